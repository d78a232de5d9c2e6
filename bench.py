#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused MI355X env step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--task T] [--lanes G] [--no-cpu-baseline]

One "step" = one env step of every resident env: AutoReset/Episode wrappers + Joystick.step +
10 x mjx.step + obs/reward, all inside one HIP kernel launch (reference joystick.py:323-481).
Workload = BASELINE.json configs[1]: open_duck_mini_v2 flat_terrain, 8192 envs per GPU, random
actions a ~ U(-1,1)^14 fresh every step, observation noise off, pushes off, imitation reward on,
auto-reset on.  Inputs (state, actions) are resident in HBM before the
timed region.  Multi-GPU: one process per GPU (torchrun), envs sharded, no data-path collective
 -> weak scaling; timing = max over ranks between barriers.

Prints ONE JSON line with the driver's keys plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "env-steps/sec open_duck_mini_v2 flat_terrain @8192 envs; 1/2/4/8-GPU scaling"
BYTES_PER_ENV_STEP = {"flat_terrain": 2844, "flat_terrain_backlash": 3564}  # SURVEY.md 8(d), algorithmic HBM bytes
FLOP_PER_ENV_STEP = 1.2e6                                                   # SURVEY.md 8(d), VALU work estimate
HBM_PEAK_GBS = 8000.0                                                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3


def usable_cores() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a box may show 256 logical
    CPUs and grant 16 cores' worth of time: threads beyond that only add throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(task: str, target_seconds: float = 12.0):
    """Times the oracle's env step (the CPU restatement, kind='port') on all host cores of this box,
    on a bounded sample of the same workload: same model, same protocol, fewer envs and steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    from open_duck_playground_amd.model import load_task_model, asset_path
    model = load_task_model(task)
    z = np.load(asset_path("prm_table.npz"))
    prm_arrays = {k: z[k] for k in z.files}
    # the float32 build of the oracle (-O3; the arithmetic type of the GPU path): ~3x the float64 checker build
    om = O.OracleModel(model.blob(), f32=True)
    prm = O.OraclePRM(prm_arrays, f32=True)
    L = O.lib(True)
    cores = usable_cores()
    nenv = 16 * cores
    rate = L.lib.odko_rollout_mt(om.h, prm.h, nenv, 20, 5, cores, 0)          # calibration (~1 s)
    nsteps = max(20, int(rate * target_seconds / nenv))
    rate = L.lib.odko_rollout_mt(om.h, prm.h, nenv, nsteps, 10, cores, 0)
    return {"value": round(rate, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/odk_oracle*.c (C restatement of the same env step, float32 -O3 build), {nenv} envs x {nsteps} steps, "
                      f"{cores} pthreads (= the cores the box grants this process: affinity mask and cgroup CPU quota; "
                      f"{os.cpu_count()} logical CPUs visible), same random-action protocol"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=8192, help="envs per GPU")
    ap.add_argument("--task", default="flat_terrain")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_api.py): run the multi-rank path on a one-GPU box -- RCCL refuses two ranks per device,
    # gloo does not; the driver's runs use neither variable
    backend = os.environ.get("ODK_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("ODK_BENCH_DEVICE", local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists for the env engine)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    model = load_task_model(args.task)
    cfg = engine.default_config()
    cfg.noise_level = 0.0
    cfg.push_enable = 0.0
    cfg.lanes_per_env = args.lanes
    batch = engine.Batch(model, args.envs, cfg, device=local_rank)
    batch.reset(seed=0, env_id_offset=rank * args.envs)
    total = args.steps + args.warmup
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # fresh action per step, generated before the timed region (HBM-resident inputs)
    chunk = min(total, 256)
    actions = torch.empty(chunk, args.envs, 14, device=dev, dtype=torch.float32).uniform_(-1.0, 1.0, generator=gen)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU's clock governor needs ~0.1 s of load to leave the idle state (a 20-step region timed right after process start-up
    # runs 1.5 % slower than the steady state): 0.3 s of unrelated matrix products first; the env steps stay exactly W + K
    ramp = torch.randn(4096, 4096, device=dev)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        torch.mm(ramp, ramp)
        torch.cuda.synchronize()
    del ramp
    for i in range(args.warmup):
        batch.step(actions[i % chunk])
    barrier()
    # HIP events around a SAMPLE of the timed launches (every 4th, every 16th in long runs): an event pair serialises the stream
    # for ~7 us, 1.2 % of this step when every launch carries one
    timing_stride = 4 if args.steps <= 64 else 16
    batch.timing(timing_stride)
    t0 = time.perf_counter()
    for i in range(args.steps):
        batch.step(actions[(args.warmup + i) % chunk])
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = batch.timing(False)
    done_frac = float(batch.done.mean().item())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * args.envs * args.steps / elapsed

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r2", "traffic.json")
        if args.task == "flat_terrain" and args.envs == 8192 and os.path.exists(tpath):
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, profiles/README.md)
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        bytes_per_launch = BYTES_PER_ENV_STEP.get(args.task, 2844) * args.envs
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        valu = args.envs * FLOP_PER_ENV_STEP / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"open_duck_mini_v2 {args.task}, {args.envs} envs/GPU, random-action rollout "
                                   "(wrappers + Joystick.step + 10 x mjx.step + obs/reward fused in one launch), "
                                   "noise off, pushes off, imitation on, auto-reset on",
                       "envs_per_gpu": args.envs, "global_envs": args.envs * world, "n_substeps": 10,
                       "lanes_per_env": batch.cfg.lanes_per_env or 32, "parallelism": f"env-sharded x{world}, no collective",
                       "done_fraction_last_step": round(done_frac, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "step_kernel", "kernel_ms": round(kernel_ms, 4), "launches_timed": launches, "timed_every": timing_stride,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "fused env step is not HBM-bound (SURVEY.md 0.4); secondary roof = FP32 VALU",
                         "valu_achieved_tflops": round(valu, 3), "valu_peak_tflops": VALU_PEAK_TFLOPS,
                         "valu_frac": round(valu / VALU_PEAK_TFLOPS, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": usable_cores(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
