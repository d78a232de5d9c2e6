#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused MI355X env step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R] [--envs E] [--task T] [--lanes G] [--scaling weak|strong]
                    [--no-cpu-baseline] [--no-secondary]
    python bench.py --mode ppo [--task flat_terrain_backlash] [--gpus N] [--steps K] [--warmup W] [--force-split]      (BASELINE configs 3 / 4 / 5)

`--gpus N` with N > 1 works both ways: under the driver's launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`: RANK / WORLD_SIZE come from the environment and WORLD_SIZE must equal N) and as a plain `python bench.py --gpus N`, which
starts the N ranks itself as a child `torch.distributed.run` BEFORE anything touches the GPU, relays rank 0's line and returns the
children's exit code.  `--scaling weak` (default): `--envs` per GPU; `--scaling strong`: `--envs` in total, split over the ranks.

One "step" = one env step of every resident env: AutoReset/Episode wrappers + Joystick.step +
10 x mjx.step + obs/reward, all inside one HIP kernel launch (reference joystick.py:323-481).
Workload = BASELINE.json configs[1]: open_duck_mini_v2 flat_terrain, 8192 envs per GPU, random
actions a ~ U(-1,1)^14 fresh every step, observation noise off, pushes off, imitation reward on,
auto-reset on.  Inputs (state, actions) are resident in HBM before the
timed region.  Multi-GPU: one process per GPU (torchrun), envs sharded, no data-path collective
 -> weak scaling; timing = max over ranks between barriers.

Default protocol = BASELINE.md section 4: 100 warm-up env steps, then 1 000 timed env steps, five times (`--reps`); `value` is the
mean over the repetitions, every repetition bracketed by barrier + synchronize.

`--mode ppo`: one "step" = one PPO training step of the reference's hyper-parameters (rollout of 20 env steps on every env with
the policy in the loop + 128 clipped-Adam minibatch steps; with N > 1 ranks the flat gradient is all-reduced over RCCL in every
minibatch step), domain randomisation on; `value` = env steps per second INCLUDING the learner.

Prints ONE JSON line with the driver's keys plus `roofline`, `cpu_baseline` (1 GPU) and `secondary`: short full-PPO legs run AFTER
the headline's timed region and outside it -- BASELINE configs 3 and 4 on one GPU, config 5's shape (backlash model, envs sharded,
flat-gradient all-reduce over RCCL per minibatch step) on N > 1 -- each leg in a fresh CHILD process (`bench.py --mode ppo`, one per
rank) with a time limit, after the headline is measured: a leg that throws, hangs or takes the GPU runtime down costs its own entry
in `secondary`, never the headline; the line is printed and THEN the exit code is 3 if a leg failed.
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


def _mujoco_baseline(task: str, cores: int, target_seconds: float):
    """MuJoCo-C on the host cores, if the box has it -- one MjData per worker thread, env step = ctrl + 10 x
    mj_step, random actions (mj_step releases the GIL).  Needs the reference's scene XML, which does not travel to the GPU box
    unless ODK_MJCF points at a copy: returns None when either is missing."""
    try:
        import mujoco  # noqa: F401
    except Exception as e:
        return None, f"import mujoco failed ({type(e).__name__}: {e})"
    xml = os.environ.get("ODK_MJCF")   # the build ships compiled models (assets/*.npz), not the reference's XML / STL tree
    if not xml or not os.path.exists(xml):
        return None, "mujoco is importable but the scene XML is not on this box (point ODK_MJCF at scene_flat_terrain.xml)"
    import threading
    import numpy as np
    m = mujoco.MjModel.from_xml_path(xml)
    m.opt.timestep = 0.002                       # reference base.py:56
    per_thread_envs = 16

    def work(tid, nsteps, out):
        rng = np.random.default_rng(tid)
        ds = [mujoco.MjData(m) for _ in range(per_thread_envs)]
        for d in ds:
            mujoco.mj_resetDataKeyframe(m, d, 0)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            for d in ds:
                d.ctrl[:] = m.key_ctrl[0] + 0.25 * rng.uniform(-1, 1, m.nu)
                for _s in range(10):
                    mujoco.mj_step(m, d)
        out[tid] = time.perf_counter() - t0

    def run(nsteps):
        out = [0.0] * cores
        th = [threading.Thread(target=work, args=(t, nsteps, out)) for t in range(cores)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        return cores * per_thread_envs * nsteps / (time.perf_counter() - t0)

    rate = run(5)
    nsteps = max(5, int(rate * target_seconds / (cores * per_thread_envs)))
    rate = run(nsteps)
    return ({"value": round(rate, 1), "unit": "env-steps/s", "cores": cores, "kind": "mujoco-c",
             "sample": f"mujoco {mujoco.__version__} mj_step x 10 per env step on {xml}, {cores} threads x {per_thread_envs} MjData x {nsteps} env steps, "
                       "random actions; physics only (no observation / reward code)"}, "ok")


def cpu_baseline(task: str, target_seconds: float = 12.0):
    """CPU baseline beside the GPU number, on the cores the box grants this process:
      1. MuJoCo-C through `import mujoco` when the box has it (kind = "mujoco-c");
      2. otherwise the oracle's env step (own C restatement of the same step, kind = "port"), float32 build, rebuilt with
         -march=native on the box when a compiler is there (flags stated in `sample`).
    A bounded sample of the same workload: same model, same random-action protocol, fewer envs and steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    cores = usable_cores()
    mj, why = None, "not tried"
    try:
        mj, why = _mujoco_baseline(task, cores, target_seconds)
    except Exception as e:
        why = f"mujoco baseline failed ({type(e).__name__}: {e})"
    if mj is not None:
        return mj
    import subprocess
    import tempfile
    import numpy as np
    flags = "gcc -O3 (shipped build, no -march: compiled in the build container)"
    native = os.path.join(tempfile.gettempdir(), f"odk_oracle_f32_native_{os.getpid()}.so")
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", f"OUT={native}"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=120)
        os.environ["ODK_ORACLE_F32_LIB"] = native
        flags = "gcc -O3 -march=native, compiled on this box"
    except Exception:
        pass
    import oracle as O
    from open_duck_playground_amd.model import load_task_model, asset_path
    model = load_task_model(task)
    z = np.load(asset_path("prm_table.npz"))
    prm_arrays = {k: z[k] for k in z.files}
    # the float32 build of the oracle (the arithmetic type of the GPU path): ~3x the float64 checker build
    om = O.OracleModel(model.blob(), f32=True)
    prm = O.OraclePRM(prm_arrays, f32=True)
    L = O.lib(True)
    nenv = 16 * cores
    rate = L.lib.odko_rollout_mt(om.h, prm.h, nenv, 20, 5, cores, 0)          # calibration (~1 s)
    nsteps = max(20, int(rate * target_seconds / nenv))
    rate = L.lib.odko_rollout_mt(om.h, prm.h, nenv, nsteps, 10, cores, 0)
    try:
        os.remove(native)
    except OSError:
        pass
    return {"value": round(rate, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/odk_oracle*.c (C restatement of the same env step, float32, {flags}), {nenv} envs x {nsteps} steps, "
                      f"{cores} pthreads (= the cores the box grants this process: affinity mask and cgroup CPU quota; "
                      f"{os.cpu_count()} logical CPUs visible), same random-action protocol.  MuJoCo-C: {why}"}


IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"
# The pool's host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL / device-memory sharing across processes fails with
# `hipIpcGetMemHandle: invalid argument` (the environment's own note; the variable is exported on the boxes).  bench.py therefore sets 0 where the
# variable is absent -- and, because no multi-GPU box has ever run this file, never lets that guess decide a run: a launch whose ranks die before a
# result line exists is started ONCE more with the variable flipped (fresh child processes, never a re-exec), and the line says which setting worked.
RCCL_FAILURE_MARKS = ("NCCL", "RCCL", "nccl", "rccl", "hipIpc", "ProcessGroupNCCL", "DistBackendError", "unhandled system error", "invalid argument")


def ipc_flipped(env: dict) -> dict:
    """a copy of `env` with the IPC mode the other way round ("0" <-> "1"; absent counts as the runtime's default, legacy = "1")"""
    e = dict(env)
    e[IPC_VAR] = "1" if e.get(IPC_VAR, "1") == "0" else "0"
    return e


def looks_like_rccl_failure(text: str) -> bool:
    return any(m in (text or "") for m in RCCL_FAILURE_MARKS)


def _relay(cmd, env, runner=None):
    """runs one launch of the ranks, relays their stdout line by line; returns (exit code, saw a result line, stderr tail)"""
    import subprocess
    if runner is not None:          # (tests: a stand-in for the child launch)
        return runner(cmd, env)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    import threading
    err_lines = []
    t = threading.Thread(target=lambda: [(err_lines.append(l), sys.stderr.write(l)) for l in p.stderr], daemon=True)
    t.start()
    saw = False
    for line in p.stdout:
        saw = saw or line.startswith("{")
        sys.stdout.write(line); sys.stdout.flush()
    rc = p.wait()
    t.join(timeout=60)      # (the pipe closes when the last rank exits: the relay ends by itself; the limit only guards against a stray grandchild)
    return rc, saw, "".join(err_lines[-200:])


def _spawn_ranks(args, argv, runner=None) -> int:
    """Plain `python bench.py --gpus N` (no launcher around it): start the N ranks as a CHILD `torch.distributed.run` -- never an
    exec, and before this process has imported torch or touched the GPU -- with the same arguments; rank 0's JSON line is relayed as
    this command's output, the children's exit code is this command's.  A launch that dies WITHOUT a result line on what looks like an
    RCCL / IPC failure is started once more with HSA_ENABLE_IPC_MODE_LEGACY flipped (ODK_BENCH_IPC_RETRIED=1 tells the ranks to say so
    in the line)."""
    import socket

    def launch(env):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        return _relay(cmd, env, runner)

    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    env.setdefault(IPC_VAR, "0")
    rc, saw, err = launch(env)
    if rc != 0 and not saw and looks_like_rccl_failure(err):
        env2 = ipc_flipped(env)
        env2["ODK_BENCH_IPC_RETRIED"] = "1"
        sys.stderr.write(f"bench.py: the ranks died before a result line on an RCCL / IPC failure with {IPC_VAR}={env[IPC_VAR]}; one more launch with {IPC_VAR}={env2[IPC_VAR]}\n")
        rc, saw, err = launch(env2)
    return rc


def _init_dist(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree "
                         "(plain `python bench.py --gpus N` starts its own ranks)")
    # test hooks (tests/test_gpu_api.py): run the multi-rank path on a one-GPU box -- RCCL refuses two ranks per device,
    # gloo does not; the driver's runs use neither variable
    backend = os.environ.get("ODK_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("ODK_BENCH_DEVICE", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists for the env engine)")
    if backend == "nccl" and world > 1 and torch.cuda.device_count() < world and os.environ.get("ODK_BENCH_ALLOW_SHARED_DEVICE") != "1":
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} HIP devices are visible (RCCL wants one rank per device)")
    torch.cuda.set_device(local_rank)
    DIST_INFO.update(backend=None, ipc_mode_legacy=os.environ.get(IPC_VAR), ipc_retried=os.environ.get("ODK_BENCH_IPC_RETRIED") == "1")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        DIST_INFO["device"] = torch.cuda.get_device_name(local_rank)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe)                                                            # the first collective builds the communicator
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"first all-reduce returned {probe.item()} on {world} ranks")
                DIST_INFO["backend"] = "nccl"
            except Exception as e:      # noqa: BLE001 -- whatever RCCL raises here, the message is what the operator needs
                DIST_INFO["rccl_error"] = rccl_diagnostics(e, rank, local_rank)
                sys.stderr.write(DIST_INFO["rccl_error"] + "\n")
                if args.mode == "ppo":      # the gradient all-reduce IS the workload: no fallback
                    raise SystemExit(DIST_INFO["rccl_error"])
                # The headline shards independent envs: its only collectives are the barrier and the max-over-ranks of the elapsed time.  Those go
                # over gloo when RCCL cannot be brought up, and the line says so.
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:      # noqa: BLE001
                    pass
                # (a store of its own: independent of the launcher's agent store and of whatever the failed attempt left in it)
                store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ.get("MASTER_PORT", "29500")) + 57, world, is_master=(rank == 0))
                dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
                DIST_INFO["backend"] = "gloo (control plane only: RCCL did not come up)"
        else:
            dist.init_process_group(backend)
            DIST_INFO["backend"] = backend
    return rank, world, local_rank, torch.device("cuda", local_rank)


DIST_INFO = {}


def rccl_diagnostics(exc, rank: int, local_rank: int) -> str:
    """what an operator needs when RCCL does not come up: the error, this rank's device, the library version, the IPC setting"""
    import torch
    try:
        name = torch.cuda.get_device_name(local_rank)
    except Exception:      # noqa: BLE001
        name = "?"
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:      # noqa: BLE001
        ver = "?"
    return (f"bench.py rank {rank} (device {local_rank}: {name}; RCCL {ver}; {IPC_VAR}={os.environ.get(IPC_VAR)}; NCCL_DEBUG={os.environ.get('NCCL_DEBUG')}): "
            f"RCCL init / first collective failed: {type(exc).__name__}: {exc}")


def reduce_max_scalar(x: float, dev) -> float:
    """max over the ranks of a host scalar, on whatever backend the group runs (RCCL wants device tensors, gloo host tensors)"""
    import torch
    import torch.distributed as dist
    on_dev = str(DIST_INFO.get("backend", "")).startswith("nccl")
    t = torch.tensor([x], device=dev if on_dev else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def agree_any(flag: bool, dev) -> bool:
    """true on every rank as soon as one rank says so (the ranks take the same exit path / retry decision)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    return reduce_max_scalar(1.0 if flag else 0.0, dev) > 0.5


def _envs_per_rank(args, world: int) -> int:
    """weak: --envs per GPU; strong: --envs in total (8192 envs split over the ranks)."""
    if args.scaling == "weak":
        return args.envs
    if args.envs % world:
        raise SystemExit(f"bench.py --scaling strong: --envs {args.envs} is not a multiple of {world} ranks")
    return args.envs // world


def _clock_ramp(dev):
    """the GPU's clock governor needs ~0.1 s of load to leave the idle state (a short region timed right after process start-up
    runs 1.5 % slower than the steady state): 0.3 s of unrelated matrix products first; the timed steps stay exactly W + K"""
    import torch
    ramp = torch.randn(4096, 4096, device=dev)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        torch.mm(ramp, ramp)
        torch.cuda.synchronize()


COUNTER_ROUNDS = ("r6", "r5", "r4", "r3", "r2")     # newest first: the newest round that holds a task's file wins


def counter_file_name(task: str) -> str:
    """The name tools/gpu_profile_round.sh gives a task's counter summary: `traffic.json` for the headline task, `traffic_<task>.json`
    for every other one (tests/test_bench_launch.py checks that every benched task resolves to a committed file)."""
    return "traffic.json" if task == "flat_terrain" else f"traffic_{task}.json"


def _counters(task: str, envs: int):
    """Per-launch counter figures of the dominant kernel from the committed PMC passes of the SAME command (rocprofv3 --pmc in
    separate passes, profiles/README.md; re-collected by tools/gpu_profile_round.sh whenever the kernel changes): HBM bytes, the
    float32 operations behind `valu_frac` and the active-lane fraction of the VALU instructions.  Counters cannot be read from
    inside this process; the newest round that holds the task's file wins and the file is named in the output.  Other sizes than
    the 8192 envs the passes ran at: nothing (`traffic` null)."""
    if envs != 8192:
        return {}, None
    name = counter_file_name(task)
    for rnd in COUNTER_ROUNDS:
        tpath = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(tpath):
            return json.load(open(tpath)), f"profiles/{rnd}/{name}"
    return {}, None


def _learner_counters():
    """HBM bytes of ONE minibatch step of the learner (all its launches) from the committed counter passes over the full-PPO loop
    (tools/gpu_learner_traffic.sh -> profiles/rN/learner_traffic.json), or ({}, None)."""
    for rnd in COUNTER_ROUNDS:
        tpath = os.path.join(ROOT, "profiles", rnd, "learner_traffic.json")
        if os.path.exists(tpath):
            return json.load(open(tpath)), f"profiles/{rnd}/learner_traffic.json"
    return {}, None


POLICY_DIMS, VALUE_DIMS = (101, 512, 256, 128, 28), (212, 512, 256, 128, 1)


def learner_flops(mb: int, nets=(POLICY_DIMS, VALUE_DIMS)) -> dict:
    """Matrix work of ONE minibatch step on `mb` samples, as the kernels do it: forward of every layer, backward-data of every layer
    BUT THE FIRST (nobody needs the gradient w.r.t. the observations: include/odk.h, odk_mlp_backward), weight gradients of every
    layer.  Reference sizes, mb = 5120: 5.03 + 3.39 + 5.03 = 13.46 GFLOP.  (Not counted: the value network's 256 bootstrap rows.)"""
    fwd = sum(2.0 * mb * a * b for d in nets for a, b in zip(d[:-1], d[1:]))
    bwd = sum(2.0 * mb * a * b for d in nets for a, b in zip(d[1:-1], d[2:]))
    return {"fwd": fwd, "bwd": bwd, "dw": fwd, "total": fwd + bwd + fwd}


def _matrix_launch_us(learner, reps: int = 50):
    """Each of the learner's three matrix launches alone on its own buffers (HIP events on torch's current stream, which is the stream
    the launches go to), microseconds per launch; `dw` includes its finishing launch.  None without the fused path."""
    import torch
    if learner is None or getattr(learner, "fused", None) is None:
        return None
    out = {}
    for name, fn in (("fwd", learner.fused.forward), ("bwd", learner.fused.backward), ("dw", learner.dw_all)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        out[name] = 1e3 * e0.elapsed_time(e1) / reps
    return out


def _split_learner_ms(net, opt, data, cfg, gen, steps: int):
    """What the data-parallel form of a minibatch step costs WITHOUT a second GPU (`--force-split`): a one-rank RCCL group and
    (a) the single-GPU step (one graph, for reference: timed the same way), (b) graph A -> host-issued all-reduce -> graph B with the
    gradient norm in its own launch (what N > 1 ranks run), (c) the same step with the all-reduce CAPTURED inside one graph (one
    replay per step again).  Learner milliseconds per training step of each (128 minibatch steps + their preparation)."""
    import torch
    import torch.distributed as dist
    from open_duck_playground_amd.ppo import train as T
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    res = {}
    for key, kw in (("learner_ms_one_graph", dict(split_update=False)), ("learner_ms_split", dict(split_update=True)),
                    ("learner_ms_allreduce_captured", dict(split_update=True, capture_allreduce=True))):
        try:
            lr = T.make_learner(net, data, cfg, 1, dist.group.WORLD, **kw)
            for _ in range(2):
                T.sgd_epoch(net, opt, data, cfg, gen, 1, dist.group.WORLD, learner=lr, meter=T.LossMeter())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                T.sgd_epoch(net, opt, data, cfg, gen, 1, dist.group.WORLD, learner=lr, meter=T.LossMeter())
            e1.record(); e1.synchronize()
            res[key] = round(e0.elapsed_time(e1) / steps, 3)
            lr.close()
        except Exception as e:
            res[key] = f"failed: {type(e).__name__}: {e}"
    dist.destroy_process_group()
    return res


def ppo_leg(ctx, task: str, envs: int, steps: int, warmup: int, scaling: str = "weak", force_split: bool = False, allreduce_form: str = "auto", probe: bool = False):
    """K full PPO training steps after W warm-up ones on this rank's `envs` envs (BASELINE configs 3 / 4 on 1 GPU, 5 on N: envs
    sharded, flat gradient all-reduced over RCCL in each of the 128 minibatch steps).  Returns the result line on rank 0."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local_rank, dev = ctx
    from open_duck_playground_amd import joystick
    from open_duck_playground_amd.ppo import train as T
    from open_duck_playground_amd.ppo.networks import PPONetworks
    env = joystick.Joystick(task=task, num_envs=envs, device=local_rank, env_id_offset=rank * envs)
    env.randomize(np.random.default_rng([0, rank, 0]))          # randomize.py domain randomisation, own draws per rank
    cfg = T.ppo_config()
    grp = dist.group.WORLD if world > 1 else None
    torch.manual_seed(0)                                        # identical initial parameters on every rank
    net = PPONetworks(env.observation_size["state"][0], env.observation_size["privileged_state"][0], env.action_size).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=cfg["learning_rate"], capturable=True)
    gen = torch.Generator(device=dev); gen.manual_seed(1000 + rank)
    state = env.reset(rank)
    learner = None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def training_step(rec=None):
        nonlocal state, learner
        if rec: rec[0].record()
        data, state = T.rollout(env, net, state, cfg["unroll_length"], gen)
        if rec: rec[1].record()
        net.norm_obs.update(data["obs"], grp); net.norm_priv.update(data["priv"], grp)
        if learner is None:
            kw = {} if (allreduce_form == "auto" or world == 1) else dict(capture_allreduce=(allreduce_form == "captured"))
            learner = T.make_learner(net, data, cfg, world, grp, **kw)
        T.sgd_epoch(net, opt, data, cfg, gen, world, grp, learner=learner, meter=T.LossMeter())
        if rec: rec[2].record()
        return data

    _clock_ramp(dev)
    if probe:      # two training steps in the requested form, then the replicas' parameters bit for bit (raises on a mismatch)
        for _ in range(2):
            training_step()
        barrier()
        if world > 1:
            T.assert_replicas_identical(net, grp)
        form = "captured" if (learner is not None and getattr(learner, "capture_allreduce", False)) else "split"
        if learner is not None:
            learner.close()
        env.close() if hasattr(env, "close") else None
        return {"probe": "ok", "allreduce_form": form, "n_gpus": world} if rank == 0 else None
    for _ in range(max(warmup, 1)):      # (the first step builds the learner and captures its graphs: never timed)
        training_step()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        data = training_step(ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        T.assert_replicas_identical(net, grp)
    rollout_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
    learner_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
    allreduce_ms = None
    nsgd = cfg["num_minibatches"] * cfg["num_updates_per_batch"]
    if world > 1:   # the collective alone, outside the timed region: 128 all-reduces of a gradient-sized buffer
        n_par = sum(p.numel() for p in net.parameters() if p.requires_grad)
        g = torch.zeros(n_par, device=dev)
        for _ in range(8): dist.all_reduce(g)
        torch.cuda.synchronize(); ta = time.perf_counter()
        for _ in range(nsgd): dist.all_reduce(g)
        torch.cuda.synchronize()
        allreduce_ms = 1e3 * (time.perf_counter() - ta)
    env_steps = world * envs * cfg["unroll_length"] * steps
    launch_us = _matrix_launch_us(learner) if rank == 0 else None
    split = _split_learner_ms(net, opt, data, cfg, gen, steps) if (force_split and world == 1) else None
    out = None
    if rank == 0:
        # learner roofline (the training step's dominant part): f32 matrix-core work of the three whole-network kernels, counted as
        # the kernels do it (no input gradient of the first layer)
        mb = envs * cfg["unroll_length"] // cfg["num_minibatches"]
        fl = learner_flops(mb, ((env.observation_size["state"][0], 512, 256, 128, 2 * env.action_size), (env.observation_size["privileged_state"][0], 512, 256, 128, 1)))
        achieved = nsgd * fl["total"] / (learner_ms * 1e-3) / 1e12 if learner_ms > 0 else 0.0
        cnt, csrc = _learner_counters() if (envs == 8192 and world == 1) else ({}, None)
        per_launch = None
        if launch_us:
            per_launch = {k: {"us": round(launch_us[k], 2), "gflop": round(fl[k] / 1e9, 3), "tflops": round(fl[k] / (launch_us[k] * 1e-6) / 1e12, 1),
                              "frac": round(fl[k] / (launch_us[k] * 1e-6) / 1e12 / VALU_PEAK_TFLOPS, 3)} for k in ("fwd", "bwd", "dw")}
        out = {
            "metric": METRIC, "value": round(env_steps / elapsed, 1), "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": max(warmup, 1),
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"full PPO (BASELINE config {'5' if world > 1 else ('4' if 'rough' in task else '3')}): open_duck_mini_v2 {task} + randomize.py domain "
                                   f"randomisation, {envs} envs/GPU, one step = rollout of {cfg['unroll_length']} env steps with the policy in the loop + {nsgd} "
                                   f"clipped-Adam minibatch steps of {mb} samples (reference hyper-parameters, common/runner.py:86-118); value counts env steps "
                                   "per second INCLUDING the learner; noise, pushes, imitation reward, auto-reset on",
                       "mode": "ppo", "task": task, "envs_per_gpu": envs, "global_envs": envs * world, "unroll_length": cfg["unroll_length"],
                       "sgd_steps_per_training_step": nsgd, "parallelism": f"env-sharded x{world}" + (", flat-gradient all-reduce (RCCL) per minibatch step" if world > 1 else ", no collective"),
                       "rollout_ms_per_training_step": round(rollout_ms, 3), "learner_ms_per_training_step": round(learner_ms, 3),
                       "allreduce_ms_per_training_step_isolated": None if allreduce_ms is None else round(allreduce_ms, 3),
                       "learner_path": "fused whole-network kernels" if (learner is not None and getattr(learner, "fused", None) is not None) else "library GEMMs / autograd",
                       "allreduce_form": None if world == 1 else ("captured" if (learner is not None and getattr(learner, "capture_allreduce", False)) else "split"),
                       "reward_per_step_last_rollout": round(float(data["reward"].mean()), 4)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / VALU_PEAK_TFLOPS, 4),
                         "traffic": cnt.get("hbm_bytes_per_sgd_step"), "counter_source": csrc,
                         "kernel": "mlp_fwd_kernel + mlp_bwd_kernel + dw_gemm_kernel (learner, f32 matrix cores)",
                         "gflop_per_sgd_step": round(fl["total"] / 1e9, 3), "matrix_launches": per_launch,
                         "note": "achieved = matrix FLOPs of the 128 minibatch steps as the kernels do them (forward + backward-data without the first layer + weight "
                                 "gradients) / learner time (HIP events), i.e. INCLUDING the element-wise launches between them; matrix_launches = each of the three "
                                 "matrix launches alone (HIP events around 50 launches on the learner's own buffers, after the timed region); traffic = HBM bytes of one "
                                 "minibatch step from the committed counter passes; peak = dense f32 MFMA"},
        }
        if split is not None:
            out["config"].update(split)
    if learner is not None:
        learner.close()
    env.close() if hasattr(env, "close") else None
    return out


def physics_leg(ctx, args, envs: int):
    """The headline: W warm-up + K timed env steps (x repetitions) of the fused env-step launch on this rank's `envs` envs."""
    import torch
    import torch.distributed as dist
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    rank, world, local_rank, dev = ctx

    model = load_task_model(args.task)
    if getattr(args, "cone", None) == "elliptic":      # (not the BASELINE workload: the duck's model with <option cone="elliptic">, the kernels' cone instantiations)
        import numpy as np
        from open_duck_playground_amd.model import Model
        model = Model({**model.a, "opt_cone": np.array([1], np.int32)})
    cfg = engine.default_config()
    cfg.noise_level = 0.0
    cfg.push_enable = 0.0
    cfg.lanes_per_env = args.lanes
    batch = engine.Batch(model, envs, cfg, device=local_rank)
    batch.reset(seed=0, env_id_offset=rank * envs)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # fresh action per step, generated before the timed region (HBM-resident inputs)
    chunk = min(args.steps + args.warmup, 256)
    actions = torch.empty(chunk, envs, 14, device=dev, dtype=torch.float32).uniform_(-1.0, 1.0, generator=gen)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _clock_ramp(dev)
    for i in range(args.warmup):
        batch.step(actions[i % chunk])
    barrier()
    # HIP events around a SAMPLE of the timed launches (every 4th, every 16th in long runs): an event pair serialises the stream
    # for ~7 us, 1.2 % of this step when every launch carries one
    timing_stride = 4 if args.steps <= 64 else 16
    batch.timing(timing_stride)
    times = []
    k = args.warmup
    for rep in range(args.reps):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            batch.step(actions[(k + i) % chunk])
        barrier()
        el = time.perf_counter() - t0
        k += args.steps
        if world > 1:
            el = reduce_max_scalar(el, dev)
        times.append(el)
    elapsed = sum(times) / len(times)
    kernel_ms, launches = batch.timing(False)
    done_frac = float(batch.done.mean().item())
    value = world * envs * args.steps / elapsed
    lanes = batch.cfg.lanes_per_env or 32
    batch.close()
    if rank != 0:
        return None
    cnt, csrc = _counters(args.task, envs) if getattr(args, "cone", None) != "elliptic" else ({}, None)      # (the committed counter passes are the pyramid kernels')
    traffic, flop_launch = cnt.get("hbm_bytes_per_launch"), cnt.get("valu_flop_per_launch")
    bytes_per_launch = BYTES_PER_ENV_STEP.get(args.task, 3564 if "backlash" in args.task else 2844) * envs
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    flop = flop_launch if flop_launch else envs * FLOP_PER_ENV_STEP
    valu = flop / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "counter_source": csrc,
            "kernel": "step_kernel", "kernel_ms": round(kernel_ms, 4), "launches_timed": launches, "timed_every": timing_stride,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "note": "fused env step is not HBM-bound (SURVEY.md 0.4); secondary roof = FP32 VALU",
            "valu_achieved_tflops": round(valu, 3), "valu_peak_tflops": VALU_PEAK_TFLOPS,
            "valu_frac": round(valu / VALU_PEAK_TFLOPS, 5),
            "valu_flop_source": "SQ_INSTS_VALU_* counters (" + csrc + "): every lane of an issued instruction counted" if flop_launch
                                else "1.2 MFLOP per env step (SURVEY.md 8d estimate)"}
    if cnt.get("valu_active_lane_fraction"):
        # active-lane view of the same figure: SQ_ACTIVE_INST_VALU-weighted thread cycles / (64 x instruction cycles)
        roof["valu_active_lane_fraction"] = cnt["valu_active_lane_fraction"]
        roof["valu_frac_active_lanes"] = round(valu / VALU_PEAK_TFLOPS * cnt["valu_active_lane_fraction"], 5)
    return {
        "metric": METRIC, "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"open_duck_mini_v2 {args.task}, {envs} envs/GPU, random-action rollout "
                               "(wrappers + Joystick.step + 10 x mjx.step + obs/reward fused in one launch), "
                               "noise off, pushes off, imitation on, auto-reset on" + (", ELLIPTIC friction cones (not the BASELINE workload)" if getattr(args, "cone", None) == "elliptic" else ""),
                   "mode": "physics", "envs_per_gpu": envs, "global_envs": envs * world, "n_substeps": 10,
                   "lanes_per_env": lanes, "parallelism": f"env-sharded x{world}, no collective",
                   "done_fraction_last_step": round(done_frac, 4),
                   "repetitions": args.reps, "value_per_repetition": [round(world * envs * args.steps / t, 1) for t in times],
                   "protocol": "BASELINE.md 4: W warm-up env steps, then K timed env steps x repetitions, value = K / mean time"},
        "roofline": roof,
    }


def _secondary_child(ctx, task: str, envs_arg: int, scaling: str, steps: int, timeout: float, leg_index: int, extra_argv=(), env_over=None, tag="leg"):
    """One full-PPO leg in a FRESH CHILD PROCESS per rank (`bench.py --mode ppo ...`, started with subprocess -- never an exec -- after
    this process has measured the headline): a hard fault in a leg (a GPU memory fault aborting the HSA runtime, a SIGSEGV in the
    library, a stuck collective) ends the child, not the process that holds the headline.  Under N > 1 ranks every rank starts its
    own child with the launcher's RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous port of its own.  Returns (result line or None,
    error text or None, path of the child's full stderr); a child past `timeout` is killed by its exact PID."""
    import subprocess
    import tempfile
    rank, world = ctx[0], ctx[1]
    env = dict(os.environ)
    env.update(env_over or {})
    if world > 1:
        env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 101 + leg_index)
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the child ranks rendezvous among themselves: rank 0's child hosts the store on the new port
        env.setdefault("NCCL_DEBUG", "WARN")               # RCCL's own words on a failure end up in the stderr file
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "ppo", "--task", task, "--gpus", str(world), "--envs", str(envs_arg), "--scaling", scaling,
           "--steps", str(steps), "--warmup", "4"] + list(extra_argv)
    log_dir = os.environ.get("ODK_BENCH_LOG_DIR") or os.path.join(tempfile.gettempdir(), "odk_bench_logs")
    os.makedirs(log_dir, exist_ok=True)
    err_path = os.path.join(log_dir, f"secondary_{task}_{tag}_rank{rank}.stderr")
    try:
        with open(err_path, "w") as ef:
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=ef, text=True)
            try:
                so, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                so, _ = p.communicate()
                return None, f"leg still running after {timeout:.0f} s: killed (stderr: {err_path})", err_path
    except OSError as e:
        return None, f"could not start the leg: {e}", err_path
    if p.returncode != 0:
        se = open(err_path, errors="replace").read()
        tail = " | ".join([l for l in se.strip().splitlines() if l.strip()][-6:])
        return None, f"leg exited with code {p.returncode}: {tail[-1200:]} (full stderr: {err_path})", err_path
    if rank != 0:
        return None, None, err_path
    lines = [l for l in so.splitlines() if l.startswith("{")]
    if not lines:
        return None, f"leg printed no result line (stderr: {err_path})", err_path
    try:
        return json.loads(lines[-1]), None, err_path
    except ValueError as e:
        return None, f"leg's result line does not parse: {e}", err_path


def choose_allreduce_form(probe_ok_everywhere: bool) -> str:
    """the data-parallel form of the timed PPO steps: the all-reduce captured inside the step graph (one replay per 32 steps: +1.3 % over a single
    GPU's step on a one-rank group) when a probe on the real ranks kept the replicas bit-identical, else graph A -> host-issued all-reduce -> graph B
    (+8.6 %)"""
    return "captured" if probe_ok_everywhere else "split"


def run_secondary_leg(ctx, dev, task, envs_arg, scaling, steps, budget, leg_index, child=None):
    """One `secondary` entry under N >= 1 ranks.  N > 1: (1) a PROBE child per rank -- two training steps with the all-reduce captured, replicas
    compared bit for bit; the ranks agree on its outcome; (2) the timed leg in the form `choose_allreduce_form` names; (3) if a child died on what looks
    like an RCCL / IPC failure, ONE more attempt with HSA_ENABLE_IPC_MODE_LEGACY flipped (all ranks take that decision together).  Returns (entry for
    `secondary` on rank 0 -- or None on the other ranks --, failed flag agreed by all ranks)."""
    child = child or _secondary_child
    rank, world = ctx[0], ctx[1]
    notes = {}
    env_over = {}
    form_argv = []
    t_probe = min(120.0, budget * 0.3)
    if world > 1:
        for attempt in range(2):
            _, perr, ppath = child(ctx, task, envs_arg, scaling, 2, t_probe, 20 + leg_index + 40 * attempt, extra_argv=["--allreduce-form", "captured", "--probe"],
                                   env_over=env_over, tag=f"probe{attempt}")
            bad = agree_any(perr is not None, dev)
            text = (perr or "") + (open(ppath, errors="replace").read()[-4000:] if (perr is not None and os.path.exists(ppath)) else "")
            rccl = agree_any(perr is not None and looks_like_rccl_failure(text), dev)
            notes[f"allreduce_probe{'_retry' if attempt else ''}"] = "ok" if not bad else (perr or "failed on another rank")
            if not bad or not rccl or attempt == 1:
                break
            env_over = {IPC_VAR: ipc_flipped(dict(os.environ, **env_over))[IPC_VAR]}      # every rank flips together
            notes["ipc_mode_legacy_flipped_to"] = env_over[IPC_VAR]
        if bad and env_over:      # the flipped setting did not help either: the timed leg runs with the environment as it was given
            env_over = {}
        form = choose_allreduce_form(not bad)
        form_argv = ["--allreduce-form", form]
        notes["allreduce_form"] = form
    leg, err, path = child(ctx, task, envs_arg, scaling, steps, budget - (t_probe if world > 1 else 0.0), leg_index, extra_argv=form_argv, env_over=env_over, tag="timed")
    failed = agree_any(err is not None, dev)
    if world > 1:
        notes["ipc_mode_legacy"] = dict(os.environ, **env_over).get(IPC_VAR)
    if rank != 0:
        return None, failed
    if err is not None or leg is None:
        return dict({"task": task, "mode": "ppo", "error": err or "a peer rank's leg failed", "stderr_file": path}, **notes), failed
    return dict(_brief(leg), **notes), failed


def _brief(leg: dict) -> dict:
    c = leg["config"]
    return {"config": c["workload"].split(":")[0].replace("full PPO (", "").rstrip(")"), "task": c["task"], "mode": "ppo", "value": leg["value"],
            "unit": leg["unit"], "n_gpus": leg["n_gpus"], "steps": leg["steps"], "warmup": leg["warmup"], "ms_per_step": leg["ms_per_step"],
            "envs_per_gpu": c["envs_per_gpu"], "rollout_ms": c["rollout_ms_per_training_step"], "learner_ms": c["learner_ms_per_training_step"],
            "allreduce_ms_isolated": c["allreduce_ms_per_training_step_isolated"], "learner_path": c["learner_path"],
            "roofline": {k: leg["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "counter_source", "gflop_per_sgd_step", "matrix_launches")}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="physics", choices=["physics", "ppo"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="physics: env steps per repetition (default 1000); ppo: training steps (default 10)")
    ap.add_argument("--warmup", type=int, default=None, help="default 100 (physics) / 3 (ppo)")
    ap.add_argument("--reps", type=int, default=None, help="physics: repetitions of the K timed steps, value = mean (default 5; 1 when --steps is given)")
    ap.add_argument("--envs", type=int, default=8192, help="envs per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--task", default=None)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--cone", default=None, choices=["pyramidal", "elliptic"], help="physics mode: the contact solver's friction cone (default: the model's own = the BASELINE workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="physics mode: skip the short full-PPO legs attached as `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--secondary-timeout", type=float, default=300.0, help="seconds for ALL secondary legs together (each child gets an equal share)")
    ap.add_argument("--allreduce-form", default="auto", choices=["auto", "split", "captured"],
                    help="ppo mode, N > 1 ranks: the gradient all-reduce between graph A and graph B issued from the host (split), or captured inside the step graph "
                         "(captured: one replay per 32 minibatch steps); auto = $ODK_LEARNER_CAPTURE_ALLREDUCE, else split.  The headline's secondary leg probes "
                         "`captured` in a child first and falls back to `split`")
    ap.add_argument("--probe", action="store_true", help="ppo mode: two training steps, replicas compared bit for bit, one line {\"probe\": \"ok\"}; no timing")
    ap.add_argument("--force-split", action="store_true", help="ppo mode, 1 GPU: also time the data-parallel form of the minibatch step on a one-rank RCCL group "
                    "(graph A -> all-reduce -> graph B, and the all-reduce captured in one graph)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args, sys.argv[1:]))          # nothing has touched the GPU yet in this process

    import torch.distributed as dist
    ctx = _init_dist(args)
    rank, world = ctx[0], ctx[1]
    envs = _envs_per_rank(args, world)

    if args.mode == "ppo":
        out = ppo_leg(ctx, args.task or "flat_terrain_backlash", envs, args.steps if args.steps is not None else 10,
                      args.warmup if args.warmup is not None else 3, args.scaling, args.force_split, allreduce_form=args.allreduce_form, probe=args.probe)
        if rank == 0:
            print(json.dumps(out), flush=True)
    else:
        args.task = args.task or "flat_terrain"
        if args.reps is None:
            args.reps = 5 if args.steps is None else 1       # an explicit --steps K is timed once: exactly K steps
        args.steps = 1000 if args.steps is None else args.steps
        args.warmup = 100 if args.warmup is None else args.warmup
        out = physics_leg(ctx, args, envs)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": usable_cores(), "kind": "port", "sample": f"failed: {e}"}
        failed = 0
        if not args.no_secondary:
            # BASELINE configs 3 / 4 (1 GPU) or 5's shape (N GPUs) as short full-PPO legs, outside the headline's timed region, each in
            # a child process of its own; the headline is already measured and is printed whatever the legs do
            tasks = ["flat_terrain_backlash", "rough_terrain_backlash"] if world == 1 else ["flat_terrain_backlash"]
            if world > 1:
                dist.barrier()          # every rank has finished the headline before any rank's child claims its GPU
            sec = []
            for k, task in enumerate(tasks):
                entry, bad = run_secondary_leg(ctx, ctx[3], task, args.envs, args.scaling, args.secondary_steps, args.secondary_timeout / len(tasks), k)
                failed += int(bad)          # (agreed across the ranks: every rank takes the same exit path below)
                if entry is not None:
                    sec.append(entry)
            if rank == 0:
                out["secondary"] = sec
        if rank == 0 and world > 1:
            out["config"]["launch"] = {k: v for k, v in DIST_INFO.items() if v is not None}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if failed:
            # the headline went out; a failed / abandoned leg still shows in the exit code (and in `secondary`).  `failed` is the same on every
            # rank (run_secondary_leg agrees on it over the live group), so every rank leaves this way together.
            sys.stdout.flush()
            os._exit(3)        # (a rank whose peers' children are stuck cannot join them through destroy_process_group)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
