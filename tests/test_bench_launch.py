"""bench.py's launch logic, as far as it runs without a GPU: the rank count it is given and the rank count it finds must agree,
and a plain `python bench.py --gpus N` starts N ranks of its own (BASELINE.json's metric is "...; 1/2/4/8-GPU scaling", and the
driver's command is the plain one).  The GPU side of the same paths: tests/test_gpu_api.py (test_bench_*)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=e, timeout=300, cwd=ROOT)


def test_gpus_flag_must_match_the_launcher():
    """Under a launcher (WORLD_SIZE set) a --gpus that disagrees is an error, not a silently smaller run."""
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0
    assert "--gpus 1 but WORLD_SIZE=2" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_plain_command_starts_n_ranks():
    """No launcher, --gpus 2: two child ranks are started (each reports the missing HIP device here -- there is no CPU path), their
    failure is this command's exit code and no JSON line is printed."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a HIP device") >= 2, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_no_gpu_no_number():
    out = _run(["--steps", "1", "--warmup", "0"])
    assert out.returncode != 0 and "needs a HIP device" in out.stderr


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("odk_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_benched_task_finds_its_counter_file():
    """bench.py names a task's counter summary the way tools/gpu_profile_round.sh writes it, and a committed file exists for each of
    the three tasks (a lookup that silently misses reports `traffic: null` and falls back to the FLOP estimate)."""
    b = _bench_module()
    for task in ("flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"):
        cnt, src = b._counters(task, 8192)
        assert src is not None and os.path.exists(os.path.join(ROOT, src)), task
        assert src.endswith(b.counter_file_name(task))
        assert cnt.get("hbm_bytes_per_launch", 0) > 1e7 and cnt.get("valu_flop_per_launch", 0) > 1e9, task
    assert b._counters("flat_terrain", 4096) == ({}, None)
    script = open(os.path.join(ROOT, "tools", "gpu_profile_round.sh")).read()
    assert '[ $TASK != flat_terrain ] && SFX="_$TASK"' in script and "$OUT/traffic$SFX.json" in script       # the naming rule the lookup mirrors


def test_learner_flop_count_is_the_work_the_kernels_do():
    """13.46 GFLOP per minibatch step at the reference sizes: forward 5.03 + backward-data WITHOUT the first layer 3.39 + weight
    gradients 5.03 (include/odk.h: odk_mlp_backward never forms the gradient w.r.t. the network input)."""
    b = _bench_module()
    fl = b.learner_flops(5120)
    pol, val = 101 * 512 + 512 * 256 + 256 * 128 + 128 * 28, 212 * 512 + 512 * 256 + 256 * 128 + 128 * 1
    assert fl["fwd"] == fl["dw"] == 2.0 * 5120 * (pol + val)
    assert fl["bwd"] == 2.0 * 5120 * (pol - 101 * 512 + val - 212 * 512)
    assert abs(fl["total"] / 1e9 - 13.46) < 0.01


# ---- the driver's first multi-GPU run must not fail for avoidable reasons: control flow, without a GPU
class _Args:
    gpus = 2


def test_launcher_retries_once_with_the_ipc_mode_flipped(monkeypatch):
    """A launch whose ranks die before a result line on an RCCL / IPC failure is started ONCE more as fresh children with
    HSA_ENABLE_IPC_MODE_LEGACY flipped; any other failure, or a failure after the line went out, is returned as it is."""
    b = _bench_module()
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    calls = []

    def runner(script):
        def run(cmd, env):
            calls.append(dict(env))
            assert "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd
            return script[len(calls) - 1]
        return run
    # RCCL failure, then success with the flipped variable
    calls.clear()
    rc = b._spawn_ranks(_Args, ["--gpus", "2"], runner=runner([(1, False, "RuntimeError: NCCL error: unhandled system error (hipIpcGetMemHandle: invalid argument)"), (0, True, "")]))
    assert rc == 0 and len(calls) == 2
    assert calls[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and calls[1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and calls[1]["ODK_BENCH_IPC_RETRIED"] == "1"
    assert "ODK_BENCH_IPC_RETRIED" not in calls[0]
    # a failure that is not RCCL's: no second launch
    calls.clear()
    assert b._spawn_ranks(_Args, [], runner=runner([(2, False, "bench.py needs a HIP device")])) == 2 and len(calls) == 1
    # a failure after the headline went out (exit code 3 of a failed secondary leg): no second launch
    calls.clear()
    assert b._spawn_ranks(_Args, [], runner=runner([(3, True, "NCCL error in a leg")])) == 3 and len(calls) == 1
    # the second launch's result is final, whatever it is
    calls.clear()
    assert b._spawn_ranks(_Args, [], runner=runner([(1, False, "ncclSystemError"), (1, False, "ncclSystemError")])) == 1 and len(calls) == 2
    # an exported setting is respected on the first launch and flipped on the second
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    calls.clear()
    b._spawn_ranks(_Args, [], runner=runner([(1, False, "RCCL"), (0, True, "")]))
    assert calls[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and calls[1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_secondary_leg_probes_the_captured_allreduce_and_falls_back(tmp_path):
    """N > 1 ranks: a probe child (two steps with the all-reduce captured, replicas compared) decides the form of the timed leg; an RCCL failure of
    the probe gets one more attempt with the IPC mode flipped; the entry says which form and which setting ran; a failed timed leg carries the path of
    its full stderr."""
    b = _bench_module()
    leg = {"value": 1.0, "unit": "env-steps/s", "n_gpus": 2, "steps": 3, "warmup": 4, "ms_per_step": 1.0,
           "config": {"workload": "full PPO (BASELINE config 5): x", "task": "flat_terrain_backlash", "envs_per_gpu": 8192, "rollout_ms_per_training_step": 1.0,
                      "learner_ms_per_training_step": 2.0, "allreduce_ms_per_training_step_isolated": 3.0, "learner_path": "fused"},
           "roofline": {"bound": "mfma"}}
    ctx = (0, 2, 0, None)
    log = []

    def child_of(script):
        def child(ctx, task, envs, scaling, steps, timeout, leg_index, extra_argv=(), env_over=None, tag="leg"):
            log.append((tag, list(extra_argv), dict(env_over or {})))
            r = script[len(log) - 1]
            p = tmp_path / f"{tag}.stderr"
            p.write_text(r[2] if len(r) > 2 else "")
            return r[0], r[1], str(p)
        return child
    # probe fine -> captured
    log.clear()
    e, failed = b.run_secondary_leg(ctx, None, "flat_terrain_backlash", 8192, "weak", 3, 300.0, 0, child=child_of([(None, None), (leg, None)]))
    assert not failed and e["allreduce_form"] == "captured" and e["allreduce_probe"] == "ok" and e["value"] == 1.0
    assert log[0][1] == ["--allreduce-form", "captured", "--probe"] and log[1][1] == ["--allreduce-form", "captured"] and log[1][2] == {}
    # probe fails on the replicas (not RCCL) -> split, no flip
    log.clear()
    e, failed = b.run_secondary_leg(ctx, None, "t", 8192, "weak", 3, 300.0, 0, child=child_of([(None, "leg exited with code 1: AssertionError: replicas differ"), (leg, None)]))
    assert not failed and e["allreduce_form"] == "split" and "replicas differ" in e["allreduce_probe"] and len(log) == 2 and log[1][1] == ["--allreduce-form", "split"]
    # probe dies on RCCL, comes up with the IPC mode flipped -> captured, timed leg with the flipped variable
    log.clear()
    e, failed = b.run_secondary_leg(ctx, None, "t", 8192, "weak", 3, 300.0, 0,
                                    child=child_of([(None, "leg exited with code 1: x", "NCCL WARN hipIpcGetMemHandle: invalid argument"), (None, None), (leg, None)]))
    flipped = log[1][2]["HSA_ENABLE_IPC_MODE_LEGACY"]
    assert not failed and e["allreduce_form"] == "captured" and e["allreduce_probe_retry"] == "ok" and e["ipc_mode_legacy_flipped_to"] == flipped == e["ipc_mode_legacy"]
    assert log[2][2] == {"HSA_ENABLE_IPC_MODE_LEGACY": flipped}
    # both probes die on RCCL -> split with the environment as given; the timed leg's failure shows, with its stderr file
    log.clear()
    e, failed = b.run_secondary_leg(ctx, None, "t", 8192, "weak", 3, 300.0, 0,
                                    child=child_of([(None, "ncclSystemError"), (None, "ncclSystemError"), (None, "leg exited with code 1: RCCL")]))
    assert failed and e["allreduce_form"] == "split" and "error" in e and e["stderr_file"].endswith("timed.stderr") and log[2][2] == {}
    # one GPU: no probe, no form
    log.clear()
    e, failed = b.run_secondary_leg((0, 1, 0, None), None, "t", 8192, "weak", 3, 300.0, 0, child=child_of([(leg, None)]))
    assert not failed and "allreduce_form" not in e and len(log) == 1 and log[0][1] == []


def test_ranks_agree_on_a_failure_over_gloo(tmp_path):
    """`agree_any` / `reduce_max_scalar` on a two-rank gloo group (the control plane the headline falls back to when RCCL does not come up):
    one rank's failed leg is every rank's exit path, and the timed region's elapsed time is the slowest rank's."""
    script = tmp_path / "agree.py"
    script.write_text(f"""
import os, sys, importlib.util
import torch.distributed as dist
spec = importlib.util.spec_from_file_location("odk_bench", {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dist.init_process_group("gloo")
b.DIST_INFO["backend"] = "gloo (control plane only: RCCL did not come up)"
r = dist.get_rank()
assert b.agree_any(r == 1, None) is True and b.agree_any(False, None) is False
assert b.reduce_max_scalar(1.0 + r, None) == 2.0
print("ok", r)
dist.destroy_process_group()
""")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29717", str(script)],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.count("ok") == 2, out.stderr[-2000:]
