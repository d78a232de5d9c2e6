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
