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
