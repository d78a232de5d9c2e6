"""The C-ABI used from C: tests/capi/consumer.c (no Python, no torch in its process) is compiled against include/odk.h, linked with
libodk.so, and run on a model blob / reference-motion table written to files; its observations, rewards, dones and qpos after 12 env
steps equal the Python host's bit for bit (same library, same seeds, same actions)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _action(t, e, a):
    M = np.uint64(0xFFFFFFFF)
    mul = lambda x, k: (x.astype(np.uint64) * np.uint64(k)) & M      # 32-bit wrap-around, as in C
    h = mul(np.full_like(e, t), 2654435761) ^ mul(e, 40503) ^ mul(a, 2246822519)
    h ^= h >> np.uint64(15); h = mul(h, 2246822519); h ^= h >> np.uint64(13)
    return (h >> np.uint64(8)).astype(np.float32) * np.float32(2.0 / 16777216.0) - np.float32(1.0)


@pytest.mark.parametrize("task", ["flat_terrain", "rough_terrain_backlash"])
def test_c_program_matches_the_python_host(tmp_path, task):
    import torch
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "the test compiles a C program with hipcc"
    engine.build_library()
    libdir = os.path.dirname(engine.LIB_PATH)
    exe = str(tmp_path / "consumer")
    subprocess.check_call([hipcc, "-x", "c", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "capi", "consumer.c"), "-o", exe,
                           "-L" + libdir, "-l:" + os.path.basename(engine.LIB_PATH), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    model = load_task_model(task)
    prm = engine.load_prm()
    (tmp_path / "model.blob").write_bytes(model.blob())
    np.ascontiguousarray(prm["table"], np.float32).tofile(tmp_path / "table.f32")
    grids = np.concatenate([np.asarray(prm[k], np.float64).ravel() for k in ("dxs", "dys", "dthetas", "dx_range", "dy_range", "dtheta_range")])
    grids.tofile(tmp_path / "grids.f64")
    n, steps, seed = 256, 12, 77
    r = subprocess.run([exe, str(tmp_path / "model.blob"), str(tmp_path / "table.f32"), str(tmp_path / "grids.f64"), str(len(prm["dxs"])), str(len(prm["dys"])),
                        str(len(prm["dthetas"])), str(int(prm["nb_steps_in_period"][0])), str(n), str(steps), str(seed), str(tmp_path / "out.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.f32", np.float32)
    b = engine.Batch(model, n)
    b.reset(seed=seed)
    e, a = np.meshgrid(np.arange(n), np.arange(14), indexing="ij")
    for t in range(steps):
        b.step(torch.tensor(_action(t, e, a), device="cuda"))
    q, _, _ = b.get_state()
    ref = np.concatenate([b.obs.cpu().numpy().ravel(), b.reward.cpu().numpy(), b.done.cpu().numpy(), q.astype(np.float32).ravel()])
    b.close()
    assert out.shape == ref.shape and np.array_equal(out, ref), (np.abs(out - ref).max(), int((out != ref).sum()))
    assert float(np.abs(out[: n * 101]).max()) > 0.1 and np.isfinite(out).all()
