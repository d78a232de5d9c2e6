"""The HIP path against MJX DIRECTLY (no oracle in between), on the fixtures of tools/dump_mjx_golden.py.  Skips while
tests/golden/mjx_<task>.npz do not exist (nothing of the jax / mujoco stack can be installed in the build container)."""
import numpy as np
import pytest

from test_mjx_golden import RTOL_Q, TASKS, _golden, _rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task", TASKS)
def test_hip_step_matches_mjx(task, parity_log):
    import torch
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    g = _golden(task)
    model = load_task_model(task)
    n = len(g["qpos"])
    ctrl = torch.tensor(np.asarray(g["ctrl"]), dtype=torch.float32, device="cuda")
    b = engine.Batch(model, n)
    b.set_state(g["qpos"], g["qvel"], g["warm"])
    b.physics_step(ctrl, 1)
    q1, v1, _ = b.get_state()
    dbg = b.get_debug()
    wq = float(_rel(q1, g["step_qpos"], 1e-2).max()); wv = float(_rel(v1, g["step_qvel"], 1.0).max())
    wa = float(_rel(dbg["qacc"], g["fwd_qacc"], 5.0).max()); ws = float(_rel(dbg["sensordata"], g["fwd_sensordata"], 1.0).max())
    idx = np.asarray(g["env10_index"], int)
    b.set_state(g["qpos"], g["qvel"], g["warm"])
    b.physics_step(ctrl, 10)
    q10, v10, _ = b.get_state()
    w10 = float(_rel(q10[idx], g["env10_qpos"], 1e-2).max())
    b.close()
    parity_log.check(f"hip_vs_mjx/{task}", dict(qpos=RTOL_Q, qvel=RTOL_Q, qacc=2e-3, sensordata=2e-3, qpos_10_substeps=5 * RTOL_Q),
                     qpos=wq, qvel=wv, qacc=wa, sensordata=ws, qpos_10_substeps=w10)


def test_hip_loader_accepts_a_fixture_of_the_dumper_s_layout(oracle_mod, parity_log, tmp_path, monkeypatch):
    """Plumbing only, NOT parity against MJX: the loader above on a file of the dumper's layout filled from the oracle (the numbers it
    compares are therefore the kernel-vs-oracle ones of tests/test_gpu_parity.py)."""
    import test_mjx_golden as T
    T._oracle_made_fixture(oracle_mod, "flat_terrain_backlash", tmp_path / "mjx_flat_terrain_backlash.npz")
    monkeypatch.setattr(T, "GOLDEN", str(tmp_path))

    class _Quiet:   # keeps the synthetic numbers out of the parity record
        def check(self, name, bounds, **vals):
            bad = {k: (v, bounds[k]) for k, v in vals.items() if k in bounds and not v <= bounds[k]}
            assert not bad, bad
    test_hip_step_matches_mjx("flat_terrain_backlash", _Quiet())
