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
