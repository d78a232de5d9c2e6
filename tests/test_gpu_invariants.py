"""GPU tests that need no oracle: symmetries of the physics itself, at BASELINE size.  The oracle's physics is unpinned against MJX
(DESIGN 2), so what the kernels compute is also checked against what rigid-body dynamics on a horizontal plane must satisfy whatever
the implementation: turning the whole scene about the vertical by quarter turns (the symmetry of the pyramidal friction cone) and
moving it along the floor changes nothing in the robot's own frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _qmul(a, b):
    w1, x1, y1, z1 = a.T; w2, x2, y2, z2 = b.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash"])
def test_yaw_and_translation_invariance_on_the_plane(torch_cuda, parity_log, task):
    """8192 random states (standing, airborne, feet in the floor) and a copy of each turned about z by a random number of quarter turns and moved up to
    3 m along the floor, same controls: after five substeps the joint coordinates, the base height, the base orientation relative to
    the turn, the body-frame angular velocity and the turned-back linear velocity agree to float32 rounding.  (World coordinates enter
    the kernels' arithmetic only through the floor plane and gravity; a frame mistake anywhere in kinematics, contacts or the
    integrator breaks this.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _random_states
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    n = 8192
    rng = np.random.default_rng(17)
    qpos, qvel = _random_states(model, n, rng)
    qvel[:] = rng.normal(0, 0.3, qvel.shape)
    # quarter turns: the pyramidal friction cone (four edges along the contact frame's world-aligned tangents) is itself only
    # symmetric under those -- under an arbitrary yaw MuJoCo's own contact forces change
    kq = rng.integers(0, 4, n); psi = kq * (np.pi / 2); sh = rng.uniform(-3, 3, (n, 2))
    c, s = np.array([1.0, 0.0, -1.0, 0.0])[kq], np.array([0.0, 1.0, 0.0, -1.0])[kq]
    q2, v2 = qpos.copy(), qvel.copy()
    q2[:, 0] = c * qpos[:, 0] - s * qpos[:, 1] + sh[:, 0]; q2[:, 1] = s * qpos[:, 0] + c * qpos[:, 1] + sh[:, 1]
    qz = np.stack([np.cos(psi / 2), 0 * psi, 0 * psi, np.sin(psi / 2)], 1)
    q2[:, 3:7] = _qmul(qz, qpos[:, 3:7])
    v2[:, 0] = c * qvel[:, 0] - s * qvel[:, 1]; v2[:, 1] = s * qvel[:, 0] + c * qvel[:, 1]     # free joint: linear velocity in the world frame, angular in the body's
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    out = []
    for q, v in ((qpos, qvel), (q2, v2)):
        b = engine.Batch(model, n)
        b.set_state(q, v, np.zeros((n, model.nv)))
        b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 5)
        gq, gv, _ = b.get_state()
        img = b.lds_image(); o = b.lds_offset("contact_dist")
        out.append((gq.astype(np.float64), gv.astype(np.float64), img[:, o: o + 12].copy()))
        b.close()
    (qa, va, ca), (qb, vb, cb) = out
    # turn the second run back
    xb = c * (qb[:, 0] - sh[:, 0]) + s * (qb[:, 1] - sh[:, 1]); yb = -s * (qb[:, 0] - sh[:, 0]) + c * (qb[:, 1] - sh[:, 1])
    qzi = qz * np.array([1, -1, -1, -1.0])
    quat_b = _qmul(qzi, qb[:, 3:7])
    sign = np.sign((quat_b * qa[:, 3:7]).sum(1))[:, None]
    vxb = c * vb[:, 0] + s * vb[:, 1]; vyb = -s * vb[:, 0] + c * vb[:, 1]
    err_q = np.maximum.reduce([np.abs(xb - qa[:, 0]), np.abs(yb - qa[:, 1]), np.abs(qb[:, 2] - qa[:, 2]), np.abs(sign * quat_b - qa[:, 3:7]).max(1), np.abs(qb[:, 7:] - qa[:, 7:]).max(1)])
    err_v = np.maximum.reduce([np.abs(vxb - va[:, 0]), np.abs(vyb - va[:, 1]), np.abs(vb[:, 2:] - va[:, 2:]).max(1)]) / np.maximum(np.abs(va).max(1), 1.0)
    in_contact = (ca[:, :8] < 0).any(1)
    # A handful of the 8192 copies decide a tie the other way (which four hull vertices carry the floor contact, a line-search branch):
    # their step is a different one, exactly as in the oracle comparison.  Judged: the 99.9 % quantile at rounding level, and how few lie
    # beyond it.  Measured: median 2.6e-6 / 4.3e-6 (flat / backlash), 99.9 % 2.4e-5 / 3.2e-5, worst 0.018 / 0.045.
    judged = ((ca < 0) == (cb < 0)).all(1)     # a foot within rounding of the floor in one copy only: a different step
    parity_log.check("invariance/" + task, dict(qvel_q999=1e-4, qpos_q999=1e-5, beyond_1e_3=0.002, contact_set_differs=0.002),
                     qvel_q999=float(np.quantile(err_v[judged], 0.999)), qpos_q999=float(np.quantile(err_q[judged], 0.999)),
                     beyond_1e_3=float((err_v[judged] > 1e-3).mean()), contact_set_differs=float(1.0 - judged.mean()),
                     qvel_median=float(np.median(err_v[judged])), qvel_worst=float(err_v[judged].max()))
    assert in_contact.mean() > 0.3, in_contact.mean()


def test_shift_and_half_turn_invariance_on_a_periodic_height_field(torch_cuda, parity_log):
    """The height-field kernel without the oracle: a terrain whose samples repeat every 8 cells and are even about the grid nodes
    (z = a cos(2 pi i / 8) + b cos(2 pi j / 8)) is the same terrain after a shift by whole periods and after a half turn about a node
    (the cells' diagonal split keeps its direction under a half turn, not under a quarter turn).  4096 states, half of them with
    feet in the terrain, and their moved / turned copies, up to 6 m apart: after five substeps they agree in the robot's frame.  What
    this exercises is the window arithmetic (cell indices, the window-relative coordinates that keep float32 digits), the prism
    construction and the contact frames at different places of the field."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _random_states
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model, load_task_model
    torch = torch_cuda
    base = load_task_model("rough_terrain_backlash")
    a = {k: np.array(v) for k, v in base.a.items()}
    nr, nc = a["hfield_data"].shape
    sx, sy, sz, _ = [float(v) for v in a["hfield_size"]]
    dx, dy = 2 * sx / (nc - 1), 2 * sy / (nr - 1)
    P = 8
    gi = 0.5 + 0.3 * np.cos(2 * np.pi * np.arange(nc) / P); gj = 0.2 * np.cos(2 * np.pi * np.arange(nr) / P)
    a["hfield_data"] = (gi[None, :] + gj[:, None] + 0.3).astype(np.float64)          # in [0, 1.3]: heights up to 1.3 cm
    model = Model(a, base.xml_path)
    n = 4096
    rng = np.random.default_rng(23)
    qpos, qvel = _random_states(model, n, rng)
    qpos[:, 0] = rng.uniform(-2.5, 2.5, n); qpos[:, 1] = rng.uniform(-2.5, 2.5, n)
    qpos[:, 2] += 0.008                                                              # about the terrain's mean height
    qvel[:] = rng.normal(0, 0.3, qvel.shape)
    turn = rng.integers(0, 2, n).astype(bool)                                         # half of the copies are turned by 180 degrees
    kx, ky = rng.integers(-4, 5, n) * P, rng.integers(-4, 5, n) * P                   # whole periods (cells)
    # x = -sx + i dx.  Shift: i' = i + kx.  Half turn about a node: i' = -i + t with t a multiple of the period -> x' = -x + (t - (nc - 1)) dx
    tx, ty = (nc - 1) // P * P + kx, (nr - 1) // P * P + ky
    c = np.where(turn, -1.0, 1.0)
    Tx = np.where(turn, (tx - (nc - 1)) * dx, kx * dx); Ty = np.where(turn, (ty - (nr - 1)) * dy, ky * dy)
    q2, v2 = qpos.copy(), qvel.copy()
    q2[:, 0] = c * qpos[:, 0] + Tx; q2[:, 1] = c * qpos[:, 1] + Ty
    qz = np.where(turn[:, None], np.array([[0.0, 0, 0, 1]]), np.array([[1.0, 0, 0, 0]]))
    q2[:, 3:7] = _qmul(qz, qpos[:, 3:7])
    v2[:, 0] = c * qvel[:, 0]; v2[:, 1] = c * qvel[:, 1]
    assert np.abs(q2[:, :2]).max() < 9.5
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    out = []
    for q, v in ((qpos, qvel), (q2, v2)):
        b = engine.Batch(model, n)
        b.set_state(q, v, np.zeros((n, model.nv)))
        b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 5)
        gq, gv, _ = b.get_state()
        img = b.lds_image(); o = b.lds_offset("contact_dist")
        out.append((gq.astype(np.float64), gv.astype(np.float64), img[:, o: o + 12].copy()))
        b.close()
    (qa, va, ca), (qb, vb, cb) = out
    xb = c * (qb[:, 0] - Tx); yb = c * (qb[:, 1] - Ty)
    quat_b = _qmul(qz * np.array([1, -1, -1, -1.0]), qb[:, 3:7])
    sign = np.sign((quat_b * qa[:, 3:7]).sum(1))[:, None]
    err_q = np.maximum.reduce([np.abs(xb - qa[:, 0]), np.abs(yb - qa[:, 1]), np.abs(qb[:, 2] - qa[:, 2]), np.abs(sign * quat_b - qa[:, 3:7]).max(1), np.abs(qb[:, 7:] - qa[:, 7:]).max(1)])
    err_v = np.maximum.reduce([np.abs(c * vb[:, 0] - va[:, 0]), np.abs(c * vb[:, 1] - va[:, 1]), np.abs(vb[:, 2:] - va[:, 2:]).max(1)]) / np.maximum(np.abs(va).max(1), 1.0)
    in_contact = (ca[:, :8] < 0).any(1)
    judged = ((ca < 0) == (cb < 0)).all(1)
    parity_log.check("invariance/periodic_height_field", dict(qvel_q99=2.5e-4, qpos_q99=4e-6, beyond_1e_2=0.005, contact_set_differs=0.01),   # measured: 7.4e-5 (shifted 2.1e-5, turned 9.2e-5), 1.0e-6, 0, 0; median 3.4e-6, worst 2.1e-3
                     qvel_q99=float(np.quantile(err_v[judged], 0.99)), qpos_q99=float(np.quantile(err_q[judged], 0.99)),
                     beyond_1e_2=float((err_v[judged] > 1e-2).mean()), contact_set_differs=float(1.0 - judged.mean()),
                     qvel_median=float(np.median(err_v[judged])), qvel_worst=float(err_v[judged].max()),
                     qvel_q99_turned=float(np.quantile(err_v[judged & turn], 0.99)), qvel_q99_shifted=float(np.quantile(err_v[judged & ~turn], 0.99)))
    assert in_contact.mean() > 0.3, in_contact.mean()
