"""Analytic invariants that pin the oracle's physics.  No MJX/MuJoCo install
exists, so these replace golden vectors: PARITY UNPINNED vs MJX, see DESIGN.md."""
import numpy as np
import pytest

from open_duck_playground_amd import mjcf

G = 9.81
MASS = 2.1071407


def _data(O, model, qpos=None, qvel=None, ctrl=None):
    om = O.OracleModel(model.blob())
    d = O.OracleData(om)
    d["qpos"][: om.nq] = model.a["key_qpos"] if qpos is None else qpos
    if qvel is not None:
        d["qvel"][: om.nv] = qvel
    d["ctrl"][: om.nu] = model.a["key_ctrl"] if ctrl is None else ctrl
    return om, d


def _random_qpos(model, rng, z=1.0):
    q = np.array(model.a["key_qpos"], float)
    q[2] = z
    quat = rng.normal(size=4); q[3:7] = quat / np.linalg.norm(quat)
    for j in range(1, model.njnt):
        a = model.a["jnt_qposadr"][j]
        lo, hi = model.a["jnt_range"][j]
        q[a] = rng.uniform(lo, hi)
    return q


@pytest.mark.parametrize("task", ["a", "b"])
def test_mass_matrix_matches_independent_jacobian_sum(oracle_mod, model_a, model_b, task):
    model = model_a if task == "a" else model_b
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = _random_qpos(model, rng)
        om, d = _data(oracle_mod, model, qpos=q)
        d.forward()
        M = d.M()
        M2, _ = mjcf.mass_matrix(model.a, q)  # sum_b J^T diag(m,I) J, not CRBA
        np.testing.assert_allclose(M, M2, atol=1e-12)
        np.testing.assert_allclose(M, M.T, atol=0)
        assert np.linalg.eigvalsh(M).min() > 0
        np.testing.assert_allclose(M[:3, :3], MASS * np.eye(3), atol=1e-12)


def test_kinematics_matches_numpy(oracle_mod, model_a):
    rng = np.random.default_rng(1)
    q = _random_qpos(model_a, rng)
    om, d = _data(oracle_mod, model_a, qpos=q)
    d.forward()
    xpos, xquat, _, _ = mjcf.body_frames(model_a.a, q)
    np.testing.assert_allclose(d["xpos"][: om.nbody * 3].reshape(-1, 3), xpos, atol=1e-12)
    np.testing.assert_allclose(d["xquat"][: om.nbody * 4].reshape(-1, 4), xquat, atol=1e-12)
    # site positions
    for s in range(om.nsite):
        b = model_a.a["site_bodyid"][s]
        p = xpos[b] + mjcf.quat_to_mat(xquat[b]) @ model_a.a["site_pos"][s]
        np.testing.assert_allclose(d["site_xpos"][3 * s: 3 * s + 3], p, atol=1e-12)


def test_free_fall_momentum_rate(oracle_mod, model_a):
    """Airborne, zero velocity: d/dt (linear momentum) = M[0:3,:] qacc = m g, whatever the joints do."""
    rng = np.random.default_rng(2)
    q = _random_qpos(model_a, rng, z=1.0)
    om, d = _data(oracle_mod, model_a, qpos=q, ctrl=rng.uniform(-1, 1, 14))
    d.forward()
    assert (d["contact_dist"][:8] > 0).all()
    M = d.M()
    np.testing.assert_allclose(M[:3] @ d["qacc"][: om.nv], [0, 0, -MASS * G], atol=1e-9)


def test_gravity_bias_matches_jacobian(oracle_mod, model_a):
    """qvel = 0: qfrc_bias = -sum_b m_b Jp_b^T g (independent Jacobians from the model compiler)."""
    rng = np.random.default_rng(3)
    q = _random_qpos(model_a, rng)
    om, d = _data(oracle_mod, model_a, qpos=q)
    d.forward()
    _, jacs = mjcf.mass_matrix(model_a.a, q)
    expect = np.zeros(om.nv)
    for b, (Jp, _) in enumerate(jacs):
        expect += model_a.a["body_mass"][b] * Jp.T @ np.array([0, 0, G])
    np.testing.assert_allclose(d["qfrc_bias"][: om.nv], expect, atol=1e-10)


def test_coriolis_matches_lagrangian_finite_difference(oracle_mod, model_a):
    """Hinge rows of the bias force with gravity off: c_i = sum_k dM_i./dq_k v_k . v - 1/2 v^T dM/dq_i v
    (only hinge velocities non-zero, so only hinge partials are needed)."""
    rng = np.random.default_rng(4)
    q = _random_qpos(model_a, rng)
    nv = model_a.nv
    v = np.zeros(nv); v[6:] = rng.normal(0, 2.0, nv - 6)
    om, d = _data(oracle_mod, model_a, qpos=q, qvel=v)
    om.f["gravity"][:] = 0
    d.forward()
    h = 1e-6
    dM = np.zeros((nv, nv, nv))
    for k in range(6, nv):
        qp, qm = q.copy(), q.copy()
        qp[k + 1] += h; qm[k + 1] -= h
        dM[k] = (mjcf.mass_matrix(model_a.a, qp)[0] - mjcf.mass_matrix(model_a.a, qm)[0]) / (2 * h)
    c = np.einsum("kij,k,j->i", dM, v, v) - 0.5 * np.einsum("ijk,j,k->i", dM, v, v)
    np.testing.assert_allclose(d["qfrc_bias"][6:nv], c[6:], atol=2e-7)


def test_static_stance_balances_weight(oracle_mod, model_a):
    om, d = _data(oracle_mod, model_a)
    d.forward()
    assert d.i("nefc") == 76 and (d.i("nf"), d.i("nl"), d.i("nc")) == (14, 14, 48)
    for _ in range(500):
        d.env_physics_step(model_a.a["key_ctrl"], 1)
    assert np.abs(d["qvel"][: om.nv]).max() < 2e-2
    J, f = d.J(), d["efc_force"][:76]
    np.testing.assert_allclose((J.T @ f)[2], MASS * G, rtol=2e-3)   # vertical support = weight
    assert (f[28:] >= 0).all()                                       # pyramid forces are unilateral
    np.testing.assert_allclose(d["sensordata"][6:9], [0, 0, G], atol=0.3)  # accelerometer ~ +g at rest
    assert d["sensordata"][11] > 0.99                                # upvector z
    assert 0.14 < d["qpos"][2] < 0.18


def test_solver_decreases_cost_and_limits_work(oracle_mod, model_a):
    q = np.array(model_a.a["key_qpos"], float)
    q[2] = 1.0
    q[10] = 2.0  # left knee beyond its +1.5708 limit
    om, d = _data(oracle_mod, model_a, qpos=q)
    d.forward()
    assert d["solver_cost1"][0] <= d["solver_cost0"][0]
    nf = d.i("nf")
    pos = d["efc_pos"][nf: nf + 14]
    assert pos[3] == pytest.approx(1.5707963267948966 - 2.0)
    assert d["efc_force"][nf + 3] > 0  # limit pushes back
    assert d["efc_J"][(nf + 3) * om.nv + 9] == -1.0


def test_backlash_model_counts(oracle_mod, model_b):
    om, d = _data(oracle_mod, model_b)
    d.forward()
    assert (om.nq, om.nv) == (31, 30)
    assert (d.i("nf"), d.i("nl"), d.i("nc")) == (14, 24, 48)
    for _ in range(300):
        d.env_physics_step(model_b.a["key_ctrl"], 1)
    assert np.isfinite(d["qpos"][: om.nq]).all() and 0.13 < d["qpos"][2] < 0.18


def test_float32_build_tracks_float64(oracle_mod, model_a):
    rng = np.random.default_rng(5)
    ctrl = np.array(model_a.a["key_ctrl"]) + rng.uniform(-0.2, 0.2, 14)
    res = []
    for f32 in (False, True):
        om = oracle_mod.OracleModel(model_a.blob(), f32=f32)
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = model_a.a["key_qpos"]
        d.env_physics_step(ctrl, 1)
        res.append((np.array(d["qpos"][: om.nq], float), np.array(d["qvel"][: om.nv], float)))
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=1e-3, atol=2e-4)


def test_height_field_one_triangle_mode(oracle_mod):
    """rough_terrain_backlash, hfield_mode = 1 (round 2's approximation, kept to measure its difference to the prism algorithm of
    tests/test_oracle_convex.py): the contact normal under each foot is the normal of the height-field triangle below
    it (independent numpy evaluation of the same rule), the robot settles on the bumps, and flat patches reduce
    to the plane case."""
    from open_duck_playground_amd.model import load_task_model
    model = load_task_model("rough_terrain_backlash")
    a = model.a
    H, size = np.asarray(a["hfield_data"]), np.asarray(a["hfield_size"])
    assert H.shape == (256, 256) and tuple(size) == (10.0, 10.0, 0.01, 0.1)            # scene_rough_terrain_backlash.xml:22
    nr, nc = H.shape
    dx, dy = 2 * size[0] / (nc - 1), 2 * size[1] / (nr - 1)

    def tri_plane(x, y):
        fx, fy = (x + size[0]) / dx, (y + size[1]) / dy
        c, r = int(np.clip(np.floor(fx), 0, nc - 2)), int(np.clip(np.floor(fy), 0, nr - 2))
        z = lambda rr, cc: H[rr, cc] * size[2]
        if (fx - c) + (fy - r) <= 1.0:
            p0 = np.array([-size[0] + c * dx, -size[1] + r * dy, z(r, c)])
            n = np.cross([dx, 0, z(r, c + 1) - z(r, c)], [0, dy, z(r + 1, c) - z(r, c)])
        else:
            p0 = np.array([-size[0] + (c + 1) * dx, -size[1] + (r + 1) * dy, z(r + 1, c + 1)])
            n = np.cross([-dx, 0, z(r + 1, c) - z(r + 1, c + 1)], [0, -dy, z(r, c + 1) - z(r + 1, c + 1)])
        return p0, n / np.linalg.norm(n)

    rng = np.random.default_rng(0)
    tilts = []
    for _ in range(12):
        q = np.array(a["key_qpos"], dtype=np.float64)
        q[0:2] = rng.uniform(-4, 4, 2); q[2] = 0.16
        _om, d = _data(oracle_mod, model, qpos=q)
        _om.set_int("hfield_mode", 1)
        d.forward()
        frames = np.array(d["contact_frame"][: 8 * 9]).reshape(8, 9)
        dist = np.array(d["contact_dist"][:8]); pos = np.array(d["contact_pos"][: 8 * 3]).reshape(8, 3)
        gx = np.array(d["geom_xpos"][:9]).reshape(3, 3); gm = np.array(d["geom_xmat"][:27]).reshape(3, 3, 3)
        hv = np.asarray(a["hull_vert"])
        for f in range(2):   # collision geoms: left foot, right foot, floor
            v = hv[a["cgeom_vertadr"][f]: a["cgeom_vertadr"][f] + a["cgeom_vertnum"][f]]
            centre = gx[f] + gm[f] @ (0.5 * (v.min(0) + v.max(0)))                   # world centre of the hull's box
            p0, n = tri_plane(centre[0], centre[1])
            nrm = frames[4 * f, :3]
            assert np.allclose(frames[4 * f: 4 * f + 4, :3], nrm)                      # one plane per foot
            np.testing.assert_allclose(nrm, n, atol=1e-9)
            wv = gx[f] + v @ gm[f].T
            depth = (p0 - wv) @ n                                                      # support of every hull vertex
            live = dist[4 * f: 4 * f + 4] < 0.9
            assert live[0]
            for k in range(4):   # manifold points are hull vertices within 1e-3 of the deepest one (_manifold_points mask)
                if live[k]:
                    assert np.abs(depth + dist[4 * f + k]).min() < 1e-12 and -dist[4 * f + k] > depth.max() - 1e-3 - 1e-12
            for k in range(4):
                if live[k]:   # contact point = vertex - dist/2 * n: it sits dist/2 off the plane along n
                    assert abs((pos[4 * f + k] - p0) @ n - 0.5 * dist[4 * f + k]) < 1e-9
            tilts.append(np.degrees(np.arccos(nrm[2])))
    assert 0.05 < max(tilts) < 10.0 and min(tilts) >= 0.0                                # gentle bumps: <= 1 cm per 7.8 cm cell
    # settles on the terrain
    _om, d = _data(oracle_mod, model, qpos=np.array(a["key_qpos"], dtype=np.float64))
    _om.set_int("hfield_mode", 1)
    ctrl = np.asarray(a["key_ctrl"], dtype=np.float64)
    for _ in range(40):
        d.env_physics_step(ctrl, 10)
    ground = H[nr // 2 - 2: nr // 2 + 2, nc // 2 - 2: nc // 2 + 2].mean() * size[2]
    assert 0.14 + ground - 0.01 < d["qpos"][2] < 0.18 + ground and d["sensordata"][11] > 0.99


def test_actuator_force_clamp_and_pd_law(oracle_mod, model_a):
    """Position actuators (open_duck_mini_v2.xml:45-50): force = kp (ctrl - q) (kv = 0), clamped to forcerange +-3.23;
    ctrl itself is clamped to the joint range (inheritrange)."""
    a = model_a.a
    q = np.array(a["key_qpos"], float); q[2] = 1.0
    ctrl = np.array(a["key_ctrl"], float)
    ctrl[3] += 0.1          # left knee: small error -> linear regime
    ctrl[12] += 5.0         # right knee: far outside its range -> ctrl clamp (ctrlrange = joint range)
    jr = int(a["actuator_trnid"][10]); lo, hi = a["jnt_range"][jr]
    q[7 + 10] = lo + 0.01   # right hip roll at its lower stop, target at the upper one -> force clamp
    ctrl[10] = hi
    om, d = _data(oracle_mod, model_a, qpos=q, ctrl=ctrl)
    d.forward()
    f = np.array(d["actuator_force"][:14])
    kp = float(a["actuator_gainprm0"][3])
    assert kp == pytest.approx(13.37)
    assert f[3] == pytest.approx(kp * 0.1, rel=1e-9)
    khi = a["jnt_range"][int(a["actuator_trnid"][12])][1]
    assert f[12] == pytest.approx(kp * (khi - q[7 + 12]), rel=1e-9)      # ctrl clamped to the joint range first
    assert kp * (hi - lo - 0.01) > 3.23 and f[10] == pytest.approx(3.23, rel=1e-12)   # forcerange
    others = [u for u in range(14) if u not in (3, 10, 12)]
    np.testing.assert_allclose(f[others], 0.0, atol=1e-12)              # ctrl == q at the keyframe
    # generalized force = gear * force on the actuated dof only
    qa = np.array(d["qfrc_actuator"][: om.nv])
    assert qa[6 + 3] == pytest.approx(f[3]) and np.abs(qa[:6]).max() == 0.0


def test_sliding_friction_opposes_motion_within_the_cone(oracle_mod, model_a):
    """Standing robot pushed sideways: the tangential contact force opposes the sliding velocity and stays inside the
    friction pyramid (mu = 0.6: floor priority 1, scene_flat_terrain.xml:35-36)."""
    om, d = _data(oracle_mod, model_a)
    for _ in range(300):                      # settle
        d.env_physics_step(model_a.a["key_ctrl"], 1)
    d["qvel"][0] = 0.5                        # base slides along +x
    d.forward()
    J, f = d.J(), np.array(d["efc_force"][:76])
    nf, nl = d.i("nf"), d.i("nl")
    Jc, fc = J[nf + nl:], f[nf + nl:]
    F = Jc.T @ fc                             # generalized contact force; rows 0..2 = world force on the floating base
    assert F[2] > 0.5 * MASS * G              # feet still carry the robot
    assert F[0] < 0                           # friction opposes +x sliding
    assert abs(F[0]) <= 0.6 * F[2] * 1.0001 and abs(F[1]) <= 0.6 * F[2] * 1.0001
    assert (fc >= 0).all()


def test_frictionloss_holds_a_joint_against_small_torque(oracle_mod, model_a):
    """dof frictionloss 0.068 (sts3215 class): an actuator torque below it, with the robot in free fall, leaves the
    joint acceleration ~0 (the friction-loss row cancels it); a torque well above it accelerates the joint."""
    a = model_a.a
    q = np.array(a["key_qpos"], float); q[2] = 1.0
    kp = float(a["actuator_gainprm0"][7])
    accs = []
    for torque in (0.03, 1.0):
        ctrl = np.array(a["key_ctrl"], float); ctrl[7] += torque / kp       # head_yaw: light distal link
        om, d = _data(oracle_mod, model_a, qpos=q, ctrl=ctrl)
        d.forward()
        accs.append(abs(d["qacc"][6 + 7]))
        assert d["actuator_force"][7] == pytest.approx(torque, rel=1e-9)
    assert accs[0] < 0.05 * accs[1] and accs[1] > 1.0


def test_newton_solver_converges_to_the_stationary_point(oracle_mod, model_a):
    """MuJoCo's soft-constraint dynamics are the minimiser of a convex cost; at the minimum
    M qacc = qfrc_smooth + J^T f(qacc).  The model runs ONE Newton iteration per step (iterations=1); driving the same
    solver code to convergence must reach that fixed point, and the one-iteration answer must lie between the
    warm start and it.  Pins gradient / Hessian / line search against each other, independently of MJX."""
    rng = np.random.default_rng(4)
    worst = 0.0
    for trial in range(6):
        q = np.array(model_a.a["key_qpos"], float)
        q[2] = 0.148 + 0.004 * trial                      # feet a few mm into the floor: contacts active
        q[7:] += rng.uniform(-0.15, 0.15, 14)
        q[10] = 1.62 if trial % 2 else q[10]              # a knee beyond its limit on odd trials
        qv = rng.normal(0, 0.5, 20)
        om, d = _data(oracle_mod, model_a, qpos=q, qvel=qv)
        d.forward()
        cost_1 = d["solver_cost1"][0]
        qacc_1 = np.array(d["qacc"][:20])
        om.set_int("iterations", 60); om.set_int("ls_iterations", 60)
        d2 = oracle_mod.OracleData(om)
        d2["qpos"][:21] = q; d2["qvel"][:20] = qv; d2["ctrl"][:14] = model_a.a["key_ctrl"]
        d2.forward()
        qacc = np.array(d2["qacc"][:20])
        M = d2.M()
        resid = M @ qacc - np.array(d2["qfrc_smooth"][:20]) - np.array(d2["qfrc_constraint"][:20])
        scale = np.abs(M @ qacc).max() + np.abs(np.array(d2["qfrc_smooth"][:20])).max()
        worst = max(worst, np.abs(resid).max() / scale)
        assert d2["solver_cost1"][0] <= cost_1 + 1e-9 * abs(cost_1)          # more iterations never cost more
        assert d2.i("nefc") == 76 and (np.array(d2["efc_force"][28:76]) >= 0).all()
        assert np.linalg.norm(qacc_1 - qacc) < np.linalg.norm(np.array(d["qacc_smooth"][:20]) - qacc) + 1e-9   # one step moves towards it
    assert worst < 1e-6, worst


def test_yaw_equivariance(oracle_mod, model_a):
    """Rotating the whole state about the vertical axis rotates the base's linear acceleration and leaves everything
    else unchanged (free-joint angular velocity is body-frame).  Airborne: any angle.  On the floor: 90 degrees, under
    which the friction pyramid's axes map onto themselves."""
    rng = np.random.default_rng(8)

    def run(q, v):
        om, d = _data(oracle_mod, model_a, qpos=q, qvel=v)
        d.forward()
        return np.array(d["qacc"][:20]), np.array(d["sensordata"][:46])

    for z, ang in ((0.6, 0.7), (0.6, 2.1), (0.15, np.pi / 2), (0.146, -np.pi / 2)):
        q = np.array(model_a.a["key_qpos"], float)
        q[0:2] = rng.uniform(-0.3, 0.3, 2); q[2] = z
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); tilt = 0.2
        q[3:7] = np.concatenate([[np.cos(tilt / 2)], np.sin(tilt / 2) * ax])
        q[7:] += rng.uniform(-0.2, 0.2, 14)
        v = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.5, 3), rng.normal(0, 1.0, 14)])
        c, s = np.cos(ang), np.sin(ang)
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        qz = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)])
        w1, x1, y1, z1 = qz; w2, x2, y2, z2 = q[3:7]
        q2 = q.copy(); v2 = v.copy()
        q2[0:3] = Rz @ q[0:3]
        q2[3:7] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                   w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
        v2[0:3] = Rz @ v[0:3]
        a1, s1 = run(q, v)
        a2, s2 = run(q2, v2)
        scale = np.abs(a1).max()
        np.testing.assert_allclose(a2[0:3], Rz @ a1[0:3], atol=1e-8 * scale + 1e-9)
        np.testing.assert_allclose(a2[3:], a1[3:], atol=1e-8 * scale + 1e-9)
        np.testing.assert_allclose(s2[0:9], s1[0:9], atol=1e-8 * np.abs(s1).max())     # gyro, local linvel, accelerometer are body-frame
        np.testing.assert_allclose(s2[9:12], Rz @ s1[9:12], atol=1e-9)                  # upvector = site z axis in the WORLD frame


def _impedance(pos, solimp):
    """MuJoCo documentation, 'Solver parameters': d(r) rises from dmin to dmax over `width` along a two-piece power curve"""
    dmin, dmax, width, mid, power = solimp
    x = abs(pos) / width
    if x >= 1.0:
        return dmax
    y = x ** power / mid ** (power - 1) if x < mid else 1.0 - (1.0 - x) ** power / (1.0 - mid) ** (power - 1)
    return dmin + y * (dmax - dmin)


def test_constraint_row_parameters_follow_the_documented_formulas(oracle_mod, model_a):
    """Every row's regulariser D = 1 / R and reference acceleration aref, recomputed here from MuJoCo's documented soft-constraint
    model (Computation chapter 'Solver parameters'; pyramidal rows as in mjx constraint.py): R = (1 - d) / d * A with A the row's
    approximate inverse inertia (dof_invweight0; for a pyramid edge (1 + mu^2) (w_1 + w_2) * 2 mu^2 / impratio), aref = -b v - k d r,
    b = 2 / (dmax timeconst), k = 1 / (dmax^2 timeconst^2 dampratio^2), timeconst >= 2 timestep (refsafe).  An independent
    restatement in numpy: a typo in the oracle's row arithmetic would pass every oracle-vs-kernel test."""
    a = model_a.a
    rng = np.random.default_rng(3)
    q = np.array(a["key_qpos"], float); q[2] = 0.148            # feet a few millimetres into the floor
    q[10] = 1.62                                                # left knee beyond its limit
    v = rng.normal(0, 0.5, model_a.nv)
    om, d = _data(oracle_mod, model_a, qpos=q, qvel=v)
    d.forward()
    nv, nf, nl, nc = om.nv, d.i("nf"), d.i("nl"), d.i("nc")
    assert (nf, nl, nc) == (14, 14, 48)
    J = d.J(); D = np.array(d["efc_D"][: nf + nl + nc]); aref = np.array(d["efc_aref"][: nf + nl + nc])
    dt = 0.002
    dof_w = np.array(om.f.view("dof_invweight0")[:nv]); body_w = np.array(om.f.view("body_invweight0")[: 2 * om.nbody]).reshape(-1, 2)
    vel = J @ v

    def row(pos, invw, solref, solimp, vrow):
        tc = max(solref[0], 2 * dt); dr = solref[1]
        dmax = solimp[1]
        imp = _impedance(pos, solimp)
        return imp / ((1 - imp) * invw), -2.0 / (dmax * tc) * vrow - imp * pos / (dmax * dmax * tc * tc * dr * dr)
    default_ref, default_imp = (0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0)      # MuJoCo defaults: the duck's XML sets none of them
    r = 0
    for i in range(nv):                                                       # friction loss: pos = 0, the dof's own velocity
        if a["dof_frictionloss"][i] > 0:
            De, ae = row(0.0, dof_w[i], default_ref, default_imp, v[i])
            assert D[r] == pytest.approx(De, rel=1e-12) and aref[r] == pytest.approx(ae, rel=1e-12, abs=1e-12)
            r += 1
    assert r == nf
    n_active_lim = 0
    for j in range(model_a.njnt):                                             # hinge limits
        if not a["jnt_limited"][j] or a["jnt_type"][j] == 0:
            continue
        qa, da = a["jnt_qposadr"][j], a["jnt_dofadr"][j]
        lo, hi = a["jnt_range"][j]
        pos = min(q[qa] - lo, hi - q[qa]); sgn = 1.0 if q[qa] - lo < hi - q[qa] else -1.0
        if pos < 0:
            n_active_lim += 1
            De, ae = row(pos, dof_w[da], default_ref, default_imp, sgn * v[da])
            assert J[r, da] == sgn and D[r] == pytest.approx(De, rel=1e-12) and aref[r] == pytest.approx(ae, rel=1e-12)
        else:
            assert not J[r].any()
        r += 1
    assert r == nf + nl and n_active_lim == 1
    cd = np.array(d["contact_dist"][:12]); n_active = 0
    floor_mu = 0.6                                                             # scene_flat_terrain.xml:35-36, priority 1
    for c in range(12):
        g2_body = [int(a["cgeom_bodyid"][g]) for g in range(3) if a["cgeom_type"][g] == 7]
        b_foot = g2_body[0] if c < 4 else g2_body[1]
        if c >= 8:
            assert cd[c] > 0                                                   # the feet do not touch each other here
            r += 4
            continue
        mu = floor_mu
        w = body_w[b_foot][0] + 0.0                                            # translational invweight of the two bodies (the floor: world)
        invw = (w + mu * mu * w) * 2 * mu * mu / 1.0                           # impratio = 1
        for k in range(4):
            if cd[c] < 0:
                n_active += 1
                De, ae = row(cd[c], invw, default_ref, default_imp, vel[r])
                assert D[r] == pytest.approx(De, rel=1e-12) and aref[r] == pytest.approx(ae, rel=1e-10)
            r += 1
    assert n_active >= 16
