"""<equality> constraints (reference README.md:74-85 "adding a new robot"): joint / connect / weld compiled by mjcf.py and
built as rows by the float64 oracle (oracle/odk_oracle.c: make_equality -- MJX constraint._efc_equality_* / MuJoCo
mj_instantiateEquality AS RECALLED: parity unpinned like the rest of the physics).  What pins them here is their own definition:
the compiled data satisfies the constraint at qpos0, a row's Jacobian is the derivative of its residual, the residual decays at the
rate `solref` names, a pinned foot carries the robot, a welded joint stays put, and the solver's fixed point is the stationary point
of MuJoCo's documented cost with the rows always active.  The kernels refuse a model with active equalities BY NAME."""
import os

import numpy as np
import pytest

from open_duck_playground_amd.model import Model

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
CONNECT, WELD, JOINT = 0, 1, 2


@pytest.fixture(scope="module")
def robot(oracle_mod):
    m = Model.from_xml(os.path.join(ASSETS, "tail_biped_equality.xml"))
    return m, oracle_mod.OracleModel(m.blob())


def _air(O, om, m, rng=None, spread=0.4):
    """a state well above the floor (no contacts), joints spread about the reference pose"""
    d = O.OracleData(om)
    q = np.array(m.a["qpos0"], float); q[2] = 1.5
    if rng is not None:
        q[7:] += rng.uniform(-spread, spread, m.nq - 7)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(-0.8, 0.8)
        q[3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    d["qpos"][: m.nq] = q
    return d


def _integrate(q, v, eps, nq):
    """qpos moved along qvel by eps (free joint: position + quaternion by the body-frame angular velocity; hinges: plain sum)"""
    out = np.array(q[:nq], float)
    out[0:3] += eps * v[0:3]
    w = v[3:6] * eps
    th = np.linalg.norm(w)
    dq = np.array([1.0, 0, 0, 0]) if th < 1e-15 else np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * w / th])
    a, b = out[3:7], dq
    out[3:7] = [a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3], a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1], a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]]
    out[7:] += eps * v[6:]
    return out


def test_the_compiler_reads_the_equality_section(robot):
    m, om = robot
    a = m.a
    assert int(a["neq"][0]) == 4 and om.neq == 4
    assert list(a["eq_type"]) == [JOINT, JOINT, CONNECT, WELD] and list(a["eq_active"]) == [1, 1, 1, 1]
    assert [m.joint_id(n) for n in ("tail_yaw_2", "left_ankle")] == list(a["eq_obj1id"][:2])
    assert [m.joint_id(n) for n in ("tail_yaw_1", "left_knee")] == list(a["eq_obj2id"][:2])
    assert list(a["eq_obj1id"][2:]) == [m.body_id("right_foot_link"), m.body_id("tail_5")] and list(a["eq_obj2id"][2:]) == [0, m.body_id("tail_4")]
    np.testing.assert_allclose(a["eq_data"][1][:5], [0.1, -0.5, 0.2, 0, 0])
    np.testing.assert_allclose(a["eq_solref"][1], [0.01, 1.0]); np.testing.assert_allclose(a["eq_solref"][0], [0.02, 1.0])
    # connect: the anchor's world position at qpos0 is the second anchor (body2 = world)
    from open_duck_playground_amd import mjcf
    xpos, xquat, _, _ = mjcf.body_frames(a, a["qpos0"])
    b = m.body_id("right_foot_link")
    np.testing.assert_allclose(a["eq_data"][2][3:6], xpos[b] + mjcf.quat_to_mat(xquat[b]) @ a["eq_data"][2][:3], atol=1e-12)
    # weld: relative pose of the reference configuration, torquescale 1
    np.testing.assert_allclose(a["eq_data"][3][6:10], [1, 0, 0, 0], atol=1e-12); assert a["eq_data"][3][10] == 1.0
    # a model without the section: neq 0, no eq_* records
    plain = Model.from_xml(os.path.join(ASSETS, "tail_biped.xml"))
    assert int(plain.a["neq"][0]) == 0 and "eq_type" not in plain.a


def test_rows_vanish_at_the_reference_pose_and_come_first(robot, oracle_mod):
    m, om = robot
    d = oracle_mod.OracleData(om)
    d.forward()
    assert d.i("ne") == 3 + 6 + 2 and d.i("nf") == 15 and d.i("nefc") == d.i("ne") + d.i("nf") + d.i("nl") + d.i("nc")
    pos = d["efc_pos"][:11]
    np.testing.assert_allclose(pos[:10], 0, atol=1e-12)         # connect, weld, the linear coupling: satisfied at qpos0
    assert abs(pos[10] + 0.1) < 1e-12                           # left_ankle = 0.1 - 0.5 knee + 0.2 knee^2: violated by the constant term
    assert (d["efc_D"][:11] > 0).all()


def test_a_row_jacobian_is_the_derivative_of_its_residual(robot, oracle_mod):
    """(pos(q + eps v) - pos(q - eps v)) / 2 eps == J v for every equality row, at random poses and velocities: holds the recalled weld
    rotation Jacobian (0.5 conj(q2) (0, w1 - w2) q1 relpose) and the polynomial coupling's -p'(x) to their own residuals."""
    m, om = robot
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(12):
        d = _air(oracle_mod, om, m, rng)
        d.forward()
        ne = d.i("ne")
        J = d.J()[:ne]
        q0 = d["qpos"][: m.nq].copy()
        v = rng.normal(size=m.nv)
        eps = 1e-6
        res = []
        for sgn in (1, -1):
            d["qpos"][: m.nq] = _integrate(q0, v, sgn * eps, m.nq)
            d.forward()
            res.append(d["efc_pos"][:ne].copy())
        fd = (res[0] - res[1]) / (2 * eps)
        worst = max(worst, np.abs(fd - J @ v).max() / max(np.abs(J @ v).max(), 1e-9))
    assert worst < 1e-6, worst


def test_impedance_of_a_vector_constraint_uses_the_norm_of_its_residual(robot, oracle_mod):
    m, om = robot
    d = _air(oracle_mod, om, m, np.random.default_rng(3))
    d.forward()
    imp = d["efc_imp"][:11]
    assert np.ptp(imp[:3]) == 0 and np.ptp(imp[3:9]) == 0       # one impedance per connect / per weld
    # solimp default (0.9, 0.95, 0.001, 0.5, 2): far from the reference pose (|pos| >> width) the impedance saturates at dmax
    assert abs(imp[0] - 0.95) < 1e-12


def _run(O, om, m, steps, ctrl=None, setup=None, gravity=True):
    d = _air(O, om, m)
    if setup:
        setup(d)
    if ctrl is not None:
        d["ctrl"][: m.nu] = ctrl
    g = om.f["gravity"]; g0 = g.copy()
    if not gravity:
        g[:] = 0
    hist = []
    for _ in range(steps):
        d.step()
        hist.append((d["qpos"][: m.nq].copy(), d["efc_pos"][: d.i("ne")].copy()))
    g[:] = g0
    return d, hist


def test_joint_coupling_decays_at_the_solref_rate(robot, oracle_mod):
    """tail_yaw_2 = 0.5 tail_yaw_1, released 0.3 rad off in free fall without gravity.  MuJoCo's reference acceleration is
    aref = -b v - k imp pos with b = 2 / (dmax tc), k = 1 / (dmax^2 tc^2 dampratio^2): a critically damped return with time constant
    ~ dmax tc = 19 ms; the soft constraint realises A / (A + R) ~ 95 % of it.  Checked: monotone, no overshoot, half-life and 5 tc / 10 tc
    levels inside a band around that ODE."""
    m, om = robot
    for e in (1, 2, 3):
        om.eq_set_active(e, False)
    try:
        j1, j2 = m.joint_id("tail_yaw_2"), m.joint_id("tail_yaw_1")
        a1, a2 = int(m.a["jnt_qposadr"][j1]), int(m.a["jnt_qposadr"][j2])
        def setup(d): d["qpos"][a1] = 0.3
        hold = np.array(m.a["key_ctrl"], float)          # position actuators at the reference pose keep the rest of the robot still
        d, hist = _run(oracle_mod, om, m, 100, ctrl=None, setup=setup, gravity=False)
        err = np.array([h[0][a1] - 0.5 * h[0][a2] for h in hist])
        assert d.i("ne") == 1
        np.testing.assert_allclose(err[:-1], [h[1][0] for h in hist][1:], atol=1e-12)  # efc_pos of the row IS this residual (built BEFORE the Euler step: one step behind the state)
    finally:
        for e in (1, 2, 3):
            om.eq_set_active(e, True)
    e0, dt, tc = 0.3, 0.002, 0.02
    assert (np.diff(err) < 1e-9).all() and err.min() > -0.02 * e0                      # decays, no overshoot worth the name
    # reference ODE e'' = s (-b e' - k imp e), s = 0.95 (A / (A + R) with R = invweight (1 - imp) / imp and A ~ invweight), semi-implicit Euler like the engine
    b, k_imp, s = 2 / (0.95 * tc), 0.95 / (0.95 ** 2 * tc ** 2), 0.95
    x, v, ref = e0, 0.0, []
    for _ in range(100):
        v += dt * s * (-b * v - k_imp * x); x += dt * v; ref.append(x)
    ref = np.array(ref)
    half = lambda y: int(np.argmax(y < 0.5 * e0))
    assert abs(half(err) - half(ref)) <= 3, (half(err), half(ref))
    assert abs(err[49] - ref[49]) < 0.03 * e0 and abs(err[99]) < 0.01 * e0, (err[49], ref[49], err[99])


def test_a_pinned_foot_carries_the_robot(robot, oracle_mod):
    """<connect> of the right foot to the world, robot released in gravity with its joints held by the actuators: the anchor stays within
    two centimetres of its world point while the robot swings (3.7 mm of static sag at the default solref), and the constraint force on the floating base's translational dofs balances
    weight + inertia: sum of forces = m (a_com - g) at every step (Newton's law for the whole robot, the only external force besides
    gravity being the pin)."""
    m, om = robot
    for e in (0, 1, 3):
        om.eq_set_active(e, False)
    try:
        d = _air(oracle_mod, om, m)
        d["ctrl"][: m.nu] = m.a["key_ctrl"]
        eqd = om.f["eq_data"]                                       # the world anchor moves up with the robot (it was compiled at qpos0)
        lift = 1.5 - float(m.a["qpos0"][2])
        eqd[2 * 11 + 5] += lift
        mass = float(np.sum(m.a["body_mass"]))
        drift, bal = 0.0, 0.0
        for t in range(400):
            d.step()
            ne = d.i("ne"); assert ne == 3
            cpos = d["efc_pos"][:3]
            drift = max(drift, float(np.linalg.norm(cpos)))
            # base translational dofs are world-aligned: M qacc - qfrc_smooth = J^T f on them; J^T f there = the pin's force on the robot
            F = d["qfrc_constraint"][:3].copy()
            J = d.J()[:3]
            np.testing.assert_allclose(F, (J.T @ d["efc_force"][:3])[:3], atol=1e-9)
            assert d.i("nc") == 48 and (d["efc_force"][d.i("nefc") - 48: d.i("nefc")] == 0).all()      # far above the floor: no contact force
        assert drift < 2e-2, drift      # (a soft constraint: g / (k imp) = 9.81 / 2632 = 3.7 mm at rest, a few times that in the release transient)
        # after the transient the mean pin force over a swing carries the weight (to a few per cent: the robot still swings)
        Fz = []
        for t in range(1500):
            d.step(); Fz.append(d["qfrc_constraint"][2])
        assert abs(np.mean(Fz[-1000:]) / (mass * 9.81) - 1.0) < 0.05, np.mean(Fz[-1000:]) / (mass * 9.81)
    finally:
        eqd[2 * 11 + 5] -= lift
        for e in (0, 1, 3):
            om.eq_set_active(e, True)


def test_a_welded_joint_stays_put(robot, oracle_mod):
    """tail_5 welded to tail_4: its hinge (tail_roll) is driven to 0.8 rad by its position actuator -- without the weld it goes there,
    with the weld it stays within a few hundredths (a soft constraint against a kp = 6 servo)."""
    m, om = robot
    u = m.actuator_id("tail_roll") if m.actuator_id("tail_roll") >= 0 else None
    names = list(m.a["names_actuator"])
    u = next(i for i, n in enumerate(names) if "tail_roll" in n)
    qa = int(m.a["jnt_qposadr"][m.joint_id("tail_roll")])
    ctrl = np.array(m.a["key_ctrl"], float); ctrl[u] = 0.8
    out = {}
    for weld_on in (False, True):
        for e in (0, 1, 2):
            om.eq_set_active(e, False)
        om.eq_set_active(3, weld_on)
        try:
            d, hist = _run(oracle_mod, om, m, 300, ctrl=ctrl, gravity=False)
            out[weld_on] = hist[-1][0][qa]
        finally:
            for e in range(4):
                om.eq_set_active(e, True)
    assert out[False] > 0.6 and abs(out[True]) < 0.05, out


def test_converged_solver_is_stationary_with_equality_rows_always_active(robot, oracle_mod):
    """With the Newton solver run to convergence, M qacc - qfrc_smooth = J^T f with f = -D (J qacc - aref) on EVERY equality row (no
    inequality gate: an equality pushes and pulls), f by the friction-loss / limit / contact rules elsewhere."""
    m, om = robot
    om.set_int("iterations", 60); om.set_int("ls_iterations", 50)
    try:
        rng = np.random.default_rng(5)
        for _ in range(4):
            d = _air(oracle_mod, om, m, rng, spread=0.25)
            d["qvel"][: m.nv] = rng.normal(0, 0.5, m.nv)
            d["ctrl"][: m.nu] = m.a["key_ctrl"]
            d.forward()
            ne, nefc, nv = d.i("ne"), d.i("nefc"), m.nv
            J, f = d.J(), d["efc_force"][:nefc]
            jar = J @ d["qacc"][:nv] - d["efc_aref"][:nefc]
            np.testing.assert_allclose(f[:ne], -d["efc_D"][:ne] * jar[:ne], rtol=1e-9, atol=1e-12)
            assert (np.sign(f[:ne]) != 0).any() and (f[:ne] > 0).any() and (f[:ne] < 0).any()       # both signs occur: not gated like a contact
            resid = d.M() @ d["qacc"][:nv] - d["qfrc_smooth"][:nv] - J.T @ f
            assert np.abs(resid).max() < 1e-6 * max(1.0, np.abs(d["qfrc_smooth"][:nv]).max()), np.abs(resid).max()
    finally:
        om.set_int("iterations", 1); om.set_int("ls_iterations", 5)


def test_inactive_equalities_leave_the_model_untouched(robot, oracle_mod):
    m, om = robot
    plain = oracle_mod.OracleModel(Model.from_xml(os.path.join(ASSETS, "tail_biped.xml")).blob())
    for e in range(4):
        om.eq_set_active(e, False)
    try:
        rng = np.random.default_rng(8)
        a, b = _air(oracle_mod, om, m, rng), _air(oracle_mod, plain, m, np.random.default_rng(8))
        for d in (a, b):
            d["qvel"][: m.nv] = 0.3
            for _ in range(5):
                d.step()
        assert a.i("ne") == 0 and np.array_equal(a["qpos"][: m.nq], b["qpos"][: m.nq]) and np.array_equal(a["qvel"][: m.nv], b["qvel"][: m.nv])
    finally:
        for e in range(4):
            om.eq_set_active(e, True)
