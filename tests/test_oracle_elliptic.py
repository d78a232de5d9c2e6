"""Elliptic friction cones in the oracle's Newton solver: rows, the three zones of the cone cost, the cone Hessian and the
exact line search (oracle/odk_oracle.c "elliptic cones": MJX solver / MuJoCo PrimalUpdateConstraint, HessianCone, PrimalEval AS RECALLED).
What pins them is the problem MuJoCo's documentation states, not the recollection: per contact the constraint force is the minimiser of
0.5 f^T R f + f^T x over the friction cone f_n >= 0, |f_t| <= mu f_n with R = diag(R_n, R_n / impratio, R_n / impratio) -- solved here
independently (closed form derived in the test + a brute-force check of that closed form) and compared with the solver's forces; the
converged solution satisfies the cone program's KKT conditions with the USER's friction coefficient; gradient and Hessian are the
derivatives of the cost; and the dynamics are equivariant under ANY rotation about the vertical (a pyramid is only under quarter turns)."""
import ctypes as C
import os

import numpy as np
import pytest

from open_duck_playground_amd.model import Model

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


@pytest.fixture(scope="module")
def robot(oracle_mod):
    m = Model.from_xml(os.path.join(ASSETS, "tail_biped_elliptic.xml"))
    return m, oracle_mod.OracleModel(m.blob())


def _standing(O, om, m, rng, sink=3e-3, vel=0.4):
    """robot on the floor, feet pressed `sink` into it, joints and velocities spread: both feet in contact, contacts sliding and sticking"""
    d = O.OracleData(om)
    q = np.array(m.a["key_qpos"], float)
    q[7:] += rng.uniform(-0.02, 0.02, m.nq - 7)
    d["qpos"][: m.nq] = q
    d.forward()
    dist = np.array(d["contact_dist"][:8])
    q[2] -= dist.min() + sink
    d["qpos"][: m.nq] = q
    v = rng.normal(0, vel, m.nv); v[2] = -abs(v[2]) * 0.2
    d["qvel"][: m.nv] = v
    d["ctrl"][: m.nu] = np.array(m.a["key_ctrl"]) + rng.uniform(-0.2, 0.2, m.nu)
    d["qacc_warmstart"][: m.nv] = rng.normal(0, 3.0, m.nv)
    return d


def _cone_min(x, Rn, Rt, mu):
    """argmin 0.5 f^T R f + f^T x over f_n >= 0, |f_t| <= mu f_n, R = diag(Rn, Rt, Rt): closed form (derived from the problem, not from the solver)"""
    xt = np.linalg.norm(x[1:])
    f = -x / np.array([Rn, Rt, Rt])
    if f[0] >= 0 and np.linalg.norm(f[1:]) <= mu * f[0]:
        return f, "interior"
    if x[0] >= mu * xt:                      # x in the dual cone: every feasible f has f . x >= 0
        return np.zeros(3), "zero"
    s = (mu * xt - x[0]) / (Rn + mu * mu * Rt)
    return np.array([s, *(-mu * s * x[1:] / xt)]), "boundary"


def _probe(O, om, d, qacc):
    L = O.lib(False)
    fn = L.lib.odko_solver_probe
    fn.restype = None; fn.argtypes = [C.c_void_p, C.c_void_p] + [C.POINTER(C.c_double)] * 4
    nv = om.nv
    qa = np.ascontiguousarray(qacc, np.float64); cost = np.zeros(1); g = np.zeros(nv); H = np.zeros(nv * nv)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    fn(om.h, d.h, p(qa), p(cost), p(g), p(H))
    return float(cost[0]), g, H.reshape(nv, nv)


def test_compiler_option_and_which_shapes_take_it(robot):
    from open_duck_playground_amd import engine
    m, om = robot
    assert int(m.a["opt_cone"][0]) == 1 and float(m.a["opt_impratio"][0]) == 3.0 and om.L.lib.odko_model_int(om.h, b"cone") == 1
    assert engine.model_reduction(m)["nvr"] == 21           # the third shape's kernels have the cone (tests/test_gpu_parity.py::test_elliptic_cones_in_the_kernels)
    plain = Model.from_xml(os.path.join(ASSETS, "tail_biped.xml"))
    assert int(plain.a["opt_cone"][0]) == 0
    from open_duck_playground_amd.model import load_task_model
    for task in ("flat_terrain", "rough_terrain_backlash"):   # the duck's shapes have their own cone instantiations (plane floor and height field)
        duck = load_task_model(task)
        assert engine.model_reduction(Model({**duck.a, "opt_cone": np.array([1], np.int32)}))["nvr"] == 20


def test_rows_of_an_elliptic_contact(robot, oracle_mod):
    m, om = robot
    d = _standing(oracle_mod, om, m, np.random.default_rng(0))
    d.forward()
    ncon, nc, nefc = d.i("ncon"), d.i("nc"), d.i("nefc")
    assert nc == 3 * ncon and ncon == 12                                   # normal + two tangents per contact (pyramidal: 4)
    r0 = nefc - nc
    J, R, pos, aref = d.J(), d["efc_R"][:nefc], d["efc_pos"][:nefc], d["efc_aref"][:nefc]
    dist = np.array(d["contact_dist"][:ncon]); fr = np.array(d["contact_frame"][: 9 * ncon]).reshape(ncon, 3, 3)
    act = dist < 0
    assert act.sum() >= 4
    for c in range(ncon):
        r = r0 + 3 * c
        np.testing.assert_allclose(R[r + 1: r + 3], R[r] / 3.0, rtol=1e-12)                      # R_t = R_n / impratio
        assert pos[r] == dist[c] and pos[r + 1] == 0 and pos[r + 2] == 0
        if not act[c]:
            assert np.abs(J[r: r + 3]).sum() == 0
            continue
        # the three rows are the contact frame's axes applied to ONE point Jacobian: J_rows = frame @ Jp  =>  frame^T J_rows has the rows of Jp,
        # and the base's translational columns of a point Jacobian are the identity
        Jp = fr[c].T @ J[r: r + 3]
        sgn = 1.0                                                            # geom2 (the foot) minus geom1 (the floor, static)
        np.testing.assert_allclose(Jp[:, :3], sgn * np.eye(3), atol=1e-12)
        vel = J[r: r + 3] @ d["qvel"][: m.nv]
        # aref of a tangent row = -b vel (no position term), the same damping b on both tangents
        assert abs(aref[r + 1] / vel[1] - aref[r + 2] / vel[2]) < 1e-9 * abs(aref[r + 1] / vel[1])


def test_forces_are_the_cone_programs_minimisers(robot, oracle_mod):
    """per contact, after one Newton iteration AND after convergence: efc_force == argmin 0.5 f^T R f + f^T x over the friction cone with the
    user's mu, x = J qacc - aref.  All three regimes must occur (sticking: interior; separating: zero; sliding: boundary)."""
    m, om = robot
    seen = {"interior": 0, "zero": 0, "boundary": 0}
    worst = 0.0
    for iters in (1, 60):
        om.set_int("iterations", iters); om.set_int("ls_iterations", 50 if iters > 1 else 5)
        try:
            rng = np.random.default_rng(3 + iters)
            for _ in range(12):
                d = _standing(oracle_mod, om, m, rng, vel=rng.choice([0.05, 0.4, 1.5]))
                d.forward()
                nefc, nc, ncon = d.i("nefc"), d.i("nc"), d.i("ncon")
                J, R, f = d.J(), d["efc_R"][:nefc], d["efc_force"][:nefc]
                x = J @ d["qacc"][: m.nv] - d["efc_aref"][:nefc]
                for c in range(ncon):
                    if d["contact_dist"][c] >= 0:
                        continue
                    r = nefc - nc + 3 * c
                    mu = float(d["contact_friction"][c])
                    ref, kind = _cone_min(x[r: r + 3], R[r], R[r + 1], mu)
                    seen[kind] += 1
                    scale = max(np.abs(ref).max(), np.abs(x[r: r + 3] / R[r: r + 3]).max(), 1e-9)
                    worst = max(worst, np.abs(f[r: r + 3] - ref).max() / scale)
        finally:
            om.set_int("iterations", 1); om.set_int("ls_iterations", 5)
    assert worst < 1e-9, worst
    assert all(v > 0 for v in seen.values()), seen


def test_the_closed_form_is_the_minimiser(robot):
    """the test's own closed form against brute force: no feasible point within reach has a lower objective"""
    rng = np.random.default_rng(1)
    for _ in range(200):
        Rn = 10 ** rng.uniform(-4, -1); Rt = Rn / rng.choice([1.0, 3.0, 10.0]); mu = rng.uniform(0.3, 1.2)
        x = rng.normal(size=3) * np.array([1.0, 1.5, 1.5])
        f, kind = _cone_min(x, Rn, Rt, mu)
        obj = lambda g: 0.5 * (Rn * g[0] ** 2 + Rt * (g[1] ** 2 + g[2] ** 2)) + g @ x
        assert f[0] >= -1e-15 and np.linalg.norm(f[1:]) <= mu * f[0] + 1e-12
        base = obj(f)
        for _ in range(60):
            g = f + rng.normal(size=3) * (0.2 * np.linalg.norm(f) + 0.05 / Rn * 0.01)
            g[0] = max(g[0], 0.0)
            t = np.linalg.norm(g[1:])
            if t > mu * g[0]:
                g[1:] *= mu * g[0] / t
            assert obj(g) >= base - 1e-9 * max(1.0, abs(base)), (kind, obj(g), base)


def test_converged_solution_satisfies_the_kkt_conditions(robot, oracle_mod):
    """f in K(mu), y = x + R f in the dual cone y_n >= mu |y_t|, f . y = 0 per contact, and M qacc - qfrc_smooth = J^T f for the whole robot"""
    m, om = robot
    om.set_int("iterations", 80); om.set_int("ls_iterations", 50)
    try:
        rng = np.random.default_rng(11)
        converged = 0
        for _ in range(16):
            d = _standing(oracle_mod, om, m, rng, vel=rng.choice([0.05, 0.5]))
            d.forward()
            nefc, nc, ncon, nv = d.i("nefc"), d.i("nc"), d.i("ncon"), m.nv
            J, R, f = d.J(), d["efc_R"][:nefc], d["efc_force"][:nefc]
            x = J @ d["qacc"][:nv] - d["efc_aref"][:nefc]
            # The bracketing line search is MJX's as recalled (shared with the pyramid, where the cost is piecewise quadratic along the
            # search).  On a cone cost whose curvature changes by an order of magnitude along the search its two Newton points can swap
            # sides for ever and the solve stops short of the optimum (2 of these 16 states; emulated with exact derivatives: the same).
            # KKT is judged where the solver says it converged; the cost, gradient and Hessian are pinned separately above / below.
            _, g, _ = _probe(oracle_mod, om, d, np.array(d["qacc"][:nv]))
            if np.abs(g).max() > 1e-5 * max(1.0, np.abs(d["qfrc_smooth"][:nv]).max()):
                continue
            converged += 1
            for c in range(ncon):
                if d["contact_dist"][c] >= 0:
                    continue
                r = nefc - nc + 3 * c
                mu = float(d["contact_friction"][c])
                fc, y = f[r: r + 3], x[r: r + 3] + R[r: r + 3] * f[r: r + 3]
                s = max(np.abs(fc).max() * max(np.abs(x[r: r + 3]).max(), np.abs(R[r: r + 3] * fc).max()), 1e-12)
                assert fc[0] >= -1e-9 and np.linalg.norm(fc[1:]) <= mu * fc[0] * (1 + 1e-7) + 1e-9
                ys = max(np.abs(x[r: r + 3]).max(), np.abs(R[r: r + 3] * fc).max(), 1e-9)      # (a sticking contact has y = 0: judge against the terms it is the difference of)
                assert y[0] - mu * np.linalg.norm(y[1:]) >= -1e-6 * ys
                assert abs(fc @ y) <= 1e-6 * s + 1e-12
            resid = d.M() @ d["qacc"][:nv] - d["qfrc_smooth"][:nv] - J.T @ f
            assert np.abs(resid).max() < 1e-5 * max(1.0, np.abs(d["qfrc_smooth"][:nv]).max())
        assert converged >= 10, converged
    finally:
        om.set_int("iterations", 1); om.set_int("ls_iterations", 5)


def test_gradient_and_hessian_are_the_costs_derivatives(robot, oracle_mod):
    """odko_solver_probe: central differences of the cost give the gradient, of the gradient the Hessian -- middle-zone contacts included
    (the cone Hessian), at points where no contact sits on a zone boundary"""
    m, om = robot
    rng = np.random.default_rng(5)
    nv = m.nv
    checked = 0
    for _ in range(10):
        d = _standing(oracle_mod, om, m, rng, vel=0.6)
        d.forward()
        qacc = np.array(d["qacc"][:nv]) + rng.normal(0, 0.5, nv)
        c0, g0, H0 = _probe(oracle_mod, om, d, qacc)
        eps = 1e-5
        gfd = np.zeros(nv); Hfd = np.zeros((nv, nv))
        for i in range(nv):
            e = np.zeros(nv); e[i] = eps
            cp, gp, _ = _probe(oracle_mod, om, d, qacc + e); cm, gm, _ = _probe(oracle_mod, om, d, qacc - e)
            gfd[i] = (cp - cm) / (2 * eps); Hfd[:, i] = (gp - gm) / (2 * eps)
        gerr = np.abs(gfd - g0).max() / max(np.abs(g0).max(), 1e-9)
        herr = np.abs(Hfd - H0).max() / np.abs(H0).max()
        if herr > 1e-4:      # a contact or a friction-loss row switched zone inside +-eps: not a point to differentiate at
            continue
        assert gerr < 1e-6, gerr
        assert np.abs(H0 - H0.T).max() < 1e-9 * np.abs(H0).max()
        assert np.linalg.eigvalsh(H0).min() > 0
        checked += 1
    assert checked >= 5, checked


def test_newton_with_the_cone_hessian_converges_fast_and_monotonically(robot, oracle_mod):
    m, om = robot
    rng = np.random.default_rng(21)
    d0 = _standing(oracle_mod, om, m, rng, vel=0.5)
    q, v, w, u = (np.array(d0[k][:n]) for k, n in (("qpos", m.nq), ("qvel", m.nv), ("qacc_warmstart", m.nv), ("ctrl", m.nu)))
    costs = []
    for iters in (1, 2, 3, 5, 8, 12, 40):
        om.set_int("iterations", iters); om.set_int("ls_iterations", 50)
        d = oracle_mod.OracleData(om)
        d["qpos"][: m.nq] = q; d["qvel"][: m.nv] = v; d["qacc_warmstart"][: m.nv] = w; d["ctrl"][: m.nu] = u
        d.forward()
        costs.append(float(d["solver_cost1"][0]))
    om.set_int("iterations", 1); om.set_int("ls_iterations", 5)
    assert all(b <= a + 1e-12 * abs(a) for a, b in zip(costs, costs[1:])), costs
    assert abs(costs[-2] - costs[-1]) < 1e-9 * max(1.0, abs(costs[-1])), costs          # 12 iterations are converged


def _yawed(q, v, ang, nq, nv):
    c, s = np.cos(ang), np.sin(ang)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    q2, v2 = q.copy(), v.copy()
    q2[:3] = Rz @ q[:3]; v2[:3] = Rz @ v[:3]                       # base position / world-frame linear velocity
    qa = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)]); qb = q[3:7]
    q2[3:7] = [qa[0]*qb[0] - qa[3]*qb[3], qa[0]*qb[1] - qa[3]*qb[2], qa[0]*qb[2] + qa[3]*qb[1], qa[0]*qb[3] + qa[3]*qb[0]]
    return q2, v2                                                  # (angular velocity is body-frame, joints are relative: unchanged)


def test_an_elliptic_cone_has_no_preferred_direction(robot, oracle_mod):
    """the same contact-rich state turned about the vertical by an arbitrary angle steps to the same result in the robot's frame (the cost
    depends on |x_t| alone); with the pyramid the same turn changes the result -- the test has teeth"""
    m, om = robot
    rng = np.random.default_rng(31)
    nq, nv = m.nq, m.nv
    for cone, expect_equal in ((1, True), (0, False)):
        om.set_int("cone", cone)
        try:
            worst = 0.0
            for _ in range(6):
                d = _standing(oracle_mod, om, m, rng, vel=0.6)
                q, v, w, u = (np.array(d[k][:n]) for k, n in (("qpos", nq), ("qvel", nv), ("qacc_warmstart", nv), ("ctrl", m.nu)))
                out = []
                for ang in (0.0, 0.6435):
                    qq, vv = _yawed(q, v, ang, nq, nv)
                    _, ww = _yawed(q, w, ang, nq, nv)                   # (the warm start's base translation is a world-frame vector too)
                    dd = oracle_mod.OracleData(om)
                    dd["qpos"][:nq] = qq; dd["qvel"][:nv] = vv; dd["qacc_warmstart"][:nv] = ww; dd["ctrl"][: m.nu] = u
                    dd.step()       # (ONE step: measured 1e-14.  Over three steps 1e-3 -- the bracketing line search takes discrete decisions, and one decided by the last bit is amplified by the next steps)
                    qo, vo = np.array(dd["qpos"][:nq]), np.array(dd["qvel"][:nv])
                    qb, vb = _yawed(qo, vo, -ang, nq, nv)
                    out.append(np.concatenate([qb[2:], vb]))           # (x, y of the base turn with the frame; compare height, attitude, joints, velocities)
                worst = max(worst, np.abs(out[0] - out[1]).max())
            if expect_equal:
                assert worst < 1e-8, worst
            else:
                assert worst > 1e-5, worst
        finally:
            om.set_int("cone", 1)
