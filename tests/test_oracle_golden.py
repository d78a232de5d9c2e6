"""Pins the oracle's reward / reference-motion halves against fixtures generated from the
reference's numpy mirrors (tools/make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _midpoint_tie(prm_arrays, q):
    """True when a query sits (to rounding) on the midpoint between two grid points: there the float32
    and float64 argmin legitimately differ; such commands have measure zero in training."""
    for v, grid, rng in ((q[0], prm_arrays["dxs"], prm_arrays["dx_range"]), (q[1], prm_arrays["dys"], prm_arrays["dy_range"]),
                         (q[2], prm_arrays["dthetas"], prm_arrays["dtheta_range"])):
        d = np.sort(np.abs(grid - np.clip(v, rng[0], rng[1])))
        if d[1] - d[0] < 1e-6:
            return True
    return False


def test_reference_motion_float64_matches_reference(oracle_mod, prm_arrays):
    g = np.load(os.path.join(GOLDEN, "reference_motion.npz"))
    prm = oracle_mod.OraclePRM(prm_arrays)
    assert prm.nsteps == int(g["nb_steps_in_period"][0]) == 27
    for q, exp, idx in zip(g["query"], g["expected"], g["index"]):
        got = prm.eval64(q[0], q[1], q[2], int(q[3]))
        np.testing.assert_allclose(got, exp, rtol=1e-9, atol=1e-9)
        if not _midpoint_tie(prm_arrays, q):  # float32 index path (what the reference's jnp arrays do)
            assert prm.index(q[0], q[1], q[2]) == tuple(idx)


def test_reference_motion_known_answer(oracle_mod, prm_arrays):
    # get_reference_motion(0.1, 0.0, 0.3, 5)[:5], nearest grid index (3,1,5)
    prm = oracle_mod.OraclePRM(prm_arrays)
    got = prm.eval64(0.1, 0.0, 0.3, 5)[:5]
    np.testing.assert_allclose(got, [0.00160981, 0.10700139, -0.80862226, 1.38789204, -0.64949825], atol=1e-7)
    assert prm.index(0.1, 0.0, 0.3) == (3, 1, 5)


def test_reference_motion_float32_path(oracle_mod, prm_arrays):
    """The shipped path (fp32 table, fp32 fma Horner, as the reference's jnp arrays) stays within the
    conditioning bound of the degree-15 polynomials (coefficients up to 2e5)."""
    g = np.load(os.path.join(GOLDEN, "reference_motion.npz"))
    prm = oracle_mod.OraclePRM(prm_arrays)
    worst = 0.0
    for q, exp in zip(g["query"], g["expected"]):
        if _midpoint_tie(prm_arrays, q):
            continue
        got = prm.eval(q[0], q[1], q[2], int(q[3]))
        worst = max(worst, float(np.abs(got - exp).max()))
    assert worst < 5e-2


def test_rewards_match_reference(oracle_mod):
    g = np.load(os.path.join(GOLDEN, "rewards.npz"))
    L = oracle_mod.lib()
    a, p = L.arr, L.ptr
    n = len(g["cmd"])
    sigma = float(g["sigma"][0])
    for i in range(n):
        cmd, lv, gy = a(g["cmd"][i]), a(g["local_vel"][i]), a(g["gyro"][i])
        tq, act, last = a(g["torques"][i]), a(g["act"][i]), a(g["last_act"][i])
        jq, jv, dp = a(g["jq"][i]), a(g["jv"][i]), a(g["default_pose"])
        bq, bv, ct, ref = a(g["base_qpos"][i]), a(g["base_qvel"][i]), a(g["contacts"][i]), a(g["ref"][i])
        tol = dict(rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(L.lib.odko_reward_tracking_lin_vel(p(cmd), p(lv), sigma), g["tracking_lin_vel"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_reward_tracking_ang_vel(p(cmd), p(gy), sigma), g["tracking_ang_vel"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_cost_torques(p(tq), 14), g["torques_cost"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_cost_action_rate(p(act), p(last), 14), g["action_rate"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_cost_stand_still(p(cmd), p(jq), p(jv), p(dp), 14), g["stand_still"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_reward_imitation(p(bq), p(bv), p(jq), p(jv), p(ct), p(ref), p(cmd)), g["imitation"][i],
                                   rtol=1e-11, atol=1e-11)
        # Standing terms (reference standing.py:585-606)
        up, ch = a(g["upvector"][i]), a(g["cmd_head"][i])
        np.testing.assert_allclose(L.lib.odko_cost_orientation(p(up)), g["orientation"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_cost_head_pos(p(jq), p(ch)), g["head_pos"][i], **tol)
        np.testing.assert_allclose(L.lib.odko_cost_stand_still_legs(p(ch), p(jq), p(jv), p(dp), 14), g["stand_still_legs"][i], **tol)
    assert g["alive"][0] == 1.0
    assert (g["head_pos"][:6] == 0).all() and (g["head_pos"][24:] == 0).all() and (g["head_pos"][12:24] > 0).all()   # move-command gate
    assert np.isnan(g["upvector"][30]).all() and g["orientation"][30] == 0.0
    assert np.isnan(g["torques"][20]).any() and g["torques_cost"][20] == 0.0  # nan_to_num branch is in the fixture
