"""GPU parity: HIP kernels (through the C-ABI, libodk.so) vs the float64 CPU oracle on the same inputs.

Tolerance: BASELINE.json north_star asks qpos/qvel within 1e-4 relative (fp32) after one step.  The
oracle itself is unpinned against MJX (no install available), so these tests pin the
kernels to the oracle, and the oracle to analytic invariants (test_oracle_physics.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RTOL_Q = 1e-4   # qpos / qvel after one mjx.step (north_star)

# Bounds = min(north_star, ~3x the worst case measured on MI355X at the start of round 3 -- profiles/r3/parity_worst.json keeps
# the measured values of the last run next to these bounds).  Relative errors use the floors given in the tests.
STAGE_BOUNDS = dict(xpos=3e-7, M=4e-4, qfs=2e-4, qas=1.5e-4, dist=4e-7, D=3e-4, aref=1.2e-3, qacc=3.5e-3, sens=5e-4, qpos=1e-5, qvel=4e-5,
                    qpos_normwise=3e-7, qvel_normwise=1.5e-5)      # the floor-free measure (_nw): ~3 x the measured 9e-8 / 4e-6; north star 1e-4
TEN_BOUNDS = dict(qpos=RTOL_Q, qvel=RTOL_Q, qpos_normwise=3e-6, qvel_normwise=5e-5)      # norm-wise: ~3 x the measured 7e-7 / 1.6e-5
FOOT_BOUNDS = dict(dist=3e-7, qacc=3e-4, qpos=1e-5, qvel=1e-5)


# the oracle's discrete decisions a test can bias (oracle/odk_oracle.c "Tie bias"; the same table as tests/test_gpu_env.py TIE_CLASSES)
ILL_CLASSES = ((4, "edge_or_face_contact"), (8, "incident_face"), (1, "separating_face"), (2, "reference_polytope"), (16, "clipping_plane_side"),
               (32, "manifold_argmax"), (64, "fourth_deepest_cut"), (128, "warm_start_pick"), (256, "line_search_bracket_end"), (512, "line_search_comparison"),
               (1024, "area_zero_cut"))


def _rel(a, b, floor=1e-3):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def _nw(a, b):
    """Norm-wise relative error of one env's vector, NO floor: max |a - b| / max |b| (the component-wise `_rel` above divides a
    velocity component below 1 rad/s by 1: for those it is an absolute bound; this is the measure without that allowance)."""
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _random_states(model, n, rng, airborne_frac=0.3):
    nq, nv = model.nq, model.nv
    qpos = np.tile(np.asarray(model.a["key_qpos"], np.float64), (n, 1))
    qvel = np.zeros((n, nv))
    for e in range(n):
        air = rng.uniform() < airborne_frac
        qpos[e, 0:2] += rng.uniform(-0.05, 0.05, 2)
        qpos[e, 2] = rng.uniform(0.3, 0.6) if air else rng.uniform(0.135, 0.17)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.25, 0.25) if not air else rng.uniform(-1.0, 1.0)
        qpos[e, 3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        for j in range(1, model.njnt):
            a = model.a["jnt_qposadr"][j]
            lo, hi = model.a["jnt_range"][j]
            if hi - lo < 0.05:   # backlash joints
                qpos[e, a] = rng.uniform(lo, hi)
            else:
                qpos[e, a] = np.clip(qpos[e, a] + rng.uniform(-0.3, 0.3), lo - 0.02, hi + 0.02)
        qvel[e, :3] = rng.normal(0, 0.3, 3)
        qvel[e, 3:6] = rng.normal(0, 1.0, 3)
        qvel[e, 6:] = rng.normal(0, 2.0, nv - 6)
    return qpos, qvel


def _contacts(d, ncon=12):
    """active contacts of the last forward pass: [(pair, dist, pos)]"""
    cd = np.array(d["contact_dist"][:ncon]); cp = np.array(d["contact_pos"][: 3 * ncon]).reshape(ncon, 3)
    return [(c // 4, cd[c], cp[c]) for c in range(ncon) if cd[c] < 0]


def _same_contacts(a, b, dtol=2e-5, ptol=2e-4):
    """the two contact lists hold the same contacts (as multisets per geom pair, up to dtol / ptol)"""
    if len(a) != len(b):
        return False
    left = list(b)
    for pr, dist, pos in a:
        hit = [k for k, (pr2, d2, p2) in enumerate(left) if pr2 == pr and abs(d2 - dist) < dtol and np.abs(p2 - pos).max() < ptol]
        if not hit:
            return False
        left.pop(hit[0])
    return True


def _contact_tie(O, om, qpos, qvel, ctrl, rng, nominal, k=8):
    """True when the ORACLE's own contact set is not stable under rounding-level noise on the state (1e-6 relative): a manifold
    arg-max, the choice of the reference face, or the cut of the four deepest height-field contacts sits on a tie.  Neighbouring
    prisms share edges and vertices, so candidates with equal depth are the rule there, not the exception; no float32
    implementation can be expected to resolve such a tie the way float64 does."""
    for _ in range(k):
        d = O.OracleData(om)
        d["qpos"][: om.nq] = qpos + 1e-6 * rng.standard_normal(om.nq) * np.maximum(np.abs(qpos), 0.1)
        d["qvel"][: om.nv] = qvel; d["ctrl"][: len(ctrl)] = ctrl
        d.forward()
        if not _same_contacts(nominal, _contacts(d)):
            return True
    return False


def _settle_on_terrain(O, om, qpos, rng, airborne):
    """moves each base height so that the deepest contact is 0.3 ... 3 mm (what a standing / walking robot sees: the solver
    holds penetrations below a millimetre), instead of the centimetres a random joint pose at a fixed height gives"""
    for e in range(len(qpos)):
        if airborne[e]:
            continue
        target = rng.uniform(3e-4, 3e-3)
        for _ in range(3):
            d = O.OracleData(om)
            d["qpos"][: om.nq] = qpos[e]
            d.forward()
            cd = np.array(d["contact_dist"][:8])
            deepest = cd.min() if (cd < 0).any() else None
            if deepest is None:
                qpos[e, 2] -= 0.004
            else:
                qpos[e, 2] += -deepest - target
    return qpos


def _oracle_step(O, om, qpos, qvel, warm, ctrl, nsub):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos; d["qvel"][: om.nv] = qvel; d["qacc_warmstart"][: om.nv] = warm
    d.env_physics_step(ctrl, nsub)
    return d


@pytest.mark.parametrize("task,lanes", [("flat_terrain", 32), ("flat_terrain", 64), ("flat_terrain_backlash", 32), ("rough_terrain_backlash", 32)])
def test_one_substep_stages(torch_cuda, oracle_mod, parity_log, task, lanes):
    """One mjx.step from random states: every comparable intermediate and the integrated state."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    n = 48
    rng = np.random.default_rng(7)
    qpos, qvel = _random_states(model, n, rng)
    om = oracle_mod.OracleModel(model.blob())
    if "rough" in task:
        qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, qpos[:, 2] > 0.25)
    warm = rng.normal(0, 5.0, (n, model.nv))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.4, 0.4, (n, 14))
    cfg = engine.default_config(); cfg.lanes_per_env = lanes
    b = engine.Batch(model, n, cfg)
    b.set_state(qpos, qvel, warm)
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    gq, gv, gw = b.get_state()
    img = b.lds_image()
    nv, nb = model.nv, model.nbody
    worst = dict(xpos=0, M=0, qfs=0, qas=0, dist=0, D=0, aref=0, qacc=0, qpos=0, qvel=0, sens=0, qpos_normwise=0, qvel_normwise=0)
    o = {k: b.lds_offset(k) for k in ("xpos", "M", "qfrc_smooth", "qacc_smooth", "contact_dist", "efc_D", "efc_aref", "qacc", "sensordata", "actuator_force")}
    from open_duck_playground_amd.tables import build_kernel_tables, reduced_layout
    tabs = build_kernel_tables(model.a)
    red = reduced_layout(model.a)     # the kernels keep the inertia on the reduced (backlash twins merged) tree
    Mi, Mj = red["ei"], red["ej"]
    if task == "flat_terrain":
        assert np.array_equal(Mi, tabs["k_M_i"]) and np.array_equal(Mj, tabs["k_M_j"])   # no twins: the model's own layout
    else:
        assert len(Mi) == 145 and (red["twin"] >= 0).sum() == 10
    nfl, nlim = len(tabs["k_fl_dof"]), len(tabs["k_lim_jnt"])
    prng = np.random.default_rng(99)
    n_tie = 0
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        L = img[e]
        xpos = L[o["xpos"]: o["xpos"] + 3 * nb].reshape(3, nb).T
        worst["xpos"] = max(worst["xpos"], np.abs(xpos - d["xpos"][: 3 * nb].reshape(nb, 3)).max())
        Md = d.M()
        worst["M"] = max(worst["M"], _rel(L[o["M"]: o["M"] + len(Mi)], Md[Mi, Mj], 1e-4).max())
        worst["qfs"] = max(worst["qfs"], _rel(L[o["qfrc_smooth"]: o["qfrc_smooth"] + nv], d["qfrc_smooth"][:nv], 1e-2).max())
        worst["qas"] = max(worst["qas"], _rel(L[o["qacc_smooth"]: o["qacc_smooth"] + nv], d["qacc_smooth"][:nv], 1.0).max())
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1      # everything downstream of the contact set is undefined at float32 resolution for this state
            continue
        cd_g, cd_o = L[o["contact_dist"]: o["contact_dist"] + 8], d["contact_dist"][:8]
        act = (cd_o < 0) | (cd_g < 0)
        if act.any():
            worst["dist"] = max(worst["dist"], np.abs(cd_g[act] - cd_o[act]).max())
        nefc = d.i("nefc")
        D_o, aref_o = d["efc_D"][:nefc].copy(), d["efc_aref"][:nefc].copy()
        D_g, aref_g = L[o["efc_D"]: o["efc_D"] + nefc], L[o["efc_aref"]: o["efc_aref"] + nefc]
        # rows the kernel leaves structurally inactive have D = 0; the oracle marks them by a zero Jacobian row
        J = d.J()
        live = np.abs(J).sum(axis=1) > 0
        assert ((D_g > 0) == live)[nfl:].all(), f"env {e}: active row sets differ"
        worst["D"] = max(worst["D"], _rel(D_g[live], D_o[live], 1e-6).max())
        worst["aref"] = max(worst["aref"], _rel(aref_g[live], aref_o[live], 1.0).max())
        worst["qacc"] = max(worst["qacc"], _rel(L[o["qacc"]: o["qacc"] + nv], d["qacc"][:nv], 5.0).max())
        worst["sens"] = max(worst["sens"], _rel(L[o["sensordata"]: o["sensordata"] + 46], d["sensordata"][:46], 1.0).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 1)
        worst["qpos"] = max(worst["qpos"], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
        worst["qvel"] = max(worst["qvel"], _rel(gv[e], ds["qvel"][:nv], 1.0).max())
        worst["qpos_normwise"] = max(worst["qpos_normwise"], _nw(gq[e], ds["qpos"][: om.nq]))
        worst["qvel_normwise"] = max(worst["qvel_normwise"], _nw(gv[e], ds["qvel"][:nv]))
    print(task, lanes, {k: float(f"{v:.3g}") for k, v in worst.items()}, "contact ties:", n_tie, "of", n)
    b.close()
    # ties (the oracle's own contact set not stable under rounding-level noise): rare since the last manifold point resolves the
    # triangle tie by rule (oracle manifold_points AREA_TIE; 21 % of the rough-terrain poses before, none now)
    parity_log.check(f"one_mjx_step/{task}/lanes{lanes}", dict(STAGE_BOUNDS, tie_fraction=0.15 if "rough" in task else 0.1), tie_fraction=n_tie / n, **worst)


def test_height_field_up_normals_option(torch_cuda, oracle_mod, parity_log):
    """odk_env_config.hfield_up_normals_only (BUILD-DEFINED opt-in, default off): a prism pair's contacts count only when its
    normal points up.  Against the oracle's hfield_mode 3 on feet pressed 0.3 ... 8 mm into the terrain at tilts up to 0.6 rad (where
    prism side faces win the separating-axis test): contact distances, one mjx.step and ten; and the option must CHANGE the contacts of
    some of these states (a kernel that ignored it would still match the default oracle), switch on and off through
    odk_batch_set_config, and leave the default path bit-identical."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model("rough_terrain_backlash")
    n = 96
    rng = np.random.default_rng(31)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.0)
    for e in range(n):      # larger tilts than the standing-ish default: a tilted sole overlaps neighbouring prisms at its rim
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(-0.6, 0.6)
        qpos[e, 3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    om0 = oracle_mod.OracleModel(model.blob())
    om3 = om0.copy(); om3.set_int("hfield_mode", 3)
    qpos = _settle_on_terrain(oracle_mod, om0, qpos, rng, np.zeros(n, bool))
    qpos[:, 2] -= rng.uniform(0.0, 0.015, n)        # ... then pressed in by up to 1.5 cm, like the reset states of the task (joystick.py:206-258 ignores the elevation)
    qvel *= 0.3
    warm = np.zeros((n, model.nv))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.2, 0.2, (n, 14))
    ctrl_t = torch.tensor(ctrl, dtype=torch.float32, device="cuda")
    cfg = engine.default_config(); cfg.hfield_up_normals_only = 1
    b = engine.Batch(model, n, cfg)
    b0 = engine.Batch(model, n)
    o_cd = b.lds_offset("contact_dist")
    res = {}
    for name, bb in (("on", b), ("off", b0)):
        bb.set_state(qpos, qvel, warm); bb.physics_step(ctrl_t, 1)
        res[name] = (bb.lds_image()[:, o_cd: o_cd + 8].copy(),) + bb.get_state()[:2]
    W = dict(dist=0.0, qpos=0.0, qvel=0.0)
    prng = np.random.default_rng(3)
    n_changed = n_tie = 0
    for e in range(n):
        d3, d0 = oracle_mod.OracleData(om3), oracle_mod.OracleData(om0)
        for d in (d3, d0):
            d["qpos"][: om0.nq] = qpos[e]; d["qvel"][: om0.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
            d.forward()
        c3, c0 = np.array(d3["contact_dist"][:8]), np.array(d0["contact_dist"][:8])
        n_changed += int(not np.allclose(np.sort(np.minimum(c3, 0)), np.sort(np.minimum(c0, 0)), atol=1e-6))
        if _contact_tie(oracle_mod, om3, qpos[e], qvel[e], ctrl[e], prng, _contacts(d3)) or _contact_tie(oracle_mod, om0, qpos[e], qvel[e], ctrl[e], prng, _contacts(d0)):
            n_tie += 1
            continue
        for name, c_o, om in (("on", c3, om3), ("off", c0, om0)):
            cd_g, gq, gv = res[name][0][e], res[name][1][e], res[name][2][e]
            act = (c_o < 0) | (cd_g < 0)
            if act.any():
                W["dist"] = max(W["dist"], np.abs(cd_g[act] - c_o[act]).max())
            ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 1)
            W["qpos"] = max(W["qpos"], _rel(gq, ds["qpos"][: om.nq], 1e-2).max()); W["qvel"] = max(W["qvel"], _rel(gv, ds["qvel"][: om.nv], 1.0).max())
    assert n_changed >= n // 12, (n_changed, n)       # the option matters on these poses (12 of 96 measured)
    # switching through odk_batch_set_config: the default batch turned on reproduces the opt-in batch bit for bit, and back
    c_on = engine.default_config(); c_on.hfield_up_normals_only = 1
    b0.set_config(c_on); b0.set_state(qpos, qvel, warm); b0.physics_step(ctrl_t, 1)
    assert np.array_equal(b0.lds_image()[:, o_cd: o_cd + 8], res["on"][0]) and np.array_equal(b0.get_state()[0], res["on"][1])
    b0.set_config(engine.default_config()); b0.set_state(qpos, qvel, warm); b0.physics_step(ctrl_t, 1)
    assert np.array_equal(b0.lds_image()[:, o_cd: o_cd + 8], res["off"][0]) and np.array_equal(b0.get_state()[0], res["off"][1])
    b.close(); b0.close()
    # (dist: these feet are pressed in by up to 1.8 cm, five times deeper than the standing-ish states of the stage test: 7e-7 measured)
    parity_log.rec("hfield_up_normals_option", None, states=n, states_whose_contacts_change=n_changed)
    parity_log.check("hfield_up_normals_option", dict(dist=1.5e-6, qpos=STAGE_BOUNDS["qpos"], qvel=STAGE_BOUNDS["qvel"], tie_fraction=0.3), tie_fraction=n_tie / n, **W)


def test_height_field_far_from_the_origin(torch_cuda, oracle_mod, parity_log):
    """The terrain spans +-10 m and a contact depth is a fraction of a millimetre: the kernel works relative to the first grid
    corner of each foot's window, so contacts 7-9 m from the origin must agree with the float64 oracle as well as those next to it."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model("rough_terrain_backlash")
    n = 32
    rng = np.random.default_rng(21)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.0)
    qpos[:, 0] += rng.choice([-1.0, 1.0], n) * rng.uniform(7.0, 9.0, n); qpos[:, 1] += rng.choice([-1.0, 1.0], n) * rng.uniform(7.0, 9.0, n)
    om = oracle_mod.OracleModel(model.blob())
    qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, np.zeros(n, bool))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.2, 0.2, (n, 14))
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel, np.zeros((n, model.nv)))
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    img = b.lds_image()
    o_cd, o_qa = b.lds_offset("contact_dist"), b.lds_offset("qacc")
    prng = np.random.default_rng(5)
    wd = wa = 0.0
    n_tie = 0
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1
            continue
        cd_o, cd_g = np.array(d["contact_dist"][:8]), img[e][o_cd: o_cd + 8]
        act = (cd_o < 0) | (cd_g < 0)
        assert act.any()
        wd = max(wd, np.abs(cd_g[act] - cd_o[act]).max())
        wa = max(wa, _rel(img[e][o_qa: o_qa + model.nv], d["qacc"][: model.nv], 5.0).max())
    b.close()
    # the base position itself is a float32 in the state record: 8 m carries 5e-7 m of rounding, which is the floor here
    parity_log.check("hfield_far_from_origin", dict(dist=6e-7, qacc=1e-3, tie_fraction=0.15), dist=wd, qacc=wa, tie_fraction=n_tie / n)


def _leaning_states(oracle_mod, model, om, n, rng):
    """per pair of envs (= per wave of the G = 32 kernels): one robot leaning on ONE foot pressed 2-8 mm into the terrain (or standing on
    both), its neighbour in the air"""
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.0)
    kinds = []
    for e in range(n):
        kind = ("one", "air") if (e // 2) % 3 == 0 else (("air", "one") if (e // 2) % 3 == 1 else ("both", "air"))
        kinds.append(kind[e % 2])
        qpos[e, 0:2] = rng.uniform(-6.0, 6.0, 2)
        if kinds[e] == "air":
            qpos[e, 2] = 0.8
            continue
        lean = rng.uniform(0.25, 0.4) * rng.choice([-1.0, 1.0]) if kinds[e] == "one" else rng.uniform(-0.03, 0.03)
        qpos[e, 3:7] = [np.cos(lean / 2), np.sin(lean / 2), 0.0, 0.0]       # roll about x: the robot leans on one foot
        qpos[e, 7:] = np.asarray(model.a["key_qpos"])[7:]
        qpos[e, 2] = 0.5
        target = rng.uniform(2e-3, 8e-3)                                   # pressed in deep: many prisms within reach
        for _ in range(14):                                                # (separated prisms report their positive separation)
            d = oracle_mod.OracleData(om)
            d["qpos"][: om.nq] = qpos[e]; d.forward()
            cd = np.array(d["contact_dist"][:8])
            qpos[e, 2] -= cd.min() + target if (cd < 0).any() else max(0.8 * cd.min(), 2e-3)
    return qpos, qvel, kinds


def test_height_field_lists_shared_across_the_wave(torch_cuda, oracle_mod, parity_log):
    """The four 16-lane rows of a wave (two envs x two feet) share their prism lists: a row with nothing open takes prisms of the
    foot with the most left.  Here the lists are as unequal as they get -- per wave one robot leaning on ONE foot pressed into the
    terrain (or standing on both), its neighbour in the air -- so that up to four rows work one foot's list at once (ranks 0 ... 3,
    merges rank by rank, hull registers reloaded across the two env images)."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model("rough_terrain_backlash")
    om = oracle_mod.OracleModel(model.blob())
    n = 64
    rng = np.random.default_rng(77)
    qpos, qvel, kinds = _leaning_states(oracle_mod, model, om, n, rng)
    ctrl = np.tile(np.asarray(model.a["key_ctrl"]), (n, 1))
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel * 0.2, np.zeros((n, model.nv)))
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    gq, gv, _ = b.get_state()
    img = b.lds_image()
    o_cd, o_qa = b.lds_offset("contact_dist"), b.lds_offset("qacc")
    prng = np.random.default_rng(5)
    W = dict(dist=0.0, qacc=0.0, qpos=0.0, qvel=0.0)
    n_tie = n_one = n_judged = 0
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = 0.2 * qvel[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        cd_o, cd_g = np.array(d["contact_dist"][:8]), img[e][o_cd: o_cd + 8]
        if kinds[e] == "air":
            assert not (cd_o < 0).any() and not (cd_g < 0).any(), e
            continue
        feet_on = [(cd_o[4 * f: 4 * f + 4] < 0).any() for f in (0, 1)]
        n_one += int(sum(feet_on) == 1)
        if _contact_tie(oracle_mod, om, qpos[e], 0.2 * qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1
            continue
        n_judged += 1
        act = (cd_o < 0) | (cd_g < 0)
        assert act.any(), e
        W["dist"] = max(W["dist"], np.abs(cd_g[act] - cd_o[act]).max())
        W["qacc"] = max(W["qacc"], _rel(img[e][o_qa: o_qa + model.nv], d["qacc"][: model.nv], 5.0).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], 0.2 * qvel[e], np.zeros(model.nv), ctrl[e], 1)
        W["qpos"] = max(W["qpos"], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
        W["qvel"] = max(W["qvel"], _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
    b.close()
    assert n_one >= n // 6 and n_judged >= n // 6, (n_one, n_judged, n_tie)
    parity_log.check("hfield_shared_lists", dict(dist=6e-7, qacc=3.5e-3, qpos=1e-5, qvel=4e-5, tie_fraction=0.15), tie_fraction=n_tie / (n - kinds.count("air")), **W)


def test_height_field_neighbour_with_touching_feet(torch_cuda, oracle_mod, parity_log):
    """The foot-foot routine runs for the whole wave as soon as one of its two envs has overlapping feet, and copies both hulls into the
    row arrays.  Whatever the height-field contacts of the OTHER env left for the row phase must not live there (the first version kept
    the per-contact frames in the jv rows: three of the neighbour's floor contacts got a hull vertex for a frame).  Per wave: one robot
    standing on the terrain, its neighbour in the air with the feet pressed together."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model("rough_terrain_backlash")
    om = oracle_mod.OracleModel(model.blob())
    n = 48
    rng = np.random.default_rng(91)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.0)
    aq = build_tables(model)["k_act_qposadr"]
    for e in range(n):
        if e % 2 == 1:
            qpos[e] = np.asarray(model.a["key_qpos"]); qpos[e, 2] = 0.5
            qpos[e, int(aq[1])] = rng.uniform(0.4, 0.6); qpos[e, int(aq[10])] = rng.uniform(-0.6, -0.4); qpos[e, int(aq[0])] += rng.uniform(-0.3, 0.3)
        else:
            qpos[e, 0:2] = rng.uniform(-6.0, 6.0, 2)
            qpos[e, 7:] = np.asarray(model.a["key_qpos"])[7:] + rng.uniform(-0.03, 0.03, model.nq - 7) * (np.asarray(model.a["key_qpos"])[7:] != 0)
            qpos[e, 2] = 0.4
            target = rng.uniform(5e-4, 4e-3)
            for _ in range(14):
                d = oracle_mod.OracleData(om)
                d["qpos"][: om.nq] = qpos[e]; d.forward()
                cd = np.array(d["contact_dist"][:8])
                qpos[e, 2] -= cd.min() + target if (cd < 0).any() else max(0.8 * cd.min(), 2e-3)
    ctrl = np.tile(np.asarray(model.a["key_ctrl"]), (n, 1))
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel, np.zeros((n, model.nv)))
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    gq, gv, _ = b.get_state()
    img = b.lds_image(); o_cd = b.lds_offset("contact_dist")
    b.close()
    prng = np.random.default_rng(92)
    W = dict(dist=0.0, qpos=0.0, qvel=0.0)
    n_ff = n_floor = n_tie = 0
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        cd_o, cd_g = np.array(d["contact_dist"][:12]), img[e][o_cd: o_cd + 12]
        n_ff += int(e % 2 == 1 and (cd_o[8:] < 0).any()); n_floor += int(e % 2 == 0 and (cd_o[:8] < 0).any())
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1
            continue
        act = (cd_o < 0) | (cd_g < 0)
        if act.any():
            W["dist"] = max(W["dist"], np.abs(cd_g[act] - cd_o[act]).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], np.zeros(model.nv), ctrl[e], 1)
        W["qpos"] = max(W["qpos"], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
        W["qvel"] = max(W["qvel"], _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
    assert n_ff >= n // 3 and n_floor >= n // 3, (n_ff, n_floor)
    parity_log.check("hfield_neighbour_touching_feet", dict(dist=6e-7, qpos=1e-5, qvel=4e-5, tie_fraction=0.15), tie_fraction=n_tie / n, **W)


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_env_step_ten_substeps(torch_cuda, oracle_mod, parity_log, task):
    """mjx_env.step (10 substeps) from standing-ish states: state after one env step within 1e-4 relative.  448 states per model
    (round 3 ran 32 and judged ~20 of them: VERDICT r3 weak 2i), of which >= 150 must be judged."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    n = 448
    rng = np.random.default_rng(11)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.2)
    om = oracle_mod.OracleModel(model.blob())
    if "rough" in task:
        qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, qpos[:, 2] > 0.25)
    qvel *= 0.3
    warm = np.zeros((n, model.nv))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.2, 0.2, (n, 14))
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel, warm)
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 10)
    gq, gv, _ = b.get_state()
    wq = wv = wqn = wvn = 0.0
    prng = np.random.default_rng(98)
    n_ill = n_classified = 0
    why_ill = {}
    for e in range(n):
        d = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 10)
        q1, v1 = np.array(d["qpos"][: om.nq]), np.array(d["qvel"][: om.nv])
        # ill-conditioned start state: the oracle's own result moves by more than half the bound when its input moves by 1e-6
        # (a contact tie or a line-search bracket flips somewhere in the ten substeps): set aside, counted
        ill = False
        for _ in range(8):
            qp = qpos[e] + 1e-6 * prng.standard_normal(om.nq) * np.maximum(np.abs(qpos[e]), 0.1)
            vp = qvel[e] + 5e-6 * prng.standard_normal(om.nv) * np.maximum(np.abs(qvel[e]), 1.0)
            dp = _oracle_step(oracle_mod, om, qp, vp, warm[e], ctrl[e], 10)
            if _rel(dp["qpos"][: om.nq], q1, 1e-2).max() > 0.5 * TEN_BOUNDS["qpos"] or _rel(dp["qvel"][: om.nv], v1, 1.0).max() > 0.5 * TEN_BOUNDS["qvel"]:
                ill = True
        if ill:
            n_ill += 1
            if n_ill <= 64:     # WHY it is ill-conditioned: which class of the oracle's discrete decisions, biased to its runner-up
                                # inside a 2e-6 band for the whole step, moves the oracle's own result by more than half a bound
                hit = []
                for bit, name in ILL_CLASSES:
                    oracle_mod.set_tie_bias(bit, 2e-6, 1e-4)
                    try:
                        db = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 10)
                    finally:
                        oracle_mod.set_tie_bias(0)
                    if _rel(db["qpos"][: om.nq], q1, 1e-2).max() > 0.5 * TEN_BOUNDS["qpos"] or _rel(db["qvel"][: om.nv], v1, 1.0).max() > 0.5 * TEN_BOUNDS["qvel"]:
                        hit.append(name)
                for name in hit or ["no_tie_class (a row switching on / off: |dist|, |limit|, J a - aref near zero)"]:
                    why_ill[name] = why_ill.get(name, 0) + 1
                n_classified += 1
            continue
        wq = max(wq, _rel(gq[e], q1, 1e-2).max())
        wv = max(wv, _rel(gv[e], v1, 1.0).max())
        wqn, wvn = max(wqn, _nw(gq[e], q1)), max(wvn, _nw(gv[e], v1))
    print(task, "10 substeps: worst rel qpos", wq, "qvel", wv, "ill-conditioned:", n_ill, "of", n)
    print(task, "ill-conditioned states by the decision class that moves the oracle's own result (of", n_classified, "examined; a state may name several):", why_ill)
    parity_log.rec(f"ten_substeps/{task}", None, ill_examined=n_classified, **{"ill_by_" + k.split(" ")[0]: v for k, v in why_ill.items()})
    b.close()
    assert n - n_ill >= 150, (n, n_ill)
    parity_log.rec(f"ten_substeps/{task}", None, states=n, judged=n - n_ill)
    parity_log.check(f"ten_substeps/{task}", dict(TEN_BOUNDS, ill_fraction=TEN_ILL_RANDOM[task]), qpos=wq, qvel=wv, qpos_normwise=wqn, qvel_normwise=wvn, ill_fraction=n_ill / n)


# the share of RANDOM start states (joint noise +-0.3 rad, qvel sigma 0.6 rad/s) the oracle sets aside by its own sensitivity: measured + 10 points
# (measured 0.154 / 0.402 / 0.482).  States a robot actually visits: `test_env_step_ten_substeps_on_rollout_states` below.
TEN_ILL_RANDOM = {"flat_terrain": 0.26, "flat_terrain_backlash": 0.51, "rough_terrain_backlash": 0.59}


def _conditioning(oracle_mod, om, q, v, w, c, q1, v1, prng, probes=6):
    """Is a ten-substep result DISCONTINUOUS in its start state at the scale of float32 rounding?  `probes` random directions, each at two scales:
    1e-7 (one float32 rounding of the state) and 1e-6 (what ten substeps of float32 arithmetic accumulate: test_one_mjx_step measures 1e-6 ... 5e-6).
    A smooth map answers in proportion (r6 ~ 10 r7) and its float32 evaluation stays within ~r7 of the float64 one; a state is SET ASIDE when
    (a) already the 1e-7 perturbation moves the oracle's own result by a quarter of the bound (no float32 evaluation can be held to the bound there), or
    (b) the 1e-6 perturbation moves it by more than half the bound AND more than 15 x the 1e-7 response: a decision (contact set, manifold arg-max,
    line-search bracket, warm-start pick) flips inside that band.
    (The random-state test's older criterion -- ANY 1e-6 response above half the bound -- also sets aside smooth states whose linear response to the
    5e-6 velocity perturbation is ~5e-5: 14 / 30 / 40 % of rollout states against 5 / 6 / 2 % by this one; tools: profiles/r6/NOTES.md.)"""
    for _ in range(probes):
        dq = prng.standard_normal(om.nq) * np.maximum(np.abs(q), 0.1); dv = prng.standard_normal(om.nv) * np.maximum(np.abs(v), 1.0)
        r = []
        for sc in (1e-7, 1e-6):
            d = _oracle_step(oracle_mod, om, q + sc * dq, v + 5 * sc * dv, w, c, 10)
            r.append(max(_rel(d["qpos"][: om.nq], q1, 1e-2).max() / TEN_BOUNDS["qpos"], _rel(d["qvel"][: om.nv], v1, 1.0).max() / TEN_BOUNDS["qvel"]))
        if r[0] > 0.25 or (r[1] > 0.5 and r[1] > 15.0 * r[0]):
            return True
    return False


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_env_step_ten_substeps_on_rollout_states(torch_cuda, oracle_mod, parity_log, task):
    """The same ten-substep comparison from states a robot VISITS: 448 snapshots of a random-action rollout through the env
    kernels (observation noise, pushes, auto-reset on; the actions of an untrained policy) -- post-reset states, states after 5 ... 60 env steps,
    and for the envs that fell the state one env step before the termination -- each with the warm start and the motor targets it had.  States at a
    float32-scale discontinuity of the map (`_conditioning`) are set aside: at most 10 %; the others are judged at the north-star bound, floored
    and norm-wise.  A judged state beyond the bound goes to the referee of the env tests' kind: the float64 oracle re-run with ONE class of its
    discrete decisions biased to the runner-up inside a 2e-6 band (whole step, then single substeps), then the oracle's float32 build -- a run that
    lands on the kernel's state explains it causally; explained states are bounded at 1 % (3 % on the height field; measured 0.2 / 0 / 1.6 %), unexplained ones at 0."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    n, T = 448, 60
    env = engine.Batch(model, n)
    env.reset(seed=31)
    g = torch.Generator(device="cuda").manual_seed(5)
    hist = []
    def snap():
        q, v, w = env.get_state()
        I = env.info()
        return q, v, w, np.array(I["motor_targets"]), np.array(I["episode_done"])
    hist.append(snap())
    for t in range(T):
        env.step(torch.empty(n, 14, device="cuda").uniform_(-1, 1, generator=g))
        hist.append(snap())
    env.close()
    done = np.stack([h[4] for h in hist])                      # [T + 1, n]: episode_done after step t
    sched = (0, 5, 10, 20, 30, 45, 60)
    pick = np.zeros(n, np.int64); kind = []
    n_pre = 0
    for i in range(n):
        fell = np.nonzero(done[:, i] != 0)[0]
        if len(fell) and fell[0] >= 3 and n_pre < 96:
            pick[i] = fell[0] - 1; kind.append("pre_termination"); n_pre += 1      # the state the terminating env step started from
        else:
            pick[i] = sched[i % len(sched)]; kind.append("post_reset" if pick[i] == 0 else "rollout")
    qpos = np.stack([hist[pick[i]][0][i] for i in range(n)]); qvel = np.stack([hist[pick[i]][1][i] for i in range(n)])
    warm = np.stack([hist[pick[i]][2][i] for i in range(n)]); ctrl = np.stack([hist[pick[i]][3][i] for i in range(n)])
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all()
    om = oracle_mod.OracleModel(model.blob())
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel, warm)
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 10)
    gq, gv, _ = b.get_state()
    b.close()
    wq = wv = wqn = wvn = 0.0
    prng = np.random.default_rng(99)
    n_ill = n_explained = n_unexplained = 0
    ill_by_kind, why = {}, {}
    om32, outliers = None, []
    for e in range(n):
        d = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 10)
        q1, v1 = np.array(d["qpos"][: om.nq]), np.array(d["qvel"][: om.nv])
        if _conditioning(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], q1, v1, prng):
            n_ill += 1
            ill_by_kind[kind[e]] = ill_by_kind.get(kind[e], 0) + 1
            continue
        eq, evv = _rel(gq[e], q1, 1e-2).max(), _rel(gv[e], v1, 1.0).max()
        if eq > TEN_BOUNDS["qpos"] or evv > TEN_BOUNDS["qvel"]:      # referee: one decision class biased, does the oracle land on the kernel's state?
            verdict = None
            # "lands on the kernel's state": within the bound -- or, for a flip early in the ten substeps (the two paths then part by ordinary float32
            # error on a changed contact set), within 5 % of the discrepancy it explains
            tol_q, tol_v = max(TEN_BOUNDS["qpos"], 0.05 * eq), max(TEN_BOUNDS["qvel"], 0.05 * evv)
            lands = lambda dd: _rel(gq[e], np.array(dd["qpos"][: om.nq], np.float64), 1e-2).max() <= tol_q and _rel(gv[e], np.array(dd["qvel"][: om.nv], np.float64), 1.0).max() <= tol_v
            for bit, name in ILL_CLASSES:
                oracle_mod.set_tie_bias(bit, 2e-6, 1e-4)
                try:
                    db = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 10)
                finally:
                    oracle_mod.set_tie_bias(0)
                if lands(db):
                    verdict = name
                    break
            if verdict is None:      # the same class in ONE substep only (a foot that rotates through a tie crosses it in one pass), bands 3e-7 / 2e-6
                for eps, eps_rel in ((3e-7, 1e-5), (2e-6, 1e-4)):
                    for bit, name in ILL_CLASSES:
                        for k in range(10):
                            oracle_mod.set_tie_bias(bit, eps, eps_rel, window=(k, k))
                            try:
                                db = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 10)
                            finally:
                                oracle_mod.set_tie_bias(0)
                            if lands(db):
                                verdict = f"{name}/substep{k}"
                                break
                        if verdict:
                            break
                    if verdict:
                        break
            if verdict is None:      # the oracle's float32 build: plain, then with one float32 rounding of noise on the state
                if om32 is None:
                    om32 = oracle_mod.OracleModel(model.blob(), f32=True)
                for k in range(13):
                    amp = 0.0 if k == 0 else (1e-7 if k < 7 else 1e-6)
                    qp = (qpos[e] + amp * prng.standard_normal(om.nq) * np.maximum(np.abs(qpos[e]), 0.1)).astype(np.float32)
                    vp = (qvel[e] + amp * prng.standard_normal(om.nv) * np.maximum(np.abs(qvel[e]), 1.0)).astype(np.float32)
                    d32 = _oracle_step(oracle_mod, om32, qp, vp, warm[e].astype(np.float32), ctrl[e].astype(np.float32), 10)
                    if lands(d32):
                        verdict = "float32_oracle" + ("" if k == 0 else f"@{amp:g}")
                        break
            if verdict is None:      # a discontinuity the six probes of `_conditioning` missed: float64 oracle, one float32 rounding of noise on the state
                for k in range(192):      # (a flip that one direction in twenty finds: 64 trials per amplitude)
                    amp = (1e-7, 3e-7, 1e-6)[k // 64]
                    qp = qpos[e] + amp * prng.standard_normal(om.nq) * np.maximum(np.abs(qpos[e]), 0.1)
                    vp = qvel[e] + 5 * amp * prng.standard_normal(om.nv) * np.maximum(np.abs(qvel[e]), 1.0)
                    if lands(_oracle_step(oracle_mod, om, qp, vp, warm[e], ctrl[e], 10)):
                        verdict = f"float64_oracle_state_noise@{amp:g}"
                        break
            outliers.append(e)
            print(f"[beyond the bound] {task} state {e} ({kind[e]}): qpos {eq:.2e} qvel {evv:.2e}; referee: {verdict}")
            if verdict:
                n_explained += 1; why[verdict] = why.get(verdict, 0) + 1
            else:
                n_unexplained += 1
            continue
        wq, wv = max(wq, eq), max(wv, evv)
        wqn, wvn = max(wqn, _nw(gq[e], q1)), max(wvn, _nw(gv[e], v1))
    if outliers:      # for replay on the CPU side (tools/replay_rollout_state.py)
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        np.savez(os.path.join(out_dir, f"rollout_state_outliers_{task}.npz"), idx=np.array(outliers), qpos=qpos[outliers], qvel=qvel[outliers], warm=warm[outliers], ctrl=ctrl[outliers],
                 gq=gq[outliers], gv=gv[outliers])
    kinds = {k: kind.count(k) for k in set(kind)}
    judged = n - n_ill
    print(task, "10 substeps from rollout states: worst rel qpos", wq, "qvel", wv, "set aside:", n_ill, "of", n, ill_by_kind, "kinds", kinds, "explained", why, "unexplained", n_unexplained)
    parity_log.rec(f"ten_substeps_rollout_states/{task}", None, states=n, judged=judged, **{"n_" + k: v for k, v in kinds.items()}, **{"ill_" + k: v for k, v in ill_by_kind.items()},
                   **{"explained_by_" + k: v for k, v in why.items()})
    assert kinds.get("post_reset", 0) >= 32 and kinds.get("rollout", 0) >= 200
    parity_log.check(f"ten_substeps_rollout_states/{task}", dict(TEN_BOUNDS, ill_fraction=0.10, explained_fraction=0.03 if "rough" in task else 0.01, unexplained=0), qpos=wq, qvel=wv, qpos_normwise=wqn,
                     qvel_normwise=wvn, ill_fraction=n_ill / n, explained_fraction=n_explained / max(judged, 1), unexplained=n_unexplained)


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_the_duck_with_elliptic_cones(torch_cuda, oracle_mod, parity_log, task):
    """`<option cone="elliptic">` on the duck itself: opt_cone = 1 (impratio 1, as the file has it), the physics kernels of the duck's shapes
    with the cone code (`ShapeAE` / `ShapeBE`; plane floor and the backlash model's height field) against the float64 oracle: the state after one
    mjx.step and after ten, at the duck's bounds; asked for 64 lanes per env the batch runs the cone kernels' 32; primitive feet are refused by name."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model, load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    model = Model({**model.a, "opt_cone": np.array([1], np.int32)})
    om = oracle_mod.OracleModel(model.blob())
    assert om.L.lib.odko_model_int(om.h, b"cone") == 1
    n = 128
    rng = np.random.default_rng(23)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.2)
    if "rough" in task:
        qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, qpos[:, 2] > 0.25)
    warm = rng.normal(0, 3.0, (n, model.nv))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    ct = torch.tensor(ctrl, dtype=torch.float32, device="cuda")
    b = engine.Batch(model, n)
    prng = np.random.default_rng(24)
    W = dict(qpos=0.0, qvel=0.0, qpos_normwise=0.0, qvel_normwise=0.0); T = dict(qpos=0.0, qvel=0.0)
    n_ill = {1: 0, 10: 0}
    for nsub, scale in ((1, 1.0), (10, 0.3)):
        v0 = scale * qvel; w0 = warm if nsub == 1 else np.zeros_like(warm)
        b.set_state(qpos, v0, w0)
        b.physics_step(ct, nsub)
        gq, gv, _ = b.get_state()
        for e in range(n):
            d = _oracle_step(oracle_mod, om, qpos[e], v0[e], w0[e], ctrl[e], nsub)
            q1, v1 = np.array(d["qpos"][: om.nq]), np.array(d["qvel"][: om.nv])
            ill = False
            if nsub == 1:           # one step: a contact-set tie (the helper of the stage tests)
                d0 = oracle_mod.OracleData(om)
                d0["qpos"][: om.nq] = qpos[e]; d0["qvel"][: om.nv] = v0[e]; d0["ctrl"][:14] = ctrl[e]; d0.forward()
                ill = _contact_tie(oracle_mod, om, qpos[e], v0[e], ctrl[e], prng, _contacts(d0))
            for _ in range(0 if nsub == 1 else 6):      # ten: the oracle's own sensitivity to 1e-6 of its input (a contact tie, a zone or a bracket flipping)
                qp = qpos[e] + 1e-6 * prng.standard_normal(om.nq) * np.maximum(np.abs(qpos[e]), 0.1); vp = v0[e] + 5e-6 * prng.standard_normal(om.nv) * np.maximum(np.abs(v0[e]), 1.0)
                dp = _oracle_step(oracle_mod, om, qp, vp, w0[e], ctrl[e], nsub)
                ill = ill or _rel(dp["qpos"][: om.nq], q1, 1e-2).max() > 0.5 * TEN_BOUNDS["qpos"] or _rel(dp["qvel"][: om.nv], v1, 1.0).max() > 0.5 * TEN_BOUNDS["qvel"]
            if ill:
                n_ill[nsub] += 1
                continue
            if nsub == 1:
                W["qpos"] = max(W["qpos"], _rel(gq[e], q1, 1e-2).max()); W["qvel"] = max(W["qvel"], _rel(gv[e], v1, 1.0).max())
                W["qpos_normwise"] = max(W["qpos_normwise"], _nw(gq[e], q1)); W["qvel_normwise"] = max(W["qvel_normwise"], _nw(gv[e], v1))
            else:
                T["qpos"] = max(T["qpos"], _rel(gq[e], q1, 1e-2).max()); T["qvel"] = max(T["qvel"], _rel(gv[e], v1, 1.0).max())
    b.close()
    print(task, "elliptic: one step", W, "ten substeps", T, "ill", n_ill, "of", n)
    parity_log.check(f"duck_elliptic/{task}/one_mjx_step", dict({k: STAGE_BOUNDS[k] for k in W}, ill_fraction=0.3), ill_fraction=n_ill[1] / n, **W)
    parity_log.check(f"duck_elliptic/{task}/ten_substeps", dict(TEN_BOUNDS, ill_fraction=0.65 if "rough" in task else 0.55), ill_fraction=n_ill[10] / n, **T)
    cfg = engine.default_config(); cfg.lanes_per_env = 64      # a hint: the cone kernels exist at 32 lanes per env, and that is what runs
    b64 = engine.Batch(model, 8, cfg)
    assert b64.lanes_per_env == 32
    b64.close()
    if task == "flat_terrain":      # sphere / capsule feet have no cone kernels: refused by name
        prim = _prim_feet_variant("flat_terrain", ("sphere", "capsule"))
        with pytest.raises(engine.OdkError, match="sphere / capsule"):
            engine.model_reduction(Model({**prim.a, "opt_cone": np.array([1], np.int32)}))


@pytest.mark.parametrize("task,lanes", [("flat_terrain", 32), ("flat_terrain", 64), ("flat_terrain_backlash", 32)])
def test_foot_foot_contacts(torch_cuda, oracle_mod, parity_log, task, lanes):
    """Feet pressed into each other (hip rolls inwards, robot lifted off the floor): the foot-foot SAT manifold,
    its contact rows and the coupled (virtual-tree) Hessian path against the oracle, one mjx.step."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model(task)
    a = model.a
    from open_duck_playground_amd.tables import build_kernel_tables
    tabs = build_kernel_tables(a)
    aq = tabs["k_act_qposadr"]          # actuator order: L hip_yaw, hip_roll, ...
    lroll, rroll, lyaw, ryaw, lpitch, rpitch = int(aq[1]), int(aq[10]), int(aq[0]), int(aq[9]), int(aq[2]), int(aq[11])
    rng = np.random.default_rng(5)
    # hip rolls press the feet together; the yaw / pitch variants cross them at an angle, which is where the best separating
    # axis is an edge pair (one contact) instead of a face (clipped manifold)
    grid = [(l, r, y, pt) for (y, pt) in ((0.0, 0.0), (0.35, 0.0), (-0.3, 0.25)) for l in (0.35, 0.4, 0.45, 0.5, 0.55, 0.6) for r in (-0.6, -0.55, -0.5, -0.45, -0.4, -0.3)]
    n = len(grid)
    qpos = np.tile(np.asarray(a["key_qpos"], np.float64), (n, 1)); qvel = np.zeros((n, model.nv))
    for e, (l, r, y, pt) in enumerate(grid):
        qpos[e, 2] = 0.3
        qpos[e, lroll] = l + rng.uniform(-0.01, 0.01); qpos[e, rroll] = r + rng.uniform(-0.01, 0.01)
        qpos[e, lyaw] += y; qpos[e, ryaw] -= y; qpos[e, lpitch] += pt; qpos[e, rpitch] -= pt
        qvel[e, 6:] = rng.normal(0, 0.5, model.nv - 6)
    warm = np.zeros((n, model.nv))
    ctrl = np.stack([qpos[e, aq] for e in range(n)])
    cfg = engine.default_config(); cfg.lanes_per_env = lanes
    b = engine.Batch(model, n, cfg)
    b.set_state(qpos, qvel, warm)
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    gq, gv, _ = b.get_state()
    img = b.lds_image()
    om = oracle_mod.OracleModel(model.blob())
    o = {k: b.lds_offset(k) for k in ("contact_dist", "efc_D", "qacc")}
    nv = model.nv
    n_pen = n_flip = n_tie = n_single = 0
    prng = np.random.default_rng(97)
    worst = dict(dist=0.0, qacc=0.0, qpos=0.0, qvel=0.0)
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        cd_o = np.array(d["contact_dist"][8:12]); cd_g = img[e][o["contact_dist"] + 8: o["contact_dist"] + 12]
        if (cd_o < 0).any():
            n_pen += 1
            n_single += int((cd_o < 0).sum() == 1)
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1      # the oracle's own manifold changes under 1e-6 noise (a sliver manifold: the arg-max of "farthest from the line a-b" is a tie)
            continue
        if set(np.flatnonzero(cd_o < 0)) != set(np.flatnonzero(cd_g < 0)):
            n_flip += 1      # a selection flip the oracle's own sensitivity does not explain: not tolerated
            continue
        both = (cd_o < 0)
        if both.any():
            worst["dist"] = max(worst["dist"], np.abs(cd_g[both] - cd_o[both]).max())
        worst["qacc"] = max(worst["qacc"], _rel(img[e][o["qacc"]: o["qacc"] + nv], d["qacc"][:nv], 5.0).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 1)
        worst["qpos"] = max(worst["qpos"], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
        worst["qvel"] = max(worst["qvel"], _rel(gv[e], ds["qvel"][:nv], 1.0).max())
    print(task, dict(n=n, penetrating=n_pen, flips=n_flip, ties=n_tie, **{k: float(f"{v:.3g}") for k, v in worst.items()}))
    b.close()
    assert n_pen >= 30, "the grid must contain penetrating poses"
    parity_log.rec(f"foot_foot/{task}/lanes{lanes}", dict(flips=0, ties=n // 20), flips=n_flip, penetrating=n_pen, ties=n_tie, poses=n, single_contact_poses=n_single)
    assert n_single >= 5, "the grid must contain edge-edge (single-contact) poses"
    assert n_flip == 0 and n_tie <= n // 20
    parity_log.check(f"foot_foot/{task}/lanes{lanes}", FOOT_BOUNDS, **worst)


def test_yaw_equivariance_full_size(torch_cuda):
    """8192 envs, size-independent property of the HIP path itself (no oracle): rotating every state by 90 degrees about
    the vertical axis (under which the friction pyramid maps onto itself) rotates base position / linear velocity after
    an env step of 10 substeps and leaves the joint state unchanged, to fp32 accuracy."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    torch = torch_cuda
    model = load_task_model("flat_terrain")
    n = 8192
    rng = np.random.default_rng(3)
    qpos, qvel = _random_states(model, 256, rng, airborne_frac=0.2)
    reps = n // 256
    qpos = np.tile(qpos, (reps, 1)); qvel = np.tile(qvel, (reps, 1)) * 0.3
    qpos[:, 0:2] += rng.uniform(-0.2, 0.2, (n, 2))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.2, 0.2, (n, 14))
    Rz = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    qz = np.array([np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    q2 = qpos.copy(); v2 = qvel.copy()
    q2[:, 0:3] = qpos[:, 0:3] @ Rz.T
    w1, x1, y1, z1 = qz
    w2, x2, y2, z2 = qpos[:, 3], qpos[:, 4], qpos[:, 5], qpos[:, 6]
    q2[:, 3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2; q2[:, 4] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    q2[:, 5] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2; q2[:, 6] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    v2[:, 0:3] = qvel[:, 0:3] @ Rz.T
    outs = []
    for q, v in ((qpos, qvel), (q2, v2)):
        b = engine.Batch(model, n)
        b.set_state(q, v, np.zeros((n, model.nv)))
        b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 10)
        outs.append(b.get_state())
        b.close()
    (qa, va, _), (qb, vb, _) = outs
    ok = np.isfinite(qa).all(axis=1) & np.isfinite(qb).all(axis=1)
    assert ok.mean() > 0.999
    qa, va, qb, vb = qa[ok], va[ok], qb[ok], vb[ok]
    err_p = np.abs(qb[:, 0:3] - qa[:, 0:3] @ Rz.T).max(axis=1)
    err_j = np.abs(qb[:, 7:] - qa[:, 7:]).max(axis=1)
    err_v = np.abs(vb[:, 0:3] - va[:, 0:3] @ Rz.T).max(axis=1)
    err_w = np.abs(vb[:, 3:] - va[:, 3:]).max(axis=1)
    # contact-manifold ties can flip between the two orientations in a handful of envs: judge the bulk
    assert np.quantile(err_p, 0.99) < 2e-5 and np.quantile(err_j, 0.99) < 2e-4, (np.quantile(err_p, 0.99), np.quantile(err_j, 0.99))
    assert np.quantile(err_v, 0.99) < 2e-3 and np.quantile(err_w, 0.99) < 2e-2, (np.quantile(err_v, 0.99), np.quantile(err_w, 0.99))
    assert np.median(err_j) < 2e-6 and np.median(err_p) < 1e-6


def _box_feet_variant(task):
    """the task's model with both foot meshes replaced by a BOX collider of the sole's size (what the compiler emits for
    <geom type="box">: eight corners, twelve outward triangles), in the same geom frame"""
    from open_duck_playground_amd.mjcf import convex_hull
    from open_duck_playground_amd.model import Model, load_task_model
    base = load_task_model(task)
    a = dict(base.a)
    v = np.asarray(a["hull_vert"])[: int(a["cgeom_vertnum"][0])]
    lo, hi = v.min(0), v.max(0)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    hv, hf = convex_hull(corners)
    a["hull_vert"] = hv; a["hull_face"] = hf
    for k, val in (("cgeom_vertadr", 0), ("cgeom_vertnum", len(hv)), ("cgeom_faceadr", 0), ("cgeom_facenum", len(hf))):
        arr = np.array(a[k]); arr[:2] = val; a[k] = arr
    return Model(a, base.xml_path)


@pytest.mark.parametrize("task", ["flat_terrain", "rough_terrain_backlash"])
def test_box_feet_variant(torch_cuda, oracle_mod, parity_log, task):
    """SURVEY 8(f).3, first step: a colliding BOX (the compiler turns it into its corner hull: tests/test_mjcf_box.py) in place of
    the foot meshes runs through the same kernels -- plane-convex on the flat floor, the prism routine on the height field, the
    convex-convex routine between the feet -- and agrees with the oracle like the mesh feet do."""
    from open_duck_playground_amd import engine
    torch = torch_cuda
    model = _box_feet_variant(task)
    assert int(model.a["cgeom_vertnum"][0]) == 8
    om = oracle_mod.OracleModel(model.blob())
    assert om.convex_counts(0) == (8, 6, 12)
    n = 48
    rng = np.random.default_rng(31)
    qpos, qvel = _random_states(model, n, rng)
    if "rough" in task:
        qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, qpos[:, 2] > 0.25)
    else:   # a third of the poses with the feet pressed against each other, off the floor
        aq = build_tables(model)["k_act_qposadr"]
        for e in range(0, n, 3):
            qpos[e] = np.asarray(model.a["key_qpos"]); qpos[e, 2] = 0.3
            qpos[e, int(aq[1])] = rng.uniform(0.4, 0.6); qpos[e, int(aq[10])] = rng.uniform(-0.6, -0.4); qpos[e, int(aq[0])] += rng.uniform(-0.3, 0.3)
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    b = engine.Batch(model, n)
    b.set_state(qpos, qvel, np.zeros((n, model.nv)))
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
    gq, gv, _ = b.get_state()
    img = b.lds_image()
    o_cd = b.lds_offset("contact_dist")
    prng = np.random.default_rng(32)
    W = dict(dist=0.0, qpos=0.0, qvel=0.0)
    n_tie = n_contact = n_ff = 0
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        cd_o = np.array(d["contact_dist"][:12]); cd_g = img[e][o_cd: o_cd + 12]
        n_contact += int((cd_o[:8] < 0).any()); n_ff += int((cd_o[8:] < 0).any())
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1
            continue
        act = cd_o < 0
        assert set(np.flatnonzero(act)) == set(np.flatnonzero(cd_g < 0)), e
        if act.any():
            W["dist"] = max(W["dist"], np.abs(cd_g[act] - cd_o[act]).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], np.zeros(model.nv), ctrl[e], 1)
        W["qpos"] = max(W["qpos"], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
        W["qvel"] = max(W["qvel"], _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
    b.close()
    assert n_contact >= n // 3 and ("rough" in task or n_ff >= 5), (n_contact, n_ff)
    # a box sole is exactly flat: on the plane its four corners tie in depth whenever the foot lies flat (rare in these random poses)
    parity_log.check(f"box_feet/{task}", dict(dist=1e-6, qpos=1e-5, qvel=4e-5, tie_fraction=0.15), tie_fraction=n_tie / n, **W)


def _substep_sensitivity(O, om, om32, qpos, qvel, ctrl, nsub, prng, tries=6):
    """oracle side only, along the float64 oracle's own trajectory: the largest change of one substep's qvel (a) under 1e-6 / 5e-6
    perturbations of that substep's state and (b) when the same substep is done by the oracle's float32 build.  Where a
    line-search branch, a contact or the pick between two candidate pairs of closest points (near-parallel capsules: a tie by
    construction, decided at the level of the formula's 1e-6 regularisers) flips there, no float32 implementation follows the
    float64 one -- MJX itself runs in float32."""
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos; d["qvel"][: om.nv] = qvel; d["qacc_warmstart"][: om.nv] = 0.0
    worst = 0.0
    for _ in range(nsub):
        q, v, w = (np.array(d[k][:n]) for k, n in (("qpos", om.nq), ("qvel", om.nv), ("qacc_warmstart", om.nv)))
        d.env_physics_step(ctrl, 1)
        v1 = np.array(d["qvel"][: om.nv])
        for _t in range(tries):
            dq = np.zeros(om.nq); dq[7:] = prng.uniform(-1e-6, 1e-6, om.nq - 7)
            dp = _oracle_step(O, om, q + dq, v + prng.uniform(-5e-6, 5e-6, om.nv), w, ctrl, 1)
            worst = max(worst, _rel(np.array(dp["qvel"][: om.nv]), v1, 1.0).max())
        d32 = _oracle_step(O, om32, q, v, w, ctrl, 1)
        worst = max(worst, _rel(np.array(d32["qvel"][: om.nv], np.float64), v1, 1.0).max())
    return worst


def _prim_feet_variant(task, kinds):
    """the task's model with the foot meshes replaced by primitive colliders (what the compiler emits for <geom type="sphere"> /
    <geom type="capsule">) at the centre of the sole's bounding box; a capsule lies along the sole's longest side"""
    from open_duck_playground_amd.mjcf import GEOM_CAPSULE, GEOM_SPHERE
    from open_duck_playground_amd.model import Model, load_task_model
    base = load_task_model(task)
    a = {k: np.array(v) for k, v in base.a.items()}
    v = np.asarray(a["hull_vert"])[: int(a["cgeom_vertnum"][0])]
    lo, hi = v.min(0), v.max(0)
    ctr, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    size = np.zeros((len(a["cgeom_type"]), 3))
    for f, kind in enumerate(kinds):
        q = a["cgeom_quat"][f]; w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        a["cgeom_pos"][f] = a["cgeom_pos"][f] + R @ ctr
        r = float(np.sort(half)[1]) * 0.6
        if kind == "capsule":
            k = int(np.argmax(half))
            ax = np.eye(3)[k]                                         # the geom frame's z axis turned onto the longest side
            w3 = np.cross([0, 0, 1.0], ax); qz = np.array([1.0 + ax[2], *w3]); qz /= np.linalg.norm(qz)
            w1, x1, y1, z1 = q; w2, x2, y2, z2 = qz
            a["cgeom_quat"][f] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
            a["cgeom_type"][f] = GEOM_CAPSULE; size[f] = [r, float(half[k]) - r, 0]
        else:
            a["cgeom_type"][f] = GEOM_SPHERE; size[f] = [r, 0, 0]
    a["cgeom_size"] = size
    for k in ("cgeom_vertnum", "cgeom_facenum"):
        a[k][:2] = 0
    return Model(a, base.xml_path)


@pytest.mark.parametrize("kinds", [("sphere", "sphere"), ("capsule", "capsule"), ("sphere", "capsule"), ("capsule", "sphere")], ids="-".join)
def test_primitive_feet_variant(torch_cuda, oracle_mod, parity_log, kinds):
    """SURVEY 8(f).3, second step: sphere / capsule foot colliders on the plane floor (plane_sphere, plane_capsule with its
    axis-aligned frame, sphere_sphere / sphere_capsule / capsule_capsule between the feet) against the oracle: contacts, then one
    substep and ten."""
    from open_duck_playground_amd import engine
    torch = torch_cuda
    model = _prim_feet_variant("flat_terrain", kinds)
    om = oracle_mod.OracleModel(model.blob())
    om32 = oracle_mod.OracleModel(model.blob(), f32=True)
    assert om.npair == 3
    n = 48
    rng = np.random.default_rng(41)
    qpos, qvel = _random_states(model, n, rng)
    aq = build_tables(model)["k_act_qposadr"]
    for e in range(0, n, 3):   # a third of the poses with the feet pressed against each other, off the floor
        qpos[e] = np.asarray(model.a["key_qpos"]); qpos[e, 2] = 0.3
        qpos[e, int(aq[1])] = rng.uniform(0.4, 0.6); qpos[e, int(aq[10])] = rng.uniform(-0.6, -0.4); qpos[e, int(aq[0])] += rng.uniform(-0.3, 0.3)
    for e in range(1, n, 3):   # a third lowered until a foot is 0.3 ... 3 mm in the floor
        d = oracle_mod.OracleData(om)
        for _ in range(4):
            d["qpos"][: om.nq] = qpos[e]; d.forward()
            qpos[e, 2] -= min(np.array(d["contact_dist"][:8]).min(), 0.05) + rng.uniform(3e-4, 3e-3)
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    W = dict(dist=0.0, pos=0.0, qpos=0.0, qvel=0.0, qpos10=0.0, qvel10=0.0)
    n_contact = n_ff = n_ill = 0
    prng = np.random.default_rng(42)
    for nsub in (1, 10):
        b = engine.Batch(model, n)
        b.set_state(qpos, qvel, np.zeros((n, model.nv)))
        b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), nsub)
        gq, gv, _ = b.get_state()
        for e in range(n):
            ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], np.zeros(model.nv), ctrl[e], nsub)
            sfx = "" if nsub == 1 else "10"
            if nsub == 10 and _rel(gv[e], ds["qvel"][: om.nv], 1.0).max() > 1.5e-4:
                # over the bound after ten substeps: accepted only if the oracle itself is that sensitive somewhere along the way
                # (_substep_sensitivity) -- and counted
                worst = _substep_sensitivity(oracle_mod, om, om32, qpos[e], qvel[e], ctrl[e], nsub, prng)
                assert worst > 1e-4, (e, worst, _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
                n_ill += 1
                continue
            W["qpos" + sfx] = max(W["qpos" + sfx], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
            W["qvel" + sfx] = max(W["qvel" + sfx], _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
        if nsub == 1:   # the contacts of the last forward pass (the LDS image is the state BEFORE the integration)
            img = b.lds_image()
            o_cd, o_cr = b.lds_offset("contact_dist"), b.lds_offset("contact_r")
            for e in range(n):
                d = oracle_mod.OracleData(om)
                d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
                d.forward()
                cd_o = np.array(d["contact_dist"][:12]); cd_g = img[e][o_cd: o_cd + 12]
                n_contact += int((cd_o[:8] < 0).any()); n_ff += int((cd_o[8:] < 0).any())
                near = np.abs(cd_o) < 0.05
                W["dist"] = max(W["dist"], np.abs(cd_g[near] - cd_o[near]).max() if near.any() else 0.0)
                assert ((cd_o == 1.0) == (cd_g == 1.0)).all(), (e, cd_o, cd_g)          # the unused slots of each pair
                for c in np.flatnonzero(cd_o < 0):
                    W["pos"] = max(W["pos"], np.abs(img[e][o_cr + 3 * c: o_cr + 3 * c + 3] + qpos[e, :3] - np.array(d["contact_pos"][3 * c: 3 * c + 3])).max())
        b.close()
    assert n_contact >= n // 4 and n_ff >= 5, (n_contact, n_ff)
    parity_log.check("prim_feet/" + "-".join(kinds), dict(dist=5e-7, pos=5e-7, qpos=3e-6, qvel=4e-5, qpos10=3e-5, qvel10=1.5e-4, ill_fraction=0.05), ill_fraction=n_ill / n, **W)


def build_tables(model):
    from open_duck_playground_amd.tables import build_kernel_tables
    return build_kernel_tables(model.a)


@pytest.mark.parametrize("task,lanes", [("flat_terrain", 32), ("flat_terrain", 64), ("flat_terrain_backlash", 32), ("rough_terrain_backlash", 32)])
def test_differential_sweep(torch_cuda, oracle_mod, parity_log, task, lanes):
    """A reduced run of tools/gpu_fuzz_parity.py: 768 random contact-rich states per model (leaning on one foot, near the home pose,
    feet pressed together in the air / on the floor -- next to a neighbour of another kind in the same wave --, airborne, far from the
    origin on the height field), one mjx.step.  Every state either agrees with the float64 oracle or is explained on the oracle side
    (contact tie under rounding-level noise; the oracle's float32 build on the kernel's side; solver branch): nothing unexplained."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gpu_fuzz_parity", os.path.join(ROOT, "tools", "gpu_fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    n = 768
    stat, worst = fz.sweep(task, n, seed=123, lanes=lanes, nsub=1, dist_tol=3e-6 if "rough" in task else 1e-6)   # (8 m from the origin float32 positions carry 5e-7 m)
    assert stat["in_contact"] > n // 2 and stat["foot_foot"] > n // 20 and stat["both_feet"] > n // 20, stat
    assert stat["unexplained"] == 0, stat
    parity_log.check(f"differential_sweep/{task}/lanes{lanes}", dict(dist=3e-6 if "rough" in task else 3e-7, qvel=1e-4, explained_fraction=0.01),
                     explained_fraction=(stat["tie"] + stat["f32_side"] + stat["solver_branch"]) / n, **worst)



def _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, xml, tag, eq_active=None, cone=False, overrides=None, dims=(22, 21, 15, 19, 16), red_dims=(21, 156, 181), ill_bound=0.5):
    """(body of the two tests below)  SURVEY 8(f).3 / reference README.md:74-85 ("adding a robot"): tests/assets/tail_biped.xml -- a biped with a five-link tail,
    written for this test: 21 dofs, 15 position actuators, 19 bodies, box feet, its own masses / lengths / axes / gains -- compiled by
    mjcf.py, its lane tables built by tables.py (nothing by hand), loaded as the kernels' third Shape and run through the PHYSICS
    kernels (odk_physics_step): every comparable stage of one mjx.step and the state after ten, against the float64 oracle, at
    the duck's bounds.  The env kernels (observations, rewards, 14 actions) stay the duck's task logic and refuse this model."""
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    from open_duck_playground_amd.tables import build_kernel_tables, reduced_layout
    from conftest import ROOT
    torch = torch_cuda
    model = Model.from_xml(os.path.join(ROOT, "tests", "assets", xml), sim_dt=0.002)
    if eq_active is not None:
        model = Model({**model.a, "eq_active": np.asarray(eq_active, np.int32)})
    if overrides:
        model = Model({**model.a, **overrides})
    assert (model.nq, model.nv, model.nu, model.nbody, model.njnt) == dims
    red = engine.model_reduction(model)
    assert red["paired"] == 0 and (red["nvr"], red["nMr"], red["nHr"]) == red_dims
    om = oracle_mod.OracleModel(model.blob())
    # equality rows come first in the oracle's row order (connects 3 rows each, welds 6, joints 1); the kernels keep theirs beside the row arrays
    ne_rows = 0 if eq_active is None else int(sum(a * {0: 3, 1: 6, 2: 1}[int(t)] for a, t in zip(eq_active, np.asarray(model.a["eq_type"]).reshape(-1))))
    n = 64
    rng = np.random.default_rng(41)
    nq, nv, nb = model.nq, model.nv, model.nbody
    qpos = np.tile(np.asarray(model.a["key_qpos"], np.float64), (n, 1)); qvel = np.zeros((n, nv))
    air = rng.uniform(size=n) < 0.25
    for e in range(n):
        qpos[e, 0:2] += rng.uniform(-0.05, 0.05, 2)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(-1.0, 1.0) if air[e] else rng.uniform(-0.2, 0.2)
        qpos[e, 3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        for j in range(1, model.njnt):
            a_, (lo, hi) = model.a["jnt_qposadr"][j], model.a["jnt_range"][j]
            qpos[e, a_] = np.clip(qpos[e, a_] + rng.uniform(-0.3, 0.3), lo - 0.02, hi + 0.02)      # some joints past their limits
        qpos[e, 2] = rng.uniform(0.5, 0.8) if air[e] else qpos[e, 2]
        qvel[e, :3] = rng.normal(0, 0.3, 3); qvel[e, 3:6] = rng.normal(0, 1.0, 3); qvel[e, 6:] = rng.normal(0, 2.0, nv - 6)
    qpos = _settle_on_terrain(oracle_mod, om, qpos, rng, air)      # (any floor: moves the base so that the deepest contact is 0.3 ... 3 mm)
    warm = rng.normal(0, 5.0, (n, nv))
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.4, 0.4, (n, model.nu))
    b = engine.Batch(model, n)
    ctrl_t = torch.tensor(ctrl, dtype=torch.float32, device="cuda")
    b.set_state(qpos, qvel, warm)
    b.physics_step(ctrl_t, 1)
    gq, gv, gw = b.get_state()
    img = b.lds_image()
    o = {k: b.lds_offset(k) for k in ("xpos", "M", "qfrc_smooth", "qacc_smooth", "contact_dist", "efc_D", "efc_aref", "qacc", "sensordata", "actuator_force", "scr")}
    tabs = build_kernel_tables(model.a); lay = reduced_layout(model.a)
    Mi, Mj = lay["ei"], lay["ej"]
    assert np.array_equal(Mi, tabs["k_M_i"]) and len(Mi) == red_dims[1]
    nfl = len(tabs["k_fl_dof"])
    W = dict(xpos=0, M=0, qfs=0, qas=0, dist=0, D=0, aref=0, qacc=0, qpos=0, qvel=0, sens=0, force=0)
    prng = np.random.default_rng(5)
    n_tie = n_contact = 0
    W_zone = {}
    for e in range(n):
        d = oracle_mod.OracleData(om)
        d["qpos"][:nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][: model.nu] = ctrl[e]
        d.forward()
        L = img[e]
        W["xpos"] = max(W["xpos"], np.abs(L[o["xpos"]: o["xpos"] + 3 * nb].reshape(3, nb).T - d["xpos"][: 3 * nb].reshape(nb, 3)).max())
        W["M"] = max(W["M"], _rel(L[o["M"]: o["M"] + len(Mi)], d.M()[Mi, Mj], 1e-4).max())
        W["qfs"] = max(W["qfs"], _rel(L[o["qfrc_smooth"]: o["qfrc_smooth"] + nv], d["qfrc_smooth"][:nv], 1e-2).max())
        W["qas"] = max(W["qas"], _rel(L[o["qacc_smooth"]: o["qacc_smooth"] + nv], d["qacc_smooth"][:nv], 1.0).max())
        W["force"] = max(W["force"], _rel(L[o["actuator_force"]: o["actuator_force"] + model.nu], d["actuator_force"][: model.nu], 1e-2).max())
        if _contact_tie(oracle_mod, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d)):
            n_tie += 1
            continue
        cd_g, cd_o = L[o["contact_dist"]: o["contact_dist"] + 12], np.array(d["contact_dist"][:12])
        act = (cd_o < 0) | (cd_g < 0)
        n_contact += int(act.any())
        if act.any():
            W["dist"] = max(W["dist"], np.abs(cd_g[act] - cd_o[act]).max())
        nefc = d.i("nefc") - ne_rows
        assert d.i("ne") == ne_rows
        live = (np.abs(d.J()).sum(axis=1) > 0)[ne_rows:]
        D_g, aref_g = L[o["efc_D"]: o["efc_D"] + nefc], L[o["efc_aref"]: o["efc_aref"] + nefc]
        if cone:      # elliptic: the oracle has three rows per contact, the kernels keep four row lanes per contact (normal, two tangents, an empty one)
            r0 = nefc - 3 * 12
            fo = np.array(d["efc_force"][ne_rows + r0: ne_rows + nefc]).reshape(12, 3)      # which zone the solver left each penetrating contact in
            for c in range(12):
                if cd_o[c] < 0:
                    ft = np.hypot(fo[c, 1], fo[c, 2])
                    zone = "top" if fo[c, 0] == 0 and ft == 0 else ("middle" if ft > 0 and abs(ft - float(d["contact_friction"][c]) * fo[c, 0]) < 1e-9 * max(ft, 1.0) else "bottom")
                    W_zone[zone] = W_zone.get(zone, 0) + 1
            take = np.concatenate([np.arange(r0), r0 + np.array([4 * c + s_ for c in range(12) for s_ in range(3)])])
            D_g, aref_g = L[o["efc_D"]: o["efc_D"] + r0 + 48][take], L[o["efc_aref"]: o["efc_aref"] + r0 + 48][take]
            assert not (L[o["efc_D"] + r0 + 3: o["efc_D"] + r0 + 48: 4] != 0).any()
        assert ((D_g > 0) == live)[nfl:].all(), f"env {e}: active row sets differ"
        W["D"] = max(W["D"], _rel(D_g[live], d["efc_D"][ne_rows: ne_rows + nefc][live], 1e-6).max())
        W["aref"] = max(W["aref"], _rel(aref_g[live], d["efc_aref"][ne_rows: ne_rows + nefc][live], 1.0).max())
        W["qacc"] = max(W["qacc"], _rel(L[o["qacc"]: o["qacc"] + nv], d["qacc"][:nv], 5.0).max())
        W["sens"] = max(W["sens"], _rel(L[o["sensordata"]: o["sensordata"] + 46], d["sensordata"][:46], 1.0).max())
        ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], warm[e], ctrl[e], 1)
        W["qpos"] = max(W["qpos"], _rel(gq[e], ds["qpos"][:nq], 1e-2).max()); W["qvel"] = max(W["qvel"], _rel(gv[e], ds["qvel"][:nv], 1.0).max())
    assert n_contact >= n // 3, n_contact
    # ten substeps (one env step of physics) from calmer states
    qv2 = 0.3 * qvel; w0 = np.zeros((n, nv)); c2 = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.2, 0.2, (n, model.nu))
    b.set_state(qpos, qv2, w0)
    b.physics_step(torch.tensor(c2, dtype=torch.float32, device="cuda"), 10)
    tq, tv, _ = b.get_state()
    T10 = dict(qpos=0.0, qvel=0.0); n_ill = 0
    for e in range(n):
        d = _oracle_step(oracle_mod, om, qpos[e], qv2[e], w0[e], c2[e], 10)
        q1, v1 = np.array(d["qpos"][:nq]), np.array(d["qvel"][:nv])
        ill = False
        for _ in range(8):
            qp = qpos[e] + 1e-6 * prng.standard_normal(nq) * np.maximum(np.abs(qpos[e]), 0.1); vp = qv2[e] + 5e-6 * prng.standard_normal(nv) * np.maximum(np.abs(qv2[e]), 1.0)
            dp = _oracle_step(oracle_mod, om, qp, vp, w0[e], c2[e], 10)
            ill = ill or _rel(dp["qpos"][:nq], q1, 1e-2).max() > 0.5 * RTOL_Q or _rel(dp["qvel"][:nv], v1, 1.0).max() > 0.5 * RTOL_Q
        if ill:
            n_ill += 1
            continue
        T10["qpos"] = max(T10["qpos"], _rel(tq[e], q1, 1e-2).max()); T10["qvel"] = max(T10["qvel"], _rel(tv[e], v1, 1.0).max())
    # the env kernels are the duck's task logic: this model must be refused there, loudly
    with pytest.raises(engine.OdkError):
        b.reset(seed=1)
    b.close()
    print(tag, {k: float(f"{v:.3g}") for k, v in W.items()}, "ties", n_tie, "ten substeps", T10, "ill", n_ill, "of", n, "cone zones", W_zone)
    if cone:      # the states must exercise the cone itself (sliding contacts: measured 58), the plain quadratic zone (sticking ones: 13) and separating contacts (11)
        assert W_zone.get("middle", 0) >= 20 and W_zone.get("bottom", 0) >= 3 and W_zone.get("top", 0) >= 4, W_zone      # (impratio 10: 67 / 4 / 11)
    parity_log.check(f"{tag}/one_mjx_step", dict(STAGE_BOUNDS, force=2e-4, tie_fraction=0.15), tie_fraction=n_tie / n, **W)
    parity_log.check(f"{tag}/ten_substeps", dict(TEN_BOUNDS, ill_fraction=ill_bound), ill_fraction=n_ill / n, **T10)
    return W, T10


def test_a_robot_that_is_not_the_duck(torch_cuda, oracle_mod, parity_log):
    """tests/assets/tail_biped.xml through the physics kernels at the duck's bounds (see the helper above)."""
    _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "tail_biped.xml", "tail_biped")


def test_a_biped_with_six_dof_legs(torch_cuda, oracle_mod, parity_log):
    """tests/assets/biped12.xml -- hip yaw / roll / pitch, knee, ankle pitch / roll per leg: what most humanoids have, one joint more per
    chain than the duck -- through the physics kernels at the duck's bounds: the fourth model shape (18 dofs, 12 actuators, 16 bodies),
    whose chain solve works blocks of six and whose contact wrenches have their own floats (16 bodies' cfrc | crb region is too small)."""
    _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "biped12.xml", "biped12", dims=(19, 18, 12, 16, 13), red_dims=(18, 135, 171))


def test_the_six_dof_biped_with_elliptic_cones_and_a_coupled_ankle(torch_cuda, oracle_mod, parity_log):
    """The optional constraint code is compiled into the fourth shape as into the third: biped12.xml with `cone="elliptic"` (impratio 2) AND a joint
    coupling inside the left leg (ankle pitch = -0.5 knee, the parallel-bar ankle of many humanoids) through the physics kernels at the duck's bounds."""
    from open_duck_playground_amd.model import Model
    import os
    from conftest import ROOT
    base = Model.from_xml(os.path.join(ROOT, "tests", "assets", "biped12.xml"), sim_dt=0.002)
    j = base.joint_id
    over = dict(opt_cone=np.array([1], np.int32), opt_impratio=np.array([2.0]),
                eq_type=np.array([2], np.int32), eq_obj1id=np.array([j("left_ankle_pitch")], np.int32), eq_obj2id=np.array([j("left_knee")], np.int32),
                eq_active=np.array([1], np.int32), eq_data=np.array([[0.0, -0.5, 0, 0, 0, 0, 0, 0, 0, 0, 0]], np.float64), eq_solref=np.array([[0.02, 1.0]]),
                eq_solimp=np.array([[0.9, 0.95, 0.001, 0.5, 2.0]]), neq=np.array([1], np.int32), names_eq=np.array(["ankle_bar"]))
    _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "biped12.xml", "biped12_elliptic_coupled", eq_active=(1,), cone=True, overrides=over,
                                       dims=(19, 18, 12, 16, 13), red_dims=(18, 135, 171))


def test_equality_joint_rows_in_the_kernels(torch_cuda, oracle_mod, parity_log):
    """SURVEY 8(f).3, <equality> (reference README.md:74-85): tests/assets/tail_biped_equality.xml with its two joint couplings ACTIVE
    (tail_yaw_2 = 0.5 tail_yaw_1; left_ankle = 0.1 - 0.5 knee + 0.2 knee^2, solref 0.01) -- rows between two dofs of one serial chain,
    always active, evaluated by both dof lanes, their Hessian term on an entry of the chain's own block (odk_kernels.h "equality rows") --
    through the physics kernels against the float64 oracle: every stage of one mjx.step (the solver's qacc with the rows in it), the state
    after one step and after ten, at the duck's bounds.  And the rows must MATTER: the same states stepped without them end up elsewhere."""
    W, T10 = _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "tail_biped_equality.xml", "tail_biped_equality", eq_active=(1, 1, 0, 0))
    # the couplings change the motion: one step with and without them from the same state
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    from conftest import ROOT
    torch = torch_cuda
    base = Model.from_xml(os.path.join(ROOT, "tests", "assets", "tail_biped_equality.xml"), sim_dt=0.002)
    out = []
    for act in ((1, 1, 0, 0), (0, 0, 0, 0)):
        m = Model({**base.a, "eq_active": np.asarray(act, np.int32)})
        b = engine.Batch(m, 4)
        q = np.tile(np.asarray(m.a["key_qpos"], np.float64), (4, 1)); q[:, 2] += 0.4
        q[:, int(m.a["jnt_qposadr"][m.joint_id("tail_yaw_1")])] = 0.4          # couplings violated: tail_yaw_2 should follow, the ankle should move
        b.set_state(q, np.zeros((4, m.nv)), np.zeros((4, m.nv)))
        b.physics_step(torch.tensor(np.tile(np.asarray(m.a["key_ctrl"]), (4, 1)), dtype=torch.float32, device="cuda"), 10)
        out.append(b.get_state()[0][0].copy())
        b.close()
    a2 = int(base.a["jnt_qposadr"][base.joint_id("tail_yaw_2")]); ak = int(base.a["jnt_qposadr"][base.joint_id("left_ankle")])
    assert abs(out[0][a2] - out[1][a2]) > 0.02 and abs(out[0][ak] - out[1][ak]) > 0.01, (out[0][a2], out[1][a2], out[0][ak], out[1][ak])


def test_connect_and_weld_rows_in_the_kernels(torch_cuda, oracle_mod, parity_log):
    """SURVEY 8(f).3, <equality><connect> and <weld> (reference README.md:74-85): tests/assets/tail_biped_equality.xml with ALL FOUR of its
    constraints active -- the two joint couplings, the right foot pinned to the world (connect: 3 rows), the tail tip welded to its parent link
    (weld: 6 rows) -- through the physics kernels against the float64 oracle at the duck's bounds.  Connect / weld run as "path rows"
    (odk_kernels.h): both bodies on one root-to-leaf path of the tree (or the world), a wrench per body and row, the Jacobian entry of a
    dof from ITS motion vector, the rows' J^T D J added to Hessian entries the tree layout already has.  (A loop between the two foot
    chains: test_a_closed_loop_between_the_feet.)  And the rows must MATTER."""
    W, T10 = _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "tail_biped_equality.xml", "tail_biped_equality_all", eq_active=(1, 1, 1, 1),
                                                ill_bound=0.6)      # (measured 31 of 64: a foot pinned centimetres from where the random state puts it is stiff)
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    from conftest import ROOT
    torch = torch_cuda
    base = Model.from_xml(os.path.join(ROOT, "tests", "assets", "tail_biped_equality.xml"), sim_dt=0.002)
    out = []
    for act in ((0, 0, 1, 1), (0, 0, 0, 0)):
        m = Model({**base.a, "eq_active": np.asarray(act, np.int32)})
        b = engine.Batch(m, 4)
        q = np.tile(np.asarray(m.a["key_qpos"], np.float64), (4, 1)); q[:, 2] += 0.3           # lifted 0.3 m above the pin's anchor (the foot's place in the key pose)
        q[:, int(m.a["jnt_qposadr"][m.joint_id("tail_roll")])] = 0.5                               # the weld is violated: the tail tip must come back
        b.set_state(q, np.zeros((4, m.nv)), np.zeros((4, m.nv)))
        b.physics_step(torch.tensor(np.tile(np.asarray(m.a["key_ctrl"]), (4, 1)), dtype=torch.float32, device="cuda"), 100)
        out.append(b.get_state()[0][0].copy())
        b.close()
    tr = int(base.a["jnt_qposadr"][base.joint_id("tail_roll")])
    assert np.isfinite(out[0]).all()
    assert out[0][2] < out[1][2] - 0.05, (out[0][2], out[1][2])                 # 0.2 s of free fall is 0.2 m; the pin pulls the foot back to its anchor faster than gravity does
    assert abs(out[0][tr]) < 0.5 * abs(out[1][tr]) + 0.05, (out[0][tr], out[1][tr])   # the welded joint is pulled towards its reference


def test_a_closed_loop_between_the_feet(torch_cuda, oracle_mod, parity_log):
    """tests/assets/tail_biped_loop.xml: the two foot links tied together by a ball joint (<equality><connect> across the two leg chains: a CLOSED
    kinematic loop) plus the welded tail tip, through the physics kernels against the float64 oracle at the duck's bounds.  The loop's rows are
    path rows whose two supports lie on different chains; their J^T D J needs Hessian entries between the two legs, which the VIRTUAL tree's
    layout has (the layout an active foot-foot contact switches to: second leg below the first foot) -- a model with such a constraint stays on
    it.  A loop through the tail's chain has no such entries and is refused by name.  And the loop must MATTER."""
    W, T10 = _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "tail_biped_loop.xml", "tail_biped_loop", eq_active=(1, 1))
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    from conftest import ROOT
    torch = torch_cuda
    base = Model.from_xml(os.path.join(ROOT, "tests", "assets", "tail_biped_loop.xml"), sim_dt=0.002)
    out = []
    for act in ((1, 0), (0, 0)):
        m = Model({**base.a, "eq_active": np.asarray(act, np.int32)})
        b = engine.Batch(m, 4)
        q = np.tile(np.asarray(m.a["key_qpos"], np.float64), (4, 1)); q[:, 2] += 0.4
        c = np.tile(np.asarray(m.a["key_ctrl"]), (4, 1))
        c[:, 1] = 0.4; c[:, 11] = -0.4          # both hip rolls driven outwards: the tied feet cannot spread
        b.set_state(q, np.zeros((4, m.nv)), np.zeros((4, m.nv)))
        b.physics_step(torch.tensor(c, dtype=torch.float32, device="cuda"), 100)
        out.append(b.get_state()[0][0].copy())
        b.close()
    lr, rr = int(base.a["jnt_qposadr"][base.joint_id("left_hip_roll")]), int(base.a["jnt_qposadr"][base.joint_id("right_hip_roll")])
    spread = lambda q: abs(q[lr] - q[rr])
    assert np.isfinite(out[0]).all() and spread(out[0]) < 0.6 * spread(out[1]), (spread(out[0]), spread(out[1]))
    loop = dict(base.a); o2 = np.array(base.a["eq_obj2id"], np.int32); o2[0] = base.body_id("tail_3"); loop["eq_obj2id"] = o2
    with pytest.raises(engine.OdkError, match="another loop has no entries"):
        engine.model_reduction(Model(loop))


@pytest.mark.parametrize("impratio", [3.0, 1.0, 10.0])
def test_elliptic_cones_in_the_kernels(torch_cuda, oracle_mod, parity_log, impratio):
    """SURVEY 8(f).3, `<option cone="elliptic">` (reference README.md:74-85): tests/assets/tail_biped_elliptic.xml (impratio 3) through the
    physics kernels -- a contact's four row lanes hold normal | tangent | tangent | nothing, the cost of a contact is the cone's three-zone
    cost, its Hessian block the cone's 3 x 3, the line search evaluates the cone exactly at every step size (odk_kernels.h "elliptic
    cones") -- against the float64 oracle, whose cone forces are pinned to the documented cone program (tests/test_oracle_elliptic.py):
    every stage of one mjx.step, the state after one step and after ten, at the duck's bounds, for three ratios of the tangents' to the
    normal's regulariser (impratio 1: mu_r = mu).  And the cone must MATTER: the same states stepped with pyramidal rows end up elsewhere."""
    W, T10 = _robot_through_the_physics_kernels(torch_cuda, oracle_mod, parity_log, "tail_biped_elliptic.xml", f"tail_biped_elliptic/impratio{impratio:g}", cone=True,
                                                overrides=None if impratio == 3.0 else dict(opt_impratio=np.array([impratio])))      # (the file says 3)
    if impratio != 3.0:
        return
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    from conftest import ROOT
    torch = torch_cuda
    base = Model.from_xml(os.path.join(ROOT, "tests", "assets", "tail_biped_elliptic.xml"), sim_dt=0.002)
    assert int(np.asarray(base.a["opt_cone"]).reshape(-1)[0]) == 1
    out = []
    for cone in (1, 0):
        m = Model({**base.a, "opt_cone": np.asarray([cone], np.int32)})
        b = engine.Batch(m, 4)
        q = np.tile(np.asarray(m.a["key_qpos"], np.float64), (4, 1))
        v = np.zeros((4, m.nv)); v[:, 0] = 0.8; v[:, 1] = -0.5          # sliding on the floor: friction decides where it ends up
        b.set_state(q, v, np.zeros((4, m.nv)))
        b.physics_step(torch.tensor(np.tile(np.asarray(m.a["key_ctrl"]), (4, 1)), dtype=torch.float32, device="cuda"), 50)
        out.append(b.get_state()[0][0].copy())
        b.close()
    assert np.isfinite(out[0]).all() and np.abs(out[0][:2] - out[1][:2]).max() > 1e-4, (out[0][:3], out[1][:3])


@pytest.mark.parametrize("kinds", [("sphere", "sphere"), ("capsule", "capsule"), ("capsule", "sphere")], ids="-".join)
def test_primitive_feet_on_a_height_field(torch_cuda, oracle_mod, parity_log, kinds):
    """sphere / capsule foot colliders on the height-field floor (mjx hfield_sphere / hfield_capsule as the oracle
    restates them: the primitive against the prism of every cell under its bounding sphere, the deepest one / two contacts kept)
    through the kernels' own instantiation (HF = 2): contacts of one forward pass, then the state after one substep and ten."""
    from open_duck_playground_amd import engine
    torch = torch_cuda
    model = _prim_feet_variant("rough_terrain_backlash", kinds)
    om = oracle_mod.OracleModel(model.blob())
    om32 = oracle_mod.OracleModel(model.blob(), f32=True)
    n = 48
    rng = np.random.default_rng(43)
    qpos, qvel = _random_states(model, n, rng)
    for e in range(n):
        qpos[e, :2] = rng.uniform(-6.0, 6.0, 2)      # anywhere on the terrain
        if e % 4 == 3:
            continue                                  # a quarter stays where the random pose put it
        d = oracle_mod.OracleData(om)
        for _ in range(5):                            # the others are lowered until a foot is 0.3 ... 3 mm in the terrain
            d["qpos"][: om.nq] = qpos[e]; d.forward()
            qpos[e, 2] -= min(np.array(d["contact_dist"][:8]).min(), 0.05) + rng.uniform(3e-4, 3e-3)
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
    W = dict(dist=0.0, pos=0.0, normal=0.0, qpos=0.0, qvel=0.0, qpos10=0.0, qvel10=0.0)
    n_contact = n_two = n_ill = 0
    prng = np.random.default_rng(44)
    for nsub in (1, 10):
        b = engine.Batch(model, n)
        b.set_state(qpos, qvel, np.zeros((n, model.nv)))
        b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), nsub)
        gq, gv, _ = b.get_state()
        for e in range(n):
            ds = _oracle_step(oracle_mod, om, qpos[e], qvel[e], np.zeros(model.nv), ctrl[e], nsub)
            sfx = "" if nsub == 1 else "10"
            if _rel(gv[e], ds["qvel"][: om.nv], 1.0).max() > (6e-5 if nsub == 1 else 1.5e-4):
                worst = _substep_sensitivity(oracle_mod, om, om32, qpos[e], qvel[e], ctrl[e], nsub, prng)
                assert worst > 1e-4, (e, nsub, worst, _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
                n_ill += 1
                continue
            W["qpos" + sfx] = max(W["qpos" + sfx], _rel(gq[e], ds["qpos"][: om.nq], 1e-2).max())
            W["qvel" + sfx] = max(W["qvel" + sfx], _rel(gv[e], ds["qvel"][: om.nv], 1.0).max())
        if nsub == 1:
            img = b.lds_image()
            o_cd, o_cr, o_fr = b.lds_offset("contact_dist"), b.lds_offset("contact_r"), b.lds_offset("scr")
            for e in range(n):
                d = oracle_mod.OracleData(om)
                d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
                d.forward()
                cd_o = np.array(d["contact_dist"][:12]); cd_g = img[e][o_cd: o_cd + 12]
                n_contact += int((cd_o[:8] < 0).any()); n_two += int((cd_o[:8] < 0).sum() >= 2)
                assert ((cd_o[:8] == 1.0) == (cd_g[:8] == 1.0)).all(), (e, cd_o, cd_g)
                near = (np.abs(cd_o) < 0.05) & (np.arange(12) < 8)
                # a candidate within 1e-6 of the kept one may take its place in float32: depth compared, position only when the depths agree
                if near.any():
                    W["dist"] = max(W["dist"], np.abs(cd_g[near] - cd_o[near]).max())
                for c in np.flatnonzero(cd_o[:8] < 0):
                    W["pos"] = max(W["pos"], np.abs(img[e][o_cr + 3 * c: o_cr + 3 * c + 3] + qpos[e, :3] - np.array(d["contact_pos"][3 * c: 3 * c + 3])).max())
        b.close()
    assert n_contact >= n // 3 and (n_two >= 4 or "capsule" not in kinds), (n_contact, n_two)
    # (the robots stand up to 8 m from the origin: float32 world coordinates carry 5e-7 m there, and a clipped capsule contact's position
    #  follows the base's x / y; measured: dist 7.8e-7, pos 5.3e-6, qpos 4.4e-6, qvel 1.9e-5, qvel10 1.1e-4, qpos10 1.3e-4 -- relative with
    #  a floor of 1e-2: 1.7e-6 rad on a backlash joint whose value is 0.013 rad, tools/gpu_prim_hfield_debug.py capsule capsule 47)
    parity_log.check("prim_feet_hfield/" + "-".join(kinds), dict(dist=1.5e-6, pos=1.5e-5, qpos=1e-5, qvel=6e-5, qpos10=3e-4, qvel10=1.5e-4, ill_fraction=0.1), ill_fraction=n_ill / (2 * n), **{k: v for k, v in W.items() if k != "normal"})
