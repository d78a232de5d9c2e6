"""Debug aid: primitive-feet variant, substep by substep against the oracle; prints where the worst env starts to differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import oracle as oracle_mod  # noqa: E402
from open_duck_playground_amd import engine  # noqa: E402
from test_gpu_parity import _prim_feet_variant, _random_states, build_tables  # noqa: E402

kinds = tuple(sys.argv[1:3]) if len(sys.argv) > 2 else ("capsule", "capsule")
model = _prim_feet_variant("flat_terrain", kinds)
om = oracle_mod.OracleModel(model.blob())
n = 48
rng = np.random.default_rng(41)
qpos, qvel = _random_states(model, n, rng)
aq = build_tables(model)["k_act_qposadr"]
for e in range(0, n, 3):
    qpos[e] = np.asarray(model.a["key_qpos"]); qpos[e, 2] = 0.3
    qpos[e, int(aq[1])] = rng.uniform(0.4, 0.6); qpos[e, int(aq[10])] = rng.uniform(-0.6, -0.4); qpos[e, int(aq[0])] += rng.uniform(-0.3, 0.3)
for e in range(1, n, 3):
    d = oracle_mod.OracleData(om)
    for _ in range(4):
        d["qpos"][: om.nq] = qpos[e]; d.forward()
        qpos[e, 2] -= min(np.array(d["contact_dist"][:8]).min(), 0.05) + rng.uniform(3e-4, 3e-3)
ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.3, 0.3, (n, 14))
b = engine.Batch(model, n)
b.set_state(qpos, qvel, np.zeros((n, model.nv)))
ds = []
for e in range(n):
    d = oracle_mod.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["ctrl"][:14] = ctrl[e]
    ds.append(d)
ct = torch.tensor(ctrl, dtype=torch.float32, device="cuda")
o_cd = b.lds_offset("contact_dist"); o_cr = b.lds_offset("contact_r"); o_scr = b.lds_offset("scr"); o_qacc = b.lds_offset("qacc")
prev_q = qpos.copy()
for k in range(10):
    b.physics_step(ct, 1)
    gq, gv, gw = b.get_state()
    img = b.lds_image()
    errs = []
    for e in range(n):
        ds[e].env_physics_step(ctrl[e], 1)
        errs.append(np.abs(gv[e] - np.array(ds[e]["qvel"][: om.nv])).max())
    w = int(np.argmax(errs))
    print(f"substep {k}: worst env {w} qvel err {errs[w]:.3e}; gpu dist {np.round(img[w][o_cd:o_cd + 12], 6)}")
    print(f"     oracle dist {np.round(np.array(ds[w]['contact_dist'][:12]), 6)}")
    if errs[w] > 1e-3:
        d = ds[w]
        print("     gpu pos8", img[w][o_cr + 24: o_cr + 27] + prev_q[w, :3], "oracle", np.array(d["contact_pos"][24:27]))
        dv = gv[w] - np.array(d["qvel"][: om.nv])
        print("     qvel diff", np.round(dv, 5))
        print("     oracle nefc", d.i("nefc"), "efc_force", np.round(np.array(d["efc_force"][: d.i("nefc")]), 3))
    prev_q = gq.copy()
b.close()

# ---- resynchronised: env 33 (or argv[3]) from the oracle's own state before each substep
w = int(sys.argv[3]) if len(sys.argv) > 3 else 33
d = oracle_mod.OracleData(om)
d["qpos"][: om.nq] = qpos[w]; d["qvel"][: om.nv] = qvel[w]
from open_duck_playground_amd.tables import build_kernel_tables  # noqa: E402
tabs = build_kernel_tables(model.a)
nfl = len(tabs["k_fl_dof"])
for k in range(10):
    q, v, wm = (np.array(d[kk][:nn]) for kk, nn in (("qpos", om.nq), ("qvel", om.nv), ("qacc_warmstart", om.nv)))
    b1 = engine.Batch(model, 1)
    b1.set_state(q[None], v[None], wm[None])
    b1.physics_step(ct[w: w + 1], 1)
    g_q, g_v, g_w = b1.get_state()
    L = b1.lds_image()[0]
    df = oracle_mod.OracleData(om)
    df["qpos"][: om.nq] = q; df["qvel"][: om.nv] = v; df["qacc_warmstart"][: om.nv] = wm; df["ctrl"][:14] = ctrl[w]
    df.forward()
    d.env_physics_step(ctrl[w], 1)
    nefc = df.i("nefc")
    o = {kk: b1.lds_offset(kk) for kk in ("efc_D", "efc_aref", "qacc", "qacc_smooth", "jar")}
    J = df.J(); live = np.abs(J).sum(axis=1) > 0
    D_g, D_o = L[o["efc_D"]: o["efc_D"] + nefc], np.array(df["efc_D"][:nefc])
    a_g, a_o = L[o["efc_aref"]: o["efc_aref"] + nefc], np.array(df["efc_aref"][:nefc])
    qa_g, qa_o = L[o["qacc"]: o["qacc"] + om.nv], np.array(df["qacc"][: om.nv])
    qs_g, qs_o = L[o["qacc_smooth"]: o["qacc_smooth"] + om.nv], np.array(df["qacc_smooth"][: om.nv])
    print(f"resync substep {k}: qvel err {np.abs(g_v[0] - np.array(d['qvel'][: om.nv])).max():.3e} rows live {live.sum()} D rel {np.abs(D_g[live] / D_o[live] - 1).max():.2e} "
          f"aref err {np.abs(a_g[live] - a_o[live]).max():.2e} active-set same {bool(((D_g > 0) == live)[nfl:].all())} qacc_smooth err {np.abs(qs_g - qs_o).max():.2e} qacc err {np.abs(qa_g - qa_o).max():.3e} |qacc| {np.abs(qa_o).max():.1f}")
    if np.abs(a_g[live] - a_o[live]).max() > 0.1:
        bad = [r for r in np.flatnonzero(live) if abs(a_g[r] - a_o[r]) > 0.01]
        print("   rows", bad, "nfl", nfl, "nlim", len(tabs["k_lim_jnt"]), "aref gpu", a_g[bad], "orc", a_o[bad], "D", D_o[bad])
        print("   contact dist orc", np.array(df["contact_dist"][:12]))
        oc, osc, ow = b1.lds_offset("contact_r"), b1.lds_offset("scr"), b1.lds_offset("W")
        print("   pos8 gpu", L[oc + 24: oc + 27] + q[:3], "orc", np.array(df["contact_pos"][24:27]))
        print("   frame gpu (scr, may be overwritten)", np.round(L[osc: osc + 9], 5))
        print("   frame orc", np.round(np.array(df["contact_frame"][72:81]), 5))
        print("   W rows gpu", np.round(L[ow + 6 * 32: ow + 6 * 36].reshape(4, 6), 5))
        r0 = nfl + len(tabs["k_lim_jnt"])
        print("   J rows orc (first bad)", np.round(J[bad[0]], 4))
        print("   efc_vel orc", (J @ v)[bad], " qpos", np.round(q, 4))
    if np.abs(qa_g - qa_o).max() > 0.5:
        print("   qacc gpu", np.round(qa_g, 2)); print("   qacc orc", np.round(qa_o, 2))
        print("   oracle solver: ", {kk: df.i(kk) if kk in ("nefc",) else None for kk in ("nefc",)})
    b1.close()
