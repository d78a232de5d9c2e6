"""Parity debugging aid (test infrastructure, not collected by pytest): per-stage max errors of the HIP forward pass vs the
oracle (prints, no asserts).  `python tests/debug_gpu_stages.py` on a GPU box.
NOTE (round 2): in the image of a forward pass that computed the sensors (the last substep), the first 46 floats of the
"jv" rows hold sensordata (csrc Shape::O_SENS aliases O_JV) and the first 2 * njnt floats of "jar" held sin/cos during
P0 / P1; "M" / "HL" / "cdof" are on the reduced (backlash twins merged) dof tree -- see tables.reduced_layout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))  # ROOT = repo root (this file lives in tests/)
import numpy as np, torch
import oracle as O
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
from open_duck_playground_amd.tables import build_kernel_tables
from test_gpu_parity import _random_states, _rel

np.set_printoptions(precision=5, suppress=True, linewidth=220)
task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = load_task_model(task)
n = 16
rng = np.random.default_rng(7)
qpos, qvel = _random_states(model, n, rng)
warm = rng.normal(0, 5.0, (n, model.nv))
ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.4, 0.4, (n, 14))
cfg = engine.default_config(); cfg.lanes_per_env = lanes
b = engine.Batch(model, n, cfg)
b.set_state(qpos, qvel, warm)
t = time.time()
b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
torch.cuda.synchronize(); print("launch+sync s", time.time() - t)
gq, gv, gw = b.get_state()
img = b.lds_image()
om = O.OracleModel(model.blob())
nv, nb = model.nv, model.nbody
tabs = build_kernel_tables(model.a)
Mi, Mj = tabs["k_M_i"], tabs["k_M_j"]
names = ("xpos", "M", "qfrc_smooth", "qacc_smooth", "contact_dist", "efc_D", "efc_aref", "qacc", "sensordata", "actuator_force", "scr", "x", "jar", "search", "Ma")
o = {k: b.lds_offset(k) for k in names}
for e in range(n):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    L = img[e]
    xpos = L[o["xpos"]: o["xpos"] + 3 * nb].reshape(3, nb).T
    r = {}
    r["xpos"] = np.abs(xpos - d["xpos"][: 3 * nb].reshape(nb, 3)).max()
    r["M"] = _rel(L[o["M"]: o["M"] + len(Mi)], d.M()[Mi, Mj], 1e-4).max()
    r["qfs"] = _rel(L[o["qfrc_smooth"]: o["qfrc_smooth"] + nv], d["qfrc_smooth"][:nv], 1e-2).max()
    r["qas"] = _rel(L[o["qacc_smooth"]: o["qacc_smooth"] + nv], d["qacc_smooth"][:nv], 1.0).max()
    r["actf"] = np.abs(L[o["actuator_force"]: o["actuator_force"] + 14] - d["actuator_force"][:14]).max()
    r["dist"] = np.abs(L[o["contact_dist"]: o["contact_dist"] + 8] - d["contact_dist"][:8]).max()
    nefc = d.i("nefc")
    live = np.abs(d.J()).sum(axis=1) > 0
    Dg = L[o["efc_D"]: o["efc_D"] + nefc]; Ag = L[o["efc_aref"]: o["efc_aref"] + nefc]
    r["rows_same"] = bool((((Dg > 0) == live)[14:]).all())
    both = live & (Dg > 0)
    r["D"] = _rel(Dg[both], d["efc_D"][:nefc][both], 1e-6).max()
    r["aref"] = _rel(Ag[both], d["efc_aref"][:nefc][both], 1.0).max()
    r["qacc"] = _rel(L[o["qacc"]: o["qacc"] + nv], d["qacc"][:nv], 5.0).max()
    r["sens"] = _rel(L[o["sensordata"]: o["sensordata"] + 46], d["sensordata"][:46], 1.0).max()
    misc = L[o["scr"] + 156: o["scr"] + 160]
    r["alpha"] = (float(misc[1]), float(d["ls_alpha"][0])); r["warm"] = (int(misc[2]), d.i("warm_used")); r["cost0"] = (float(misc[3]), float(d["solver_cost0"][0]))
    d2 = O.OracleData(om)
    d2["qpos"][: om.nq] = qpos[e]; d2["qvel"][:nv] = qvel[e]; d2["qacc_warmstart"][:nv] = warm[e]
    d2.env_physics_step(ctrl[e], 1)
    r["qpos"] = _rel(gq[e], d2["qpos"][: om.nq], 1e-2).max(); r["qvel"] = _rel(gv[e], d2["qvel"][:nv], 1.0).max()
    print(e, {k: (float(f"{v:.3g}") if isinstance(v, (float, np.floating)) else v) for k, v in r.items()})
    if e == 0:
        print(" qacc gpu", L[o["qacc"]: o["qacc"] + nv]); print(" qacc ora", d["qacc"][:nv])
        print(" sens gpu", L[o["sensordata"]: o["sensordata"] + 46]); print(" sens ora", d["sensordata"][:46])

r0c = 14 + len(tabs["k_lim_jnt"])
# ---- solver internals for env 1: expected gradient / search from the oracle's dense quantities
for e in (0, 3):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    nefc = d.i("nefc"); J = d.J(); D = d["efc_D"][:nefc]; aref = d["efc_aref"][:nefc]; R = d["efc_R"][:nefc]; fl = d["efc_frictionloss"][:nefc]
    M_ = d.M(); qfs = d["qfrc_smooth"][:nv]; qas = d["qacc_smooth"][:nv]
    x0 = warm[e] if d.i("warm_used") else qas
    jar = J @ x0 - aref
    f = np.zeros(nefc); act = np.zeros(nefc)
    for r in range(nefc):
        if r < 14:
            rf = R[r] * fl[r]
            if jar[r] <= -rf: f[r] = fl[r]
            elif jar[r] >= rf: f[r] = -fl[r]
            else: f[r] = -D[r] * jar[r]; act[r] = 1
        elif jar[r] < 0 and np.abs(J[r]).sum() > 0:
            f[r] = -D[r] * jar[r]; act[r] = 1
    grad = M_ @ x0 - qfs - J.T @ f
    H = M_ + J.T @ np.diag(D * act) @ J
    search = -np.linalg.solve(H, grad)
    L = img[e]
    sg = L[o["search"]: o["search"] + nv]
    print("env", e, "search gpu", sg); print("       search ora", search)
    print("  active contact rows", np.nonzero(act[r0c:])[0])

# ---- K / FF blocks of env 1 from the kernel's own W, D, JAR (consistency of the reductions)
e = 1
L = img[e]
oW, oD, oJ, oJV, oS = b.lds_offset("W"), b.lds_offset("efc_D"), b.lds_offset("jar"), b.lds_offset("jv"), b.lds_offset("scr")
W = L[oW: oW + 288].reshape(48, 6); Dr = L[oD + r0c: oD + r0c + 48]; jar = L[oJ + r0c: oJ + r0c + 48]
act = np.where((Dr > 0) & (jar < 0), Dr, 0.0)
for f in range(2):
    K = sum(act[r] * np.outer(W[r], W[r]) for r in range(16 * f, 16 * f + 16))
    Kg = L[oS + 24 + 36 * f: oS + 24 + 36 * f + 36].reshape(6, 6)
    print("foot", f, "K err", np.abs(K - Kg).max(), "K max", np.abs(K).max())
print("jar contact", jar[:32]); print("act", act[:32])

# ---- reconstruct H = L^T D L from the kernel's factor (virtual layout) and compare with the dense expectation
Hi, Hj = tabs["k_H_i"], tabs["k_H_j"]
oH = b.lds_offset("HL")
for e in (0, 3):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    nefc = d.i("nefc"); J = d.J(); D = d["efc_D"][:nefc]; aref = d["efc_aref"][:nefc]; R = d["efc_R"][:nefc]; fl = d["efc_frictionloss"][:nefc]
    x0 = warm[e] if d.i("warm_used") else d["qacc_smooth"][:nv]
    jar = J @ x0 - aref
    act = np.zeros(nefc)
    for r in range(nefc):
        if r < 14:
            act[r] = 1.0 if abs(jar[r]) < R[r] * fl[r] else 0.0
        elif jar[r] < 0 and np.abs(J[r]).sum() > 0:
            act[r] = 1
    H = d.M() + J.T @ np.diag(D * act) @ J
    Lm = np.eye(nv); Dm = np.zeros(nv)
    HLg = img[e][oH: oH + len(Hi)]
    for p, (i, j) in enumerate(zip(Hi, Hj)):
        if i == j: Dm[i] = HLg[p]
        else: Lm[i, j] = HLg[p]
    Hg = Lm.T @ np.diag(Dm) @ Lm
    err = np.abs(Hg - H)
    print("env", e, "H recon err max", err.max(), "at", np.unravel_index(err.argmax(), err.shape), "H max", np.abs(H).max())
    bad = np.argwhere(err > 1e-3 * np.abs(H).max())
    print("  bad entries", bad[:20].tolist())

# ---- gradient and foot wrench sums
oMa = b.lds_offset("Ma")
for e in (0, 3):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    nefc = d.i("nefc"); J = d.J(); D = d["efc_D"][:nefc]; aref = d["efc_aref"][:nefc]; R = d["efc_R"][:nefc]; fl = d["efc_frictionloss"][:nefc]
    x0 = warm[e] if d.i("warm_used") else d["qacc_smooth"][:nv]
    jar = J @ x0 - aref
    f = np.zeros(nefc)
    for r in range(nefc):
        if r < 14:
            rf = R[r] * fl[r]
            f[r] = fl[r] if jar[r] <= -rf else (-fl[r] if jar[r] >= rf else -D[r] * jar[r])
        elif jar[r] < 0 and np.abs(J[r]).sum() > 0:
            f[r] = -D[r] * jar[r]
    grad = d.M() @ x0 - d["qfrc_smooth"][:nv] - J.T @ f
    gg = img[e][oMa: oMa + nv]
    print("env", e, "grad gpu", gg); print("      grad ora", grad)
    Wk = img[e][oW: oW + 288].reshape(48, 6); fk = img[e][oJV + r0c: oJV + r0c + 48]
    print("  contact forces gpu", fk[:32]); print("  contact forces ora", f[r0c:r0c+32])
    for ft in range(2):
        FFe = sum(Wk[r] * fk[r] for r in range(16 * ft, 16 * ft + 16))
        print("  FF", ft, "gpu", img[e][oS + 12 + 6 * ft: oS + 18 + 6 * ft], "exp", FFe)

# ---- line-search start point: derivative and curvature at alpha = 0
for e in (0, 3):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][:nv] = qvel[e]; d["qacc_warmstart"][:nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    nefc = d.i("nefc"); J = d.J(); D = d["efc_D"][:nefc]; aref = d["efc_aref"][:nefc]; R = d["efc_R"][:nefc]; fl = d["efc_frictionloss"][:nefc]
    x0 = warm[e] if d.i("warm_used") else d["qacc_smooth"][:nv]
    jar = J @ x0 - aref
    sg = img[e][o["search"]: o["search"] + nv].astype(np.float64)
    jv = J @ sg
    Ma = d.M() @ x0; qfs_ = d["qfrc_smooth"][:nv]
    q1 = sg @ Ma - sg @ qfs_; q2 = 0.5 * sg @ (d.M() @ sg)
    for r in range(nefc):
        if r < 14:
            rf = R[r] * fl[r]
            if jar[r] <= -rf: q1 += -fl[r] * jv[r]
            elif jar[r] >= rf: q1 += fl[r] * jv[r]
            else: q1 += D[r] * jv[r] * jar[r]; q2 += 0.5 * D[r] * jv[r] ** 2
        elif jar[r] < 0 and np.abs(J[r]).sum() > 0:
            q1 += D[r] * jv[r] * jar[r]; q2 += 0.5 * D[r] * jv[r] ** 2
    misc = img[e][oS + 156: oS + 164]
    print("env", e, "p0 deriv0 gpu", misc[4], "exp", q1, "| deriv1 gpu", misc[5], "exp", 2 * q2, "| alpha gpu", misc[1], "newton", -q1 / (2 * q2), "gtol", misc[7])
    jvg = img[e][oJV + r0c: oJV + r0c + 48]; jarg = img[e][oJ + r0c: oJ + r0c + 48]; Dg = img[e][oD + r0c: oD + r0c + 48]
    print("  jv contact gpu", jvg[:32]); print("  jv contact exp", jv[r0c:r0c+32]); print("  jar gpu", jarg[:16]); print("  jar exp", jar[r0c:r0c+16])
    print("  twist gpu", img[e][oS: oS + 12])
    misc = img[e][oS + 156: oS + 172]
    print("  qg1 gpu", misc[13], "exp", sg @ Ma - sg @ qfs_, "qg2 gpu", misc[14], "exp", 0.5 * sg @ (d.M() @ sg))
    # friction + limit row contributions
    c1 = c2 = 0.0
    for r in range(r0c):
        if r < 14:
            rf = R[r] * fl[r]
            if jar[r] <= -rf: c1 += -fl[r] * jv[r]
            elif jar[r] >= rf: c1 += fl[r] * jv[r]
            else: c1 += D[r] * jv[r] * jar[r]; c2 += 0.5 * D[r] * jv[r] ** 2
        elif jar[r] < 0 and np.abs(J[r]).sum() > 0:
            c1 += D[r] * jv[r] * jar[r]; c2 += 0.5 * D[r] * jv[r] ** 2
    cc1 = sum(D[r] * jv[r] * jar[r] for r in range(r0c, nefc) if jar[r] < 0 and np.abs(J[r]).sum() > 0)
    print("  rows: fl+lim d0", c1, "contact d0", cc1, "gpu rows total", misc[4] - misc[13])
    print("  contact d0 from gpu arrays", float(np.sum(Dg * jvg * jarg * (jarg < 0) * (Dg > 0))))
    # limit rows from gpu arrays: D, aref rows 14..27
    Dl = img[e][oD + 14: oD + r0c]; print("  lim D gpu", Dl); print("  lim D exp", D[14:r0c] * (np.abs(J[14:r0c]).sum(axis=1) > 0)); print("  lim jar exp", jar[14:r0c]); print("  lim jv exp", jv[14:r0c])
    print("  fl jar exp", jar[:14]); print("  fl rf", (R * fl)[:14]); print("  fl jv exp", jv[:14])
    print("  per-row exp", [float(f"{D[r] * jv[r] * jar[r]:.1f}") for r in range(r0c, nefc) if jar[r] < 0 and np.abs(J[r]).sum() > 0], [r for r in range(r0c, nefc) if jar[r] < 0 and np.abs(J[r]).sum() > 0])
    print("  per-row gpu", [float(f"{v:.1f}") for v in (Dg * jvg * jarg * (jarg < 0) * (Dg > 0)) if v != 0])
    print("  D exp", D[r0c:r0c+16], "D gpu", Dg[:16])
    print("  CHECK", Dg[:4], jvg[:4], jarg[:4], (Dg * jvg * jarg)[:4], "| exp jv", jv[r0c:r0c+4], "exp jar", jar[r0c:r0c+4])
    oC = b.lds_offset("cdof"); cd = img[e][oC: oC + 6 * nv].reshape(6, nv)
    fm = tabs["k_foot_dofmask"]
    for ft in range(2):
        print("  twist", ft, "gpu", img[e][oS + 6 * ft: oS + 6 * ft + 6], "exp", (cd * fm[ft][None, :]) @ sg)
    print("  W row0 gpu", img[e][oW: oW + 6], " J row 28 exp via W.twist", Wk[0] @ ((cd * fm[0][None, :]) @ sg), "J@s", jv[r0c])
    Cm = cd * fm[0][None, :]
    cands = {"search": sg, "x0": x0, "qacc": img[e][o["qacc"]: o["qacc"] + nv], "qas": d["qacc_smooth"][:nv], "warm": warm[e], "grad": img[e][oMa: oMa + nv], "qvel": qvel[e]}
    for nm, vec in cands.items():
        print("   cand", nm, Cm @ np.asarray(vec, np.float64))
    print("  FULL jar gpu", jarg[:32]); print("  FULL jar exp", jar[r0c:r0c + 32]); print("  FULL D gpu", Dg[:32]); print("  FULL D exp", (D * (np.abs(J).sum(axis=1) > 0))[r0c:r0c + 32])
    print("  FULL jv gpu", jvg[:32]); print("  FULL jv exp", jv[r0c:r0c + 32])
