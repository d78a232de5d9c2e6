"""Diagnostic (GPU box): replays ONE env step of the resynchronised sequence substep by substep -- GPU and oracle start every
substep from the oracle's state -- and prints where the two part.  python tools/gpu_env_outlier_substeps.py task t env [standing]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import oracle as O
from open_duck_playground_amd import engine
import test_gpu_env as T

task, t_target, i_target = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
standing = len(sys.argv) > 4
seed_reset, seed_act, nsteps_ep = (11, 2, 25) if standing else (9, 0, 25)
O.build()
from open_duck_playground_amd.model import load_task_model
model = load_task_model(task)
om = O.OracleModel(model.blob()); prm = O.OraclePRM(engine.load_prm())
envs = [O.OracleEnv(om, prm, standing=standing) for _ in range(32)]
for i, e in enumerate(envs):
    e.cfg["episode_length"][0] = nsteps_ep
    e.reset(seed_reset, i)
rng = np.random.default_rng(seed_act)
for t in range(t_target + 1):
    act = rng.uniform(-1, 1, (32, 14)).astype(np.float32)
    if t == t_target:
        break
    for i, e in enumerate(envs):
        e.step(act[i])
e = envs[i_target]
c0 = e.clone(); c0.cfg["n_substeps"][0] = 0; c0.step(act[i_target])
mt = np.array(c0["motor_targets"][:14])
d = O.OracleData(om)
d["qpos"][: om.nq] = c0.data["qpos"][: om.nq]; d["qvel"][: om.nv] = c0.data["qvel"][: om.nv]; d["qacc_warmstart"][: om.nv] = c0.data["qacc_warmstart"][: om.nv]
b = engine.Batch(model, 1)
nv = model.nv
o = {k: b.lds_offset(k) for k in ("contact_dist", "efc_D", "efc_aref", "qacc", "qacc_smooth", "jar", "contact_r")}
ctrl = torch.tensor(mt[None], dtype=torch.float32, device="cuda")
for s in range(10):
    q0 = np.array(d["qpos"][: om.nq]); v0 = np.array(d["qvel"][:nv]); w0 = np.array(d["qacc_warmstart"][:nv])
    b.set_state(q0[None], v0[None], w0[None])
    b.physics_step(ctrl, 1)
    gq, gv, gw = b.get_state()
    img = b.lds_image()[0]
    d["ctrl"][:14] = mt
    d.step()
    nefc = d.i("nefc")
    qacc_g = img[o["qacc"]: o["qacc"] + nv]; qacc_o = np.array(d["qacc"][:nv])
    qas_g = img[o["qacc_smooth"]: o["qacc_smooth"] + nv]; qas_o = np.array(d["qacc_smooth"][:nv])
    cd_g = img[o["contact_dist"]: o["contact_dist"] + 12]; cd_o = np.array(d["contact_dist"][:12])
    D_g = img[o["efc_D"]: o["efc_D"] + nefc]; D_o = np.array(d["efc_D"][:nefc]); J = d.J(); live = np.abs(J).sum(axis=1) > 0
    ar_g = img[o["efc_aref"]: o["efc_aref"] + nefc]; ar_o = np.array(d["efc_aref"][:nefc])
    eq = np.abs(qacc_g - qacc_o) / np.maximum(np.abs(qacc_o), 5.0)
    print(f"substep {s}: qacc err {eq.max():.2e} (dof {eq.argmax()})  qacc_smooth err {(np.abs(qas_g - qas_o) / np.maximum(np.abs(qas_o), 1)).max():.2e}  "
          f"warm_used {d.i('warm_used')} ls_iters {d.i('ls_iters')} alpha {d['ls_alpha'][0]:.6g} cost0 {d['solver_cost0'][0]:.8g} cost1 {d['solver_cost1'][0]:.8g}")
    print("    oracle dist", np.round(cd_o[:8], 7).tolist(), "\n    gpu    dist", np.round(cd_g[:8].astype(float), 7).tolist())
    act_rows_g = set(np.flatnonzero(D_g > 0)); act_rows_o = set(np.flatnonzero(live))
    if act_rows_g != act_rows_o:
        print("    ACTIVE ROW SETS DIFFER: gpu-only", sorted(act_rows_g - act_rows_o), "oracle-only", sorted(act_rows_o - act_rows_g))
    both = sorted(act_rows_g & act_rows_o)
    if both:
        print(f"    D err {(np.abs(D_g[both] - D_o[both]) / np.abs(D_o[both])).max():.2e}  aref err {(np.abs(ar_g[both] - ar_o[both]) / np.maximum(np.abs(ar_o[both]), 1)).max():.2e}")
    if both and (np.abs(ar_g[both] - ar_o[both]) / np.maximum(np.abs(ar_o[both]), 1)).max() > 0.05:
        bad = [r for r in both if abs(ar_g[r] - ar_o[r]) / max(abs(ar_o[r]), 1) > 0.05]
        nf, nl = d.i("nf"), d.i("nl")
        print("    rows whose aref differs:", bad, "(friction rows", nf, "limit rows", nl, "-> contact row r belongs to contact (r - nf - nl) // 4)")
        for r in bad[:8]:
            print(f"      row {r}: aref gpu {ar_g[r]:.5g} oracle {ar_o[r]:.5g}  D gpu {D_g[r]:.6g} oracle {D_o[r]:.6g}  efc_pos oracle {d['efc_pos'][r]:.7g}")
        cp = np.array(d["contact_pos"][:36]).reshape(12, 3); fr = np.array(d["contact_frame"][:108]).reshape(12, 9)
        for c in range(8):
            print(f"      oracle contact {c}: dist {cd_o[c]:.7f} pos {np.round(cp[c], 5).tolist()} normal {np.round(fr[c][:3], 5).tolist()}")
        if "contact_r" in o:
            print("      gpu contact_r:", np.round(img[o["contact_r"]: o["contact_r"] + 36].reshape(12, 3)[:8], 5).tolist())
    print(f"    state after: qpos err {np.abs(gq[0] - d['qpos'][:om.nq]).max():.2e} qvel err {(np.abs(gv[0] - d['qvel'][:nv]) / np.maximum(np.abs(d['qvel'][:nv]), 1)).max():.2e}")

print("---- free-running k substeps from the env step's initial state (no resync)")
d2 = O.OracleData(om)
q0 = np.array(c0.data["qpos"][: om.nq]); v0 = np.array(c0.data["qvel"][:nv]); w0 = np.array(c0.data["qacc_warmstart"][:nv])
d2["qpos"][: om.nq] = q0; d2["qvel"][:nv] = v0; d2["qacc_warmstart"][:nv] = w0
# perturbed oracle twins: how fast does rounding-level noise grow in the ORACLE itself?
tw = []
prng = np.random.default_rng(5)
for k in range(4):
    dd = O.OracleData(om)
    dd["qpos"][: om.nq] = q0 + 1e-6 * prng.standard_normal(om.nq) * np.maximum(np.abs(q0), 0.1)
    dd["qvel"][:nv] = v0 + 5e-6 * prng.standard_normal(nv) * np.maximum(np.abs(v0), 1.0); dd["qacc_warmstart"][:nv] = w0
    tw.append(dd)
for k in range(1, 11):
    b.set_state(q0[None], v0[None], w0[None])
    b.physics_step(ctrl, k)
    gq, gv, gw = b.get_state()
    img = b.lds_image()[0]
    d2["ctrl"][:14] = mt; d2.step()
    for dd in tw:
        dd["ctrl"][:14] = mt; dd.step()
    qacc_g = img[o["qacc"]: o["qacc"] + nv]; qacc_o = np.array(d2["qacc"][:nv])
    e_acc = (np.abs(qacc_g - qacc_o) / np.maximum(np.abs(qacc_o), 5.0)).max()
    e_v = (np.abs(gv[0] - d2["qvel"][:nv]) / np.maximum(np.abs(d2["qvel"][:nv]), 1)).max()
    t_acc = max((np.abs(np.array(dd["qacc"][:nv]) - qacc_o) / np.maximum(np.abs(qacc_o), 5.0)).max() for dd in tw)
    t_v = max((np.abs(np.array(dd["qvel"][:nv]) - d2["qvel"][:nv]) / np.maximum(np.abs(d2["qvel"][:nv]), 1)).max() for dd in tw)
    if e_acc > 1e-3:
        cd_g = img[o["contact_dist"]: o["contact_dist"] + 12]
        print("    nominal dist", np.round(np.array(d2["contact_dist"][:8]), 7).tolist(), "warm_used", d2.i("warm_used"), "alpha", d2["ls_alpha"][0], "cost0", d2["solver_cost0"][0], "cost1", d2["solver_cost1"][0])
        for dd in tw:
            print("    twin    dist", np.round(np.array(dd["contact_dist"][:8]), 7).tolist(), "warm_used", dd.i("warm_used"), "alpha", dd["ls_alpha"][0], "cost0", dd["solver_cost0"][0], "cost1", dd["solver_cost1"][0])
        print("    gpu     dist", np.round(cd_g[:8].astype(float), 7).tolist())
        nf = d2.i("nf"); nl = d2.i("nl")
        print("    nominal limit pos", np.round(np.array(d2["efc_pos"][nf:nf + nl]), 7).tolist())
        print("    twin0   limit pos", np.round(np.array(tw[0]["efc_pos"][nf:nf + nl]), 7).tolist())
    print(f"k={k}: gpu-vs-oracle qacc {e_acc:.2e} qvel {e_v:.2e} | oracle twins qacc {t_acc:.2e} qvel {t_v:.2e} | ls_iters {d2.i('ls_iters')} twins {[dd.i('ls_iters') for dd in tw]}")
