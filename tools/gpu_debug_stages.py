"""First-contact GPU debug: per-stage max errors of the HIP forward pass vs the oracle (prints, no asserts)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
