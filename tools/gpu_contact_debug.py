"""Diagnostic (GPU box): contact distances / positions / normals of the HIP collision path next to the oracle's for a few states.
python tools/gpu_contact_debug.py footfoot|rough"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import oracle as O
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
from open_duck_playground_amd.tables import build_kernel_tables
import test_gpu_parity as TP
np.set_printoptions(precision=6, suppress=True, linewidth=220)
mode = sys.argv[1] if len(sys.argv) > 1 else "footfoot"
O.build()
if mode == "footfoot":
    model = load_task_model("flat_terrain"); a = model.a
    aq = build_kernel_tables(a)["k_act_qposadr"]; lroll, rroll = int(aq[1]), int(aq[10])
    grid = [(0.45, -0.45), (0.5, -0.5), (0.55, -0.55), (0.6, -0.6), (0.4, -0.55), (0.6, -0.3)]
    n = len(grid)
    qpos = np.tile(np.asarray(a["key_qpos"], np.float64), (n, 1)); qvel = np.zeros((n, model.nv))
    for e, (l, r) in enumerate(grid):
        qpos[e, 2] = 0.3; qpos[e, lroll] = l; qpos[e, rroll] = r
    c0 = 8
else:
    model = load_task_model("rough_terrain_backlash"); a = model.a
    n = 6
    rng = np.random.default_rng(7)
    qpos, qvel = TP._random_states(model, n, rng, airborne_frac=0.0); qvel *= 0
    c0 = 0
ctrl = np.asarray(a["key_ctrl"])[None].repeat(n, 0)
b = engine.Batch(model, n)
b.set_state(qpos, qvel, np.zeros((n, model.nv)))
b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
img = b.lds_image()
om = O.OracleModel(model.blob())
o_cd, o_cr, o_jv, o_scr = b.lds_offset("contact_dist"), b.lds_offset("contact_r"), b.lds_offset("jv"), b.lds_offset("scr")
for e in range(n):
    d = O.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    nc = 4 if mode == "footfoot" else 8
    cd_o = np.array(d["contact_dist"][c0:c0 + nc]); cp_o = np.array(d["contact_pos"][3 * c0: 3 * (c0 + nc)]).reshape(nc, 3)
    fr_o = np.array(d["contact_frame"][9 * c0: 9 * (c0 + nc)]).reshape(nc, 9)[:, :3]
    L = img[e]
    cd_g = L[o_cd + c0: o_cd + c0 + nc]; cr_g = L[o_cr + 3 * c0: o_cr + 3 * (c0 + nc)].reshape(nc, 3) + qpos[e, :3]
    if mode == "footfoot":
        fr_g = np.tile(L[o_scr: o_scr + 3], (4, 1))
    else:
        fr_g = L[o_jv: o_jv + 72].reshape(8, 9)[:, :3]
    print(f"--- env {e}")
    print(" oracle dist", cd_o); print(" gpu    dist", cd_g)
    print(" oracle pos\n", cp_o); print(" gpu pos\n", cr_g)
    print(" oracle n\n", fr_o); print(" gpu n\n", fr_g)
