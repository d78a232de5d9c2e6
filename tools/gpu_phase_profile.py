"""Per-phase shader-clock breakdown of the fused step kernel (needs the -DODK_PROFILE build):
    ODK_LIB=open_duck_playground_amd/csrc/libodk_prof.so python tools/gpu_phase_profile.py [task] [lanes] [nenv]
Prints mean cycles per phase per env step (10 forwards), measured by lane 0 of every env."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model

task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nenv = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
NAMES = ["P1 kinematics+cinert+cdof", "P2 crb*cdof, vel prefix", "P3 M entries", "P4 cvel/cacc/cfrc", "P5 bias+act (copy M)",
         "P6a factor M", "P7 plane-convex x2", "P7b foot-foot OBB", "P8 rows", "P9a warm/smooth costs", "P9b forces,K,grad,T",
         "P9c Hessian entries", "P9d factor H", "P9e ls setup (mv, jv)", "P9f line search", "P10 sensors", "P6b solve M", "P9d' solve H"]
ORDER = [0, 1, 2, 3, 4, 15, 16, 5, 6, 7, 8, 9, 10, 11, 12, 17, 13, 14]
IDX = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 15: 5, 16: 16, 5: 6, 6: 7, 7: 8, 8: 9, 9: 10, 10: 11, 11: 12, 12: 17, 17: 15, 13: 13, 14: 14}
model = load_task_model(task)
cfg = engine.default_config(); cfg.noise_level = 0.0; cfg.push_enable = 0.0; cfg.lanes_per_env = lanes
b = engine.Batch(model, nenv, cfg)
b.reset(0)
act = torch.empty(nenv, 14, device="cuda").uniform_(-1, 1)
for _ in range(20):
    b.step(act.uniform_(-1, 1))
b.L.odk_set_debug_dump(1)
b.step(act.uniform_(-1, 1))
torch.cuda.synchronize()
img = b.lds_image()
o = b.lds_offset("scr") + 172
prof = img[:, o:o + 20].astype(np.float64)
prof2 = img[:, o + 20:o + 36].astype(np.float64).mean(axis=0)
mean = prof.mean(axis=0)
tot = mean[:18].sum()
labels = ["P0 sincos", "P1 top-down sweep (pose,cdof,cvel,cacc,cinert)", "P2 bottom-up sweep (crb,cfrc)", "P3 per-dof bias/act/qfrc_smooth",
          "P4 M entries", "P5 factor M", "P5 solve M + dense M row", "P7 plane-convex x2", "P7 foot-foot OBB cull", "P8 constraint rows",
          "P9 candidate twists/costs", "P9 forces, K blocks, grad, K*cdof", "P9 Hessian entries", "P9 factor H", "P9 solve H",
          "P9 line-search setup", "P9 line search", "P10 sensors (+tail)"]
print(f"{task} G={lanes} nenv={nenv}: cycles per env step (lane-0 clock, 10 forwards), total {tot:,.0f}")
for i in range(18):
    print(f"  {labels[i]:48s} {mean[i]:12,.0f}  {100 * mean[i] / tot:5.1f}%")
print(f"  {'env-step prologue (kernel start -> first substep)':48s} {mean[18]:12,.0f}  {100 * mean[18] / tot:5.1f}% of the substeps' total")
print(f"  {'env-step epilogue (last Euler -> kernel end)':48s} {mean[19]:12,.0f}  {100 * mean[19] / tot:5.1f}% of the substeps' total")
if "rough" in task:
    print(f"  height-field contacts (per env step = 10 forwards): hull setup {prof2[0]:,.0f}  cull pass {prof2[1]:,.0f}  "
          f"pair loop {prof2[3]:,.0f} cycles; loop iterations {prof2[4] / 10:.2f} per forward (longest row of the wave), list length of foot 0 {prof2[5] / 10:.2f}, pairs of foot 0 given the full test {prof2[6] / 10:.2f}")
    it = max(prof2[4], 1.0)
    names = ["select + prism", "face query (hull faces)", "27 Gauss-map tests", "passing pairs", "faces / polygons", "clip + manifold selection", "contact writes + merge"]
    print("  per pair-loop iteration (cycles): " + ", ".join(f"{nm} {prof2[8 + i] / it:,.0f}" for i, nm in enumerate(names)))
    print(f"  passing edge pairs per iteration (lane 0's row): {prof2[2] / it:.1f}")
    print(f"  of which: select + prism up to the end of the row switch {prof2[7] / it:,.0f}; writes + merge before the merges {prof2[15] / it:,.0f}")
