"""Debug aid: height-field contacts of the leaning-robot states (tests/test_gpu_parity.py::_leaning_states), HIP path vs oracle,
contact by contact, for the envs whose contact distances differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import oracle as oracle_mod  # noqa: E402
from open_duck_playground_amd import engine  # noqa: E402
from open_duck_playground_amd.model import load_task_model  # noqa: E402
from test_gpu_parity import _contact_tie, _contacts, _leaning_states  # noqa: E402

model = load_task_model("rough_terrain_backlash")
om = oracle_mod.OracleModel(model.blob())
n = 64
rng = np.random.default_rng(77)
qpos, qvel, kinds = _leaning_states(oracle_mod, model, om, n, rng)
ctrl = np.tile(np.asarray(model.a["key_ctrl"]), (n, 1))
b = engine.Batch(model, n)
b.set_state(qpos, qvel * 0.2, np.zeros((n, model.nv)))
b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), 1)
img = b.lds_image()
o_cd, o_cr = b.lds_offset("contact_dist"), b.lds_offset("contact_r")
prng = np.random.default_rng(5)
for e in range(n):
    if kinds[e] == "air":
        continue
    d = oracle_mod.OracleData(om)
    d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = 0.2 * qvel[e]; d["ctrl"][:14] = ctrl[e]
    d.forward()
    cd_o, cd_g = np.array(d["contact_dist"][:8]), img[e][o_cd: o_cd + 8]
    act = (cd_o < 0) | (cd_g < 0)
    err = np.abs(cd_g[act] - cd_o[act]).max() if act.any() else 0.0
    tie = _contact_tie(oracle_mod, om, qpos[e], 0.2 * qvel[e], ctrl[e], prng, _contacts(d))
    print(f"env {e} {kinds[e]} tie {tie} err {err:.2e}")
    if err > 1e-5:
        print("   oracle dist", np.round(cd_o, 5)); print("   gpu    dist", np.round(cd_g, 5))
        po = np.array(d["contact_pos"][:24]).reshape(8, 3); pg = img[e][o_cr: o_cr + 24].reshape(8, 3) + qpos[e, :3]
        for c in range(8):
            if act[c]:
                print(f"   c{c} oracle pos {np.round(po[c], 4)} n {np.round(np.array(d['contact_frame'][9 * c: 9 * c + 3]), 4)} | gpu pos {np.round(pg[c], 4)}")
b.close()
