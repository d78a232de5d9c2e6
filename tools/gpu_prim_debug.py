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
o_cd = b.lds_offset("contact_dist")
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
b.close()
