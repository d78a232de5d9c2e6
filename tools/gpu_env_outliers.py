"""Diagnostic (GPU box): which env steps of the resynchronised GPU-vs-oracle sequence exceed the parity bounds, in which slots, and
what the contact distances looked like on both sides.  python tools/gpu_env_outliers.py [task] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import oracle as O
from open_duck_playground_amd import engine
import test_gpu_env as T

task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
O.build()
def edit(cfg): cfg.episode_length = 25
torch_, model, b, envs, keep = T._mk(O, task, 32, edit)
engine.load_library().odk_set_debug_dump(1)
n = len(envs)
b.reset(seed=9)
for i, e in enumerate(envs): e.reset(9, i)
rng = np.random.default_rng(0)
names = [(0, 3, "gyro"), (3, 6, "acc"), (6, 13, "cmd"), (13, 27, "jpos"), (27, 41, "jvel"), (41, 83, "acts"), (83, 97, "mt"), (97, 99, "contact"), (99, 101, "phase"),
         (101, 104, "p.gyro"), (104, 107, "p.acc"), (107, 110, "p.grav"), (110, 113, "p.linvel"), (113, 116, "p.angvel"), (116, 130, "p.jpos"), (130, 144, "p.jvel"),
         (144, 145, "p.height"), (145, 159, "p.force"), (159, 161, "p.contact"), (161, 167, "p.feetvel"), (167, 169, "p.air"), (169, 209, "p.ref"), (209, 212, "p.imi")]
nout = 0
for t in range(steps):
    T._resync(b, envs, model)
    act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
    b.step(torch.tensor(act, device="cuda"))
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy(); rew = b.reward.cpu().numpy(); met = b.metrics.cpu().numpy()
    dbg = b.get_debug()
    for i, e in enumerate(envs):
        e.step(act[i])
        full = np.concatenate([obs[i][:0], priv[i]]); ref = np.array(e["priv"][:212])
        err = T._rel1(priv[i], ref)
        errs = {nm: float(err[a:bb].max()) for a, bb, nm in names}
        lim = {nm: (5e-3 if "acc" in nm else 5e-4) for _, _, nm in names}
        bad = {k: f"{v:.2e}" for k, v in errs.items() if v > lim[k]}
        r = float(T._rel1(rew[i], e["reward"][0])); m = T._rel1(met[i], e["metrics"][:8])
        if bad or r > 5e-4 or m.max() > 1e-3:
            nout += 1
            cd_o = np.array(e.data["contact_dist"][:12]); cd_g = dbg["contact_dist"][i]
            print(f"t={t} env={i} done={e['done'][0]} bad={bad} rew={r:.2e} met={np.round(m, 5).tolist()}")
            print("   oracle dist", np.round(cd_o[:8], 6).tolist()); print("   gpu    dist", np.round(cd_g[:8], 6).tolist())
print("outlier env-steps:", nout, "of", steps * n)
