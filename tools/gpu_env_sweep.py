"""The resynchronised env-step comparison of tests/test_gpu_env.py::test_step_sequence_with_resync at a larger size (GPU box):
    python tools/gpu_env_sweep.py [task = flat_terrain | a robot's XML under tests/assets/, e.g. biped12.xml] [n_envs = 256] [steps = 30] [seed = 9] [cone = pyramidal | elliptic]
prints the judged worst errors, the fraction of env steps set aside by the oracle's own sensitivity and the outliers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O  # noqa: E402
import test_gpu_env as T  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 9
cone = sys.argv[5] if len(sys.argv) > 5 else "pyramidal"
O.build()


def edit(cfg):
    cfg.episode_length = 25


torch, model, b, envs, keep = T._mk(O, task, n, edit, model_edit=dict(opt_cone=np.array([1], np.int32)) if cone == "elliptic" else None)
b.reset(seed=seed)
for i, e in enumerate(envs):
    e.reset(seed, i)
rng = np.random.default_rng(seed + 100)
W = T._new_W()
nobs, npriv, nu = b.nobs, b.npriv, model.nu      # (the duck: 101 / 212 / 14)
W["reset_ill"] = T._ill_resets(envs, model, nobs)
for t in range(steps):
    T._resync(b, envs, model)
    act = rng.uniform(-1, 1, (n, nu)).astype(np.float32)
    T._step_and_compare(torch, b, envs, act, nobs, npriv, t, W)
print(task, cone, {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in T._errs(W).items()}, "done", W["n_done"], "trunc", W["n_trunc"])
b.close()
