"""Replays ONE state of a `gpu_fuzz_parity.py` sweep (GPU box): the contact distances and the step's qvel on the HIP path -- for every library
listed in ODK_REPLAY_LIBS (names in open_duck_playground_amd/csrc/, default libodk.so), each in a child process -- against the float64 oracle, the
float32 build of the oracle, and the oracle under rounding-level noise on the state (how many of 64 trials change the foot-foot contact set).
    python tools/gpu_fuzz_replay.py <task> <n_states> <seed> <env>
The states are regenerated exactly as the sweep made them (same n, same seed)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]

CHILD = """
import sys, json, numpy as np, torch
sys.path[:0] = [%r]
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
z = np.load(%r)
model = load_task_model(%r)
cfg = engine.default_config(); cfg.lanes_per_env = 32
n = 64
b = engine.Batch(model, n, cfg)
b.set_state(np.repeat(z['qpos'][None], n, 0), np.repeat(z['qvel'][None], n, 0), np.repeat(z['warm'][None], n, 0))
b.physics_step(torch.tensor(np.repeat(z['ctrl'][None], n, 0), dtype=torch.float32, device='cuda'), 1)
gq, gv, _ = b.get_state()
img = b.lds_image(); o = b.lds_offset('contact_dist')
print(json.dumps(dict(cd=[float(x) for x in img[5][o:o + 12]], qvel=[float(x) for x in gv[5]], same=bool((img[5][o:o + 12] == img[37][o:o + 12]).all()))))
"""


def main():
    task, n, seed, env = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import oracle as O
    from gpu_fuzz_parity import make_states
    from test_gpu_parity import _oracle_step, _rel
    model, om, om32, qpos, qvel, warm, ctrl = make_states(task, n, seed)
    path = os.path.join(ROOT, "gpurun_out", f"fuzz_state_{task}_{seed}_{env}.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, qpos=qpos[env], qvel=qvel[env], warm=warm[env], ctrl=ctrl[env])

    def fwd(m, qp, qv):
        d = O.OracleData(m)
        d["qpos"][: m.nq] = qp; d["qvel"][: m.nv] = qv; d["qacc_warmstart"][: m.nv] = warm[env]; d["ctrl"][:14] = ctrl[env]
        d.forward()
        return np.array(d["contact_dist"][:12], np.float64)

    cd64, cd32 = fwd(om, qpos[env], qvel[env]), fwd(om32, qpos[env], qvel[env])
    print("oracle f64", np.round(cd64, 6).tolist())
    print("oracle f32", np.round(cd32, 6).tolist())
    ds = _oracle_step(O, om, qpos[env], qvel[env], warm[env], ctrl[env], 1)
    rng = np.random.default_rng(0)
    for amp in (1e-7, 1e-6, 1e-5):
        sets = {}
        for _ in range(64):
            c = fwd(om, qpos[env] + np.concatenate([np.zeros(7), rng.uniform(-amp, amp, om.nq - 7)]), qvel[env])
            key = tuple(np.round(np.sort(c[8:12]), 4))
            sets[key] = sets.get(key, 0) + 1
        print(f"oracle f64, joint noise {amp:g}: foot-foot contact sets (4 decimals) -> trials:", {str(k): v for k, v in sets.items()})
    for lib in os.environ.get("ODK_REPLAY_LIBS", "libodk.so").split():
        env_ = dict(os.environ, ODK_LIB=os.path.join(ROOT, "open_duck_playground_amd", "csrc", lib))
        o = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path, task)], capture_output=True, text=True, env=env_)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", o.stderr[-500:]); continue
        r = json.loads(line[-1])
        verr = _rel(np.array(r["qvel"]), np.array(ds["qvel"][: om.nv]), 1.0).max()
        print(f"{lib:16s}", np.round(r["cd"], 6).tolist(), f"qvel err vs oracle {verr:.2e}", "(all envs of the batch agree)" if r["same"] else "(ENVS OF ONE BATCH DIFFER)")


if __name__ == "__main__":
    main()
