"""Generates tests/golden/*.npz by IMPORTING the reference's numpy mirrors in the build
container (they cannot travel to the GPU box).  Data only: inputs + expected outputs.

    python tools/make_golden.py [/root/reference]

Sources (reference, imported not copied):
  playground/common/rewards_numpy.py                         -> rewards.npz
  playground/open_duck_mini_v2/custom_rewards_numpy.py       -> rewards.npz (imitation)
  playground/common/poly_reference_motion_numpy.py           -> reference_motion.npz
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

from playground.common import rewards_numpy as R  # noqa: E402
from playground.common.poly_reference_motion_numpy import PolyReferenceMotion  # noqa: E402
from playground.open_duck_mini_v2.custom_rewards_numpy import reward_imitation  # noqa: E402

rng = np.random.default_rng(20251001)
N = 48

# ---------------------------------------------------------------- reference motion
with contextlib.redirect_stdout(io.StringIO()):
    prm = PolyReferenceMotion(os.path.join(REF, "playground/open_duck_mini_v2/data/polynomial_coefficients.pkl"))
queries = []
for _ in range(64):  # random, incl. out-of-range values that get clipped
    queries.append((rng.uniform(-0.3, 0.35), rng.uniform(-0.25, 0.25), rng.uniform(-1.6, 1.6), int(rng.integers(0, 80))))
for dy in (0.0, 0.037, -0.037, 0.074, -0.074):  # grid ties on dy (first index wins)
    queries.append((0.1, dy, 0.3, 5))
for dx in (-0.111, -0.037, 0.037, 0.111, 0.185):  # ties on dx
    queries.append((dx, 0.05, -0.2, 13))
queries.append((0.1, 0.0, 0.3, 5))  # SURVEY section 4 known answer
queries.append((0.0, 0.0, 0.0, 0))
queries.append((1.0, 1.0, 5.0, 26))
queries.append((-1.0, -1.0, -5.0, 27))
q = np.array(queries, dtype=np.float64)
exp = np.array([np.asarray(prm.get_reference_motion(a, b, c, int(i)), dtype=np.float64) for a, b, c, i in queries])
idx = np.array([[int(v) for v in prm.vel_to_index(a, b, c)] for a, b, c, _ in queries], dtype=np.int32)
np.savez(os.path.join(OUT, "reference_motion.npz"), query=q, expected=exp, index=idx,
         nb_steps_in_period=np.array([prm.nb_steps_in_period]))
print("reference_motion", q.shape, exp.shape, exp[-5, :5], idx[-5])

# ---------------------------------------------------------------- rewards
cmd = rng.uniform(-1, 1, size=(N, 7)) * np.array([0.15, 0.2, 1.0, 1.0, 0.78, 1.5, 0.5])
cmd[:6] = 0.0                      # zero-command branch (10% of training steps)
cmd[6:10, :3] *= 0.01              # |cmd[:3]| around the 0.01 gate
local_vel = rng.normal(0, 0.2, size=(N, 3))
local_vel[10:14, 1] = cmd[10:14, 1] + rng.uniform(-0.12, 0.12, size=4)  # y dead-band edge
gyro = rng.normal(0, 0.8, size=(N, 3))
torques = rng.uniform(-3.23, 3.23, size=(N, 14))
act = rng.uniform(-1, 1, size=(N, 14)); last_act = rng.uniform(-1, 1, size=(N, 14))
jq = rng.uniform(-1.0, 1.4, size=(N, 14)); jv = rng.normal(0, 2.0, size=(N, 14))
default_pose = np.array([0.002, 0.053, -0.63, 1.368, -0.784, 0, 0, 0, 0, -0.003, -0.065, 0.635, 1.379, -0.796])
base_qpos = np.concatenate([rng.normal(0, 0.1, size=(N, 3)), rng.normal(0, 1, size=(N, 4))], axis=1)
base_qpos[:, 3:] /= np.linalg.norm(base_qpos[:, 3:], axis=1, keepdims=True)
base_qvel = rng.normal(0, 0.5, size=(N, 6))
contacts = rng.integers(0, 2, size=(N, 2)).astype(bool)
ref = rng.normal(0, 0.6, size=(N, 40)); ref[:, 32:34] = rng.uniform(0, 1, size=(N, 2))
sigma = 0.01
# one NaN case to pin nan_to_num
torques[20, 3] = np.nan
out = dict(cmd=cmd, local_vel=local_vel, gyro=gyro, torques=torques, act=act, last_act=last_act, jq=jq, jv=jv,
           default_pose=default_pose, base_qpos=base_qpos, base_qvel=base_qvel, contacts=contacts.astype(np.float64), ref=ref,
           sigma=np.array([sigma]))
with np.errstate(all="ignore"):
    out["tracking_lin_vel"] = np.array([R.reward_tracking_lin_vel(cmd[i], local_vel[i], sigma) for i in range(N)], dtype=np.float64)
    out["tracking_ang_vel"] = np.array([R.reward_tracking_ang_vel(cmd[i], gyro[i], sigma) for i in range(N)], dtype=np.float64)
    out["torques_cost"] = np.array([R.cost_torques(torques[i]) for i in range(N)], dtype=np.float64)
    out["action_rate"] = np.array([R.cost_action_rate(act[i], last_act[i]) for i in range(N)], dtype=np.float64)
    out["stand_still"] = np.array([R.cost_stand_still(cmd[i], jq[i], jv[i], default_pose, ignore_head=False) for i in range(N)], dtype=np.float64)
    out["alive"] = np.array([R.reward_alive()], dtype=np.float64)
    out["imitation"] = np.array([reward_imitation(base_qpos[i], base_qvel[i], jq[i], jv[i], contacts[i], ref[i], cmd[i], True)
                                 for i in range(N)], dtype=np.float64)
# Standing terms (reference standing.py:585-606): drawn AFTER everything above so the earlier arrays keep their values
upvec = rng.normal(0, 0.5, size=(N, 3)); upvec[:, 2] = np.abs(upvec[:, 2])
upvec[30] = np.nan   # nan_to_num branch
cmd_head = cmd.copy(); cmd_head[24:, :3] = 0.0   # Standing's own command: no move part -> head_pos gate closed
out["upvector"] = upvec; out["cmd_head"] = cmd_head
with np.errstate(all="ignore"):
    out["orientation"] = np.array([R.cost_orientation(upvec[i]) for i in range(N)], dtype=np.float64)
    out["head_pos"] = np.array([R.cost_head_pos(jq[i], jv[i], cmd_head[i]) for i in range(N)], dtype=np.float64)
    out["stand_still_legs"] = np.array([R.cost_stand_still(cmd_head[i], jq[i], jv[i], default_pose, ignore_head=True) for i in range(N)], dtype=np.float64)
np.savez(os.path.join(OUT, "rewards.npz"), **out)
print("rewards", {k: v.shape for k, v in out.items() if k in ("imitation", "stand_still", "tracking_lin_vel")})
print(out["imitation"][:12], out["stand_still"][:8])
