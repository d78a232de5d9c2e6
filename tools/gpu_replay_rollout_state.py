"""One outlier of tests/test_gpu_parity.py::test_env_step_ten_substeps_on_rollout_states (gpurun_out/rollout_state_outliers_<task>.npz) substep by
substep: kernel (one substep at a time from ITS OWN states, and from the oracle's) against the float64 oracle -- where do the contacts / qacc part?
    python tools/gpu_replay_rollout_state.py <task> <state index>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import oracle as O
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model

task, idx = sys.argv[1], int(sys.argv[2])
p = os.path.join(ROOT, "gpurun_out", f"rollout_state_outliers_{task}.npz")
z = np.load(p if os.path.exists(p) else os.path.join(ROOT, "profiles", "r6", f"rollout_state_outliers_{task}.npz"))
k = list(z["idx"]).index(idx)
q, v, w, c = z["qpos"][k], z["qvel"][k], z["warm"][k], z["ctrl"][k]
model = load_task_model(task)
om = O.OracleModel(model.blob())
b = engine.Batch(model, 2)
ct = torch.tensor(np.stack([c, c]), dtype=torch.float32, device="cuda")
rel = lambda a, bb, f: float((np.abs(a - bb) / np.maximum(np.abs(bb), f)).max())
qo, vo, wo = q.astype(np.float64), v.astype(np.float64), w.astype(np.float64)
qk, vk, wk = q.copy(), v.copy(), w.copy()
np.set_printoptions(precision=4, suppress=True, linewidth=220)
for s in range(10):
    # env 0: the kernel from its own state; env 1: the kernel from the oracle's state of this substep
    b.set_state(np.stack([qk, qo]), np.stack([vk, vo]), np.stack([wk, wo]))
    b.physics_step(ct, 1)
    gq, gv, gw = b.get_state()
    dbg = b.get_debug()
    d = O.OracleData(om); d["qpos"][: om.nq] = qo; d["qvel"][: om.nv] = vo; d["qacc_warmstart"][: om.nv] = wo
    d.env_physics_step(c, 1)
    q1, v1, w1 = np.array(d["qpos"][: om.nq]), np.array(d["qvel"][: om.nv]), np.array(d["qacc_warmstart"][: om.nv])
    cd_o = np.array(d["contact_dist"][:12]); cd_k = dbg["contact_dist"][1]
    print(f"substep {s}: kernel(from oracle state) vs oracle: qvel {rel(gv[1], v1, 1.0):.2e} qacc {rel(dbg['qacc'][1], np.array(d['qacc'][: om.nv]), 5.0):.2e} | kernel own path vs oracle: qvel {rel(gv[0], v1, 1.0):.2e}")
    print("   contact dist mm oracle", cd_o[:8] * 1e3)
    print("   contact dist mm kernel", cd_k[:8] * 1e3)
    fr = np.array(d["contact_frame"][:72]).reshape(8, 9)[:, :3]
    print("   oracle normals z", fr[:, 2], " pos z", np.array(d["contact_pos"][:24]).reshape(8, 3)[:, 2])
    qo, vo, wo = q1, v1, w1
    qk, vk, wk = gq[0], gv[0], gw[0]
b.close()
