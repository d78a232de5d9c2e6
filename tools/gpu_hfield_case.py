"""Diagnostic (GPU box): one env step of the resynchronised test sequence where GPU and oracle part in FREE-RUNNING substeps although
every resynchronised substep agrees.  Finds the first substep k whose contact distances differ, takes the GPU's OWN state before that
substep and evaluates the oracle (float64 and float32 builds) AT that state: if the oracle then reports the GPU's contacts, the
difference is the 1e-7 state difference sitting on a selection tie; if not, the kernel's collision routine disagrees with the oracle
on identical input.   python tools/gpu_hfield_case.py task t env"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import oracle as O
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model

task, t_target, i_target = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
standing = len(sys.argv) > 4 and sys.argv[4] == "standing"          # the sequence of test_standing_env_matches_oracle instead
dr = len(sys.argv) > 4 and sys.argv[4] == "dr"                      # the sequence of test_env_step_with_domain_randomisation (per-env models)
seed_reset, seed_act = (11, 2) if standing else ((21, 6) if dr else (9, 0))
O.build()
model = load_task_model(task)
om = O.OracleModel(model.blob()); prm = O.OraclePRM(engine.load_prm())
om32 = O.OracleModel(model.blob(), f32=True)
if dr:
    import test_gpu_env as T
    from open_duck_playground_amd import randomize
    fields, _ = randomize.domain_randomize(model, np.random.default_rng(17), 32)
    def edit(cfg): cfg.episode_length = 20
    _t, _m, b_dr, envs, keep = T._mk(O, task, 32, edit, dr_fields=fields)
    om = keep[2][i_target]
    om32 = T._dr_model(model, om32, fields, i_target)
else:
    envs = [O.OracleEnv(om, prm, standing=standing) for _ in range(32)]
for i, e in enumerate(envs):
    if not dr:
        e.cfg["episode_length"][0] = 25
    e.reset(seed_reset, i)
rng = np.random.default_rng(seed_act)
for t in range(t_target + 1):
    act = rng.uniform(-1, 1, (32, 14)).astype(np.float32)
    if t == t_target:
        break
    for i, e in enumerate(envs):
        e.step(act[i])
e = envs[i_target]
c0 = e.clone(); c0.cfg["n_substeps"][0] = 0; c0.step(act[i_target])
mt = np.array(c0["motor_targets"][:14])
nq, nv = om.nq, om.nv
q0 = np.array(c0.data["qpos"][:nq]); v0 = np.array(c0.data["qvel"][:nv]); w0 = np.array(c0.data["qacc_warmstart"][:nv])
b = b_dr if dr else engine.Batch(model, 1)
NB = 32 if dr else 1
IT = i_target if dr else 0
o_cd = b.lds_offset("contact_dist"); o_cr = b.lds_offset("contact_r")
ctrl = torch.tensor(np.tile(mt[None], (NB, 1)), dtype=torch.float32, device="cuda")


def oracle_at(omx, q, v, w):
    d = O.OracleData(omx)
    d["decision_margin"][:] = 1e30
    d["qpos"][:nq] = q; d["qvel"][:nv] = v; d["qacc_warmstart"][:nv] = w; d["ctrl"][:14] = mt
    d.forward()
    return d


gq, gv, gw = q0[None].astype(np.float32), v0[None].astype(np.float32), w0[None].astype(np.float32)
d_free = O.OracleData(om)
d_free["qpos"][:nq] = q0; d_free["qvel"][:nv] = v0; d_free["qacc_warmstart"][:nv] = w0
for k in range(1, 11):
    # the GPU's own state before substep k
    b.set_state(np.tile(q0[None], (NB, 1)), np.tile(v0[None], (NB, 1)), np.tile(w0[None], (NB, 1)))
    if k > 1:
        b.physics_step(ctrl, k - 1)
    sq, sv, sw = b.get_state()
    sq, sv, sw = sq[IT:IT + 1], sv[IT:IT + 1], sw[IT:IT + 1]
    b.physics_step(ctrl, 1)
    img = b.lds_image()[IT]
    cd_g = img[o_cd: o_cd + 12].astype(np.float64)
    d64 = oracle_at(om, sq[0].astype(np.float64), sv[0].astype(np.float64), sw[0].astype(np.float64))
    d32 = oracle_at(om32, sq[0], sv[0], sw[0])
    cd_64 = np.array(d64["contact_dist"][:12]); cd_32 = np.array(d32["contact_dist"][:12], np.float64)
    q_free = np.array(d_free["qpos"][:nq]); v_free = np.array(d_free["qvel"][:nv])
    act_ = (cd_64[:8] < 0) | (cd_g[:8] < 0)
    err = np.abs(np.where(act_, cd_g[:8] - cd_64[:8], 0)).max()
    print(f"k={k}: GPU state vs free-running oracle: qpos {np.abs(sq[0] - q_free).max():.2e} qvel {np.abs(sv[0] - v_free).max():.2e} | "
          f"contacts AT the GPU's state: |gpu - oracle64| {err:.2e}  |oracle32 - oracle64| {np.abs(np.where(act_, cd_32[:8] - cd_64[:8], 0)).max():.2e}")
    if err > 2e-6:
        # which class of near-ties, biased to the runner-up (odko_set_tie_bias), makes the float64 oracle report the kernel's contacts
        expl = []
        for eps in (3e-7, 2e-6, 1e-5):
            for mask in (1, 2, 4, 8, 16, 32, 64, 128, 12, 9, 72, 24, 5, 68, 127):
                O.set_tie_bias(mask, eps, 1e-5)
                db = oracle_at(om, sq[0].astype(np.float64), sv[0].astype(np.float64), sw[0].astype(np.float64))
                O.set_tie_bias(0)
                cdb = np.array(db["contact_dist"][:12])
                eb = np.abs(np.where(act_, cd_g[:8] - cdb[:8], 0)).max()
                if eb < 2e-6:
                    expl.append((mask, eps))
        print("   oracle64 solver: warm_used", d64.i("warm_used"), "ls_iters", d64.i("ls_iters"), "alpha", float(d64["ls_alpha"][0]))
        print("   tie bias (mask, eps) that reproduces the kernel's contacts at this state:", expl[:6], "margins", np.array(d64["decision_margin"][:5]))
        print("   gpu      ", np.round(cd_g[:8], 7).tolist())
        print("   oracle64 ", np.round(cd_64[:8], 7).tolist())
        print("   oracle32 ", np.round(cd_32[:8], 7).tolist())
        cp64 = np.array(d64["contact_pos"][:36]).reshape(12, 3); fr64 = np.array(d64["contact_frame"][:108]).reshape(12, 9)
        print("   oracle64 pos / normal of contacts 4..7:")
        for c in range(4, 8):
            print("      ", np.round(cp64[c], 5).tolist(), np.round(fr64[c][:3], 4).tolist())
        np.savez(os.path.join(ROOT, "gpurun_out", f"hfield_case_{task}_{t_target}_{i_target}_k{k}.npz"), qpos=sq[0], qvel=sv[0], warm=sw[0], ctrl=mt, gpu_dist=cd_g, o64_dist=cd_64)
        print("   state saved to gpurun_out/")
    d_free["ctrl"][:14] = mt; d_free.step()
