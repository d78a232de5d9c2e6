"""How far round 2's height-field approximation (the plane of ONE triangle under the foot, oracle hfield_mode = 1) is from the
prism algorithm (hfield_mode = 0: the reference's hfield_convex as restated in oracle/odk_oracle_convex.inc), on the real terrain
of scene_rough_terrain_backlash.xml.  CPU only (oracle): a 1000-step random-action rollout under the prism algorithm; every state
it visits is evaluated by one forward pass under both modes.  Writes profiles/r3/hfield_mode_deviation.json.

    python tools/hfield_mode_deviation.py [envs] [steps]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O
from open_duck_playground_amd.model import load_task_model, asset_path

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
O.build()
model = load_task_model("rough_terrain_backlash")
z = np.load(asset_path("prm_table.npz")); prm = O.OraclePRM({k: z[k] for k in z.files})
om0 = O.OracleModel(model.blob()); om1 = om0.copy(); om1.set_int("hfield_mode", 1)
envs = [O.OracleEnv(om0, prm) for _ in range(nenv)]
for i, e in enumerate(envs):
    e.reset(3, i)
rng = np.random.default_rng(0)
nv, nq = om0.nv, om0.nq
d0, d1 = O.OracleData(om0), O.OracleData(om1)
flag_diff = n_states = n_contact_states = 0
qacc_rel, dist_diff, nact, shallow = [], [], [], []
t0 = time.time()
for t in range(steps):
    act = rng.uniform(-1, 1, (nenv, 14))
    for i, e in enumerate(envs):
        e.step(act[i])
        res = []
        for d in (d0, d1):
            d["qpos"][:nq] = e.data["qpos"][:nq]; d["qvel"][:nv] = e.data["qvel"][:nv]; d["qacc_warmstart"][:nv] = e.data["qacc_warmstart"][:nv]
            d["ctrl"][:14] = e["motor_targets"][:14]
            d.forward()
            cd = np.array(d["contact_dist"][:8])
            res.append((cd, np.array(d["qacc"][:nv])))
        (cd0, qa0), (cd1, qa1) = res
        f0 = [cd0[:4].min() < 0, cd0[4:].min() < 0]; f1 = [cd1[:4].min() < 0, cd1[4:].min() < 0]
        n_states += 1
        flag_diff += int(f0 != f1)
        if any(f0) or any(f1):
            n_contact_states += 1
            qacc_rel.append(float(np.abs(qa0 - qa1).max() / max(np.abs(qa0).max(), 5.0)))
            dist_diff.append(float(abs(min(cd0.min(), 0) - min(cd1.min(), 0))))
            nact.append((int((cd0 < 0).sum()), int((cd1 < 0).sum())))
            # the regime a standing / walking robot lives in: upright, feet pressed in by less than 3 mm under either rule
            shallow.append(bool(e.data["sensordata"][11] > 0.9 and min(cd0.min(), cd1.min()) > -0.003))
qr = np.array(qacc_rel); dd = np.array(dist_diff); na = np.array(nact); sh = np.array(shallow)
stats = lambda x: dict(median=float(np.median(x)), p90=float(np.quantile(x, 0.9)), p99=float(np.quantile(x, 0.99)), max=float(x.max())) if len(x) else None
out = dict(task="rough_terrain_backlash", envs=nenv, steps=steps, states=n_states, states_with_contact=n_contact_states,
           foot_contact_flag_differs_fraction=flag_diff / n_states,
           qacc_rel_diff=dict(median=float(np.median(qr)), p90=float(np.quantile(qr, 0.9)), p99=float(np.quantile(qr, 0.99)), max=float(qr.max())),
           deepest_dist_abs_diff_m=dict(median=float(np.median(dd)), p90=float(np.quantile(dd, 0.9)), p99=float(np.quantile(dd, 0.99)), max=float(dd.max())),
           mean_active_contacts=dict(prisms=float(na[:, 0].mean()), one_triangle=float(na[:, 1].mean())),
           upright_shallow_states=dict(count=int(sh.sum()), qacc_rel_diff=stats(qr[sh]), deepest_dist_abs_diff_m=stats(dd[sh]),
                                       note="upvector.z > 0.9 and no contact deeper than 3 mm under either rule"),
           note="qacc_rel_diff = max_dof |qacc_prisms - qacc_one_triangle| / max(max_dof |qacc_prisms|, 5) on the SAME state; float64 oracle both ways",
           seconds=round(time.time() - t0, 1))
os.makedirs(os.path.join(ROOT, "profiles", "r3"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r3", "hfield_mode_deviation.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
