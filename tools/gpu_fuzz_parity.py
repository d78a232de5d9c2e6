"""Differential sweep (GPU box): many random contact-rich states per model, one mjx.step on the HIP path against the float64 oracle.
Mismatches beyond the parity bounds are classified on the oracle side (contact-set tie under rounding-level noise; float32 build of
the oracle on the kernel's side) and whatever stays unexplained is printed with what is needed to replay it.
    python tools/gpu_fuzz_parity.py [n_states = 1024] [seed = 0] [task ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402

import oracle as O  # noqa: E402
from open_duck_playground_amd import engine  # noqa: E402
from open_duck_playground_amd.model import load_task_model  # noqa: E402
from test_gpu_parity import _contact_tie, _contacts, _oracle_step, _random_states, _rel  # noqa: E402

def make_states(task, n=1024, seed=0, variant=None):
    """CPU only: the model (variant: None | "elliptic" | "box" | "sphere-capsule" ...: the foot colliders replaced as in tests/test_gpu_parity.py),
    its oracle models and the random contact-rich states of the sweep"""
    if variant is None:
        model = load_task_model(task)
    elif variant == "elliptic":      # the task's own model with <option cone="elliptic"> (the cone instantiations of the kernels)
        from open_duck_playground_amd.model import Model
        base = load_task_model(task)
        model = Model({**base.a, "opt_cone": np.array([1], np.int32)})
    elif variant == "box":
        from test_gpu_parity import _box_feet_variant
        model = _box_feet_variant(task)
    else:
        from test_gpu_parity import _prim_feet_variant
        model = _prim_feet_variant(task, tuple(variant.split("-")))
    om = O.OracleModel(model.blob()); om32 = O.OracleModel(model.blob(), f32=True)
    rng = np.random.default_rng(seed)
    qpos, qvel = _random_states(model, n, rng, airborne_frac=0.1)
    from open_duck_playground_amd.tables import build_kernel_tables
    aq = build_kernel_tables(model.a)["k_act_qposadr"]
    for e in range(n):   # press the feet 0.2 ... 6 mm into the floor, a third of the robots leaning, some far from the origin
        if qpos[e, 2] > 0.25:
            continue
        if e % 3 == 0:
            lean = rng.uniform(-0.4, 0.4); q = qpos[e, 3:7].copy()
            qpos[e, 3:7] = [np.cos(lean / 2) * q[0] - np.sin(lean / 2) * q[1], np.cos(lean / 2) * q[1] + np.sin(lean / 2) * q[0],
                            np.cos(lean / 2) * q[2] + np.sin(lean / 2) * q[3], np.cos(lean / 2) * q[3] - np.sin(lean / 2) * q[2]]
        if e % 4 == 1:      # near the home pose: both feet down
            qpos[e, 7:] = np.asarray(model.a["key_qpos"])[7:] + rng.uniform(-0.03, 0.03, model.nq - 7) * (np.asarray(model.a["key_qpos"])[7:] != 0)
        if "rough" in task:
            qpos[e, 0:2] = rng.uniform(-8.0, 8.0, 2)
        if e % 16 in (7, 15):     # feet pressed against each other; off the floor (7) or standing on it (15)
            xy = qpos[e, 0:2].copy()
            qpos[e] = np.asarray(model.a["key_qpos"]); qpos[e, 2] = 0.5 if "rough" in task else 0.3
            qpos[e, int(aq[1])] = rng.uniform(0.35, 0.6); qpos[e, int(aq[10])] = rng.uniform(-0.6, -0.35)
            qpos[e, int(aq[0])] += rng.uniform(-0.4, 0.4); qpos[e, int(aq[9])] += rng.uniform(-0.4, 0.4); qpos[e, int(aq[4])] += rng.uniform(-0.3, 0.3)
            if e % 16 == 7:
                continue
            qpos[e, 0:2] = xy
        qpos[e, 2] = 0.4
        target = rng.uniform(2e-4, 6e-3)
        for _ in range(14):
            d = O.OracleData(om); d["qpos"][: om.nq] = qpos[e]; d.forward()
            cd = np.array(d["contact_dist"][:8])
            qpos[e, 2] -= cd.min() + target if (cd < 0).any() else max(0.8 * min(cd.min(), 0.2), 2e-3)
    ctrl = np.asarray(model.a["key_ctrl"])[None] + rng.uniform(-0.4, 0.4, (n, 14))
    warm = rng.normal(0, 3.0, (n, model.nv))
    return model, om, om32, qpos, qvel, warm, ctrl


def sweep(task, n=1024, seed=0, lanes=32, nsub=1, dist_tol=1e-6, verbose=True, variant=None, dr=False):
    """one model: returns (counts, worst errors of the states that agree); dr: every env with its own randomize.py model fields"""
    model, om_base, om32_base, qpos, qvel, warm, ctrl = make_states(task, n, seed, variant)
    fields = None
    if dr:
        from open_duck_playground_amd import randomize
        from test_gpu_env import _dr_model
        fields, _ = randomize.domain_randomize(model, np.random.default_rng(seed + 7), n)
    cfg = engine.default_config(); cfg.lanes_per_env = lanes if "rough" not in task else 32
    b = engine.Batch(model, n, cfg)
    if dr:
        randomize.apply(b, fields)
    b.set_state(qpos, qvel, warm)
    b.physics_step(torch.tensor(ctrl, dtype=torch.float32, device="cuda"), nsub)
    gq, gv, _ = b.get_state()
    img = b.lds_image(); o_cd = b.lds_offset("contact_dist")
    b.close()
    prng = np.random.default_rng(seed + 1)
    stat = dict(ok=0, tie=0, f32_side=0, solver_branch=0, unexplained=0, in_contact=0, both_feet=0, foot_foot=0)
    worst = dict(dist=0.0, qvel=0.0)
    for e in range(n):
        om = _dr_model(model, om_base, fields, e) if dr else om_base
        om32 = _dr_model(model, om32_base, fields, e) if dr else om32_base
        d = O.OracleData(om)
        d["qpos"][: om.nq] = qpos[e]; d["qvel"][: om.nv] = qvel[e]; d["qacc_warmstart"][: om.nv] = warm[e]; d["ctrl"][:14] = ctrl[e]
        d.forward()
        cd_o = np.array(d["contact_dist"][:12]); cd_g = img[e][o_cd: o_cd + 12]
        stat["in_contact"] += int((cd_o[:8] < 0).any()); stat["both_feet"] += int((cd_o[:4] < 0).any() and (cd_o[4:8] < 0).any()); stat["foot_foot"] += int((cd_o[8:] < 0).any())
        act = (cd_o < 0) | (cd_g < 0)
        # per geom pair as multisets: which slot a manifold point lands in may differ where two candidates tie in area (a rectangle's
        # two far corners), the set of contacts -- the physics -- is the same
        so, sg = np.concatenate([np.sort(cd_o[4 * p: 4 * p + 4]) for p in range(3)]), np.concatenate([np.sort(cd_g[4 * p: 4 * p + 4]) for p in range(3)])
        acts = (so < 0) | (sg < 0)
        derr = np.abs(sg[acts] - so[acts]).max() if acts.any() else 0.0
        ds = _oracle_step(O, om, qpos[e], qvel[e], warm[e], ctrl[e], nsub)
        verr = _rel(gv[e], np.array(ds["qvel"][: om.nv]), 1.0).max()
        if nsub > 1:
            derr = 0.0   # the image holds the contacts of the LAST forward pass
        if derr < dist_tol and verr < (1e-4 if nsub == 1 else 1e-3):
            stat["ok"] += 1; worst["dist"] = max(worst["dist"], derr); worst["qvel"] = max(worst["qvel"], verr)
            continue
        if _contact_tie(O, om, qpos[e], qvel[e], ctrl[e], prng, _contacts(d), k=16):
            stat["tie"] += 1
            continue
        d32 = O.OracleData(om32)
        d32["qpos"][: om.nq] = qpos[e]; d32["qvel"][: om.nv] = qvel[e]; d32["qacc_warmstart"][: om.nv] = warm[e]; d32["ctrl"][:14] = ctrl[e]
        d32.forward()
        cd_32 = np.array(d32["contact_dist"][:12], np.float64)
        s32 = np.concatenate([np.sort(cd_32[4 * p: 4 * p + 4]) for p in range(3)])
        if derr >= dist_tol and (not acts.any() or np.abs(sg[acts] - s32[acts]).max() < 2e-6):
            stat["f32_side"] += 1
            continue
        if derr < dist_tol:   # same contacts, different solve: the oracle's own step under perturbation
            sens = 0.0
            for _ in range(16):
                dp = _oracle_step(O, om, qpos[e] + np.concatenate([np.zeros(7), prng.uniform(-1e-6, 1e-6, om.nq - 7)]), qvel[e] + prng.uniform(-5e-6, 5e-6, om.nv), warm[e], ctrl[e], nsub)
                sens = max(sens, _rel(np.array(dp["qvel"][: om.nv]), np.array(ds["qvel"][: om.nv]), 1.0).max())
            d32s = _oracle_step(O, om32, qpos[e], qvel[e], warm[e], ctrl[e], nsub)
            sens = max(sens, _rel(np.array(d32s["qvel"][: om.nv], np.float64), np.array(ds["qvel"][: om.nv]), 1.0).max())
            if sens > 3e-5:
                stat["solver_branch"] += 1
                continue
        # the referee of the parity tests (oracle "Tie bias"): one class of the oracle's discrete decisions taking its runner-up inside a rounding-level
        # band reproduces what the kernel did (round 6: a foot-foot pair whose separating face is one of the sole's two coplanar triangles)
        from test_gpu_parity import ILL_CLASSES
        hit = None
        for eps, eps_rel in ((1e-7, 1e-6), (2e-6, 1e-4)):
            for bit, name in ILL_CLASSES:
                O.set_tie_bias(bit, eps, eps_rel)
                try:
                    db = O.OracleData(om)
                    db["qpos"][: om.nq] = qpos[e]; db["qvel"][: om.nv] = qvel[e]; db["qacc_warmstart"][: om.nv] = warm[e]; db["ctrl"][:14] = ctrl[e]
                    db.forward()
                    cb = np.array(db["contact_dist"][:12])
                    sb = np.concatenate([np.sort(cb[4 * p: 4 * p + 4]) for p in range(3)])
                    ab = (sb < 0) | (sg < 0)
                    dsb = _oracle_step(O, om, qpos[e], qvel[e], warm[e], ctrl[e], nsub)
                finally:
                    O.set_tie_bias(0)
                if (nsub > 1 or not ab.any() or np.abs(sg[ab] - sb[ab]).max() < 2e-6) and _rel(gv[e], np.array(dsb["qvel"][: om.nv]), 1.0).max() < (1e-4 if nsub == 1 else 1e-3):
                    hit = f"{name}@{eps:g}"
                    break
            if hit:
                break
        if hit:
            stat["tie_class"] = stat.get("tie_class", 0) + 1
            stat.setdefault("tie_classes", {})[hit] = stat.get("tie_classes", {}).get(hit, 0) + 1
            continue
        stat["unexplained"] += 1
        if verbose:
            print(f"  UNEXPLAINED {task} seed {seed} env {e}: dist err {derr:.2e} qvel err {verr:.2e}")
            print("     oracle", np.round(cd_o, 6).tolist()); print("     gpu   ", np.round(cd_g.astype(float), 6).tolist()); print("     f32   ", np.round(cd_32, 6).tolist())
    return stat, worst


if __name__ == "__main__":
    n_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seed_ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    O.build()
    for task_ in sys.argv[3:] or ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"]:
        st, w = sweep(task_, n_, seed_, int(os.environ.get("ODK_FUZZ_LANES", "32")), int(os.environ.get("ODK_FUZZ_SUBSTEPS", "1")), variant=os.environ.get("ODK_FUZZ_VARIANT") or None, dr=bool(int(os.environ.get("ODK_FUZZ_DR", "0"))))
        print(task_, f"n={n_}", st, {k: float(f"{v:.2e}") for k, v in w.items()}, flush=True)
