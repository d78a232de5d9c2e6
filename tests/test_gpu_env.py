"""GPU parity of the full env step (wrappers + Joystick.step + obs/reward) vs the CPU oracle env,
with observation noise, action delay and pushes ON (both sides draw from the same counter RNG).

Error measure: |gpu - oracle| / max(|oracle|, 1) per element ("relative with floor 1").  The accelerometer slots are judged
separately: they are linear in qacc, which one Newton iteration in fp32 resolves to ~1e-3 relative (test_gpu_parity.py).
Bounds = ~3x the worst case measured on MI355X; the measured values of the last run are written next to the bounds into
gpurun_out/parity_worst.json (committed copy: profiles/r3/parity_worst.json).

The physics state is re-synchronised from the oracle before every step, but ten substeps of a one-iteration Newton solver with a
five-iteration line search are not a smooth map: 2-5 % of random-action env steps sit on a branch point (a contact row switching
on, a manifold tie, a line-search bracket) where the fp64 oracle ITSELF changes its answer by 1e-2 ... 1 when its input moves by
1e-6.  Those env steps are detected on the oracle side alone (ten perturbed oracle runs per env step) and set aside -- counted,
bounded in number, not judged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# obs / priv without the accelerometer, accelerometer slots, reward, metrics (BASELINE north_star: 1e-4 on the state after one
# mjx.step; an env step is 10 of them plus sensors).  Measured on MI355X over the well-conditioned env steps (round 3):
# obs <= 7e-5, accelerometer <= 3.2e-4, reward <= 1.5e-5, metrics <= 3.2e-4.
ENV_BOUNDS = dict(obs=2.5e-4, acc=1e-3, reward=5e-5, metrics=1e-3)
# An env step is set aside as ill-conditioned when the ORACLE's own outputs move by more than this under rounding-level input
# noise (`_perturbed_oracle_steps`); measured: 2-5 % of the env steps of a random-action rollout.
SENSITIVITY = dict(obs=2.5e-4, acc=2.5e-3, reward=2.5e-4, metrics=5e-4)
# Judged env steps beyond a bound ("outliers": a line-search / manifold branch flipped between fp32 and fp64 without any of the
# perturbed oracle runs flipping it) may be at most this fraction; measured 1 in 1841 (flat), 3 in 1847 (rough terrain).
# Every such step goes to a referee (`_F32.adjudicate`), in this order: (1) the float64 oracle re-run with ONE class of its discrete
# decisions -- collision: separating face, reference polytope, edge-or-face contact, incident face, clipping-plane side, manifold
# arg-max, the cut behind the fourth-deepest height-field contact; solver: warm-start pick, line-search bracket -- biased to the
# runner-up inside a band of 1e-7, then 3e-7 (the fourth-deepest cut also 1e-6) around a tie, for the whole env step or for one substep (odko_set_tie_bias): if that
# reproduces the kernel's outputs within ENV_BOUNDS the step is EXPLAINED, causally ("tie_<class>"); (2) the oracle's float32 build,
# plain / with rounding-level noise on the state / on the hull vertices ("agrees_with_kernel", "departs_too").  What neither
# explains is `outlier_fraction` (<= 0.2 % of the judged steps on every floor); a flipped contact branch moves the accelerometer by
# O(1), so there is no magnitude cap on those few -- each one is printed and its size recorded (`outlier_over_bound`).
SET_ASIDE = dict(ill_fraction=0.08, outlier_fraction=0.002, explained_fraction=0.004)      # explained: measured <= 0.0011 (round 5: bound 0.01 -> 0.004)
# Height-field floor: every reset starts with the feet 1-3 cm inside the terrain (joystick.py:206-258 knows nothing of the
# elevation) and many prisms give candidates.  Measured: 2-5 % of the env steps of the random-action sequence are ill-conditioned
# by the oracle's own sensitivity (63-75 % until the last manifold point resolved the triangle tie by rule: oracle manifold_points
# AREA_TIE), 0.3-0.5 % are judged and beyond a bound (near-ties at the 1e-7 level, e.g. which of two hull faces is the more
# anti-parallel to a prism's side wall: tools/gpu_env_outlier_substeps.py).
# Round 4: those are now adjudicated by the float32 oracle like on the flat floor, and the allowance for UNEXPLAINED ones is the
# flat floor's.
SET_ASIDE_ROUGH = dict(ill_fraction=0.15, outlier_fraction=0.002, explained_fraction=0.012)      # explained: measured <= 0.0056 (round 5: bound 0.02 -> 0.012)


def _set_aside(task):
    return SET_ASIDE_ROUGH if "rough" in task else SET_ASIDE


def _ill_resets(envs, model, nobs):
    """envs whose reset-time forward pass sits on a contact tie: the oracle's own accelerometer at the reset state moves by more
    than half the bound under 1e-6 noise on qpos.  Their first observation (which auto-reset hands back later) is not judged on
    the accelerometer slots."""
    rng = np.random.default_rng(7)
    ill = set()
    for i, e in enumerate(envs):
        adr = int(e.ints("adr_accelerometer")[0])
        acc0 = np.array(e.data["sensordata"][adr: adr + 3])
        for _ in range(8):
            c = e.clone()
            q = c.data["qpos"][: model.nq]
            q += 1e-6 * rng.standard_normal(model.nq) * np.maximum(np.abs(q), 0.1)
            c.data.forward()
            if _rel1(np.array(c.data["sensordata"][adr: adr + 3]), acc0).max() > 0.5 * RESET_BOUNDS["acc"]:
                ill.add(i)
                break
    return ill

RESET_BOUNDS = dict(obs=1e-5, acc=1e-3, qpos=1e-6, qvel=1e-6)   # measured: 3.5e-6, 3.2e-4, 1.8e-7, 2e-9


def _xml_model(xml):
    import os
    from open_duck_playground_amd.model import Model
    return Model.from_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", xml), sim_dt=0.002)


def _mk(oracle_mod, task, n, cfg_edit=None, standing=False, dr_fields=None, model_edit=None):
    """`task`: a shipped task name, or the file name of a robot under tests/assets/ (a robot that is not the duck: no imitation reward)"""
    import torch
    from open_duck_playground_amd import engine, randomize
    from open_duck_playground_amd.model import load_task_model
    robot = task.endswith(".xml")
    model = _xml_model(task) if robot else load_task_model(task)
    if model_edit:      # the task's model with some arrays replaced (e.g. opt_cone)
        from open_duck_playground_amd.model import Model
        model = Model({**model.a, **model_edit})
    cfg = engine.default_config(standing)
    if robot:
        cfg.use_imitation = 0
    if cfg_edit:
        cfg_edit(cfg)
    b = engine.Batch(model, n, cfg)
    base = oracle_mod.OracleModel(model.blob())
    prm = oracle_mod.OraclePRM(engine.load_prm())
    oms = [base] * n
    if dr_fields is not None:     # per-env model fields of randomize.py:119-144 on both sides
        randomize.apply(b, dr_fields)
        oms = [_dr_model(model, base, dr_fields, e) for e in range(n)]
    envs = _Envs(oracle_mod.OracleEnv(oms[i], prm, standing=standing) for i in range(n))
    envs.f32 = _F32(oracle_mod, model, standing, dr_fields)
    for e in envs:
        e.cfg["episode_length"][0] = cfg.episode_length
        e.cfg["noise_level"][0] = cfg.noise_level
        e.cfg["push_enable"][0] = cfg.push_enable
        e.cfg["use_imitation"][0] = cfg.use_imitation
    return torch, model, b, envs, (base, prm, oms)


class _Envs(list):
    """the oracle envs of a test + the float32 adjudicator that belongs to them"""
    f32 = None


_CFG_FIELDS = ("ctrl_dt", "action_scale", "dof_vel_scale", "max_motor_velocity", "noise_level", "noise_gyro", "noise_accelerometer", "noise_gravity",
               "noise_joint_vel", "qpos_noise_scale", "reward_scales", "tracking_sigma", "push_enable", "push_interval_range", "push_magnitude_range",
               "cmd_range", "use_imitation", "use_motor_speed_limits", "autoreset", "episode_length", "n_substeps", "env_kind", "reset_base_qvel")
_ENV_FIELDS = ("command", "last_act", "last_last_act", "last_last_last_act", "motor_targets", "feet_air_time", "swing_peak", "push", "action_history",
               "imu_history", "current_reference_motion", "imitation_phase", "ep_metrics", "first_qpos", "first_qvel", "first_warmstart", "first_obs",
               "first_priv", "obs", "priv", "metrics", "contact", "reward", "done", "ep_steps", "truncation", "episode_done", "ep_sum_reward", "ep_length")
# decision classes of the oracle's collision routines that the referee may bias (oracle/odk_oracle.c "Tie bias")
TIE_CLASSES = ((4, "edge_or_face_contact"), (8, "incident_face"), (1, "separating_face"), (2, "reference_polytope"), (16, "clipping_plane_side"),
               (32, "manifold_argmax"), (64, "fourth_deepest_cut"), (128, "warm_start_pick"), (256, "line_search_bracket_end"),
               (512, "line_search_comparison"), (1024, "area_zero_cut"))
_ENV_INTS = ("last_contact", "key", "step", "push_step", "push_interval_steps", "imitation_i", "rng_ctr")


class _F32:
    """The oracle's float32 build (oracle/libodk_oracle_f32.so: the same C source with `real` = float) as the referee of judged env
    steps beyond a bound.  `twin` copies a float64 oracle env -- config, carried info, wrapper state, physics state -- into a float32
    one; `adjudicate` steps it with the same action and says whether float32 evaluation of the ORACLE'S OWN code explains the
    disagreement."""

    def __init__(self, O, model, standing, dr_fields):
        self.O, self.model, self.standing, self.dr = O, model, standing, dr_fields
        self.base = self.prm = None
        self.models = {}

    def _model(self, i):
        from open_duck_playground_amd import engine
        if self.base is None:
            self.base = self.O.OracleModel(self.model.blob(), f32=True)
            self.prm = self.O.OraclePRM(engine.load_prm(), f32=True)
        if self.dr is None:
            return self.base
        if i not in self.models:
            self.models[i] = _dr_model(self.model, self.base, self.dr, i)
        return self.models[i]

    def twin(self, e, i, om=None):
        om = om if om is not None else self._model(i)
        t = self.O.OracleEnv(om, self.prm, standing=self.standing)
        t._keep_model = om
        for nm in _CFG_FIELDS:
            t.cfg[nm][:] = e.cfg[nm]
        for nm in _ENV_FIELDS:
            t[nm][:] = e[nm]
        for nm in _ENV_INTS:
            t.ints(nm)[:] = e.ints(nm)
        for nm in ("qpos", "qvel", "qacc_warmstart", "time"):
            t.data[nm][:] = e.data[nm]
        return t

    TRIES = 13

    def adjudicate(self, pre, i, act, gpu, e64, nobs, npriv, rng):
        """`pre`: the float64 env before the step; `gpu`: (obs, priv, reward, metrics, done) of the kernel; `e64`: the float64 env after
        it.  Returns "tie_<class>" when the float64 oracle with that class of collision near-ties biased the other way reproduces
        the kernel's outputs within ENV_BOUNDS; "agrees_with_kernel" when a float32 run of the oracle reproduces the kernel's outputs within ENV_BOUNDS -- the
        plain run, one of six with its input state moved by one float32 rounding (1e-7 relative), or one of six with the state moved by
        1e-6 AND the hull vertices by 2e-7 relative (the kernels carry the model's constants rounded to float32 from a float64
        host computation, the oracle derives its face normals itself: near-ties BETWEEN HULL FEATURES -- which of two foot faces
        is the more anti-parallel to a prism wall, edge axis vs face axis at the EDGE_TOL threshold -- are decided by those last
        bits and blind to any perturbation of the state alone); "departs_too" when the plain float32 run leaves the float64
        result by more than a bound as well, elsewhere; None otherwise."""
        obs, priv, rew, met, done = gpu

        def same_as_kernel(t):
            o, a = _obs_err(obs, priv, t, nobs, npriv)
            r = float(_rel1(rew, t["reward"][0])); m = float(_rel1(met, np.array(t["metrics"][:8], np.float64)).max())
            return done == t["done"][0] and o <= ENV_BOUNDS["obs"] and a <= ENV_BOUNDS["acc"] and r <= ENV_BOUNDS["reward"] and m <= ENV_BOUNDS["metrics"]

        # 1. the float64 oracle with ONE class of collision near-ties biased to the runner-up for the whole env step (odko_set_tie_bias:
        #    band 3e-7 m / 3e-7 in a cosine / 1e-5 relative for the manifold's area steps): deterministic and causal -- "the kernel's
        #    answer is the oracle's own algorithm with that tie falling the other way"
        #    Singles first, then pairs of classes; band 3e-7, then 2e-6 (the terrain spans +-10 m: one float32 ulp of a coordinate there
        #    is 1e-6 m, and the kernels' window-relative coordinates only remove part of that).
        #    A tie between two hull features persists through the env step (bias on in all ten collision passes); a foot that rotates
        #    THROUGH a tie crosses it in one substep (bias on in that pass only).
        #    Round 5: the search is what the recorded cases needed and no wider.  Of the 31 explained steps on record
        #    (20 in the suite, 11 in the sweeps) none needed a PAIR of classes, and only the cut behind the fourth-deepest height-field
        #    contact needed a band beyond 3e-7 (a contact depth compared across prisms whose coordinates are metres from the origin):
        #    singles only (pairs: ODK_REFEREE_PAIRS=1, tools), a ladder of bands 1e-7 -> 3e-7 for every class, 1e-6 for that class
        #    alone (measured with the ladder in place, profiles/r5/referee_log.jsonl: 13 of 18 tie verdicts at 1e-7, the five at 1e-6 all
        #    `fourth_deepest_cut`; the 2e-6 band of round 4 is gone); the band that sufficed and the oracle's own smallest decision margins of the step go to the referee log.
        import os
        singles = [(bit, name) for bit, name in TIE_CLASSES]
        pairs = [(b1 | b2, n1 + "+" + n2) for i, (b1, n1) in enumerate(TIE_CLASSES) for (b2, n2) in TIE_CLASSES[i + 1:]] if os.environ.get("ODK_REFEREE_PAIRS") == "1" else []
        nsub = int(pre.cfg["n_substeps"][0])
        self.last = None
        for eps, eps_rel, tag, classes in ((1e-7, 1e-5, "@1e-7", None), (3e-7, 1e-5, "", None), (1e-6, 1e-4, "@1e-6", (64,))):
            for window in [None] + [(k, k) for k in range(nsub)]:
                for mask, name in (singles + pairs if window is None else singles):
                    if classes is not None and mask not in classes:
                        continue
                    t = pre.clone()
                    self.O.set_tie_bias(mask, eps, eps_rel, window=window)
                    try:
                        t.step(act)
                    finally:
                        self.O.set_tie_bias(0)
                    if same_as_kernel(t):
                        u = pre.clone()                          # the unbiased step's own smallest decision margins, per category
                        u.data["decision_margin"][:] = 1e30
                        u.step(act)
                        self.last = dict(band=eps, margins=[float(x) for x in u.data["decision_margin"][:5]], window=None if window is None else window[0])
                        return "tie_" + name + tag + ("" if window is None else f"/substep{window[0]}")
        verdict = None
        for k in range(self.TRIES):
            om = None
            if k >= 7:
                om = self._model(i).copy()
                om.jitter_hulls(1000 * k + i, 2e-7)
            t = self.twin(pre, i, om)
            if k:
                amp = 1e-7 if k < 7 else 1e-6
                q = t.data["qpos"][: self.model.nq]; v = t.data["qvel"][: self.model.nv]
                q += (amp * rng.standard_normal(self.model.nq) * np.maximum(np.abs(q), 0.1)).astype(np.float32)
                v += (amp * rng.standard_normal(self.model.nv) * np.maximum(np.abs(v), 1.0)).astype(np.float32)
            t.step(act)
            if same_as_kernel(t):
                return "agrees_with_kernel"
            if k == 0:
                o2, a2 = _obs_err(np.array(t["obs"][:nobs], np.float64), np.array(t["priv"][:npriv], np.float64), e64, nobs, npriv)
                r2 = float(_rel1(t["reward"][0], e64["reward"][0])); m2 = float(_rel1(np.array(t["metrics"][:8], np.float64), e64["metrics"][:8]).max())
                if t["done"][0] != e64["done"][0] or o2 > ENV_BOUNDS["obs"] or a2 > ENV_BOUNDS["acc"] or r2 > ENV_BOUNDS["reward"] or m2 > ENV_BOUNDS["metrics"]:
                    verdict = "departs_too"
        return verdict


def _dr_model(model, base, fields, e):
    act_jnt = np.asarray(model.a["actuator_trnid"]); dofs = np.asarray(model.a["jnt_dofadr"])[act_jnt]; qadr = np.asarray(model.a["jnt_qposadr"])[act_jnt]
    om = base.copy()
    om.f["body_mass"][:] = fields["body_mass"][e]
    om.f["body_ipos"][3:6] = fields["body_ipos"][e]
    om.f["dof_frictionloss"][dofs] = fields["dof_frictionloss"][e]
    om.f["dof_armature"][dofs] = fields["dof_armature"][e]
    om.f["qpos0"][qadr] = fields["qpos0"][e]
    om.f["actuator_gainprm0"][:] = fields["actuator_gainprm"][e]
    om.f["actuator_biasprm"][1::3] = fields["actuator_biasprm"][e]
    return om


def _rel1(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


def _obs_err(obs, priv, env, nobs, npriv):
    """(worst error over obs / priv outside the accelerometer slots, worst error over the accelerometer slots)"""
    eo = _rel1(obs, env["obs"][:nobs]); ep = _rel1(priv, env["priv"][:npriv])
    acc = max(eo[3:6].max(), ep[nobs + 3: nobs + 6].max())
    eo[3:6] = 0; ep[3:6] = 0; ep[nobs + 3: nobs + 6] = 0
    return max(eo.max(), ep.max()), acc


def _resync(b, envs, model):
    qp = np.stack([np.array(e.data["qpos"][: model.nq]) for e in envs])
    qv = np.stack([np.array(e.data["qvel"][: model.nv]) for e in envs])
    wm = np.stack([np.array(e.data["qacc_warmstart"][: model.nv]) for e in envs])
    b.set_state(qp, qv, wm)


PERTURB_CLONES = 10


def _perturbed_oracle_steps(e, act, model, rng):
    """How far the ORACLE's own outputs of this env step move when its input state is perturbed at the level of fp32 rounding
    through a substep (qpos 1e-6, qvel 5e-6 relative: what test_gpu_parity measures after one mjx.step).  An env step that sits on
    a discontinuity of the model -- a contact or limit row switching on (|dist| ~ 0), a manifold tie, the warm-start pick -- moves
    by orders of magnitude more than a smooth one; no fp32 implementation can agree with an fp64 one there."""
    clones = []
    for _ in range(PERTURB_CLONES):
        c = e.clone()
        q = c.data["qpos"][: model.nq]; v = c.data["qvel"][: model.nv]
        q += 1e-6 * rng.standard_normal(model.nq) * np.maximum(np.abs(q), 0.1)
        v += 5e-6 * rng.standard_normal(model.nv) * np.maximum(np.abs(v), 1.0)
        c.step(act)
        clones.append(c)
    return clones


def _push_info(b, envs, model, which):
    """Overwrites the carried info of the envs in `which` with the oracle's (after an ill-conditioned step the two sides may
    legitimately disagree on a contact flag, and air time / swing peak / last_contact carry that forward).  All other envs keep
    the info the GPU produced itself.  Fields are addressed by name (`Batch.info()` -> `odk_record_field`)."""
    if not which:
        return
    I = b.info()
    same = ("last_act", "last_last_act", "last_last_last_act", "motor_targets", "feet_air_time", "swing_peak", "push", "action_history", "imu_history")
    for i in which:
        e = envs[i]
        I["command"][i] = e["command"][:7]
        for nm in same:
            I[nm][i] = e[nm][: I[nm].shape[1]]
        I["steps"][i] = e["ep_steps"][0]; I["truncation"][i] = e["truncation"][0]; I["episode_done"][i] = e["episode_done"][0]
        I["episode_metrics/sum_reward"][i] = e["ep_sum_reward"][0]; I["episode_metrics/length"][i] = e["ep_length"][0]
        I["episode_metrics/reward_terms"][i] = e["ep_metrics"][:8]
        I["rng"][i, 2] = e.ints("rng_ctr")[0]
        for nm in ("step", "push_step", "push_interval_steps", "imitation_i"):
            I[nm][i] = e.ints(nm)[0]
        lc = e.ints("last_contact")
        I["last_contact"][i] = int(lc[0] != 0) | (int(lc[1] != 0) << 1)
    b.set_records(I["_records"])


def _step_and_compare(torch, b, envs, act, nobs, npriv, t, W, model=None):
    """One env step on both sides.  Well-conditioned env steps: done / truncation must agree exactly and the errors are folded into
    W (dict of maxima).  Ill-conditioned ones (`_perturbed_oracle_steps`: the oracle's own outputs move by more than half a bound under
    rounding-level input noise) are counted in W["n_ill"] and not judged; their carried info is re-synchronised before the next step."""
    model = model if model is not None else b.model
    _push_info(b, envs, model, W.pop("resync_info", []))
    rng = np.random.default_rng(1000 + t)
    clones = [_perturbed_oracle_steps(e, act[i], model, rng) for i, e in enumerate(envs)]
    pre = [e.clone() for e in envs]          # the float64 envs before this step: what the float32 referee starts from
    b.step(torch.tensor(act, device="cuda"))
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy(); rew = b.reward.cpu().numpy(); done = b.done.cpu().numpy()
    trunc = b.truncation.cpu().numpy(); met = b.metrics.cpu().numpy()
    ill_envs = []
    for i, e in enumerate(envs):
        e.step(act[i])
        sens = dict(obs=0.0, acc=0.0, reward=0.0, metrics=0.0, done=0.0)
        for c in clones[i]:
            so, sa = _obs_err(np.array(c["obs"][:nobs]), np.array(c["priv"][:npriv]), e, nobs, npriv)
            sr = float(_rel1(c["reward"][0], e["reward"][0])); sm = float(_rel1(np.array(c["metrics"][:8]), e["metrics"][:8]).max())
            for k, v in (("obs", so), ("acc", sa), ("reward", sr), ("metrics", sm), ("done", float(c["done"][0] != e["done"][0]))):
                sens[k] = max(sens[k], v)
        ill = sens["done"] > 0 or any(sens[k] > SENSITIVITY[k] for k in ("obs", "acc", "reward", "metrics"))
        W["n"] += 1
        W["n_done"] += int(done[i]); W["n_trunc"] += int(trunc[i])
        if ill:
            W["n_ill"] += 1
            ill_envs.append(i)
            continue
        assert done[i] == e["done"][0], (t, i)
        assert trunc[i] == e["truncation"][0], (t, i)
        o, a = _obs_err(obs[i], priv[i], e, nobs, npriv)
        if e["done"][0] != 0 and i in W.get("reset_ill", ()):
            a = 0.0     # auto-reset handed back the first observation of an ill-conditioned reset state
        r = float(_rel1(rew[i], e["reward"][0])); m = float(_rel1(met[i], e["metrics"][:8]).max())
        err = dict(obs=o, acc=a, reward=r, metrics=m)
        if any(err[k] > ENV_BOUNDS[k] for k in err):
            why = envs.f32.adjudicate(pre[i], i, act[i], (obs[i], priv[i], rew[i], met[i], done[i]), e, nobs, npriv, rng) if getattr(envs, "f32", None) else None
            print(f"[outlier] t={t} env={i} err={ {k: float(f'{v:.2e}') for k, v in err.items()} } oracle sensitivity={ {k: float(f'{v:.2e}') for k, v in sens.items()} } "
                  f"float32 oracle: {why}")
            if why:
                W["n_explained"] += 1
                W["n_explained_" + why] = W.get("n_explained_" + why, 0) + 1
                tag = "explained_"
                last = getattr(envs.f32, "last", None)
                if last is not None and why.startswith("tie_"):
                    W["explained_band_max"] = max(W.get("explained_band_max", 0.0), last["band"])
                _referee_log(dict(t=int(t), env=int(i), verdict=why, err={k: float(v) for k, v in err.items()}, **(last or {})))
            else:
                W["n_outlier"] += 1
                tag = "outlier_"
            for k, v in err.items():
                W[tag + k] = max(W.get(tag + k, 0.0), v)
                if not why:     # how far out, in units of the bound: capped
                    W["outlier_over_bound"] = max(W.get("outlier_over_bound", 0.0), v / ENV_BOUNDS[k])
            ill_envs.append(i)     # carried flags may differ after a flipped branch: info re-synchronised like for the set-aside steps
            continue
        for k, v in err.items():
            W[k] = max(W[k], v)
    W["resync_info"] = ill_envs
    return W


def _referee_log(entry):
    """every explained env step with the band that sufficed and the oracle's smallest decision margins of that step ([0] collision
    lengths m, [1] normal cosines, [2] clipping-plane distances m, [3] manifold arg-max steps (relative), [4] warm-start pick (relative)):
    gpurun_out/referee_log.jsonl (committed copy: profiles/r<N>/referee_log.jsonl)"""
    import json, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(root, exist_ok=True)
    entry["test"] = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    with open(os.path.join(root, "referee_log.jsonl"), "a") as f:
        f.write(json.dumps(entry) + "\n")


def _new_W():
    return dict(obs=0.0, acc=0.0, reward=0.0, metrics=0.0, n_done=0, n_trunc=0, n=0, n_ill=0, n_outlier=0, n_explained=0)


def _errs(W):
    """the judged quantities: worst errors over the well-conditioned env steps inside the bounds, the fraction of env steps set
    aside as ill-conditioned, the fraction of judged ones beyond a bound (and how far they were)"""
    judged = max(W["n"] - W["n_ill"], 1)
    out = dict({k: W[k] for k in ("obs", "acc", "reward", "metrics")}, ill_fraction=W["n_ill"] / max(W["n"], 1),
               outlier_fraction=W["n_outlier"] / judged, explained_fraction=W["n_explained"] / judged, env_steps=W["n"], judged_env_steps=judged,
               outliers_unexplained=W["n_outlier"], explained_by_f32_oracle=W["n_explained"], outlier_over_bound=W.get("outlier_over_bound", 0.0))
    out.update({k: v for k, v in W.items() if k.startswith(("outlier_", "explained_", "n_explained_"))})
    if "reset_ill" in W:
        out["ill_resets"] = len(W["reset_ill"])
    return out


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash"])
def test_reset_matches_oracle(oracle_mod, parity_log, task):
    torch, model, b, envs, keep = _mk(oracle_mod, task, 32)
    b.reset(seed=5, env_id_offset=100)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    qpos, qvel, warm = b.get_state()
    I = b.info()
    W = dict(obs=0.0, acc=0.0, qpos=0.0, qvel=0.0)
    for i, e in enumerate(envs):
        e.reset(5, 100 + i)
    for i, e in enumerate(envs):
        W["qpos"] = max(W["qpos"], np.abs(qpos[i] - e.data["qpos"][: model.nq]).max())
        W["qvel"] = max(W["qvel"], np.abs(qvel[i] - e.data["qvel"][: model.nv]).max())
        # the accelerometer spikes to O(100) m/s^2 at reset (feet start 1.5 cm inside the floor): judged relatively, own bound
        o, a = _obs_err(obs[i], priv[i], e, 101, 212)
        W["obs"] = max(W["obs"], o); W["acc"] = max(W["acc"], a)
        np.testing.assert_allclose(I["command"][i], e["command"], rtol=1e-6, atol=1e-7)
        assert int(I["push_interval_steps"][i]) == int(e.ints("push_interval_steps")[0])
    b.close()
    parity_log.check(f"reset/{task}", RESET_BOUNDS, **W)


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_step_sequence_with_resync(oracle_mod, parity_log, task):
    """60 env steps with random actions.  The physics state is re-synchronised from the oracle before every
    step (fp32-vs-fp64 chaos through contact would otherwise dominate), everything else -- info ring
    buffers, RNG counters, episode counters, auto-reset -- runs free on the GPU."""
    def edit(cfg):
        cfg.episode_length = 25   # exercise truncation + auto-reset
    torch, model, b, envs, keep = _mk(oracle_mod, task, 32, edit)
    n = len(envs)
    b.reset(seed=9)
    for i, e in enumerate(envs):
        e.reset(9, i)
    rng = np.random.default_rng(0)
    W = _new_W()
    W["reset_ill"] = _ill_resets(envs, model, 101)
    for t in range(60):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, 101, 212, t, W)
    assert W["n_done"] > 0 and W["n_trunc"] > 0, "sequence must cross terminations and truncations"
    I = b.info()
    for i, e in enumerate(envs):
        np.testing.assert_allclose(I["last_act"][i], e["last_act"][:14], atol=1e-6)
        np.testing.assert_allclose(I["action_history"][i], e["action_history"][:42], atol=1e-6)
        assert int(I["rng"][i, 2]) == int(e.ints("rng_ctr")[0])
        assert int(I["imitation_i"][i]) == int(e.ints("imitation_i")[0])
    b.close()
    parity_log.check(f"env_step/{task}", {**ENV_BOUNDS, **_set_aside(task)}, **_errs(W))


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_step_sequence_of_the_duck_with_elliptic_cones(oracle_mod, parity_log, task):
    """`<option cone="elliptic">` on the duck itself: the model's `opt_cone` switched to 1 (impratio as in the file: 1), the env
    kernels' own instantiation with the cone code (`ShapeAE` / `ShapeBE`, odk_engine.hip) against the oracle env with the same model: reset, then
    40 env steps with random actions, physics re-synchronised before every step as in the test above, at the same bounds."""
    def edit(cfg):
        cfg.episode_length = 25
    torch, model, b, envs, keep = _mk(oracle_mod, task, 32, edit, model_edit=dict(opt_cone=np.array([1], np.int32)))
    n = len(envs)
    b.reset(seed=13)
    for i, e in enumerate(envs):
        e.reset(13, i)
    rng = np.random.default_rng(2)
    W = _new_W()
    W["reset_ill"] = _ill_resets(envs, model, 101)
    for t in range(40):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, 101, 212, t, W)
    assert W["n_done"] > 0 and W["n_trunc"] > 0
    b.close()
    parity_log.check(f"env_step_elliptic/{task}", {**ENV_BOUNDS, **_set_aside(task)}, **_errs(W))


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash"])
def test_in_step_command_resample(oracle_mod, parity_log, task):
    """sample_command inside step (joystick.py:456-466): info["step"] is preset to 499 / 500 / 123 on both sides through the raw
    record; the step that takes it past 500 must draw the same new command (same counter-RNG draws, incl. the all-zero
    branch) and reset the counter to 0, the others keep theirs, and the next step's obs carry the new command."""
    torch, model, b, envs, keep = _mk(oracle_mod, task, 64)
    n = len(envs)
    b.reset(seed=13)
    for i, e in enumerate(envs):
        e.reset(13, i)
    I = b.info()
    preset = np.where(np.arange(n) % 4 == 0, 123, np.where(np.arange(n) % 4 == 1, 499, 500)).astype(np.int32)
    I["step"][:] = preset                 # info["step"], by name
    b.set_records(I["_records"])
    for i, e in enumerate(envs):
        e.ints("step")[0] = int(preset[i])
    old_cmd = np.stack([np.array(e["command"][:7]) for e in envs])
    rng = np.random.default_rng(4)
    W = _new_W()
    n_resampled = n_zero = 0
    for t in range(3):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, 101, 212, t, W)
        I = b.info()
        for i, e in enumerate(envs):
            cmd_g = I["command"][i]; cmd_o = np.array(e["command"][:7])
            np.testing.assert_allclose(cmd_g, cmd_o, rtol=1e-6, atol=1e-7)
            assert int(I["step"][i]) == int(e.ints("step")[0]), (t, i)
            if t == 0:
                if preset[i] == 500:     # 501 > 500: resampled, counter back to 0
                    assert int(e.ints("step")[0]) == 0
                    n_resampled += int(not np.array_equal(cmd_o, old_cmd[i])); n_zero += int(np.all(cmd_o == 0))
                else:                    # 124 / 500: not yet
                    assert np.array_equal(cmd_o, old_cmd[i]) and (int(e.ints("step")[0]) == preset[i] + 1 or e["done"][0] != 0)
            if t == 1 and preset[i] == 499 and envs[i]["done"][0] == 0:
                assert int(e.ints("step")[0]) == 0
    assert n_resampled >= n // 2 - 1 and W["n_done"] < n // 2
    b.close()
    parity_log.rec(f"command_resample/{task}", None, resampled=n_resampled, all_zero_draws=n_zero)
    parity_log.check(f"command_resample/{task}", {**ENV_BOUNDS, **SET_ASIDE}, **_errs(W))


@pytest.mark.parametrize("task,standing", [("flat_terrain_backlash", False), ("flat_terrain", False), ("rough_terrain_backlash", False), ("flat_terrain_backlash", True)])
def test_env_step_with_domain_randomisation(oracle_mod, parity_log, task, standing):
    """randomize.py's per-env model fields through odk_reset / odk_step (not only the physics call): reset obs, then 30
    resynchronised env steps, obs / reward / metrics against oracle envs that each own a randomised copy of the model."""
    from open_duck_playground_amd import randomize
    from open_duck_playground_amd.model import load_task_model
    n = 32
    fields, _ = randomize.domain_randomize(load_task_model(task), np.random.default_rng(17), n)

    def edit(cfg):
        cfg.episode_length = 20
    torch, model, b, envs, keep = _mk(oracle_mod, task, n, edit, standing=standing, dr_fields=fields)
    nobs, npriv = (85, 153) if standing else (101, 212)
    b.reset(seed=21)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    WR = dict(obs=0.0, acc=0.0)
    for i, e in enumerate(envs):
        e.reset(21, i)
    ill = _ill_resets(envs, model, nobs)
    for i, e in enumerate(envs):
        o, a = _obs_err(obs[i], priv[i], e, nobs, npriv)
        WR["obs"] = max(WR["obs"], o); WR["acc"] = max(WR["acc"], 0.0 if i in ill else a)
    rng = np.random.default_rng(6)
    W = _new_W()
    W["reset_ill"] = ill
    for t in range(30):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, nobs, npriv, t, W)
    b.close()
    tag = f"env_step_dr/{task}/{'standing' if standing else 'joystick'}"
    parity_log.check(tag + "/reset", dict(obs=RESET_BOUNDS["obs"], acc=RESET_BOUNDS["acc"]), **WR)
    parity_log.check(tag, {**ENV_BOUNDS, **_set_aside(task)}, **_errs(W))


def test_domain_randomisation_changes_the_env_step(oracle_mod):
    """guard for the test above: with the SAME seed and actions, randomised parameters must move the GPU outputs (a kernel that
    ignored the per-env block would still match a nominal oracle)."""
    import torch
    from open_duck_playground_amd import engine, randomize
    from open_duck_playground_amd.model import load_task_model
    model = load_task_model("flat_terrain_backlash")
    n = 32
    fields, _ = randomize.domain_randomize(model, np.random.default_rng(17), n)
    outs = []
    for dr in (False, True):
        b = engine.Batch(model, n)
        if dr:
            randomize.apply(b, fields)
        b.reset(seed=21)
        g = torch.Generator(device="cuda").manual_seed(0)
        for _ in range(3):
            b.step(torch.empty(n, 14, device="cuda").uniform_(-1, 1, generator=g))
        outs.append((b.priv.clone(), b.reward.clone()))
        b.close()
    assert float((outs[0][0] - outs[1][0]).abs().max()) > 1e-2 and float((outs[0][1] - outs[1][1]).abs().max()) > 1e-4


def test_free_running_rollout_stays_close(oracle_mod, parity_log):
    """5 free-running env steps (50 substeps): median state error small, no NaNs."""
    def edit(cfg):
        cfg.noise_level = 0.0
        cfg.push_enable = 0.0
    torch, model, b, envs, keep = _mk(oracle_mod, "flat_terrain", 64, edit)
    n = len(envs)
    b.reset(seed=1)
    for i, e in enumerate(envs):
        e.reset(1, i)
    rng = np.random.default_rng(1)
    for t in range(5):
        act = (0.3 * rng.uniform(-1, 1, (n, 14))).astype(np.float32)
        b.step(torch.tensor(act, device="cuda"))
        for i, e in enumerate(envs):
            e.step(act[i])
    qpos, qvel, _ = b.get_state()
    ref = np.stack([np.array(e.data["qpos"][: model.nq]) for e in envs])
    err = np.abs(qpos - ref).max(axis=1)
    assert np.isfinite(qpos).all()
    b.close()
    parity_log.check("free_running_5_steps/flat_terrain", dict(median_qpos_abs=5e-5, frac_above_5e3=0.1),
                     median_qpos_abs=float(np.median(err)), frac_above_5e3=float((err >= 5e-3).mean()), max_qpos_abs=float(err.max()))


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_standing_env_matches_oracle(oracle_mod, parity_log, task):
    """Standing (reference standing.py): reset + 40 resynchronised steps; obs rows are 85 / 153 floats wide."""
    def edit(cfg):
        cfg.episode_length = 25
    torch, model, b, envs, keep = _mk(oracle_mod, task, 32, edit, standing=True)
    n = len(envs)
    assert tuple(b.obs.shape) == (n, 85) and tuple(b.priv.shape) == (n, 153)
    b.reset(seed=11)
    for i, e in enumerate(envs):
        e.reset(11, i)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    qpos, qvel, _ = b.get_state()
    WR = dict(obs=0.0, acc=0.0, qvel=0.0)
    ill = _ill_resets(envs, model, 85)
    for i, e in enumerate(envs):
        WR["qvel"] = max(WR["qvel"], np.abs(qvel[i] - e.data["qvel"][: model.nv]).max())
        o, a = _obs_err(obs[i], priv[i], e, 85, 153)
        WR["obs"] = max(WR["obs"], o); WR["acc"] = max(WR["acc"], 0.0 if i in ill else a)
    assert np.abs(qvel[:, :6]).max() > 0.05      # the Standing reset range
    rng = np.random.default_rng(2)
    W = _new_W()
    W["reset_ill"] = ill
    for t in range(40):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, 85, 153, t, W)
    assert W["n_done"] > 0
    b.close()
    parity_log.check(f"standing/{task}/reset", dict(obs=RESET_BOUNDS["obs"], acc=RESET_BOUNDS["acc"], qvel=RESET_BOUNDS["qvel"]), **WR)
    parity_log.check(f"standing/{task}", {**ENV_BOUNDS, **_set_aside(task)}, **_errs(W))


def test_standing_python_env_surface():
    from open_duck_playground_amd import standing
    env = standing.Standing(task="flat_terrain", num_envs=64)
    assert env.observation_size == {"state": (85,), "privileged_state": (153,)}
    st = env.reset(0)
    import torch
    st = env.step(st, torch.zeros(64, 14, device="cuda"))
    assert set(st.metrics) == {"cost/orientation", "cost/head_pos", "cost/torques", "cost/action_rate", "cost/stand_still", "reward/alive", "swing_peak"}
    assert tuple(st.obs["state"].shape) == (64, 85) and torch.isfinite(st.obs["privileged_state"]).all()
    assert float(st.metrics["reward/alive"].min()) == 20.0 and float(st.metrics["cost/head_pos"].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------------------
# Robots that are not the duck through odk_reset / odk_step (reference README.md:74-85 "Adding a new robot"): the env kernels
# instantiated for their shapes -- observation / action / history sizes from the robot's actuator count, actuators, default pose, feet and
# imu sites, sensor addresses from the ModelBlob -- against the nu-generic oracle env, at the duck's bounds.
ROBOTS = [("biped12.xml", 12, 89, 194), ("tail_biped.xml", 15, 107, 221)]
# The error bounds are the duck's (measured: obs 2.3e-5, accelerometer 2.0e-4, reward 5.4e-6, metrics 2.0e-4; 0 unexplained).  What differs is how
# many env steps the ORACLE sets aside by its own sensitivity: these robots stand on BOX feet, whose four sole vertices touch a plane floor at the
# same depth -- the manifold's arg-max steps tie by construction whenever a foot lies flat (measured 8.8 % / 10.4 % / 12.6 % of the env steps
# against the duck's 2-5 % on its 17-vertex hull): bound = measured + ~7 points.
SET_ASIDE_BOX = dict(SET_ASIDE, ill_fraction=0.2)


@pytest.mark.parametrize("xml,nu,nobs,npriv", ROBOTS)
def test_reset_of_a_robot_that_is_not_the_duck(oracle_mod, parity_log, xml, nu, nobs, npriv):
    torch, model, b, envs, keep = _mk(oracle_mod, xml, 32)
    assert model.nu == nu and (b.nobs, b.npriv) == (nobs, npriv) == (envs[0].nobs, envs[0].npriv) and tuple(b.obs.shape) == (32, nobs)
    assert b.lanes_per_env == 32
    b.reset(seed=5, env_id_offset=100)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    qpos, qvel, warm = b.get_state()
    I = b.info()
    W = dict(obs=0.0, acc=0.0, qpos=0.0, qvel=0.0)
    for i, e in enumerate(envs):
        e.reset(5, 100 + i)
    ill = _ill_resets(envs, model, nobs)
    for i, e in enumerate(envs):
        W["qpos"] = max(W["qpos"], np.abs(qpos[i] - e.data["qpos"][: model.nq]).max())
        W["qvel"] = max(W["qvel"], np.abs(qvel[i] - e.data["qvel"][: model.nv]).max())
        o, a = _obs_err(obs[i], priv[i], e, nobs, npriv)
        W["obs"] = max(W["obs"], o); W["acc"] = max(W["acc"], 0.0 if i in ill else a)
        np.testing.assert_allclose(I["command"][i], e["command"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(I["motor_targets"][i], e["motor_targets"][:nu], rtol=1e-6, atol=1e-7)
        assert I["action_history"].shape[1] == 3 * nu
        assert int(I["push_interval_steps"][i]) == int(e.ints("push_interval_steps")[0])
    b.close()
    parity_log.check(f"robot_env/{xml}/reset", RESET_BOUNDS, **W)


@pytest.mark.parametrize("xml,nu,nobs,npriv", ROBOTS)
def test_step_sequence_of_a_robot_that_is_not_the_duck(oracle_mod, parity_log, xml, nu, nobs, npriv):
    """`test_step_sequence_with_resync` for a robot that is not the duck: 60 env steps with random actions, observation noise, pushes, truncation
    and auto-reset on; physics re-synchronised before every step, the carried info (rings sized by the robot's actuator count, RNG counters,
    episode counters) running free on the GPU."""
    def edit(cfg):
        cfg.episode_length = 25
    torch, model, b, envs, keep = _mk(oracle_mod, xml, 32, edit)
    n = len(envs)
    b.reset(seed=9)
    for i, e in enumerate(envs):
        e.reset(9, i)
    rng = np.random.default_rng(0)
    W = _new_W()
    W["reset_ill"] = _ill_resets(envs, model, nobs)
    for t in range(60):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, nu)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, nobs, npriv, t, W)
    assert W["n_trunc"] > 0, "sequence must cross truncations"
    I = b.info()
    for i, e in enumerate(envs):
        np.testing.assert_allclose(I["last_act"][i], e["last_act"][:nu], atol=1e-6)
        np.testing.assert_allclose(I["last_last_last_act"][i], e["last_last_last_act"][:nu], atol=1e-6)
        np.testing.assert_allclose(I["action_history"][i], e["action_history"][:3 * nu], atol=1e-6)
        assert int(I["rng"][i, 2]) == int(e.ints("rng_ctr")[0])
        assert int(I["imitation_i"][i]) == 0
    b.close()
    parity_log.check(f"robot_env/{xml}/env_step", {**ENV_BOUNDS, **SET_ASIDE_BOX}, **_errs(W))


def test_a_robot_that_is_not_the_duck_with_domain_randomisation_and_command_resampling(oracle_mod, parity_log):
    """biped12 with randomize.py's per-env model fields through odk_reset / odk_step, and info["step"] preset so that some envs resample their
    command inside the sequence (draws 13 + 2 nu ...)."""
    from open_duck_playground_amd import randomize
    xml, nu, nobs, npriv = ROBOTS[0]
    n = 32
    fields, _ = randomize.domain_randomize(_xml_model(xml), np.random.default_rng(17), n)

    def edit(cfg):
        cfg.episode_length = 20
    torch, model, b, envs, keep = _mk(oracle_mod, xml, n, edit, dr_fields=fields)
    b.reset(seed=21)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    WR = dict(obs=0.0, acc=0.0)
    for i, e in enumerate(envs):
        e.reset(21, i)
    ill = _ill_resets(envs, model, nobs)
    for i, e in enumerate(envs):
        o, a = _obs_err(obs[i], priv[i], e, nobs, npriv)
        WR["obs"] = max(WR["obs"], o); WR["acc"] = max(WR["acc"], 0.0 if i in ill else a)
    I = b.info()
    preset = np.where(np.arange(n) % 2 == 0, 500, 7).astype(np.int32)
    I["step"][:] = preset
    b.set_records(I["_records"])
    old_cmd = np.stack([np.array(e["command"][:7]) for e in envs])
    for i, e in enumerate(envs):
        e.ints("step")[0] = int(preset[i])
    rng = np.random.default_rng(6)
    W = _new_W()
    W["reset_ill"] = ill
    n_resampled = 0
    for t in range(30):
        _resync(b, envs, model)
        act = rng.uniform(-1, 1, (n, nu)).astype(np.float32)
        _step_and_compare(torch, b, envs, act, nobs, npriv, t, W)
        if t == 0:
            I = b.info()
            for i, e in enumerate(envs):
                np.testing.assert_allclose(I["command"][i], np.array(e["command"][:7]), rtol=1e-6, atol=1e-7)
                n_resampled += int(preset[i] == 500 and not np.array_equal(np.array(e["command"][:7]), old_cmd[i]))
    assert n_resampled >= n // 2 - 2
    b.close()
    parity_log.check(f"robot_env/{xml}/dr/reset", dict(obs=RESET_BOUNDS["obs"], acc=RESET_BOUNDS["acc"]), **WR)
    parity_log.check(f"robot_env/{xml}/dr/env_step", {**ENV_BOUNDS, **SET_ASIDE_BOX}, **_errs(W))


def test_python_env_surface_of_a_robot_from_its_xml():
    """`Joystick(xml_path=...)` (runner --xml): sizes, stepping, the evaluation sibling, and the refusals (Standing / imitation are the duck's)."""
    import os
    import torch
    from open_duck_playground_amd import engine, joystick, standing
    xml = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "biped12.xml")
    env = joystick.Joystick(xml_path=xml, num_envs=64)
    assert env.action_size == 12 and env.observation_size == {"state": (89,), "privileged_state": (194,)} and env.xml_path == xml
    st = env.reset(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(5):
        st = env.step(st, torch.empty(64, 12, device="cuda").uniform_(-1, 1, generator=g))
    assert tuple(st.obs["state"].shape) == (64, 89) and torch.isfinite(st.obs["privileged_state"]).all() and torch.isfinite(st.reward).all()
    assert float(st.metrics["reward/alive"].min()) == 20.0 and float(st.metrics["reward/imitation"].abs().max()) == 0.0
    assert tuple(st.info["action_history"].shape) == (64, 36) and tuple(st.info["motor_targets"].shape) == (64, 12)
    ev = env.make_eval_env(128)      # asks for 64 lanes per env; this robot's kernels exist at 32
    assert ev.batch.lanes_per_env == 32 and ev.action_size == 12
    ev.reset(1)
    ev.step(None, torch.zeros(128, 12, device="cuda"))
    assert torch.isfinite(ev.batch.obs).all()
    env.randomize(np.random.default_rng(0))      # randomize.py's fields apply to any robot
    env.step(st, torch.zeros(64, 12, device="cuda"))
    assert torch.isfinite(env.batch.reward).all()
    with pytest.raises(engine.OdkError, match="not the duck"):
        standing.Standing(xml_path=xml, num_envs=8).reset(0)


def test_a_model_whose_own_xml_sets_the_elliptic_cone_evaluates_at_32_lanes(tmp_path):
    """the cone of a model may come from its XML (`<option cone="elliptic">`, the reference's way) and not from the
    `cone` config switch; the evaluation sibling asks for 64 lanes per env on small batches and must still get the cone kernels (32)."""
    import os
    import torch
    from open_duck_playground_amd import joystick
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "biped12.xml")).read()
    xml = tmp_path / "biped12_elliptic.xml"
    xml.write_text(src.replace('<option timestep="0.002"', '<option cone="elliptic" impratio="2" timestep="0.002"'))
    env = joystick.Joystick(xml_path=str(xml), num_envs=32)
    assert int(env.mj_model.a["opt_cone"][0]) == 1
    ev = env.make_eval_env(64)
    assert ev.batch.lanes_per_env == 32
    st = ev.reset(0)
    st = ev.step(st, torch.zeros(64, 12, device="cuda"))
    assert torch.isfinite(st.obs["state"]).all()
    from open_duck_playground_amd.model import Model, load_task_model
    duck = load_task_model("flat_terrain")
    env2 = joystick.Joystick(model=Model({**duck.a, "opt_cone": np.array([1], np.int32)}), num_envs=32)      # the duck with the cone in the model itself
    ev2 = env2.make_eval_env(64)
    assert ev2.batch.lanes_per_env == 32
    ev2.step(ev2.reset(0), torch.zeros(64, 14, device="cuda"))
    assert torch.isfinite(ev2.batch.obs).all()
