"""GPU parity of the full env step (wrappers + Joystick.step + obs/reward) vs the CPU oracle env,
with observation noise, action delay and pushes ON (both sides draw from the same counter RNG)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(oracle_mod, task, n, cfg_edit=None, standing=False):
    import torch
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import load_task_model
    model = load_task_model(task)
    cfg = engine.default_config(standing)
    if cfg_edit:
        cfg_edit(cfg)
    b = engine.Batch(model, n, cfg)
    om = oracle_mod.OracleModel(model.blob())
    prm = oracle_mod.OraclePRM(engine.load_prm())
    envs = [oracle_mod.OracleEnv(om, prm, standing=standing) for _ in range(n)]
    for e in envs:
        e.cfg["episode_length"][0] = cfg.episode_length
        e.cfg["noise_level"][0] = cfg.noise_level
        e.cfg["push_enable"][0] = cfg.push_enable
    return torch, model, b, envs, (om, prm)


def _close(a, b, rtol, atol):
    return np.abs(a - b) <= atol + rtol * np.abs(b)


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash"])
def test_reset_matches_oracle(oracle_mod, task):
    torch, model, b, envs, keep = _mk(oracle_mod, task, 32)
    b.reset(seed=5, env_id_offset=100)
    obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy()
    qpos, qvel, warm = b.get_state()
    rec = b.records()
    for i, e in enumerate(envs):
        e.reset(5, 100 + i)
        np.testing.assert_allclose(qpos[i], e.data["qpos"][: model.nq], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(qvel[i], e.data["qvel"][: model.nv], rtol=1e-5, atol=1e-6)
        # accelerometer spikes to O(100) m/s^2 at reset (feet start 1.5 cm inside the floor): compare relatively
        assert _close(obs[i], e["obs"][:101], 2e-3, 2e-3).all(), (i, np.abs(obs[i] - e["obs"][:101]).max())
        assert _close(priv[i], e["priv"][:212], 2e-3, 2e-3).all()
        info = rec[i, model.nq + 2 * model.nv:]
        np.testing.assert_allclose(info[0:7], e["command"], rtol=1e-6, atol=1e-7)
        assert int(info[138:139].view(np.int32)[0]) == int(e.ints("push_interval_steps")[0])
    b.close()


@pytest.mark.parametrize("task", ["flat_terrain", "flat_terrain_backlash", "rough_terrain_backlash"])
def test_step_sequence_with_resync(oracle_mod, task):
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
    n_done = n_trunc = 0
    for t in range(60):
        qp = np.stack([np.array(e.data["qpos"][: model.nq]) for e in envs])
        qv = np.stack([np.array(e.data["qvel"][: model.nv]) for e in envs])
        wm = np.stack([np.array(e.data["qacc_warmstart"][: model.nv]) for e in envs])
        b.set_state(qp, qv, wm)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        b.step(torch.tensor(act, device="cuda"))
        obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy(); rew = b.reward.cpu().numpy(); done = b.done.cpu().numpy()
        trunc = b.truncation.cpu().numpy(); met = b.metrics.cpu().numpy()
        bad = 0
        for i, e in enumerate(envs):
            e.step(act[i])
            assert done[i] == e["done"][0], (t, i)
            assert trunc[i] == e["truncation"][0], (t, i)
            ok = _close(obs[i], e["obs"][:101], 5e-3, 5e-3).all() and _close(priv[i], e["priv"][:212], 5e-3, 5e-3).all()
            ok = ok and _close(rew[i], e["reward"][0], 5e-3, 1e-3) and _close(met[i], e["metrics"][:8], 1e-2, 2e-3).all()
            bad += 0 if ok else 1
            n_done += int(done[i]); n_trunc += int(trunc[i])
        assert bad <= 1, (t, bad)   # a contact-manifold tie may flip between fp32 and fp64 once in a while
    assert n_done > 0 and n_trunc > 0, "sequence must cross terminations and truncations"
    rec = b.records()
    for i, e in enumerate(envs):
        info = rec[i, model.nq + 2 * model.nv:]
        np.testing.assert_allclose(info[7:21], e["last_act"][:14], atol=1e-6)
        np.testing.assert_allclose(info[69:111], e["action_history"][:42], atol=1e-6)
        assert int(info[135:136].view(np.int32)[0]) == int(e.ints("rng_ctr")[0])
        assert int(info[139:140].view(np.int32)[0]) == int(e.ints("imitation_i")[0])
    b.close()


def test_free_running_rollout_stays_close(oracle_mod):
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
    assert np.median(err) < 2e-4 and (err < 5e-3).mean() > 0.9, (np.median(err), err.max())
    b.close()


@pytest.mark.parametrize("task", ["flat_terrain", "rough_terrain_backlash"])
def test_standing_env_matches_oracle(oracle_mod, task):
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
    for i, e in enumerate(envs):
        np.testing.assert_allclose(qvel[i], e.data["qvel"][: model.nv], rtol=1e-5, atol=1e-6)
        assert _close(obs[i], e["obs"][:85], 2e-3, 2e-3).all(), (i, np.abs(obs[i] - e["obs"][:85]).max())
        assert _close(priv[i], e["priv"][:153], 2e-3, 2e-3).all()
    assert np.abs(qvel[:, :6]).max() > 0.05      # the Standing reset range
    rng = np.random.default_rng(2)
    n_done = 0
    for t in range(40):
        qp = np.stack([np.array(e.data["qpos"][: model.nq]) for e in envs])
        qv = np.stack([np.array(e.data["qvel"][: model.nv]) for e in envs])
        wm = np.stack([np.array(e.data["qacc_warmstart"][: model.nv]) for e in envs])
        b.set_state(qp, qv, wm)
        act = rng.uniform(-1, 1, (n, 14)).astype(np.float32)
        b.step(torch.tensor(act, device="cuda"))
        obs = b.obs.cpu().numpy(); priv = b.priv.cpu().numpy(); rew = b.reward.cpu().numpy(); done = b.done.cpu().numpy()
        met = b.metrics.cpu().numpy()
        bad = 0
        for i, e in enumerate(envs):
            e.step(act[i])
            assert done[i] == e["done"][0], (t, i)
            ok = _close(obs[i], e["obs"][:85], 5e-3, 5e-3).all() and _close(priv[i], e["priv"][:153], 5e-3, 5e-3).all()
            ok = ok and _close(rew[i], e["reward"][0], 5e-3, 1e-3) and _close(met[i], e["metrics"][:8], 1e-2, 2e-3).all()
            bad += 0 if ok else 1
            n_done += int(done[i])
        assert bad <= 1, (t, bad)
    assert n_done > 0
    b.close()


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
