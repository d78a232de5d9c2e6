"""CPU tests of the PPO learner (brax semantics) and its data-parallel path
(world_size 2 over gloo)."""
import math
import os

import numpy as np
import pytest
import torch

from open_duck_playground_amd.ppo import train as T
from open_duck_playground_amd.ppo.networks import PPONetworks, RunningStats, tanh_normal_entropy, tanh_normal_log_prob


def test_network_shapes_and_parameter_counts():
    net = PPONetworks(101, 212, 14)
    assert sum(p.numel() for p in net.policy.parameters()) == 220060
    assert sum(p.numel() for p in net.value.parameters()) == 273409
    loc, scale = net.dist_params(torch.randn(5, 101))
    assert loc.shape == (5, 14) and (scale > 0.001).all()
    assert net.values(torch.randn(5, 212)).shape == (5,)


def test_tanh_normal_log_prob_matches_torch_distributions():
    torch.manual_seed(0)
    loc, scale, raw = torch.randn(7, 14), torch.rand(7, 14) + 0.1, torch.randn(7, 14)
    base = torch.distributions.Normal(loc, scale)
    a = torch.tanh(raw)
    ref = (base.log_prob(raw) - torch.log(1 - a ** 2 + 1e-12)).sum(-1)
    np.testing.assert_allclose(tanh_normal_log_prob(loc, scale, raw).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.isfinite(tanh_normal_entropy(loc, scale, raw)).all()


def test_gae_matches_naive_recursion():
    rng = np.random.default_rng(0)
    Tn, B = 20, 6
    rew, val = rng.normal(size=(Tn, B)), rng.normal(size=(Tn, B))
    boot = rng.normal(size=B)
    term = (rng.uniform(size=(Tn, B)) < 0.1).astype(float)
    trunc = (rng.uniform(size=(Tn, B)) < 0.05).astype(float) * (1 - term)
    lam, disc = 0.95, 0.97
    vs, adv = T.compute_gae(*(torch.tensor(x) for x in (trunc, term, rew, val)), torch.tensor(boot), lam, disc)
    v1 = np.concatenate([val[1:], boot[None]])
    mask = 1 - trunc
    delta = (rew + disc * (1 - term) * v1 - val) * mask
    acc = np.zeros(B); vmv = np.zeros((Tn, B))
    for t in reversed(range(Tn)):
        acc = delta[t] + disc * (1 - term[t]) * mask[t] * lam * acc
        vmv[t] = acc
    vs_ref = vmv + val
    adv_ref = (rew + disc * (1 - term) * np.concatenate([vs_ref[1:], boot[None]]) - val) * mask
    np.testing.assert_allclose(vs.numpy(), vs_ref, atol=1e-12)
    np.testing.assert_allclose(adv.numpy(), adv_ref, atol=1e-12)


def test_running_stats_matches_numpy():
    rs = RunningStats(5)
    rng = np.random.default_rng(1)
    chunks = [rng.normal(2.0, 3.0, size=(50, 5)).astype(np.float32) for _ in range(4)]
    for c in chunks:
        rs.update(torch.tensor(c))
    allx = np.concatenate(chunks)
    np.testing.assert_allclose(rs.mean.numpy(), allx.mean(0), rtol=1e-5)
    np.testing.assert_allclose(rs.std.numpy(), allx.std(0), rtol=1e-4)


def _fake_rollout(B, Tn, gen):
    r = lambda *s: torch.randn(*s, generator=gen)
    return dict(obs=r(B, Tn, 101), priv=r(B, Tn, 212), raw_action=r(B, Tn, 14), log_prob=r(B, Tn) * 0.1 - 14.0,
                reward=torch.rand(B, Tn, generator=gen), done=(torch.rand(B, Tn, generator=gen) < 0.05).float(),
                truncation=torch.zeros(B, Tn), last_priv=r(B, 212))


def test_sgd_epoch_reduces_value_loss():
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(0)
    net = PPONetworks(101, 212, 14, (64, 32), (64, 32))
    data = _fake_rollout(64, 10, gen)
    cfg = T.ppo_config(); cfg.update(num_minibatches=4, num_updates_per_batch=4)
    opt = torch.optim.Adam(net.parameters(), 1e-3)
    first = T.ppo_loss(net, data, cfg)[1]["v_loss"]
    for _ in range(5):
        T.sgd_epoch(net, opt, data, cfg, gen)
    assert T.ppo_loss(net, data, cfg)[1]["v_loss"] < first


def _dist_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = PPONetworks(101, 212, 14, (32, 16), (32, 16))
    gen = torch.Generator().manual_seed(100 + rank)          # different rollouts per rank
    data = _fake_rollout(32, 8, gen)
    net.norm_obs.update(data["obs"], dist.group.WORLD); net.norm_priv.update(data["priv"], dist.group.WORLD)
    cfg = T.ppo_config(); cfg.update(num_minibatches=4, num_updates_per_batch=2)
    opt = torch.optim.Adam(net.parameters(), 1e-3)
    m = T.sgd_epoch(net, opt, data, cfg, torch.Generator().manual_seed(7), world=world, group=dist.group.WORLD)
    T.assert_replicas_identical(net, dist.group.WORLD)          # passes: gradients and moments were all-reduced
    ml = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(ml, m["total_loss"])
    assert torch.equal(ml[0], ml[1])                            # the reported losses are means over the ranks
    if rank == 1:
        net.policy.layers[0].bias.data[0] += 1e-3
    try:
        T.assert_replicas_identical(net, dist.group.WORLD)
        diverged = False
    except RuntimeError:
        diverged = True
    assert diverged                                             # ... and a drifting replica is caught on every rank
    if rank == 1:
        net.policy.layers[0].bias.data[0] -= 1e-3
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    stats = torch.cat([net.norm_obs.mean, net.norm_obs.std])
    gs = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gs, stats)
    if rank == 0:
        out.put((bool(all(torch.equal(gathered[0], g) for g in gathered)), bool(all(torch.equal(gs[0], g) for g in gs)),
                 float(net.norm_obs.count), data["obs"].reshape(-1, 101).mean(0)[:3].tolist(), net.norm_obs.mean[:3].tolist()))
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """Gradients are all-reduced every SGD step and normaliser moments every rollout: parameters and statistics stay
    bit-identical across ranks although each rank trains on its own env shard."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    same_params, same_stats, count, local_mean, global_mean = res
    assert same_params and same_stats
    assert count == 2 * 32 * 8                                   # both shards counted
    assert not np.allclose(local_mean, global_mean, atol=1e-6)   # statistics are global, not rank-local


class _ToyEnv:
    """CPU stand-in with the Joystick surface: reward is high when tanh-action[0] tracks obs[0]."""

    def __init__(self, num_envs=64, seed=0):
        self.num_envs, self.device = num_envs, torch.device("cpu")
        self.action_size = 2
        self.observation_size = {"state": (5,), "privileged_state": (7,)}
        self.g = torch.Generator().manual_seed(seed)
        self.resets = 0

    def make_eval_env(self, n):
        return _ToyEnv(n, seed=99)

    def _state(self, reward, done):
        from open_duck_playground_amd.joystick import State
        self.target = torch.rand(self.num_envs, generator=self.g) * 1.6 - 0.8
        obs = torch.cat([self.target[:, None], torch.randn(self.num_envs, 4, generator=self.g) * 0.1], 1)
        priv = torch.cat([obs, torch.zeros(self.num_envs, 2)], 1)
        return State(data=None, obs={"state": obs, "privileged_state": priv}, reward=reward, done=done,
                     metrics={"reward/track": reward.clone()}, info={"truncation": torch.zeros(self.num_envs)})

    def reset(self, seed):
        self.resets += 1
        self.t = torch.zeros(self.num_envs)
        return self._state(torch.zeros(self.num_envs), torch.zeros(self.num_envs))

    def step(self, state, action):
        reward = 1.0 - (action[:, 0] - self.target).abs()
        self.t += 1
        done = (self.t >= 10).float()
        self.t = self.t * (1 - done)
        return self._state(reward, done)


def test_train_loop_epochs_evaluator_and_callbacks(tmp_path):
    env = _ToyEnv(64)
    seen, saved = [], []
    net, metrics = T.train(env, num_timesteps=64 * 5 * 12, progress_fn=lambda s, m: seen.append((s, dict(m))),
                           policy_params_fn=lambda s, n: saved.append(s), seed=0, log_path=str(tmp_path / "m.jsonl"),
                           num_evals=4, unroll_length=5, num_minibatches=4, num_updates_per_batch=2, episode_length=20, num_eval_envs=16,
                           learning_rate=3e-3, network_factory=dict(policy_hidden_layer_sizes=(32, 32), value_hidden_layer_sizes=(32, 32)))
    # brax epoch structure: 1 initial evaluation + (num_evals - 1) epochs of ceil(3840 / (3 * 320)) = 4 training steps
    assert [s for s, _ in seen] == [0, 1280, 2560, 3840] and saved == [1280, 2560, 3840]
    assert env.resets == 1 + 3                                  # initial reset + num_resets_per_eval = 1 per epoch
    first, last = seen[0][1], seen[-1][1]
    for k in ("eval/episode_reward", "eval/episode_reward_std", "eval/episode_reward/track", "eval/avg_episode_length", "eval/sps"):
        assert k in first and k in last
    assert first["eval/avg_episode_length"] == 10.0              # EvalWrapper: only the first episode of each env counts
    assert abs(first["eval/episode_reward"] - first["eval/episode_reward/track"]) < 1e-5
    assert "training/sps" in last and "training/total_loss" in last and "training/sps" not in first
    assert last["eval/episode_reward"] > first["eval/episode_reward"]       # it learns to track
    import json
    lines = [json.loads(l) for l in open(tmp_path / "m.jsonl")]
    assert [l["step"] for l in lines] == [0, 1280, 2560, 3840]


def _train_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = _ToyEnv(32, seed=10 + rank)                             # every rank steps its own env shard
    seen = []
    net, metrics = T.train(env, num_timesteps=2 * 32 * 5 * 6, progress_fn=lambda s, m: seen.append((s, dict(m))), seed=0, num_evals=3,
                           unroll_length=5, num_minibatches=4, num_updates_per_batch=2, episode_length=20, num_eval_envs=8, learning_rate=3e-3,
                           network_factory=dict(policy_hidden_layer_sizes=(16,), value_hidden_layer_sizes=(16,)))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [net.norm_obs.mean, net.norm_obs.std])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out.put((rank, [s for s, _ in seen], bool(torch.equal(gathered[0], gathered[1])), float(net.norm_obs.count),
             sorted(k for k in metrics if k.startswith("training/"))))
    dist.destroy_process_group()


def test_train_loop_two_ranks_gloo():
    """The whole training loop under data parallelism (what `torchrun ... runner` runs per GPU, with gloo instead of RCCL):
    env shards per rank, all-reduced gradients / normaliser moments / loss means, evaluation and callbacks on rank 0 only,
    and the per-epoch replica check (train.assert_replicas_identical) passing on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs: p.join(timeout=60)
    (r0, steps0, same0, count0, keys0), (r1, steps1, same1, count1, keys1) = res
    assert (r0, r1) == (0, 1) and same0 and same1
    assert steps0 == [0, 960, 1920] and steps1 == []              # progress / evaluation on rank 0 only; env steps count both shards
    assert count0 == count1 == 1920                               # normaliser statistics cover both ranks' rollouts
    assert "training/total_loss" in keys0 and "training/sps" in keys0


def test_loss_meter_and_sgd_epoch_report_means_over_all_steps():
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(0)
    net = PPONetworks(101, 212, 14, (32, 16), (32, 16))
    data = _fake_rollout(32, 6, gen)
    cfg = T.ppo_config(); cfg.update(num_minibatches=4, num_updates_per_batch=2)
    opt = torch.optim.Adam(net.parameters(), 1e-3)
    meter = T.LossMeter()
    assert T.sgd_epoch(net, opt, data, cfg, gen, meter=meter) is None and meter.n == 8     # sums keep running in the meter
    T.sgd_epoch(net, opt, data, cfg, gen, meter=meter)
    assert meter.n == 16
    m = meter.mean()
    assert set(m) == set(T.LossMeter.KEYS) and meter.n == 0 and meter.mean() == {}
    torch.testing.assert_close(m["total_loss"], m["policy_loss"] + m["v_loss"] + m["entropy_loss"], rtol=1e-5, atol=1e-6)
    own = T.sgd_epoch(net, opt, data, cfg, gen)                 # without a meter: the mean over this call's 8 steps
    assert set(own) == set(T.LossMeter.KEYS)


def test_every_evaluation_gets_its_own_seed_and_the_eval_envs_their_own_randomisation():
    env = _ToyEnv(32)
    seeds, streams = [], []
    ev = _ToyEnv(8, seed=99)
    real_reset = ev.reset
    ev.reset = lambda seed: (seeds.append(seed), real_reset(seed))[1]
    T.train(env, num_timesteps=32 * 5 * 6, seed=11, eval_env=None, num_evals=4, unroll_length=5, num_minibatches=4, num_updates_per_batch=1,
            episode_length=10, num_eval_envs=8, randomization_fn=lambda e, stream: streams.append((e.num_envs, stream)),
            network_factory=dict(policy_hidden_layer_sizes=(8,), value_hidden_layer_sizes=(8,)))
    assert streams == [(32, 0), (8, 1)]                         # training envs: stream 0, evaluation envs: stream 1
    env2 = _ToyEnv(32)
    T.train(env2, num_timesteps=32 * 5 * 6, seed=11, eval_env=ev, num_evals=4, unroll_length=5, num_minibatches=4, num_updates_per_batch=1,
            episode_length=10, num_eval_envs=8, network_factory=dict(policy_hidden_layer_sizes=(8,), value_hidden_layer_sizes=(8,)))
    assert seeds == [12, 13, 14, 15]                            # seed + 1 + evaluation index


def test_tensorboard_writer_round_trip(tmp_path):
    from open_duck_playground_amd.tb_writer import SummaryWriter, crc32c, read_scalars
    assert crc32c(b"123456789") == 0xE3069283                   # CRC-32C check value
    w = SummaryWriter(str(tmp_path))
    w.add_scalar("eval/episode_reward", 3.5, 100); w.add_scalar("training/sps", 1.0e6, 163840); w.close()
    assert list(read_scalars(w.path)) == [("eval/episode_reward", 100, 3.5), ("training/sps", 163840, 1.0e6)]
