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
    T.sgd_epoch(net, opt, data, cfg, torch.Generator().manual_seed(7), world=world)
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
