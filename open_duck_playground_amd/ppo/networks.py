"""Policy / value networks and the tanh-normal action distribution (brax ppo.networks counterpart;
architecture corroborated by reference playground/common/export_onnx.py:71-102: swish MLPs, policy
output split into (loc, scale), deterministic action = tanh(loc))."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


_ENGINE = []


def _engine():
    """the engine module when the HIP library is there (GPU rollouts), else None -- the torch path below is the CPU / fallback arithmetic"""
    if not _ENGINE:
        try:
            from .. import engine
            engine.load_library()
            _ENGINE.append(engine)
        except Exception:
            _ENGINE.append(None)
    return _ENGINE[0]


class MLP(nn.Module):
    def __init__(self, sizes):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])])
        for lin in self.layers:  # lecun_uniform, zero bias (brax / flax default for these nets)
            bound = math.sqrt(3.0 / lin.in_features)
            nn.init.uniform_(lin.weight, -bound, bound)
            nn.init.zeros_(lin.bias)

    def forward(self, x):
        for lin in self.layers[:-1]:
            x = F.silu(lin(x))
        return self.layers[-1](x)


class RunningStats(nn.Module):
    """brax running_statistics: per-feature mean / std from (count, mean, summed_variance); buffers are
    replicated across ranks (batch moments are all-reduced before the update)."""

    def __init__(self, size: int, std_min: float = 1e-6, std_max: float = 1e6):
        super().__init__()
        self.register_buffer("count", torch.zeros((), dtype=torch.float64))
        self.register_buffer("mean", torch.zeros(size))
        self.register_buffer("summed_variance", torch.zeros(size))
        self.register_buffer("std", torch.ones(size))
        self.std_min, self.std_max = std_min, std_max

    @torch.no_grad()
    def update(self, batch: torch.Tensor, group=None):
        x = batch.reshape(-1, batch.shape[-1]).to(torch.float32)
        eng = _engine() if (x.is_cuda and x.is_contiguous() and x.shape[0] >= 4096) else None
        if eng is not None and group is None:
            # single rank: the whole update on the device -- one pass over the rollout with float64 accumulators, then the running statistics
            # in float64 (csrc/odk_learner.hip col_moments_kernel / moments_update_kernel): 0.23 ms of ~25 tiny launches and a host-to-device
            # copy -> three launches
            eng.running_stats_update(x, self.count, self.mean, self.summed_variance, self.std, self.std_min, self.std_max)
            return
        n = torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device)
        # float32 sums over chunks of <= 256 rows (one fused read each, no float64 copy of the 10^7-element rollout), the
        # chunk sums folded in float64: each chunk sum / norm carries float32 rounding (~1e-7 relative; the squared norm
        # ~2e-7), which is far below what the normaliser needs but NOT double accuracy; 2.2 ms -> 0.3 ms per update
        rows = x.shape[0]
        chunk = next((c for c in (256, 128, 64, 32) if rows % c == 0), 0)
        if eng is not None:
            s, s2 = eng.col_moments(x)
        elif chunk and rows > chunk:
            xv = x.view(rows // chunk, chunk, x.shape[1])
            s = xv.sum(1).double().sum(0)
            s2 = torch.linalg.vector_norm(xv, ord=2, dim=1).double().square().sum(0)
        else:
            s, s2 = x.sum(0).double(), (x.double() ** 2).sum(0)
        if group is not None:
            import torch.distributed as dist
            packed = torch.cat([n, s, s2])
            dist.all_reduce(packed, group=group)
            n, s, s2 = packed[:1], packed[1:1 + s.numel()], packed[1 + s.numel():]
        count = self.count + n[0]
        bmean = s / n[0]
        delta = bmean - self.mean.double()
        new_mean = self.mean.double() + delta * (n[0] / count)
        # summed_variance += sum (x - old_mean)(x - new_mean)
        sv = self.summed_variance.double() + (s2 - s * (self.mean.double() + new_mean) + n[0] * self.mean.double() * new_mean)
        self.count.copy_(count); self.mean.copy_(new_mean.float()); self.summed_variance.copy_(sv.float())
        self.std.copy_(torch.sqrt(torch.clamp(sv / count, min=0)).float().clamp(self.std_min, self.std_max))

    def forward(self, x, out=None):
        if out is None:
            return (x - self.mean) / self.std
        torch.sub(x, self.mean, out=out)        # the same two roundings, written into the caller's buffer
        return out.div_(self.std)


class PPONetworks(nn.Module):
    def __init__(self, obs_size: int, priv_size: int, action_size: int, policy_hidden=(512, 256, 128), value_hidden=(512, 256, 128)):
        super().__init__()
        self.action_size = action_size
        self.policy = MLP([obs_size, *policy_hidden, 2 * action_size])
        self.value = MLP([priv_size, *value_hidden, 1])
        self.norm_obs = RunningStats(obs_size)
        self.norm_priv = RunningStats(priv_size)

    def dist_params(self, obs):
        out = self.policy(self.norm_obs(obs))
        loc, raw = out[..., : self.action_size], out[..., self.action_size:]
        return loc, F.softplus(raw) + 0.001

    def values(self, priv):
        return self.value(self.norm_priv(priv)).squeeze(-1)


LOG2 = math.log(2.0)


def tanh_log_det_jac(x):
    return 2.0 * (LOG2 - x - F.softplus(-2.0 * x))


def tanh_normal_log_prob(loc, scale, raw_action):
    lp = -0.5 * ((raw_action - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)
    return (lp - tanh_log_det_jac(raw_action)).sum(-1)


def tanh_normal_entropy(loc, scale, sample):
    ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(scale)
    return (ent + tanh_log_det_jac(sample)).sum(-1)
