"""PPO on PyTorch-ROCm with brax semantics (brax.training.agents.ppo.train counterpart, reached by the
reference through playground/common/runner.py:104-118).  Hyper-parameters: the BerkeleyHumanoid table the
reference asks for (common/runner.py:87-89; values in SURVEY.md Appendix G, [UPSTREAM-MEMORY]).

Data parallelism: one process per GPU, envs sharded, parameters replicated; per SGD step one all-reduce
(mean) of the flat gradient buffer and, per training step, one all-reduce of the normaliser moments
(RCCL over xGMI when launched with torchrun on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import json
import os
import time
import weakref
from typing import Callable, Dict, Optional

import torch

from .networks import PPONetworks, tanh_normal_entropy, tanh_normal_log_prob


def ppo_config() -> Dict:
    """locomotion_params.brax_ppo_config("BerkeleyHumanoidJoystickFlatTerrain")."""
    return dict(num_timesteps=150_000_000, num_evals=15, reward_scaling=1.0, episode_length=1000, normalize_observations=True,
                action_repeat=1, unroll_length=20, num_minibatches=32, num_updates_per_batch=4, discounting=0.97, learning_rate=3e-4,
                entropy_cost=0.005, num_envs=8192, batch_size=256, max_grad_norm=1.0, clipping_epsilon=0.2, num_resets_per_eval=1,
                gae_lambda=0.95, normalize_advantage=True,
                network_factory=dict(policy_hidden_layer_sizes=(512, 256, 128), value_hidden_layer_sizes=(512, 256, 128),
                                     policy_obs_key="state", value_obs_key="privileged_state"))


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_, discount):
    """brax ppo.losses.compute_gae: time-major [T, B] tensors."""
    T = rewards.shape[0]
    trunc_mask = 1.0 - truncation
    values_t1 = torch.cat([values[1:], bootstrap_value[None]], 0)
    deltas = (rewards + discount * (1.0 - termination) * values_t1 - values) * trunc_mask
    acc = torch.zeros_like(bootstrap_value)
    vs_minus_v = []
    for t in range(T - 1, -1, -1):
        acc = deltas[t] + discount * (1.0 - termination[t]) * trunc_mask[t] * lambda_ * acc
        vs_minus_v.append(acc)
    vs_minus_v = torch.stack(vs_minus_v[::-1], 0)
    vs = vs_minus_v + values
    vs_t1 = torch.cat([vs[1:], bootstrap_value[None]], 0)
    advantages = (rewards + discount * (1.0 - termination) * vs_t1 - values) * trunc_mask
    return vs.detach(), advantages.detach()


def ppo_loss(net: PPONetworks, mb: Dict[str, torch.Tensor], cfg: Dict):
    """brax ppo.losses.compute_ppo_loss on a minibatch of trajectories ([B, T, ...] tensors).  Autograd reference
    of the fused GPU step in ppo/learner.py (and the CPU path)."""
    obs, priv = mb["obs"], mb["priv"]
    loc, scale = net.dist_params(obs)
    baseline = net.values(priv)
    bootstrap = net.values(mb["last_priv"])
    rewards = mb["reward"] * cfg["reward_scaling"]
    termination = mb["done"] * (1.0 - mb["truncation"])
    tm = lambda x: x.transpose(0, 1)
    vs, adv = compute_gae(tm(mb["truncation"]), tm(termination), tm(rewards), tm(baseline.detach()), bootstrap.detach(),
                          cfg["gae_lambda"], cfg["discounting"])
    vs, adv = tm(vs), tm(adv)
    if cfg["normalize_advantage"]:
        adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)   # jnp.std: ddof 0
    logp = tanh_normal_log_prob(loc, scale, mb["raw_action"])
    rho = torch.exp(logp - mb["log_prob"])
    eps = cfg["clipping_epsilon"]
    policy_loss = -torch.min(rho * adv, rho.clamp(1 - eps, 1 + eps) * adv).mean()
    v_loss = ((vs - baseline) ** 2).mean() * 0.5 * 0.5
    noise = mb["noise"] if "noise" in mb else torch.randn_like(loc)   # tests inject the entropy sample
    entropy = tanh_normal_entropy(loc, scale, loc + scale * noise).mean()
    ent_loss = -cfg["entropy_cost"] * entropy
    total = policy_loss + v_loss + ent_loss
    return total, dict(total_loss=total.detach(), policy_loss=policy_loss.detach(), v_loss=v_loss.detach(), entropy_loss=ent_loss.detach())


def _allreduce_grads(params, world: int, group=None):
    import torch.distributed as dist
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, group=group)   # RCCL (backend "nccl") on GPUs, gloo on CPU
    flat /= world
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


def clip_by_global_norm(params, max_norm: float):
    """optax.clip_by_global_norm: g *= max_norm / ||g|| when ||g|| >= max_norm."""
    grads = [p.grad for p in params if p.grad is not None]
    norm = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.where(norm < max_norm, torch.ones_like(norm), max_norm / norm)
    for g in grads:
        g.mul_(coef)
    return norm


class LossMeter:
    """Mean of the loss scalars over every SGD step since the last reset (brax returns the metrics of all
    num_updates_per_batch x num_minibatches steps of every training step of the epoch and averages them; with several
    ranks they are pmean'ed as well)."""
    KEYS = ("total_loss", "policy_loss", "v_loss", "entropy_loss")

    def __init__(self):
        self.sum, self.n = None, 0

    def add(self, metrics: Dict[str, torch.Tensor]):
        v = torch.stack([metrics[k].detach().float() for k in self.KEYS])
        self.sum = v if self.sum is None else self.sum + v
        self.n += 1

    def mean(self, world: int = 1, group=None, reset: bool = True) -> Dict[str, torch.Tensor]:
        if self.sum is None:
            return {}
        v = self.sum / float(self.n)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(v, group=group)
            v = v / world
        if reset:
            self.sum, self.n = None, 0
        return {k: v[i] for i, k in enumerate(self.KEYS)}


def sgd_epoch(net, opt, data: Dict[str, torch.Tensor], cfg: Dict, gen: torch.Generator, world: int = 1, group=None, learner=None,
              meter: Optional[LossMeter] = None):
    """num_updates_per_batch x num_minibatches clipped-Adam steps over one rollout ([B, T, ...] per rank).
    With a `learner` (ppo.learner.FlatLearner, GPU) every step is one HIP-graph replay; otherwise autograd.
    Returns the mean losses over this call's steps (averaged over the ranks).  With a `meter` nothing is reduced or returned:
    the sums keep running (in the meter, or in the learner) until the caller asks for the mean -- `train` does so once per
    epoch, like brax."""
    B = data["reward"].shape[0]
    nmb = cfg["num_minibatches"]
    if learner is not None:
        from .learner import prepare_rollout
        learner.sync_weights()                        # (parameters written from outside since the last step, e.g. a restored checkpoint)
        nup = cfg["num_updates_per_batch"]
        if getattr(learner, "indexed", False) and nup * nmb <= learner.steps_cap:
            # the rollout goes into the learner's resident buffers once, the training step's shuffles become ONE device array of
            # trajectory indices, and every minibatch step is one graph replay -- no gather, no host-side call between two steps
            learner.load_rollout_from(net, data, cfg)
            # the training step's shuffles as ONE batched sort of uniform keys (four torch.randperm calls were 0.18 ms of sort launches)
            perms = torch.rand(nup, B, generator=gen, device=data["reward"].device).argsort(dim=1)
            learner.set_schedule(perms.reshape(-1))
            learner.run(nup * nmb)
            return learner.metrics() if meter is None else None
        prep = prepare_rollout(net, data, cfg)
        for _ in range(nup):
            perm = torch.randperm(B, generator=gen, device=data["reward"].device)
            for mbi in perm.chunk(nmb):
                learner.load_minibatch(prep, mbi)
                learner.step()
        return learner.metrics() if meter is None else None
    params = [p for p in net.parameters() if p.requires_grad]
    own = meter is None
    meter = LossMeter() if own else meter
    for _ in range(cfg["num_updates_per_batch"]):
        perm = torch.randperm(B, generator=gen, device=data["reward"].device)
        for mbi in perm.chunk(nmb):
            mb = {k: v[mbi] for k, v in data.items()}
            loss, metrics = ppo_loss(net, mb, cfg)
            opt.zero_grad(set_to_none=False)
            loss.backward()
            if world > 1:
                _allreduce_grads(params, world, group)
            if cfg.get("max_grad_norm"):
                clip_by_global_norm(params, cfg["max_grad_norm"])
            opt.step()
            meter.add(metrics)
    return meter.mean(world, group) if own else None


def make_learner(net, data, cfg, world: int = 1, group=None, **learner_kw):
    """FlatLearner for CUDA rollouts whose trajectory count divides into the minibatches; None (autograd path) otherwise.
    `learner_kw`: FlatLearner options (`split_update`, `capture_allreduce`, `fused_norm`)."""
    B = data["reward"].shape[0]
    if not data["reward"].is_cuda or B % cfg["num_minibatches"] != 0 or not cfg.get("use_graphs", True):
        return None
    from .learner import FlatLearner
    return FlatLearner(net, cfg, B // cfg["num_minibatches"], data["reward"].shape[1], world, group,
                       use_graph=os.environ.get("ODK_LEARNER_GRAPH", "1") != "0", **learner_kw)


class _RolloutBuffers:
    """Time-major history of one unroll of an engine-backed env ([T, N, ...] per field) and the per-step snapshot launches: the
    env's output tensors are persistent, so step t's five snapshots (reward / done / truncation of step t, the observations
    step t + 1 starts from) are ONE launch with fixed addresses instead of five `clone`s."""

    def __init__(self, state, T: int, A: int):
        from .. import engine
        obs, priv = state.obs["state"], state.obs["privileged_state"]
        N, dev = obs.shape[0], obs.device
        E = lambda *s: torch.empty(*s, device=dev)
        self.T, self.src = T, (obs.data_ptr(), priv.data_ptr(), state.reward.data_ptr(), state.done.data_ptr(), state.info["truncation"].data_ptr())
        self.shape = (N, obs.shape[1], priv.shape[1], A)
        self.buf = dict(obs=E(T, N, obs.shape[1]), priv=E(T, N, priv.shape[1]), raw_action=E(T, N, A), log_prob=E(T, N), reward=E(T, N), done=E(T, N),
                        truncation=E(T, N))
        self.action = E(N, A)
        b = self.buf
        self.first = engine.MultiCopy([(obs, b["obs"][0]), (priv, b["priv"][0])])
        self.after = []
        for t in range(T):
            pairs = [(state.reward, b["reward"][t]), (state.done, b["done"][t]), (state.info["truncation"], b["truncation"][t])]
            if t + 1 < T:
                pairs += [(obs, b["obs"][t + 1]), (priv, b["priv"][t + 1])]
            self.after.append(engine.MultiCopy(pairs))

    def matches(self, state, T: int, A: int) -> bool:
        obs, priv = state.obs["state"], state.obs["privileged_state"]
        return (T == self.T and self.shape == (obs.shape[0], obs.shape[1], priv.shape[1], A)
                and self.src == (obs.data_ptr(), priv.data_ptr(), state.reward.data_ptr(), state.done.data_ptr(), state.info["truncation"].data_ptr()))


# Rollout history buffers per network, held weakly OUTSIDE the module: ~220 MB of [T, N, ...] history and ctypes descriptor arrays
# hidden in net.__dict__ would ride along with copy.deepcopy(net) / torch.save(net) -- and the ctypes pointers cannot be pickled.
_ROLLOUT_BUFFERS = weakref.WeakKeyDictionary()


@torch.no_grad()
def _rollout_engine(env, net: PPONetworks, state, unroll_length: int, gen: torch.Generator, deterministic: bool):
    """`rollout` for an engine-backed env on the GPU: per step the policy's whole-network launch (when the architecture allows),
    the sampling launch writing straight into the history, the fused env-step launch and one snapshot launch."""
    from .. import engine
    from .learner import fused_policy
    T, A = unroll_length, net.action_size
    rb = _ROLLOUT_BUFFERS.get(net)
    if rb is None or not rb.matches(state, T, A):
        rb = _ROLLOUT_BUFFERS[net] = _RolloutBuffers(state, T, A)
    N = state.obs["state"].shape[0]
    fp = fused_policy(net, N)
    if fp is not None:
        fp.refresh()
    noise = torch.zeros(T, N, A, device=state.reward.device) if deterministic else torch.randn(T, N, A, generator=gen, device=state.reward.device)
    b = rb.buf
    rb.first()
    for t in range(T):
        logits = fp(state.obs["state"]) if fp is not None else net.policy(net.norm_obs(state.obs["state"]))
        engine.policy_sample(logits, noise[t], out=(b["raw_action"][t], rb.action, b["log_prob"][t]))
        state = env.step(state, rb.action)
        rb.after[t]()
    data = {k: v.transpose(0, 1).contiguous() for k, v in b.items()}
    data["last_priv"] = state.obs["privileged_state"].clone()
    return data, state


@torch.no_grad()
def rollout(env, net: PPONetworks, state, unroll_length: int, gen: torch.Generator, deterministic: bool = False, engine_path: bool = True):
    """brax acting.generate_unroll: returns ([B, T, ...] transition tensors, final state).  `engine_path` false forces the
    generic loop below for an engine-backed env too (tests compare the two)."""
    if engine_path and state.obs["state"].is_cuda and hasattr(env, "batch") and "truncation" in state.info:
        return _rollout_engine(env, net, state, unroll_length, gen, deterministic)
    keys = ("obs", "priv", "raw_action", "log_prob", "reward", "done", "truncation")
    buf = {k: [] for k in keys}
    for _ in range(unroll_length):
        obs, priv = state.obs["state"].clone(), state.obs["privileged_state"].clone()
        loc, scale = net.dist_params(obs)
        raw = loc if deterministic else loc + scale * torch.randn(loc.shape, generator=gen, device=loc.device)
        logp = tanh_normal_log_prob(loc, scale, raw)
        action = torch.tanh(raw).contiguous()
        state = env.step(state, action)
        buf["obs"].append(obs); buf["priv"].append(priv); buf["raw_action"].append(raw); buf["log_prob"].append(logp)
        buf["reward"].append(state.reward.clone()); buf["done"].append(state.done.clone()); buf["truncation"].append(state.info["truncation"].clone())
    data = {k: torch.stack(v, 1) for k, v in buf.items()}
    data["last_priv"] = state.obs["privileged_state"].clone()
    return data, state


def train(environment, num_timesteps: int, progress_fn: Optional[Callable] = None, policy_params_fn: Optional[Callable] = None,
          restore_checkpoint_path: Optional[str] = None, seed: int = 0, randomization_fn: Optional[Callable] = None,
          log_path: Optional[str] = None, eval_env=None, **overrides):
    """Trains on `environment` (a batched Joystick / Standing).  Returns (networks, metrics).

    Epoch structure of brax ppo.train: `num_evals - 1` epochs of `ceil(num_timesteps / (epochs * env_steps_per_iter *
    num_resets_per_eval))` training steps, the env re-reset `num_resets_per_eval` times per epoch, one evaluation
    (Evaluator, deterministic policy, `num_eval_envs` envs) before training and after every epoch, then
    `progress_fn(env_steps, metrics)` and `policy_params_fn(env_steps, networks)` on rank 0."""
    import torch.distributed as dist
    cfg = ppo_config()
    cfg.update({k: v for k, v in overrides.items() if v is not None})
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = environment.batch.obs.device if hasattr(environment, "batch") else environment.device
    nf = cfg["network_factory"]
    torch.manual_seed(seed)   # identical initial parameters on every rank
    net = PPONetworks(environment.observation_size["state"][0], environment.observation_size["privileged_state"][0], environment.action_size,
                      nf["policy_hidden_layer_sizes"], nf["value_hidden_layer_sizes"]).to(dev)
    if restore_checkpoint_path:
        net.load_state_dict(torch.load(restore_checkpoint_path, map_location=dev)["networks"])
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=cfg["learning_rate"], capturable=dev.type == "cuda")
    gen = torch.Generator(device=dev); gen.manual_seed(seed * 1000 + rank)
    learner = None
    if randomization_fn is not None:
        _randomize(randomization_fn, environment, 0)
    n_local = environment.num_envs
    steps_per_iter = n_local * world * cfg["unroll_length"] * cfg["action_repeat"]
    num_evals_after_init = max(cfg["num_evals"] - 1, 1)
    resets = max(cfg["num_resets_per_eval"], 1)
    iters_per_epoch = max(1, -(-num_timesteps // (num_evals_after_init * steps_per_iter * resets)))
    evaluator = None
    if cfg.get("num_eval_envs", 128) and rank == 0:
        if eval_env is None and hasattr(environment, "make_eval_env"):
            eval_env = environment.make_eval_env(cfg.get("num_eval_envs", 128))
            if randomization_fn is not None:
                _randomize(randomization_fn, eval_env, 1)    # its own draws, not a copy of the first training envs'
        if eval_env is not None:
            from .evaluator import Evaluator
            evaluator = Evaluator(eval_env, cfg["episode_length"], cfg["action_repeat"])
    if dev.type == "cuda" and cfg.get("tune_gemms", True):
        from .learner import tune_inference_shapes
        tune_inference_shapes(net, [n_local] + ([cfg.get("num_eval_envs", 128)] if evaluator is not None else []))
    log = open(log_path, "a") if (log_path and rank == 0) else None
    t0 = time.time(); done_steps = 0; metrics = {}

    eval_count = 0

    def report(training_metrics):
        nonlocal eval_count
        # brax's Evaluator splits its key on every run: each evaluation gets fresh initial states, commands and pushes
        m = evaluator.run_evaluation(net, training_metrics, seed=seed + 1 + eval_count) if evaluator is not None else dict(training_metrics)
        eval_count += 1
        if rank == 0:
            if log:
                log.write(json.dumps({"step": done_steps, **m}) + "\n"); log.flush()
            if progress_fn:
                progress_fn(done_steps, m)
        return m

    if cfg["num_evals"] > 1 and evaluator is not None:
        metrics = report({})
    state = environment.reset(seed)
    reset_count = 0
    meter = LossMeter()
    grp = dist.group.WORLD if world > 1 else None
    for epoch in range(num_evals_after_init):
        t_epoch = time.time()
        for _ in range(resets):
            for _ in range(iters_per_epoch):
                data, state = rollout(environment, net, state, cfg["unroll_length"], gen)
                if cfg["normalize_observations"]:
                    net.norm_obs.update(data["obs"], grp); net.norm_priv.update(data["priv"], grp)
                if learner is None and done_steps == 0:
                    learner = make_learner(net, data, cfg, world)
                sgd_epoch(net, opt, data, cfg, gen, world, grp, learner=learner, meter=meter)
                done_steps += steps_per_iter
            if cfg["num_resets_per_eval"] > 0:
                reset_count += 1
                state = environment.reset(seed + 7919 * reset_count)
        loss_metrics = learner.metrics() if learner is not None else meter.mean(world, grp)
        if world > 1:
            assert_replicas_identical(net, grp)
        ep_rew = (data["reward"].sum(1)).mean()
        m = torch.stack([ep_rew, data["done"].mean()])
        if world > 1:
            dist.all_reduce(m); m /= world
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        epoch_time = time.time() - t_epoch
        training_metrics = {"training/sps": iters_per_epoch * resets * steps_per_iter / epoch_time, "training/walltime": time.time() - t0,
                            "training/unroll_reward": float(m[0]), "training/done_rate": float(m[1]),
                            **{f"training/{k}": float(v) for k, v in loss_metrics.items()}}
        metrics = report(training_metrics)
        if rank == 0 and policy_params_fn:
            policy_params_fn(done_steps, net)
    if log:
        log.close()
    if learner is not None:
        learner.close()
    return net, metrics


def _randomize(randomization_fn: Callable, env, stream: int):
    """`randomization_fn(env, stream)`: stream 0 = the training envs, 1 = the evaluation envs (distinct draws).  One-argument
    callables are still accepted."""
    import inspect
    try:
        two = len(inspect.signature(randomization_fn).parameters) >= 2
    except (TypeError, ValueError):
        two = False
    return randomization_fn(env, stream) if two else randomization_fn(env)


@torch.no_grad()
def assert_replicas_identical(net: PPONetworks, group=None):
    """Data-parallel invariant: parameters and normaliser statistics are BIT-identical on every rank (same initial values,
    all-reduced gradients and moments, fixed-order reductions).  Two all-reduces (min / max of the bit patterns) of ~2 MB
    once per epoch; raises on the first divergence instead of training replicas that silently drift apart."""
    import torch.distributed as dist
    flat = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + [b for b in net.buffers() if b.dtype == torch.float32]])
    bits = flat.view(torch.int32)
    lo, hi = bits.clone(), bits.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    bad = int((lo != hi).sum())
    if bad:
        raise RuntimeError(f"data-parallel replicas diverged: {bad} of {bits.numel()} parameter / normaliser words differ across ranks")


def save_checkpoint(path: str, net: PPONetworks):
    """(normalizer, policy, value) triple like the reference's orbax checkpoint (common/runner.py:68-76)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"networks": net.state_dict()}, path)
