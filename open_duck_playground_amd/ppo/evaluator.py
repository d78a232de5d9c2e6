"""brax `acting.Evaluator` + `EvalWrapper` semantics for the batched env (the reference gets them from
brax ppo.train, common/runner.py:104-118; its progress_fn prints `eval/episode_reward` and
`eval/episode_reward_std`, common/runner.py:62-65).

One evaluation = reset `num_eval_envs` envs, run the DETERMINISTIC policy (action = tanh(loc)) for
`episode_length` steps, and sum reward / metrics over each env's FIRST episode only
(`episode_metrics += metrics * active; active *= 1 - done`)."""
from __future__ import annotations

import time
from typing import Dict

import torch


class Evaluator:
    def __init__(self, eval_env, episode_length: int, action_repeat: int = 1):
        self.env, self.episode_length, self.action_repeat = eval_env, int(episode_length), int(action_repeat)
        self._steps_per_unroll = self.episode_length * eval_env.num_envs
        self._eval_walltime = 0.0

    @torch.no_grad()
    def run_evaluation(self, net, training_metrics: Dict[str, float], seed: int = 0, aggregate_episodes: bool = True) -> Dict[str, float]:
        t0 = time.time()
        state = self.env.reset(seed)
        n = self.env.num_envs
        dev = state.reward.device
        active = torch.ones(n, device=dev)
        sums = {"reward": torch.zeros(n, device=dev), **{k: torch.zeros(n, device=dev) for k in state.metrics}}
        steps = torch.zeros(n, device=dev)
        for _ in range(self.episode_length // self.action_repeat):
            loc, _ = net.dist_params(state.obs["state"])
            state = self.env.step(state, torch.tanh(loc).contiguous())
            sums["reward"] += state.reward * active
            for k, v in state.metrics.items():
                sums[k] += v * active
            steps += active
            active = active * (1.0 - state.done)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dt = time.time() - t0
        self._eval_walltime += dt
        out = {}
        for name, v in sums.items():
            if aggregate_episodes:
                out[f"eval/episode_{name}"] = float(v.mean())
                out[f"eval/episode_{name}_std"] = float(v.std(unbiased=False))
            else:
                out[f"eval/episode_{name}"] = v.cpu().numpy()
        out["eval/avg_episode_length"] = float(steps.mean())
        out["eval/std_episode_length"] = float(steps.std(unbiased=False))
        out["eval/epoch_eval_time"] = dt
        out["eval/sps"] = self._steps_per_unroll / dt
        out["eval/walltime"] = self._eval_walltime
        return {**out, **training_metrics}
