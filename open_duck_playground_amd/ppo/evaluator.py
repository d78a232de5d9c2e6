"""brax `acting.Evaluator` + `EvalWrapper` semantics for the batched env (the reference gets them from
brax ppo.train, common/runner.py:104-118; its progress_fn prints `eval/episode_reward` and
`eval/episode_reward_std`, common/runner.py:62-65).

One evaluation = reset `num_eval_envs` envs, run the DETERMINISTIC policy (action = tanh(loc)) for
`episode_length` steps, and sum reward / metrics over each env's FIRST episode only
(`episode_metrics += metrics * active; active *= 1 - done`).

On the GPU one evaluation step (policy inference, the fused env-step launch, the accumulator updates: ~45 small
launches) is captured once as a HIP graph and replayed `episode_length` times: with 128 envs the step is launch
bound, not compute bound."""
from __future__ import annotations

import time
from typing import Dict

import torch


class Evaluator:
    def __init__(self, eval_env, episode_length: int, action_repeat: int = 1, use_graph: bool = True):
        self.env, self.episode_length, self.action_repeat = eval_env, int(episode_length), int(action_repeat)
        self.use_graph = use_graph
        self._steps_per_unroll = self.episode_length * eval_env.num_envs
        self._eval_walltime = 0.0
        self._acc = None              # persistent accumulators (the captured graph updates them in place)
        self._graph = None
        self._graph_key = None
        self._fp = None

    def _one_step(self, net):
        st = self._state
        # all four policy layers in one launch when the architecture allows (csrc/odk_mlp.hip, inference mode); the mode needs
        # no scale (softplus skipped)
        logits = self._fp(st.obs["state"]) if self._fp is not None else net.policy(net.norm_obs(st.obs["state"]))
        loc = logits[..., : net.action_size]
        st = self.env.step(st, torch.tanh(loc).contiguous())
        a = self._acc
        if a["matrix"] is not None:
            # the engine hands the metrics over as one [N, K] tensor: 4 launches per step for the bookkeeping instead of
            # 2 per metric (the evaluation step is launch-bound)
            a["matrix"].addcmul_(st.data.metrics, a["active"].unsqueeze(1))
            a["sums"]["reward"].addcmul_(st.reward, a["active"])
        else:
            a["sums"]["reward"] += st.reward * a["active"]
            for k, v in st.metrics.items():
                a["sums"][k] += v * a["active"]
        a["steps"] += a["active"]
        a["active"].addcmul_(a["active"], st.done, value=-1.0)   # active *= 1 - done
        self._state = st

    @torch.no_grad()
    def run_evaluation(self, net, training_metrics: Dict[str, float], seed: int = 0, aggregate_episodes: bool = True) -> Dict[str, float]:
        t0 = time.time()
        self._state = self.env.reset(seed)
        n = self.env.num_envs
        dev = self._state.reward.device
        if self._acc is None:
            z = lambda: torch.zeros(n, device=dev)
            m = getattr(getattr(self._state, "data", None), "metrics", None)
            names = getattr(self.env, "METRIC_NAMES", None)
            if torch.is_tensor(m) and m.dim() == 2 and names is not None and len(names) == m.shape[1]:
                matrix = torch.zeros_like(m)     # column i accumulates METRIC_NAMES[i]; the dict entries are views of it
                sums = {"reward": z(), **{nm: matrix[:, i] for i, nm in enumerate(names) if nm is not None}}
            else:
                matrix, sums = None, {"reward": z(), **{k: z() for k in self._state.metrics}}
            self._acc = dict(active=z(), steps=z(), sums=sums, matrix=matrix)
        a = self._acc
        a["active"].fill_(1.0); a["steps"].zero_()
        if a["matrix"] is not None:
            a["matrix"].zero_(); a["sums"]["reward"].zero_()
        else:
            for v in a["sums"].values():
                v.zero_()
        nsteps = self.episode_length // self.action_repeat
        if dev.type == "cuda":
            from .learner import fused_policy
            self._fp = fused_policy(net, n)
            if self._fp is not None:
                self._fp.refresh()                               # its packed weight copy <- the current parameters
        if dev.type == "cuda" and self.use_graph:
            # the env's outputs are persistent buffers and the accumulators are updated in place, so one step replays as
            # a graph; re-captured when the parameters move (FlatLearner re-homes them in its flat buffer)
            key = tuple(p.data_ptr() for p in net.policy.parameters())
            done = 0
            if self._graph is None or self._graph_key != key:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._one_step(net)                          # warm-up: this IS step 1 of the evaluation
                torch.cuda.current_stream().wait_stream(side)
                self._graph = torch.cuda.CUDAGraph()
                import torch.distributed as _dist
                live_pg = _dist.is_available() and _dist.is_initialized()     # (the group's watchdog thread must not trip the capture: learner._capture)
                if live_pg:
                    torch.cuda.synchronize()
                with torch.cuda.graph(self._graph, **(dict(capture_error_mode="thread_local") if live_pg else {})):
                    self._one_step(net)                          # captured, not executed
                self._graph_key = key
                done = 1
            for _ in range(nsteps - done):
                self._graph.replay()
        else:
            for _ in range(nsteps):
                self._one_step(net)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dt = time.time() - t0
        self._eval_walltime += dt
        out = {}
        for name, v in a["sums"].items():
            if aggregate_episodes:
                out[f"eval/episode_{name}"] = float(v.mean())
                out[f"eval/episode_{name}_std"] = float(v.std(unbiased=False))
            else:
                out[f"eval/episode_{name}"] = v.cpu().numpy()
        out["eval/avg_episode_length"] = float(a["steps"].mean())
        out["eval/std_episode_length"] = float(a["steps"].std(unbiased=False))
        out["eval/epoch_eval_time"] = dt
        out["eval/sps"] = self._steps_per_unroll / dt
        out["eval/walltime"] = self._eval_walltime
        return {**out, **training_metrics}
