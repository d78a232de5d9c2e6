"""GPU tests of the two artefacts a training run hands to the outside world:

* the ONNX policy file (reference playground/common/export_onnx.py:170-183: input "obs" (1, obs_size), output
  "continuous_actions" = tanh(loc), opset 11; consumer playground/open_duck_mini_v2/mujoco_infer.py:67-103) against the HIP policy
  path the rollout / evaluator use, read back by an INDEPENDENT decoder: the field numbers of onnx.proto3 are written out below,
  nothing is imported from the exporter's own reader;
* the evaluator's numbers (reference common/runner.py:56-66 prints eval/episode_reward from brax acting.Evaluator: sums over each
  env's FIRST episode under the deterministic policy) against an oracle-side accumulation of the same episodes."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# ---- onnx.proto3 (onnx 1.x, IR version 6): field numbers of the messages a policy file uses
MODEL = dict(ir_version=1, producer_name=2, producer_version=3, domain=4, model_version=5, doc_string=6, graph=7, opset_import=8)
OPSET_ID = dict(domain=1, version=2)
GRAPH = dict(node=1, name=2, initializer=5, doc_string=10, input=11, output=12, value_info=13)
NODE = dict(input=1, output=2, name=3, op_type=4, attribute=5, doc_string=6, domain=7)
ATTR = dict(name=1, f=2, i=3, s=4, t=5, g=6, floats=7, ints=8, strings=9, type=20)
ATTR_TYPE = dict(FLOAT=1, INT=2)
TENSOR = dict(dims=1, data_type=2, float_data=4, name=8, raw_data=9)
VALUE_INFO = dict(name=1, type=2)
TYPE = dict(tensor_type=1)
TYPE_TENSOR = dict(elem_type=1, shape=2)
SHAPE = dict(dim=1)
DIM = dict(dim_value=1, dim_param=2)
DT_FLOAT = 1


def _fields(buf):
    """protobuf wire format -> [(field number, wire type, value)]; value = int (varint), bytes (length-delimited), 4 / 8 raw bytes"""
    out, pos = [], 0
    while pos < len(buf):
        key, shift = 0, 0
        while True:
            b = buf[pos]; pos += 1
            key |= (b & 0x7F) << shift; shift += 7
            if not b & 0x80:
                break
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, shift = 0, 0
            while True:
                b = buf[pos]; pos += 1
                v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
        elif wire == 2:
            n, shift = 0, 0
            while True:
                b = buf[pos]; pos += 1
                n |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            v = bytes(buf[pos: pos + n]); pos += n
            assert len(v) == n, "length-delimited field runs past the buffer"
        elif wire == 5:
            v = bytes(buf[pos: pos + 4]); pos += 4
        elif wire == 1:
            v = bytes(buf[pos: pos + 8]); pos += 8
        else:
            raise AssertionError(f"wire type {wire}")
        out.append((field, wire, v))
    assert pos == len(buf)
    return out


def _get(fs, num, wire=None):
    return [v for f, w, v in fs if f == num and (wire is None or w == wire)]


def _tensor(buf):
    fs = _fields(buf)
    assert set(f for f, _, _ in fs) <= set(TENSOR.values()), "unknown TensorProto field"
    dims = _get(fs, TENSOR["dims"], 0)
    assert _get(fs, TENSOR["data_type"], 0) == [DT_FLOAT]
    raw = _get(fs, TENSOR["raw_data"], 2)
    assert len(raw) == 1 and len(raw[0]) == 4 * int(np.prod(dims))
    return _get(fs, TENSOR["name"], 2)[0].decode(), np.frombuffer(raw[0], "<f4").reshape(dims)


def _value_info(buf):
    fs = _fields(buf)
    name = _get(fs, VALUE_INFO["name"], 2)[0].decode()
    tt = _fields(_get(_fields(_get(fs, VALUE_INFO["type"], 2)[0]), TYPE["tensor_type"], 2)[0])
    assert _get(tt, TYPE_TENSOR["elem_type"], 0) == [DT_FLOAT]
    dims = [_get(_fields(d), DIM["dim_value"], 0)[0] for d in _get(_fields(_get(tt, TYPE_TENSOR["shape"], 2)[0]), SHAPE["dim"], 2)]
    return name, dims


def _decode_model(blob):
    fs = _fields(blob)
    assert set(f for f, _, _ in fs) <= set(MODEL.values()), "unknown ModelProto field"
    m = dict(ir_version=_get(fs, MODEL["ir_version"], 0)[0])
    ops = [_fields(o) for o in _get(fs, MODEL["opset_import"], 2)]
    m["opsets"] = [((_get(o, OPSET_ID["domain"], 2) or [b""])[0].decode(), _get(o, OPSET_ID["version"], 0)[0]) for o in ops]
    g = _fields(_get(fs, MODEL["graph"], 2)[0])
    assert set(f for f, _, _ in g) <= set(GRAPH.values()), "unknown GraphProto field"
    m["init"] = dict(_tensor(t) for t in _get(g, GRAPH["initializer"], 2))
    m["inputs"] = [_value_info(v) for v in _get(g, GRAPH["input"], 2)]
    m["outputs"] = [_value_info(v) for v in _get(g, GRAPH["output"], 2)]
    m["nodes"] = []
    for nb in _get(g, GRAPH["node"], 2):
        n = _fields(nb)
        assert set(f for f, _, _ in n) <= set(NODE.values()), "unknown NodeProto field"
        attrs = {}
        for ab in _get(n, NODE["attribute"], 2):
            a = _fields(ab)
            nm = _get(a, ATTR["name"], 2)[0].decode(); ty = _get(a, ATTR["type"], 0)[0]
            if ty == ATTR_TYPE["FLOAT"]:
                attrs[nm] = struct.unpack("<f", _get(a, ATTR["f"], 5)[0])[0]
            else:
                assert ty == ATTR_TYPE["INT"]
                attrs[nm] = _get(a, ATTR["i"], 0)[0]
        m["nodes"].append(dict(op=_get(n, NODE["op_type"], 2)[0].decode(), inputs=[x.decode() for x in _get(n, NODE["input"], 2)],
                               outputs=[x.decode() for x in _get(n, NODE["output"], 2)], attrs=attrs))
    return m


def _run(m, obs):
    """numpy evaluation of the decoded graph with the operator semantics of the ONNX spec (opset 11)"""
    env = dict(m["init"]); env[m["inputs"][0][0]] = obs.astype(np.float32)
    for n in m["nodes"]:
        x = [env[i] for i in n["inputs"]]
        if n["op"] == "Sub": y = x[0] - x[1]
        elif n["op"] == "Div": y = x[0] / x[1]
        elif n["op"] == "Mul": y = x[0] * x[1]
        elif n["op"] == "Sigmoid": y = 1.0 / (1.0 + np.exp(-x[0].astype(np.float64))); y = y.astype(np.float32)
        elif n["op"] == "Tanh": y = np.tanh(x[0])
        elif n["op"] == "Gemm":   # Y = alpha A' B' + beta C
            a = x[0].T if n["attrs"].get("transA", 0) else x[0]; b = x[1].T if n["attrs"].get("transB", 0) else x[1]
            y = n["attrs"].get("alpha", 1.0) * (a @ b) + n["attrs"].get("beta", 1.0) * x[2]
        else:
            raise AssertionError(f"operator {n['op']} is not one a policy file may contain")
        env[n["outputs"][0]] = y.astype(np.float32)
    return env[m["outputs"][0][0]]


@pytest.mark.parametrize("kind", ["joystick", "standing"])
def test_onnx_file_matches_the_hip_policy_path(tmp_path, kind):
    """export -> independent decode -> numpy run == tanh(loc) of the whole-network HIP launch (what rollout / evaluator use), on 256
    observations; obs 101 (Joystick) and 85 (Standing)."""
    import torch
    from open_duck_playground_amd import export_onnx as X
    from open_duck_playground_amd.ppo.learner import fused_policy
    from open_duck_playground_amd.ppo.networks import PPONetworks
    nobs, npriv = (101, 212) if kind == "joystick" else (85, 153)
    torch.manual_seed(3)
    net = PPONetworks(nobs, npriv, 14).cuda()
    net.norm_obs.update(torch.randn(8192, 1, nobs, device="cuda") * 2.0 + 0.3)
    path = X.export_onnx(net, str(tmp_path / "policy.onnx"))
    m = _decode_model(open(path, "rb").read())
    assert m["ir_version"] == 6 and m["opsets"] == [("", 11)]                                   # export_onnx.py:177
    assert m["inputs"] == [("obs", [1, nobs])] and m["outputs"] == [("continuous_actions", [1, 14])]   # export_onnx.py:170-175
    obs = (torch.randn(256, nobs, device="cuda") * 2.0 + 0.3).contiguous()
    fp = fused_policy(net, 256)
    assert fp is not None, "the whole-network kernel must serve this architecture"
    fp.refresh()
    act_hip = torch.tanh(fp(obs)[:, :14]).cpu().numpy()
    with torch.no_grad():
        act_torch = torch.tanh(net.dist_params(obs)[0]).cpu().numpy()
    o = obs.cpu().numpy()
    act_onnx = np.concatenate([_run(m, o[i: i + 1]) for i in range(256)])
    assert np.abs(act_onnx - act_hip).max() < 1e-5, np.abs(act_onnx - act_hip).max()
    assert np.abs(act_onnx - act_torch).max() < 1e-5


class _ShadowEnv:
    """The evaluator's env with an oracle in its shadow: every step first re-synchronises the GPU's physics state from the oracle
    envs, then both sides take the evaluator's action; the oracle side keeps its own first-episode sums."""

    def __init__(self, env, oracle_mod, standing=False):
        import oracle as O  # noqa: F401
        from open_duck_playground_amd import engine
        self.env, self.num_envs, self.METRIC_NAMES = env, env.num_envs, env.METRIC_NAMES
        self.model = env.mj_model
        om = oracle_mod.OracleModel(self.model.blob()); prm = oracle_mod.OraclePRM(engine.load_prm())
        self.keep = (om, prm)
        self.o = [oracle_mod.OracleEnv(om, prm, standing=standing) for _ in range(self.num_envs)]
        for e in self.o:
            e.cfg["episode_length"][0] = env.batch.cfg.episode_length
        self.offset = env._env_id_offset

    def reset(self, seed):
        st = self.env.reset(seed)
        n = self.num_envs
        for i, e in enumerate(self.o):
            e.reset(seed, self.offset + i)
        self.active = np.ones(n); self.sum_reward = np.zeros(n); self.sum_metrics = np.zeros((n, 8)); self.steps = np.zeros(n)
        return st

    def step(self, state, action):
        import torch
        m = self.model
        b = self.env.batch
        b.set_state(np.stack([np.array(e.data["qpos"][: m.nq]) for e in self.o]), np.stack([np.array(e.data["qvel"][: m.nv]) for e in self.o]),
                    np.stack([np.array(e.data["qacc_warmstart"][: m.nv]) for e in self.o]))
        a = action.cpu().numpy()
        for i, e in enumerate(self.o):
            e.step(a[i])
            self.sum_reward[i] += e["reward"][0] * self.active[i]
            self.sum_metrics[i] += np.array(e["metrics"][:8]) * self.active[i]
            self.steps[i] += self.active[i]
            self.active[i] *= 1.0 - e["done"][0]
        return self.env.step(state, action)


def test_evaluator_sums_match_an_oracle_side_accumulation(oracle_mod, parity_log):
    """eval/episode_reward, the per-term eval/episode_reward/* / cost/* sums and eval/avg_episode_length of `Evaluator` (first
    episode of every env, deterministic policy) vs the same episodes accumulated from the ORACLE's rewards and dones."""
    import torch
    from open_duck_playground_amd import joystick
    from open_duck_playground_amd.ppo.evaluator import Evaluator
    from open_duck_playground_amd.ppo.networks import PPONetworks
    torch.manual_seed(0)
    net = PPONetworks(101, 212, 14).cuda()
    env = joystick.Joystick(task="flat_terrain", num_envs=48, config_overrides={"episode_length": 40})
    sh = _ShadowEnv(env, oracle_mod)
    ev = Evaluator(sh, 40, use_graph=False)     # (the graph replay is compared with plain launches in test_gpu_api.py)
    out = ev.run_evaluation(net, {}, seed=4, aggregate_episodes=False)
    rew = out["eval/episode_reward"]; steps_g = ev._acc["steps"].cpu().numpy()
    assert np.array_equal(steps_g, sh.steps), "first-episode lengths differ"
    assert steps_g.min() >= 1 and (steps_g < 40).any() and (steps_g == 40).any()      # both terminated and full-length episodes
    rel = np.abs(rew - sh.sum_reward) / np.maximum(np.abs(sh.sum_reward), 1.0)
    worst_terms = 0.0
    for k, name in enumerate(env.METRIC_NAMES):
        g = out[f"eval/episode_{name}"]
        worst_terms = max(worst_terms, float(np.quantile(np.abs(g - sh.sum_metrics[:, k]) / np.maximum(np.abs(sh.sum_metrics[:, k]), 1.0), 0.9)))
    mean_err = abs(rew.mean() - sh.sum_reward.mean()) / abs(sh.sum_reward.mean())
    # an env step that sits on a branch point of the solver (tests/test_gpu_env.py) moves one episode's sum by a few 1e-3; the
    # bulk of the episodes and the mean over the envs (what the runner prints) must agree
    parity_log.check("evaluator/flat_terrain", dict(reward_p90=5e-4, reward_mean=1e-3, terms_p90=1e-3),
                     reward_p90=float(np.quantile(rel, 0.9)), reward_max=float(rel.max()), reward_mean=float(mean_err), terms_p90=worst_terms)
    # and the aggregate form the runner logs
    agg = Evaluator(sh, 40, use_graph=False).run_evaluation(net, {}, seed=4)
    assert agg["eval/episode_reward"] == pytest.approx(float(rew.mean()), rel=1e-6)
    assert agg["eval/avg_episode_length"] == pytest.approx(float(steps_g.mean()), rel=1e-6)
