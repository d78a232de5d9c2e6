"""Policy export to ONNX (counterpart of reference playground/common/export_onnx.py:7-189).

The reference rebuilds the policy in TensorFlow and converts it with tf2onnx; neither is needed here: the graph is
six op types, so the ONNX ModelProto is written directly in protobuf wire format (no `onnx` package in this
image).  Contract kept from the reference (export_onnx.py:170-183): input `obs` of shape (1, obs_size) float32,
output `continuous_actions` = tanh(loc) of shape (1, action_size), opset 11, so the reference's
`mujoco_infer.py` / `onnx_infer.py` can run a policy trained here unchanged.

Graph: (obs - mean) / std -> [Gemm(transB=1) -> x * Sigmoid(x)] x hidden -> Gemm (loc half of the last layer) -> Tanh.
`load_onnx` / `run_onnx` are a minimal reader + numpy evaluator for the ops emitted here (used by the tests and
for a self-check after export); they read the same wire format back, field numbers from onnx.proto3.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

IR_VERSION, OPSET = 6, 11     # opset 11 "matches isaac lab" (export_onnx.py:177)
FLOAT = 1                     # TensorProto.DataType.FLOAT


# ---------------------------------------------------------------- protobuf wire helpers
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _int(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(int(v))


def _bytes(field: int, v: bytes) -> bytes:
    return _key(field, 2) + _varint(len(v)) + v


def _str(field: int, v: str) -> bytes:
    return _bytes(field, v.encode())


def _f32(field: int, v: float) -> bytes:
    return _key(field, 5) + struct.pack("<f", v)


# ---------------------------------------------------------------- onnx messages (field numbers: onnx.proto3)
def _tensor(name: str, arr: np.ndarray) -> bytes:           # TensorProto: dims=1, data_type=2, name=8, raw_data=9
    arr = np.ascontiguousarray(arr, np.float32)
    return b"".join(_int(1, d) for d in arr.shape) + _int(2, FLOAT) + _str(8, name) + _bytes(9, arr.tobytes())


def _value_info(name: str, shape) -> bytes:                  # ValueInfoProto: name=1, type=2{tensor_type=1{elem_type=1, shape=2{dim=1{dim_value=1}}}}
    dims = b"".join(_bytes(1, _int(1, d)) for d in shape)
    return _str(1, name) + _bytes(2, _bytes(1, _int(1, FLOAT) + _bytes(2, dims)))


def _attr_int(name: str, v: int) -> bytes:                   # AttributeProto: name=1, i=3, type=20 (INT=2)
    return _str(1, name) + _int(3, v) + _int(20, 2)


def _attr_float(name: str, v: float) -> bytes:               # f=2, type FLOAT=1
    return _str(1, name) + _f32(2, v) + _int(20, 1)


def _node(op: str, inputs: List[str], outputs: List[str], name: str, attrs: List[bytes] = ()) -> bytes:   # NodeProto: input=1, output=2, name=3, op_type=4, attribute=5
    return b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs) + _str(3, name) + _str(4, op) + \
        b"".join(_bytes(5, a) for a in attrs)


def policy_to_onnx(mean: np.ndarray, std: np.ndarray, weights: List[np.ndarray], biases: List[np.ndarray], action_size: int) -> bytes:
    """weights[i]: [out, in] (torch Linear layout); the last layer has 2*action_size rows, of which the `loc` half is kept."""
    obs_size = int(mean.shape[0])
    weights = [np.asarray(w, np.float32) for w in weights]; biases = [np.asarray(b, np.float32) for b in biases]
    weights[-1], biases[-1] = weights[-1][:action_size], biases[-1][:action_size]   # loc, _ = split(logits, 2) (export_onnx.py:71)
    inits = [_tensor("obs_mean", mean.reshape(1, -1)), _tensor("obs_std", std.reshape(1, -1))]
    nodes = [_node("Sub", ["obs", "obs_mean"], ["centered"], "normalize_sub"), _node("Div", ["centered", "obs_std"], ["h0"], "normalize_div")]
    x = "h0"
    for i, (W, b) in enumerate(zip(weights, biases)):
        inits += [_tensor(f"hidden_{i}/kernel", W), _tensor(f"hidden_{i}/bias", b)]
        z = f"z{i}"
        nodes.append(_node("Gemm", [x, f"hidden_{i}/kernel", f"hidden_{i}/bias"], [z], f"hidden_{i}",
                           [_attr_float("alpha", 1.0), _attr_float("beta", 1.0), _attr_int("transA", 0), _attr_int("transB", 1)]))
        if i + 1 < len(weights):   # swish (export_onnx.py:100)
            nodes += [_node("Sigmoid", [z], [f"s{i}"], f"swish_{i}/sigmoid"), _node("Mul", [z, f"s{i}"], [f"h{i + 1}"], f"swish_{i}/mul")]
            x = f"h{i + 1}"
        else:
            nodes.append(_node("Tanh", [z], ["continuous_actions"], "tanh_loc"))
    graph = b"".join(_bytes(1, n) for n in nodes) + _str(2, "open_duck_policy") + b"".join(_bytes(5, t) for t in inits) + \
        _bytes(11, _value_info("obs", (1, obs_size))) + _bytes(12, _value_info("continuous_actions", (1, action_size)))
    # ModelProto: ir_version=1, producer_name=2, producer_version=3, graph=7, opset_import=8{domain=1, version=2}
    return _int(1, IR_VERSION) + _str(2, "open_duck_playground_amd") + _str(3, "1") + _bytes(7, graph) + _bytes(8, _str(1, "") + _int(2, OPSET))


def export_onnx(net, output_path: str = "ONNX.onnx", check: bool = True) -> str:
    """Writes the deterministic policy of a `ppo.networks.PPONetworks` (reference signature: export_onnx(params, act_size,
    ppo_params, obs_size, output_path); here the module carries all four)."""
    mean = net.norm_obs.mean.detach().cpu().numpy(); std = net.norm_obs.std.detach().cpu().numpy()
    Ws = [l.weight.detach().cpu().numpy() for l in net.policy.layers]; bs = [l.bias.detach().cpu().numpy() for l in net.policy.layers]
    blob = policy_to_onnx(mean, std, Ws, bs, net.action_size)
    if check:
        # The reference prints the graph's prediction for an all-ones observation; here the graph must also EQUAL the module.
        # The comparison runs at an in-distribution point (mean +- std: normalised inputs of +-1): with all ones, an
        # observation entry that never varies (std clamped to 1e-6, e.g. Standing's unused command slots) normalises to 1e6 and
        # the two float32 evaluations then differ by rounding alone -- that input only has to give finite actions.
        import torch
        sign = np.where(np.arange(mean.shape[0]) % 2 == 0, 1.0, -1.0).astype(np.float32)
        x = (mean + sign * std).astype(np.float32)[None, :]
        model = load_onnx(blob)
        got = run_onnx(model, x)
        with torch.no_grad():
            loc, _ = net.dist_params(torch.from_numpy(x).to(net.norm_obs.mean.device))
        ref = torch.tanh(loc).cpu().numpy()
        if not np.allclose(got, ref, rtol=1e-4, atol=1e-5):
            raise RuntimeError("exported ONNX graph disagrees with the policy module")
        if not np.all(np.isfinite(run_onnx(model, np.ones((1, mean.shape[0]), np.float32)))):
            raise RuntimeError("exported ONNX graph returns non-finite actions for an all-ones observation")
    with open(output_path, "wb") as f:
        f.write(blob)
    return output_path


# ---------------------------------------------------------------- minimal reader + evaluator (tests / self-check)
def _parse(buf: bytes) -> List[Tuple[int, int, object]]:
    out, i = [], 0
    while i < len(buf):
        k = 0; sh = 0
        while True:
            b = buf[i]; i += 1
            k |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                break
        field, wire = k >> 3, k & 7
        if wire == 0:
            v = 0; sh = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            out.append((field, wire, v))
        elif wire == 2:
            n = 0; sh = 0
            while True:
                b = buf[i]; i += 1
                n |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            out.append((field, wire, buf[i:i + n])); i += n
        elif wire == 5:
            out.append((field, wire, struct.unpack("<f", buf[i:i + 4])[0])); i += 4
        elif wire == 1:
            out.append((field, wire, buf[i:i + 8])); i += 8
        else:
            raise ValueError(f"unsupported wire type {wire}")
    return out


def load_onnx(blob: bytes) -> Dict:
    model = _parse(blob)
    g = _parse(next(v for f, _, v in model if f == 7))
    opset = [dict((f, v) for f, _, v in _parse(v)) for f, _, v in model if f == 8]
    inits = {}
    for f, _, v in g:
        if f == 5:
            t = _parse(v)
            dims = [x for ff, _, x in t if ff == 1]
            name = next(x for ff, _, x in t if ff == 8).decode()
            assert next(x for ff, _, x in t if ff == 2) == FLOAT
            inits[name] = np.frombuffer(next(x for ff, _, x in t if ff == 9), np.float32).reshape(dims)
    nodes = []
    for f, _, v in g:
        if f == 1:
            n = _parse(v)
            attrs = {}
            for ff, _, a in n:
                if ff == 5:
                    ap = _parse(a)
                    nm = next(x for q, _, x in ap if q == 1).decode()
                    ty = next(x for q, _, x in ap if q == 20)
                    attrs[nm] = next(x for q, _, x in ap if q == (3 if ty == 2 else 2))
            nodes.append(dict(op=next(x for ff, _, x in n if ff == 4).decode(), inputs=[x.decode() for ff, _, x in n if ff == 1],
                              outputs=[x.decode() for ff, _, x in n if ff == 2], attrs=attrs))

    def vi(v):
        p = _parse(v)
        name = next(x for f, _, x in p if f == 1).decode()
        tt = _parse(next(x for f, _, x in _parse(next(x for f, _, x in p if f == 2)) if f == 1))
        shape = [next(x for f, _, x in _parse(d) if f == 1) for f2, _, d in _parse(next(x for f, _, x in tt if f == 2)) if f2 == 1]
        return name, shape
    return dict(ir_version=next(v for f, _, v in model if f == 1), opset=opset[0].get(2), nodes=nodes, initializers=inits,
                inputs=[vi(v) for f, _, v in g if f == 11], outputs=[vi(v) for f, _, v in g if f == 12])


def run_onnx(model: Dict, obs: np.ndarray) -> np.ndarray:
    env = dict(model["initializers"]); env[model["inputs"][0][0]] = np.asarray(obs, np.float32)
    for n in model["nodes"]:
        a = [env[i] for i in n["inputs"]]
        if n["op"] == "Sub": r = a[0] - a[1]
        elif n["op"] == "Div": r = a[0] / a[1]
        elif n["op"] == "Mul": r = a[0] * a[1]
        elif n["op"] == "Sigmoid": r = 1.0 / (1.0 + np.exp(-a[0]))
        elif n["op"] == "Tanh": r = np.tanh(a[0])
        elif n["op"] == "Gemm":
            A = a[0].T if n["attrs"].get("transA") else a[0]; B = a[1].T if n["attrs"].get("transB") else a[1]
            r = n["attrs"].get("alpha", 1.0) * (A @ B) + n["attrs"].get("beta", 1.0) * a[2]
        else:
            raise ValueError(f"op {n['op']} not emitted by this exporter")
        env[n["outputs"][0]] = r.astype(np.float32)
    return env[model["outputs"][0][0]]
