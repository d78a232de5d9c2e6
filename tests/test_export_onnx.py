"""ONNX policy export (reference playground/common/export_onnx.py): contract = input "obs" (1, obs_size), output
"continuous_actions" = tanh(loc) (1, action_size), opset 11."""
import numpy as np
import torch

from open_duck_playground_amd import export_onnx as X
from open_duck_playground_amd.ppo.networks import PPONetworks


def test_export_matches_policy_module(tmp_path):
    torch.manual_seed(0)
    net = PPONetworks(101, 212, 14)
    net.norm_obs.update(torch.randn(4096, 101) * 3.0 + 0.5)
    path = X.export_onnx(net, str(tmp_path / "policy.onnx"))
    blob = open(path, "rb").read()
    m = X.load_onnx(blob)
    assert m["ir_version"] == 6 and m["opset"] == 11                                   # export_onnx.py:177
    assert m["inputs"] == [("obs", [1, 101])] and m["outputs"] == [("continuous_actions", [1, 14])]   # :170-175
    ops = [n["op"] for n in m["nodes"]]
    assert ops == ["Sub", "Div", "Gemm", "Sigmoid", "Mul", "Gemm", "Sigmoid", "Mul", "Gemm", "Sigmoid", "Mul", "Gemm", "Tanh"]
    assert m["initializers"]["hidden_3/kernel"].shape == (14, 128)                     # loc half only (:71)
    rng = np.random.default_rng(0)
    for _ in range(5):
        obs = rng.normal(0, 2, (1, 101)).astype(np.float32)
        with torch.no_grad():
            loc, _ = net.dist_params(torch.from_numpy(obs))
        np.testing.assert_allclose(X.run_onnx(m, obs), torch.tanh(loc).numpy(), rtol=1e-4, atol=1e-5)


def test_wire_format_is_valid_protobuf():
    """Independent check of the byte stream with the protobuf runtime: every length-delimited field nests cleanly."""
    from google.protobuf.internal import decoder
    blob = X.policy_to_onnx(np.zeros(5, np.float32), np.ones(5, np.float32), [np.eye(4, 5, dtype=np.float32), np.ones((6, 4), np.float32)],
                            [np.zeros(4, np.float32), np.zeros(6, np.float32)], 3)
    pos, fields = 0, []
    while pos < len(blob):
        tag, pos = decoder._DecodeVarint(blob, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            _, pos = decoder._DecodeVarint(blob, pos)
        else:
            assert wire == 2
            n, pos = decoder._DecodeVarint(blob, pos)
            pos += n
        fields.append(field)
    assert pos == len(blob) and fields == [1, 2, 3, 7, 8]
    m = X.load_onnx(blob)
    out = X.run_onnx(m, np.array([[1, 2, 3, 4, 5]], np.float32))
    x = np.array([1, 2, 3, 4]) ; h = x / (1 + np.exp(-x)) * 1.0
    np.testing.assert_allclose(out, np.tanh(np.full((1, 3), h.sum())), rtol=1e-6)


def test_self_check_survives_constant_observation_entries(tmp_path):
    """Standing's unused command slots never vary: their std clamps to 1e-6, an all-ones observation normalises to 1e6 there
    and float32 evaluations of the same network differ by rounding alone.  The export's self-check (which once ended a
    training run at its last checkpoint) compares at mean +- std and only asks the all-ones input for finite actions."""
    torch.manual_seed(1)
    net = PPONetworks(85, 153, 14)
    obs = torch.randn(4096, 85) * 2.0
    obs[:, 10:17] = 0.0                      # entries that never vary
    net.norm_obs.update(obs)
    assert abs(float(net.norm_obs.std[10]) - 1e-6) < 1e-12
    with torch.no_grad():                    # large first-layer weights on those entries: rounding is amplified
        net.policy.layers[0].weight[:, 10:17] *= 50.0
    path = X.export_onnx(net, str(tmp_path / "policy.onnx"), check=True)
    m = X.load_onnx(open(path, "rb").read())
    assert np.all(np.isfinite(X.run_onnx(m, np.ones((1, 85), np.float32))))


def test_file_read_back_by_an_independent_decoder(tmp_path):
    """The same file through the decoder of tests/test_gpu_export_eval.py -- onnx.proto3 field numbers written out there, nothing
    shared with the exporter's own load_onnx -- and a numpy run with the operator semantics of the ONNX spec."""
    import test_gpu_export_eval as D
    torch.manual_seed(3)
    net = PPONetworks(85, 153, 14)
    net.norm_obs.update(torch.randn(4096, 1, 85) * 2.0 + 0.3)
    m = D._decode_model(open(X.export_onnx(net, str(tmp_path / "policy.onnx")), "rb").read())
    assert m["ir_version"] == 6 and m["opsets"] == [("", 11)]
    assert m["inputs"] == [("obs", [1, 85])] and m["outputs"] == [("continuous_actions", [1, 14])]
    assert m["nodes"][2]["attrs"] == {"alpha": 1.0, "beta": 1.0, "transA": 0, "transB": 1}
    obs = torch.randn(16, 85) * 2.0
    with torch.no_grad():
        ref = torch.tanh(net.dist_params(obs)[0]).numpy()
    got = np.concatenate([D._run(m, obs.numpy()[i: i + 1]) for i in range(16)])
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
