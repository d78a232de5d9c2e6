"""Times the learner's network launches alone at the reference sizes (policy 5120 x 101 -> 28, value 5376 x 212 -> 1):
    python tools/gpu_mlp_bench.py      -> us per launch: forward (training / inference), backward, weight gradients"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine

g = torch.Generator(device="cuda").manual_seed(0)
nets, dw_layers, goff = [], [], 0
tot = 0
specs = ((5120, 101, 28), (5376, 212, 1))
entries = []
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    for l in range(4):
        entries.append((tot, widths[l + 1], widths[l], l > 0)); tot += widths[l + 1] * widths[l] + widths[l + 1]
table = engine.WeightTable(entries)
flat = torch.randn(tot, device="cuda", generator=g) * 0.05
pf, pb = torch.zeros(table.fwd_size, device="cuda"), torch.zeros(table.bwd_size, device="cuda")
engine.pack_weights(flat, pf, pb, table)
flat_g = torch.zeros(tot, device="cuda")
k = 0
for n, n_in, n_out in specs:
    widths = (n_in,) + engine.MLP_HIDDEN + (n_out,)
    tiles = (n + 31) // 32
    E = lambda *s: torch.empty(*s, device="cuda")
    x = torch.randn(n, n_in, device="cuda", generator=g)
    b = [flat[entries[k + l][0] + widths[l + 1] * widths[l]:][:widths[l + 1]] for l in range(4)]
    d = dict(x=x, wf=[table.fwd_view(pf, k + l) for l in range(4)], wb=[table.bwd_view(pb, k + l) for l in range(4)], b=b, out=E(n, n_out),
             dout=torch.randn(n, n_out, device="cuda", generator=g) * 1e-3, **engine.FusedMLP.train_buffers(n, n_in, n_out, "cuda"))
    nets.append(d)
    hs = [d["xp"]] + d["h"]; dzs = d["dz"] + [d["doutp"]]
    dw_layers += [(dzs[l], hs[l], widths[l + 1], widths[l], entries[k + l][0]) for l in range(4)]
    k += 4
train = engine.FusedMLP(nets)
infer = engine.FusedMLP([dict(x=d["x"], wf=d["wf"], b=d["b"], out=d["out"]) for d in nets])
one = [engine.FusedMLP([d]) for d in nets]
KS = int(os.environ.get("ODK_DW_KS", "8"))   # row slices of the weight-gradient launch
ws = torch.empty(KS * engine.DwGemm.workspace_stride(tot), device="cuda")
dw = engine.DwGemm(dw_layers, flat_g, ws, KS)


if os.environ.get("ODK_MLP_DIAG"):      # diagnostic variants of the network launches (ODK_LIB=.../libodk_mlpdiag.so; results are WRONG): csrc/odk_mlp.hip
    import ctypes
    L = engine.load_library()
    L.odk_mlp_set_diag.argtypes = [ctypes.c_int]
    L.odk_mlp_set_diag(int(os.environ["ODK_MLP_DIAG"]))


def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


flops_f = sum(2 * n * sum(a * b for a, b in zip((i,) + engine.MLP_HIDDEN, engine.MLP_HIDDEN + (o,))) for n, i, o in specs)
res = {"fwd_train_us": timeit(train.forward), "fwd_infer_us": timeit(infer.forward), "bwd_us": timeit(train.backward), "dw_us": timeit(dw),
       "fwd_policy_alone_us": timeit(one[0].forward), "fwd_value_alone_us": timeit(one[1].forward),
       "bwd_policy_alone_us": timeit(one[0].backward), "bwd_value_alone_us": timeit(one[1].backward)}
res = {k: round(v, 1) for k, v in res.items()}
res["fwd_tflops"] = round(flops_f / res["fwd_train_us"] / 1e6, 1)
res["dw_tflops"] = round(flops_f / res["dw_us"] / 1e6, 1)
print(res)
