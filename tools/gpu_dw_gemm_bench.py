"""odk_dw_gemm (split-K weight-gradient GEMMs on the f32 matrix cores) against torch.mm: accuracy and time.
    python tools/gpu_dw_gemm_bench.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine
torch.manual_seed(0)
dev = "cuda"
def run(n, shapes, kslices=16, iters=200):
    tot = sum((o * i + 7) // 4 * 4 for o, i in shapes) + 1001
    flat = torch.zeros(tot, device=dev); ws = torch.zeros(kslices * engine.DwGemm.workspace_stride(tot), device=dev)
    layers, off = [], 8
    for o, i in shapes:
        dz = torch.randn(n, o, device=dev); h = torch.randn(n, i, device=dev)
        layers.append((dz, h, off)); off += (o * i + 7) // 4 * 4
    g = engine.DwGemm(layers, flat, ws, kslices)
    g(); torch.cuda.synchronize()
    worst = 0
    for dz, h, o in layers:
        ref = (dz.double().t() @ h.double())
        got = flat[o:o + ref.numel()].view_as(ref)
        worst = max(worst, float((got.double() - ref).abs().max() / ref.abs().max()))
    for _ in range(20): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): g()
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / iters
    outs = [torch.empty(o, i, device=dev) for o, i in shapes]
    for _ in range(20):
        for (dz, h, _), out in zip(layers, outs): torch.mm(dz.t(), h, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        for (dz, h, _), out in zip(layers, outs): torch.mm(dz.t(), h, out=out)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / iters
    fl = 2 * n * sum(o * i for o, i in shapes)
    print(f"n={n} shapes={shapes} ks={kslices}: rel err {worst:.2e}; odk {t1*1e6:.1f} us ({fl/t1/1e12:.1f} TF), torch.mm x{len(shapes)} {t2*1e6:.1f} us ({fl/t2/1e12:.1f} TF)")
run(5120, [(512, 101), (256, 512), (128, 256), (28, 128)])
run(5376, [(512, 212), (256, 512), (128, 256), (1, 128)])
run(5120, [(512, 101), (256, 512), (128, 256), (28, 128)], kslices=8)
run(5120, [(512, 101), (256, 512), (128, 256), (28, 128)], kslices=32)
run(1280, [(38, 70), (64, 33)], kslices=8)
