import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from open_duck_playground_amd import engine
ks = int(sys.argv[1]); which = sys.argv[2]
n, shapes = (5120, [(512, 101), (256, 512), (128, 256), (28, 128)]) if which == "P" else (5376, [(512, 212), (256, 512), (128, 256), (1, 128)])
tot = sum(o * i for o, i in shapes) + 1000
flat = torch.zeros(tot, device="cuda"); ws = torch.zeros(ks * engine.DwGemm.workspace_stride(tot), device="cuda")
layers, off = [], 0
for o, i in shapes:
    layers.append((torch.randn(n, o, device="cuda"), torch.randn(n, i, device="cuda"), off)); off += o * i
g = engine.DwGemm(layers, flat, ws, ks)
for _ in range(int(os.environ.get("DW_ITERS", "200"))): g()
torch.cuda.synchronize()
