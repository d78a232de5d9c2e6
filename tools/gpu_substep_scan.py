"""Per-substep cost vs per-env-step overhead of the fused step kernel (n_substeps = 1, 2, 10):  python tools/gpu_substep_scan.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
for task in ("flat_terrain", "flat_terrain_backlash"):
    model = load_task_model(task)
    res = {}
    for ns in (0, 1, 2, 10):
        cfg = engine.default_config(); cfg.noise_level = 0.0; cfg.push_enable = 0.0; cfg.n_substeps = ns
        b = engine.Batch(model, 8192, cfg); b.reset(0)
        act = torch.empty(64, 8192, 14, device="cuda").uniform_(-1, 1)
        for i in range(20): b.step(act[i % 64])
        torch.cuda.synchronize(); b.timing(True)
        for i in range(100): b.step(act[i % 64])
        torch.cuda.synchronize()
        ms, n = b.timing(False)
        res[ns] = ms
        b.close()
    per = (res[10] - res[2]) / 8
    print(task, {k: round(v, 4) for k, v in res.items()}, "per-substep ms", round(per, 4), "overhead ms (ns=1 minus one substep)", round(res[1] - per, 4))
