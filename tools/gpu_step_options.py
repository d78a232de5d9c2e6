"""Env-step launch time with the training-time options switched on one at a time (observation noise, pushes, domain
randomisation; the headline bench runs with all three off):  python tools/gpu_step_options.py [task]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine, randomize
from open_duck_playground_amd.model import load_task_model
task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain_backlash"
model = load_task_model(task)
SCALE = float(os.environ.get("ODK_ACT_SCALE", "1"))   # 1: the bench's uniform actions; 0.3 ~ an untrained policy's (robots mostly stay up)
act = torch.empty(64, 8192, 14, device="cuda").uniform_(-1, 1) * SCALE
for name, noise, push, dr in (("bench (all off)", 0, 0, 0), ("+ noise", 1, 0, 0), ("+ pushes", 0, 1, 0), ("+ domain randomisation", 0, 0, 1), ("training (all on)", 1, 1, 1)):
    cfg = engine.default_config()
    if not noise: cfg.noise_level = 0.0
    if not push: cfg.push_enable = 0.0
    b = engine.Batch(model, 8192, cfg)
    if dr:
        fields, _ = randomize.domain_randomize(model, np.random.default_rng(0), 8192)
        randomize.apply(b, fields)
    b.reset(0)
    for i in range(30): b.step(act[i % 64])
    torch.cuda.synchronize(); b.timing(4)
    for i in range(200): b.step(act[i % 64])
    torch.cuda.synchronize()
    ms, n = b.timing(False)
    print(f"{task:24s} {name:26s} {ms:.4f} ms  done {float(b.done.mean()):.3f}", flush=True)
    b.close()
