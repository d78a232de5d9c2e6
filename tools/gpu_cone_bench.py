#!/usr/bin/env python3
"""Env-step rate of the duck with pyramidal and with elliptic friction cones (the env kernels' ShapeA / ShapeAE, ShapeB / ShapeBE instantiations):
    python tools/gpu_cone_bench.py [task] [envs] [steps]      (bench.py's physics protocol: random actions, noise off, pushes off)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_duck_playground_amd import engine  # noqa: E402
from open_duck_playground_amd.model import Model, load_task_model  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
base = load_task_model(task)
for cone in (0, 1, 0, 1):
    model = Model({**base.a, "opt_cone": np.array([cone], np.int32)})
    cfg = engine.default_config()
    cfg.noise_level = 0.0; cfg.push_enable = 0.0
    b = engine.Batch(model, envs, cfg)
    b.reset(seed=0)
    act = torch.empty(64, envs, 14, device="cuda").uniform_(-1.0, 1.0)
    for t in range(100):
        b.step(act[t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        b.step(act[t % 64])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{task} cone={'elliptic' if cone else 'pyramidal'}: {envs * steps / dt / 1e6:.2f} M env-steps/s ({dt / steps * 1e3:.4f} ms per step), reward mean {float(b.reward.mean()):.4f}", flush=True)
    b.close()
