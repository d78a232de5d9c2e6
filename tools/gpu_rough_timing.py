"""Rough-terrain step time vs robot state (GPU box): zero actions (robots stand) vs random actions (robots topple), 10-step windows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
task = sys.argv[1] if len(sys.argv) > 1 else "rough_terrain_backlash"
model = load_task_model(task)
n = 8192
for mode in ("zero", "random"):
    cfg = engine.default_config(); cfg.noise_level = 0.0; cfg.push_enable = 0.0
    b = engine.Batch(model, n, cfg)
    b.reset(0)
    act = torch.zeros(n, 14, device="cuda")
    for w in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dones = 0.0
        e0.record()
        for _ in range(10):
            if mode == "random":
                act.uniform_(-1, 1)
            b.step(act)
        e1.record(); torch.cuda.synchronize()
        up = float((b.priv[:, 101 + 8] < -0.9).float().mean())      # gravity z in the imu frame < -0.9: upright
        print(f"{task} {mode} steps {10 * w:3d}-{10 * w + 9:3d}: {e0.elapsed_time(e1) / 10:.3f} ms/step, upright fraction {up:.2f}, done now {float(b.done.mean()):.3f}")
    b.close()
