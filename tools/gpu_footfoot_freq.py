"""How often the foot-foot pair needs more than the bounding-sphere test in a random-action rollout (fractions of envs / of
waves whose boxes overlap or whose feet penetrate):  python tools/gpu_footfoot_freq.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_duck_playground_amd import engine
from open_duck_playground_amd.model import load_task_model
model = load_task_model("flat_terrain")
cfg = engine.default_config(); cfg.noise_level = 0.0; cfg.push_enable = 0.0
b = engine.Batch(model, 8192, cfg); b.reset(0)
act = torch.empty(8192, 14, device="cuda")
b.L.odk_set_debug_dump(1)
o = b.lds_offset("scr") + 156
tot = ov = pen = 0
for t in range(60):
    b.step(act.uniform_(-1, 1))
    if t % 5 == 4:
        img = b.lds_image()
        v = img[:, o]
        cd = img[:, b.lds_offset("contact_dist") + 8: b.lds_offset("contact_dist") + 12]
        tot += len(v); ov += int((v <= 0).sum()); pen += int((cd.min(axis=1) < 0).sum())
        # waves = pairs of envs
        w = (v.reshape(-1, 2) <= 0).any(axis=1).mean()
        print(t, "overlap frac", (v <= 0).mean(), "wave frac", w, "penetrating", (cd.min(axis=1) < 0).mean())
