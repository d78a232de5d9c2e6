"""Full-PPO throughput (BASELINE.json configs[2]): flat_terrain_backlash + domain randomisation, 8192 envs,
Appendix-G hyper-parameters; reports env-steps/s including the learner, and the rollout/learner split."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_duck_playground_amd import joystick
from open_duck_playground_amd.ppo import train as T
from open_duck_playground_amd.ppo.networks import PPONetworks

task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain_backlash"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
env = joystick.Joystick(task=task, num_envs=8192)
env.randomize(np.random.default_rng(0))
cfg = T.ppo_config()
dev = env.batch.obs.device
torch.manual_seed(0)
net = PPONetworks(101, 212, 14).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=cfg["learning_rate"], capturable=True)
learner = None
gen = torch.Generator(device=dev); gen.manual_seed(0)
state = env.reset(0)
tr = tl = 0.0
WARM = 2   # untimed: learner construction, library handles, selection file
for it in range(iters + WARM):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data, state = T.rollout(env, net, state, cfg["unroll_length"], gen)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    net.norm_obs.update(data["obs"]); net.norm_priv.update(data["priv"])
    if it == 0:
        learner = T.make_learner(net, data, cfg) if os.environ.get("ODK_EAGER_LEARNER") != "1" else None
    m = T.sgd_epoch(net, opt, data, cfg, gen, learner=learner)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= WARM:
        tr += t1 - t0; tl += t2 - t1
steps = iters * 8192 * cfg["unroll_length"]
print(json.dumps({"task": task, "iters": iters, "env_steps_per_s_total": steps / (tr + tl), "rollout_env_steps_per_s": steps / tr,
                  "rollout_s_per_iter": tr / iters, "learner_s_per_iter": tl / iters, "last_loss": float(m["total_loss"]),
                  "reward_per_step": float(data["reward"].mean())}))
