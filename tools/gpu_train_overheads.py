"""Where a training step's time goes OUTSIDE the env-step and minibatch-step launches (config 3, 8192 envs): HIP-event timings of the pieces of
`rollout` and `sgd_epoch` around them.   python tools/gpu_train_overheads.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_duck_playground_amd import joystick
from open_duck_playground_amd.ppo import train as T
from open_duck_playground_amd.ppo.networks import PPONetworks

env = joystick.Joystick(task="flat_terrain_backlash", num_envs=8192)
env.randomize(np.random.default_rng(0))
cfg = T.ppo_config(); dev = env.batch.obs.device
torch.manual_seed(0)
net = PPONetworks(101, 212, 14).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(0)
state = env.reset(0)
data, state = T.rollout(env, net, state, 20, gen)
net.norm_obs.update(data["obs"]); net.norm_priv.update(data["priv"])
lr = T.make_learner(net, data, cfg)
T.sgd_epoch(net, None, data, cfg, gen, learner=lr, meter=T.LossMeter())


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return round(a.elapsed_time(b) / reps, 3)


B = data["reward"].shape[0]
out = {}
def roll():
    global data, state
    data, state = T.rollout(env, net, state, 20, gen)
out["rollout_ms"] = timed(roll, 5)
out["norm_obs_update_ms"] = timed(lambda: net.norm_obs.update(data["obs"]))
out["norm_priv_update_ms"] = timed(lambda: net.norm_priv.update(data["priv"]))
out["sync_weights_ms"] = timed(lr.sync_weights)
out["load_rollout_from_ms"] = timed(lambda: lr.load_rollout_from(net, data, cfg))
perms = torch.cat([torch.randperm(B, generator=gen, device=dev) for _ in range(4)])
out["randperm_x4_cat_ms"] = timed(lambda: torch.cat([torch.randperm(B, generator=gen, device=dev) for _ in range(4)]))
out["set_schedule_ms"] = timed(lambda: lr.set_schedule(perms))
def steps():
    lr.set_schedule(perms); lr.run(128)
out["set_schedule_plus_128_steps_ms"] = timed(steps, 5)
out["sgd_epoch_ms"] = timed(lambda: T.sgd_epoch(net, None, data, cfg, gen, learner=lr, meter=T.LossMeter()), 5)
b = data  # transposes inside rollout: time them alone
rb = T._ROLLOUT_BUFFERS.get(net)
out["transpose_contiguous_ms"] = timed(lambda: {k: v.transpose(0, 1).contiguous() for k, v in rb.buf.items()})
print(json.dumps(out))
