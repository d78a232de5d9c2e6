"""Runs training for Open Duck Mini V2 (mirror of reference playground/open_duck_mini_v2/runner.py and
playground/common/runner.py).  Same flags as the reference (runner.py:36-56) plus additive ones
(--num_envs, --seed, --device, --no_randomize).  Launch with torchrun for multi-GPU data parallelism.

    python -m open_duck_playground_amd.runner --task flat_terrain --num_timesteps 150000000
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import numpy as np


class OpenDuckMiniV2Runner:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from . import joystick, standing
        from .ppo import train as ppo_train
        self.args = args
        self.output_dir = os.path.join(os.getcwd(), args.output_dir)
        available_envs = {"joystick": joystick.Joystick, "standing": standing.Standing}   # reference runner.py:14-17
        if args.env not in available_envs:
            raise ValueError(f"Unknown env {args.env}")
        self.world = int(os.environ.get("WORLD_SIZE", "1")); self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = args.device if self.world == 1 else local_rank
        torch.cuda.set_device(device)
        n_local = args.num_envs // self.world
        overrides = {"hfield_up_normals_only": True} if getattr(args, "hfield_up_normals_only", False) else None
        if getattr(args, "cone", None):
            overrides = dict(overrides or {}, cone=args.cone)
        extra = {}
        if getattr(args, "xml", None):      # a robot of one's own (reference README.md:74-85 "Adding a new robot"): its MJCF instead of a shipped task
            extra["xml_path"] = args.xml
        self.env = available_envs[args.env](task=args.task, num_envs=n_local, device=device, env_id_offset=self.rank * n_local, config_overrides=overrides, **extra)
        self.action_size = self.env.action_size
        self.obs_size = int(self.env.observation_size["state"][0])
        # one generator per (seed, rank, stream): stream 0 = training envs, 1 = evaluation envs (ppo/train.py)
        self.randomizer = None if args.no_randomize else (lambda env, stream=0: env.randomize(np.random.default_rng([args.seed, self.rank, stream])))
        self.ppo = ppo_train
        print(f"Observation size: {self.obs_size}")

    def progress_callback(self, num_steps, metrics):   # rank 0 only (ppo/train.py)
        for name, value in metrics.items():   # reference common/runner.py:58-60
            self.writer.add_scalar(name, value, num_steps)
        self.writer.flush()
        print("-----------")
        print(f'STEP: {num_steps} reward: {metrics.get("eval/episode_reward")} reward_std: {metrics.get("eval/episode_reward_std")}'
              f' sps: {metrics.get("training/sps", 0.0):.0f}')
        print("-----------")

    def policy_params_fn(self, current_step, net):
        d = datetime.now().strftime("%Y_%m_%d_%H%M%S")
        path = f"{self.output_dir}/{d}_{current_step}.pt"
        print(f"Saving checkpoint (step: {current_step}): {path}")
        self.ppo.save_checkpoint(path, net)
        from .export_onnx import export_onnx   # reference common/runner.py:77-84
        export_onnx(net, output_path=f"{self.output_dir}/{d}_{current_step}.onnx")

    def train(self):
        self.writer = None
        if self.rank == 0:   # one writer, one output directory: the other ranks only train
            os.makedirs(self.output_dir, exist_ok=True)
            from .tb_writer import SummaryWriter
            self.writer = SummaryWriter(self.output_dir)   # reference common/runner.py:38-39 (tensorboardX)
        try:
            return self.ppo.train(self.env, num_timesteps=self.args.num_timesteps, progress_fn=self.progress_callback,
                                  policy_params_fn=self.policy_params_fn, restore_checkpoint_path=self.args.restore_checkpoint_path,
                                  seed=self.args.seed, randomization_fn=self.randomizer, log_path=os.path.join(self.output_dir, "metrics.jsonl"),
                                  num_envs=self.args.num_envs)
        finally:
            if self.writer is not None:
                self.writer.close()

    def close(self):
        import torch.distributed as dist
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def main():
    parser = argparse.ArgumentParser(description="Open Duck Mini Runner Script")
    parser.add_argument("--output_dir", type=str, default="checkpoints", help="Where to save the checkpoints")
    parser.add_argument("--num_timesteps", type=int, default=150000000)
    parser.add_argument("--env", type=str, default="joystick", help="env")
    parser.add_argument("--task", type=str, default="flat_terrain", help="Task to run")
    parser.add_argument("--restore_checkpoint_path", type=str, default=None, help="Resume training from this checkpoint")
    parser.add_argument("--num_envs", type=int, default=8192)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--no_randomize", action="store_true")
    parser.add_argument("--hfield_up_normals_only", action="store_true",
                        help="height-field floors: count a prism pair's contacts only when the normal points up (BUILD-DEFINED opt-in, DESIGN 2; "
                             "the reading under which rough_terrain_backlash trains: profiles/r4/hfield_variants.json)")
    parser.add_argument("--cone", choices=["pyramidal", "elliptic"], default=None,
                        help="friction cone of the contact solver (what <option cone=...> in the robot's XML sets; default: the model's own, pyramidal for the duck)")
    parser.add_argument("--xml", type=str, default=None,
                        help="train a robot of your own: path of its MJCF (additive; the reference's recipe is a copy of this package per robot, README.md:74-85). "
                             "The XML carries the names constants.py looks up (sites imu / left_foot / right_foot, geoms left_foot_bottom_tpu / right_foot_bottom_tpu / "
                             "floor, the 15 sensors, keyframe home); its model shape needs a compiled kernel: tools/new_shape.py <xml> prints the lines to add")
    args = parser.parse_args()
    runner = OpenDuckMiniV2Runner(args)
    try:
        runner.train()
    finally:
        runner.close()


if __name__ == "__main__":
    main()
