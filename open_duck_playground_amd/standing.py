"""Standing task for Open Duck Mini V2 -- batched, device-resident mirror of reference
playground/open_duck_mini_v2/standing.py (`Standing`:103, `default_config`:44-100).

Same fused kernel as `Joystick` with `env_kind = ODK_ENV_STANDING`: no imitation reward / phase, no motor speed
limit (standing.py:42,377-380), observation without motor targets (85 floats, :524-540; privileged 153, :548-565),
rewards orientation / torques / action_rate / alive / stand_still(ignore_head) / head_pos (:585-606), no move
command (:652-654), base-velocity reset noise +-0.5 (:247).
"""
from __future__ import annotations

from . import engine
from .joystick import ConfigDict, Joystick, State, to_engine_config  # noqa: F401

USE_IMITATION_REWARD = False      # reference standing.py:42

# reward slot order of the engine for this env kind (include/odk.h: odk_env_config.reward_scales)
REWARD_SLOTS = ("orientation", "head_pos", "torques", "action_rate", "stand_still", "alive", None)
METRIC_NAMES = ("cost/orientation", "cost/head_pos", "cost/torques", "cost/action_rate", "cost/stand_still", "reward/alive", None, "swing_peak")


def default_config() -> ConfigDict:
    """reference standing.py:44-100, key for key."""
    C = ConfigDict
    return C(
        ctrl_dt=0.02, sim_dt=0.002, episode_length=1000, action_repeat=1, action_scale=0.25, dof_vel_scale=0.05, history_len=0,
        soft_joint_pos_limit_factor=0.95,
        noise_config=C(level=1.0, action_min_delay=0, action_max_delay=3, imu_min_delay=0, imu_max_delay=3,
                       scales=C(hip_pos=0.03, knee_pos=0.05, ankle_pos=0.08, joint_vel=2.5, gravity=0.1, linvel=0.1, gyro=0.05, accelerometer=0.005)),
        reward_config=C(scales=C(orientation=-0.5, torques=-1.0e-3, action_rate=-0.375, stand_still=-0.3, alive=20.0, head_pos=-2.0),
                        tracking_sigma=0.01),
        push_config=C(enable=True, interval_range=[5.0, 10.0], magnitude_range=[0.1, 1.0]),
        neck_pitch_range=[-0.34, 1.1], head_pitch_range=[-0.78, 0.78], head_yaw_range=[-2.7, 2.7], head_roll_range=[-0.5, 0.5],
        head_range_factor=1.0,
    )


class Standing(Joystick):
    """Standing policy (reference standing.py:103)."""

    METRIC_NAMES = METRIC_NAMES

    def _default_config(self) -> ConfigDict:
        return default_config()

    def _engine_config(self, autoreset: bool, lanes_per_env: int) -> engine.EnvConfig:
        return to_engine_config(self._config, autoreset, lanes_per_env, standing=True, reward_slots=REWARD_SLOTS,
                                use_imitation=USE_IMITATION_REWARD, use_motor_speed_limits=False)
