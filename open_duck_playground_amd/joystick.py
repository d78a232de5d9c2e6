"""Joystick task for Open Duck Mini V2 -- batched, device-resident mirror of the reference env.

Same surface as reference playground/open_duck_mini_v2/joystick.py (`Joystick`, `default_config`,
`reset`, `step`, `action_size`, `observation_size`, `dt`, `sim_dt`, `n_substeps`) and base.py
(`mj_model`, `xml_path`), but *batched and stateful on the GPU*: `reset(seed)` / `step(state, action)`
operate on all `num_envs` envs at once and every tensor in `State` is a torch device tensor.  The
brax Vmap / Episode / AutoReset wrappers (reference common/runner.py:117) are fused into the same
kernel launch (config keys `episode_length`, `autoreset`).  All arithmetic happens in csrc/ (HIP).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import numpy as np

from . import constants, engine

USE_IMITATION_REWARD = True       # reference joystick.py:45
USE_MOTOR_SPEED_LIMITS = True     # reference joystick.py:46


class ConfigDict(dict):
    """Tiny ml_collections.ConfigDict stand-in: attribute + item access, nested."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def default_config() -> ConfigDict:
    """reference joystick.py:49-102, key for key."""
    C = ConfigDict
    return C(
        ctrl_dt=0.02, sim_dt=0.002, episode_length=1000, action_repeat=1, action_scale=0.25, dof_vel_scale=0.05, history_len=0,
        soft_joint_pos_limit_factor=0.95, max_motor_velocity=5.24,
        noise_config=C(level=1.0, action_min_delay=0, action_max_delay=3, imu_min_delay=0, imu_max_delay=3,
                       scales=C(hip_pos=0.03, knee_pos=0.05, ankle_pos=0.08, joint_vel=2.5, gravity=0.1, linvel=0.1, gyro=0.1, accelerometer=0.05)),
        reward_config=C(scales=C(tracking_lin_vel=2.5, tracking_ang_vel=6.0, torques=-1.0e-3, action_rate=-0.5, stand_still=-0.2, alive=20.0, imitation=1.0),
                        tracking_sigma=0.01),
        push_config=C(enable=True, interval_range=[5.0, 10.0], magnitude_range=[0.1, 1.0]),
        lin_vel_x=[-0.15, 0.15], lin_vel_y=[-0.2, 0.2], ang_vel_yaw=[-1.0, 1.0], neck_pitch_range=[-0.34, 1.1], head_pitch_range=[-0.78, 0.78],
        head_yaw_range=[-1.5, 1.5], head_roll_range=[-0.5, 0.5], head_range_factor=1.0,
    )


def _merge(cfg: ConfigDict, overrides: Optional[Dict[str, Any]]) -> ConfigDict:
    cfg = copy.deepcopy(cfg)
    for k, v in (overrides or {}).items():
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


REWARD_SLOTS = ("tracking_lin_vel", "tracking_ang_vel", "torques", "action_rate", "stand_still", "alive", "imitation")


def to_engine_config(cfg: ConfigDict, autoreset: bool = True, lanes_per_env: int = 0, standing: bool = False,
                     reward_slots=REWARD_SLOTS, use_imitation: bool = USE_IMITATION_REWARD,
                     use_motor_speed_limits: bool = USE_MOTOR_SPEED_LIMITS, joints_order_no_head=None) -> engine.EnvConfig:
    """reference config -> odk_env_config (include/odk.h).  `joints_order_no_head`: the robot's leg joints in actuator order (what a new robot's
    constants.py sets, reference README.md:74-85); default: the duck's."""
    c = engine.default_config(standing)
    c.ctrl_dt, c.action_scale, c.dof_vel_scale = cfg.ctrl_dt, cfg.action_scale, cfg.dof_vel_scale
    c.max_motor_velocity = cfg.get("max_motor_velocity", 0.0)
    n = cfg.noise_config
    if (n.action_min_delay, n.action_max_delay, n.imu_min_delay, n.imu_max_delay) != (0, 3, 0, 3):
        raise ValueError("the kernels implement the reference's delay ring of depth 3")
    c.noise_level, c.noise_gyro, c.noise_accelerometer = n.level, n.scales.gyro, n.scales.accelerometer
    c.noise_gravity, c.noise_joint_vel = n.scales.gravity, n.scales.joint_vel
    # BUG-COMPAT (joystick.py:184-200): indices taken on the 10-entry JOINTS_ORDER_NO_HEAD, written into the 14-entry array
    scale = np.zeros(16, np.float32)
    for idx, j in enumerate((constants.JOINTS_ORDER_NO_HEAD if joints_order_no_head is None else list(joints_order_no_head))[:16]):
        scale[idx] = n.scales.hip_pos if "_hip" in j else (n.scales.knee_pos if "_knee" in j else n.scales.ankle_pos)
    for i in range(16):
        c.qpos_noise_scale[i] = float(scale[i])
    s = cfg.reward_config.scales
    for i, k in enumerate(reward_slots):
        c.reward_scales[i] = float(s[k]) if k is not None else 0.0
    c.tracking_sigma = cfg.reward_config.tracking_sigma
    c.push_enable = 1.0 if cfg.push_config.enable else 0.0
    for i in range(2):
        c.push_interval_range[i] = cfg.push_config.interval_range[i]
        c.push_magnitude_range[i] = cfg.push_config.magnitude_range[i]
    f = cfg.head_range_factor
    zero = [0.0, 0.0]   # Standing has no move command (standing.py:652-654)
    ranges = [cfg.get("lin_vel_x", zero), cfg.get("lin_vel_y", zero), cfg.get("ang_vel_yaw", zero), [cfg.neck_pitch_range[0] * f, cfg.neck_pitch_range[1] * f],
              [cfg.head_pitch_range[0] * f, cfg.head_pitch_range[1] * f], [cfg.head_yaw_range[0] * f, cfg.head_yaw_range[1] * f],
              [cfg.head_roll_range[0] * f, cfg.head_roll_range[1] * f]]
    for i, r in enumerate(ranges):
        c.cmd_range[i][0], c.cmd_range[i][1] = float(r[0]), float(r[1])
    c.use_imitation = int(use_imitation)
    c.use_motor_speed_limits = int(use_motor_speed_limits)
    c.autoreset = int(autoreset)
    c.episode_length = int(cfg.episode_length)
    c.n_substeps = int(round(cfg.ctrl_dt / cfg.sim_dt))
    c.lanes_per_env = lanes_per_env
    # additive key (not in the reference's config): on a height-field floor, prism contacts count only when their normal points up
    c.hfield_up_normals_only = int(bool(cfg.get("hfield_up_normals_only", False)))
    return c


@dataclass
class State:
    """mjx_env.State counterpart (reference joystick.py:321): tensors are views of the engine's output buffers,
    valid until the next reset/step call."""
    data: Any                       # the engine Batch (physics state lives on the device)
    obs: Dict[str, Any]             # {"state": [N,101], "privileged_state": [N,212]}
    reward: Any                     # [N]
    done: Any                       # [N]
    metrics: Dict[str, Any]         # reward/* cost/* swing_peak, each [N]
    info: Dict[str, Any] = field(default_factory=dict)   # {"truncation": [N]} + every other key of the reference's info, fetched on first use (_Info)


class _Info(dict):
    """`State.info` (reference joystick.py:278-302 + the wrappers' additions): "truncation" is the engine's output tensor; every
    other key of the reference's dict -- rng, step, command, last_act, last_last_act, last_last_last_act, motor_targets,
    feet_air_time, last_contact, swing_peak, push, push_step, push_interval_steps, action_history, imu_history, imitation_i, and
    steps / episode_done / episode_metrics/* -- lives in the engine's per-env record on the device and is copied out ON FIRST USE
    (one synchronous read of the records per State, `Batch.info()` -> `odk_record_field`): reading `info["command"]` works like
    in the reference, it is just not free.  Every way of reading a dict triggers that fetch: `[]`, `get`, `in`, iteration, `keys` /
    `values` / `items`, `len`, `dict(info)`, `copy`.  Values are tensors on the env's device, [N, ...]; `last_contact` is bool [N, 2].

    The reference's info is an immutable snapshot; the engine is stateful and keeps ONE set of records.  A State whose info has not
    been fetched yet when the env moves on (a later `reset` / `step` / `set_records`) can no longer produce its own step's values:
    reading it then raises `RuntimeError` instead of silently returning the newer step's (fetch what you need before stepping on).
    Writing goes through `Batch.info()` / `set_records` (the reference's functional update has no counterpart on a stateful engine)."""

    def __init__(self, batch, truncation):
        super().__init__(truncation=truncation)
        self._batch = batch
        self._generation = batch.generation
        self._fetched = False

    def _fetch(self):
        if self._fetched:
            return
        if self._batch.generation != self._generation:
            raise RuntimeError("State.info: the env has been stepped / reset since this State was made and its info was never read; "
                               "the engine holds only the current records (read info before the next step)")
        import torch
        I = self._batch.info()
        dev = dict.__getitem__(self, "truncation").device
        for k in self._batch.INFO_FIELDS:
            if k == "truncation":
                continue
            v = self._batch.last_contact_bool(I) if k == "last_contact" else I[k]
            dict.__setitem__(self, k, torch.from_numpy(np.ascontiguousarray(v)).to(dev))
        self._fetched = True

    def __missing__(self, key):
        if key in self._batch.INFO_FIELDS:
            self._fetch()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._batch.INFO_FIELDS

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __iter__(self):
        self._fetch()
        return dict.__iter__(self)

    def __len__(self):
        self._fetch()
        return dict.__len__(self)

    def keys(self):
        self._fetch()
        return dict.keys(self)

    def values(self):
        self._fetch()
        return dict.values(self)

    def items(self):
        self._fetch()
        return dict.items(self)

    def copy(self):
        self._fetch()
        return dict(dict.items(self))


class Joystick:
    """Track a joystick command (reference joystick.py:105)."""

    METRIC_NAMES = engine.METRIC_NAMES

    def __init__(self, task: str = "flat_terrain", config: Optional[ConfigDict] = None, config_overrides: Optional[Dict[str, Any]] = None,
                 num_envs: int = 8192, device: int = 0, autoreset: bool = True, lanes_per_env: int = 0, env_id_offset: int = 0,
                 xml_path: Optional[str] = None, model=None):
        """`xml_path` / `model` (additive): a robot of one's own instead of a shipped task -- the MJCF (compiled by mjcf.py) or a compiled `Model`.
        The reference's recipe for a new robot (README.md:74-85) copies base.py / constants.py / joystick.py and edits names; here the names the
        reference looks up (constants.py: sites `imu`, `left_foot`, `right_foot`; geoms `left_foot_bottom_tpu`, `right_foot_bottom_tpu`, `floor`;
        the 15 sensors; keyframe `home`) are what the XML must carry, and the index tables of base.py:63-125 / joystick.py:121-200 come out of the
        compiled model.  Such a robot runs the Joystick task without the imitation reward (the reference-motion table is the duck's)."""
        self._config = _merge(config if config is not None else self._default_config(), config_overrides)
        if model is not None:
            self._model = model
        elif xml_path is not None:
            from .model import Model
            self._model = Model.from_xml(xml_path, sim_dt=float(self._config.sim_dt))
        else:
            self._model = constants.task_to_model(task)      # KeyError for unknown task names
        self._robot = constants.robot_of(self._model)
        cone = self._config.get("cone", None)
        if cone is not None:      # BUILD-DEFINED switch (the reference edits the XML's <option cone=...>): "pyramidal" | "elliptic", optional "impratio"
            if cone not in ("pyramidal", "elliptic"):
                raise ValueError(f"config cone = {cone!r}: 'pyramidal' or 'elliptic'")
            import numpy as _np
            from .model import Model
            edit = {"opt_cone": _np.array([1 if cone == "elliptic" else 0], _np.int32)}
            if self._config.get("impratio", None) is not None:
                edit["opt_impratio"] = _np.array([float(self._config.get("impratio"))], _np.float64)
            self._model = Model({**self._model.a, **edit}, xml_path=self._model.xml_path)
        # (lanes_per_env is a hint: a model with elliptic cones -- from this switch or from its own XML --, a height field or a robot that is not the
        # duck runs 32 lanes per env whatever is asked here; `batch.lanes_per_env` reports it)
        self._task = task
        self.num_envs = int(num_envs)
        self._env_id_offset = int(env_id_offset)
        self._batch = engine.Batch(self._model, self.num_envs, self._engine_config(autoreset, lanes_per_env), device=device)

    def _default_config(self) -> ConfigDict:
        return default_config()

    def _engine_config(self, autoreset: bool, lanes_per_env: int) -> engine.EnvConfig:
        return to_engine_config(self._config, autoreset, lanes_per_env, use_imitation=USE_IMITATION_REWARD and self._robot.is_open_duck,
                                joints_order_no_head=self._robot.joints_order_no_head)

    # ---- reference accessors (base.py:277-291, MjxEnv)
    @property
    def xml_path(self) -> str: return self._model.xml_path
    @property
    def action_size(self) -> int: return self._model.nu
    @property
    def mj_model(self): return self._model
    @property
    def mjx_model(self): return self._batch
    @property
    def observation_size(self): return {"state": (self._batch.nobs,), "privileged_state": (self._batch.npriv,)}
    @property
    def dt(self) -> float: return self._config.ctrl_dt
    @property
    def sim_dt(self) -> float: return self._config.sim_dt
    @property
    def n_substeps(self) -> int: return int(round(self._config.ctrl_dt / self._config.sim_dt))
    @property
    def unwrapped(self): return self
    @property
    def batch(self) -> "engine.Batch": return self._batch

    def make_eval_env(self, num_envs: int = 128):
        """Sibling env for the Evaluator (brax wraps `eval_env or environment` a second time with num_eval_envs)."""
        # a few hundred envs leave most SIMDs empty, so an evaluation of 1000 sequential steps is bound by ONE wave's latency:
        # 64 lanes per env (one env per wave) finish a step 7 % sooner than two envs sharing a wave (0.194 vs 0.209 ms at 128 envs)
        return type(self)(task=self._task, config=self._config, num_envs=num_envs, device=self._batch.device, autoreset=True,
                          lanes_per_env=64 if num_envs <= 1024 else 0, env_id_offset=1 << 24, model=self._model)

    def randomize(self, rng: np.random.Generator):
        """randomization_fn hook of brax ppo.train (reference runner.py:26, common/runner.py:108)."""
        from . import randomize
        fields, _ = randomize.domain_randomize(self._model, rng, self.num_envs)
        randomize.apply(self._batch, fields)
        return fields

    def _state(self) -> State:
        b = self._batch
        metrics = {name: b.metrics[:, i] for i, name in enumerate(self.METRIC_NAMES) if name is not None}
        return State(data=b, obs={"state": b.obs, "privileged_state": b.priv}, reward=b.reward, done=b.done, metrics=metrics,
                     info=_Info(b, b.truncation))

    def reset(self, rng: int) -> State:
        """reference joystick.py:206: `rng` is an integer seed; env e draws from key(seed, env_id_offset + e)."""
        self._batch.reset(int(rng), self._env_id_offset)
        return self._state()

    def step(self, state: State, action) -> State:
        """reference joystick.py:323 (+ Episode/AutoReset wrappers): one fused kernel launch."""
        self._batch.step(action)
        return self._state()
