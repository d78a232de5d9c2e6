"""Domain randomization (host mirror of reference playground/common/randomize.py:26-146).

`domain_randomize(model, rng, num_envs)` samples, once per env for the whole run, the same fields
with the same distributions as the reference and returns them with an `in_axes`-style dict that
names the batched fields.  `apply(batch, fields)` hands them to the engine (odk_batch_set_param).

Bug-compatibility kept on purpose:
  * FLOOR_GEOM_ID = 0 is a visual trunk mesh, so the sampled friction has no physical effect: it is
    returned but not sent to the engine;
  * TORSO_BODY_ID = 1 is the massless `base` body: its ipos is jittered and its mass becomes
    0 * U(0.9,1.1) + U(-0.1,0.1), which can be negative; inertias and invweights are not updated.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import engine
from .model import Model

FLOOR_GEOM_ID = 0
TORSO_BODY_ID = 1


def domain_randomize(model: Model, rng: np.random.Generator, num_envs: int) -> Tuple[Dict[str, np.ndarray], Dict[str, int]]:
    a = model.a
    nu, nbody = model.nu, model.nbody
    act_jnt = np.asarray(a["actuator_trnid"])
    dof_addr = np.asarray(a["jnt_dofadr"])[act_jnt]          # dofs with frictionloss (backlash joints have none)
    joint_addr = np.asarray(a["jnt_qposadr"])[act_jnt]
    U = lambda lo, hi, shape: rng.uniform(lo, hi, size=(num_envs,) + shape)
    geom_friction0 = U(0.5, 1.0, ())                                              # randomize.py:42-45 (no-op geom)
    frictionloss = np.asarray(a["dof_frictionloss"])[dof_addr][None] * U(0.9, 1.1, (nu,))   # :48-52
    armature = np.asarray(a["dof_armature"])[dof_addr][None] * U(1.0, 1.05, (nu,))          # :55-59
    dpos = U(-0.05, 0.05, (3,))                                                    # :62-66
    body_ipos_torso = np.asarray(a["body_ipos"])[TORSO_BODY_ID][None] + dpos
    body_mass = np.asarray(a["body_mass"])[None] * U(0.9, 1.1, (nbody,))           # :69-71
    body_mass[:, TORSO_BODY_ID] += U(-0.1, 0.1, ())                                # :74-76
    qpos0 = np.asarray(a["qpos0"])[joint_addr][None] + U(-0.03, 0.03, (nu,))       # :79-86
    factor = U(0.9, 1.1, (nu,))                                                    # :89-95
    kp = np.asarray(a["actuator_gainprm0"])[None] * factor
    fields = {
        "geom_friction": geom_friction0, "body_ipos": body_ipos_torso, "dof_frictionloss": frictionloss, "dof_armature": armature,
        "body_mass": body_mass, "qpos0": qpos0, "actuator_gainprm": kp, "actuator_biasprm": -kp,
    }
    in_axes = {k: 0 for k in fields}   # every returned field is batched over envs (randomize.py:119-131)
    return fields, in_axes


def apply(batch: "engine.Batch", fields: Dict[str, np.ndarray]) -> None:
    batch.set_param(engine.PARAM_BODY_MASS, fields["body_mass"])
    batch.set_param(engine.PARAM_BODY_IPOS_TORSO, fields["body_ipos"])
    batch.set_param(engine.PARAM_DOF_FRICTIONLOSS, fields["dof_frictionloss"])
    batch.set_param(engine.PARAM_DOF_ARMATURE, fields["dof_armature"])
    batch.set_param(engine.PARAM_QPOS0, fields["qpos0"])
    batch.set_param(engine.PARAM_KP, fields["actuator_gainprm"])
