"""Run this where `pip install mujoco mujoco-mjx jax playground` is possible (NOT in the build container) to close
the physics parity gap: dumps (qpos, qvel, qacc_warmstart, ctrl) -> (qpos', qvel', sensordata,
actuator_force, contact.dist, qacc) of `mjx.step` on JAX-CPU for ~100 states of each scene into
tests/golden/mjx_step_<task>.npz.  tests/test_oracle_physics.py picks the files up when they exist."""
import sys

import numpy as np


def main(ref_root: str):
    import jax
    import jax.numpy as jp
    import mujoco
    from mujoco import mjx
    jax.config.update("jax_platform_name", "cpu")
    xml_dir = f"{ref_root}/playground/open_duck_mini_v2/xmls"
    for task, scene in (("flat_terrain", "scene_flat_terrain.xml"), ("flat_terrain_backlash", "scene_flat_terrain_backlash.xml")):
        m = mujoco.MjModel.from_xml_path(f"{xml_dir}/{scene}")
        m.opt.timestep = 0.002
        mx = mjx.put_model(m)
        rng = np.random.default_rng(0)
        home = m.keyframe("home").qpos
        rows = {k: [] for k in ("qpos", "qvel", "warm", "ctrl", "qpos1", "qvel1", "sensordata", "actuator_force", "dist", "qacc")}
        step = jax.jit(mjx.step)
        for i in range(100):
            qpos = home.copy(); qpos[2] = rng.uniform(0.135, 0.4); qpos[7:] += rng.uniform(-0.2, 0.2, m.nq - 7)
            qvel = rng.normal(0, 0.5, m.nv); warm = rng.normal(0, 2.0, m.nv)
            ctrl = m.keyframe("home").ctrl + rng.uniform(-0.3, 0.3, m.nu)
            d = mjx.make_data(mx).replace(qpos=jp.array(qpos), qvel=jp.array(qvel), qacc_warmstart=jp.array(warm), ctrl=jp.array(ctrl))
            d1 = step(mx, d)
            for k, v in (("qpos", qpos), ("qvel", qvel), ("warm", warm), ("ctrl", ctrl), ("qpos1", d1.qpos), ("qvel1", d1.qvel),
                         ("sensordata", d1.sensordata), ("actuator_force", d1.actuator_force), ("dist", d1.contact.dist), ("qacc", d1.qacc)):
                rows[k].append(np.asarray(v))
        np.savez(f"tests/golden/mjx_step_{task}.npz", **{k: np.stack(v) for k, v in rows.items()})
        print(task, "dumped")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
