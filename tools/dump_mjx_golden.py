"""Run this where `pip install mujoco mujoco-mjx jax playground` is possible (NOT in the build container: no network, nothing of
that stack installed) to close the physics parity gap.  Written for this build, not taken from the reference.

    python tools/dump_mjx_golden.py /path/to/Open_Duck_Playground [out_dir = tests/golden]

For each of the three scenes (flat_terrain, flat_terrain_backlash, rough_terrain_backlash) it writes tests/golden/mjx_<task>.npz:

  const_*   compiler-derived model constants the build's own MJCF compiler has to reproduce (mjcf.py): dof_invweight0,
            body_invweight0, stat.meaninertia, body_mass / ipos / iquat / inertia, qpos0, the collision geoms' pos / quat /
            size / rbound (the recentred foot mesh frame), the foot mesh's vertices, the height field's size and samples
  fwd_*     `mjx_env.init` = mjx.forward (no integration) on 100 states, 60 of them contact-rich (feet pressed 0.3 ... 3 mm into
            the floor / terrain, some with the feet pressed against each other): qacc_smooth, qacc, sensordata, actuator_force,
            contact dist / pos / frame / geom, efc_force
  step_*    one mjx.step from the same states: qpos', qvel', qacc_warmstart'
  env10_*   mjx_env.step semantics (ctrl held, 10 x mjx.step) from 40 of the states

All arithmetic on JAX-CPU in float32 (the reference never enables x64: common/runner.py:47-54).  Loaders: tests/test_mjx_golden.py
(oracle, CPU) and tests/test_gpu_mjx_golden.py (HIP path, directly); both skip while the files are absent."""
import os
import sys

import numpy as np

SCENES = (("flat_terrain", "scene_flat_terrain.xml"), ("flat_terrain_backlash", "scene_flat_terrain_backlash.xml"),
          ("rough_terrain_backlash", "scene_rough_terrain_backlash.xml"))


def _states(m, mujoco, rng, n=100, n_contact=60):
    """random states around the home keyframe; the first n_contact are lowered until the deepest contact is 0.3 ... 3 mm"""
    home_q = np.array(m.keyframe("home").qpos); home_c = np.array(m.keyframe("home").ctrl)
    d = mujoco.MjData(m)
    qs, vs, ws, cs = [], [], [], []
    hip_roll = [i for i in range(m.njnt) if "hip_roll" in (m.joint(i).name or "") and "backlash" not in m.joint(i).name]
    for i in range(n):
        q = home_q.copy()
        q[0:2] += rng.uniform(-0.5, 0.5, 2) if i % 3 else 0.0
        ang = rng.uniform(-0.2, 0.2); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        q[3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        for j in range(1, m.njnt):
            a = m.jnt_qposadr[j]; lo, hi = m.jnt_range[j]
            q[a] = rng.uniform(lo, hi) if hi - lo < 0.05 else np.clip(q[a] + rng.uniform(-0.25, 0.25), lo - 0.01, hi + 0.01)
        if i % 10 == 9 and len(hip_roll) == 2:      # feet pressed against each other (mesh-mesh contacts), off the floor
            q[m.jnt_qposadr[hip_roll[0]]] = rng.uniform(0.4, 0.6); q[m.jnt_qposadr[hip_roll[1]]] = rng.uniform(-0.6, -0.4); q[2] = 0.3
        elif i < n_contact:
            q[2] = 0.2
            target = rng.uniform(3e-4, 3e-3)
            for _ in range(6):                      # lower the base until the deepest contact is `target` deep (MuJoCo-C distances)
                d.qpos[:] = q; d.qvel[:] = 0; mujoco.mj_forward(m, d)
                dist = min([c.dist for c in d.contact[: d.ncon]] + [0.05])
                q[2] -= dist + target
        else:
            q[2] = rng.uniform(0.3, 0.5)
        qs.append(q); vs.append(rng.normal(0, 0.5, m.nv)); ws.append(rng.normal(0, 2.0, m.nv)); cs.append(home_c + rng.uniform(-0.3, 0.3, m.nu))
    return map(np.array, (qs, vs, ws, cs))


def main(ref_root: str, out_dir: str):
    import jax
    import jax.numpy as jp
    import mujoco
    from mujoco import mjx
    jax.config.update("jax_platform_name", "cpu")
    xml_dir = f"{ref_root}/playground/open_duck_mini_v2/xmls"
    os.makedirs(out_dir, exist_ok=True)
    for task, scene in SCENES:
        m = mujoco.MjModel.from_xml_path(f"{xml_dir}/{scene}")
        m.opt.timestep = 0.002                       # reference base.py:56
        mx = mjx.put_model(m)
        out = {}
        # ---- constants
        cg = [g for g in range(m.ngeom) if m.geom_contype[g] or m.geom_conaffinity[g]]
        out.update(const_dof_invweight0=m.dof_invweight0, const_body_invweight0=m.body_invweight0, const_meaninertia=np.array([m.stat.meaninertia]),
                   const_body_mass=m.body_mass, const_body_ipos=m.body_ipos, const_body_iquat=m.body_iquat, const_body_inertia=m.body_inertia,
                   const_qpos0=m.qpos0, const_cgeom_id=np.array(cg), const_cgeom_type=m.geom_type[cg], const_cgeom_pos=m.geom_pos[cg],
                   const_cgeom_quat=m.geom_quat[cg], const_cgeom_size=m.geom_size[cg], const_cgeom_rbound=m.geom_rbound[cg],
                   const_cgeom_bodyid=m.geom_bodyid[cg], const_dof_armature=m.dof_armature, const_dof_damping=m.dof_damping,
                   const_dof_frictionloss=m.dof_frictionloss, const_jnt_range=m.jnt_range, const_actuator_gainprm0=m.actuator_gainprm[:, 0],
                   const_actuator_biasprm=m.actuator_biasprm[:, :3], const_key_qpos=np.array(m.keyframe("home").qpos), const_key_ctrl=np.array(m.keyframe("home").ctrl))
        mesh_geoms = [g for g in cg if m.geom_type[g] == mujoco.mjtGeom.mjGEOM_MESH]
        if mesh_geoms:
            mid = m.geom_dataid[mesh_geoms[0]]
            out["const_foot_mesh_vert"] = m.mesh_vert[m.mesh_vertadr[mid]: m.mesh_vertadr[mid] + m.mesh_vertnum[mid]]
        if m.nhfield:
            out["const_hfield_size"] = m.hfield_size[0]; out["const_hfield_data"] = m.hfield_data.reshape(m.hfield_nrow[0], m.hfield_ncol[0])
        # ---- states
        rng = np.random.default_rng(0)
        qpos, qvel, warm, ctrl = _states(m, mujoco, rng)
        fwd, step = jax.jit(mjx.forward), jax.jit(mjx.step)
        rows = {}

        def put(prefix, **kv):
            for k, v in kv.items():
                rows.setdefault(prefix + k, []).append(np.asarray(v))
        for i in range(len(qpos)):
            d0 = mjx.make_data(mx).replace(qpos=jp.array(qpos[i], jp.float32), qvel=jp.array(qvel[i], jp.float32), qacc_warmstart=jp.array(warm[i], jp.float32),
                                           ctrl=jp.array(ctrl[i], jp.float32))
            df = fwd(mx, d0)
            put("fwd_", qacc_smooth=df.qacc_smooth, qacc=df.qacc, sensordata=df.sensordata, actuator_force=df.actuator_force, dist=df.contact.dist,
                pos=df.contact.pos, frame=df.contact.frame, geom=df.contact.geom, efc_force=df.efc_force, xpos=df.xpos, site_xpos=df.site_xpos)
            d1 = step(mx, d0)
            put("step_", qpos=d1.qpos, qvel=d1.qvel, warm=d1.qacc_warmstart)
            if i % 5 < 2:
                d = d0
                for _ in range(10):                 # mjx_env.step: the same ctrl for n_substeps
                    d = step(mx, d)
                put("env10_", index=i, qpos=d.qpos, qvel=d.qvel, warm=d.qacc_warmstart, sensordata=d.sensordata, dist=d.contact.dist)
        out.update(qpos=qpos, qvel=qvel, warm=warm, ctrl=ctrl, **{k: np.stack(v) for k, v in rows.items()})
        out["versions"] = np.array([f"mujoco {mujoco.__version__}", f"jax {jax.__version__}"])
        np.savez_compressed(os.path.join(out_dir, f"mjx_{task}.npz"), **out)
        print(task, "dumped:", len(qpos), "states")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", sys.argv[2] if len(sys.argv) > 2 else "tests/golden")
