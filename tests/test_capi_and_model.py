"""CPU checks: the C-ABI library loads and exports every symbol include/odk.h declares (no compute calls
without a GPU); model compiler / kernel tables / blob are self-consistent and match SURVEY Appendix A."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_libodk_exports_every_header_symbol():
    from open_duck_playground_amd import engine
    engine.build_library()
    lib = ctypes.CDLL(engine.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "odk.h")).read()
    declared = sorted(set(re.findall(r"\b(odk_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libodk.so does not export {name}"
    assert set(engine.EXPORTED_SYMBOLS) <= set(declared)


def test_engine_refuses_to_run_without_gpu(model_a):
    import torch
    from open_duck_playground_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.OdkError):
        engine.Batch(model_a, 4)


def test_default_config_matches_reference_defaults():
    from open_duck_playground_amd import engine
    c = engine.default_config()   # odk_default_config == joystick.py:49-102
    assert (c.ctrl_dt, c.action_scale, c.dof_vel_scale) == pytest.approx((0.02, 0.25, 0.05))
    assert c.max_motor_velocity == pytest.approx(5.24)
    assert list(c.reward_scales) == pytest.approx([2.5, 6.0, -1e-3, -0.5, -0.2, 20.0, 1.0])
    assert list(c.qpos_noise_scale)[:14] == pytest.approx([0.03, 0.03, 0.03, 0.05, 0.08, 0.03, 0.03, 0.03, 0.05, 0.08, 0, 0, 0, 0])  # BUG-COMPAT
    assert (c.episode_length, c.n_substeps, c.autoreset) == (1000, 10, 1)


def test_model_dimension_table(model_a, model_b):
    # SURVEY.md Appendix A
    assert (model_a.nq, model_a.nv, model_a.nu, model_a.nbody, model_a.njnt) == (21, 20, 14, 18, 15)
    assert (model_b.nq, model_b.nv, model_b.nu, model_b.nbody, model_b.njnt) == (31, 30, 14, 18, 25)
    for m in (model_a, model_b):
        assert m.nsensordata == 46 and int(m.a["ngeom"][0]) == 47
        assert m.a["body_mass"].sum() == pytest.approx(2.10714, abs=1e-4)
        assert m.geom_id("floor") == 46 and m.geom_id("left_foot_bottom_tpu") == 18 and m.geom_id("right_foot_bottom_tpu") == 43
        assert list(m.a["cgeom_vertnum"]) == [17, 17, 0] and list(m.a["cgeom_facenum"]) == [30, 30, 0]
        assert m.a["opt_iterations"][0] == 1 and m.a["opt_ls_iterations"][0] == 5 and m.a["opt_eulerdamp"][0] == 0
        assert m.a["opt_timestep"][0] == pytest.approx(0.002)
        assert m.body_id("base") == 1 and m.a["body_mass"][1] == 0.0
        assert list(m.a["sensor_adr"]) == [0, 3, 6, 9, 12, 15, 18, 21, 24, 28, 31, 34, 37, 40, 43]
        np.testing.assert_allclose(m.a["cgeom_friction"][2], [0.6, 0.005, 0.0001])
    assert model_a.a["actuator_gainprm0"][0] == pytest.approx(13.37) and model_b.a["actuator_gainprm0"][0] == pytest.approx(17.11)
    np.testing.assert_allclose(model_a.a["key_qpos"][:7], [0, 0, 0.15, 1, 0, 0, 0])
    np.testing.assert_allclose(model_a.a["actuator_ctrlrange"], model_a.a["jnt_range"][1:])   # inheritrange="1"


def test_blob_roundtrip_and_tables(model_a, model_b):
    from open_duck_playground_amd.model import unpack_blob
    from open_duck_playground_amd.tables import build_kernel_tables
    for m in (model_a, model_b):
        arrays = unpack_blob(m.blob())
        np.testing.assert_array_equal(arrays["body_parentid"], m.a["body_parentid"])
        np.testing.assert_allclose(arrays["qpos0"], m.a["qpos0"])
        t = build_kernel_tables(m.a)
        nv = m.nv
        depth, adr, Mi, Mj = t["k_dof_depth"], t["k_dof_Madr"], t["k_M_i"], t["k_M_j"]
        assert int(t["k_nM"][0]) == int((depth + 1).sum()) == len(Mi)
        parent = m.a["dof_parentid"]
        for i in range(nv):
            # row i lists its ancestors root-first; the diagonal sits at adr + depth
            assert Mi[adr[i] + depth[i]] == i and Mj[adr[i] + depth[i]] == i
            j, c = i, depth[i]
            while j >= 0:
                assert Mj[adr[i] + c] == j
                j, c = parent[j], c - 1
            anc = {int(Mj[adr[i] + c]) for c in range(depth[i])}
            assert {b for b in range(nv) if (int(t["k_dof_ancmask"][i]) >> b) & 1} == anc
        for i in range(nv):
            desc = {k for k in range(nv) if (int(t["k_dof_ancmask"][k]) >> i) & 1}
            assert {b for b in range(nv) if (int(t["k_dof_descmask"][i]) >> b) & 1} == desc
        # virtual tree: every kinematic ancestor stays an ancestor; the right leg additionally hangs below the left leg
        for i in range(nv):
            assert (int(t["k_vdof_ancmask"][i]) & int(t["k_dof_ancmask"][i])) == int(t["k_dof_ancmask"][i])
        src = t["k_H_src"]
        for p, (i, j) in enumerate(zip(t["k_H_i"], t["k_H_j"])):
            if src[p] >= 0:
                assert (Mi[src[p]], Mj[src[p]]) == (i, j)
        assert (t["k_foot_dofmask"].sum(axis=1) == [6 + (nv - 6 - 4) // 2] * 2).all()


def test_twin_dof_reduction_of_the_loader_matches_the_python_mirror(model_a, model_b):
    """csrc build_reduced_tables (host side of the kernels' reduced tree) == tables.reduced_layout; the backlash model
    reduces to the 20-dof robot's own tree (145 / 170 entries) and both LDS images fit 8 workgroups per CU."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.tables import build_kernel_tables, reduced_layout
    for m, paired in ((model_a, 0), (model_b, 1)):
        got, ref = engine.model_reduction(m), reduced_layout(m.a)
        assert got["paired"] == paired and got["nvr"] == 20 and got["nMr"] == ref["nnz"] == 145 and got["nHr"] == 170
        assert got["main"] == list(ref["main"]) and got["twin"] == list(ref["twin"])
        assert got["env_lds_floats"] * 4 * 2 <= 160 * 1024 // 8          # two envs per single-wave workgroup, 8 workgroups per CU
    ref_b = reduced_layout(model_b.a)
    assert list(ref_b["twin"][6:11]) == [7, 9, 11, 13, 15] and list(ref_b["twin"][11:15]) == [-1] * 4   # legs have twins, the head has none
    # reduced entries address the inertia element of the main dofs; a pair's diagonal is the (twin, main) element (no armature)
    ta = build_kernel_tables(model_a.a)
    ref_a = reduced_layout(model_a.a)
    assert np.array_equal(ref_a["ei"], ta["k_M_i"]) and np.array_equal(ref_a["ej"], ta["k_M_j"])
    for p, (i, j) in enumerate(zip(ref_b["ei"], ref_b["ej"])):
        assert ref_b["kind"][j] != 2 and (ref_b["kind"][i] != 2 or i == j + 1)


def test_unknown_task_and_env_errors():
    from open_duck_playground_amd import constants
    with pytest.raises(KeyError):     # reference constants.py:28-34
        constants.task_to_model("no_such_task")
    with pytest.raises(KeyError):     # rough_terrain's XML is missing in the reference too (constants.py:23)
        constants.task_to_model("rough_terrain")


def test_domain_randomize_host_mirror(model_b):
    from open_duck_playground_amd import randomize
    f, axes = randomize.domain_randomize(model_b, np.random.default_rng(0), 256)
    assert set(axes) == {"geom_friction", "body_ipos", "dof_frictionloss", "dof_armature", "body_mass", "qpos0", "actuator_gainprm", "actuator_biasprm"}
    assert f["body_mass"].shape == (256, 18) and f["dof_frictionloss"].shape == (256, 14)
    assert ((f["dof_frictionloss"] >= 0.068 * 0.9 - 1e-9) & (f["dof_frictionloss"] <= 0.068 * 1.1 + 1e-9)).all()
    assert ((f["dof_armature"] >= 0.027 - 1e-9) & (f["dof_armature"] <= 0.027 * 1.05 + 1e-9)).all()
    assert (np.abs(f["body_mass"][:, 1]) <= 0.1).all() and (f["body_mass"][:, 1] < 0).any()   # BUG-COMPAT: massless base gets +-0.1 kg
    np.testing.assert_allclose(f["actuator_biasprm"], -f["actuator_gainprm"])
    assert (np.abs(f["qpos0"]) <= 0.03 + 1e-12).all()


def test_standing_config_matches_reference_defaults():
    """odk_default_config_standing / standing.default_config == reference standing.py:44-100 (+ :42, :377-380, :652-654)."""
    from open_duck_playground_amd import engine, joystick, standing
    c = engine.default_config(standing=True)
    assert c.env_kind == 1 and c.reset_base_qvel == pytest.approx(0.5)
    assert (c.noise_gyro, c.noise_accelerometer) == pytest.approx((0.05, 0.005))
    assert list(c.reward_scales) == pytest.approx([-0.5, -2.0, -1e-3, -0.375, -0.3, 20.0, 0.0])   # orientation, head_pos, torques, action_rate, stand_still, alive
    assert c.use_imitation == 0 and c.use_motor_speed_limits == 0
    assert [list(r) for r in c.cmd_range][:3] == [[0.0, 0.0]] * 3 and list(c.cmd_range[5]) == pytest.approx([-2.7, 2.7])
    assert engine.obs_sizes(1) == (85, 153) and engine.obs_sizes(0) == (101, 212)
    cfg = standing.default_config()
    assert set(cfg.reward_config.scales) == {"orientation", "torques", "action_rate", "stand_still", "alive", "head_pos"}
    assert "max_motor_velocity" not in cfg and "lin_vel_x" not in cfg and cfg.head_yaw_range == [-2.7, 2.7]
    e = joystick.to_engine_config(cfg, standing=True, reward_slots=standing.REWARD_SLOTS, use_imitation=False, use_motor_speed_limits=False)
    for name, _t in engine.EnvConfig._fields_:
        a, b = getattr(e, name), getattr(c, name)
        flat = lambda v: [y for x in v for y in (x if hasattr(x, "__len__") else [x])] if hasattr(v, "__len__") else [v]
        assert flat(a) == pytest.approx(flat(b)), name   # the Python config and the C default agree field by field


@pytest.mark.parametrize("task,xml", [("flat_terrain", "scene_flat_terrain.xml"), ("flat_terrain_backlash", "scene_flat_terrain_backlash.xml")])
def test_shipped_assets_are_what_the_compiler_makes_from_the_reference_xml(task, xml):
    """The shipped assets/<task>.npz are data compiled from the reference's MJCF (tools/compile_models.py); where the
    reference is mounted (build container only) the compiler must reproduce them."""
    path = os.path.join("/root/reference/playground/open_duck_mini_v2/xmls", xml)
    if not os.path.exists(path):
        pytest.skip("reference not mounted (GPU box)")
    from open_duck_playground_amd import mjcf
    from open_duck_playground_amd.model import load_task_model
    fresh = mjcf.compile_mjcf(path, sim_dt=0.002)
    shipped = load_task_model(task).a
    for k in ("nq", "nv", "body_mass", "body_pos", "body_quat", "body_ipos", "jnt_axis", "jnt_range", "dof_armature", "dof_damping", "dof_frictionloss",
              "dof_invweight0", "body_invweight0", "hull_vert", "hull_face", "key_qpos", "actuator_gainprm0", "actuator_biasprm", "sensor_adr", "site_pos"):
        np.testing.assert_allclose(np.asarray(fresh[k], dtype=np.float64), np.asarray(shipped[k], dtype=np.float64), rtol=1e-12, atol=1e-14, err_msg=k)


def test_env_kernels_have_no_scratch_and_fit_two_waves_per_simd(tmp_path):
    """measured HBM traffic equals the algorithmic traffic only because no env kernel spills; the claim is
    checked on the shipped code object (kernel metadata of libodk.so), not on a recompilation."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not present")
    lib = os.path.join(ROOT, "open_duck_playground_amd", "csrc", "libodk.so")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    # one offload bundle per translation unit (engine, learner kernels, network kernels), concatenated in the section
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    kernels = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        piece, co = str(tmp_path / f"bundle{n}.bin"), str(tmp_path / f"dev{n}.co")
        open(piece, "wb").write(blob[a:b])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([tools[2], "--notes", co], text=True)
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1); kernels[name] = {}
            m = re.match(r"\s+\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|group_segment_fixed_size):\s+(\d+)", line)
            if m and name:
                kernels[name][m.group(1)] = int(m.group(2))
    env = {k: v for k, v in kernels.items() if "step_kernel" in k or "reset_kernel" in k}
    assert len(env) >= 8, sorted(kernels)          # A / B / B+hfield at 32 lanes, A at 64 lanes: step + reset each
    for k, v in env.items():
        if k.endswith("ELi32ELi2EEv5KArgs"):       # sphere / capsule feet on a height field (8(f).3): two out-of-line calls, no traffic claim rests on it
            continue
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
        assert v["vgpr_count"] <= 256, (k, v)      # 2 waves per SIMD (launch bounds 64, 2)
    # the learner's whole-network kernels run four workgroups of four waves per CU (<= 128 registers), the
    # weight-gradient kernel keeps its 128 accumulators beside <= 128 other registers; none of them spills
    mlp = {k: v for k, v in kernels.items() if "mlp_fwd_kernel" in k or "mlp_bwd_kernel" in k}
    dw = {k: v for k, v in kernels.items() if "dw_gemm_kernel" in k}
    assert len(mlp) == 2 and len(dw) == 1, sorted(kernels)
    for k, v in {**mlp, **dw}.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
    for k, v in mlp.items():
        assert v["vgpr_count"] <= 128, (k, v)


@pytest.mark.parametrize("extra, what", [
    ("<equality><tendon tendon1='t'/></equality>", "equality"),
    ("<tendon><fixed name='t'><joint joint='j' coef='1'/></fixed></tendon>", "tendon"),
    ("<contact><exclude body1='a' body2='b'/></contact>", "contact"),
])
def test_compiler_refuses_sections_the_kernels_do_not_model(tmp_path, extra, what):
    """Data-format boundary: MJCF features that would change the physics are refused, not silently dropped."""
    from open_duck_playground_amd import mjcf
    xml = f"""<mujoco><compiler angle="radian"/><worldbody>
      <body name="a"><freejoint/><inertial pos="0 0 0" mass="1" fullinertia="1 1 1 0 0 0"/>
        <body name="b" pos="0 0 0.1"><joint name="j" type="hinge" axis="0 1 0"/><inertial pos="0 0 0" mass="1" fullinertia="1 1 1 0 0 0"/></body>
      </body></worldbody>{extra}</mujoco>"""
    path = tmp_path / "m.xml"
    path.write_text(xml)
    with pytest.raises(NotImplementedError, match=what):
        mjcf.compile_mjcf(str(path))


def test_loader_takes_joint_couplings_and_refuses_other_equalities_by_name():
    """<equality> joint / connect / weld compile (mjcf.py) and the float64 oracle steps all of them (tests/test_oracle_equality.py).  The
    kernels of the third model shape model <equality><joint> rows between two hinges of one serial chain and connect / weld rows between two
    bodies of one root-to-leaf path (or a body and the world): `odk_model_load` takes those, (or the two foot chains: a closed loop), and refuses -- naming the constraint -- a
    connect / weld closing another loop, a coupling across two chains, a joint in two rows, a third row, and any equality on the duck's shapes;
    a model whose equalities are all switched off (eq_active = 0: MuJoCo's own off switch) loads."""
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    m = Model.from_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "tail_biped_equality.xml"))
    act = lambda *a: Model({**m.a, "eq_active": np.array(a, np.int32)})
    assert engine.model_reduction(m)["nvr"] == 21                         # all four (two couplings, the pinned foot, the welded tail tip): taken
    assert engine.model_reduction(act(1, 1, 0, 0))["nvr"] == 21          # the two joint couplings alone
    # a connect / weld between the two FEET closes a loop over the two leg chains: taken (the virtual tree's layout has the entries);
    # one between a foot and the tail closes a loop the Hessian's layouts have no entries for: refused by name
    for k, kind in ((2, "connect"), (3, "weld")):
        for other, ok in (("left_foot_link", True), ("tail_3", False)):
            loop = dict(m.a); o2 = np.array(m.a["eq_obj2id"], np.int32); o1 = np.array(m.a["eq_obj1id"], np.int32)
            o1[k] = m.body_id("right_foot_link"); o2[k] = m.body_id(other)
            loop["eq_obj1id"] = o1; loop["eq_obj2id"] = o2; loop["eq_active"] = np.array([0, 0, k == 2, k == 3], np.int32)
            if ok:
                assert engine.model_reduction(Model(loop))["nvr"] == 21
            else:
                with pytest.raises(engine.OdkError, match=rf"<equality><{kind}> \(constraint {k}\): the two bodies must lie on one root-to-leaf path.*another loop has no entries"):
                    engine.model_reduction(Model(loop))
    assert engine.model_reduction(act(0, 0, 0, 0))["nvr"] == 21
    # a coupling across two chains (left knee <- right knee): no entry of the tree layout
    cross = dict(m.a); cross["eq_obj2id"] = np.array([m.a["eq_obj2id"][0], m.joint_id("right_knee"), 0, 0], np.int32); cross["eq_active"] = np.array([1, 1, 0, 0], np.int32)
    with pytest.raises(engine.OdkError, match="one serial chain"):
        engine.model_reduction(Model(cross))
    # a joint in two rows
    twice = dict(m.a); twice["eq_obj1id"] = np.array([m.a["eq_obj1id"][0], m.a["eq_obj1id"][0], 0, 0], np.int32); twice["eq_active"] = np.array([1, 1, 0, 0], np.int32)
    with pytest.raises(engine.OdkError, match="at most one equality row"):
        engine.model_reduction(Model(twice))
    # the duck's shapes carry no equality code
    from open_duck_playground_amd.model import load_task_model
    duck = load_task_model("flat_terrain")
    j = lambda n: duck.joint_id(n)
    eq = dict(duck.a, eq_type=np.array([2], np.int32), eq_obj1id=np.array([j("left_ankle")], np.int32), eq_obj2id=np.array([j("left_knee")], np.int32),
              eq_active=np.array([1], np.int32), eq_data=np.array([[0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]], np.float64), eq_solref=np.array([[0.02, 1.0]]),
              eq_solimp=np.array([[0.9, 0.95, 0.001, 0.5, 2.0]]), neq=np.array([1], np.int32))
    with pytest.raises(engine.OdkError, match="third and fourth model shapes only"):
        engine.model_reduction(Model(eq))


def test_loader_takes_a_biped_with_six_dof_legs():
    """tests/assets/biped12.xml (hip yaw / roll / pitch, knee, ankle pitch / roll per leg): compiled by mjcf.py, matched by `odk_model_load` against
    the fourth instantiated shape (18 dofs, 12 actuators, 16 bodies, chains of six), which carries the optional constraint code (equality rows,
    elliptic cones) like the third."""
    import os
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model
    m = Model.from_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "biped12.xml"))
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt) == (19, 18, 12, 16, 13)
    red = engine.model_reduction(m)
    assert (red["paired"], red["nvr"], red["nMr"], red["nHr"]) == (0, 18, 135, 171)
    assert engine.model_reduction(Model({**m.a, "opt_cone": np.array([1], np.int32)}))["nvr"] == 18      # elliptic cones: compiled into this shape too
    j = m.joint_id
    eq = dict(m.a, eq_type=np.array([2], np.int32), eq_obj1id=np.array([j("left_ankle_pitch")], np.int32), eq_obj2id=np.array([j("left_knee")], np.int32),
              eq_active=np.array([1], np.int32), eq_data=np.array([[0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]], np.float64), eq_solref=np.array([[0.02, 1.0]]),
              eq_solimp=np.array([[0.9, 0.95, 0.001, 0.5, 2.0]]), neq=np.array([1], np.int32))
    assert engine.model_reduction(Model(eq))["nvr"] == 18          # a joint coupling inside one leg: taken, like the third shape's


def test_observation_sizes_and_robot_constants_follow_the_model(model_a, model_b):
    """`odk_model_obs_sizes` (what `observation_size` of the reference's env reports, joystick.py:570-615 with the robot's actuator count) and
    `constants.robot_of` (what a new robot's constants.py spells out, reference README.md:74-85), host-only: the duck 101 / 212 and 85 / 153,
    a twelve-actuator biped 89 / 194, the tail biped (15 actuators) 107 / 221."""
    import os
    from open_duck_playground_amd import constants, engine
    from open_duck_playground_amd.model import Model
    assets = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
    assert engine.model_obs_sizes(model_a, 0) == engine.obs_sizes(0) == (101, 212) and engine.model_obs_sizes(model_b, 1) == engine.obs_sizes(1) == (85, 153)
    b12 = Model.from_xml(os.path.join(assets, "biped12.xml"))
    tb = Model.from_xml(os.path.join(assets, "tail_biped.xml"))
    assert engine.model_obs_sizes(b12, 0) == (89, 194) and engine.model_obs_sizes(tb, 0) == (107, 221)
    assert constants.robot_of(model_a).is_open_duck and constants.robot_of(model_b).is_open_duck
    assert constants.robot_of(model_a).joints_order_no_head == constants.JOINTS_ORDER_NO_HEAD
    r = constants.robot_of(b12)
    assert not r.is_open_duck and r.joints_order_no_head == [str(n) for n in b12.a["names_actuator"]]      # every actuator of this robot is a leg joint
    r = constants.robot_of(tb)
    assert not r.is_open_duck and len(r.joints_order_no_head) == 10 and all("tail" not in j for j in r.joints_order_no_head)
    # the engine config of such a robot: no imitation reward, qpos noise by the robot's own joint names
    from open_duck_playground_amd import joystick
    cfg = joystick.to_engine_config(joystick.default_config(), use_imitation=False, joints_order_no_head=constants.robot_of(b12).joints_order_no_head)
    assert cfg.use_imitation == 0
    np.testing.assert_allclose(list(cfg.qpos_noise_scale)[:12], [0.03, 0.03, 0.03, 0.05, 0.08, 0.08] * 2, rtol=1e-6)


def test_new_shape_tool_names_the_lines_a_new_robot_needs(tmp_path, capsys):
    """tools/new_shape.py (reference README.md:74-85 "Adding a new robot"): a robot a compiled shape takes -> its env sizes and the runner line;
    a robot with another model shape (biped12 with an extra torso joint chain) -> the `using Shape<...>` line and the ODK_SHAPES entry."""
    import importlib.util, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("new_shape", os.path.join(root, "tools", "new_shape.py"))
    ns = importlib.util.module_from_spec(spec); spec.loader.exec_module(ns)
    import sys
    argv = sys.argv
    try:
        sys.argv = ["new_shape.py", os.path.join(root, "tests", "assets", "biped12.xml")]
        assert ns.main() == 0
        out = capsys.readouterr().out
        assert "a compiled kernel shape takes it" in out and "action 12, observation 89, privileged observation 194" in out and "--xml" in out
        # the same robot with a two-link neck on the trunk (two more hinges + actuators): 20 dofs, 14 actuators, but not the duck's tree
        src = open(os.path.join(root, "tests", "assets", "biped12.xml")).read()
        neck = ('<body name="neck" pos="0.02 0 0.12"><inertial pos="0 0 0.02" mass="0.05" fullinertia="2e-5 2e-5 1e-5 0 0 0"/><joint name="neck_a" axis="0 1 0" range="-0.5 0.5"/>'
                '<body name="head" pos="0 0 0.04"><inertial pos="0 0 0.02" mass="0.08" fullinertia="4e-5 4e-5 3e-5 0 0 0"/><joint name="neck_b" axis="0 0 1" range="-0.8 0.8"/></body></body>')
        src = src.replace('<body name="left_hip_yaw_link"', neck + '<body name="left_hip_yaw_link"', 1)
        src = src.replace('</actuator>', '<position name="neck_a" joint="neck_a" kp="5" forcerange="-1 1" inheritrange="1"/><position name="neck_b" joint="neck_b" kp="5" forcerange="-1 1" inheritrange="1"/></actuator>')
        src = src.replace('0 0 -0.4 0.8 -0.4 0  0 0 -0.4 0.8 -0.4 0" ctrl="0 0 -0.4 0.8 -0.4 0  0 0 -0.4 0.8 -0.4 0"', '0 0  0 0 -0.4 0.8 -0.4 0  0 0 -0.4 0.8 -0.4 0" ctrl="0 0 -0.4 0.8 -0.4 0  0 0 -0.4 0.8 -0.4 0  0 0"')
        xml = tmp_path / "biped12_neck.xml"
        xml.write_text(src)
        sys.argv = ["new_shape.py", str(xml)]
        assert ns.main() == 1
        out = capsys.readouterr().out
        m = re.search(r"using ShapeX = Shape<([^>]*)>;", out)
        assert m, out
        dims = [t.strip() for t in m.group(1).split(",")]
        assert dims[:5] == ["21", "20", "18", "14", "15"] and dims[-3:] == ["false", "6", "true"]      # nq, nv, nbody, nu, njnt ...; chains of six; optional constraint code
        assert "X(4, ShapeX)" in out and "ODK_SHAPES" in out
        # `--add` without touching the tree: the user header in a directory of its own, the kernels' static_asserts checked by a syntax-only compile
        hdr = tmp_path / "odk_shapes_user.h"
        alias = ns.add_user_shape(m.group(1), header=str(hdr))
        assert alias == "ShapeU0" and ns.add_user_shape(m.group(1), header=str(hdr)) == "ShapeU0"      # idempotent
        text = hdr.read_text()
        assert f"using ShapeU0 = Shape<{m.group(1)}>;" in text and "#define ODK_USER_SHAPES(X) X(4, ShapeU0)" in text
        ok, err = ns.syntax_check(include_dir=str(tmp_path))
        assert ok, err
        # a shape the kernels cannot take (60 constraint rows: the foot-foot routine's scratch does not fit) stops at a static_assert, by name
        bad = m.group(1).split(",")
        bad[7] = " 60"
        ns.add_user_shape(",".join(bad), header=str(hdr))
        ok, err = ns.syntax_check(include_dir=str(tmp_path))
        assert not ok and "static assertion" in err
    finally:
        sys.argv = argv


def test_compiler_refuses_colliding_primitives(tmp_path):
    from open_duck_playground_amd import mjcf
    xml = """<mujoco><compiler angle="radian"/><worldbody>
      <geom name="floor" type="plane" size="0 0 0.01"/>
      <body name="a"><freejoint/><inertial pos="0 0 0" mass="1" fullinertia="1 1 1 0 0 0"/><geom name="ball" type="ellipsoid" size="0.1 0.1 0.2"/></body>
      </worldbody></mujoco>"""
    path = tmp_path / "m.xml"
    path.write_text(xml)
    with pytest.raises(NotImplementedError, match="ball"):
        mjcf.compile_mjcf(str(path))
    path.write_text(xml.replace('type="ellipsoid" size="0.1 0.1 0.2"', 'type="cylinder" size="0.1 0.1"'))
    with pytest.raises(NotImplementedError, match="ball"):
        mjcf.compile_mjcf(str(path))
    # spheres and capsules collide (tests/test_mjcf_box.py), but not against a height field
    path.write_text(xml.replace('type="ellipsoid" size="0.1 0.1 0.2"', 'type="sphere" size="0.1"'))
    a = mjcf.compile_mjcf(str(path))
    assert list(a["cgeom_type"]) == [0, mjcf.GEOM_SPHERE] and a["cgeom_size"][1][0] == pytest.approx(0.1)


def _with_feet(model, verts, margin=None):
    """a copy of `model` whose two foot colliders are the convex hull of `verts` (geom frame)"""
    from open_duck_playground_amd import mjcf
    from open_duck_playground_amd.model import Model
    hv, hf = mjcf.convex_hull(np.asarray(verts, np.float64))
    a = dict(model.a)
    a["hull_vert"], a["hull_face"] = hv, hf.astype(np.int32)
    vn, fn = a["cgeom_vertnum"].copy(), a["cgeom_facenum"].copy()
    for g in a["k_foot_cgeom"] if "k_foot_cgeom" in a else (0, 1):
        vn[g], fn[g] = len(hv), len(hf)
    a["cgeom_vertadr"], a["cgeom_faceadr"] = np.zeros_like(a["cgeom_vertadr"]), np.zeros_like(a["cgeom_faceadr"])
    a["cgeom_vertnum"], a["cgeom_facenum"] = vn, fn
    if margin is not None:
        a["cgeom_margin"] = np.asarray(margin, np.float64)
    return Model(a)


def test_loader_refuses_hulls_beyond_the_kernels_regions(model_a):
    """The kernels' LDS regions hold 17 hull vertices / 30 merged faces / 48 edges per foot (odk_model.h HULL_MAX*): a larger hull is
    ODK_ERR_UNSUPPORTED at odk_model_load, never a silent overrun (include/odk.h).  The duck's own hull is exactly 17 / 30 / 45."""
    from open_duck_playground_amd import engine
    engine.model_reduction(model_a)                                   # the shipped model loads
    ang = np.linspace(0, 2 * np.pi, 9, endpoint=False)
    prism18 = np.array([[0.05 * np.cos(t), 0.03 * np.sin(t), z] for z in (-0.01, 0.01) for t in ang])     # 18 vertices, 11 faces
    with pytest.raises(engine.OdkError, match="18 vertices"):
        engine.model_reduction(_with_feet(model_a, prism18))
    ang = np.linspace(0, 2 * np.pi, 8, endpoint=False)
    prism16 = np.array([[0.05 * np.cos(t), 0.03 * np.sin(t), z] for z in (-0.01, 0.01) for t in ang])     # 16 vertices, octagon faces
    with pytest.raises(engine.OdkError, match="4-vertex faces"):
        engine.model_reduction(_with_feet(model_a, prism16))
    box = np.array([[sx * 0.05, sy * 0.03, sz * 0.01] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    engine.model_reduction(_with_feet(model_a, box))                  # 8 vertices / 6 faces / 12 edges: fine
    with pytest.raises(engine.OdkError, match="margin"):
        engine.model_reduction(_with_feet(model_a, box, margin=[0.0, 0.002, 0.0]))


def test_loader_refuses_feet_wider_than_the_prism_window():
    """hfield_contacts walks a window of <= 3 x 3 cells under the foot: a foot whose box spans two cells or more (a finer height
    field, a larger foot) is refused at load instead of losing cells silently (MJX sizes its sub-grid from the same ratio)."""
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.model import Model, load_task_model
    m = load_task_model("rough_terrain_backlash")
    engine.model_reduction(m)                                          # the duck's foot: 0.12 m across, cells 0.078 m
    big = np.array([[sx * 0.09, sy * 0.05, sz * 0.01] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    with pytest.raises(engine.OdkError, match="smaller than two cells"):
        engine.model_reduction(_with_feet(m, big))
    a = dict(m.a)
    a["hfield_size"] = np.array(a["hfield_size"], np.float64) * np.array([0.5, 0.5, 1, 1])     # same samples on half the area: cells 0.039 m
    with pytest.raises(engine.OdkError, match="smaller than two cells"):
        engine.model_reduction(Model(a))


def test_loader_takes_primitive_feet_on_a_height_field_but_not_beside_a_hull():
    """two sphere / capsule feet on the height-field floor load (their own kernel instantiation); a hull foot beside a
    primitive one does not (hfield_contacts and hfield_prim_floor each work both feet)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _prim_feet_variant
    from open_duck_playground_amd import engine
    from open_duck_playground_amd.mjcf import GEOM_MESH
    from open_duck_playground_amd.model import Model, load_task_model
    engine.model_reduction(_prim_feet_variant("rough_terrain_backlash", ("capsule", "sphere")))
    base = load_task_model("rough_terrain_backlash")
    prim = _prim_feet_variant("rough_terrain_backlash", ("sphere", "sphere"))
    a = {k: np.array(v) for k, v in prim.a.items()}
    for k in ("cgeom_type", "cgeom_pos", "cgeom_quat", "cgeom_vertnum", "cgeom_facenum", "cgeom_size"):   # foot 1 back to its hull
        if k in base.a:
            a[k][1] = np.asarray(base.a[k])[1]
        else:
            a[k][1] = 0
    assert a["cgeom_type"][1] == GEOM_MESH
    with pytest.raises(engine.OdkError, match="both feet are hulls"):
        engine.model_reduction(Model(a, prim.xml_path))
