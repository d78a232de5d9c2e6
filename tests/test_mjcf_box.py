"""SURVEY 8(f).3, first step: colliding BOX geoms through the MJCF compiler and the oracle (a box collides as the convex hull of its
eight corners, which is how MJX collides boxes with planes, height fields and meshes), on a toy robot that is not the duck
(tests/assets/toy_box_hopper.xml: written for this test).  The HIP kernels stay instances over the duck's shapes; a duck whose feet
are boxes runs through them in tests/test_gpu_parity.py::test_box_feet_variant."""
import os

import numpy as np
import pytest

from conftest import ROOT


def _toy():
    from open_duck_playground_amd import mjcf
    return mjcf.compile_mjcf(os.path.join(ROOT, "tests", "assets", "toy_box_hopper.xml"))


def test_box_geoms_compile_to_hulls():
    a = _toy()
    assert int(a["nq"][0]) == 8 and int(a["nv"][0]) == 7 and int(a["nu"][0]) == 1
    assert list(a["cgeom_type"]) == [7, 7, 0]                       # two boxes (as convex meshes) and the plane
    assert list(a["cgeom_vertnum"][:2]) == [8, 8] and list(a["cgeom_facenum"][:2]) == [12, 12]
    v = np.asarray(a["hull_vert"])
    assert np.allclose(np.abs(v[:8]), [0.05, 0.04, 0.03]) and np.allclose(np.abs(v[8:16]), [0.04, 0.02, 0.01])
    # outward triangles: every face normal points away from the centre
    f = np.asarray(a["hull_face"])[:12]
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert ((n * v[f[:, 0]]).sum(1) > 0).all()


FOOT_BOX = 'type="box" pos="0 0 -0.1" size="0.04 0.02 0.01"'
TRUNK_BOX = 'type="box" size="0.05 0.04 0.03"'


def _toy_variant(tmp_path, foot, trunk=None, name="v"):
    from open_duck_playground_amd import mjcf
    xml = open(os.path.join(ROOT, "tests", "assets", "toy_box_hopper.xml")).read().replace(FOOT_BOX, foot)
    if trunk is not None:
        xml = xml.replace(TRUNK_BOX, trunk)
    p = tmp_path / f"toy_{name}.xml"
    p.write_text(xml)
    return mjcf.compile_mjcf(str(p))


def test_spheres_and_capsules_compile(tmp_path):
    from open_duck_playground_amd import mjcf
    a = _toy_variant(tmp_path, 'type="sphere" pos="0 0 -0.1" size="0.02"')
    assert list(a["cgeom_type"]) == [7, mjcf.GEOM_SPHERE, 0] and np.allclose(a["cgeom_size"][1], [0.02, 0, 0]) and int(a["cgeom_vertnum"][1]) == 0
    a = _toy_variant(tmp_path, 'type="capsule" pos="0 0 -0.1" quat="0.7071067811865476 0 0.7071067811865476 0" size="0.015 0.03"')
    assert list(a["cgeom_type"]) == [7, mjcf.GEOM_CAPSULE, 0] and np.allclose(a["cgeom_size"][1], [0.015, 0.03, 0])
    # fromto: position = the midpoint, z axis of the geom frame from -> to, half length derived
    a = _toy_variant(tmp_path, 'type="capsule" fromto="-0.03 0 -0.1 0.03 0 -0.1" size="0.015"')
    assert np.allclose(a["cgeom_size"][1], [0.015, 0.03, 0]) and np.allclose(a["cgeom_pos"][1], [0, 0, -0.1])
    q = a["cgeom_quat"][1]
    zax = np.array([2 * (q[1] * q[3] + q[0] * q[2]), 2 * (q[2] * q[3] - q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)])
    assert np.allclose(zax, [1, 0, 0], atol=1e-12)
    for bad, what in (('type="ellipsoid" pos="0 0 -0.1" size="0.02 0.02 0.03"', "ellipsoid"), ('type="cylinder" pos="0 0 -0.1" size="0.02 0.03"', "cylinder")):
        with pytest.raises(NotImplementedError, match=what):
            _toy_variant(tmp_path, bad)


def _settle(oracle_mod, a, steps=1500):
    from open_duck_playground_amd.model import pack_blob
    om = oracle_mod.OracleModel(pack_blob(a))
    d = oracle_mod.OracleData(om)
    d["qpos"][: om.nq] = a["key_qpos"]
    for _ in range(steps):
        d.env_physics_step(np.zeros(1), 1)
    return om, d


def test_toy_robot_settles_on_a_sphere_and_on_a_capsule_foot(oracle_mod, tmp_path):
    """the parent-child pair (trunk, foot) is filtered as MuJoCo does: two pairs, floor-trunk and floor-foot"""
    weight = 1.2 * 9.81
    for foot, rest, nact in (('type="sphere" pos="0 0 -0.1" size="0.02"', 0.17, 1),
                             ('type="capsule" pos="0 0 -0.1" quat="0.7071067811865476 0 0.7071067811865476 0" size="0.015 0.03"', 0.165, 2)):
        om, d = _settle(oracle_mod, _toy_variant(tmp_path, foot))
        assert om.npair == 2
        assert rest - 1e-3 < d["qpos"][2] < rest and np.abs(d["qvel"][: om.nv]).max() < 0.05
        cd = np.array(d["contact_dist"][:8])
        assert (cd[:4] > 0).all() and (cd[4: 4 + nact] < 0).all() and (cd[4 + nact: 8] == 1.0).all()
        nefc = d.i("nefc")
        assert (d.J().T @ np.array(d["efc_force"][:nefc]))[2] == pytest.approx(weight, rel=5e-2)
        if nact == 2:   # the capsule lies along world x: contact frame (n, axis direction, n x axis), contacts under its two ends
            fr = np.array(d["contact_frame"][4 * 9: 5 * 9]).reshape(3, 3)
            assert np.allclose(fr[0], [0, 0, 1], atol=1e-6) and abs(abs(fr[1][0]) - 1) < 1e-3 and np.allclose(fr[2], np.cross(fr[0], fr[1]), atol=1e-12)
            pos = np.array(d["contact_pos"][4 * 3: 6 * 3]).reshape(2, 3)
            assert abs(pos[0][0] - pos[1][0]) == pytest.approx(0.06, abs=1e-3) and np.abs(pos[:, 2]).max() < 1e-3


def _two_feet(tmp_path, oracle_mod, g1, g2, q1, q2):
    """two free bodies, one primitive each, no floor contact: the oracle's body-body contact for poses q1 / q2 (pos + quat)"""
    from open_duck_playground_amd import mjcf
    from open_duck_playground_amd.model import pack_blob
    xml = f"""<mujoco><compiler angle="radian"/><worldbody>
      <geom name="floor" type="plane" size="0 0 0.01" pos="0 0 -10"/>
      <body name="a"><freejoint/><inertial pos="0 0 0" mass="1" fullinertia="1 1 1 0 0 0"/><geom name="ga" {g1}/></body>
      <body name="b"><freejoint/><inertial pos="0 0 0" mass="1" fullinertia="1 1 1 0 0 0"/><geom name="gb" {g2}/></body>
      </worldbody></mujoco>"""
    p = tmp_path / "two.xml"
    p.write_text(xml)
    a = mjcf.compile_mjcf(str(p))
    om = oracle_mod.OracleModel(pack_blob(a))
    d = oracle_mod.OracleData(om)
    d["qpos"][:14] = np.concatenate([q1, q2])
    d.forward()
    assert om.npair == 3
    c = 8
    return d["contact_dist"][c], np.array(d["contact_pos"][3 * c: 3 * c + 3]), np.array(d["contact_frame"][9 * c: 9 * c + 9]).reshape(3, 3), np.array(d["contact_dist"][c + 1: c + 4])


def test_primitive_pairs_closed_forms(oracle_mod, tmp_path):
    I = [1.0, 0, 0, 0]
    rng = np.random.default_rng(0)
    # sphere - sphere
    for _ in range(20):
        p1, p2 = rng.uniform(-0.1, 0.1, 3), rng.uniform(-0.1, 0.1, 3)
        dist, pos, fr, rest = _two_feet(tmp_path, oracle_mod, 'type="sphere" size="0.05"', 'type="sphere" size="0.03"', [*p1, *I], [*p2, *I])
        L = np.linalg.norm(p2 - p1); n = (p2 - p1) / L
        assert dist == pytest.approx(L - 0.08, abs=1e-12) and np.allclose(fr[0], n, atol=1e-12) and (rest == 1.0).all()
        assert np.allclose(pos, p1 + n * (0.05 + 0.5 * dist), atol=1e-12)
        assert np.allclose(fr @ fr.T, np.eye(3), atol=1e-12) and np.linalg.det(fr) == pytest.approx(1.0)
    # sphere - capsule (capsule along its local z, rotated about y by 90 deg: along world x), both orders
    qy = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]
    for sx, exp_pt in ((0.0, 0.0), (0.02, 0.02), (0.2, 0.05), (-0.3, -0.05)):
        sp = np.array([sx, 0.0, 0.06])
        dist, pos, fr, _ = _two_feet(tmp_path, oracle_mod, 'type="sphere" size="0.02"', 'type="capsule" size="0.01 0.05"', [*sp, *I], [0, 0, 0, *qy])
        pt = np.array([exp_pt, 0, 0]); L = np.linalg.norm(pt - sp)
        assert dist == pytest.approx(L - 0.03, abs=2e-6) and np.allclose(fr[0], (pt - sp) / L, atol=2e-4)      # (the 1e-6 in closest_segment_point's denominator: a 5 um shift along the axis)
        dist2, pos2, fr2, _ = _two_feet(tmp_path, oracle_mod, 'type="capsule" size="0.01 0.05"', 'type="sphere" size="0.02"', [0, 0, 0, *qy], [*sp, *I])
        assert dist2 == pytest.approx(dist, abs=1e-12) and np.allclose(fr2, fr, atol=1e-12) and np.allclose(pos2, pos, atol=1e-12)     # geoms ordered by type: the sphere is geom 1 either way
    # capsule - capsule: crossed (closest points inside both), parallel offset, end to end
    dist, pos, fr, _ = _two_feet(tmp_path, oracle_mod, 'type="capsule" size="0.01 0.05"', 'type="capsule" size="0.02 0.05"', [0, 0, 0, *qy], [0.01, 0, 0.025, np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0])
    assert dist == pytest.approx(0.025 - 0.03, abs=1e-7) and np.allclose(fr[0], [0, 0, 1], atol=5e-4) and np.allclose(pos, [0.01, 0, 0.01 + 0.5 * dist], atol=2e-5)      # (the same 1e-6: points 5 um along the axes)
    dist, pos, fr, _ = _two_feet(tmp_path, oracle_mod, 'type="capsule" size="0.01 0.05"', 'type="capsule" size="0.02 0.05"', [0, 0, 0, *I], [0, 0, 0.2, *I])
    assert dist == pytest.approx(0.1 - 0.03, abs=2e-6) and np.allclose(fr[0], [0, 0, 1], atol=5e-4)                                  # end to end along z
    dist, pos, fr, _ = _two_feet(tmp_path, oracle_mod, 'type="capsule" size="0.01 0.05"', 'type="capsule" size="0.02 0.05"', [0, 0, 0, *I], [0.04, 0, 0.01, *I])
    assert dist == pytest.approx(0.04 - 0.03, abs=2e-6) and abs(fr[0][0]) == pytest.approx(1.0, abs=1e-4)                           # parallel: some pair of closest points, 4 cm apart


def test_toy_robot_settles_on_its_box_foot(oracle_mod):
    from open_duck_playground_amd.model import pack_blob
    a = _toy()
    om = oracle_mod.OracleModel(pack_blob(a))
    assert om.nv == 7 and om.ncgeom == 3
    assert om.convex_counts(0) == (8, 6, 12) and om.convex_counts(1) == (8, 6, 12)      # coplanar triangles merged: six quads, twelve edges
    d = oracle_mod.OracleData(om)
    d["qpos"][: om.nq] = a["key_qpos"]
    for _ in range(1500):
        d.env_physics_step(np.zeros(1), 1)
    # base 0.05 above the knee, the foot box's centre 0.1 below it, half height 0.01: the base rests at 0.16 minus the static penetration
    assert 0.1590 < d["qpos"][2] < 0.1600 and abs(d["qpos"][3] - 1.0) < 1e-3 and np.abs(d["qvel"][: om.nv]).max() < 0.05      # (one Newton iteration per step: a slow residual rocking on the small foot)
    cd = np.array(d["contact_dist"][:12])
    foot = np.sort(cd[4:8])
    assert (foot[:2] < 0).all() and foot[0] > -5e-4                                       # the foot's bottom face carries the robot
    # static equilibrium: the contact forces hold the weight
    weight = 1.2 * 9.81
    nefc = d.i("nefc")
    J = d.J(); f = np.array(d["efc_force"][:nefc])
    assert (J.T @ f)[2] == pytest.approx(weight, rel=5e-2)
