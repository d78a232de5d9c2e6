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


def test_spheres_and_capsules_are_still_refused(tmp_path):
    from open_duck_playground_amd import mjcf
    xml = open(os.path.join(ROOT, "tests", "assets", "toy_box_hopper.xml")).read().replace('type="box" pos="0 0 -0.1" size="0.04 0.02 0.01"', 'type="sphere" pos="0 0 -0.1" size="0.02"')
    p = tmp_path / "toy_sphere.xml"
    p.write_text(xml)
    with pytest.raises(NotImplementedError, match="spheres, capsules"):
        mjcf.compile_mjcf(str(p))


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
