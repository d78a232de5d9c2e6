"""CPU tests of the oracle's convex-convex narrow phase (oracle/odk_oracle_convex.inc): separating-axis test with Gauss-map edge
pruning, clipped face manifolds, edge-edge contacts, the polygon / edge tables of the foot hull, and the height-field prisms.
The oracle restates mujoco-mjx collision_convex.py from memory (parity unpinned); these tests pin it to independent numpy
evaluations and to closed-form cases."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull
from scipy.spatial.transform import Rotation


def _hull(pts):
    h = ConvexHull(pts)
    keep = np.sort(h.vertices)
    remap = -np.ones(len(pts), int); remap[keep] = np.arange(len(keep))
    f = remap[h.simplices]; v = pts[keep]; c = v.mean(0)
    for i, t in enumerate(f):
        a, b, cc = v[t]
        if np.dot(np.cross(b - a, cc - a), a - c) < 0:
            f[i] = t[[0, 2, 1]]
    return v, f.astype(np.int32)


def _normals(v, f):
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    return n / np.linalg.norm(n, axis=1)[:, None]


def _brute_force_sat(va, fa, vb, fb):
    """largest separation over ALL face normals and ALL edge-pair cross products, by projecting both vertex sets: the textbook
    test the Gauss-map version must agree with whenever the hulls overlap"""
    def edge_dirs(v, f):
        e = set()
        for t in f:
            for i in range(3):
                a, b = t[i], t[(i + 1) % 3]; e.add((min(a, b), max(a, b)))
        return np.array([v[b] - v[a] for a, b in e])
    cr = np.cross(edge_dirs(va, fa)[:, None, :], edge_dirs(vb, fb)[None, :, :]).reshape(-1, 3)
    ln = np.linalg.norm(cr, axis=1); cr = cr[ln > 1e-9] / ln[ln > 1e-9][:, None]
    ax = np.concatenate([_normals(va, fa), -_normals(vb, fb), cr, -cr])
    sep = (ax @ vb.T).min(1) - (ax @ va.T).max(1)
    return sep.max()


def _inside(v, f, p, tol):
    n = _normals(v, f)
    return bool((((p[None] - v[f[:, 0]]) * n).sum(1) <= tol).all())


CUBE = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)])


def test_sat_matches_brute_force_and_manifold_points_lie_on_the_hulls(oracle_mod):
    rng = np.random.default_rng(0)
    n_pen = 0
    kinds = [0, 0, 0]
    for it in range(400):
        va, fa = _hull(rng.normal(size=(12, 3)) * [0.5, 0.3, 0.2]); vb, fb = _hull(rng.normal(size=(10, 3)) * 0.3)
        Ra = Rotation.random(random_state=it).as_matrix(); Rb = Rotation.random(random_state=1000 + it).as_matrix()
        ta = rng.normal(size=3) * 0.1; tb = rng.normal(size=3) * 0.3
        r = oracle_mod.convex_pair(va, fa, ta, Ra, vb, fb, tb, Rb)
        wa = va @ Ra.T + ta; wb = vb @ Rb.T + tb
        ref = _brute_force_sat(wa, fa, wb, fb)
        got = max(r["sep_a"], r["sep_b"], r["sep_e"])
        if ref > 0:      # separated: every contact slot inactive (the best axis of a separated pair need not be a SAT axis)
            assert got > 0 and (r["dist"] > 0).all()
            continue
        n_pen += 1
        kinds[r["kind"]] += 1
        assert got == pytest.approx(ref, abs=1e-12), it
        assert abs(np.linalg.norm(r["normal"]) - 1) < 1e-12
        # the normal points from A to B: pushing B along it by the penetration depth separates the pair (to rounding)
        assert _brute_force_sat(wa, fa, wb + (abs(got) + 1e-9) * r["normal"], fb) > -1e-12, it
        live = r["dist"] < 0
        assert live.any(), it
        if r["kind"] == 2:   # one contact: midpoint of the closest points of the two edges, depth = the axis's penetration
            assert live.sum() == 1 and r["dist"][0] == pytest.approx(r["sep_e"], abs=1e-12)
            assert _inside(wa, fa, r["pos"][0], 0.5 * abs(got) + 1e-9) and _inside(wb, fb, r["pos"][0], 0.5 * abs(got) + 1e-9)
        else:
            n_ref = r["normal"] if r["kind"] == 0 else -r["normal"]
            Rv, Rf, Iv, If = (wa, fa, wb, fb) if r["kind"] == 0 else (wb, fb, wa, fa)
            assert r["dist"][live].min() >= got - 1e-9       # nothing deeper than the penetration along the reference normal
            for k in np.flatnonzero(live):
                ref_pt = r["pos"][k] - 0.5 * r["dist"][k] * n_ref; inc_pt = r["pos"][k] + 0.5 * r["dist"][k] * n_ref
                assert _inside(Rv, Rf, ref_pt, 1e-9), (it, k)      # on the reference face (inside its polygon: clipped)
                assert _inside(Iv, If, inc_pt, 1e-9), (it, k)      # on the incident face
    assert n_pen > 100 and min(kinds) > 10, (n_pen, kinds)


def test_box_on_box_gives_the_overlap_rectangle(oracle_mod):
    v, f = _hull(CUBE)
    r = oracle_mod.convex_pair(v, f, [0, 0, 0], np.eye(3), v, f, [0.5, 0.3, 0.9], np.eye(3))
    assert r["kind"] == 0 and np.allclose(r["normal"], [0, 0, 1])
    assert np.allclose(r["dist"], -0.1)
    corners = {(0.0, -0.2), (0.5, -0.2), (0.5, 0.5), (0.0, 0.5)}
    got = {(round(float(p[0]), 9), round(float(p[1]), 9)) for p in r["pos"]}
    assert got <= corners and len(got) >= 3           # the 4th pick of _manifold_points ties on a perfect rectangle (area measure)
    assert np.allclose(r["pos"][:, 2], 0.45)          # halfway between the two faces
    # rotated about the vertical by 30 degrees and pressed in by 2 cm: still a face contact with 4 distinct points
    Rz = Rotation.from_euler("z", 30, degrees=True).as_matrix()
    r = oracle_mod.convex_pair(v, f, [0, 0, 0], np.eye(3), v, f, [0.1, 0.05, 0.98], Rz)
    assert r["kind"] in (0, 1) and np.allclose(np.abs(r["normal"]), [0, 0, 1]) and np.allclose(r["dist"], -0.02)
    assert len({tuple(np.round(p, 9)) for p in r["pos"]}) == 4


def test_crossed_edges_give_one_edge_contact(oracle_mod):
    v, f = _hull(CUBE)
    # B balanced on an edge (rotated 45 degrees about y) and turned by 90 degrees about z against A's top edge direction:
    # its lowest edge runs along y, A is rotated 45 degrees about x so that its highest edge runs along x
    Ra = Rotation.from_euler("x", 45, degrees=True).as_matrix(); Rb = Rotation.from_euler("y", 45, degrees=True).as_matrix()
    h = np.sqrt(0.5)
    r = oracle_mod.convex_pair(v, f, [0, 0, 0], Ra, v, f, [0, 0, 2 * h - 0.03], Rb)
    assert r["kind"] == 2
    assert r["dist"][0] == pytest.approx(-0.03, abs=1e-12) and (r["dist"][1:] == 1.0).all()
    assert np.allclose(r["normal"], [0, 0, 1]) and np.allclose(r["pos"][0], [0, 0, h - 0.015])


def test_foot_hull_polygons_and_edges(oracle_mod, model_a):
    om = oracle_mod.OracleModel(model_a.blob())
    for g in (0, 1):
        nv, nf, ne = om.convex_counts(g)
        assert nv == 17 and nv - ne + nf == 2 and nf <= 30      # Euler's formula: a closed polytope, every edge has two faces


def _prism_candidates(oracle_mod, a, foot_v, foot_f, foot_pos, foot_mat, centre, rad):
    """independent numpy statement of `_hfield_collision`: the prisms of the cells under the bounding sphere as 6-point hulls
    (scipy), convex_pair per prism, all 4-slot results in row / column / triangle order"""
    H, size = np.asarray(a["hfield_data"]), np.asarray(a["hfield_size"])
    nr, nc = H.shape
    dx, dy = 2 * size[0] / (nc - 1), 2 * size[1] / (nr - 1)
    cmin, cmax = int(np.floor((centre[0] - rad + size[0]) / dx)), int(np.floor((centre[0] + rad + size[0]) / dx))
    rmin, rmax = int(np.floor((centre[1] - rad + size[1]) / dy)), int(np.floor((centre[1] + rad + size[1]) / dy))
    out = []
    for r in range(max(rmin, 0), min(rmax, nr - 2) + 1):
        for c in range(max(cmin, 0), min(cmax, nc - 2) + 1):
            for tri in ((c, r), (c + 1, r), (c, r + 1)), ((c + 1, r + 1), (c, r + 1), (c + 1, r)):
                top = np.array([[-size[0] + cc * dx, -size[1] + rr * dy, H[rr, cc] * size[2]] for cc, rr in tri])
                bot = top.copy(); bot[:, 2] = -size[3]
                # same vertex and triangle order as the oracle's prism (the manifold's first point is "the first masked candidate":
                # the result depends on where each polygon starts)
                pv = np.concatenate([top, bot])
                pf = np.array([[0, 1, 2], [3, 5, 4], [0, 3, 4], [0, 4, 1], [1, 4, 5], [1, 5, 2], [2, 5, 3], [2, 3, 0]], np.int32)
                res = oracle_mod.convex_pair(pv, pf, [0, 0, 0], np.eye(3), foot_v, foot_f, foot_pos, foot_mat)
                for k in range(4):
                    out.append((res["dist"][k], res["pos"][k], res["normal"]))
    return out


def test_height_field_prisms_against_numpy(oracle_mod):
    """rough_terrain_backlash (scene_rough_terrain_backlash.xml:22: 256 x 256 samples, size 10 10 .01 0.1): the oracle's
    hfield_convex == the four deepest contacts of an independent numpy enumeration of the prisms; normals are those of the prism
    test that produced the contact; the robot settles on the bumps."""
    from open_duck_playground_amd.model import load_task_model
    model = load_task_model("rough_terrain_backlash")
    a = model.a
    H, size = np.asarray(a["hfield_data"]), np.asarray(a["hfield_size"])
    assert H.shape == (256, 256) and tuple(size) == (10.0, 10.0, 0.01, 0.1)
    om = oracle_mod.OracleModel(model.blob())
    rng = np.random.default_rng(0)
    hv = np.asarray(a["hull_vert"]); hf = np.asarray(a["hull_face"])
    n_live = 0
    tilts = []
    for trial in range(16):
        q = np.array(a["key_qpos"], dtype=np.float64)
        q[0:2] = rng.uniform(-4, 4, 2); q[2] = rng.uniform(0.145, 0.165)
        ang = rng.uniform(-0.2, 0.2); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        yaw = rng.uniform(-np.pi, np.pi)
        qt = Rotation.from_rotvec(ang * ax) * Rotation.from_euler("z", yaw)
        q[3:7] = np.roll(qt.as_quat(), 1)      # scipy: x y z w -> w x y z
        d = oracle_mod.OracleData(om)
        d["qpos"][: om.nq] = q
        d.forward()
        frames = np.array(d["contact_frame"][: 8 * 9]).reshape(8, 9)
        dist = np.array(d["contact_dist"][:8]); pos = np.array(d["contact_pos"][: 8 * 3]).reshape(8, 3)
        gx = np.array(d["geom_xpos"][:9]).reshape(3, 3); gm = np.array(d["geom_xmat"][:27]).reshape(3, 3, 3)
        for f in range(2):   # collision geoms: left foot, right foot, floor (identity pose)
            v = hv[a["cgeom_vertadr"][f]: a["cgeom_vertadr"][f] + a["cgeom_vertnum"][f]]
            tris = hf[a["cgeom_faceadr"][f]: a["cgeom_faceadr"][f] + a["cgeom_facenum"][f]]
            wv = gx[f] + v @ gm[f].T
            centre = gx[f] + gm[f] @ (0.5 * (v.min(0) + v.max(0))); rad = np.linalg.norm(0.5 * (v.max(0) - v.min(0)))
            cand = _prism_candidates(oracle_mod, a, v, tris, gx[f], gm[f], centre, rad)
            order = sorted(range(len(cand)), key=lambda i: (cand[i][0], i))[:4]       # four deepest, ties to the lower index
            for k, i in enumerate(order):
                assert dist[4 * f + k] == pytest.approx(cand[i][0], abs=1e-12), (trial, f, k)
                if cand[i][0] < 0:
                    n_live += 1
                    np.testing.assert_allclose(pos[4 * f + k], cand[i][1], atol=1e-12)
                    np.testing.assert_allclose(frames[4 * f + k, :3], cand[i][2], atol=1e-12)
                    fr = frames[4 * f + k].reshape(3, 3)
                    np.testing.assert_allclose(fr @ fr.T, np.eye(3), atol=1e-12)      # an orthonormal contact frame
                    tilts.append(np.degrees(np.arccos(min(1.0, frames[4 * f + k, 2]))))
    assert n_live > 30
    # gentle bumps (<= 1 cm per 7.8 cm cell): normals within a few degrees of vertical -- except where the foot's rim overlaps a
    # neighbouring prism by less than it is pressed in: the least-penetration axis of THAT prism test is its vertical side face
    # (the per-prism algorithm's "internal edge" contacts; these poses are pressed in by up to 1.5 cm)
    tilts = np.array(tilts)
    assert 0.01 < np.median(tilts) < 8.0 and (tilts > 12.0).mean() < 0.15      # the rest: side faces and edge-edge axes
    # settles on the terrain
    d = oracle_mod.OracleData(om)
    d["qpos"][: om.nq] = np.array(a["key_qpos"], dtype=np.float64)
    ctrl = np.asarray(a["key_ctrl"], dtype=np.float64)
    for _ in range(40):
        d.env_physics_step(ctrl, 10)
    nr, nc = H.shape
    ground = H[nr // 2 - 2: nr // 2 + 2, nc // 2 - 2: nc // 2 + 2].mean() * size[2]
    assert 0.14 + ground - 0.01 < d["qpos"][2] < 0.18 + ground and d["sensordata"][11] > 0.99


def test_round2_one_triangle_mode_is_still_available(oracle_mod):
    """hfield_mode = 1 keeps round 2's approximation (plane of the triangle under the hull centre) so that the difference can
    be measured (tools/hfield_mode_deviation.py); on a patch that is flat under the whole foot both modes report the same depth."""
    from open_duck_playground_amd.model import load_task_model
    model = load_task_model("rough_terrain_backlash")
    a = model.a
    flat = dict(a); flat["hfield_data"] = np.full_like(np.asarray(a["hfield_data"]), 0.5)
    from open_duck_playground_amd.model import Model
    mflat = Model(flat)
    res = []
    for mode in (0, 1):
        om = oracle_mod.OracleModel(mflat.blob()); om.set_int("hfield_mode", mode)
        d = oracle_mod.OracleData(om)
        q = np.array(a["key_qpos"], dtype=np.float64); q[2] = 0.152
        d["qpos"][: om.nq] = q
        d.forward()
        res.append((np.array(d["contact_dist"][:8]), np.array(d["contact_frame"][: 8 * 9]).reshape(8, 9)))
    for f in range(2):
        assert res[0][0][4 * f: 4 * f + 4].min() == pytest.approx(res[1][0][4 * f: 4 * f + 4].min(), abs=1e-9)
        live = res[0][0][4 * f: 4 * f + 4] < 0
        nz = res[0][1][4 * f: 4 * f + 4][live][:, 2]
        # the deepest contact is a face contact with the flat top; a slot may hold an edge-edge contact against the cell's
        # diagonal (an internal edge of the per-prism algorithm), a few degrees off the vertical
        assert nz[0] == pytest.approx(1.0, abs=1e-12) and (nz > 0.99).all()


def test_sweep_state_generator_makes_contact_rich_states(oracle_mod):
    """tools/gpu_fuzz_parity.make_states (CPU part of the differential sweep, tests/test_gpu_parity.py::test_differential_sweep):
    most states touch the floor, some with both feet, some with the feet against each other, some airborne -- on every model."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gpu_fuzz_parity", os.path.join(ROOT, "tools", "gpu_fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    for task in ("flat_terrain", "rough_terrain_backlash"):
        model, om, om32, qpos, qvel, warm, ctrl = fz.make_states(task, 96, seed=5)
        floor = both = ff = air = 0
        for e in range(96):
            d = oracle_mod.OracleData(om)
            d["qpos"][: om.nq] = qpos[e]
            d.forward()
            cd = np.array(d["contact_dist"][:12])
            f0, f1 = (cd[:4] < 0).any(), (cd[4:8] < 0).any()
            floor += int(f0 or f1); both += int(f0 and f1); ff += int((cd[8:] < 0).any()); air += int(not (cd < 0).any())
            if f0 or f1:
                assert cd[:8].min() > -8e-3, (task, e, cd[:8].min())          # pressed in by at most the target depth
        assert floor >= 60 and both >= 5 and ff >= 6 and air >= 5, (task, floor, both, ff, air)


def _hfield_prim_model(O, kinds, heights):
    """rough-terrain model with primitive feet (the variant the GPU test uses) and the given height samples"""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _prim_feet_variant
    from open_duck_playground_amd.model import Model
    m = _prim_feet_variant("rough_terrain_backlash", kinds)
    a = {k: np.array(v) for k, v in m.a.items()}
    a["hfield_data"] = np.asarray(heights, np.float64)
    return Model(a, m.xml_path)


@pytest.mark.parametrize("kind", ["sphere", "capsule"])
def test_primitive_on_a_planar_height_field_is_the_plane_collider(oracle_mod, kind):
    """hfield_sphere / hfield_capsule (oracle/odk_oracle.c: hfield_prim) on a height field whose samples lie on ONE plane: away from
    prism edges the answer is plane_sphere / plane_capsule against that plane -- depth = distance of the sphere centre / capsule end to
    the plane minus the radius, normal = the plane's, position half a depth under the surface point.  (What the routine does near
    a prism's edges, and which contacts it keeps among the prisms, is the recollection of MJX this build cannot pin.)"""
    O = oracle_mod
    nr = nc = 256
    sx = 10.0; dxy = 2 * sx / (nc - 1)
    sz = 0.01
    gx, gy = 0.02, -0.035                                   # slopes (dimensionless) of the plane z = gx x + gy y + z0
    xs = -sx + dxy * np.arange(nc); ys = -sx + dxy * np.arange(nr)
    z = gx * xs[None, :] + gy * ys[:, None]
    z0 = -z.min() + 0.001
    model = _hfield_prim_model(O, (kind, kind), (z + z0) / sz)
    om = O.OracleModel(model.blob())
    nrm = np.array([-gx, -gy, 1.0]); nrm /= np.linalg.norm(nrm)
    rng = np.random.default_rng(3)
    checked = 0
    for trial in range(60):
        d = O.OracleData(om)
        q = np.asarray(model.a["key_qpos"], np.float64).copy()
        q[:2] = rng.uniform(-5, 5, 2); q[2] = gx * q[0] + gy * q[1] + z0 + 0.3
        ang = rng.uniform(-0.3, 0.3); ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        q[3:7] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        for _ in range(4):                                  # lower the robot until a foot is ~1 mm in the plane
            d["qpos"][: om.nq] = q; d.forward()
            q[2] -= min(np.array(d["contact_dist"][:8]).min(), 0.05) + 1e-3
        d["qpos"][: om.nq] = q; d.forward()
        cd = np.array(d["contact_dist"][:8]); cp = np.array(d["contact_pos"][:24]).reshape(8, 3); fr = np.array(d["contact_frame"][:72]).reshape(8, 9)
        gpos = np.array(d["geom_xpos"]).reshape(-1, 3); gmat = np.array(d["geom_xmat"]).reshape(-1, 9)
        for f in range(2):
            g = f                                            # collision geoms: left foot, right foot, floor
            r, hl = float(model.a["cgeom_size"][f][0]), float(model.a["cgeom_size"][f][1])
            ends = [gpos[g]] if kind == "sphere" else [gpos[g] + s * hl * gmat[g].reshape(3, 3)[:, 2] for s in (-1, 1)]
            plane = lambda x: float(nrm @ x - z0 * nrm[2])          # signed distance to the plane through (0, 0, z0)
            # only configurations where every kept contact is a face contact of a prism's TOP: normals equal the plane's
            tops = [k for k in range(4 * f, 4 * f + len(ends)) if cd[k] < 0.5 and np.abs(fr[k][:3] - nrm).max() < 1e-9]
            if len(tops) != len(ends):
                continue
            # the deepest contact is the deeper end's (the sphere's centre's): plane_sphere against the plane
            np.testing.assert_allclose(min(cd[k] for k in tops), min(plane(e) - r for e in ends), atol=2e-9)
            for k in tops:
                # every kept contact sits under a point of the axis (a capsule's second contact is where a prism's side planes cut
                # the axis, not the far end): that point is on the segment, its depth is the plane's, the position half a depth
                # below the surface point
                ax_pt = cp[k] + nrm * (r + 0.5 * cd[k])
                a0, a1 = ends[0], ends[-1]
                t = 0.0 if len(ends) == 1 else float(np.clip((ax_pt - a0) @ (a1 - a0) / ((a1 - a0) @ (a1 - a0)), 0, 1))
                np.testing.assert_allclose(ax_pt, a0 + t * (a1 - a0), atol=2e-9)
                np.testing.assert_allclose(cd[k], plane(ax_pt) - r, atol=2e-9)
            assert (cd[4 * f + len(ends): 4 * f + 4] == 1.0).all()
            checked += 1
    assert checked >= 40, checked
