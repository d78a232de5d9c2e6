"""MJCF subset compiler: XML -> dict of numpy arrays (the "ModelBlob" source).

The reference obtains its model by calling the MuJoCo C compiler
(`mujoco.MjModel.from_xml_string`, reference playground/open_duck_mini_v2/base.py:53-55)
and `mjx.put_model` (base.py:61).  Neither library exists in this build, so this module
re-states the part of the MJCF compiler the Open Duck scenes actually use
(SURVEY.md section 7 step 1):

  <include>, nested <default class>, childclass, <option>/<flag>, <compiler angle meshdir>,
  bodies with <inertial fullinertia>, <freejoint>, hinge joints, mesh / box / plane / hfield geoms
  with contype/conaffinity/priority/friction/condim, sites, the 9 sensor types of
  xmls/open_duck_mini_v2.xml:26-42, <position> actuators with kp/kv/forcerange/inheritrange,
  <keyframe>.

Derived constants MuJoCo computes in mj_setConst (dof_invweight0, body_invweight0,
stat.meaninertia, body_subtreemass) are recomputed here with a small float64 numpy
rigid-body routine (`mass_matrix_qpos0`).

Everything is float64 / int32 numpy; `model.pack_blob` turns the dict into the wire format the
C-ABI (`odk_model_load`, include/odk.h) and the oracle consume.
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

# MuJoCo enum values kept so index tables read like the reference's (base.py:88-100).
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_MESH = 0, 1, 2, 3, 7   # mjtGeom

# sensor type codes (own numbering; order = first appearance in open_duck_mini_v2.xml:26-42)
SENS_GYRO, SENS_VELOCIMETER, SENS_ACCELEROMETER = 0, 1, 2
SENS_FRAMEZAXIS, SENS_FRAMEXAXIS, SENS_FRAMELINVEL, SENS_FRAMEANGVEL = 3, 4, 5, 6
SENS_FRAMEPOS, SENS_FRAMEQUAT = 7, 8
_SENSOR_TAGS = {
    "gyro": (SENS_GYRO, 3), "velocimeter": (SENS_VELOCIMETER, 3),
    "accelerometer": (SENS_ACCELEROMETER, 3), "framezaxis": (SENS_FRAMEZAXIS, 3),
    "framexaxis": (SENS_FRAMEXAXIS, 3), "framelinvel": (SENS_FRAMELINVEL, 3),
    "frameangvel": (SENS_FRAMEANGVEL, 3), "framepos": (SENS_FRAMEPOS, 3),
    "framequat": (SENS_FRAMEQUAT, 4),
}

# MuJoCo defaults (mjmodel.h / XML reference)
DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
DEFAULT_GEOM_FRICTION = (1.0, 0.005, 0.0001)


# ----------------------------------------------------------------------------- small math
def quat_mul(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, float)
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def _normalize(v):
    v = np.asarray(v, float)
    n = np.linalg.norm(v)
    return v / n if n > 0 else v


def _floats(s: Optional[str], n: Optional[int] = None, default=None):
    if s is None:
        return None if default is None else np.array(default, float)
    v = np.array([float(t) for t in s.split()], float)
    if n is not None and len(v) < n:  # MuJoCo pads partially specified vectors with defaults
        base = np.array(default if default is not None else [0.0] * n, float)
        base[: len(v)] = v
        v = base
    return v


# ----------------------------------------------------------------------------- XML loading
def _load_tree(path: str) -> ET.Element:
    """Parses `path` and splices every <include file=...> in place (MuJoCo semantics:
    the included file's root children replace the include element)."""
    root = ET.parse(path).getroot()
    base = os.path.dirname(path)

    def expand(elem: ET.Element):
        new_children = []
        for ch in list(elem):
            if ch.tag == "include":
                inc = _load_tree(os.path.join(base, ch.attrib["file"]))
                new_children.extend(list(inc))
            else:
                expand(ch)
                new_children.append(ch)
        for ch in list(elem):
            elem.remove(ch)
        for ch in new_children:
            elem.append(ch)

    expand(root)
    return root


class _Defaults:
    """Default-class tree. classes[name][tag] -> attribute dict (already merged with parents)."""

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def ingest(self, elem: ET.Element, parent: str = "main", top: bool = True):
        name = elem.attrib.get("class", "main" if top else None)
        if name is None:
            raise ValueError("nested <default> needs a class name")
        if name not in self.classes:
            self.classes[name] = {k: dict(v) for k, v in self.classes[parent].items()}
        cur = self.classes[name]
        for ch in elem:
            if ch.tag == "default":
                continue
            cur.setdefault(ch.tag, {}).update(ch.attrib)
        for ch in elem:
            if ch.tag == "default":
                self.ingest(ch, parent=name, top=False)

    def get(self, cls: Optional[str], tag: str) -> Dict[str, str]:
        return dict(self.classes.get(cls or "main", self.classes["main"]).get(tag, {}))


# ----------------------------------------------------------------------------- STL / hull
def load_stl_vertices(path: str) -> np.ndarray:
    """Binary STL -> unique vertices in order of first appearance (float32 values as float64)."""
    data = open(path, "rb").read()
    ntri = struct.unpack("<I", data[80:84])[0]
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    tri = np.frombuffer(data[84:], dtype=rec, count=ntri)
    seen, out = {}, []
    for p in tri["v"].reshape(-1, 3):
        key = tuple(p.tolist())
        if key not in seen:
            seen[key] = len(out)
            out.append(key)
    return np.array(out, dtype=np.float64)


def convex_hull(verts: np.ndarray):
    """Hull vertices (kept in the input order), outward triangle faces re-indexed to the hull
    vertex list.  scipy/qhull here; MuJoCo also uses qhull (mesh_graph)."""
    from scipy.spatial import ConvexHull

    h = ConvexHull(verts)
    keep = np.sort(h.vertices)
    remap = -np.ones(len(verts), dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    faces = remap[h.simplices]
    hv = verts[keep]
    centre = hv.mean(axis=0)
    for i, f in enumerate(faces):  # orient outward
        a, b, c = hv[f]
        n = np.cross(b - a, c - a)
        if np.dot(n, a - centre) < 0:
            faces[i] = f[[0, 2, 1]]
    return hv, faces.astype(np.int32)


# ----------------------------------------------------------------------------- compiler
def compile_mjcf(xml_path: str, sim_dt: Optional[float] = None) -> Dict[str, np.ndarray]:
    """Compiles the scene at `xml_path` into flat arrays.  `sim_dt` overrides opt.timestep the
    way reference base.py:56 does (`self._mj_model.opt.timestep = self.sim_dt`)."""
    root = _load_tree(os.path.abspath(xml_path))
    xml_dir = os.path.dirname(os.path.abspath(xml_path))

    # ---- compiler / option
    meshdir, angle = "", "degree"
    for c in root.findall("compiler"):
        meshdir = c.attrib.get("meshdir", meshdir)
        angle = c.attrib.get("angle", angle)
    if angle != "radian":
        raise NotImplementedError("only <compiler angle='radian'> is supported")
    opt = dict(timestep=0.002, iterations=100, ls_iterations=50, tolerance=1e-8, ls_tolerance=0.01,
               impratio=1.0, gravity=np.array([0, 0, -9.81]), eulerdamp=1, solver=2, cone=0,
               integrator=0)
    for o in root.findall("option"):
        for k in ("timestep", "tolerance", "ls_tolerance", "impratio"):
            if k in o.attrib:
                opt[k] = float(o.attrib[k])
        for k in ("iterations", "ls_iterations"):
            if k in o.attrib:
                opt[k] = int(o.attrib[k])
        if "gravity" in o.attrib:
            opt["gravity"] = _floats(o.attrib["gravity"])
        for k in ("solver", "integrator"):
            if k in o.attrib:
                raise NotImplementedError(f"<option {k}=...> not supported (defaults: Newton / Euler)")
        if "cone" in o.attrib:      # mjtCone: pyramidal 0, elliptic 1 (elliptic: the float64 oracle and the kernels have it; `odk_model_load` refuses it with sphere / capsule feet)
            if o.attrib["cone"] not in ("pyramidal", "elliptic"):
                raise ValueError(f"<option cone='{o.attrib['cone']}'>")
            opt["cone"] = 1 if o.attrib["cone"] == "elliptic" else 0
        for f in o.findall("flag"):
            if f.attrib.get("eulerdamp") == "disable":
                opt["eulerdamp"] = 0
    if sim_dt is not None:
        opt["timestep"] = float(sim_dt)

    # ---- defaults
    dfl = _Defaults()
    for d in root.findall("default"):
        dfl.ingest(d)

    # ---- assets
    meshes: Dict[str, str] = {}
    hfields: Dict[str, dict] = {}
    for a in root.findall("asset"):
        for m in a.findall("mesh"):
            f = m.attrib["file"]
            name = m.attrib.get("name", os.path.splitext(os.path.basename(f))[0])
            meshes[name] = os.path.join(xml_dir, meshdir, f)
        for h in a.findall("hfield"):
            hfields[h.attrib["name"]] = dict(file=os.path.join(xml_dir, h.attrib["file"]),
                                             size=_floats(h.attrib["size"]))

    # ---- kinematic tree (depth first == MuJoCo id order)
    bodies: List[dict] = [dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                               mass=0.0, ipos=np.zeros(3), inertia=np.zeros((3, 3)), jnts=[])]
    joints: List[dict] = []
    geoms: List[dict] = []
    sites: List[dict] = []

    def attrs(elem, tag, childclass):
        cls = elem.attrib.get("class", childclass)
        a = dfl.get(cls, tag)
        a.update(elem.attrib)
        return a

    def add_geom(g, bid, childclass):
        a = attrs(g, "geom", childclass)
        gtype = a.get("type", "sphere")
        geoms.append(dict(
            name=a.get("name", ""), body=bid, type=gtype,
            contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
            priority=int(a.get("priority", 0)), condim=int(a.get("condim", 3)),
            friction=_floats(a.get("friction"), 3, DEFAULT_GEOM_FRICTION),
            pos=_floats(a.get("pos"), 3, [0, 0, 0]),
            quat=_normalize(_floats(a.get("quat"), 4, [1, 0, 0, 0])),
            mesh=a.get("mesh"), hfield=a.get("hfield"), size=_floats(a.get("size"), None, []) if a.get("size") else np.zeros(0),
            fromto=_floats(a.get("fromto"), 6) if a.get("fromto") else None,
            solref=_floats(a.get("solref"), 2, DEFAULT_SOLREF),
            solimp=_floats(a.get("solimp"), 5, DEFAULT_SOLIMP),
            solmix=float(a.get("solmix", 1.0)), margin=float(a.get("margin", 0.0)),
            gap=float(a.get("gap", 0.0)),
        ))

    def walk(elem, parent_id, childclass):
        for ch in elem:
            if ch.tag == "geom":
                add_geom(ch, parent_id, childclass)
            elif ch.tag == "site":
                a = attrs(ch, "site", childclass)
                sites.append(dict(name=a.get("name", ""), body=parent_id,
                                  pos=_floats(a.get("pos"), 3, [0, 0, 0]),
                                  quat=_normalize(_floats(a.get("quat"), 4, [1, 0, 0, 0]))))
            elif ch.tag == "body":
                bid = len(bodies)
                cc = ch.attrib.get("childclass", childclass)
                b = dict(name=ch.attrib.get("name", ""), parent=parent_id,
                         pos=_floats(ch.attrib.get("pos"), 3, [0, 0, 0]),
                         quat=_normalize(_floats(ch.attrib.get("quat"), 4, [1, 0, 0, 0])),
                         mass=0.0, ipos=np.zeros(3), inertia=np.zeros((3, 3)), jnts=[])
                bodies.append(b)
                for sub in ch:
                    if sub.tag == "inertial":
                        b["mass"] = float(sub.attrib["mass"])
                        b["ipos"] = _floats(sub.attrib.get("pos"), 3, [0, 0, 0])
                        if "fullinertia" in sub.attrib:
                            xx, yy, zz, xy, xz, yz = _floats(sub.attrib["fullinertia"])
                            I = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
                        else:
                            I = np.diag(_floats(sub.attrib["diaginertia"]))
                        if "quat" in sub.attrib:  # inertial frame orientation
                            R = quat_to_mat(_normalize(_floats(sub.attrib["quat"])))
                            I = R @ I @ R.T
                        b["inertia"] = I
                    elif sub.tag in ("joint", "freejoint"):
                        jid = len(joints)
                        if sub.tag == "freejoint":
                            j = dict(name=sub.attrib.get("name", ""), type=JNT_FREE, body=bid,
                                     pos=np.zeros(3), axis=np.array([0.0, 0, 1]), range=np.zeros(2),
                                     limited=0, damping=0.0, armature=0.0, frictionloss=0.0,
                                     solref_limit=np.array(DEFAULT_SOLREF), solimp_limit=np.array(DEFAULT_SOLIMP),
                                     solref_fric=np.array(DEFAULT_SOLREF), solimp_fric=np.array(DEFAULT_SOLIMP),
                                     margin=0.0)
                        else:
                            a = attrs(sub, "joint", cc)
                            jt = a.get("type", "hinge")
                            if jt == "free":
                                jtype = JNT_FREE
                            elif jt == "hinge":
                                jtype = JNT_HINGE
                            else:
                                raise NotImplementedError(f"joint type {jt}")
                            rng = _floats(a.get("range"), 2, [0, 0])
                            lim = a.get("limited", "auto")
                            limited = int((lim == "true") or (lim == "auto" and "range" in a))
                            j = dict(name=a.get("name", ""), type=jtype, body=bid,
                                     pos=_floats(a.get("pos"), 3, [0, 0, 0]),
                                     axis=_normalize(_floats(a.get("axis"), 3, [0, 0, 1])),
                                     range=rng, limited=limited,
                                     damping=float(a.get("damping", 0.0)),
                                     armature=float(a.get("armature", 0.0)),
                                     frictionloss=float(a.get("frictionloss", 0.0)),
                                     solref_limit=_floats(a.get("solreflimit"), 2, DEFAULT_SOLREF),
                                     solimp_limit=_floats(a.get("solimplimit"), 5, DEFAULT_SOLIMP),
                                     solref_fric=_floats(a.get("solreffriction"), 2, DEFAULT_SOLREF),
                                     solimp_fric=_floats(a.get("solimpfriction"), 5, DEFAULT_SOLIMP),
                                     margin=float(a.get("margin", 0.0)))
                        joints.append(j)
                        b["jnts"].append(jid)
                walk(ch, bid, cc)

    for wb in root.findall("worldbody"):
        walk(wb, 0, None)

    nbody, njnt = len(bodies), len(joints)

    # ---- addresses
    qposadr, dofadr = [], []
    nq = nv = 0
    for j in joints:
        qposadr.append(nq); dofadr.append(nv)
        if j["type"] == JNT_FREE:
            nq += 7; nv += 6
        else:
            nq += 1; nv += 1
    jnt_qposadr = np.array(qposadr, np.int32); jnt_dofadr = np.array(dofadr, np.int32)

    body_parentid = np.array([b["parent"] for b in bodies], np.int32)
    body_jntadr = np.array([b["jnts"][0] if b["jnts"] else -1 for b in bodies], np.int32)
    body_jntnum = np.array([len(b["jnts"]) for b in bodies], np.int32)
    body_dofadr = np.array([jnt_dofadr[b["jnts"][0]] if b["jnts"] else -1 for b in bodies], np.int32)
    body_dofnum = np.array([sum(6 if joints[j]["type"] == JNT_FREE else 1 for j in b["jnts"]) for b in bodies], np.int32)
    body_rootid = np.zeros(nbody, np.int32)
    body_weldid = np.zeros(nbody, np.int32)
    for i in range(1, nbody):
        p = body_parentid[i]
        body_rootid[i] = i if p == 0 else body_rootid[p]
        body_weldid[i] = i if body_jntnum[i] > 0 else body_weldid[p]

    dof_bodyid = np.zeros(nv, np.int32); dof_jntid = np.zeros(nv, np.int32)
    dof_parentid = -np.ones(nv, np.int32)
    dof_armature = np.zeros(nv); dof_damping = np.zeros(nv); dof_frictionloss = np.zeros(nv)
    for jid, j in enumerate(joints):
        n = 6 if j["type"] == JNT_FREE else 1
        for k in range(n):
            d = jnt_dofadr[jid] + k
            dof_bodyid[d] = j["body"]; dof_jntid[d] = jid
            dof_armature[d] = j["armature"]; dof_damping[d] = j["damping"]
            dof_frictionloss[d] = j["frictionloss"]
    # parent dof: previous dof on same body, else last dof of the nearest jointed ancestor
    last_dof_of_body = -np.ones(nbody, np.int32)
    for b in range(nbody):
        if body_dofnum[b] > 0:
            last_dof_of_body[b] = body_dofadr[b] + body_dofnum[b] - 1
    for d in range(nv):
        b = dof_bodyid[d]
        if d > body_dofadr[b]:
            dof_parentid[d] = d - 1
        else:
            p = body_parentid[b]
            while p > 0 and body_dofnum[p] == 0:
                p = body_parentid[p]
            dof_parentid[d] = last_dof_of_body[p] if p > 0 else -1

    qpos0 = np.zeros(nq)
    for jid, j in enumerate(joints):
        if j["type"] == JNT_FREE:
            b = bodies[j["body"]]
            qpos0[jnt_qposadr[jid]: jnt_qposadr[jid] + 3] = b["pos"]
            qpos0[jnt_qposadr[jid] + 3: jnt_qposadr[jid] + 7] = b["quat"]

    # ---- actuators (position servos)
    act = []
    joint_by_name = {j["name"]: i for i, j in enumerate(joints)}
    for an in root.findall("actuator"):
        for p in an:
            if p.tag != "position":
                raise NotImplementedError(f"actuator <{p.tag}>")
            a = attrs(p, "position", None)
            jid = joint_by_name[a["joint"]]
            kp = float(a.get("kp", 1.0)); kv = float(a.get("kv", 0.0))
            if "ctrlrange" in a:
                cr = _floats(a["ctrlrange"]); cl = 1
            elif float(a.get("inheritrange", 0)) > 0:
                r = joints[jid]["range"]; ir = float(a["inheritrange"])
                mean, half = 0.5 * (r[0] + r[1]), 0.5 * (r[1] - r[0]) * ir
                cr = np.array([mean - half, mean + half]); cl = 1
            else:
                cr = np.zeros(2); cl = 0
            fr = _floats(a.get("forcerange"), 2, [0, 0]); fl = int("forcerange" in a)
            act.append(dict(name=a.get("name", ""), jnt=jid, kp=kp, kv=kv, ctrlrange=cr, ctrllimited=cl,
                            forcerange=fr, forcelimited=fl, gear=float(a.get("gear", "1").split()[0])))
    nu = len(act)

    # ---- keyframes
    key = {}
    for kf in root.findall("keyframe"):
        for k in kf.findall("key"):
            key[k.attrib.get("name", "")] = dict(
                qpos=_floats(k.attrib.get("qpos"), nq, qpos0), ctrl=_floats(k.attrib.get("ctrl"), nu, [0] * nu))

    # ---- sections that would change the physics but have no counterpart anywhere in this build: refuse, never ignore
    for tag in ("tendon", "contact"):
        if any(len(e) for e in root.findall(tag)):
            raise NotImplementedError(f"<{tag}> is not supported by this engine (Open Duck scenes have none)")

    # ---- <equality>: joint / connect / weld (reference README.md:74-85 "adding a new robot"; the duck's own <equality/> is empty:
    # open_duck_mini_v2.xml:502).  Compiled into mjModel's eq_* arrays (mjtEq: connect 0, weld 1, joint 2); the float64 oracle builds their
    # rows, `odk_model_load` accepts what the kernels' row code expresses and refuses the rest by name.
    body_by_name = {b["name"]: i for i, b in enumerate(bodies) if b["name"]}
    jnt_by_name = {j["name"]: i for i, j in enumerate(joints) if j["name"]}
    eqs = []
    for eq_root in root.findall("equality"):
        for e in eq_root:
            a = dfl.get(e.attrib.get("class"), "equality")
            a.update(e.attrib)
            common = dict(solref=_floats(a.get("solref"), 2, DEFAULT_SOLREF), solimp=_floats(a.get("solimp"), 5, DEFAULT_SOLIMP),
                          active=int(a.get("active", "true") == "true"), name=a.get("name", ""))
            data = np.zeros(11)
            if e.tag == "joint":
                j1 = jnt_by_name[a["joint1"]]
                j2 = jnt_by_name[a["joint2"]] if "joint2" in a else -1
                if joints[j1]["type"] != JNT_HINGE or (j2 >= 0 and joints[j2]["type"] != JNT_HINGE):
                    raise NotImplementedError("<equality><joint> couples scalar (hinge) joints")
                data[:5] = _floats(a.get("polycoef"), 5, [0, 1, 0, 0, 0])
                eqs.append(dict(type=2, obj1=j1, obj2=j2, data=data, **common))
            elif e.tag in ("connect", "weld"):
                if "site1" in a or "site2" in a:
                    raise NotImplementedError(f"<equality><{e.tag}> by sites: give body1 / body2 (+ anchor)")
                b1 = body_by_name[a["body1"]]
                b2 = body_by_name[a["body2"]] if "body2" in a else 0
                eqs.append(dict(type=0 if e.tag == "connect" else 1, obj1=b1, obj2=b2, data=data, attrs=a, **common))
            else:
                raise NotImplementedError(f"<equality><{e.tag}> (joint, connect and weld are compiled)")

    # ---- collision geoms (everything with contype|conaffinity != 0)
    col = [g for g in geoms if (g["contype"] or g["conaffinity"])]
    for g in col:
        if g["type"] == "box":
            # a colliding box = the convex hull of its eight corners, which is how MJX collides a box with a plane, a height field
            # or a mesh (collision_driver: plane_convex / hfield_convex / convex_convex take boxes as convex meshes)
            if len(g["size"]) != 3:
                raise ValueError(f"box geom '{g['name']}' needs size='hx hy hz'")
            g["box_corners"] = np.array([[sx * g["size"][0], sy * g["size"][1], sz * g["size"][2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64)
        elif g["type"] == "sphere":
            if len(g["size"]) < 1:
                raise ValueError(f"sphere geom '{g['name']}' needs size='r'")
        elif g["type"] == "capsule":
            if g["fromto"] is not None:      # MuJoCo: the geom frame's z axis runs from -> to, pos = the midpoint, size = radius (half length derived)
                p0, p1 = g["fromto"][:3], g["fromto"][3:]
                ax = p1 - p0; hl = 0.5 * float(np.linalg.norm(ax)); ax = ax / (2 * hl)
                g["pos"] = 0.5 * (p0 + p1)
                z = np.array([0.0, 0.0, 1.0]); c = float(z @ ax)
                if c < -1 + 1e-12:
                    g["quat"] = np.array([0.0, 1.0, 0.0, 0.0])
                else:
                    w = np.cross(z, ax); q = np.array([1.0 + c, w[0], w[1], w[2]]); g["quat"] = q / np.linalg.norm(q)
                g["size"] = np.array([float(g["size"][0]), hl])
            if len(g["size"]) < 2:
                raise ValueError(f"capsule geom '{g['name']}' needs size='r half_length' or fromto")
        elif g["type"] not in ("plane", "hfield", "mesh"):
            raise NotImplementedError(f"colliding geom '{g['name']}' of type {g['type']}: plane / hfield floors, convex meshes, boxes, "
                                      "spheres and capsules collide in this engine; ellipsoids and cylinders do not (give visual "
                                      "primitives contype=conaffinity=0)")
    for g in col:
        if g.get("margin", 0.0) != 0.0 or g.get("gap", 0.0) != 0.0:
            raise NotImplementedError(f"colliding geom '{g['name']}' has margin / gap: contacts are detected at distance 0 in this engine "
                                      "(the height-field cull and the foot-foot box cull drop separated pairs)")
    if any(g["type"] == "hfield" for g in col):
        kinds = {("prim" if g["type"] in ("sphere", "capsule") else "hull") for g in col if g["type"] != "hfield"}
        if len(kinds) > 1:   # (odk_model_load repeats the check on the blob)
            raise NotImplementedError("on a height-field floor both feet are hulls (meshes / boxes: hfield_convex) or both are spheres / "
                                      "capsules (hfield_sphere / hfield_capsule)")
    col_ids = [i for i, g in enumerate(geoms) if (g["contype"] or g["conaffinity"])]
    geom_name2id = {g["name"]: i for i, g in enumerate(geoms) if g["name"]}

    out: Dict[str, np.ndarray] = {}
    I32 = lambda x: np.asarray(x, np.int32)
    F64 = lambda x: np.asarray(x, np.float64)
    out["nq"], out["nv"], out["nu"], out["nbody"], out["njnt"] = I32([nq]), I32([nv]), I32([nu]), I32([nbody]), I32([njnt])
    out["ngeom"], out["nsite"] = I32([len(geoms)]), I32([len(sites)])
    out["opt_timestep"] = F64([opt["timestep"]]); out["opt_gravity"] = F64(opt["gravity"])
    out["opt_tolerance"] = F64([opt["tolerance"]]); out["opt_ls_tolerance"] = F64([opt["ls_tolerance"]])
    out["opt_impratio"] = F64([opt["impratio"]]); out["opt_cone"] = I32([opt["cone"]])
    out["opt_iterations"] = I32([opt["iterations"]]); out["opt_ls_iterations"] = I32([opt["ls_iterations"]])
    out["opt_eulerdamp"] = I32([opt["eulerdamp"]])

    out["body_parentid"], out["body_rootid"], out["body_weldid"] = body_parentid, body_rootid, body_weldid
    out["body_jntadr"], out["body_jntnum"], out["body_dofadr"], out["body_dofnum"] = body_jntadr, body_jntnum, body_dofadr, body_dofnum
    out["body_pos"] = F64([b["pos"] for b in bodies]); out["body_quat"] = F64([b["quat"] for b in bodies])
    out["body_mass"] = F64([b["mass"] for b in bodies]); out["body_ipos"] = F64([b["ipos"] for b in bodies])
    # full symmetric inertia about the body COM, in the body frame: xx yy zz xy xz yz
    out["body_inertia_full"] = F64([[b["inertia"][0, 0], b["inertia"][1, 1], b["inertia"][2, 2],
                                     b["inertia"][0, 1], b["inertia"][0, 2], b["inertia"][1, 2]] for b in bodies])
    # MuJoCo's own representation (principal moments + frame) for readers used to mjModel
    iq, idiag = [], []
    for b in bodies:
        w, V = np.linalg.eigh(b["inertia"])
        order = np.argsort(-w)  # MuJoCo sorts principal moments in decreasing order
        w, V = w[order], V[:, order]
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        idiag.append(w); iq.append(_mat_to_quat(V))
    out["body_inertia"] = F64(idiag); out["body_iquat"] = F64(iq)

    out["jnt_type"] = I32([j["type"] for j in joints]); out["jnt_bodyid"] = I32([j["body"] for j in joints])
    out["jnt_qposadr"], out["jnt_dofadr"] = jnt_qposadr, jnt_dofadr
    out["jnt_pos"] = F64([j["pos"] for j in joints]); out["jnt_axis"] = F64([j["axis"] for j in joints])
    out["jnt_range"] = F64([j["range"] for j in joints]); out["jnt_limited"] = I32([j["limited"] for j in joints])
    out["jnt_solref"] = F64([j["solref_limit"] for j in joints]); out["jnt_solimp"] = F64([j["solimp_limit"] for j in joints])
    out["jnt_margin"] = F64([j["margin"] for j in joints])
    out["dof_bodyid"], out["dof_jntid"], out["dof_parentid"] = dof_bodyid, dof_jntid, dof_parentid
    out["dof_armature"], out["dof_damping"], out["dof_frictionloss"] = dof_armature, dof_damping, dof_frictionloss
    out["dof_solref"] = F64([joints[dof_jntid[d]]["solref_fric"] for d in range(nv)])
    out["dof_solimp"] = F64([joints[dof_jntid[d]]["solimp_fric"] for d in range(nv)])
    out["qpos0"] = qpos0

    out["actuator_trnid"] = I32([a["jnt"] for a in act])
    out["actuator_gainprm0"] = F64([a["kp"] for a in act])
    out["actuator_biasprm"] = F64([[0.0, -a["kp"], -a["kv"]] for a in act])
    out["actuator_ctrlrange"] = F64([a["ctrlrange"] for a in act]); out["actuator_ctrllimited"] = I32([a["ctrllimited"] for a in act])
    out["actuator_forcerange"] = F64([a["forcerange"] for a in act]); out["actuator_forcelimited"] = I32([a["forcelimited"] for a in act])
    out["actuator_gear"] = F64([a["gear"] for a in act])

    home = key.get("home", dict(qpos=qpos0, ctrl=np.zeros(nu)))
    out["key_qpos"] = F64(home["qpos"]); out["key_ctrl"] = F64(home["ctrl"])

    out["site_bodyid"] = I32([s["body"] for s in sites]); out["site_pos"] = F64([s["pos"] for s in sites])
    out["site_quat"] = F64([s["quat"] for s in sites])

    # sensors
    site_by_name = {s["name"]: i for i, s in enumerate(sites)}
    stype, sobj, sadr, sdim = [], [], [], []
    adr = 0
    sensor_names = []
    for sn in root.findall("sensor"):
        for s in sn:
            if s.tag not in _SENSOR_TAGS:
                raise NotImplementedError(f"sensor <{s.tag}>")
            code, dim = _SENSOR_TAGS[s.tag]
            if "site" in s.attrib:
                obj = site_by_name[s.attrib["site"]]
            else:
                if s.attrib.get("objtype") != "site":
                    raise NotImplementedError("frame sensors on non-site objects")
                obj = site_by_name[s.attrib["objname"]]
            stype.append(code); sobj.append(obj); sadr.append(adr); sdim.append(dim); adr += dim
            sensor_names.append(s.attrib.get("name", ""))
    out["sensor_type"], out["sensor_objid"], out["sensor_adr"], out["sensor_dim"] = I32(stype), I32(sobj), I32(sadr), I32(sdim)
    out["nsensordata"] = I32([adr])

    # collision geoms: the scenes have exactly {plane|hfield floor, 2 convex foot meshes}
    ncol = len(col)
    out["cgeom_id"] = I32(col_ids)
    out["cgeom_type"] = I32([{"plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "mesh": GEOM_MESH, "box": GEOM_MESH, "sphere": GEOM_SPHERE,
                              "capsule": GEOM_CAPSULE}[g["type"]] for g in col])
    out["cgeom_size"] = F64([list(g["size"][:3]) + [0.0] * (3 - min(len(g["size"]), 3)) if g["type"] in ("sphere", "capsule") else [0.0, 0.0, 0.0] for g in col])
    out["cgeom_bodyid"] = I32([g["body"] for g in col])
    out["cgeom_pos"] = F64([g["pos"] for g in col]); out["cgeom_quat"] = F64([g["quat"] for g in col])
    out["cgeom_friction"] = F64([g["friction"] for g in col])
    out["cgeom_priority"] = I32([g["priority"] for g in col]); out["cgeom_condim"] = I32([g["condim"] for g in col])
    out["cgeom_contype"] = I32([g["contype"] for g in col]); out["cgeom_conaffinity"] = I32([g["conaffinity"] for g in col])
    out["cgeom_solref"] = F64([g["solref"] for g in col]); out["cgeom_solimp"] = F64([g["solimp"] for g in col])
    out["cgeom_solmix"] = F64([g["solmix"] for g in col])
    out["cgeom_margin"] = F64([g.get("margin", 0.0) for g in col])      # always 0 (refused above); odk_model_load checks it again
    # convex hulls (all collision meshes here share one asset, but keep it general: concat + adr)
    vadr, vnum, fadr, fnum, allv, allf = [], [], [], [], [], []
    cache = {}
    for g in col:
        if g["type"] in ("mesh", "box"):
            ckey = g["mesh"] if g["type"] == "mesh" else ("box",) + tuple(g["size"])
            if ckey not in cache:
                hv, hf = convex_hull(load_stl_vertices(meshes[g["mesh"]]) if g["type"] == "mesh" else g["box_corners"])
                cache[ckey] = (sum(len(v) for v in allv), len(hv), sum(len(f) for f in allf), len(hf))
                allv.append(hv); allf.append(hf)
            va, vn, fa, fn = cache[ckey]
        else:
            va, vn, fa, fn = 0, 0, 0, 0
        vadr.append(va); vnum.append(vn); fadr.append(fa); fnum.append(fn)
    out["cgeom_vertadr"], out["cgeom_vertnum"] = I32(vadr), I32(vnum)
    out["cgeom_faceadr"], out["cgeom_facenum"] = I32(fadr), I32(fnum)
    out["hull_vert"] = F64(np.concatenate(allv)) if allv else np.zeros((0, 3))
    out["hull_face"] = I32(np.concatenate(allf)) if allf else np.zeros((0, 3), np.int32)
    # height field (rough terrain): raw elevation grid, MuJoCo normalises PNG data to [0,1]
    for g in col:
        if g["type"] == "hfield":
            hf = hfields[g["hfield"]]
            out["hfield_size"] = F64(hf["size"])
            out["hfield_data"] = _load_hfield_png(hf["file"])
    # named ids the env needs (reference joystick.py:158-181)
    out["id_floor_geom"] = I32([geom_name2id.get("floor", -1)])
    out["names_body"] = np.array([b["name"] for b in bodies])
    out["names_jnt"] = np.array([j["name"] for j in joints])
    out["names_geom"] = np.array([g["name"] for g in geoms])
    out["names_site"] = np.array([s["name"] for s in sites])
    out["names_sensor"] = np.array(sensor_names)
    out["names_actuator"] = np.array([a["name"] for a in act])

    # equality constraints: the second anchor / the relative pose come from the reference configuration qpos0 (MuJoCo's compiler does
    # the same when they are not given), so that the constraint is satisfied there
    if eqs:
        xpos0, xquat0, _, _ = body_frames(out, qpos0)
        for q in eqs:
            if q["type"] == 2:
                continue
            a, b1, b2 = q["attrs"], q["obj1"], q["obj2"]
            R1, R2 = quat_to_mat(xquat0[b1]), quat_to_mat(xquat0[b2])
            anchor = _floats(a.get("anchor"), 3, [0, 0, 0])
            if q["type"] == 0:          # connect: anchor in body1's frame; data[3:6] = the same world point in body2's frame
                q["data"][0:3] = anchor
                q["data"][3:6] = R2.T @ (xpos0[b1] + R1 @ anchor - xpos0[b2])
            else:                       # weld: anchor in body2's frame (data[0:3]); data[3:6] = that point in body1's frame; data[6:10] = body2 relative to body1
                rel = _floats(a.get("relpose"), 7, [0, 1, 0, 0, 0, 0, 0])
                q["data"][0:3] = anchor
                if np.any(rel[3:] != 0):
                    rq = _normalize(rel[3:])
                    q["data"][3:6] = rel[:3] + quat_to_mat(rq) @ anchor
                    q["data"][6:10] = rq
                else:
                    q["data"][3:6] = R1.T @ (xpos0[b2] + R2 @ anchor - xpos0[b1])
                    q["data"][6:10] = quat_mul(np.array([xquat0[b1][0], -xquat0[b1][1], -xquat0[b1][2], -xquat0[b1][3]]), xquat0[b2])
                q["data"][10] = float(a.get("torquescale", 1.0))
        out["eq_type"] = I32([q["type"] for q in eqs]); out["eq_obj1id"] = I32([q["obj1"] for q in eqs]); out["eq_obj2id"] = I32([q["obj2"] for q in eqs])
        out["eq_data"] = F64([q["data"] for q in eqs]); out["eq_solref"] = F64([q["solref"] for q in eqs]); out["eq_solimp"] = F64([q["solimp"] for q in eqs])
        out["eq_active"] = I32([q["active"] for q in eqs])
        out["names_eq"] = np.array([q["name"] for q in eqs])
    out["neq"] = I32([len(eqs)])

    _set_const(out)
    return out


def _mat_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def _load_hfield_png(path: str) -> np.ndarray:
    from PIL import Image

    img = np.asarray(Image.open(path).convert("L"), dtype=np.float64)
    img = img[::-1]  # MuJoCo flips image rows so that row 0 is the -y edge
    lo, hi = img.min(), img.max()
    return (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)


# ----------------------------------------------------------------------------- mj_setConst
def body_frames(m: Dict[str, np.ndarray], qpos: np.ndarray):
    """World position/rotation of every body at `qpos` (numpy float64; used at compile time
    and by tests as an independent check of the oracle's kinematics)."""
    nbody = int(m["nbody"][0])
    xpos = np.zeros((nbody, 3)); xquat = np.zeros((nbody, 4)); xquat[0, 0] = 1
    xanchor = np.zeros((int(m["njnt"][0]), 3)); xaxis = np.zeros((int(m["njnt"][0]), 3))
    for b in range(1, nbody):
        p = m["body_parentid"][b]
        jn, ja = m["body_jntnum"][b], m["body_jntadr"][b]
        if jn == 1 and m["jnt_type"][ja] == JNT_FREE:
            a = m["jnt_qposadr"][ja]
            pos = qpos[a:a + 3].copy(); quat = _normalize(qpos[a + 3:a + 7])
            xanchor[ja] = pos; xaxis[ja] = [0, 0, 1]
        else:
            pos = xpos[p] + quat_to_mat(xquat[p]) @ m["body_pos"][b]
            quat = quat_mul(xquat[p], m["body_quat"][b])
            for j in range(ja, ja + jn):
                a = m["jnt_qposadr"][j]
                R = quat_to_mat(quat)
                xanchor[j] = R @ m["jnt_pos"][j] + pos
                xaxis[j] = R @ m["jnt_axis"][j]
                ang = qpos[a] - m["qpos0"][a]
                ax = m["jnt_axis"][j]
                qj = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
                quat = quat_mul(quat, qj)
                # correct for off-centre rotation
                pos = xanchor[j] - quat_to_mat(quat) @ m["jnt_pos"][j]
        xpos[b], xquat[b] = pos, _normalize(quat)
    return xpos, xquat, xanchor, xaxis


def dof_jacobians(m, qpos):
    """For every dof d: world angular axis w_d and a point a_d so that a point p rigidly attached
    to a descendant body moves with  v = lin_d + w_d x (p - a_d)  per unit qvel[d]."""
    nv = int(m["nv"][0])
    xpos, xquat, xanchor, xaxis = body_frames(m, qpos)
    w = np.zeros((nv, 3)); lin = np.zeros((nv, 3)); anchor = np.zeros((nv, 3))
    for j in range(int(m["njnt"][0])):
        d = m["jnt_dofadr"][j]; b = m["jnt_bodyid"][j]
        if m["jnt_type"][j] == JNT_FREE:
            R = quat_to_mat(xquat[b])
            for k in range(3):
                lin[d + k] = np.eye(3)[k]
                w[d + 3 + k] = R[:, k]; anchor[d + 3 + k] = xpos[b]
        else:
            w[d] = xaxis[j]; anchor[d] = xanchor[j]
    return xpos, xquat, w, lin, anchor


def mass_matrix(m, qpos, body_mass=None):
    """Joint-space inertia via the Jacobian sum  M = sum_b Jb^T diag(m, I_b) Jb + armature
    (deliberately NOT the composite-rigid-body algorithm the oracle/kernels use)."""
    nv, nbody = int(m["nv"][0]), int(m["nbody"][0])
    mass = m["body_mass"] if body_mass is None else body_mass
    xpos, xquat, w, lin, anchor = dof_jacobians(m, qpos)
    M = np.diag(np.asarray(m["dof_armature"], float).copy())
    jacs = []
    for b in range(nbody):
        R = quat_to_mat(xquat[b])
        com = xpos[b] + R @ m["body_ipos"][b]
        f = m["body_inertia_full"][b]
        Ib = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
        Iw = R @ Ib @ R.T
        Jp = np.zeros((3, nv)); Jr = np.zeros((3, nv))
        d = -1
        # dofs affecting body b: walk up from its last dof
        bb = b
        while bb > 0 and m["body_dofnum"][bb] == 0:
            bb = m["body_parentid"][bb]
        if bb > 0:
            d = m["body_dofadr"][bb] + m["body_dofnum"][bb] - 1
        while d >= 0:
            Jr[:, d] = w[d]
            Jp[:, d] = lin[d] + np.cross(w[d], com - anchor[d])
            d = m["dof_parentid"][d]
        jacs.append((Jp, Jr))
        M += mass[b] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    return M, jacs


def _set_const(m: Dict[str, np.ndarray]):
    """dof_invweight0 / body_invweight0 / stat.meaninertia / body_subtreemass at qpos0,
    following MuJoCo's mj_setConst (engine_setconst.c, [UPSTREAM-MEMORY])."""
    nv, nbody = int(m["nv"][0]), int(m["nbody"][0])
    M, jacs = mass_matrix(m, m["qpos0"])
    Minv = np.linalg.inv(M)
    inv = np.zeros(nv)
    for j in range(int(m["njnt"][0])):
        d = m["jnt_dofadr"][j]
        if m["jnt_type"][j] == JNT_FREE:
            inv[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
            inv[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
        else:
            inv[d] = Minv[d, d]
    m["dof_invweight0"] = inv
    biw = np.zeros((nbody, 2))
    for b in range(1, nbody):
        if m["body_weldid"][b] == 0:
            continue  # static body
        Jp, Jr = jacs[b]
        biw[b, 0] = np.trace(Jp @ Minv @ Jp.T) / 3.0
        biw[b, 1] = np.trace(Jr @ Minv @ Jr.T) / 3.0
    m["body_invweight0"] = biw
    m["stat_meaninertia"] = np.array([np.mean(np.diag(M))])
    sub = np.asarray(m["body_mass"], float).copy()
    for b in range(nbody - 1, 0, -1):
        sub[m["body_parentid"][b]] += sub[b]
    m["body_subtreemass"] = sub
