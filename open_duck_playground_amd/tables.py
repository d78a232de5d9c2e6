"""Static topology tables for the HIP kernels, derived from the compiled model.

The kernels map one lane to one body / dof / matrix entry / constraint row; everything that
would be a pointer chase (`parent[parent[...]]`) is flattened here once, at model-load time,
into small integer tables that ride in the ModelBlob as `k_*` records (the native loader in
csrc/odk_engine.hip copies them verbatim).  Layout conventions:

* Sparse inertia `M` uses MuJoCo's qM layout: row i starts at `Madr[i]` and holds
  M[i, anc_0(i)=i], M[i, anc_1(i)=parent], ... up to the root.
* The Newton Hessian `H` uses the same layout over a *virtual* tree in which the second foot's
  leg chain hangs below the first foot's last dof.  Foot-foot contacts couple the two legs; in
  the virtual tree that coupling is part of the ancestor pattern, so the same fill-free
  L^T D L works ("Hessian sparsity").
"""
from __future__ import annotations

from typing import Dict

import numpy as np

MAXV = 32     # dofs
MAXB = 20     # bodies
MAXCHAIN = 8  # bodies between the floating base and a leaf
MAXNZ = 512   # sparse entries
JNT_FREE = 0


def _anc_lists(parent: np.ndarray):
    nv = len(parent)
    anc = -np.ones((nv, MAXV), np.int32)
    depth = np.zeros(nv, np.int32)
    for i in range(nv):
        k, j = 0, i
        while j >= 0:
            anc[i, k] = j
            k += 1
            j = parent[j]
        depth[i] = k - 1
    return anc, depth


def _sparse_layout(parent: np.ndarray):
    anc, depth = _anc_lists(parent)
    nv = len(parent)
    adr = np.zeros(nv, np.int32)
    n = 0
    for i in range(nv):
        adr[i] = n
        n += depth[i] + 1
    ei, ej = np.zeros(n, np.int32), np.zeros(n, np.int32)
    # depth-indexed rows: entry c of row i is the column of i's ancestor at depth c (c = depth[i] is the diagonal)
    ancd = -np.ones((nv, MAXV), np.int32)
    ancmask = np.zeros(nv, np.int64); descmask = np.zeros(nv, np.int64)
    for i in range(nv):
        for k in range(depth[i] + 1):
            c = depth[i] - k
            ei[adr[i] + c] = i
            ej[adr[i] + c] = anc[i, k]
            ancd[i, c] = anc[i, k]
            if k > 0:
                ancmask[i] |= (1 << int(anc[i, k])); descmask[anc[i, k]] |= (1 << i)
    # descendants: for column j, every row k>j with j in anc(k), plus the address of entry (k, j)
    ndesc = np.zeros(nv, np.int32)
    desc = -np.ones((nv, MAXV), np.int32)
    desc_adr = -np.ones((nv, MAXV), np.int32)
    for k in range(nv):
        for q in range(1, depth[k] + 1):
            j = anc[k, q]
            desc[j, ndesc[j]] = k
            desc_adr[j, ndesc[j]] = adr[k] + depth[j]
            ndesc[j] += 1
    # per-step ancestor row addresses for the factorisation: anc_adr[k, m] = adr[anc_m(k)]
    anc_adr = -np.ones((nv, MAXV), np.int32)
    for k in range(nv):
        for q in range(depth[k] + 1):
            anc_adr[k, q] = adr[anc[k, q]]
    to_i32 = lambda v: np.array([int(x) - (1 << 32) if int(x) >= (1 << 31) else int(x) for x in v], np.int32)
    return dict(anc=anc, depth=depth, adr=adr, nnz=n, ei=ei, ej=ej, ndesc=ndesc, desc=desc, desc_adr=desc_adr, anc_adr=anc_adr,
                ancd=ancd, ancmask=to_i32(ancmask), descmask=to_i32(descmask))


def reduced_layout(a: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Python mirror of csrc/odk_engine.hip build_reduced_tables (twin dofs merged into their main dof): used by the
    parity tests to address the kernels' reduced inertia `M`.  A twin is a hinge declared right after another hinge on
    the same body with the same axis (backlash joints): identical motion column, so the kernels keep one column per
    pair and work on the twin-free tree.  Returns `main`, `twin` (-1: none) per reduced dof and, per entry of the
    reduced row layout (row r = r's ancestors by depth), the pair of FULL dof indices (`ei`, `ej`) whose inertia
    element the entry holds -- for a pair's diagonal that is (twin, main), the element without armature."""
    nv, nj = int(a["nv"][0]), int(a["njnt"][0])
    parent = np.asarray(a["dof_parentid"], np.int64)
    dof_jnt = -np.ones(nv, np.int64)
    for j in range(1, nj):
        dof_jnt[a["jnt_dofadr"][j]] = j
    kind = np.zeros(nv, np.int64)
    for v in range(7, nv):
        u = v - 1
        ju, jv = dof_jnt[u], dof_jnt[v]
        if ju < 0 or jv < 0 or kind[u] != 0:
            continue
        if a["dof_bodyid"][u] == a["dof_bodyid"][v] and parent[v] == u and np.array_equal(a["jnt_axis"][ju], a["jnt_axis"][jv]):
            kind[u], kind[v] = 1, 2
    red = np.zeros(nv, np.int64); main, twin = [], []
    for d in range(nv):
        if kind[d] == 2:
            red[d] = red[d - 1]
            continue
        red[d] = len(main); main.append(d); twin.append(d + 1 if kind[d] == 1 else -1)
    rparent = np.array([-1 if parent[u] < 0 else red[parent[u]] for u in main], np.int32)
    lay = _sparse_layout(rparent)
    ei = np.array([twin[i] if (i == j and twin[i] >= 0) else main[i] for i, j in zip(lay["ei"], lay["ej"])], np.int64)
    ej = np.array([main[j] for j in lay["ej"]], np.int64)
    return dict(main=np.array(main), twin=np.array(twin), red=red, kind=kind, ei=ei, ej=ej, nnz=lay["nnz"])


def build_kernel_tables(a: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    nv, nb, nj, nu = int(a["nv"][0]), int(a["nbody"][0]), int(a["njnt"][0]), int(a["nu"][0])
    if nv > MAXV or nb > MAXB:
        raise ValueError("model too large for the kernel tables")
    if a["jnt_type"][0] != JNT_FREE or (a["jnt_type"][1:] == JNT_FREE).any():
        raise ValueError("kernels expect exactly one free joint, first")
    parent = np.asarray(a["dof_parentid"], np.int32)
    bparent = np.asarray(a["body_parentid"], np.int32)
    base = int(a["jnt_bodyid"][0])
    out: Dict[str, np.ndarray] = {}
    I = lambda x: np.asarray(x, np.int32)

    # ---- bodies: chain below the floating base, affecting dofs, subtree
    chain = -np.ones((nb, MAXCHAIN), np.int32); chain_len = np.zeros(nb, np.int32)
    in_tree = np.zeros(nb, np.int32)
    for b in range(nb):
        path, c = [], b
        while c > 0 and c != base:
            path.append(c); c = bparent[c]
        if c == base:
            in_tree[b] = 1
            path.reverse()
            if len(path) > MAXCHAIN:
                raise ValueError("kinematic chain too deep")
            chain[b, :len(path)] = path; chain_len[b] = len(path)
    out["k_base_body"] = I([base]); out["k_body_in_tree"] = in_tree
    out["k_body_chain"] = chain; out["k_body_chain_len"] = chain_len
    ancdof = -np.ones((nb, MAXV), np.int32); nancdof = np.zeros(nb, np.int32)
    for b in range(nb):
        if not in_tree[b]:
            continue
        c = b
        while c > 0 and a["body_dofnum"][c] == 0:
            c = bparent[c]
        d = a["body_dofadr"][c] + a["body_dofnum"][c] - 1
        lst = []
        while d >= 0:
            lst.append(d); d = parent[d]
        lst.reverse()
        ancdof[b, :len(lst)] = lst; nancdof[b] = len(lst)
    out["k_body_ancdof"] = ancdof; out["k_body_nancdof"] = nancdof
    level = -np.ones(nb, np.int32); children = -np.ones((nb, 3), np.int32); nchild = np.zeros(nb, np.int32)
    for b in range(nb):
        if in_tree[b]:
            level[b] = chain_len[b]
            if b != base:
                p_ = bparent[b]
                if nchild[p_] >= 3:
                    raise ValueError("more than three child bodies")
                children[p_, nchild[p_]] = b; nchild[p_] += 1
    out["k_body_level"] = level; out["k_body_children"] = children; out["k_body_nchild"] = nchild
    out["k_max_level"] = I([int(level.max())])
    # "path" bodies: the subtree below (and including) the body is a serial chain with consecutive ids, so subtree
    # sums are suffix sums over neighbouring lanes (3 Hillis-Steele steps) instead of one tree level per step.
    # bit s of pathmask[b]: body b + 2^s belongs to b's chain below it.
    is_path = np.zeros(nb, np.int32)
    for b in range(nb - 1, 0, -1):
        if in_tree[b] and (nchild[b] == 0 or (nchild[b] == 1 and children[b, 0] == b + 1 and is_path[b + 1])):
            is_path[b] = 1
    plen = np.zeros(nb, np.int32)   # bodies in the chain from b downwards
    for b in range(nb - 1, 0, -1):
        if is_path[b]:
            plen[b] = 1 + (plen[b + 1] if nchild[b] == 1 else 0)
    if plen.max() > 8:
        raise ValueError("serial body chain longer than 8")
    pathmask = np.zeros(nb, np.int32)
    for b in range(nb):
        for si in range(3):
            if is_path[b] and (1 << si) < plen[b]:
                pathmask[b] |= 1 << si
    # upward masks for prefix scans from the chain head: bit s: body b - 2^s is in b's chain above it
    head = np.zeros(nb, np.int32); upmask = np.zeros(nb, np.int32); is_head = np.zeros(nb, np.int32)
    for b in range(1, nb):
        if is_path[b]:
            par = bparent[b]
            if is_path[par] and par == b - 1:
                head[b] = head[par]
            else:
                head[b] = b; is_head[b] = 1
            for si in range(3):
                if b - (1 << si) >= head[b]:
                    upmask[b] |= 1 << si
    out["k_body_upmask"] = upmask; out["k_body_path_head"] = is_head
    out["k_body_pathmask"] = pathmask; out["k_body_is_path"] = is_path
    nonpath_levels = [int(level[b]) for b in range(nb) if in_tree[b] and not is_path[b] and nchild[b] > 0]
    out["k_max_nonpath_level"] = I([max(nonpath_levels) if nonpath_levels else -1])
    sub = -np.ones((nb, MAXB), np.int32); nsub = np.zeros(nb, np.int32)
    for c in range(1, nb):
        b = c
        while b > 0:
            sub[b, nsub[b]] = c; nsub[b] += 1
            b = bparent[b]
    out["k_body_sub"] = sub; out["k_body_nsub"] = nsub

    # ---- dofs
    true = _sparse_layout(parent)
    out["k_dof_depth"] = true["depth"]; out["k_dof_anc"] = true["anc"]; out["k_dof_Madr"] = true["adr"]
    out["k_nM"] = I([true["nnz"]]); out["k_M_i"] = true["ei"]; out["k_M_j"] = true["ej"]
    out["k_dof_ndesc"] = true["ndesc"]; out["k_dof_desc"] = true["desc"]; out["k_dof_desc_adr"] = true["desc_adr"]
    out["k_dof_anc_adr"] = true["anc_adr"]
    out["k_dof_ancmask"] = true["ancmask"]; out["k_dof_descmask"] = true["descmask"]
    # chains: maximal runs of dofs i, i+1, ... with parent(i+1) = i whose first dof hangs off the last base dof.
    # A "tree of chains" lets one lane factor a whole chain block in registers (csrc chain_solve).
    chain_first, chain_len = [], []
    d = 6
    is_chain_tree = True
    while d < nv:
        if parent[d] != 5:
            is_chain_tree = False
            break
        e = d
        while e + 1 < nv and parent[e + 1] == e:
            e += 1
        chain_first.append(d); chain_len.append(e - d + 1)
        d = e + 1
    if not is_chain_tree or len(chain_first) > 3:
        chain_first, chain_len = [], []
    out["k_chain_first"] = I(chain_first + [0] * (3 - len(chain_first))); out["k_chain_len"] = I(chain_len + [0] * (3 - len(chain_len)))
    out["k_nchain"] = I([len(chain_first)])
    # velocity prefix (mj_comVel): strict ancestors, except that the free joint's rotational dofs
    # see only its three translational dofs
    prefix = -np.ones((nv, MAXV), np.int32); nprefix = np.zeros(nv, np.int32)
    for d in range(nv):
        if d < 3:
            lst = []
        elif d < 6:
            lst = [0, 1, 2]
        else:
            lst = list(true["anc"][d, 1:true["depth"][d] + 1][::-1])
        prefix[d, :len(lst)] = lst; nprefix[d] = len(lst)
    out["k_dof_prefix"] = prefix; out["k_dof_nprefix"] = nprefix
    # symmetric row lists for M @ v
    nsym = np.zeros(nv, np.int32); sym_dof = -np.ones((nv, MAXV), np.int32); sym_adr = -np.ones((nv, MAXV), np.int32)
    for p in range(true["nnz"]):
        i, j = true["ei"][p], true["ej"][p]
        sym_dof[i, nsym[i]] = j; sym_adr[i, nsym[i]] = p; nsym[i] += 1
        if i != j:
            sym_dof[j, nsym[j]] = i; sym_adr[j, nsym[j]] = p; nsym[j] += 1
    out["k_dof_nsym"] = nsym; out["k_dof_sym_dof"] = sym_dof; out["k_dof_sym_adr"] = sym_adr

    # ---- actuators / backlash twins (reference base.py:63-125)
    act_jnt = np.asarray(a["actuator_trnid"], np.int32)
    is_act = np.zeros(nj, bool); is_act[act_jnt] = True
    act_q = a["jnt_qposadr"][act_jnt]; act_d = a["jnt_dofadr"][act_jnt]
    bl_q = -np.ones(nu, np.int32)
    for u, j in enumerate(act_jnt):
        if j + 1 < nj and not is_act[j + 1] and a["jnt_bodyid"][j + 1] == a["jnt_bodyid"][j]:
            bl_q[u] = a["jnt_qposadr"][j + 1]
    out["k_act_qposadr"] = I(act_q); out["k_act_dofadr"] = I(act_d); out["k_act_backlash_qposadr"] = bl_q
    dof_act = -np.ones(nv, np.int32)
    dof_act[act_d] = np.arange(nu)
    out["k_dof_act"] = dof_act

    # ---- constraint rows: friction loss (dofs with frictionloss > 0), hinge limits, contacts
    fl_dofs = [d for d in range(nv) if a["dof_frictionloss"][d] > 0]
    lim_jnts = [j for j in range(nj) if a["jnt_limited"][j] and a["jnt_type"][j] != JNT_FREE]
    out["k_fl_dof"] = I(fl_dofs); out["k_lim_jnt"] = I(lim_jnts)
    dof_flrow = -np.ones(nv, np.int32); dof_limrow = -np.ones(nv, np.int32)
    for r, d in enumerate(fl_dofs):
        dof_flrow[d] = r
    for r, j in enumerate(lim_jnts):
        dof_limrow[a["jnt_dofadr"][j]] = r
    out["k_dof_flrow"] = dof_flrow; out["k_dof_limrow"] = dof_limrow

    # ---- feet / floor (collision geoms: exactly two foot colliders -- convex meshes / box hulls, or spheres / capsules -- and
    # one plane / hfield)
    ctype = a["cgeom_type"]
    feet = [g for g in range(len(ctype)) if ctype[g] in (2, 3, 7)]
    floor = [g for g in range(len(ctype)) if ctype[g] in (0, 1)]
    if len(feet) != 2 or len(floor) != 1:
        raise ValueError("kernels expect two foot colliders and one floor geom")
    out["k_foot_cgeom"] = I(feet); out["k_floor_cgeom"] = I(floor)
    foot_body = [int(a["cgeom_bodyid"][g]) for g in feet]
    out["k_foot_body"] = I(foot_body)
    # membership masks: dof d moves foot f
    mask = np.zeros((2, nv), np.int32)
    for f, b in enumerate(foot_body):
        mask[f, ancdof[b, :nancdof[b]]] = 1
    out["k_foot_dofmask"] = mask
    # hull AABB (geom-local) for the foot-foot cull
    obb_c, obb_h = [], []
    for g in feet:
        if ctype[g] != 7:   # primitive: the box around the sphere / capsule (its axis is the geom frame's z)
            r, hl = float(a["cgeom_size"][g][0]), float(a["cgeom_size"][g][1]) if ctype[g] == 3 else 0.0
            obb_c.append(np.zeros(3)); obb_h.append(np.array([r, r, r + hl]))
            continue
        v = a["hull_vert"][a["cgeom_vertadr"][g]: a["cgeom_vertadr"][g] + a["cgeom_vertnum"][g]]
        obb_c.append(0.5 * (v.min(0) + v.max(0))); obb_h.append(0.5 * (v.max(0) - v.min(0)))
    out["k_foot_obb_center"] = np.asarray(obb_c, np.float64); out["k_foot_obb_half"] = np.asarray(obb_h, np.float64)

    # ---- virtual tree for the Hessian: second leg below the first foot's last dof
    vparent = parent.copy()
    l_last = int(ancdof[foot_body[0], nancdof[foot_body[0]] - 1])
    r_chain = [d for d in ancdof[foot_body[1], :nancdof[foot_body[1]]] if not mask[0, d]]
    if r_chain and r_chain[0] > l_last:
        vparent[r_chain[0]] = l_last
    virt = _sparse_layout(vparent)
    if virt["nnz"] > MAXNZ:
        raise ValueError("Hessian pattern too large")
    out["k_vdof_depth"] = virt["depth"]; out["k_vdof_anc"] = virt["anc"]; out["k_vdof_Madr"] = virt["adr"]
    out["k_nH"] = I([virt["nnz"]]); out["k_H_i"] = virt["ei"]; out["k_H_j"] = virt["ej"]
    out["k_vdof_ndesc"] = virt["ndesc"]; out["k_vdof_desc"] = virt["desc"]; out["k_vdof_desc_adr"] = virt["desc_adr"]
    out["k_vdof_anc_adr"] = virt["anc_adr"]
    out["k_vdof_ancmask"] = virt["ancmask"]; out["k_vdof_descmask"] = virt["descmask"]
    # source address in the true M for every H entry (-1 where M is structurally zero)
    src = -np.ones(virt["nnz"], np.int32)
    lookup = {(int(i), int(j)): p for p, (i, j) in enumerate(zip(true["ei"], true["ej"]))}
    for p in range(virt["nnz"]):
        src[p] = lookup.get((int(virt["ei"][p]), int(virt["ej"][p])), -1)
    out["k_H_src"] = src

    # ---- triangular enumeration for the factorisation: t -> (m, q), 1 <= m <= q
    tm, tq = [], []
    for q in range(1, MAXV):
        for m_ in range(1, q + 1):
            tm.append(m_); tq.append(q)
    out["k_tri_m"] = I(tm[:MAXNZ]); out["k_tri_q"] = I(tq[:MAXNZ])

    # ---- sites / sensors the env reads (reference joystick.py:158-181, base.py:234-273)
    names_site = list(a["names_site"])
    names_sensor = list(a["names_sensor"])
    out["k_site_imu"] = I([names_site.index("imu")])
    out["k_site_feet"] = I([names_site.index("left_foot"), names_site.index("right_foot")])
    sadr = lambda n: int(a["sensor_adr"][names_sensor.index(n)])
    out["k_adr"] = I([sadr("gyro"), sadr("local_linvel"), sadr("accelerometer"), sadr("upvector"), sadr("global_angvel"),
                      sadr("left_foot_global_linvel"), sadr("right_foot_global_linvel")])
    return out
