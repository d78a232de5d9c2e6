// odk_model.h -- device-resident model: constants + static topology tables (tables.py).
// Replaces mjx.Model (reference base.py:61) for the kernels in odk_engine.hip.
#pragma once
#include <stdint.h>

namespace odk {

constexpr int EQ_MAX = 2;     // <equality><joint> rows per model in the kernels
constexpr int EQP_MAX = 2, EQP_ROWS = 9;   // path constraints (connect / weld) and their rows (3 / 6 each)
constexpr int MAXV = 32;     // dofs
constexpr int MAXQ = 32;     // qpos
constexpr int MAXB = 20;     // bodies
constexpr int MAXJ = 26;     // joints
constexpr int MAXU = 16;     // actuators
constexpr int MAXNZ = 512;   // sparse matrix entries
constexpr int MAXHV = 20;    // hull vertices per foot
constexpr int MAXHF = 40;    // hull faces per foot
constexpr int MAXHE = 64;    // unique hull edges per foot
// What the kernels' LDS regions hold (odk_kernels.h: [2][HULL_MAXV][3] vertices | [2][HULL_MAXF][3] normals | [HULL_MAXF] plane
// offsets, the plane pass' 16 lanes + ONE extra vertex, FaceRegs<2> = two faces per lane of a 16-lane row, 3 edges per lane):
// odk_model_load refuses a hull beyond these (ODK_ERR_UNSUPPORTED) -- the blob arrays above are merely the wire format's bound.
constexpr int HULL_MAXV = 17;    // vertices
constexpr int HULL_MAXF = 30;    // faces after the coplanar merge
constexpr int HULL_MAXE = 48;    // unique edges
static_assert(HULL_MAXV <= MAXHV && HULL_MAXF <= MAXHF && HULL_MAXE <= MAXHE && HULL_MAXF <= 32 && HULL_MAXE <= 48 && HULL_MAXV <= 17, "kernel hull regions");
constexpr int MAXCHAIN = 8;
constexpr int MAXSITE = 8;
constexpr int MAXSENS = 16;
constexpr int NCON = 12;     // 3 geom pairs x 4 contacts
constexpr int NSENSD = 46;

// Per-body constants of the kinematics / inertia sweeps, one record per body (+ one "no body" record at index MAXB for the lanes
// past the last body), filled on the host (odk_engine.hip fill_body_st): the kernel fetches a lane's record with ONE address
// computation and no dependent loads (as separate tables the joint-derived fields were three dependent loads deep).
struct BodySt {
  int level, parent, nchild, child[3], njnt, jd[2], jj[2], jr[2], pathmask, is_path, upmask, path_head;   // jr: CDOF column of the joint's dof (-1: twin, its main dof's column is the same vector)
  float pos[3], quat[4], ipos[3], inertia[6], ax[2][3];
  int pad;   // 40 dwords
};

// Per-lane statics of the step kernels (odk_kernels.h Statics = this + compile-time counts), host-built per lane like the body
// records: compute_statics runs on the host once per model, the kernels copy their lane's record.
struct LaneSt {
  int j_qadr, j_dadr;                    // joint role (sin/cos phase, Euler)
  // reduced-dof role (lane = reduced dof, DevModel::paired; a model without twins: reduced dof = dof)
  int cs_pk;                             // chain_solve roles: lane 8 c + b -> chain c's length | head depth << 3 | first dof << 8 | row address << 14; lane q -> base entry (tb << 24 | tb2 << 27)
  int ch_first, ch_len;                  // the serial chain this lane's reduced dof belongs to (ch_len 0: base dof / no chain)
  int r_on, r_depth, r_Madr, r_ancmask, r_descmask, r_foot;   // reduced tree layout (virtual-tree statics: fetched in the rare path)
  int m_adr[16];                         // M v product: byte offset (from the env's LDS image) of M's entry (lane, j), two per register; unrelated dofs -> a structural zero
  // dof role (lane = dof)
  int d_on, d_body;
  int d_act, d_flrow, d_limrow, d_foot;  // d_foot: bit0 moves left foot, bit1 right foot
  int d_qadr, d_lim_on;                  // hinge qpos address (-1: free joint); has a limit row
  int d_tkind, d_red;                    // twin dofs (DevModel::paired): 0 unpaired / 1 main (twin = dof + 1) / 2 twin; reduced dof
  float d_damping, d_lo, d_hi;           // joint range of the dof's hinge
};

struct DevModel {
  BodySt body_st[MAXB + 1];
  LaneSt lane_st[64];
  int nq, nv, nu, nb, nj, nM, nH, nfl, nlim, nrow, nsite, nsensor;
  float dt, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int ls_iterations, iterations;
  // bodies
  int base_body, body_in_tree[MAXB], body_parent[MAXB], body_jntadr[MAXB], body_jntnum[MAXB];
  int max_level, body_level[MAXB], body_children[MAXB][3], body_nchild[MAXB];
  int max_nonpath_level, body_pathmask[MAXB], body_is_path[MAXB], body_upmask[MAXB], body_path_head[MAXB];   // serial body chains (tables.py)
  // bodies above the serial chains that have children: the subtree sum of np_body[i] = sum over np_src[i][0 .. np_nsrc) of
  // own values (itself, non-chain bodies below it) and chain-head sums (chain heads below it)
  int np_count, np_body[4], np_nsrc[4], np_src[4][MAXB];
  int body_chain[MAXB][MAXCHAIN], body_chain_len[MAXB];
  int body_ancdof[MAXB][MAXV], body_nancdof[MAXB];
  int body_sub[MAXB][MAXB], body_nsub[MAXB];
  float body_pos[MAXB][3], body_quat[MAXB][4], body_ipos[MAXB][3], body_mass[MAXB], body_inertia[MAXB][6];
  // joints
  int jnt_qposadr[MAXJ], jnt_dofadr[MAXJ], jnt_bodyid[MAXJ];
  float jnt_axis[MAXJ][3], jnt_pos[MAXJ][3], jnt_range[MAXJ][2], qpos0[MAXQ];
  // dofs
  int dof_body[MAXV], dof_depth[MAXV], dof_anc[MAXV][MAXV], dof_Madr[MAXV], dof_anc_adr[MAXV][MAXV];
  int dof_ndesc[MAXV], dof_desc[MAXV][MAXV], dof_desc_adr[MAXV][MAXV];
  int dof_nprefix[MAXV], dof_prefix[MAXV][MAXV];
  int dof_nsym[MAXV], dof_sym_dof[MAXV][MAXV], dof_sym_adr[MAXV][MAXV];
  int dof_act[MAXV], dof_flrow[MAXV], dof_limrow[MAXV];
  int dof_ancmask[MAXV], dof_descmask[MAXV], vdof_ancmask[MAXV], vdof_descmask[MAXV];
  int dof_qadr[MAXV], dof_jnt[MAXV];   // hinge dofs: qpos address / joint id (-1 for the free joint)
  float dof_range[MAXV][2];
  float dof_armature[MAXV], dof_damping[MAXV], dof_frictionloss[MAXV], dof_invweight0[MAXV];
  int M_i[MAXNZ], M_j[MAXNZ];
  int M_ent[MAXNZ];   // packed entry: i | j << 5 | feet moved by dof i << 10 | feet moved by dof j << 12
  int nchain, chain_first[3], chain_len[3];   // tree of chains below the floating base (0 chains: generic tree)
  // Twin dofs (backlash joints): a hinge v declared right after hinge u on the same body, same anchor, same axis, has the
  // same motion column, cdof_v == cdof_u, so M = P Mr P^T + diag(armature) with P copying each reduced column onto the
  // pair (and likewise the Newton Hessian: contact rows see the pair through the same column, friction-loss / limit rows are
  // diagonal).  All matrix work of the kernels runs on the REDUCED tree (twins merged into their main dof); a model without
  // twins reduces to itself.  Built at load (odk_engine.hip: build_reduced_tables); kernel side: odk_kernels.h.
  int paired, nvr, nMr, nHr;
  int dof_tkind[MAXV];       // 0: unpaired, 1: main dof of a pair (its twin is dof + 1), 2: twin
  int dof_red[MAXV];         // reduced dof of this dof (a twin shares its main dof's)
  int red_main[MAXV], red_twin[MAXV];                                  // per reduced dof: main dof, twin dof (-1: none)
  int red_depth[MAXV], red_Madr[MAXV], red_ancmask[MAXV], red_descmask[MAXV], red_foot[MAXV];   // reduced tree layout (as dof_*); feet moved (bit f)
  int rv_depth[MAXV], rv_Madr[MAXV], rv_ancmask[MAXV], rv_descmask[MAXV];                     // reduced VIRTUAL tree (as vdof_*)
  int nrchain, rchain_first[3], rchain_len[3];                         // serial chains of the reduced tree below the floating base
  int foot_rchain_first[2], foot_rchain_len[2];                        // the chain that carries foot f: the reduced dofs above the foot are 0..5 + this chain
  // packed entries of the reduced layouts: ri | rj << 5 | feet of ri << 10 | feet of rj << 12 | diagonal << 14 | pair << 15 |
  // main dof of ri << 16; virtual tree: | (address in the reduced M + 1) << 21 (0: structurally zero in M)
  int R_ent[MAXNZ], RH_ent[MAXNZ];
  // virtual tree (Hessian)
  int vdof_depth[MAXV], vdof_anc[MAXV][MAXV], vdof_Madr[MAXV], vdof_anc_adr[MAXV][MAXV];
  int vdof_ndesc[MAXV], vdof_desc[MAXV][MAXV], vdof_desc_adr[MAXV][MAXV];
  int H_i[MAXNZ], H_j[MAXNZ], H_src[MAXNZ];
  int tri_m[MAXNZ], tri_q[MAXNZ];
  // actuators
  int act_qposadr[MAXU], act_dofadr[MAXU], act_backlash_qposadr[MAXU];
  float act_kp[MAXU], act_bias1[MAXU], act_bias2[MAXU], act_ctrlrange[MAXU][2], act_forcerange[MAXU][2];
  int act_ctrllimited[MAXU], act_forcelimited[MAXU];
  float key_qpos[MAXQ], key_ctrl[MAXU];
  // constraint rows
  int fl_dof[MAXV], lim_jnt[MAXJ];
  float fl_D[MAXV], fl_R[MAXV], fl_b[MAXV];             // friction-loss rows: pos = 0 -> constant impedance
  float lim_solref[MAXJ][2], lim_solimp[MAXJ][5], lim_invweight[MAXJ];
  float pair_solref[3][2], pair_solimp[3][5], pair_mu[3], pair_invweight[3];  // pairs: Lfoot-floor, Rfoot-floor, Lfoot-Rfoot
  // solref / solimp folded into per-row constants at load (odk_engine.hip: pack_imp): k, b, dmin, dmax, 1/width, mid,
  // power, 1/mid^(power-1), 1/(1-mid)^(power-1)
  float lim_imp[MAXJ][9], pair_imp[3][9];
  // feet / floor
  int foot_body[2], foot_nvert[2], foot_nface[2], foot_dofmask[2][MAXV];
  float foot_vert[2][MAXHV][3];  // hull vertices in the BODY frame (geom pos/quat folded in)
  int foot_face[2][MAXHF][3];
  float foot_obb_center[2][3], foot_obb_half[2][3], foot_obb_axes[2][9];  // body-frame OBB (columns = axes)
  float foot_sphere_r[2];        // |foot_obb_half|: the bounding sphere of the foot-foot cull
  // convex-convex narrow phase (odk_convex.h): face polygons after the coplanar merge (count, then <= 4 vertices counter-clockwise
  // seen from outside), their outward normals in the BODY frame, unique edges (va, vb, face running va -> vb, face running
  // vb -> va), an interior point; built at load (odk_engine.hip build_convex_tables).
  int foot_npoly[2], foot_nedge[2];
  int foot_poly[2][MAXHF][5];
  float foot_fnorm[2][MAXHF][3];
  float foot_foff[2][MAXHF];       // plane offsets n . v of the hull's faces in the body frame (height-field cull: hull face query per prism)
  int foot_edge[2][MAXHE][4];
  int foot_lane_rec[2][16][8];   // per 16-lane row lane j: the hull edges j, j + 16, j + 32 (face a | face b << 8 | va << 16 | vb << 24) and faces j, j + 16
                                 // (count | v0 << 3 | v1 << 8 | v2 << 13 | v3 << 18); bit 31: the hull has no such edge / face (the record is edge / face 0's)
  float foot_centroid[2][3];
  // primitive colliders in place of the foot hulls (mjtGeom: 2 sphere, 3 capsule; 7 = convex hull, the duck's own): centre and, for a
  // capsule, the axis (the geom frame's z) in the BODY frame; size = radius, half length.  foot_prim = some foot is a primitive.
  int foot_prim, foot_gtype[2];
  float foot_gpos[2][3], foot_gaxis[2][3], foot_gsize[2][3];
  float plane_pos[3], plane_n[3], plane_frame[9];
  int floor_is_plane;
  // height-field floor (rough terrain): geom frame = (plane_pos, floor_mat); samples live in HBM (KArgs.hfield)
  int hfield_nrow, hfield_ncol;
  float hfield_size[4], floor_mat[9];
  int hfield_filter;   // per batch (odk_env_config.hfield_up_normals_only): 0 = none, 3 = a pair's contacts count only when its normal points up
  // sites / sensors
  int site_body[MAXSITE], site_imu, site_feet[2];
  float site_pos[MAXSITE][3], site_mat[MAXSITE][9], site_quat[MAXSITE][4];
  int sensor_type[MAXSENS], sensor_site[MAXSENS], sensor_adr[MAXSENS];
  int adr_gyro, adr_local_linvel, adr_accelerometer, adr_upvector, adr_global_angvel, adr_foot_linvel[2];
  // <equality><joint> rows the kernels model (odk_kernels.h "equality rows"; shapes with S::EQ): q1 - q1_0 = poly(q2 - q2_0) between two
  // hinges of ONE serial chain -- its Hessian term -D c then falls on an entry the chain's block already has -- or one hinge held at
  // poly[0].  At most EQ_MAX rows, a dof in at most one.  eq_key: low ten bits (i | j << 5) of the packed reduced entry the coupling's
  // off-diagonal term lands on (-1: single-joint row); dof_eqrow: the row a dof takes part in (-1: none).
  // <equality><connect | weld> whose two bodies lie on ONE root-to-leaf path of the tree (or body2 = the world): "path rows" (shapes with S::EQ;
  // odk_kernels.h).  A row's Jacobian entry for dof i is m1_i (w1 . cdof_i) - m2_i (w2 . cdof_i) with a wrench per body and m = "dof i is above
  // the body"; both supports lie on the path, so J^T D J only touches entries the tree layout has.  dof_eqp: bit 2c = above body1 of
  // constraint c, bit 2c + 1 = above body2.
  int eqp_cross;     // some path constraint ties the two foot chains together (a closed loop): its rows' J^T D J needs the VIRTUAL tree's entries (second leg below the first foot)
  int neqp, eqp_nrow, eqp_type[EQP_MAX], eqp_b1[EQP_MAX], eqp_b2[EQP_MAX], eqp_row0[EQP_MAX], dof_eqp[MAXV];
  float eqp_a1[EQP_MAX][3], eqp_a2[EQP_MAX][3], eqp_relq[EQP_MAX][4], eqp_ts[EQP_MAX], eqp_imp[EQP_MAX][9], eqp_invw[EQP_MAX][2];
  int cone;      // <option cone>: 0 pyramidal, 1 elliptic (shapes with S::ELL; odk_kernels.h "elliptic cones")
  int neq, eq_dof1[EQ_MAX], eq_dof2[EQ_MAX], eq_qadr1[EQ_MAX], eq_qadr2[EQ_MAX], eq_key[EQ_MAX], dof_eqrow[MAXV];
  float eq_poly[EQ_MAX][5], eq_imp[EQ_MAX][9], eq_invweight[EQ_MAX];
};

// Topology of a height-field prism (vertices 0..2 = top triangle counter-clockwise seen from above, 3..5 below them; faces: top,
// bottom, the sides over the edges 0-1, 1-2, 2-0): unique edges (va, vb, face running va -> vb, face running vb -> va) and face
// polygons (count, vertices).  Compile-time for the kernels (odk_convex.h unrolls over them); odk_model_load checks at run time that
// build_convex_tables -- the routine that prepares the foot hulls -- makes exactly these tables of a prism's eight triangles.
constexpr int PRISM_EDGE[9][4] = {{0, 1, 0, 2}, {1, 2, 0, 3}, {3, 5, 1, 4}, {0, 3, 2, 4}, {3, 4, 2, 1}, {1, 4, 3, 2}, {4, 5, 3, 1}, {2, 5, 4, 3}, {0, 2, 4, 0}};
constexpr int PRISM_POLY[5][5] = {{3, 0, 1, 2, 0}, {3, 3, 5, 4, 3}, {4, 0, 3, 4, 1}, {4, 1, 4, 5, 2}, {4, 2, 5, 3, 0}};

// Height-field pair loop: the assignment of the wave's four rows to feet as a function of the four rows' open-entry counts, each capped at four (index = c0 + 5 (c1 + 5 (c2 + 5 c3)):
// odk_kernels.h hf_assign_index).  A row works its own foot while that has open entries; an idle row goes where most are left (lowest foot among equals).  Entry: foot of row r
// (2 bits each) | rank among the rows on that foot << 8 (2 bits each) | row has work << 16 | (most rows on one foot - 1) << 20.  Built at COMPILE time and kept in constant memory:
// the kernel's lookup is one scalar load (as a field of DevModel it was a vector load from global memory per iteration: round 6).
struct HfAssign { int v[625]; };
constexpr HfAssign make_hf_assign() {
  HfAssign T{};
  for (int idx = 0; idx < 625; idx++) {
    const int c[4] = {idx % 5, (idx / 5) % 5, (idx / 25) % 5, idx / 125};
    int asg[4] = {0, 0, 0, 0}, tgt[4] = {0, 1, 2, 3}, rnk[4] = {0, 0, 0, 0}, on[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; r++) if (c[r] > 0) { asg[r] = 1; on[r] = 1; }
    for (int r = 0; r < 4; r++) {
      if (on[r]) continue;
      int bt = -1, bl = 0;
      for (int t = 0; t < 4; t++) if (c[t] - asg[t] > bl) { bl = c[t] - asg[t]; bt = t; }
      if (bt >= 0) { tgt[r] = bt; rnk[r] = asg[bt]; asg[bt]++; on[r] = 1; }
    }
    int mq = 1;
    for (int t = 0; t < 4; t++) mq = asg[t] > mq ? asg[t] : mq;
    unsigned w = 0;
    for (int r = 0; r < 4; r++) w |= (unsigned)tgt[r] << (2 * r) | (unsigned)rnk[r] << (8 + 2 * r) | (unsigned)on[r] << (16 + r);
    T.v[idx] = (int)(w | (unsigned)(mq - 1) << 20);
  }
  return T;
}

// reference-motion table header (poly_reference_motion.py)
struct DevPRM {
  int nx, ny, nth, nsteps;
  float dxs[16], dys[16], dths[16], ranges[6];
};

}  // namespace odk
