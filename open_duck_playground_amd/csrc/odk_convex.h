// odk_convex.h -- convex-convex narrow phase on 16-lane rows (gfx950): separating-axis test over face normals and the edge pairs
// that form a face of the Minkowski difference (Gauss-map test), clipped 4-point face manifolds, edge-edge contacts, and the
// height-field floor as prisms.  What the reference runs here is mujoco-mjx collision_convex.py (`convex_convex`,
// `_create_contact_manifold`, `hfield_convex`; third party, reached through mjx.step at playground/open_duck_mini_v2/joystick.py:420)
// for the foot-foot mesh pair (open_duck_mini_v2.xml:203-205,408-410) and the terrain of scene_rough_terrain_backlash.xml:22.
//
// Mapping: one polytope pair is worked on by ONE 16-lane DPP row (lane j of the row): face queries put a face per lane, the edge
// query keeps the second polytope's edges in registers (3 per lane) and walks the first one's, the clipped manifold puts one
// candidate point per lane, so every arg-max / arg-min is a row-local DPP reduction (no LDS, no readlane).  The two feet of an env
// are two rows, i.e. they share every instruction.  Geometry sits in LDS that is dead between the inertia and the constraint
// phases; topology tables (face polygons, unique edges with their two faces) are built at model load (odk_engine.hip).
#pragma once

namespace odk {

// one polytope of a pair: geometry in LDS (work frame), topology in global memory
struct Cvx {
  const float* V;     // [nv][3]
  const float* N;     // [nf][3] outward unit normals
  const int* poly;    // [nf][5]: vertex count (3 or 4), then the vertices counter-clockwise seen from outside
  const int* edge;    // [ne][4]: va, vb, face that runs va -> vb, face that runs vb -> va
  int nv, nf, ne;
  float c[3];         // an interior point
};

#define ODK_ROWBASE ((int)(threadIdx.x & 48u))
__device__ __forceinline__ float row_get(float v, int src) { return __shfl(v, ODK_ROWBASE | src, 64); }
// lowest index among the row's maxima of (v, i); vmax = the maximum
__device__ __forceinline__ int row_argmax(float v, int i, float& vmax) {
  const unsigned k = fkey(v), mx = rreduce_u<true>(k);
  vmax = fkey_inv(mx);
  return (int)rreduce_u<false>(k == mx ? (unsigned)i : 0x7FFFFFFFu);
}
// lowest index among the row's minima of (v, i); vmin = the minimum
__device__ __forceinline__ int row_argmin(float v, int i, float& vmin) {
  const unsigned k = fkey(v), mn = rreduce_u<false>(k);
  vmin = fkey_inv(mn);
  return (int)rreduce_u<false>(k == mn ? (unsigned)i : 0x7FFFFFFFu);
}
__device__ __forceinline__ void sub3(float* r, const float* a, const float* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
__device__ __forceinline__ void ld3(float* r, const float* p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; }

// largest over the faces of P of the smallest signed distance of Q's vertices to the face plane (first maximum in face order)
__device__ __forceinline__ void face_query_row(const Cvx& P, const Cvx& Q, int j, float& sep, int& face) {
  float best = -3.0e38f;
  int bi = 0x7FFFFFFF;
  for (int f = j; f < P.nf; f += 16) {
    float n[3], v0[3];
    ld3(n, P.N + 3 * f); ld3(v0, P.V + 3 * P.poly[5 * f + 1]);
    float smin = 3.0e38f;
    for (int q = 0; q < Q.nv; q++) {
      float t[3];
      sub3(t, Q.V + 3 * q, v0);
      smin = fminf(smin, dot3(t, n));
    }
    if (smin > best) { best = smin; bi = f; }
  }
  face = row_argmax(best, bi, sep);
}

// per edge of A: b x a of its two face normals and its direction: 6 floats, written by the row
__device__ __forceinline__ void edge_prepare_row(const Cvx& A, float* AE, int j, bool act) {
  for (int i = j; i < A.ne; i += 16) {
    const int* e = A.edge + 4 * i;
    float a[3], b[3], bxa[3], d[3];
    ld3(a, A.N + 3 * e[2]); ld3(b, A.N + 3 * e[3]);
    cross3(bxa, b, a);
    sub3(d, A.V + 3 * e[1], A.V + 3 * e[0]);
    if (act) {
      float* o = AE + 6 * i;
      o[0] = bxa[0]; o[1] = bxa[1]; o[2] = bxa[2]; o[3] = d[0]; o[4] = d[1]; o[5] = d[2];
    }
  }
}

// the edges of B this lane owns (edge j + 16 s), negated face normals and their cross product ready for the Gauss-map test
template <int NSLOT> struct EdgeRegs {
  float c[NSLOT][3], d[NSLOT][3], dxc[NSLOT][3];   // (direction and a point of the edge are fetched only for the few pairs that pass the test)
  int vv[NSLOT];                                   // va | vb << 8
  bool on[NSLOT];
};
template <int NSLOT> __device__ __forceinline__ void edge_regs_load(EdgeRegs<NSLOT>& R, const Cvx& B, int j) {
#pragma unroll
  for (int s = 0; s < NSLOT; s++) {
    const int jb = j + 16 * s;
    R.on[s] = jb < B.ne;
    const int* e = B.edge + 4 * (R.on[s] ? jb : 0);
    for (int k = 0; k < 3; k++) { R.c[s][k] = -B.N[3 * e[2] + k]; R.d[s][k] = -B.N[3 * e[3] + k]; }
    R.vv[s] = e[0] | (e[1] << 8);
    cross3(R.dxc[s], R.d[s], R.c[s]);
  }
}

// the same from the row lane's packed record (DevModel::foot_lane_rec: one 32-byte load instead of five dependent table rows)
template <int NSLOT> __device__ __forceinline__ void edge_regs_from_rec(EdgeRegs<NSLOT>& R, const Cvx& B, const int* rec) {
#pragma unroll
  for (int s = 0; s < NSLOT; s++) {
    const unsigned e = (unsigned)rec[s];
    R.on[s] = !(e >> 31);
    const int fa = e & 255u, fb = (e >> 8) & 255u;
    for (int k = 0; k < 3; k++) { R.c[s][k] = -B.N[3 * fa + k]; R.d[s][k] = -B.N[3 * fb + k]; }
    R.vv[s] = (int)((e >> 16) & 0x7FFFu);   // va | vb << 8
    cross3(R.dxc[s], R.d[s], R.c[s]);
  }
}
// Largest separation over the edge pairs (edge i of A, edge jb of B) whose Gauss-map arcs cross (they span a face of the Minkowski
// difference); ties to the lowest (i, jb).  pair = i << 8 | jb (0x7FFFFFFF: none), axis oriented away from A's interior.
// Edges closer to parallel than 1e-4 (sine) give no axis.
template <int NSLOT>
__device__ __forceinline__ void edge_query_row(const Cvx& A, const Cvx& B, const float* AE, const EdgeRegs<NSLOT>& R, int j, float& sep, int& pair, float* axis) {
  float best = -3.0e38f, bax[3] = {0.0f, 0.0f, 1.0f};
  int bi = 0x7FFFFFFF;
  for (int i = 0; i < A.ne; i++) {
    const float* ae = AE + 6 * i;
    const float bxa[3] = {ae[0], ae[1], ae[2]}, ea[3] = {ae[3], ae[4], ae[5]};
    const int* et = A.edge + 4 * i;   // uniform address (i and the table are the same in every lane): scalar loads
    float a[3], b[3], pa[3], ta[3];
    ld3(a, A.N + 3 * et[2]); ld3(b, A.N + 3 * et[3]); ld3(pa, A.V + 3 * et[0]);
    sub3(ta, pa, A.c);
    const float ea2 = dot3(ea, ea);
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
      const float cba = dot3(R.c[s], bxa), dba = dot3(R.d[s], bxa), adc = dot3(a, R.dxc[s]), bdc = dot3(b, R.dxc[s]);
      if (R.on[s] && cba * dba < 0.0f && adc * bdc < 0.0f && cba * bdc > 0.0f) {
        float ax[3], t[3], pb[3], eb[3];
        const int* be = B.edge + 4 * (j + 16 * s);
        ld3(pb, B.V + 3 * be[0]);
        sub3(eb, B.V + 3 * be[1], pb);
        cross3(ax, ea, eb);
        const float len = sqrtf(dot3(ax, ax));
        if (len >= 1e-4f * sqrtf(ea2 * dot3(eb, eb)) + 1e-30f) {
          const float inv = 1.0f / len;
          ax[0] *= inv; ax[1] *= inv; ax[2] *= inv;
          if (dot3(ax, ta) < 0.0f) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; }
          sub3(t, pb, pa);
          const float sp = dot3(ax, t);
          if (sp > best) { best = sp; bi = (i << 8) | (j + 16 * s); bax[0] = ax[0]; bax[1] = ax[1]; bax[2] = ax[2]; }
        }
      }
    }
  }
  pair = row_argmax(best, bi, sep);
  const int src = pair & 15;   // jb = j + 16 s: the owner's row lane
  for (int k = 0; k < 3; k++) axis[k] = row_get(bax[k], src);
}

// mjx _manifold_points over the row's candidates (lane j = candidate j, `cand`: the lane holds one): four points of roughly
// maximal area among the masked ones; every arg-max takes the lowest index among ties, like jp.argmax
__device__ __forceinline__ void manifold4_row(const float* p, bool cand, bool mask, const float* n, int np, int j, int* idx) {
  const float dm = cand ? (mask ? 0.0f : -1e6f) : -3.0e38f;
  float vm;
  idx[0] = row_argmax(dm, j, vm);
  float a[3], b[3], c[3];
  for (int k = 0; k < 3; k++) a[k] = row_get(p[k], idx[0]);
  float ap[3];
  sub3(ap, a, p);
  idx[1] = row_argmax(cand ? dot3(ap, ap) + dm : -3.0e38f, j, vm);
  for (int k = 0; k < 3; k++) b[k] = row_get(p[k], idx[1]);
  float amb[3], ab[3];
  sub3(amb, a, b);
  cross3(ab, n, amb);
  idx[2] = row_argmax(cand ? area0(fabsf(dot3(ap, ab))) + dm : -3.0e38f, j, vm);
  for (int k = 0; k < 3; k++) c[k] = row_get(p[k], idx[2]);
  float amc[3], bmc[3], ac[3], bc[3], bp[3];
  sub3(amc, a, c); sub3(bmc, b, c);
  cross3(ac, n, amc); cross3(bc, n, bmc);
  sub3(bp, b, p);
  // last point: argmax over concat([first half: area against bc, second half: against ac]), lowest index among the values within
  // AREA_TIE of the maximum (oracle manifold_points: for a triangle of candidates the two largest entries are equal in exact arithmetic)
  const float v1 = area0(fabsf(dot3(bp, bc))) + dm, v2 = area0(fabsf(dot3(ap, ac))) + dm;
  const float M = fkey_inv(rreduce_u<true>(fkey(cand ? fmaxf(v1, v2) : -3.0e38f))) - AREA_TIE;
  const unsigned id = !cand ? 0x7FFFFFFFu : (v1 >= M ? (unsigned)j : (v2 >= M ? (unsigned)(np + j) : 0x7FFFFFFFu));
  const unsigned best = rreduce_u<false>(id);
  idx[3] = best == 0x7FFFFFFFu ? 0 : (int)(best >= (unsigned)np ? best - (unsigned)np : best);
}

// mjx _clip_edge_to_planes: the edge (p0, p1) against the side planes of polygon Q (nq vertices at QP, normal qn); the `which`-th
// of the two clipped points, returns the mask
template <bool FULL4 = false>      // FULL4: all four planes for every lane (the height-field loop, below)
__device__ __forceinline__ bool clip_edge_row(const float* p0, const float* p1, const float* QP, int nq, const float* qn, int which, float* out) {
  // Every point the routine can return lies on the edge: p0 + t (p1 - p0).  What is kept per plane is the parameter t, not the
  // point: "most along the edge" is t |d|^2 (from p0) and (1 - t) |d|^2 (from p1) without forming the candidate, the second end's
  // signed distance to a plane is the first end's plus pn . d, and the two clipped points have crossed when t0 > t1.
  float d01[3];
  sub3(d01, p1, p0);
  const float dd = dot3(d01, d01);
  float best0 = -3.0e38f, best1 = -3.0e38f, tb0 = 0.0f, tb1 = 1.0f;
  bool both = false;
  // (the polygon is read once, up front: in a rolled loop every plane waited for its own LDS round trip)
  float Q[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++) ld3(Q[k], QP + 3 * (k < nq ? k : 0));
#pragma unroll
  for (int k = 0; k < 4; k++) {
    // (nq is 3 or 4.  FULL4: no early exit for a triangle -- its fourth plane is its first again (Q[3] is a copy of Q[0], so the edge is Q[2] -> Q[0]) and a
    // plane met twice changes nothing below (strict comparisons); the rows of a wave mix triangles and quads, so the exit only cost its branch: round 6.)
    if (!FULL4 && k >= nq) break;
    const float* pa = k == 0 ? (nq == 4 ? Q[3] : Q[2]) : Q[k - 1]; const float* pb = Q[k];
    float e[3], pn[3], t0[3];
    sub3(e, pb, pa);
    cross3(pn, e, qn);
    sub3(t0, p0, pa);
    const float a0 = dot3(t0, pn), denom = dot3(pn, d01), a1 = a0 + denom;
    const bool f0 = a0 > 1e-6f, f1 = a1 > 1e-6f;
    both = both || (f0 && f1);
    // _closest_segment_point_plane (v_rcp_f32, 1 ulp: an IEEE quotient is ten instructions)
    float t = -a0 * __builtin_amdgcn_rcpf(denom + (denom == 0.0f ? 1e-6f : 0.0f));
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float s0 = f0 ? t * dd : 0.0f, s1 = f1 ? (1.0f - t) * dd : 0.0f;   // {clipped point where the end is in front, else the end}
    if (s0 > best0) { best0 = s0; tb0 = f0 ? t : 0.0f; }
    if (s1 > best1) { best1 = s1; tb1 = f1 ? t : 1.0f; }
  }
  const bool mask = !both && !(tb0 > tb1);
  const float tt = !both ? (which ? tb1 : tb0) : (which ? 1.0f : 0.0f);   // (crossed points are returned as they are, masked out)
  for (int k = 0; k < 3; k++) out[k] = tt == 1.0f ? p1[k] : p0[k] + tt * d01[k];
  return mask;
}

// Scratch of one row (floats): RP | IP = reference / incident polygon of the current face contact ([4][3] each), NEW = this pair's
// four contacts.  A contact = dist, pos[3], normal[3], (candidate index: the caller's) -- 8 floats.
struct RowScratch { float* RP; float* IP; float* NEW; float* PW; float* VV; };   // PW [16] | VV [48], contiguous: the row's list of passing edge pairs (height field only)

// mjx _create_contact_manifold on the polygons in S.RP (rcnt vertices, normal n_ref) / S.IP (icnt, n_inc): lane j = candidate j of
// _clip -- (incident edge e clipped by the reference side planes) x 2, then (reference edge, projected on the incident plane along
// the reference normal, clipped by the incident side planes) x 2 -- projected on the reference plane, four of them by
// _manifold_points, written to S.NEW with the normal sg * n_ref (skip: an edge contact replaces them).
template <bool PULL = false>      // PULL: lanes 0 .. 3 fetch the four chosen candidates from their lanes and write them in ONE block (the height-field loop)
__device__ __forceinline__ void manifold_row(const RowScratch& S, int rcnt, int icnt, const float* n_ref, const float* n_inc, float sg, bool skip, int j, bool act, float cidx0 = 0.0f) {
  const int e = j >> 1, which = j & 1, np = 2 * (icnt + rcnt);
  const bool cand = j < np;
  float pt[3] = {0.0f, 0.0f, 0.0f};
  bool mask = false;
  {
    // ONE clip per lane: the two kinds of candidate differ only in their operands (as two branches every lane walked both)
    const bool subj = e < icnt;
    const int er = subj ? e : e - icnt, cntp = subj ? icnt : rcnt;
    const float* src = subj ? S.IP : S.RP;
    float a[3], b[3];
    ld3(a, src + 3 * (er == 0 ? cntp - 1 : er - 1)); ld3(b, src + 3 * (er < cntp ? er : 0));
    if (PULL || !subj) {   // reference edge: projected on the incident plane along the reference normal (PULL: every lane walks through it with a zero step for the
                           // incident edges -- some lane of the wave always takes the branch, and as an if / else it cost its exec-mask bookkeeping on top)
      const float d = dot3(S.IP, n_inc), den = dot3(n_ref, n_inc), dinv = __builtin_amdgcn_rcpf(den + (den == 0.0f ? 1e-6f : 0.0f));
      const float ta = subj ? 0.0f : (d - dot3(a, n_inc)) * dinv, tb = subj ? 0.0f : (d - dot3(b, n_inc)) * dinv;
      for (int k = 0; k < 3; k++) { a[k] += ta * n_ref[k]; b[k] += tb * n_ref[k]; }
    }
    const float qn[3] = {subj ? n_ref[0] : n_inc[0], subj ? n_ref[1] : n_inc[1], subj ? n_ref[2] : n_inc[2]};
    const bool m1 = clip_edge_row<PULL>(a, b, subj ? S.RP : S.IP, subj ? rcnt : icnt, qn, which, pt);
    mask = cand && m1;
  }
  float t0[3], pref[3];
  sub3(t0, pt, S.RP);
  const float off = dot3(t0, n_ref);
  for (int k = 0; k < 3; k++) pref[k] = pt[k] - off * n_ref[k];
  mask = mask && (-off > 1e-6f);   // behind the reference plane
  int idx[4];
  manifold4_row(pref, cand, mask, n_ref, np, j, idx);
  ODK_SYNC();
  if constexpr (PULL) {
    // (four predicated blocks of seven LDS writes each -> four lane fetches and one block: round 6)
    const float pen = -off;
    int src = idx[3];      // (one select per statement: nested, they become branches)
    src = j == 2 ? idx[2] : src; src = j == 1 ? idx[1] : src; src = j == 0 ? idx[0] : src;
    const float w0 = row_get(mask ? -pen : 1.0f, src);
    float w[3];
#pragma unroll
    for (int t = 0; t < 3; t++) w[t] = row_get(pref[t] - 0.5f * pen * n_ref[t], src);
    if (act && !skip && j < 4) {
      float* o = S.NEW + 8 * j;
      o[0] = w0;
      for (int t = 0; t < 3; t++) { o[1 + t] = w[t]; o[4 + t] = sg * n_ref[t]; }
      o[7] = cidx0 + (float)j;      // candidate index in MJX's list: prism-major, then the pair's four slots
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (act && !skip && j == idx[k]) {
        const float pen = -off;
        float* o = S.NEW + 8 * k;
        o[0] = mask ? -pen : 1.0f;
        for (int t = 0; t < 3; t++) { o[1 + t] = pref[t] - 0.5f * pen * n_ref[t]; o[4 + t] = sg * n_ref[t]; }
      }
    }
  }
}

// ONE contact at the closest points of the edges (p1, q1) / (p2, q2) (Ericson 5.1.9), depth sep, normal ax; slots 1..3 inactive
template <bool REGS = false>      // REGS (the height-field loop): the end points in registers before the writes, the clamping cases as selects
__device__ __forceinline__ void edge_contact_row(const float* p1_, const float* q1, const float* p2_, const float* q2, float sep, const float* ax, const RowScratch& S, int j, bool act, float cidx0 = 0.0f) {
  float d1[3], d2[3], r[3];
  // (REGS: read once -- through the pointers every use after a write to S.NEW is a new LDS round trip, the writes may alias them for all the compiler knows)
  float p1r[3], p2r[3];
  if constexpr (REGS) { ld3(p1r, p1_); ld3(p2r, p2_); }
  const float* p1 = REGS ? p1r : p1_; const float* p2 = REGS ? p2r : p2_;
  sub3(d1, q1, p1); sub3(d2, q2, p2); sub3(r, p1, p2);
  const float a = dot3(d1, d1), ee = dot3(d2, d2), f = dot3(d2, r), c = dot3(d1, r), b = dot3(d1, d2), den = a * ee - b * b;
  const float ainv = __builtin_amdgcn_rcpf(a > 1e-30f ? a : 1.0f), einv = __builtin_amdgcn_rcpf(ee > 1e-30f ? ee : 1.0f);
  float s = den > 1e-30f ? (b * f - c * ee) * __builtin_amdgcn_rcpf(den) : 0.0f;
  s = fminf(fmaxf(s, 0.0f), 1.0f);
  float t = (b * s + f) * einv;
  if constexpr (REGS) {
    const bool lo = t < 0.0f, hi = t > 1.0f;
    const float s_lo = fminf(fmaxf(-c * ainv, 0.0f), 1.0f), s_hi = fminf(fmaxf((b - c) * ainv, 0.0f), 1.0f);
    s = hi ? s_hi : s; s = lo ? s_lo : s;
    t = hi ? 1.0f : t; t = lo ? 0.0f : t;
  } else {
    if (t < 0.0f) { t = 0.0f; s = fminf(fmaxf(-c * ainv, 0.0f), 1.0f); }
    else if (t > 1.0f) { t = 1.0f; s = fminf(fmaxf((b - c) * ainv, 0.0f), 1.0f); }
  }
  if (act && j < 4) {
    float* o = S.NEW + 8 * j;
    o[0] = j == 0 ? sep : 1.0f;
    for (int k = 0; k < 3; k++) { o[1 + k] = 0.5f * ((p1[k] + s * d1[k]) + (p2[k] + t * d2[k])); o[4 + k] = ax[k]; }
    if constexpr (REGS) o[7] = cidx0 + (float)j;
  }
}

// One polytope pair on one row, both given by tables: SAT decision + contacts into S.NEW[4][7] (normal pointing from A to B).
// All lanes of the row call this together; `act` gates the LDS writes (rows that have nothing to do run along).
template <int NSLOT>
__device__ __forceinline__ void sat_pair_row(const Cvx& A, const Cvx& B, const float* AE, const EdgeRegs<NSLOT>& RB, const RowScratch& S, int j, bool act) {
  float sep_a, sep_b, sep_e, eax[3];
  int face_a, face_b, pair;
  face_query_row(A, B, j, sep_a, face_a);
  face_query_row(B, A, j, sep_b, face_b);
  edge_query_row<NSLOT>(A, B, AE, RB, j, sep_e, pair, eax);
  const bool ref_a = sep_a >= sep_b;
  const float face_sep = ref_a ? sep_a : sep_b;
  const bool is_edge = pair != 0x7FFFFFFF && sep_e > face_sep + 1e-5f;
  // ---- face contact (computed by every row; an edge contact replaces it below)
  const Cvx& R = ref_a ? A : B; const Cvx& I = ref_a ? B : A;
  int rf = ref_a ? face_a : face_b;
  rf = (unsigned)rf < (unsigned)R.nf ? rf : 0;   // (rows that only run along may hold garbage: keep every table index in range)
  float n_ref[3];
  ld3(n_ref, R.N + 3 * rf);
  int inc;
  {
    float best = -3.0e38f, vm; int bi = 0x7FFFFFFF;
    for (int f = j; f < I.nf; f += 16) { const float s = -dot3(I.N + 3 * f, n_ref); if (s > best) { best = s; bi = f; } }
    inc = row_argmax(best, bi, vm);
    inc = (unsigned)inc < (unsigned)I.nf ? inc : 0;
  }
  float n_inc[3];
  ld3(n_inc, I.N + 3 * inc);
  const int rcnt = R.poly[5 * rf], icnt = I.poly[5 * inc];
  ODK_SYNC();
  if (act && j < 8) {
    const bool isr = j < 4; const int k = j & 3;
    const int cnt = isr ? rcnt : icnt;
    if (k < cnt) {
      const int v = isr ? R.poly[5 * rf + 1 + k] : I.poly[5 * inc + 1 + k];
      const float* src = (isr ? R.V : I.V) + 3 * v;
      float* dst = (isr ? S.RP : S.IP) + 3 * k;
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
  }
  ODK_SYNC();
  manifold_row(S, rcnt, icnt, n_ref, n_inc, ref_a ? 1.0f : -1.0f, is_edge, j, act);
  if (is_edge) {   // row-uniform
    const int ia = pair >> 8, ib = pair & 255;
    const int* ea = A.edge + 4 * (ia < A.ne ? ia : 0); const int* eb = B.edge + 4 * (ib < B.ne ? ib : 0);
    edge_contact_row(A.V + 3 * ea[0], A.V + 3 * ea[1], B.V + 3 * eb[0], B.V + 3 * eb[1], sep_e, eax, S, j, act);
  }
  ODK_SYNC();
}

// ---- height-field prisms: the first polytope's topology is known at compile time and its geometry fits registers
// vertices 0..2 = top triangle (counter-clockwise seen from above), 3..5 below them at z = -base; faces: top, bottom, the sides over
// the edges 0-1, 1-2, 2-0 (what build_convex_tables makes of the prism's eight triangles: odk_engine.hip)
struct Prism {
  float x[3], y[3], z[3], base;
  float nt[3];      // top normal
  float ns[3][2];   // side normals (x, y)
};
__device__ __forceinline__ void prism_vert(const Prism& P, int k, float* o) { const int t = k % 3; o[0] = P.x[t]; o[1] = P.y[t]; o[2] = k < 3 ? P.z[t] : -P.base; }
__device__ __forceinline__ void prism_norm(const Prism& P, int f, float* o) {
  if (f == 0) { o[0] = P.nt[0]; o[1] = P.nt[1]; o[2] = P.nt[2]; }
  else if (f == 1) { o[0] = 0.0f; o[1] = 0.0f; o[2] = -1.0f; }
  else { o[0] = P.ns[f - 2][0]; o[1] = P.ns[f - 2][1]; o[2] = 0.0f; }
}
// (PRISM_EDGE / PRISM_POLY: odk_model.h; only ever indexed with compile-time constants here)

// the faces of the second polytope this lane owns (face j + 16 s): normal, plane offset n . v0, polygon packed 3 | 5 x 4 bits
template <int NFS> struct FaceRegs { float d[NFS]; int poly[NFS]; bool on[NFS]; };   // (the normals are re-read from LDS per pair: registers are the scarce resource)
template <int NFS> __device__ __forceinline__ void face_regs_from_rec(FaceRegs<NFS>& R, const Cvx& B, const int* rec, int j) {
#pragma unroll
  for (int s = 0; s < NFS; s++) {
    const unsigned pk = (unsigned)rec[3 + s];
    R.on[s] = !(pk >> 31);
    R.d[s] = dot3(B.V + 3 * ((pk >> 3) & 31u), B.N + 3 * (R.on[s] ? j + 16 * s : 0));
    R.poly[s] = (int)(pk & 0x7FFFFFu);
  }
}

// prism P (first polytope, in registers; its own face query already done: sep_a, face_a) against the hull B (geometry in LDS,
// faces / edges of this lane in FB / RB); PV: the prism's vertices in LDS for the polygon fetch
template <int NFS, int NSLOT>
__device__ __forceinline__ void sat_prism_row(const Prism& P, const float* pc, const float* PV, const Cvx& B, const FaceRegs<NFS>& FB, const EdgeRegs<NSLOT>& RB,
                                              float sep_a, int face_a, const RowScratch& S, int j, bool act, float cidx0, int knock   // cidx0: candidate index of the pair's first contact; knock: 0 outside the timing experiment (ODK_HF_KNOCK)
#ifdef ODK_PROFILE
                                              , float* prof, long long& tp
#endif
                                              ) {
#ifdef ODK_PROFILE
#define SAT_PROF(i) do { const long long _t = clock64(); if (prof) prof[i] += (float)(_t - tp); tp = _t; } while (0)
#elif defined(ODK_MARK)
#define SAT_PROF(i) asm volatile("; SAT_MARK " #i ::: "memory")
#else
#define SAT_PROF(i) do { } while (0)
#endif
  // ---- face query of the hull against the prism's six vertices
  float sep_b; int face_b;
  float fn[NFS][3];   // this lane's hull face normals (slot s = face j + 16 s; lanes without a second face read face 0's)
#pragma unroll
  for (int s = 0; s < NFS; s++) ld3(fn[s], B.N + 3 * (FB.on[s] ? j + 16 * s : 0));
  HF_REP(11) {
    HF_TOUCH(fn[0][0]);
    // (vertices 3..5 sit under 0..2 at z = -base: the x / y part of n . v is shared by a column, as in the cull pass)
    float best = -3.0e38f; int bi = 0x7FFFFFFF;
#pragma unroll
    for (int s = 0; s < NFS; s++) {
      const float h0 = fn[s][0] * P.x[0] + fn[s][1] * P.y[0], h1 = fn[s][0] * P.x[1] + fn[s][1] * P.y[1], h2 = fn[s][0] * P.x[2] + fn[s][1] * P.y[2];
      const float top = fminf(fminf(h0 + fn[s][2] * P.z[0], h1 + fn[s][2] * P.z[1]), h2 + fn[s][2] * P.z[2]);
      const float bot = fminf(fminf(h0, h1), h2) - fn[s][2] * P.base;
      const float smin = fminf(top, bot) - FB.d[s];
      if (FB.on[s] && smin > best) { best = smin; bi = j + 16 * s; }
    }
    face_b = row_argmax(best, bi, sep_b);
  }
  SAT_PROF(1);
  // ---- edge query: the prism's nine edges (compile-time topology) against the hull edges of this lane.  The Gauss-map test of all
  // 27 pairs of a lane first (a pass bit each), then the few passing pairs in a short loop: taken inline, some lane of the wave
  // passes nearly every test, so every lane would walk through all 27 axis computations.
  float sep_e, eax[3]; int pair;   // pair: the winning list entry (prism edge << 22 | hull edge << 16 | its vertices va | vb << 8), 0x7FFFFFFF: none
  {
    static_assert(NSLOT == 3, "pass bit = 9 slot + i");
    unsigned pass = 0;
    if (!(knock & 8)) HF_REP(12) {
      // The prism's face normals are T = nt, (0, 0, -1), S0 = (0, a0, 0), S1 = (ux, uy, 0), S2 = (b2, 0, 0): written out, the nine
      // b x a and the projections lose their zero terms (the compiler may not drop x * 0), and the three vertical edges' b x a are
      // positive multiples of z -- only signs enter the test, so their c . (b x a) is c.z itself.
      float tx = P.nt[0]; HF_TOUCH(tx);
      const float ty = P.nt[1], tz = P.nt[2], a0 = P.ns[0][1], ux = P.ns[1][0], uy = P.ns[1][1], b2 = P.ns[2][0];
      const float e0x = a0 * tz, e0z = -a0 * tx;                                   // S0 x T
      const float e1x = uy * tz, e1y = -ux * tz, e1z = ux * ty - uy * tx;          // S1 x T
      const float e8y = tz * b2, e8z = -ty * b2;                                   // T x S2
      constexpr int FA[9] = {0, 0, 1, 2, 2, 3, 3, 4, 4}, FBK[9] = {2, 3, 4, 4, 1, 2, 1, 3, 0};   // PRISM_EDGE[i][2], [3]
      static_assert(PRISM_EDGE[0][2] == 0 && PRISM_EDGE[0][3] == 2 && PRISM_EDGE[4][3] == 1 && PRISM_EDGE[8][2] == 4 && PRISM_EDGE[8][3] == 0, "edge faces");
#pragma unroll
      for (int s = 0; s < NSLOT; s++) {
        const float* c = RB.c[s]; const float* d = RB.d[s]; const float* x = RB.dxc[s];
        const float pr[5] = {tx * x[0] + ty * x[1] + tz * x[2], -x[2], a0 * x[1], ux * x[0] + uy * x[1], b2 * x[0]};   // face normal . (d x c)
        const float cb[9] = {c[0] * e0x + c[2] * e0z, c[0] * e1x + c[1] * e1y + c[2] * e1z, c[1] * b2, c[2], c[0] * a0, c[2], c[0] * uy - c[1] * ux, c[2], c[1] * e8y + c[2] * e8z};
        const float db[9] = {d[0] * e0x + d[2] * e0z, d[0] * e1x + d[1] * e1y + d[2] * e1z, d[1] * b2, d[2], d[0] * a0, d[2], d[0] * uy - d[1] * ux, d[2], d[1] * e8y + d[2] * e8z};
#ifndef ODK_GAUSS_PRODUCTS
        // The three conditions are conditions on SIGNS: taken from the operands' sign bits -- two 3-input bit operations and one funnel shift per pair instead of three
        // products, a max3, a compare, a select and an or (round 6: -1.9 % of the rough-terrain launch).  Where it differs from the products (-DODK_GAUSS_PRODUCTS builds
        // those): an operand that is exactly +-0 counts as a signed infinitesimal instead of failing the test, i.e. two arcs that merely TOUCH may pass.  Such a pair's
        // edges still carry the support points along their common normal (the closed form of the same condition), so its separation is a true separation along a real
        // axis and cannot beat the exact maximum; it can only tie it (deviations).
        unsigned sp9 = 0u;
#pragma unroll
        for (int i = 8; i >= 0; i--) {
          const unsigned ucb = __float_as_uint(cb[i]), udb = __float_as_uint(db[i]), ua = __float_as_uint(pr[FA[i]]), ub = __float_as_uint(pr[FBK[i]]);
          // sign bit of r: cb db < 0 and adc bdc < 0 and cb bdc > 0.  Two v_bitop3_b32 (truth tables over A = 0xF0, B = 0xCC, C = 0xAA):
          // f1 = (cb ^ db) & ~(cb ^ bdc) = 0x24 on (cb, db, bdc);  r = f1 & (adc ^ bdc) = 0x60 on (f1, adc, bdc)
          const unsigned r = (unsigned)__builtin_amdgcn_bitop3_b32((int)__builtin_amdgcn_bitop3_b32((int)ucb, (int)udb, (int)ub, 0x24), (int)ua, (int)ub, 0x60);
          sp9 = __builtin_amdgcn_alignbit(sp9, r, 31);                   // (sp9 << 1) | sign(r): pair i ends at bit i
        }
        pass |= sp9 << (9 * s);
#else
#pragma unroll
        for (int i = 0; i < 9; i++) {
          const float adc = pr[FA[i]], bdc = pr[FBK[i]];
          // cb db < 0 and adc bdc < 0 and cb bdc > 0  <=>  max(cb db, adc bdc, -(cb bdc)) < 0: one v_max3_f32 + one compare instead of three compares and
          // two scalar ANDs per pair (round 6)
          const bool ok = fmaxf(fmaxf(cb[i] * db[i], adc * bdc), -(cb[i] * bdc)) < 0.0f;
          pass |= ok ? (1u << (9 * s + i)) : 0u;
        }
#endif
        pass &= RB.on[s] ? ~0u : ~(0x1FFu << (9 * s));      // (a lane without a hull edge in this slot passes nothing)
      }
    }
    SAT_PROF(2);
    HF_REP(13) {
    HF_TOUCH(pass);
    HF_REP_SYNC();
    // The passing pairs of the ROW are shared out evenly: a lane's own count varies from 0 to a dozen, and worked off lane by lane
    // the loop ran as long as the unluckiest lane of the wave.  Every lane PUSHES its passing pairs into the row's list at its
    // place in the row's running count (prefix sum over the row by DPP): one word per pair -- prism edge << 22 | hull edge << 16 |
    // the hull edge's two vertices -- so that lane j then works entries j, j + 16, ... with one LDS read each and no search.  The
    // list holds 64 pairs (S.PW | S.VV); a longer one (rare) is worked off in windows of 64.
    int incl = __popc(pass);
    const int cnt = incl;
    incl += (int)ODK_DPPU(incl, 0x111); incl += (int)ODK_DPPU(incl, 0x112); incl += (int)ODK_DPPU(incl, 0x114); incl += (int)ODK_DPPU(incl, 0x118);   // row_shr 1, 2, 4, 8
    const int total = (knock & 16) ? 0 : __shfl(incl, ODK_ROWBASE | 15, 64);
    float* PL = S.PW;
#ifdef ODK_PROFILE
    if (prof) prof[-6] += (float)total;   // (S_PROF2 + 2: passing edge pairs of the row, summed over the iterations)
#endif
    float best = -3.0e38f, bax[3] = {0.0f, 0.0f, 1.0f};
    int bi = 0x7FFFFFFF;
    for (int wb = 0; __builtin_amdgcn_ballot_w64(wb < total) != 0; wb += 64) {
      int pos = incl - cnt - wb;   // this lane's first place, relative to the window
#pragma unroll
      for (int s = 0; s < NSLOT; s++) {
        const unsigned base = (unsigned)RB.vv[s] | ((unsigned)(j + 16 * s) << 16);
#pragma unroll 1
        for (unsigned bits = (pass >> (9 * s)) & 0x1FFu; __builtin_amdgcn_ballot_w64(bits != 0u) != 0; ) {
          // (two bits per trip: half the loop-control round trips; predicated writes, everything else unconditional: 0 & (0 - 1) stays 0)
          const bool on0 = bits != 0u;
          const int i0 = __ffs((int)bits) - 1;
          bits &= bits - 1u;
          const bool on1 = bits != 0u;
          const int i1 = __ffs((int)bits) - 1;
          bits &= bits - 1u;
          if (on0 & ((unsigned)pos < 64u)) PL[pos] = __uint_as_float(base | ((unsigned)i0 << 22));
          if (on1 & ((unsigned)(pos + 1) < 64u)) PL[pos + 1] = __uint_as_float(base | ((unsigned)i1 << 22));
          pos += (on0 ? 1 : 0) + (on1 ? 1 : 0);
        }
      }
      ODK_SYNC();
      // one pair of the list: written without branches (a lane without an entry, or with edges closer to parallel than 1e-4 (sine), leaves `best` alone)
      auto eval_pair = [&](unsigned en, bool has) {
        const int i = en >> 22;
        // prism edge i: vertices from the packed table (va | vb << 3, 6 bits per edge), geometry from the LDS copy
        const unsigned long long PKE = 0ull | (0ull | 1ull << 3) | ((1ull | 2ull << 3) << 6) | ((3ull | 5ull << 3) << 12) | ((0ull | 3ull << 3) << 18) |
                                       ((3ull | 4ull << 3) << 24) | ((1ull | 4ull << 3) << 30) | ((4ull | 5ull << 3) << 36) | ((2ull | 5ull << 3) << 42) | ((0ull | 2ull << 3) << 48);
        const int ve = (int)((PKE >> (6 * i)) & 63ull);
        float pa[3], qa[3], pb[3], qb[3], ea[3], eb[3], ta[3], ax[3], tt[3];
        ld3(pa, PV + 3 * (ve & 7)); ld3(qa, PV + 3 * (ve >> 3)); ld3(pb, B.V + 3 * (en & 255u)); ld3(qb, B.V + 3 * ((en >> 8) & 255u));
        sub3(ea, qa, pa); sub3(eb, qb, pb); sub3(ta, pa, pc);
        cross3(ax, ea, eb);
        const float l2 = dot3(ax, ax);
        const bool ok = has & (l2 >= 1e-8f * dot3(ea, ea) * dot3(eb, eb)) & (l2 > 1e-30f);
        const float inv = rsqrtf(l2);
        ax[0] *= inv; ax[1] *= inv; ax[2] *= inv;
        const bool flip = dot3(ax, ta) < 0.0f;
        ax[0] = flip ? -ax[0] : ax[0]; ax[1] = flip ? -ax[1] : ax[1]; ax[2] = flip ? -ax[2] : ax[2];
        sub3(tt, pb, pa);
        const float sp = dot3(ax, tt);
        const bool better = ok & ((sp > best) | ((sp == best) & ((int)en < bi)));   // (entries order like (i, hull edge))
        best = better ? sp : best; bi = better ? (int)en : bi;
        bax[0] = better ? ax[0] : bax[0]; bax[1] = better ? ax[1] : bax[1]; bax[2] = better ? ax[2] : bax[2];
      };
      // two entries per trip (j + 16 t, j + 16 (t + 1)): both entries' LDS reads are in flight together; a row's list holds 17 pairs on average, so the
      // second one is rarely idle for the whole wave (round 6; one entry per trip before: two dependent LDS round trips per 16 pairs)
#pragma unroll 1
      for (int t = 0; t < 4 && __builtin_amdgcn_ballot_w64(wb + 16 * t < total) != 0; t += 2) {
        const bool has0 = wb + 16 * t + j < total, has1 = wb + 16 * (t + 1) + j < total;
        // (both entries read by every lane and pinned: the list's 64 slots always exist; as `has ? read : 0` each read sat under its own exec mask)
        const float r0 = PL[(16 * t + j) & 63], r1 = PL[(16 * (t + 1) + j) & 63];
        asm volatile("" :: "v"(r0), "v"(r1));
        const unsigned en0 = has0 ? __float_as_uint(r0) : 0u, en1 = has1 ? __float_as_uint(r1) : 0u;
        eval_pair(en0, has0);
        eval_pair(en1, has1);
      }
      ODK_SYNC();
    }
    pair = row_argmax(best, bi, sep_e);
    // the lane that worked the winning pair hands its axis to the row
    const unsigned own = (unsigned)((__builtin_amdgcn_ballot_w64(bi == pair) >> (threadIdx.x & 48u)) & 0xFFFFull);
    const int src = own ? __ffs((int)own) - 1 : 0;
    for (int k = 0; k < 3; k++) eax[k] = row_get(bax[k], src);
    }
  }
  SAT_PROF(3);
  const bool ref_a = sep_a >= sep_b;
  const float face_sep = ref_a ? sep_a : sep_b;
  const bool is_edge = pair != 0x7FFFFFFF && sep_e > face_sep + 1e-5f;
  // ---- reference / incident faces
  face_b = (unsigned)face_b < (unsigned)B.nf ? face_b : 0;
  face_a = (unsigned)face_a < 5u ? face_a : 0;
  float n_ref[3], n_inc[3];
  int pf_pk;    // the hull face involved (reference or incident), packed polygon
  int pa_f;     // the prism face involved
  // The two cases -- the prism's face is the reference (ref_a) or the hull's -- are ONE sequence with the roles swapped at the end: the rows of a
  // wave rarely agree on the case, and as two row-uniform branches every row walked through both (round 6).  Prism face: face_a, or the one most
  // anti-parallel to the hull's separating face (first minimum); hull face: face_b, or the one most anti-parallel to that prism face.
  {
    float nhb[3];
    ld3(nhb, B.N + 3 * face_b);
    float bestp = 3.0e38f; int pf = 0;
#pragma unroll
    for (int f = 0; f < 5; f++) { float nf[3]; prism_norm(P, f, nf); const float sc = dot3(nf, nhb); if (sc < bestp) { bestp = sc; pf = f; } }
    pa_f = ref_a ? face_a : pf;
    // (one select per statement: the nested five-way form was compiled into basic blocks with exec-mask bookkeeping, ~40 instructions a component)
    float npr[3] = {P.ns[2][0], P.ns[2][1], 0.0f};
    npr[0] = pa_f == 3 ? P.ns[1][0] : npr[0]; npr[1] = pa_f == 3 ? P.ns[1][1] : npr[1];
    npr[0] = pa_f == 2 ? P.ns[0][0] : npr[0]; npr[1] = pa_f == 2 ? P.ns[0][1] : npr[1];
    npr[0] = pa_f == 1 ? 0.0f : npr[0]; npr[1] = pa_f == 1 ? 0.0f : npr[1]; npr[2] = pa_f == 1 ? -1.0f : npr[2];
    npr[0] = pa_f == 0 ? P.nt[0] : npr[0]; npr[1] = pa_f == 0 ? P.nt[1] : npr[1]; npr[2] = pa_f == 0 ? P.nt[2] : npr[2];
    float best = -3.0e38f, vm; int bi = 0x7FFFFFFF;
#pragma unroll
    for (int s = 0; s < NFS; s++) { const float sc = -dot3(fn[s], npr); if (FB.on[s] && sc > best) { best = sc; bi = j + 16 * s; } }
    int inc = row_argmax(best, bi, vm);
    inc = (unsigned)inc < (unsigned)B.nf ? inc : 0;
    const int hface = ref_a ? inc : face_b;
    float nh[3];
    ld3(nh, B.N + 3 * hface);
    pf_pk = __float_as_int(row_get(__int_as_float((hface >> 4) == 0 ? FB.poly[0] : FB.poly[NFS - 1]), hface & 15));
#pragma unroll
    for (int k = 0; k < 3; k++) { n_ref[k] = ref_a ? npr[k] : nh[k]; n_inc[k] = ref_a ? nh[k] : npr[k]; }
  }
  static_assert(NFS == 2, "face slots: face = lane + 16 slot, two slots");
  // packed prism polygons (count | v0 << 3 | v1 << 6 | v2 << 9 | v3 << 12), selected without a table load
  constexpr unsigned long long PPK = (unsigned long long)(3 | 0 << 3 | 1 << 6 | 2 << 9 | 0 << 12) | (unsigned long long)(3 | 3 << 3 | 5 << 6 | 4 << 9 | 3 << 12) << 15 |
                                     (unsigned long long)(4 | 0 << 3 | 3 << 6 | 4 << 9 | 1 << 12) << 30 | (unsigned long long)(4 | 1 << 3 | 4 << 6 | 5 << 9 | 2 << 12) << 45;
  const int pp_pk = pa_f == 4 ? (4 | 2 << 3 | 5 << 6 | 3 << 9 | 0 << 12) : (int)((PPK >> (15 * (pa_f & 3))) & 0x7FFFull);      // (a shift, not five-way selects: those became branches)
  const int pcnt = pp_pk & 7, fcnt = pf_pk & 7;
  const int rcnt = ref_a ? pcnt : fcnt, icnt = ref_a ? fcnt : pcnt;
  ODK_SYNC();
  if (act && j < 8) {
    const bool isr = j < 4; const int k = j & 3;
    const bool from_prism = isr == ref_a;
    const int cnt = from_prism ? pcnt : fcnt;
    if (k < cnt) {
      const int v = from_prism ? (pp_pk >> (3 + 3 * k)) & 7 : (pf_pk >> (3 + 5 * k)) & 31;
      const float* src = (from_prism ? PV : B.V) + 3 * v;
      float* dst = (isr ? S.RP : S.IP) + 3 * k;
      const float c0 = src[0], c1 = src[1], c2 = src[2];      // (all three reads, then the writes: element by element each write waited for its own read)
      dst[0] = c0; dst[1] = c1; dst[2] = c2;
    }
  }
  ODK_SYNC();
  SAT_PROF(4);
  if (!(knock & 32)) HF_REP(14) { HF_TOUCH(n_ref[0]); manifold_row<true>(S, rcnt, icnt, n_ref, n_inc, ref_a ? 1.0f : -1.0f, is_edge, j, act, cidx0); }
  SAT_PROF(5);
  if (is_edge) {   // row-uniform
    int ia = pair >> 22;
    ia = ia < 9 ? ia : 0;
    const int evv = pair & 0xFFFF;
    // PRISM_EDGE[ia][0] / [1] from packed tables, three bits per edge (nested selects became branches)
    constexpr unsigned VA = 0u | 1u << 3 | 3u << 6 | 0u << 9 | 3u << 12 | 1u << 15 | 4u << 18 | 2u << 21 | 0u << 24;
    constexpr unsigned VB = 1u | 2u << 3 | 5u << 6 | 3u << 9 | 4u << 12 | 4u << 15 | 5u << 18 | 5u << 21 | 2u << 24;
    static_assert(PRISM_EDGE[0][0] == 0 && PRISM_EDGE[2][0] == 3 && PRISM_EDGE[6][0] == 4 && PRISM_EDGE[7][0] == 2 && PRISM_EDGE[8][0] == 0 && PRISM_EDGE[0][1] == 1 &&
                  PRISM_EDGE[2][1] == 5 && PRISM_EDGE[3][1] == 3 && PRISM_EDGE[5][1] == 4 && PRISM_EDGE[8][1] == 2, "packed prism edge vertices");
    const int va = (int)((VA >> (3 * ia)) & 7u), vb = (int)((VB >> (3 * ia)) & 7u);
    edge_contact_row<true>(PV + 3 * va, PV + 3 * vb, B.V + 3 * (evv & 255), B.V + 3 * ((evv >> 8) & 255), sep_e, eax, S, j, act, cidx0);
  }
  ODK_SYNC();
}

// keeps the four smallest of TOP[4] ++ NEW[4] in TOP; entries are 8 floats (dist, pos[3], normal[3], candidate index) ordered by
// (dist, candidate index): the order of a stable sort of the whole candidate list, whatever order the pairs were worked in
__device__ __forceinline__ void merge_top4_row(float* TOP, const float* NEW, int j, bool act) {
  float mine[8], dist[8], cidx[8];
  const float* src = j < 4 ? TOP + 8 * j : NEW + 8 * ((j - 4) & 3);
  for (int k = 0; k < 8; k++) mine[k] = src[k];
  for (int k = 0; k < 8; k++) { const float* e = k < 4 ? TOP + 8 * k : NEW + 8 * (k - 4); dist[k] = e[0]; cidx[k] = e[7]; }
  int rank = 0;
  // (bitwise, not short-circuit: `||` / `&&` here were eight branches per merge)
  for (int k = 0; k < 8; k++) rank += (int)((dist[k] < mine[0]) | ((dist[k] == mine[0]) & (cidx[k] < mine[7])));
  ODK_SYNC();
  if (act && j < 8 && rank < 4) { float* o = TOP + 8 * rank; for (int k = 0; k < 8; k++) o[k] = mine[k]; }
  ODK_SYNC();
}

__device__ __forceinline__ void make_frame_dev(const float* n, float* fr) {   // mju_makeFrame / mjx math.make_frame
  float b[3] = {0.0f, 0.0f, 0.0f}, cc[3];
  if (fabsf(n[1]) < 0.5f) b[1] = 1.0f; else b[2] = 1.0f;
  const float dtb = dot3(n, b);
  b[0] -= dtb * n[0]; b[1] -= dtb * n[1]; b[2] -= dtb * n[2];
  const float inv = 1.0f / sqrtf(dot3(b, b));
  b[0] *= inv; b[1] *= inv; b[2] *= inv;
  cross3(cc, n, b);
  for (int t = 0; t < 3; t++) { fr[t] = n[t]; fr[3 + t] = b[t]; fr[6 + t] = cc[t]; }
}

}  // namespace odk
