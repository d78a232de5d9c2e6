// odk_kernels.h -- device code of the fused env step (included by odk_engine.hip only).
//
// Geometry: one workgroup = one wavefront (64 lanes) = 64/G environments, G lanes per env
// (G = 32 or 64).  Every per-env array lives in LDS for the whole env step; HBM is touched once at
// the start (state record, action) and once at the end (state record, obs, reward...).  Lanes map
// to bodies / dofs / sparse-matrix entries / constraint rows by phase.  Physics follows mjx.step as
// restated in oracle/odk_oracle.c; the reference reaches it through mjx_env.step at
// playground/open_duck_mini_v2/joystick.py:420.
//
// What makes it fast (all exact in real arithmetic; DESIGN.md 4.1 has the measurements):
//  * occupancy by construction: <= 256 VGPRs => 2 waves/SIMD = 8 workgroups/CU, and the LDS image is sized so that 8
//    workgroups fit (shape A: 2 296 floats/env) => 8192 envs are exactly two rounds.  Two waves keep the VALU ~70 % busy;
//    each wave is paced by its own chain of dependent instructions and LDS round trips (a second wave stretches every phase
//    by 1.0-1.2x only): with 2 ... 8 resident workgroups per CU the launch takes ceil(16 / w) rounds of ~0.2 ms
//    (profiles/r3/occupancy_scan.txt), so a third wave per SIMD (12 resident: still two rounds) would not shorten it.  What
//    pays is fewer instructions and fewer serialised round trips -- not fewer loads from the model, those are covered;
//  * per-lane statics (a lane's dof depth, row address, ancestor / descendant masks...) are read once per launch into
//    registers (struct Statics); phase-local constants are re-fetched from the L2-resident model where they are used;
//    the model pointer is made opaque once per substep so that table addresses are not hoisted into scratch;
//  * the body recursions (pose, cvel, cacc top-down; composite inertia, bias force bottom-up) are cross-lane prefix /
//    suffix scans over the serial chains of the tree (3 ds_bpermute steps each), not one LDS round trip per level;
//  * spatial quantities are expressed about the floating-base origin instead of the subtree COM;
//  * contact Jacobian rows are never formed: J_r = w_r . cdof[d] for dofs d above the foot, with the
//    6-vector w_r = [r x dir; dir], so J x, J^T f and J^T D J collapse to 6-vector / 6x6 algebra;
//  * both feet's plane-convex manifolds run in one pass, one 16-lane DPP row per foot;
//  * inertia and Newton Hessian share a tree-sparse row layout (row i = its ancestors by depth) and a fill-free
//    L^T D L: floating base + serial chains are solved by chain_solve -- the seven columns [C | y] of each chain's block
//    in seven lanes, the base block's Schur complement one entry per lane; generic trees eliminate the chains' pivots
//    together (factor_chains);
//  * the dense symmetric row of M sits in registers through the solver (gathered once per substep): both M v products
//    are 20 FMAs on LDS broadcast reads;
//  * the line search's bracketing iterations evaluate derivatives only (two sums per step size), the three costs that
//    pick the step are one evaluation at the end;
//  * group reductions are fused v_add_f32_dpp / v_max_u32_dpp butterflies + gfx950 v_permlane16/32_swap; whole-vector
//    broadcasts are LDS broadcast reads (the VALU is the busy unit, the LDS pipe is not).
#pragma once
#include <hip/hip_runtime.h>

#include "odk_model.h"

namespace odk {

// One workgroup == one wavefront: DS (LDS) instructions of a wave are issued and serviced in order, so
// cross-lane hand-offs through LDS need no s_barrier and no s_waitcnt -- only a compiler barrier that keeps
// the LDS accesses in program order (and stops values being cached in registers across the hand-off).
// NOTE: never put __restrict__ on an LDS pointer: with noalias the optimiser may treat the barrier as not
// touching that memory and reuse a value loaded before another lane's store.
#define ODK_SYNC() asm volatile("" ::: "memory")

constexpr float MINVAL_F = 1e-15f;
constexpr float PI_F = 3.14159265358979323846f;

struct EnvCfg {  // device copy of odk_env_config
  float ctrl_dt, action_scale, dof_vel_scale, max_motor_velocity;
  float noise_level, noise_gyro, noise_accelerometer, noise_gravity, noise_joint_vel;
  float qpos_noise_scale[16];
  float reward_scales[7];
  float tracking_sigma;
  float push_enable, push_interval_range[2], push_magnitude_range[2];
  float cmd_range[7][2];
  int use_imitation, use_motor_speed_limits, autoreset, episode_length, n_substeps;
  int kind, nobs, npriv;   // env kind (0 Joystick, 1 Standing) and its output row strides
  float reset_base_qvel;
};

// ------------------------------------------------------------------------------------------------
// LDS layout (floats), per environment.  Component-major (SoA) arrays: X[k * N + item].
template <int NQ_, int NV_, int NB_, int NU_, int NJ_, int NM_, int NH_, int NROW_, int DT_, int DV_, bool CONE_ = false, int CL_ = -1, bool OPT_ = false>
struct Shape {
  static constexpr int NQ = NQ_, NV = NV_, NB = NB_, NU = NU_, NJ = NJ_, NM = NM_, NH = NH_, NROW = NROW_;
  static constexpr int DT = DT_;    // max dof depth, kinematic tree
  static constexpr int DV = DV_;    // max dof depth, virtual (Hessian) tree
  static constexpr int NCROW = 48;  // contact rows
  // Twin dofs (the backlash model, nv 30): solves run on the reduced tree -- twins merged into their main dof, which is
  // the 20-dof robot's own tree (DevModel::paired; reduced_rhs / reduced_expand below).
  static constexpr bool PAIRED = (NV_ == 30);
  // <equality><joint> rows (DevModel::neq) are compiled into this shape's kernels: the third shape (tests/assets/tail_biped*.xml) -- the
  // duck's shapes have no equality and do not pay for the code
  static constexpr bool EQ = (NV_ == 21) || OPT_;      // (OPT_: a shape that asks for the optional constraint code -- equality rows, elliptic cones as a runtime switch)
  // <option cone="elliptic"> (DevModel::cone) is compiled into the same shape's kernels: a contact's four row lanes hold normal | tangent 1 |
  // tangent 2 | nothing instead of the four pyramid edges, and the cost of a contact is the cone's (odk_kernels.h "elliptic cones")
  static constexpr bool CONE = CONE_;                   // an instantiation that is ONLY launched for cone = 1 models: the pyramid code is compiled out
  static constexpr bool ELL = (NV_ == 21) || CONE_ || OPT_;    // (the duck's shapes: their own instantiations with CONE_ = true, launched for models with cone = 1 only)
  static constexpr int NVR = PAIRED ? 20 : NV_;    // reduced dofs
  static constexpr int NMR = PAIRED ? 145 : NM_;   // entries of the reduced tree layout
  static constexpr int NHR = PAIRED ? 170 : NH_;   // entries of the reduced virtual-tree layout
  static constexpr int DVR = PAIRED ? 15 : DV_;    // max dof depth, reduced virtual tree
  // workgroup-shared LDS tables behind the envs' images (odk_engine.hip load_shared): the packed reduced entries and the
  // contact-row constants.  (Friction-loss rows, actuator constants and the foot hull were tried there too: no gain, their
  // loads from the L2-resident model are already covered.)
  static constexpr int SH_CT = NMR, SHARED = SH_CT + 42;
  static constexpr int CL = CL_ >= 0 ? CL_ : ((NV_ == 20 || NV_ == 21 || PAIRED) ? 5 : 0);   // max (reduced) chain length for chain_solve: every compiled shape is a floating base + <= 3 serial chains of <= 5 (reduced) dofs (0: generic path)
  // persistent over the env step
  static constexpr int O_QPOS = 0;
  static constexpr int O_QVEL = O_QPOS + NQ;
  static constexpr int O_WARM = O_QVEL + NV;
  static constexpr int O_CTRL = O_WARM + NV;
  // per-env effective model parameters (domain randomisation lands here)
  static constexpr int O_Q0 = O_CTRL + NU;     // qpos0
  static constexpr int O_MASS = O_Q0 + NQ;     // body_mass
  static constexpr int O_ARM = O_MASS + NB;    // dof_armature
  static constexpr int O_FRL = O_ARM + NV;     // dof_frictionloss
  static constexpr int O_KP = O_FRL + NV;      // actuator kp
  static constexpr int O_IPOS1 = O_KP + NU;    // body_ipos[1]
  // tree sweeps (component-major by body)
  static constexpr int O_XPOS = O_IPOS1 + 3;         // [3][NB]
  static constexpr int O_XQUAT = O_XPOS + 3 * NB;    // [4][NB]
  static constexpr int O_CVEL = O_XQUAT + 4 * NB;    // [6][NB]
  static constexpr int O_CACC = O_CVEL + 6 * NB;     // [6][NB] velocity-dependent part of cacc (gravity folded in)
  static constexpr int O_CFRC = O_CACC + 6 * NB;     // [6][NB] local then subtree-accumulated bias force
  static constexpr int O_CRB = O_CFRC + 6 * NB;      // [10][NB] cinert then composite inertia
  // matrix work runs on the REDUCED dofs (twins merged, DevModel::paired): one motion column per reduced dof
  static constexpr int O_CDOF = O_CRB + 10 * NB;     // [6][NVR]
  static constexpr int O_BUF6 = O_CDOF + 6 * NVR;    // [6][NVR] crb*cdof -> K_L*cdof
  static constexpr int O_BUF6B = O_BUF6 + 6 * NVR;   // [6][NVR] K_R*cdof
  static constexpr int O_M = O_BUF6B + 6 * NVR;      // [NMR] sparse reduced inertia (rows by ancestor depth; twin pairs: no armature)
  static constexpr int O_HL = O_M + NMR;             // [NHR] reduced inertia / Hessian with the diagonal terms, and its factor
  // dof vectors
  static constexpr int O_QFS = O_HL + NHR;           // qfrc_smooth
  static constexpr int O_QAS = O_QFS + NV;           // qacc_smooth
  static constexpr int O_X = O_QAS + NV;             // current qacc iterate
  static constexpr int O_MA = O_X + NV;              // scratch: per-dof friction-row hand-over
  static constexpr int O_GRAD = O_MA + NV;           // search direction
  static constexpr int O_MV = O_GRAD + NV;           // scratch: per-dof friction-row hand-over
  // constraint rows
  static constexpr int O_D = O_MV + NV;              // efc_D (0 = structurally inactive row)
  static constexpr int O_AREF = O_D + NROW;
  static constexpr int O_JAR = O_AREF + NROW;        // J qacc - aref (contact rows)
  static constexpr int O_JV = O_JAR + NROW;          // J search (scratch: candidate Jaref, forces, Hessian diagonal addend)
  static constexpr int O_SC = O_JAR;                 // [2][NJ] sin/cos of the half joint angles: P0 -> P1 only, ALIASES jar (born in P9)
  static_assert(2 * NJ <= NROW, "sin/cos must fit in the jar rows");
  // [NCROW][6] contact row wrenches [r x dir; dir]: ALIAS cfrc|crb, which are dead once the bias forces and M entries exist (P3/P4; W is
  // born in P8) -- where that region is large enough (18 bodies); a smaller robot gets its own floats behind the image
  static constexpr bool W_FITS = 6 * NCROW <= 16 * NB;
  static constexpr int O_CDIST = O_JV + NROW;        // [12]
  static constexpr int O_CR = O_CDIST + NCON;        // [12][3] contact position relative to the base origin
  static constexpr int O_SCR = O_CR + 3 * NCON;      // scratch: foot twists, wrenches, 6x6 blocks, sensor inputs
  // scratch sub-offsets
  static constexpr int S_VF = 0;      // [2][6] foot twist of the current vector
  static constexpr int S_FF = 12;     // [2][6] foot wrench sums
  static constexpr int S_K = 24;      // [3][36] K_L, K_R, K_X
  static constexpr int S_FR = S_K;    // [8][9] frames of the floor contacts (height field / primitive feet): contact phase -> row phase only, ALIASES
                                      // the K blocks (born in the solver).  NOT in the row arrays: the foot-foot routine's hull copies live there.
  static constexpr int S_VF2 = 132;   // [2][6] second foot twist (warmstart candidate)
  static constexpr int S_FFX = 144;   // [6] wrench sum of the foot-foot rows
  static constexpr int S_EQ = 150;    // [EQ_MAX] equality rows: the Hessian's off-diagonal addend -D c (force phase -> Hessian entries; the solve's scratch runs over it afterwards)
  // the chain solve's scratch starts at S_K and takes 3 CL 7 + 27 floats (156 for chains of five: up to S_MISC; 177 for chains of six): it runs over
  // S_VF2 / S_FFX / S_EQ, which are dead by then -- but never over the misc scalars, which the env kernels' epilogue reads (foot heights, gravity)
  static constexpr int S_SOLVE_END = S_K + 21 * CL + 27;
  static constexpr int S_MISC = S_SOLVE_END > 156 ? S_SOLVE_END : 156;  // misc scalars (16)
  static constexpr int S_PROF = S_MISC + 16;   // [20] per-phase cycle counters (ODK_PROFILE builds)
  static constexpr int S_PROF2 = S_PROF + 20;  // [16] height-field contacts: hull setup, cull pass, register loads, pair loop, iterations, list length, -, -,
                                               //      then inside a pair: select + prism, face query, Gauss-map tests, passing pairs, faces / polygons, clip + manifold, merge
#ifdef ODK_PROFILE
  static constexpr int N_SCR = S_PROF2 + 16;
#else
  static constexpr int N_SCR = ((S_MISC + 16 + 3) / 4) * 4;      // no S_PROF slots outside profile builds (172 for chains of five)
#endif
  static constexpr int O_SENS = O_JV;                // sensordata[46]: born after the line search (P10, last substep only), ALIASES jv
  static_assert(NSENSD <= NROW, "sensordata must fit in the jv rows");
  static constexpr int O_ACTF = O_SCR + N_SCR;       // actuator_force
  static constexpr int O_QACC = O_X;                 // qacc of the last forward == final iterate
  // path rows of shapes with EQ (<equality><connect | weld>, DevModel::neqp): wrenches [EQP_ROWS][12] | J [EQP_ROWS][NV] | D | aref | residual
  static constexpr int O_EQP = ((O_ACTF + NU + 3) / 4) * 4;
  static constexpr int EQP_W = 0, EQP_J = 12 * EQP_ROWS, EQP_D = EQP_J + EQP_ROWS * NV, EQP_AREF = EQP_D + EQP_ROWS, EQP_POS = EQP_AREF + EQP_ROWS;
  static constexpr int N_EQP = EQ ? ((EQP_POS + EQP_ROWS + 3) / 4) * 4 : 0;
  static constexpr int O_WX = O_EQP + N_EQP;         // the contact row wrenches of a robot with fewer than 18 bodies
  static constexpr int O_W = W_FITS ? O_CFRC : O_WX;
  static constexpr int TOTAL = O_WX + (W_FITS ? 0 : 6 * NCROW);
  // the env logic's floats behind the physics image (odk_engine.hip EnvL): the carried info (43 + 7 NU floats, rec_lay) and this step's action + the
  // imitation phase, both rounded up to whole float4s (the duck: 144 + 16)
  static constexpr int N_INFO = ((43 + 7 * NU + 3) / 4) * 4, N_ACT = ((NU + 2 + 3) / 4) * 4;
  static constexpr int ENV_STRIDE = TOTAL + N_INFO + N_ACT;   // floats between the images of the two envs of a workgroup (odk_engine.hip EnvL::TOTAL)
};

// Phase timing (build with -DODK_PROFILE): lane 0 accumulates shader-clock deltas per phase into the
// scratch area, which the debug LDS image carries out.  Zero cost when the macro is off.
#ifdef ODK_PROFILE
#define ODK_PROF(i) do { if (lane == 0) { long long _t = clock64(); SCR[S::S_PROF + (i)] += (float)(_t - _tprev); _tprev = _t; } } while (0)
#define ODK_PROF_BEGIN() long long _tprev = clock64()
#elif defined(ODK_MARK)   // phase markers in the ISA listing (tools/isa_phase_stats.py): comments only
#define ODK_PROF(i) asm volatile("; ODK_PHASE_END " #i ::: "memory")
#define ODK_PROF_BEGIN() asm volatile("; ODK_PHASE_BEGIN" ::: "memory")
#else
#define ODK_PROF(i) do { } while (0)
#define ODK_PROF_BEGIN() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Cross-lane all-reduce over the G lanes of one env with DPP (no LDS traffic): xor-1 / xor-2 quad permutes,
// row_half_mirror, row_mirror give every lane its 16-lane row total; rows are then combined with row_bcast15
// (/31) and the group total is read back with v_readlane.  MUST be called in uniform control flow.
// old = 0 with bound_ctrl lets the compiler fold the permute into the consuming VALU op (v_add_f32_dpp ...): one
// instruction per butterfly stage, no copies, no hazard nops.  (All four patterns read only enabled in-row lanes.)
#define ODK_DPP(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, true))
struct OpSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
struct OpMin { static __device__ __forceinline__ float f(float a, float b) { return fminf(a, b); } };
typedef unsigned int odk_u2 __attribute__((ext_vector_type(2)));
template <int G, class Op> __device__ __forceinline__ float greduce(float v) {
  v = Op::f(v, ODK_DPP(v, 0xB1, 0xF));    // quad_perm [1,0,3,2]
  v = Op::f(v, ODK_DPP(v, 0x4E, 0xF));    // quad_perm [2,3,0,1]
  v = Op::f(v, ODK_DPP(v, 0x141, 0xF));   // row_half_mirror
  v = Op::f(v, ODK_DPP(v, 0x140, 0xF));   // row_mirror: every lane holds its row's total
  // gfx950 v_permlane16_swap: rows (r0 r1 r2 r3),(s0 s1 s2 s3) -> (r0 s0 r2 s2),(r1 s1 r3 s3); with both operands = v the
  // two results are (x0 x0 x2 x2) and (x1 x1 x3 x3): their combination is the 32-lane group total in every lane.
  const unsigned iv = __float_as_uint(v);
  const odk_u2 h = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  v = Op::f(__uint_as_float(h[0]), __uint_as_float(h[1]));
  if (G == 64) {   // v_permlane32_swap: halves (lo hi),(lo hi) -> (lo lo),(hi hi)
    const unsigned iw = __float_as_uint(v);
    const odk_u2 w = __builtin_amdgcn_permlane32_swap(iw, iw, false, false);
    v = Op::f(__uint_as_float(w[0]), __uint_as_float(w[1]));
  }
  return v;
}
template <int G> __device__ __forceinline__ float gsum(float v) { return greduce<G, OpSum>(v); }
// N sums at once (G = 32): the four in-row butterfly stages on the VALU as above, then the two rows are exchanged by
// ds_swizzle (xor 16 inside each 32-lane group: the LDS crossbar, no memory access and no VALU slot) instead of
// v_permlane16_swap (~8 cycles of VALU issue each); the N swizzles are in flight together, so their latency is paid once.
template <int G, int N> __device__ __forceinline__ void gsum_n(float* v) {
  if constexpr (G == 32) {
    float o[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      float x = v[i];
      x += ODK_DPP(x, 0xB1, 0xF); x += ODK_DPP(x, 0x4E, 0xF); x += ODK_DPP(x, 0x141, 0xF); x += ODK_DPP(x, 0x140, 0xF);
      v[i] = x;
      o[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));   // bit mode: and 0x1F, or 0, xor 0x10
    }
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += o[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = gsum<G>(v[i]);
  }
}
// max / min / argmax run on order-preserving unsigned keys: integer v_max_u32 / v_min_u32 fold the DPP permute into the
// op (float max needs a canonicalising v_max x, x per stage under IEEE mode, and then the permute stays a separate mov)
__device__ __forceinline__ unsigned fkey(float v) { const unsigned b = __float_as_uint(v); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
#define ODK_DPPU(v, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true))
template <int G, bool MAX> __device__ __forceinline__ unsigned greduce_u(unsigned v) {
  auto op = [](unsigned a, unsigned b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = op(v, ODK_DPPU(v, 0xB1));
  v = op(v, ODK_DPPU(v, 0x4E));
  v = op(v, ODK_DPPU(v, 0x141));
  v = op(v, ODK_DPPU(v, 0x140));
  const odk_u2 h = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = op(h[0], h[1]);
  if (G == 64) {
    const odk_u2 w = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = op(w[0], w[1]);
  }
  return v;
}
template <int G> __device__ __forceinline__ float gmax(float v) { return fkey_inv(greduce_u<G, true>(fkey(v))); }
template <int G> __device__ __forceinline__ float gmin(float v) { return fkey_inv(greduce_u<G, false>(fkey(v))); }
// argmax, lowest index wins ties (jnp.argmax semantics): max of the values, then min index among the maxima
template <int G> __device__ __forceinline__ int gargmax(float v, int i) {
  const unsigned k = fkey(v);
  const unsigned mx = greduce_u<G, true>(k);
  return (int)greduce_u<G, false>(k == mx ? (unsigned)i : 0x7FFFFFFFu);
}
// value held by lane j (uniform j < G) of this env's lane group, via v_readlane (no LDS)
template <int G> __device__ __forceinline__ float bcast(float v, int j) {
  const int iv = __float_as_int(v);
  if (G == 64) return __int_as_float(__builtin_amdgcn_readlane(iv, j));
  const int lo = __builtin_amdgcn_readlane(iv, j), hi = __builtin_amdgcn_readlane(iv, j + 32);
  return __int_as_float((threadIdx.x & 32) ? hi : lo);
}
// uniform integer static of lane j (identical in every env of the wave)
__device__ __forceinline__ int ubcast(int v, int j) { return __builtin_amdgcn_readlane(v, j); }

__device__ __forceinline__ void cross3(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void qmul(float* r, const float* a, const float* b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void qnormalize(float* q) {   // v_rsq_f32 (1 ulp) instead of IEEE sqrt + division (~20 instructions)
  const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 < MINVAL_F * MINVAL_F) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  const float inv = __builtin_amdgcn_rsqf(n2);
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
__device__ __forceinline__ void qrot(float* r, const float* q, const float* v) {  // r = R(q) v
  float t[3], u[3] = {q[1], q[2], q[3]}, c[3];
  cross3(t, u, v);
  t[0] *= 2; t[1] *= 2; t[2] *= 2;
  cross3(c, u, t);
  r[0] = v[0] + q[0] * t[0] + c[0]; r[1] = v[1] + q[0] * t[1] + c[1]; r[2] = v[2] + q[0] * t[2] + c[2];
}
__device__ __forceinline__ void q2mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
// spatial inertia (10: Ixx Iyy Izz Ixy Ixz Iyz mcx mcy mcz m) times motion [ang; lin] (mju_mulInertVec)
__device__ __forceinline__ void inert_mul(float* res, const float* i, const float* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float* res, const float* vel, const float* v) {  // mju_crossMotion
  float a[3], b[3];
  cross3(res, vel, v);
  cross3(a, vel, v + 3);
  cross3(b, vel + 3, v);
  res[3] = a[0] + b[0]; res[4] = a[1] + b[1]; res[5] = a[2] + b[2];
}

// ---- RNG: threefry2x32-20, stream definition shared with oracle/odk_oracle_env.c
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ inline void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
  for (int blk = 0; blk < 5; blk++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      x0 += x1;
      x1 = rotl32(x1, R[(blk & 1) * 4 + r]);
      x1 ^= x0;
    }
    x0 += ks[(blk + 1) % 3];
    x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  o0 = x0; o1 = x1;
}
__device__ inline float rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr, uint32_t idx) {
  uint32_t a, b;
  threefry2x32(k0, k1, ctr, idx >> 1, a, b);
  return (float)(((idx & 1) ? b : a) >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ int randint3(float u) { int i = (int)(u * 3.0f); return i > 2 ? 2 : i; }

// ------------------------------------------------------------------------------------------------
// Per-lane static data, loaded once per launch.  A lane plays several roles (body `lane`, dof `lane`,
// joint `lane`, hull vertex `lane`, a few matrix entries and constraint rows); roles beyond the
// model's counts are disabled by their flags.
// Persistent part: what the factor / solve / solver phases need all the time (kept small: register pressure
// decides whether two wavefronts fit on a SIMD).
template <class S, int G>
struct Statics : LaneSt {   // (data members: odk_model.h LaneSt)
  static constexpr int NME = (S::NMR + G - 1) / G;  // reduced inertia entries per lane
  static constexpr int NHE = (S::NHR + G - 1) / G;  // reduced virtual-tree Hessian entries per lane
  static_assert((S::NVR + 1) / 2 <= 16, "LaneSt::m_adr");
};
// Phase-local statics: fetched from the model (L1/L2-resident, one batch of loads per phase and substep)
// right where they are used, so they do not occupy registers for the rest of the substep.
// (BodySt: odk_model.h -- the per-body records are part of the device model)
struct ActSt { float bias2, clo, chi, flo, fhi; int climited, flimited; };
struct FlSt { float D, R, b; int dof; };

// The same three records held over the whole launch instead (plane-floor kernels: ~40 registers are free there since round 3, and
// each of these loads sat exposed right behind a phase hand-off in every substep)
struct HotSt { ActSt as; FlSt fs; float vb[3]; };   // (the non-chain bodies' source lists the same way: +1 %, the limit rows' records: over the register budget)

// Device side: the lane's host-built record (DevModel::lane_st, filled by compute_statics below at model load)
template <class S, int G>
__device__ __forceinline__ void load_statics(Statics<S, G>& st, const DevModel* __restrict__ m, int lane) {
  static_cast<LaneSt&>(st) = m->lane_st[lane];
}
template <class S>
__host__ __device__ inline void compute_statics(LaneSt& st, const DevModel* m, int lane) {
  {
    const bool on = lane >= 1 && lane < S::NJ;
    st.j_qadr = on ? m->jnt_qposadr[on ? lane : 0] : -1;
    st.j_dadr = m->jnt_dofadr[on ? lane : 0];
  }
  {
    st.cs_pk = 0;
    const int c = lane >> 3;
    if (c < 3 && c < m->nrchain) {
      const int f = m->rchain_first[c];
      st.cs_pk = m->rchain_len[c] | (m->red_depth[f] << 3) | (f << 8) | (m->red_Madr[f] << 14);
    }
    int tb = 0, tb2 = 0;
    if (lane < 21) { while ((tb + 1) * (tb + 2) / 2 <= lane) tb++; tb2 = lane - tb * (tb + 1) / 2; }
    else if (lane < 27) { tb = lane - 21; tb2 = 6; }
    st.cs_pk |= (tb << 24) | (tb2 << 27);
  }
  st.ch_first = 0; st.ch_len = 0;
  for (int c = 0; c < 3; c++)
    if (c < m->nrchain && lane >= m->rchain_first[c] && lane < m->rchain_first[c] + m->rchain_len[c]) { st.ch_first = m->rchain_first[c]; st.ch_len = m->rchain_len[c]; }
  {
    const int r = lane < S::NVR ? lane : 0;
    st.r_on = lane < S::NVR;
    st.r_depth = m->red_depth[r]; st.r_Madr = m->red_Madr[r]; st.r_ancmask = m->red_ancmask[r]; st.r_descmask = m->red_descmask[r];
    st.r_foot = st.r_on ? m->red_foot[r] : 0;
    if (!st.r_on) { st.r_ancmask = 0; st.r_descmask = 0; st.r_depth = 0; }
    // dofs are numbered parents first: j < lane can only be an ancestor of this lane's dof (entry in this lane's row, column
    // depth_j), j > lane only a descendant (entry in j's row, column depth_lane), j == lane the diagonal.  Unrelated pairs
    // (and lanes past the reduced dofs) point at CDOF[0][0], the angular x component of the base's x translation: always 0.
    const int rel = st.r_ancmask | st.r_descmask | (st.r_on ? (1 << r) : 0);
    for (int jp = 0; jp < (S::NVR + 1) / 2; jp++) {
      int pk = 0;
      for (int h = 0; h < 2; h++) {
        const int j = 2 * jp + h;
        int off = S::O_CDOF * 4;
        if (j < S::NVR && ((rel >> j) & 1)) off = (S::O_M + (j < r ? st.r_Madr + m->red_depth[j] : m->red_Madr[j] + st.r_depth)) * 4;
        pk |= off << (16 * h);
      }
      st.m_adr[jp] = pk;
    }
  }
  const int i = lane < S::NV ? lane : 0;
  st.d_on = lane < S::NV;
  st.d_body = m->dof_body[i];
  st.d_act = st.d_on ? m->dof_act[i] : -1;
  st.d_flrow = st.d_on ? m->dof_flrow[i] : -1;
  st.d_limrow = st.d_on ? m->dof_limrow[i] : -1;
  st.d_foot = st.d_on ? (m->foot_dofmask[0][i] | (m->foot_dofmask[1][i] << 1)) : 0;
  st.d_damping = m->dof_damping[i];
  st.d_lim_on = st.d_limrow >= 0;
  st.d_qadr = m->dof_qadr[i];
  st.d_lo = m->dof_range[i][0]; st.d_hi = m->dof_range[i][1];
  st.d_tkind = (S::PAIRED && st.d_on) ? m->dof_tkind[i] : 0;
  st.d_red = S::PAIRED ? m->dof_red[i] : i;   // column of this dof's motion vector in CDOF / BUF6
}
template <class S>
__device__ __forceinline__ void load_body(BodySt& b, const DevModel* __restrict__ m, int lane) {
  b = m->body_st[lane < S::NB ? lane : MAXB];   // one record per body, host-built (DevModel::body_st)
}
__device__ __forceinline__ void load_act(ActSt& a, const DevModel* __restrict__ m, int act) {
  const int u = act >= 0 ? act : 0;
  a.bias2 = m->act_bias2[u]; a.clo = m->act_ctrlrange[u][0]; a.chi = m->act_ctrlrange[u][1];
  a.flo = m->act_forcerange[u][0]; a.fhi = m->act_forcerange[u][1];
  a.climited = m->act_ctrllimited[u]; a.flimited = m->act_forcelimited[u];
}
__device__ __forceinline__ void load_fl(FlSt& f, const DevModel* __restrict__ m, int lane) {
  const int r = lane < m->nfl ? lane : 0;
  f.D = m->fl_D[r]; f.R = m->fl_R[r]; f.b = m->fl_b[r]; f.dof = m->fl_dof[r];
}
template <class S, int G>
__device__ __forceinline__ void load_hot(HotSt& h, const DevModel* __restrict__ m, const Statics<S, G>& st, int lane) {
  load_act(h.as, m, st.d_act);
  load_fl(h.fs, m, lane);
  const int f = (lane >> 4) & 1, j = lane & 15;   // plane-convex: foot f = 16-lane row f, lane j = hull vertex j
  const int jv = (lane < 32 && j < m->foot_nvert[f]) ? j : 0;
  for (int k = 0; k < 3; k++) h.vb[k] = m->foot_vert[f][jv][k];
}
// packed entries of the reduced layouts (DevModel::R_ent / RH_ent), -1 beyond the layout
__device__ __forceinline__ int load_rent(const DevModel* __restrict__ m, int p, int n) { const int e = m->R_ent[p < n ? p : 0]; return p < n ? e : -1; }
__device__ __forceinline__ int load_rhent(const DevModel* __restrict__ m, int p, int n) { const int e = m->RH_ent[p < n ? p : 0]; return p < n ? e : -1; }

// ------------------------------------------------------------------------------------------------
// Sparse L^T D L on the tree layout (row i = [L(i, ancestor at depth 0..d-1), D_i]).  Each lane keeps its
// own row in registers; at step k lane k publishes its finished row to LDS and the ancestors of k fold it in.
template <int G, int DMAX, int NVT>
__device__ __forceinline__ void factor_rows(float* A, int lane, int on, int di, int ai, int descmask, int depth_st, int madr_st) {
  // all LDS loads are unconditional (in-bounds of the env's LDS image even when they run past a row) and
  // masked afterwards with selects: a conditional load costs a branch and a full LDS round trip each
  float row[DMAX > 0 ? DMAX : 1];
#pragma unroll
  for (int c = 0; c < DMAX; c++) row[c] = A[ai + c];
  float diag = A[ai + di];
  if (!on) diag = 1.0f;
  for (int k = NVT - 1; k > 0; k--) {
    const int Dk = ubcast(depth_st, k);
    if (Dk == 0) continue;
    const int ak = ubcast(madr_st, k);
    if (lane == k) {
      const float inv = __builtin_amdgcn_rcpf(diag);
#pragma unroll
      for (int c = 0; c < DMAX; c++)
        if (c < Dk) { row[c] *= inv; A[ak + c] = row[c]; }   // Dk is scalar: plain scalar branches
      A[ak + Dk] = diag;
    }
    ODK_SYNC();
    {
      const bool anc = (descmask >> k) & 1;  // this lane's dof is a strict ancestor of k
      float rk[DMAX > 0 ? DMAX : 1];
#pragma unroll
      for (int c = 0; c < DMAX; c++) rk[c] = A[ak + c];
      const float Lki = A[ak + di], dk = A[ak + Dk];
      const float t = anc ? Lki * dk : 0.0f;
#pragma unroll
      for (int c = 0; c < DMAX; c++) row[c] = anc ? fmaf(-t, rk[c], row[c]) : row[c];
      diag = anc ? fmaf(-t, Lki, diag) : diag;
    }
    ODK_SYNC();
  }
  if (on) A[ai + di] = diag;  // rows at depth 0 were never published
  ODK_SYNC();
}

// Same factorisation for a "floating base + serial dof chains" tree (this robot with backlash joints: 6 + 10 + 4 + 10):
// the chains are independent of each other, so their pivots at equal distance from the leaf are eliminated in the
// same step (MAXLEN steps instead of NV - 1), each chain lane folding only its own chain's pivot row; the base block
// receives the chains' Schur complement afterwards and is finished with the generic loop (NBASE - 1 steps).
// Produces the same LDS image as factor_rows (row k = scaled L entries, then D_k).
template <int G, int DMAX, int NVT, int MAXLEN, int NBASE>
__device__ __forceinline__ void factor_chains(float* A, int lane, int on, int di, int ai, int ch_first, int ch_len, int descmask,
                                              int depth_st, int madr_st) {
  float row[DMAX > 0 ? DMAX : 1];
#pragma unroll
  for (int c = 0; c < DMAX; c++) row[c] = A[ai + c];
  float diag = A[ai + di];
  if (!on) diag = 1.0f;
  const int a_loc = lane - ch_first;
#pragma unroll
  for (int s = 0; s < MAXLEN; s++) {            // unrolled: in step s no pivot is deeper than DMAX - s, rows shrink with s
    const int kl = ch_len - 1 - s;              // local index of this chain's pivot in step s
    const int t = kl - a_loc;                   // pivot dof = lane + t
    const bool act = ch_len > 0 && kl >= 0;
    if (act && t == 0) {                        // the pivot publishes its finished row
      const float inv = __builtin_amdgcn_rcpf(diag);
#pragma unroll
      for (int c = 0; c < DMAX; c++)
        if (c < DMAX - s && c < di) { row[c] *= inv; A[ai + c] = row[c]; }
      A[ai + di] = diag;
    }
    ODK_SYNC();
    {
      const bool anc = act && t > 0;            // this lane's dof is above the pivot in the same chain
      const int tk = anc ? t : 0;
      const int ak = ai + tk * (di + 1) + (tk * (tk - 1)) / 2, Dk = di + tk;   // consecutive dofs: rows are adjacent, one entry longer each
      float rk[DMAX > 0 ? DMAX : 1];
#pragma unroll
      for (int c = 0; c < DMAX; c++)
        if (c < DMAX - s - 1) rk[c] = A[ak + c];   // an ancestor's own row is at least one entry shorter than the pivot's
      const float Lki = A[ak + di], dk = A[ak + Dk];
      const float tt = anc ? Lki * dk : 0.0f;
#pragma unroll
      for (int c = 0; c < DMAX; c++)
        if (c < DMAX - s - 1) row[c] = anc ? fmaf(-tt, rk[c], row[c]) : row[c];
      diag = anc ? fmaf(-tt, Lki, diag) : diag;
    }
    ODK_SYNC();
  }
  // Schur complement of the chains on the base block: every chain dof still holds its finished row in registers
  // (scaled entries of the NBASE base columns, D_k), so entry (a, b) is a group sum of row[a] D row[b] over the chain
  // lanes -- 21 register reductions, no LDS reads; lane q = a (a + 1) / 2 + b applies entry q.
  {
    const bool chain = on && ch_len > 0;
    float mine = 0.0f;
    int q = 0;
#pragma unroll
    for (int a = 0; a < NBASE; a++) {
      const float la = chain ? row[a] * diag : 0.0f;
#pragma unroll
      for (int b2 = 0; b2 <= a; b2++) {
        const float sq = gsum<G>(la * row[b2]);
        mine = lane == q ? sq : mine;
        q++;
      }
    }
    const int ra = ubcast(madr_st, 0);   // base rows are stored first: entry q of the packed lower triangle is at ra + q
    ODK_SYNC();
    if (lane < (NBASE * (NBASE + 1)) / 2) A[ra + lane] -= mine;
  }
  ODK_SYNC();
  // base block: reload the updated rows and finish as in factor_rows
  if (lane < NBASE) {
#pragma unroll
    for (int c = 0; c < DMAX; c++) row[c] = A[ai + c];
    diag = A[ai + di];
  }
  for (int k = NBASE - 1; k > 0; k--) {
    const int Dk = ubcast(depth_st, k), ak = ubcast(madr_st, k);
    if (lane == k) {
      const float inv = __builtin_amdgcn_rcpf(diag);
#pragma unroll
      for (int c = 0; c < DMAX; c++)
        if (c < Dk) { row[c] *= inv; A[ak + c] = row[c]; }
      A[ak + Dk] = diag;
    }
    ODK_SYNC();
    {
      const bool anc = lane < k;
      float rk[NBASE];
#pragma unroll
      for (int c = 0; c < NBASE; c++) rk[c] = A[ak + c];
      const float Lki = A[ak + di], dk = A[ak + Dk];
      const float tt = anc ? Lki * dk : 0.0f;
#pragma unroll
      for (int c = 0; c < NBASE; c++) row[c] = anc ? fmaf(-tt, rk[c], row[c]) : row[c];
      diag = anc ? fmaf(-tt, Lki, diag) : diag;
    }
    ODK_SYNC();
  }
  if (lane == 0) A[ai + di] = diag;
  ODK_SYNC();
}

// x <- (L^T D L)^-1 x for the vector whose i-th element is held by lane i (returned the same way)
template <int G, int NVT>
__device__ __forceinline__ float solve_rows(const float* A, float xi, int lane, int on, int di, int ai, int ancmask, int descmask,
                                            int depth_st, int madr_st) {
  float Lcol[NVT], Lrow[NVT];
#pragma unroll
  for (int k = 0; k < NVT; k++) {
    const int ak = ubcast(madr_st, k), dk = ubcast(depth_st, k);
    const float lc = A[ak + di], lr = A[ai + dk];           // unconditional loads, masked below
    Lcol[k] = ((descmask >> k) & 1) ? lc : 0.0f;           // L(k, i), k below i
    Lrow[k] = ((ancmask >> k) & 1) ? lr : 0.0f;            // L(i, k), k above i
  }
  const float dg = A[ai + di];
  const float dinv = on ? __builtin_amdgcn_rcpf(dg) : 0.0f;
#pragma unroll
  for (int k = NVT - 1; k > 0; k--) xi -= Lcol[k] * bcast<G>(xi, k);
  xi *= dinv;
#pragma unroll
  for (int j = 0; j < NVT - 1; j++) xi -= Lrow[j] * bcast<G>(xi, j);
  return xi;
}

// ------------------------------------------------------------------------------------------------
// Solve of (L^T D L) x = b for a "tree of chains" (floating base + up to three serial chains), the shape of this
// robot.  The matrices are tiny (6 + 5 + 4 + 5): eliminating lane-parallel by dof is bound by ~19 dependent hand-offs, and
// one lane per chain doing its whole block in registers (the previous version) issues ~600 VALU instructions per solve for
// three busy lanes -- and the kernel is bound by VALU issue.  Now the block of chain c, [T | C | y] with T the chain's own
// 5 x 5 part, C its 5 x 6 coupling to the base dofs and y its right-hand side, is eliminated by COLUMN: lane 8 c + b holds
// column b of [C | y] (b = 6: y) and a private copy of T, so the seven columns of the three chains go through the same ~50
// instructions at once.  Then 27 lanes form the base block's Schur complement and right-hand side (one entry each, 15
// FMAs), every lane factors and solves the 6 x 6 base system (redundantly: no hand-off), and the y-lanes back-substitute.
// Three LDS hand-offs.  A is the tree layout (read only); VEC holds b on entry and x on exit (LDS); XCH: 132 floats and
// XSV: 105 floats of LDS scratch.  cs = Statics::cs_pk.
template <class S, int G>
__device__ __forceinline__ void chain_solve(const float* A, float* VEC, float* XCH, float* XSV, int cs, int lane) {
  constexpr int CL = S::CL > 0 ? S::CL : 1;
  constexpr int NB6 = 6, P = 7, BASE = 3 * CL * P;   // P: pitch of one (chain, row) record; BASE: base block behind the records
  const int len = cs & 7, d0 = (cs >> 3) & 31, first = (cs >> 8) & 63, madr = (cs >> 14) & 1023;
  const int c = (lane >> 3) < 3 ? (lane >> 3) : 0, bcol = lane & 7;
  const bool col = lane < 24 && bcol < P;   // lanes that hold a column (all zeros when the chain does not exist: len = 0)
  float T[CL][CL], v[CL], dinv[CL];
  // ---- load T and this lane's column (rows beyond the chain's length are identity padding)
  // (All reads first, unconditionally, and PINNED: left alone the optimiser turns every `valid ? read : constant` into a read under its own exec mask -- twenty blocks of
  // `v_mov default; s_and_saveexec; ds_read; s_or exec` with branches between them, twice per substep: round 6.)
  float tr[CL][CL], cvr[CL];
#pragma unroll
  for (int p = 0; p < CL; p++) {
    const bool valid = p < len;
    const int radr = valid ? madr + p * (d0 + 1) + (p * (p - 1)) / 2 : 0;
#pragma unroll
    for (int q = 0; q <= p; q++) tr[p][q] = A[radr + d0 + q];
    const float* src = bcol < NB6 ? A + radr + bcol : VEC + (valid ? first + p : 0);
    cvr[p] = *src;
  }
  static_assert(CL <= 6, "pins below");
#pragma unroll
  for (int p = 0; p < CL; p++) {
    switch (p) {
      case 0: asm volatile("" :: "v"(tr[0][0]), "v"(cvr[0])); break;
      case 1: asm volatile("" :: "v"(tr[1][0]), "v"(tr[1][1]), "v"(cvr[1])); break;
      case 2: asm volatile("" :: "v"(tr[2][0]), "v"(tr[2][1]), "v"(tr[2][2]), "v"(cvr[2])); break;
      case 3: asm volatile("" :: "v"(tr[3][0]), "v"(tr[3][1]), "v"(tr[3][2]), "v"(tr[3][3]), "v"(cvr[3])); break;
      case 4: asm volatile("" :: "v"(tr[4][0]), "v"(tr[4][1]), "v"(tr[4][2]), "v"(tr[4][3]), "v"(tr[4][4]), "v"(cvr[4])); break;
      default: asm volatile("" :: "v"(tr[CL - 1][0]), "v"(tr[CL - 1][1]), "v"(tr[CL - 1][2]), "v"(tr[CL - 1][3]), "v"(tr[CL - 1][4]), "v"(tr[CL - 1][CL - 1]), "v"(cvr[CL - 1])); break;
    }
  }
#pragma unroll
  for (int p = 0; p < CL; p++) {
    const bool valid = p < len;
#pragma unroll
    for (int q = 0; q <= p; q++) T[p][q] = valid ? tr[p][q] : (q == p ? 1.0f : 0.0f);
    v[p] = valid ? cvr[p] : 0.0f;
  }
  // ---- eliminate the chain from its leaf: L entries replace T, the column becomes L^-T (column)
#pragma unroll
  for (int k = CL - 1; k >= 0; k--) {
    const float inv = __builtin_amdgcn_rcpf(T[k][k]);
    dinv[k] = inv;
#pragma unroll
    for (int i = k - 1; i >= 0; i--) {   // towards the root: row k's entries j <= i are still unscaled
      const float l = T[k][i] * inv;
#pragma unroll
      for (int j = 0; j <= i; j++) T[i][j] = fmaf(-l, T[k][j], T[i][j]);
      v[i] = fmaf(-l, v[k], v[i]);
      T[k][i] = l;
    }
  }
  float sv[CL];
#pragma unroll
  for (int k = 0; k < CL; k++) sv[k] = v[k] * dinv[k];
  if (col) {
#pragma unroll
    for (int k = 0; k < CL; k++) { XCH[(c * CL + k) * P + bcol] = v[k]; XSV[(c * CL + k) * P + bcol] = sv[k]; }
  }
  ODK_SYNC();
  // ---- base block: entry (tb, tb2) of B - sum_chains C^T T^-1 C (lanes 0-20, lower triangle) and component tb of
  // b_base - sum_chains C^T T^-1 y (lanes 21-26: tb2 = 6, the y column)
  {
    const int tb = (cs >> 24) & 7, tb2 = (cs >> 27) & 7;
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 3 * CL; r++) sum = fmaf(XSV[r * P + tb], XCH[r * P + tb2], sum);
    const float* src = lane < 21 ? A + lane : VEC + (lane < 27 ? lane - 21 : 0);
    const float a0 = *src;
    if (lane < 27) XCH[BASE + lane] = a0 - sum;
  }
  ODK_SYNC();
  // ---- base 6x6, in every lane: factor, solve
  float xb[NB6];
  {
    float Bm[NB6][NB6], bb[NB6], binv[NB6];
    {
      int q = 0;
#pragma unroll
      for (int b = 0; b < NB6; b++) {
#pragma unroll
        for (int b2 = 0; b2 <= b; b2++) Bm[b][b2] = XCH[BASE + q++];
        bb[b] = XCH[BASE + 21 + b];
      }
    }
#pragma unroll
    for (int k = NB6 - 1; k >= 0; k--) {
      const float inv = __builtin_amdgcn_rcpf(Bm[k][k]);
      binv[k] = inv;
#pragma unroll
      for (int i = k - 1; i >= 0; i--) {
        const float l = Bm[k][i] * inv;
#pragma unroll
        for (int j = 0; j <= i; j++) Bm[i][j] = fmaf(-l, Bm[k][j], Bm[i][j]);
        bb[i] = fmaf(-l, bb[k], bb[i]);
        Bm[k][i] = l;
      }
    }
#pragma unroll
    for (int k = 0; k < NB6; k++) {
      float x = bb[k] * binv[k];
#pragma unroll
      for (int i = 0; i < k; i++) x = fmaf(-Bm[k][i], xb[i], x);
      xb[k] = x;
    }
    if (lane == 0) {   // (b[0..5] was read before the hand-off above)
#pragma unroll
      for (int b = 0; b < NB6; b++) VEC[b] = xb[b];
    }
  }
  // ---- the y-lanes back-substitute their chain with the base solution
  {
    float x[CL];
    const bool ycol = lane < 24 && bcol == NB6;
#pragma unroll
    for (int k = 0; k < CL; k++) {
      float val = sv[k];
#pragma unroll
      for (int b = 0; b < NB6; b++) val = fmaf(-XSV[(c * CL + k) * P + b], xb[b], val);
#pragma unroll
      for (int i = 0; i < k; i++) val = fmaf(-T[k][i], x[i], val);
      x[k] = val;
      if (ycol && k < len) VEC[first + k] = val;
    }
  }
  ODK_SYNC();
}

// Twin dofs (DevModel::paired): solving (P Hr P^T + diag(d)) x = g, with P copying each reduced column onto its pair.
// With z = P^T x (one entry per reduced dof), E = P^T diag(1/d) P = diag(1/d_u + 1/d_v) and s = P^T diag(1/d) g:
//     (Hr + E^-1) z = E^-1 s      -- the twin-free robot's own system with the pair's series stiffness d_u d_v / (d_u + d_v)
//                                    on the diagonal (pair_einv) and the right-hand side pair_rhs,
//     x_i = (g_i - t_r) / d_i,  t = E^-1 s - E^-1 z     (unpaired dofs: x = z)   -- pair_expand.
// d_i = ARM[i] (+ DX[i]: the Newton Hessian's diagonal rows); the twin of main dof u is u + 1 (build_reduced_tables).
__device__ __forceinline__ void pair_terms(const float* GV, const float* ARM, const float* DX, int u, float& rhs, float& einv, float& du, float& dv) {
  du = ARM[u] + (DX ? DX[u] : 0.0f); dv = ARM[u + 1] + (DX ? DX[u + 1] : 0.0f);
  const float inv = __builtin_amdgcn_rcpf(du + dv);
  rhs = (GV[u] * dv + GV[u + 1] * du) * inv;
  einv = du * dv * inv;
}
__device__ __forceinline__ float pair_einv(const float* ARM, const float* DX, int u) {
  const float du = ARM[u] + (DX ? DX[u] : 0.0f), dv = ARM[u + 1] + (DX ? DX[u + 1] : 0.0f);
  return du * dv * __builtin_amdgcn_rcpf(du + dv);
}
// right-hand side of the reduced system for reduced dof `lane` (GV: g per dof, LDS)
template <class S>
__device__ __forceinline__ float pair_rhs(const float* GV, const float* ARM, const float* DX, const DevModel* __restrict__ m, int lane) {
  float rhs = 0.0f;
  if (lane < S::NVR) {
    const int u = m->red_main[lane];
    rhs = GV[u];
    if (m->red_twin[lane] >= 0) { float einv, du, dv; pair_terms(GV, ARM, DX, u, rhs, einv, du, dv); }
  }
  return rhs;
}
// x of this lane's dof from the reduced solution Z (LDS, one entry per reduced dof); gi = g of this dof
template <class S, int G>
__device__ __forceinline__ float pair_expand(const float* Z, const float* GV, float gi, const float* ARM, const float* DX, const Statics<S, G>& st, int lane) {
  float x = 0.0f;
  if (st.d_on) {
    const float z = Z[st.d_red];
    x = z;
    if (st.d_tkind != 0) {
      float rhs, einv, du, dv;
      pair_terms(GV, ARM, DX, st.d_tkind == 1 ? lane : lane - 1, rhs, einv, du, dv);
      x = (gi - (rhs - einv * z)) * __builtin_amdgcn_rcpf(st.d_tkind == 1 ? du : dv);
    }
  }
  return x;
}
// P^T v for reduced dof `lane`: v_u (+ v_twin)
template <class S>
__device__ __forceinline__ float pair_sum(const float* V, const DevModel* __restrict__ m, int lane) {
  float r = 0.0f;
  if (lane < S::NVR) {
    const int u = m->red_main[lane];
    r = V[u];
    if (m->red_twin[lane] >= 0) r += V[u + 1];
  }
  return r;
}

// impedance / stiffness of one constraint row (mjx constraint._row); returns D = 1/R and aref
// efc_D and efc_aref of one active row from the constants packed at load (DevModel::*_imp): impedance sigmoid of
// |pos| / width, R = invweight (1 - imp) / imp, aref = -b vel - k imp pos  (mjx constraint._efc_row / mju_makeImpedance)
// a row of a vector constraint: the impedance from pos_imp (the norm of the constraint's residual), the reference from the row's own pos
__device__ __forceinline__ void row_params_imp(const float* P, float pos, float pos_imp, float invweight, float vel, float& D, float& aref) {
  const float k = P[0], b = P[1], dmin = P[2], dmax = P[3], mid = P[5], power = P[6];
  const float x = fabsf(pos_imp) * P[4];
  float y;
  if (power == 2.0f) y = x < mid ? x * x * P[7] : 1.0f - (1.0f - x) * (1.0f - x) * P[8];
  else y = x < mid ? powf(x, power) * P[7] : 1.0f - powf(1.0f - x, power) * P[8];
  float imp = fminf(fmaxf(dmin + y * (dmax - dmin), dmin), dmax);
  if (x > 1.0f) imp = dmax;
  D = imp * __builtin_amdgcn_rcpf(fmaxf(invweight * (1.0f - imp), MINVAL_F * imp));
  aref = -b * vel - k * imp * pos;
}
__device__ __forceinline__ void row_params(const float* P, float pos, float invweight, float vel, float& D, float& aref) {
  const float k = P[0], b = P[1], dmin = P[2], dmax = P[3], mid = P[5], power = P[6];
  const float x = fabsf(pos) * P[4];
  float y;
  if (power == 2.0f) {
    y = x < mid ? x * x * P[7] : 1.0f - (1.0f - x) * (1.0f - x) * P[8];
  } else {
    y = x < mid ? powf(x, power) * P[7] : 1.0f - powf(1.0f - x, power) * P[8];
  }
  float imp = fminf(fmaxf(dmin + y * (dmax - dmin), dmin), dmax);
  if (x > 1.0f) imp = dmax;
  D = imp * __builtin_amdgcn_rcpf(fmaxf(invweight * (1.0f - imp), MINVAL_F * imp));   // = 1 / max(invweight (1 - imp) / imp, MINVAL); v_rcp_f32 (1 ulp), not the ten-instruction IEEE quotient
  aref = -b * vel - k * imp * pos;
}

// ------------------------------------------------------------------------------------------------
// One mjx.forward for one env (all G lanes of the group call this together).
//   flags bit0: compute sensordata / debug outputs (last substep only)
// mjx collision_convex._manifold_points: 4 support points of approximately maximal area among the vertices within
// 1e-3 of the deepest one (lane = vertex; `n` = contact normal)
// area measures below 1e-7 m^2 count as zero (oracle manifold_points AREA0: collinear / coincident candidates tie like in exact
// arithmetic instead of by rounding residue)
__device__ __forceinline__ float area0(float v) { return v < 1e-7f ? 0.0f : v; }
// last manifold point: values within AREA_TIE of the maximum count as the maximum, the lowest index wins (oracle manifold_points)
constexpr float AREA_TIE = 1e-7f;
template <int G>
__device__ __forceinline__ void select4(const float* w, bool has, float sup, int nvt, const float* n, int* idx, int lane) {
  const float smax = gmax<G>(sup);
  const float thr = fmaxf(0.0f, smax - 1e-3f);
  const float dm = has ? ((sup > thr) ? 0.0f : -1e6f) : -3.0e38f;
  idx[0] = gargmax<G>(dm, lane);
  float a[3] = {__shfl(w[0], idx[0], G), __shfl(w[1], idx[0], G), __shfl(w[2], idx[0], G)};
  float ap[3] = {a[0] - w[0], a[1] - w[1], a[2] - w[2]};
  idx[1] = gargmax<G>(has ? dot3(ap, ap) + dm : -3.0e38f, lane);
  float bq[3] = {__shfl(w[0], idx[1], G), __shfl(w[1], idx[1], G), __shfl(w[2], idx[1], G)};
  float amb[3] = {a[0] - bq[0], a[1] - bq[1], a[2] - bq[2]}, ab[3];
  cross3(ab, n, amb);
  idx[2] = gargmax<G>(has ? area0(fabsf(dot3(ap, ab))) + dm : -3.0e38f, lane);
  float cq[3] = {__shfl(w[0], idx[2], G), __shfl(w[1], idx[2], G), __shfl(w[2], idx[2], G)};
  float amc[3] = {a[0] - cq[0], a[1] - cq[1], a[2] - cq[2]}, bmc[3] = {bq[0] - cq[0], bq[1] - cq[1], bq[2] - cq[2]}, ac[3], bc[3];
  cross3(ac, n, amc);
  cross3(bc, n, bmc);
  float bp[3] = {bq[0] - w[0], bq[1] - w[1], bq[2] - w[2]};
  const float v1 = area0(fabsf(dot3(bp, bc))) + dm, v2 = area0(fabsf(dot3(ap, ac))) + dm;
  const float M = gmax<G>(has ? fmaxf(v1, v2) : -3.0e38f) - AREA_TIE;
  const unsigned id = !has ? 0x7FFFFFFFu : (v1 >= M ? (unsigned)lane : (v2 >= M ? (unsigned)(nvt + lane) : 0x7FFFFFFFu));
  const unsigned best = greduce_u<G, false>(id);
  idx[3] = best == 0x7FFFFFFFu ? 0 : (int)(best >= (unsigned)nvt ? best - (unsigned)nvt : best);
}

// 16-lane row reductions on unsigned keys (the first four butterfly stages of greduce_u: every lane ends with its row's value)
template <bool MAX> __device__ __forceinline__ unsigned rreduce_u(unsigned v) {
  auto op = [](unsigned a, unsigned b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = op(v, ODK_DPPU(v, 0xB1));
  v = op(v, ODK_DPPU(v, 0x4E));
  v = op(v, ODK_DPPU(v, 0x141));
  v = op(v, ODK_DPPU(v, 0x140));
  return v;
}
// _manifold_points for BOTH feet at once: foot f lives in the 16-lane row f (lane = 16 f + vertex, vertices 0..15) and
// every lane of the row also carries the hull's 17th vertex (index 16) itself, so all arg-max steps are row-local DPP
// reductions and the two feet share every instruction.  we / supe: the extra vertex (has_e: the hull has one).
__device__ __forceinline__ void select4_rows(const float* w, bool has, float sup, const float* we, bool has_e, float supe, int nvt,
                                             const float* n, int* idx, int lane) {
  const int j = lane & 15, rowbase = lane & ~15;
  auto amax = [&](float v, int i, float ve, int ie) -> int {   // lowest index among the maxima of {(v, i) over the row} + (ve, ie)
    const unsigned k = fkey(v), ke = fkey(ve);
    const unsigned rm = rreduce_u<true>(k);
    const unsigned mx = rm > ke ? rm : ke;
    const unsigned ri = rreduce_u<false>(k == mx ? (unsigned)i : 0x7FFFFFFFu);
    const unsigned ei = ke == mx ? (unsigned)ie : 0x7FFFFFFFu;
    return (int)(ri < ei ? ri : ei);
  };
  auto fetch = [&](int id, float* o) {   // position of vertex id of this row's hull
    const int src = rowbase + (id & 15);
#pragma unroll
    for (int k = 0; k < 3; k++) { const float t = __shfl(w[k], src, 32); o[k] = id == 16 ? we[k] : t; }   // src: relative to the env's 32 lanes
  };
  const unsigned ksm = rreduce_u<true>(fkey(sup)), kse = fkey(supe);
  const float smax = fkey_inv(ksm > kse ? ksm : kse);
  const float thr = fmaxf(0.0f, smax - 1e-3f);
  const float dm = has ? ((sup > thr) ? 0.0f : -1e6f) : -3.0e38f;
  const float dme = has_e ? ((supe > thr) ? 0.0f : -1e6f) : -3.0e38f;
  idx[0] = amax(dm, j, dme, 16);
  float a[3];
  fetch(idx[0], a);
  const float ap[3] = {a[0] - w[0], a[1] - w[1], a[2] - w[2]}, ape[3] = {a[0] - we[0], a[1] - we[1], a[2] - we[2]};
  idx[1] = amax(has ? dot3(ap, ap) + dm : -3.0e38f, j, has_e ? dot3(ape, ape) + dme : -3.0e38f, 16);
  float bq[3];
  fetch(idx[1], bq);
  const float amb[3] = {a[0] - bq[0], a[1] - bq[1], a[2] - bq[2]};
  float ab[3];
  cross3(ab, n, amb);
  idx[2] = amax(has ? area0(fabsf(dot3(ap, ab))) + dm : -3.0e38f, j, has_e ? area0(fabsf(dot3(ape, ab))) + dme : -3.0e38f, 16);
  float cq[3];
  fetch(idx[2], cq);
  const float amc[3] = {a[0] - cq[0], a[1] - cq[1], a[2] - cq[2]}, bmc[3] = {bq[0] - cq[0], bq[1] - cq[1], bq[2] - cq[2]};
  float ac[3], bc[3];
  cross3(ac, n, amc);
  cross3(bc, n, bmc);
  auto last = [&](const float* x, const float* apx, float dmx, float& v1, float& v2) {
    const float bp[3] = {bq[0] - x[0], bq[1] - x[1], bq[2] - x[2]};
    v1 = area0(fabsf(dot3(bp, bc))) + dmx; v2 = area0(fabsf(dot3(apx, ac))) + dmx;
  };
  float v1, v2, v1e, v2e;
  last(w, ap, dm, v1, v2);
  last(we, ape, dme, v1e, v2e);
  const unsigned km = rreduce_u<true>(fkey(has ? fmaxf(v1, v2) : -3.0e38f)), kme = fkey(has_e ? fmaxf(v1e, v2e) : -3.0e38f);
  const float M = fkey_inv(km > kme ? km : kme) - AREA_TIE;
  const unsigned id = !has ? 0x7FFFFFFFu : (v1 >= M ? (unsigned)j : (v2 >= M ? (unsigned)(nvt + j) : 0x7FFFFFFFu));
  const unsigned ide = !has_e ? 0x7FFFFFFFu : (v1e >= M ? 16u : (v2e >= M ? (unsigned)(nvt + 16) : 0x7FFFFFFFu));
  const unsigned ri = rreduce_u<false>(id);
  const unsigned best = ri < ide ? ri : ide;
  idx[3] = best == 0x7FFFFFFFu ? 0 : (int)(best >= (unsigned)nvt ? best - (unsigned)nvt : best);
}

}  // namespace odk
// Timing experiment (make libodk_knock.so, tools/gpu_hf_knock.sh): bits 8.. of the batch's filter word switch parts of the routine
// off (bits 0-6: the results are WRONG) or run them twice (bits 8-16: same results, the launch grows by that part's cost).
#ifdef ODK_HF_KNOCK
#define HF_KNOCK(bit) ((m->hfield_filter >> (8 + (bit))) & 1)
#define HF_FILTER(m) ((m)->hfield_filter & 255)
#define HF_REP(bit) for (int _rep = 0; _rep < 1 + ((knock >> (bit)) & 1); _rep++)
#define HF_TOUCH(x) asm volatile("" : "+v"(x))
#define HF_REP_SYNC() ODK_SYNC()
#else
#define HF_KNOCK(bit) 0
#define HF_FILTER(m) ((m)->hfield_filter)
#define HF_REP(bit)
#define HF_TOUCH(x) do { } while (0)
#define HF_REP_SYNC() do { } while (0)
#endif
#include "odk_convex.h"
namespace odk {

// Foot-foot (mesh-mesh) contacts: mjx convex_convex on the two hulls in the world frame (odk_convex.h), worked on by the first
// 16-lane row of every env whose boxes overlap; the other rows (and envs whose boxes are separated: `overlap` false) run along
// with their writes gated off.  Called by every lane of the wave (wave-uniform branch at the caller).  Rare path, out of line:
// it must not take part in the hot path's register allocation.
// LDS (all dead between the inertia phase and the constraint rows): world vertices / face normals of both feet in the row
// region D | aref | jar | jv, the first hull's edge data in cfrc | crb, polygons + this pair's contacts in BUF6.
template <class S, int G>
__device__ __noinline__ void foot_foot_sat(float* L, const DevModel* __restrict__ m, int lane, bool overlap) {
  constexpr int NB = S::NB;
  static_assert(4 * S::NROW >= 282 && (16 * S::NB >= 6 * 48 || !S::W_FITS) && 6 * S::NVR >= 56, "foot-foot scratch does not fit");
  float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR; float* SCR = L + S::O_SCR;
  const float* XPOS = L + S::O_XPOS; const float* XQUAT = L + S::O_XQUAT; const float* QPOS = L + S::O_QPOS;
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  float* FVb = L + S::O_D;       // [2][17][3] vertices | [2][30][3] normals
  float* AE = L + S::O_W;        // [<= 48][6]: cfrc | crb (dead here), or the wrench floats of a robot with fewer than 18 bodies (born later: P8)
  float* RS = L + S::O_BUF6;     // RP 12 | IP 12 | NEW 32
  const int row = lane >> 4, j = lane & 15;
  const bool act = overlap && row == 0;
  float cw[2][3];
#pragma unroll
  for (int f = 0; f < 2; f++) {
    const int fb = m->foot_body[f];
    float q[4], R[9], P[3];
    for (int k = 0; k < 4; k++) q[k] = XQUAT[k * NB + fb];
    for (int k = 0; k < 3; k++) P[k] = XPOS[k * NB + fb] - ref[k];   // relative to the base: a robot metres from the origin keeps float32 digits for the hull
    q2mat(R, q);
    for (int v = lane; v < m->foot_nvert[f]; v += G) {
      const float vb[3] = {m->foot_vert[f][v][0], m->foot_vert[f][v][1], m->foot_vert[f][v][2]};
      for (int k = 0; k < 3; k++) FVb[f * 51 + 3 * v + k] = P[k] + R[3 * k] * vb[0] + R[3 * k + 1] * vb[1] + R[3 * k + 2] * vb[2];
    }
    for (int t = lane; t < m->foot_npoly[f]; t += G) {
      const float nb[3] = {m->foot_fnorm[f][t][0], m->foot_fnorm[f][t][1], m->foot_fnorm[f][t][2]};
      for (int k = 0; k < 3; k++) FVb[102 + f * 90 + 3 * t + k] = R[3 * k] * nb[0] + R[3 * k + 1] * nb[1] + R[3 * k + 2] * nb[2];
    }
    for (int k = 0; k < 3; k++) cw[f][k] = P[k] + R[3 * k] * m->foot_centroid[f][0] + R[3 * k + 1] * m->foot_centroid[f][1] + R[3 * k + 2] * m->foot_centroid[f][2];
  }
  ODK_SYNC();
  Cvx A = {FVb, FVb + 102, &m->foot_poly[0][0][0], &m->foot_edge[0][0][0], m->foot_nvert[0], m->foot_npoly[0], m->foot_nedge[0], {cw[0][0], cw[0][1], cw[0][2]}};
  Cvx B = {FVb + 51, FVb + 192, &m->foot_poly[1][0][0], &m->foot_edge[1][0][0], m->foot_nvert[1], m->foot_npoly[1], m->foot_nedge[1], {cw[1][0], cw[1][1], cw[1][2]}};
  EdgeRegs<3> RB;
  edge_regs_load<3>(RB, B, j);
  edge_prepare_row(A, AE, j, row == 0);
  ODK_SYNC();
  RowScratch RSS = {RS, RS + 12, RS + 24, nullptr, nullptr};
  sat_pair_row<3>(A, B, AE, RB, RSS, j, row == 0);
  if (act && j < 4) {
    const float* o = RS + 24 + 8 * j;
    const int c = 8 + j;
    CDIST[c] = o[0];
    for (int t = 0; t < 3; t++) CR[3 * c + t] = o[1 + t];
    if (j == 0) make_frame_dev(o + 4, SCR + S::S_VF);   // the pair's contact frame (one normal for all four), consumed by P8
  }
}

// Height-field floor (scene_rough_terrain_backlash.xml:22): mjx hfield_convex -- the prisms of the cells under the foot's bounding
// sphere, convex_convex per prism, the four deepest contacts kept, each with the normal of its own prism test (odk_convex.h).
// Foot f = 16-lane row f of the env.  A prism whose own faces already separate it from the foot (face query of the prism
// against the hull's vertices > 0) cannot contribute an active contact (MJX reports dist > 0 for it) and is dropped before the
// pair loop; the loop runs over each row's surviving prisms, as many iterations as the longest list in the wave.
// The prism of an iteration lives in registers (compile-time topology: sat_prism_row); the hull's faces and edges of a lane stay in
// registers over the whole loop.  LDS: hull vertices / face normals in the height field's frame in cfrc | crb; per row the prism's
// vertices + the two polygons of the face contact in BUF6 / BUF6B; per row prism list, running best four, current four in
// D | aref | jar | jv; at the end the eight contact frames go to jv ([8][9], read by the constraint-row phase).
__constant__ HfAssign ODK_HF_ASSIGN = make_hf_assign();
// index into that table from the four rows' open-entry counts (one byte each), each capped at four
__host__ __device__ __forceinline__ int hf_assign_index(unsigned n_pk) {
  const unsigned c0 = min(n_pk & 255u, 4u), c1 = min((n_pk >> 8) & 255u, 4u), c2 = min((n_pk >> 16) & 255u, 4u), c3 = min(n_pk >> 24, 4u);
  return (int)(c0 + 5u * (c1 + 5u * (c2 + 5u * c3)));
}
template <class S, int G>
__device__ __forceinline__ void hfield_contacts(float* L, const DevModel* __restrict__ m, const float* __restrict__ hf, int lane) {
  constexpr int NB = S::NB;
  static_assert(G == 32, "height-field floors run 32 lanes per env (two 16-lane rows = two feet)");
  static_assert(4 * S::NROW >= 344 && 16 * S::NB >= 288 && 6 * S::NVR >= 106 && S::NROW >= 72, "height-field scratch does not fit");
  float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR;
  const float* XPOS = L + S::O_XPOS; const float* XQUAT = L + S::O_XQUAT; const float* QPOS = L + S::O_QPOS;
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  const int f = (lane >> 4) & 1, j = lane & 15;
#ifdef ODK_HF_KNOCK
  const int knock = m->hfield_filter >> 8;
#else
  constexpr int knock = 0;
#endif
#ifdef ODK_PROFILE
  long long _hp = clock64();
#define HF_PROF(i) do { if (lane == 0) { const long long _t = clock64(); L[S::O_SCR + S::S_PROF2 + (i)] += (float)(_t - _hp); _hp = _t; } } while (0)
#define HF_COUNT(i, v) do { if (lane == 0) L[S::O_SCR + S::S_PROF2 + (i)] += (float)(v); } while (0)
#ifdef ODK_PROF_CULL      // (investigation build: the cull pass's own sub-phases in slots 8 .. 13, separate clock; the pair loop's sub-timers then mean nothing)
#define HF_CULL(i) do { if (lane == 0) { const long long _t = clock64(); L[S::O_SCR + S::S_PROF2 + 8 + (i)] += (float)(_t - _hc); _hc = _t; } } while (0)
#else
#define HF_CULL(i) do { } while (0)
#endif
#else
#define HF_CULL(i) do { } while (0)
#define HF_PROF(i) do { } while (0)
#define HF_COUNT(i, v) do { } while (0)
#endif
  float* FV = L + S::O_CFRC + f * 54; float* FN = L + S::O_CFRC + 108 + f * 90;      // [2][18][3] vertices (entries past a hull's last: copies of vertex 0) | [2][30][3] normals
  float* RS = L + (f ? S::O_BUF6B : S::O_BUF6);
  float* PV = RS;                                                      // prism vertices [6][3]
  // [<= 30][4] the hull's faces as (normal, plane offset) records for the cull pass's face loop: ONE 16-byte LDS read per face instead of four dwords
  // (round 6).  Foot 0: the H / factor region (dead between the solve for qacc_smooth and the Hessian's entries), foot 1: x | Ma | search | mv (dead
  // until the solver); both 16-byte aligned.
  float* FN4 = L + (f ? S::O_X : ((S::O_HL + 3) & ~3));
  static_assert(4 * S::NV >= 4 * HULL_MAXF && S::NHR >= 4 * HULL_MAXF + 3 && S::O_X % 4 == 0 && S::ENV_STRIDE % 4 == 0, "hull face records");
  float* META = RS + 106;                                              // for the other rows: ncw, hull centroid [3]  (the loop's bookkeeping -- open entries, entries taken -- lives in scalar registers)
  static_assert(6 * S::NVR >= 111, "row scratch + window record");
  float* RL = L + S::O_D + f * 172;
  float* LIST = RL; float* TOP = RL + 108; float* NEW = RL + 140;      // [18][6] | [4][8] | [4][8]
  const RowScratch RSS = {RS + 18, RS + 30, NEW, RS + 42, RS + 58};    // RP [4][3], IP [4][3], the row's list of passing edge pairs [64] (PW | VV)
  const float* Rh = m->floor_mat; const float ph[3] = {m->plane_pos[0], m->plane_pos[1], m->plane_pos[2]};
  const int nvt = m->foot_nvert[f], nfc = m->foot_npoly[f];
  // ---- the hull in the height field's frame: v_h = Rh^T (P + R v - ph)
  float Rw[9], Pw[3], cl[3];
  {
    const int fb = m->foot_body[f];
    float q[4], R[9], P[3];
    for (int k = 0; k < 4; k++) q[k] = XQUAT[k * NB + fb];
    for (int k = 0; k < 3; k++) P[k] = XPOS[k * NB + fb] - ph[k];
    q2mat(R, q);
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) Rw[3 * a + b] = Rh[a] * R[b] + Rh[3 + a] * R[3 + b] + Rh[6 + a] * R[6 + b];
      Pw[a] = Rh[a] * P[0] + Rh[3 + a] * P[1] + Rh[6 + a] * P[2];
    }
  }
  for (int k = 0; k < 3; k++) cl[k] = Pw[k] + Rw[3 * k] * m->foot_obb_center[f][0] + Rw[3 * k + 1] * m->foot_obb_center[f][1] + Rw[3 * k + 2] * m->foot_obb_center[f][2];
  // ---- cells under the bounding sphere
  const int nc = m->hfield_ncol, nr = m->hfield_nrow;
  const float sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2], base = m->hfield_size[3];
  const float dx = 2.0f * sx / (float)(nc - 1), dy = 2.0f * sy / (float)(nr - 1);
  // MJX takes the cells under the bounding sphere; every prism outside the hull's own x / y extent is separated from it by one of its
  // axis-aligned or diagonal side faces (dist > 0: no force), so the window is the extent of the hull's oriented box (<= the sphere's)
  float ex = 0.0f, ey = 0.0f;
  for (int a = 0; a < 3; a++) {
    const float* ax = m->foot_obb_axes[f];
    ex += fabsf(Rw[0] * ax[a] + Rw[1] * ax[3 + a] + Rw[2] * ax[6 + a]) * m->foot_obb_half[f][a];
    ey += fabsf(Rw[3] * ax[a] + Rw[4] * ax[3 + a] + Rw[5] * ax[6 + a]) * m->foot_obb_half[f][a];
  }
  int cmin = (int)floorf((cl[0] - ex + sx) / dx), cmax = (int)floorf((cl[0] + ex + sx) / dx);
  int rmin = (int)floorf((cl[1] - ey + sy) / dy), rmax = (int)floorf((cl[1] + ey + sy) / dy);
  cmin = cmin < 0 ? 0 : cmin; rmin = rmin < 0 ? 0 : rmin; cmax = cmax > nc - 2 ? nc - 2 : cmax; rmax = rmax > nr - 2 ? nr - 2 : rmax;
  int ncw = cmax - cmin + 1, nrw = rmax - rmin + 1;
  ncw = ncw > 3 ? 3 : ncw; nrw = nrw > 3 ? 3 : nrw;   // the sphere (radius < a cell) spans at most 3 cells per axis
  // Everything below works relative to the window's first grid corner: the terrain spans +-10 m, a contact depth is a fraction of
  // a millimetre, and float32 coordinates in the height field's own frame would carry ~1e-6 m of rounding each.
  const float org[2] = {-sx + (float)cmin * dx, -sy + (float)rmin * dy};
  Pw[0] -= org[0]; Pw[1] -= org[1];
  // this lane's share of the hull in the body frame (vertices / faces j and j + 16; past the last: vertex 0 / face 0), all of it asked for at once
  float hb_v[2][3], hb_n[2][3], hb_off[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int vs = j + 16 * u < nvt ? j + 16 * u : 0, ts = j + 16 * u < nfc ? j + 16 * u : 0;
    for (int k = 0; k < 3; k++) { hb_v[u][k] = m->foot_vert[f][vs][k]; hb_n[u][k] = m->foot_fnorm[f][ts][k]; }
    hb_off[u] = m->foot_foff[f][ts];
  }
  HF_REP(8) {
  HF_TOUCH(Pw[0]);
  // (the cull pass's loops run unmasked over 18 vertices / the faces in fives: the entries past a hull's last are copies of vertex 0 -- no change to
  // a minimum -- and face records that separate nothing)
  // A lane's two vertices (j, j + 16) and two faces; their body-frame data were read at the top of the routine (hb_*), all at once: as two rolled loops each
  // element's reads waited for their own round trip to the L2 -- eight exposed latencies per foot and substep (round 6).
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int v = j + 16 * u, t = j + 16 * u;
    float vw[3], nw[3];
    for (int k = 0; k < 3; k++) {
      vw[k] = Pw[k] + Rw[3 * k] * hb_v[u][0] + Rw[3 * k + 1] * hb_v[u][1] + Rw[3 * k + 2] * hb_v[u][2];
      nw[k] = Rw[3 * k] * hb_n[u][0] + Rw[3 * k + 1] * hb_n[u][1] + Rw[3 * k + 2] * hb_n[u][2];
    }
    // the face's plane offset in the window frame: n . (P + R v) = n_b . v_b + n . P; past the hull's last face: a record that separates nothing
    const bool real = t < nfc;
    const float4 rec = make_float4(real ? nw[0] : 0.0f, real ? nw[1] : 0.0f, real ? nw[2] : 0.0f, real ? hb_off[u] + dot3(nw, Pw) : 3.0e38f);
    if (u == 0 || v < 18) { FV[3 * v] = vw[0]; FV[3 * v + 1] = vw[1]; FV[3 * v + 2] = vw[2]; }
    if (u == 0 || t < HULL_MAXF) { FN[3 * t] = nw[0]; FN[3 * t + 1] = nw[1]; FN[3 * t + 2] = nw[2]; *reinterpret_cast<float4*>(FN4 + 4 * t) = rec; }      // (normals past the last face are never read)
  }
  }
  float fc[3];
  for (int k = 0; k < 3; k++) fc[k] = Pw[k] + Rw[3 * k] * m->foot_centroid[f][0] + Rw[3 * k + 1] * m->foot_centroid[f][1] + Rw[3 * k + 2] * m->foot_centroid[f][2];
  if (j < 4) { float* o = TOP + 8 * j; o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f; o[5] = 0.0f; o[6] = 1.0f; o[7] = 1.0e9f + (float)j; }
  if (j == 0) { META[0] = __int_as_float(ncw); META[1] = fc[0]; META[2] = fc[1]; META[3] = fc[2]; }
  ODK_SYNC();
  const float idiag = 1.0f / sqrtf(dx * dx + dy * dy);
  // prism p = 2 (ri ncw + ci) + tri of this row's window: grid corners of its top triangle (counter-clockwise seen from above)
  auto corners = [&](int p, int ncw, int* cc, int* rr) {
    const int q = p >> 1, tri = p & 1;
    // q / ncw for q <= 8, ncw = 1 / 2 / 3 as a multiply and a shift (q 32 >> 5, q 16 >> 5, q 11 >> 5): the nested selects this replaces were compiled into
    // four basic blocks with exec-mask bookkeeping, once per pair-loop iteration and once per cull pass (round 6)
    const int ri = (q * (ncw == 1 ? 32 : (ncw == 2 ? 16 : 11))) >> 5;
    const int c = q - ri * ncw, r = ri;   // relative to (cmin, rmin)
    cc[0] = tri ? c + 1 : c; rr[0] = tri ? r + 1 : r; cc[1] = tri ? c : c + 1; rr[1] = tri ? r + 1 : r; cc[2] = tri ? c + 1 : c; rr[2] = tri ? r : r + 1;
  };
  auto prism = [&](int p, int ncw, const float* z, Prism& P) {
    int cc[3], rr[3];
    corners(p, ncw, cc, rr);
    for (int k = 0; k < 3; k++) { P.x[k] = (float)cc[k] * dx; P.y[k] = (float)rr[k] * dy; P.z[k] = z[k]; }
    P.base = base;
    const float e1[3] = {P.x[1] - P.x[0], P.y[1] - P.y[0], P.z[1] - P.z[0]}, e2[3] = {P.x[2] - P.x[0], P.y[2] - P.y[0], P.z[2] - P.z[0]};
    cross3(P.nt, e1, e2);
    const float inv = __builtin_amdgcn_rsqf(dot3(P.nt, P.nt));
    P.nt[0] *= inv; P.nt[1] *= inv; P.nt[2] *= inv;
    // side over the edge a -> b of the top triangle: (e.y, -e.x, 0) / |e|; the edges run along x, the diagonal, along y
    const float sg = (p & 1) ? -1.0f : 1.0f;
    P.ns[0][0] = 0.0f; P.ns[0][1] = -sg;
    P.ns[1][0] = sg * dy * idiag; P.ns[1][1] = sg * dx * idiag;
    P.ns[2][0] = -sg; P.ns[2][1] = 0.0f;
  };
  // Four of a prism's five faces are shared by every prism of the window up to an offset: the bottom (normal -z), the sides along x, along y and
  // along the cell diagonal.  The smallest signed distance of the hull's vertices to such a face is an EXTENT of the hull along that axis -- one
  // row reduction per foot (lane = vertex) instead of a loop over the vertices in every prism's lane (round 6; VERDICT r5 #2b).  Sign flips are exact,
  // so min over v of (sg n . v) = sg > 0 ? min (n . v) : -max (n . v).
  float e_xmin, e_xmax, e_ymin, e_ymax, e_wmin, e_wmax, e_zmax;
  {
    const float wa = dy * idiag, wb = dx * idiag;
    float xmn = 3.0e38f, xmx = -3.0e38f, ymn = 3.0e38f, ymx = -3.0e38f, wmn = 3.0e38f, wmx = -3.0e38f, zmx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int v = j + 16 * t;
      if (v < nvt) {
        const float x = FV[3 * v], y = FV[3 * v + 1], z = FV[3 * v + 2], w = fmaf(wa, x, wb * y);
        xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y); wmn = fminf(wmn, w); wmx = fmaxf(wmx, w); zmx = fmaxf(zmx, z);
      }
    }
    e_xmin = fkey_inv(rreduce_u<false>(fkey(xmn))); e_xmax = fkey_inv(rreduce_u<true>(fkey(xmx)));
    e_ymin = fkey_inv(rreduce_u<false>(fkey(ymn))); e_ymax = fkey_inv(rreduce_u<true>(fkey(ymx)));
    e_wmin = fkey_inv(rreduce_u<false>(fkey(wmn))); e_wmax = fkey_inv(rreduce_u<true>(fkey(wmx)));
    e_zmax = fkey_inv(rreduce_u<true>(fkey(zmx)));
  }
  HF_PROF(0);
  // ---- pass over the window, lane = prism: the prism's own face query against the hull vertices; survivors into the list
  int cnt = 0;
  const int nprism = (ncw > 0 && nrw > 0) ? 2 * ncw * nrw : 0;
  {      // (prisms 0 .. 15: one pass; a 3 x 3 window's last two are done below, lane = hull vertex / face)
    constexpr int pass = 0;
    const int p = j;
    const bool valid = p < nprism;
#if defined(ODK_PROFILE) && defined(ODK_PROF_CULL)
    long long _hc = clock64();
#endif
    float z[3] = {0.0f, 0.0f, 0.0f};
    if (valid) { int cc[3], rr[3]; corners(p, ncw, cc, rr); for (int k = 0; k < 3; k++) z[k] = hf[(rmin + rr[k]) * nc + cmin + cc[k]] * sz; }
    HF_CULL(0);      // heights
    Prism P;
    prism(valid ? p : 0, ncw, z, P);
    HF_CULL(1);      // prism
    // plane offsets n . v0 of the five faces (v0: vertex 0 / 3 / 0 / 1 / 2), then min over the hull's vertices of n . v - offset: the top face
    // in a loop over the vertices (batches of six: the LDS reads of a batch are in flight together), the other four from the hull's extents
    float d5[5], s5[5];
    d5[0] = P.nt[0] * P.x[0] + P.nt[1] * P.y[0] + P.nt[2] * P.z[0]; d5[1] = base;
    d5[2] = P.ns[0][0] * P.x[0] + P.ns[0][1] * P.y[0]; d5[3] = P.ns[1][0] * P.x[1] + P.ns[1][1] * P.y[1]; d5[4] = P.ns[2][0] * P.x[2] + P.ns[2][1] * P.y[2];
    {
      const bool up = !(p & 1);      // (sg = +1 for the first triangle of a cell)
      s5[1] = -e_zmax;
      s5[2] = up ? -e_ymax : e_ymin;      // normal (0, -sg)
      s5[3] = up ? e_wmin : -e_wmax;      // normal sg (dy, dx) / |diag|
      s5[4] = up ? -e_xmax : e_xmin;      // normal (-sg, 0)
    }
    const int nvu = max(m->foot_nvert[0], m->foot_nvert[1]);   // (wave-uniform trip count, as for the faces below)
    HF_REP(9) {
    HF_TOUCH(P.nt[0]);
    float stop = 3.0e38f;
#pragma unroll 1
    for (int q0 = 0; q0 < nvu; q0 += 6) {
      float vv[6][3];
#pragma unroll
      for (int t = 0; t < 6; t++) { const int q = q0 + t; vv[t][0] = FV[3 * q]; vv[t][1] = FV[3 * q + 1]; vv[t][2] = FV[3 * q + 2]; }      // (q <= 17: padded with copies of vertex 0)
#pragma unroll
      for (int t = 0; t < 6; t++) stop = fminf(stop, dot3(P.nt, vv[t]));
    }
    s5[0] = stop;
    }
    HF_CULL(2);      // vertex loop
    float sep = -3.0e38f; int face = 0;
#pragma unroll
    for (int fa = 0; fa < 5; fa++) { const float sv = s5[fa] - d5[fa]; if (sv > sep) { sep = sv; face = fa; } }
    // the hull's face query against this prism's six vertices: with the prism's own it bounds the pair's best separating axis,
    // hence every contact of the pair, from below -- a prism beside the foot (inside the window, outside the slab) is separated by
    // one of the hull's side faces and never enters the list; the others are ranked by the larger of the two bounds.  (The six
    // faces of the hull's oriented box instead: a third of the cost, 0.4 more pairs per foot in the loop -- no gain.)
    float sep_b = -3.0e38f;
    if (!HF_KNOCK(1)) {
      const float zb = -base;
      // (wave-uniform trip count: with this row's own face count the loop is divergent across the rows and costs three times as much)
      const int nfu = max(m->foot_npoly[0], m->foot_npoly[1]);
      HF_REP(10) {
      HF_TOUCH(z[0]);
#pragma unroll 1
      for (int t0 = 0; t0 < nfu; t0 += 10) {      // (batches of ten faces: ten 16-byte LDS reads in flight together, three waits per pass)
        float4 fnv[10];
#pragma unroll
        for (int u = 0; u < 10; u++) fnv[u] = *reinterpret_cast<const float4*>(FN4 + 4 * (t0 + u));      // (t0 + u <= 29: records past the hull's last separate nothing)
#pragma unroll
        for (int u = 0; u < 10; u++) {
          const float n0 = fnv[u].x, n1 = fnv[u].y, n2 = fnv[u].z, dd = fnv[u].w;
          const float h0 = n0 * P.x[0] + n1 * P.y[0], h1 = n0 * P.x[1] + n1 * P.y[1], h2 = n0 * P.x[2] + n1 * P.y[2];
          const float top = fminf(fminf(h0 + n2 * z[0], h1 + n2 * z[1]), h2 + n2 * z[2]), bot = fminf(fminf(h0, h1), h2) + n2 * zb;
          const float sv = fminf(top, bot) - dd;
          sep_b = fmaxf(sep_b, sv);
        }
      }
      }
    }
    HF_CULL(3);      // face loop
    const float bound = fmaxf(sep, sep_b);
    const bool keep = valid && !(bound > 0.0f);
    const unsigned rowmask = (unsigned)((__builtin_amdgcn_ballot_w64(keep) >> (threadIdx.x & 48u)) & 0xFFFFull);
    const int pos = cnt + __popc(rowmask & ((1u << j) - 1u));
    if (keep) { float* o = LIST + 6 * pos; o[0] = __int_as_float(p | (face << 8)); o[1] = z[0]; o[2] = z[1]; o[3] = z[2]; o[4] = sep; o[5] = bound; }
    cnt += __popc(rowmask);
    HF_CULL(4);      // list write
  }
  // Prisms 16 / 17 (a 3 x 3 window only): a second pass of the loops above would run them for two lanes of sixteen.  Here the prism is the ROW's
  // and the lanes are the hull's: lane = vertex for the top plane (row minimum), lane = two face records for the hull's query (row maximum) --
  // the same minima / maxima, ~150 instructions per prism instead of a ~1 000-instruction pass (round 6).
  if (__builtin_amdgcn_ballot_w64(nprism > 16) != 0 && !HF_KNOCK(0)) {
#pragma unroll 1
    for (int p = 16; p < 18; p++) {
      const bool valid = p < nprism;      // (row-uniform)
      float z[3] = {0.0f, 0.0f, 0.0f};
      if (valid) { int cc[3], rr[3]; corners(p, ncw, cc, rr); for (int k = 0; k < 3; k++) z[k] = hf[(rmin + rr[k]) * nc + cmin + cc[k]] * sz; }
      Prism P;
      prism(valid ? p : 0, ncw, z, P);
      float d5[5], s5[5];
      d5[0] = P.nt[0] * P.x[0] + P.nt[1] * P.y[0] + P.nt[2] * P.z[0]; d5[1] = base;
      d5[2] = P.ns[0][0] * P.x[0] + P.ns[0][1] * P.y[0]; d5[3] = P.ns[1][0] * P.x[1] + P.ns[1][1] * P.y[1]; d5[4] = P.ns[2][0] * P.x[2] + P.ns[2][1] * P.y[2];
      {
        const bool up = !(p & 1);
        s5[1] = -e_zmax; s5[2] = up ? -e_ymax : e_ymin; s5[3] = up ? e_wmin : -e_wmax; s5[4] = up ? -e_xmax : e_xmin;
        const int v1 = j < 2 ? j + 16 : 0;      // (18 vertex slots, the last ones copies of vertex 0)
        const float va[3] = {FV[3 * j], FV[3 * j + 1], FV[3 * j + 2]}, vb[3] = {FV[3 * v1], FV[3 * v1 + 1], FV[3 * v1 + 2]};
        s5[0] = fkey_inv(rreduce_u<false>(fkey(fminf(dot3(P.nt, va), dot3(P.nt, vb)))));
      }
      float sep = -3.0e38f; int face = 0;
#pragma unroll
      for (int fa = 0; fa < 5; fa++) { const float sv = s5[fa] - d5[fa]; if (sv > sep) { sep = sv; face = fa; } }
      float sb = -3.0e38f;
      {
        const float zb = -base;
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int t = j + 16 * u < HULL_MAXF ? j + 16 * u : 0;      // (30 records, those past the hull's last separate nothing)
          const float4 fr = *reinterpret_cast<const float4*>(FN4 + 4 * t);
          const float n0 = fr.x, n1 = fr.y, n2 = fr.z, dd = fr.w;
          const float h0 = n0 * P.x[0] + n1 * P.y[0], h1 = n0 * P.x[1] + n1 * P.y[1], h2 = n0 * P.x[2] + n1 * P.y[2];
          const float top = fminf(fminf(h0 + n2 * z[0], h1 + n2 * z[1]), h2 + n2 * z[2]), bot = fminf(fminf(h0, h1), h2) + n2 * zb;
          sb = fmaxf(sb, fminf(top, bot) - dd);
        }
      }
      const float sep_b = fkey_inv(rreduce_u<true>(fkey(sb)));
      const float bound = fmaxf(sep, sep_b);
      const bool keep = valid && !(bound > 0.0f);
      if (keep && j == 0) { float* o = LIST + 6 * cnt; o[0] = __int_as_float(p | (face << 8)); o[1] = z[0]; o[2] = z[1]; o[3] = z[2]; o[4] = sep; o[5] = bound; }
      cnt += keep ? 1 : 0;
    }
  }
  ODK_SYNC();
  {   // the list in ascending (sep, prism) order: an entry's place is the number of entries before it
    float e0[6], e1[6];
    {   // (entries 16 / 17 exist for lanes 0 / 1 only: read by every lane through a clamped index and pinned -- as `j < 2 ? read : 0` each of the six became a read under its own exec mask)
      const int j1 = j < 2 ? j + 16 : 0;
      for (int t = 0; t < 6; t++) { e0[t] = LIST[6 * j + t]; e1[t] = LIST[6 * j1 + t]; }
      asm volatile("" :: "v"(e1[0]), "v"(e1[1]), "v"(e1[2]), "v"(e1[3]), "v"(e1[4]), "v"(e1[5]));
      for (int t = 0; t < 6; t++) e1[t] = j < 2 ? e1[t] : 0.0f;
    }
    int rk0 = 0, rk1 = 0;
    // (trip count: the longest list of the wave's four rows -- three entries on average, not the list's capacity)
    const int cmax = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)), max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
#pragma unroll 1
    for (int q = 0; q < cmax; q++) {
      const float sq = LIST[6 * q + 5];
      const bool in = q < cnt;
      rk0 += (in && (sq < e0[5] || (sq == e0[5] && q < j))) ? 1 : 0;
      rk1 += (in && (sq < e1[5] || (sq == e1[5] && q < j + 16))) ? 1 : 0;
    }
    ODK_SYNC();
    if (j < cnt) for (int t = 0; t < 6; t++) LIST[6 * rk0 + t] = e0[t];
    if (j + 16 < cnt) for (int t = 0; t < 6; t++) LIST[6 * rk1 + t] = e1[t];
  }
  ODK_SYNC();
  HF_PROF(1);
  HF_COUNT(5, cnt);
  // ---- pair loop.  Every contact of a pair is at least as far out as the pair's best separating axis, which is at least the prism's
  // own face separation `sep` (kept in the list): a prism whose sep lies beyond the fourth-deepest contact found so far cannot enter
  // the best four.  So a foot's prisms are worked from the deepest sep on, and those that cannot matter any more are left out; the
  // result does not depend on the order (entries are ranked by (dist, candidate index), like a stable sort of MJX's whole candidate
  // list).  Typically three or four of the four to eight overlapping prisms of a foot get the full test.
  //
  // The four rows of the wave (two envs x two feet) share the work: a row whose own list has nothing open takes a prism of the
  // foot with the most open entries -- left to itself the loop runs as long as the longest of the four lists (6.0 iterations per
  // forward pass for 3.6 pairs per foot).  Per iteration: every row counts the open entries of its own list and publishes the
  // count; every lane derives the same assignment (row -> foot, rank among the rows on that foot) from the four counts; a row on a
  // new foot reloads that hull's faces / edges into its registers; the row takes the entry of rank `rank` among the open ones (by
  // sep, then index), marks it taken, works the pair into its own NEW block; the blocks are merged into the foot's TOP rank by rank.
  const int r_own = (threadIdx.x >> 4) & 3;
  float* Lw = L - ((threadIdx.x >> 5) & 1) * S::ENV_STRIDE;   // env 0 of the workgroup
  auto tgt_L = [&](int t) -> float* { return Lw + (t >> 1) * S::ENV_STRIDE; };
  auto tgt_meta = [&](int t) -> float* { return tgt_L(t) + ((t & 1) ? S::O_BUF6B : S::O_BUF6) + 106; };
  int tg = -1;                // the foot whose hull this row holds in registers
  int ncw_t = 1;
  Cvx B = {FV, FN, &m->foot_poly[f][0][0], &m->foot_edge[f][0][0], nvt, nfc, m->foot_nedge[f], {fc[0], fc[1], fc[2]}};
  EdgeRegs<3> RB;
  FaceRegs<2> FB;
  float* LISTt = LIST; float* TOPt = TOP;
  const float s0 = j < cnt ? LIST[6 * j + 5] : 3.0e38f, s1 = j + 16 < cnt ? LIST[6 * (j + 16) + 5] : 3.0e38f;   // own list's bounds (sorted)
  // Bookkeeping of the loop in SCALAR registers (round 6): which entries of the four lists have been taken (two 64-bit words: foot t's 18 bits at
  // 32 (t & 1) of word t >> 1), and what every row took in the iteration before (read from the rows' registers by v_readlane) -- no exchange through
  // LDS, no barrier, nothing of it on the vector pipe.
  unsigned long long taken01 = 0ull, taken23 = 0ull;
  int last_pk = 31;           // row-uniform: the entry this row took in the iteration before (foot << 8 | entry); none: entry 31 of foot 0, which no list has
  const bool up_only = HF_FILTER(m) == 3;   // (read once: every compiler barrier in the loop would fetch it again)
#ifdef ODK_PROFILE
  _hp = clock64();   // (slot 2 counts the passing edge pairs of lane 0's row: odk_convex.h)
#endif
#ifdef ODK_PROFILE
  const long long _loop0 = clock64();
#endif
#pragma unroll 1
  for (;;) {
    asm volatile("; HF_LOOP_BEGIN" ::: "memory");
    unsigned open_t[4], n_pk = 0u;
    {   // open entries of every row's own list: the prefix of the sorted list within reach of the fourth-deepest contact so far (one wave-wide
        // ballot holds all four rows' bits), less the entries taken (by any row) in the iterations before
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const unsigned pk = (unsigned)__builtin_amdgcn_readlane(last_pk, 16 * r);      // ("none" = foot 0, entry 31: a bit no list has)
        const unsigned long long bit = 1ull << ((pk & 31u) + ((pk >> 8) & 1u) * 32u);
        taken01 |= (pk >> 9) ? 0ull : bit; taken23 |= (pk >> 9) ? bit : 0ull;
      }
      const float lim = TOP[8 * 3];
      const unsigned long long b0 = __builtin_amdgcn_ballot_w64(!(s0 > lim)), b1 = __builtin_amdgcn_ballot_w64(!(s1 > lim));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const unsigned tk = (unsigned)((t < 2 ? taken01 : taken23) >> (32 * (t & 1)));
        open_t[t] = ((unsigned)((b0 >> (16 * t)) & 0xFFFFull) | ((unsigned)((b1 >> (16 * t)) & 0x3ull) << 16)) & ~tk;
        n_pk |= (unsigned)__popc(open_t[t]) << (8 * t);
      }
    }
    if (n_pk == 0u || HF_KNOCK(2)) break;
    // assignment: a row works its own foot while that has open entries, an idle row goes where most are left.  The rule is a function of the four
    // counts capped at four (a foot can use its own row and three helpers): ONE scalar load from a 625-entry table in constant memory
    // (odk_model.h make_hf_assign; the ~300 dependent scalar instructions that derived it in place were the largest single block of the loop's scalar work: round 6)
    const unsigned aw = (unsigned)ODK_HF_ASSIGN.v[hf_assign_index(n_pk)];
    const unsigned maxq = (aw >> 20) & 3u;
    const int my_tg = (int)((aw >> (2 * r_own)) & 3u), my_q = (int)((aw >> (8 + 2 * r_own)) & 3u);
    const bool my_on = (aw >> (16 + r_own)) & 1u;
    // a row on a new foot: that hull's tables, the window's width and centroid, its faces / edges into the registers
    const bool sw = my_on && my_tg != tg;
    if (__builtin_amdgcn_ballot_w64(sw) != 0) {
      if (sw) {
        tg = my_tg;
        float* Lt = tgt_L(tg); const int ft = tg & 1;
        const float* mt = tgt_meta(tg);
        ncw_t = __float_as_int(mt[0]);
        B.V = Lt + S::O_CFRC + ft * 54; B.N = Lt + S::O_CFRC + 108 + ft * 90;
        B.poly = &m->foot_poly[ft][0][0]; B.edge = &m->foot_edge[ft][0][0];
        B.nv = m->foot_nvert[ft]; B.nf = m->foot_npoly[ft]; B.ne = m->foot_nedge[ft];
        B.c[0] = mt[1]; B.c[1] = mt[2]; B.c[2] = mt[3];
        LISTt = Lt + S::O_D + ft * 172; TOPt = LISTt + 108;
        int rec[5];
        { const int* rp = &m->foot_lane_rec[ft][j][0]; for (int k = 0; k < 5; k++) rec[k] = rp[k]; }
        edge_regs_from_rec<3>(RB, B, rec);
        face_regs_from_rec<2>(FB, B, rec, j);
      }
      ODK_SYNC();
    }
#ifdef ODK_PROFILE
    if (lane == 0) L[S::O_SCR + S::S_PROF2 + 7] += (float)(clock64() - _hp);   // (of `select + prism`: up to the end of the row switch)
#endif
    // the open entry of rank my_q of that foot's (sorted) list
    unsigned wsel = open_t[3];      // (one select per statement: the nested form became branches)
    wsel = my_tg == 2 ? open_t[2] : wsel; wsel = my_tg == 1 ? open_t[1] : wsel; wsel = my_tg == 0 ? open_t[0] : wsel;
    for (int q = 0; q < 3; q++) wsel = q < my_q ? wsel & (wsel - 1u) : wsel;
    const bool act = my_on && wsel != 0u;
    const int kk = act ? __ffs((int)wsel) - 1 : 0;
    last_pk = act ? (my_tg << 8) | kk : 31;
    const float* en = LISTt + 6 * kk;
    const int pf = act ? __float_as_int(en[0]) : 0;
    const int p = pf & 255, face_a = pf >> 8;
    const float z[3] = {en[1], en[2], en[3]};
    const float sep_a = en[4];
    Prism P;
    prism(p, ncw_t, z, P);
    {   // lane j < 6 writes vertex j for the polygon fetch (selects, no per-lane indexing of register arrays: that is scratch)
      const int t = j < 3 ? j : j - 3;
      const float vx = t == 0 ? P.x[0] : (t == 1 ? P.x[1] : P.x[2]), vy = t == 0 ? P.y[0] : (t == 1 ? P.y[1] : P.y[2]);
      const float vz = j >= 3 ? -base : (j == 0 ? z[0] : (j == 1 ? z[1] : z[2]));
      if (act && j < 6) { PV[3 * j] = vx; PV[3 * j + 1] = vy; PV[3 * j + 2] = vz; }
    }
    const float pc[3] = {(P.x[0] + P.x[1] + P.x[2]) * (1.0f / 3.0f), (P.y[0] + P.y[1] + P.y[2]) * (1.0f / 3.0f), (z[0] + z[1] + z[2] - 3.0f * base) * (1.0f / 6.0f)};
#ifdef ODK_PROFILE
    float* prof = lane == 0 ? L + S::O_SCR + S::S_PROF2 + 8 : nullptr;
    long long tp = clock64();
    if (prof) { prof[0] += (float)(tp - _hp); }
    sat_prism_row<2, 3>(P, pc, PV, B, FB, RB, sep_a, face_a, RSS, j, act, (float)(4 * p), 0, prof, tp);
#else
    sat_prism_row<2, 3>(P, pc, PV, B, FB, RB, sep_a, face_a, RSS, j, act, (float)(4 * p), knock);
#endif
    // opt-in (odk_env_config.hfield_up_normals_only; oracle hfield_mode 3; default off): a pair's contacts count only when its
    // normal points up -- one wave-uniform flag from the batch's model, nothing on the default path but the test
    if (up_only) {
      ODK_SYNC();
      if (act && j < 4 && !(NEW[8 * j + 6] > 0.5f)) NEW[8 * j] = 1.0f;
    }
#if defined(ODK_HF_VARIANT) && ODK_HF_VARIANT == 4
    // hypothesis sweep (make libodk_hfv4.so; oracle hfield_mode 4): one contact per prism, its deepest (first of equals)
    ODK_SYNC();
    {
      const float d0 = NEW[0], d1 = NEW[8], d2 = NEW[16], d3 = NEW[24];
      int kb = 0; float db = d0;
      if (d1 < db) { db = d1; kb = 1; }
      if (d2 < db) { db = d2; kb = 2; }
      if (d3 < db) { db = d3; kb = 3; }
      ODK_SYNC();
      if (act && j < 4 && j != kb) NEW[8 * j] = 1.0f;
    }
#endif
    ODK_SYNC();
#ifdef ODK_PROFILE
    if (prof) prof[7] += (float)(clock64() - tp);   // (of `writes + merge`: the edge contact and the writes, before the merges)
#endif
    if (!HF_KNOCK(6)) for (unsigned q = 0; q <= maxq; q++) merge_top4_row(TOPt, NEW, j, act && my_q == (int)q);
#ifdef ODK_PROFILE
    { const long long t2 = clock64(); if (prof) prof[6] += (float)(t2 - tp); _hp = t2; }
#endif
    HF_COUNT(4, 1);
    HF_COUNT(6, act ? 1 : 0);
    asm volatile("; HF_LOOP_END" ::: "memory");
  }
#ifdef ODK_PROFILE
  if (lane == 0) { const long long _t = clock64(); L[S::O_SCR + S::S_PROF2 + 3] += (float)(_t - _loop0); _hp = _t; }
#endif
  // ---- the best four: contact distance, position (relative to the base origin), frame; back in the world frame
  float out[7];
  for (int k = 0; k < 7; k++) out[k] = TOP[8 * (j & 3) + k];
  ODK_SYNC();
  if (j < 4) {
    const int c = 4 * f + j;
    float pw[3], nw[3];
    for (int a = 0; a < 3; a++) {
      pw[a] = ph[a] + Rh[3 * a] * (out[1] + org[0]) + Rh[3 * a + 1] * (out[2] + org[1]) + Rh[3 * a + 2] * out[3];
      nw[a] = Rh[3 * a] * out[4] + Rh[3 * a + 1] * out[5] + Rh[3 * a + 2] * out[6];
    }
    CDIST[c] = out[0];
    for (int t = 0; t < 3; t++) CR[3 * c + t] = pw[t] - ref[t];
    make_frame_dev(nw, L + S::O_SCR + S::S_FR + 9 * c);
  }
}

// Primitive foot colliders (sphere / capsule feet on the plane floor): mjx collision_primitive.py as the oracle restates
// it -- plane_sphere, plane_capsule (two contacts, frame aligned with the capsule axis), sphere_sphere, sphere_capsule,
// capsule_capsule.  A handful of scalar operations: lane 0 of the env does them; out of line so that the duck's own kernels pay one
// uniform branch.  Writes all twelve contact slots, the eight floor-contact frames (S_FR, as the height-field path) and the
// foot-foot frame (S_VF, as the convex-convex path).
template <class S, int G>
__device__ __noinline__ void prim_contacts(float* L, const DevModel* __restrict__ m, int lane, bool floor) {
  constexpr int NB = S::NB;
  float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR; float* SCR = L + S::O_SCR; float* FR = L + S::O_SCR + S::S_FR;
  const float* XPOS = L + S::O_XPOS; const float* XQUAT = L + S::O_XQUAT; const float* QPOS = L + S::O_QPOS;
  if (lane != 0) return;
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  const float pn[3] = {m->plane_n[0], m->plane_n[1], m->plane_n[2]}, pp[3] = {m->plane_pos[0], m->plane_pos[1], m->plane_pos[2]};
  float cw[2][3], aw[2][3];
  for (int f = 0; f < 2; f++) {
    const int fb = m->foot_body[f];
    float q[4], R[9];
    for (int t = 0; t < 4; t++) q[t] = XQUAT[t * NB + fb];
    q2mat(R, q);
    for (int t = 0; t < 3; t++) {
      cw[f][t] = XPOS[t * NB + fb] + R[3 * t] * m->foot_gpos[f][0] + R[3 * t + 1] * m->foot_gpos[f][1] + R[3 * t + 2] * m->foot_gpos[f][2];
      aw[f][t] = R[3 * t] * m->foot_gaxis[f][0] + R[3 * t + 1] * m->foot_gaxis[f][1] + R[3 * t + 2] * m->foot_gaxis[f][2];
    }
  }
  auto put = [&](int c, float dist, const float* pos) { CDIST[c] = dist; CR[3 * c] = pos[0] - ref[0]; CR[3 * c + 1] = pos[1] - ref[1]; CR[3 * c + 2] = pos[2] - ref[2]; };
  const float zero3[3] = {ref[0], ref[1], ref[2]};
  // ---- floor (a plane; the height-field floor's contacts are hfield_prim_floor's)
  for (int f = 0; f < 2 && floor; f++) {
    const float r = m->foot_gsize[f][0], hl = m->foot_gsize[f][1];
    const bool cap = m->foot_gtype[f] == 3;
    float fr[9];
    if (cap) {   // frame aligned with the capsule axis: b = axis - n (n . axis), y / z when the capsule stands on end
      float b[3];
      const float na = dot3(pn, aw[f]);
      for (int t = 0; t < 3; t++) b[t] = aw[f][t] - pn[t] * na;
      const float bn = sqrtf(dot3(b, b));
      if (bn < 0.5f) { const bool yy = -0.5f < pn[1] && pn[1] < 0.5f; b[0] = 0.0f; b[1] = yy ? 1.0f : 0.0f; b[2] = yy ? 0.0f : 1.0f; }
      else { const float inv = 1.0f / bn; b[0] *= inv; b[1] *= inv; b[2] *= inv; }
      fr[0] = pn[0]; fr[1] = pn[1]; fr[2] = pn[2]; fr[3] = b[0]; fr[4] = b[1]; fr[5] = b[2];
      cross3(fr + 6, pn, b);
    } else {
      make_frame_dev(pn, fr);
    }
    const int nco = cap ? 2 : 1;
    for (int s = 0; s < 4; s++) {
      const int c = 4 * f + s;
      if (s < nco) {
        const float sg = cap ? (s == 0 ? hl : -hl) : 0.0f;
        const float e[3] = {cw[f][0] + sg * aw[f][0], cw[f][1] + sg * aw[f][1], cw[f][2] + sg * aw[f][2]};
        const float dist = (e[0] - pp[0]) * pn[0] + (e[1] - pp[1]) * pn[1] + (e[2] - pp[2]) * pn[2] - r;
        const float pos[3] = {e[0] - pn[0] * (r + 0.5f * dist), e[1] - pn[1] * (r + 0.5f * dist), e[2] - pn[2] * (r + 0.5f * dist)};
        put(c, dist, pos);
      } else {
        put(c, 1.0f, zero3);
      }
      for (int t = 0; t < 9; t++) FR[9 * c + t] = fr[t];
    }
  }
  // ---- foot against foot: closest points of the two axes' segments (a sphere: a segment of length zero), then sphere_sphere
  auto seg_point = [](const float* a, const float* b, const float* pt, float* out) {   // math.closest_segment_point
    float ab[3], t[3];
    sub3(ab, b, a); sub3(t, pt, a);
    float tt = dot3(t, ab) / (dot3(ab, ab) + 1e-6f);
    tt = fminf(fmaxf(tt, 0.0f), 1.0f);
    for (int i = 0; i < 3; i++) out[i] = a[i] + tt * ab[i];
  };
  float p1[3], p2[3];
  const bool c1 = m->foot_gtype[0] == 3, c2 = m->foot_gtype[1] == 3;
  const float h1 = m->foot_gsize[0][1], h2 = m->foot_gsize[1][1];
  float a0[3], a1[3], b0[3], b1[3];
  for (int t = 0; t < 3; t++) { a0[t] = cw[0][t] - h1 * aw[0][t]; a1[t] = cw[0][t] + h1 * aw[0][t]; b0[t] = cw[1][t] - h2 * aw[1][t]; b1[t] = cw[1][t] + h2 * aw[1][t]; }
  if (c1 && c2) {   // math.closest_segment_to_segment_points
    // the same formula in a form that keeps float32 digits when the capsules are near parallel (two feet side by side):
    // 1 - (da.db)^2 = |da x db|^2 and -da.diff + (da.db)(db.diff) = -(db x (da x db)).diff, with the unit axes taken from the
    // rotation matrices instead of normalised end-point differences (rounding error ~1e-7 / sin(angle), not / sin^2)
    const float* da = aw[0]; const float* db = aw[1]; const float* am = cw[0]; const float* bm = cw[1];
    const float la = 2.0f * h1, lb = 2.0f * h2;
    float diff[3], cx[3], u[3];
    sub3(diff, am, bm);
    cross3(cx, da, db); cross3(u, db, cx);
    const float dotb = dot3(db, diff), dab = dot3(da, db);
    const float ota = -dot3(u, diff) / (dot3(cx, cx) + 1e-6f), otb = dotb + ota * dab;
    const float ta = fminf(fmaxf(ota, -0.5f * la), 0.5f * la), tb = fminf(fmaxf(otb, -0.5f * lb), 0.5f * lb);
    // (ca, cb) = the clamped points; each re-projected on the other segment (closest_segment_point, its 1e-6 included) moves
    // by da_ / db_ along its axis; the pair with the smaller distance wins.  Away from the clamps the two distances differ only
    // through the regularisers (1e-6 ... 1e-4 relative): d1 - d2 is formed from the shifts, not from two rounded distances.
    float w[3];
    for (int t = 0; t < 3; t++) w[t] = diff[t] + ta * da[t] - tb * db[t];
    const float wda = dot3(w, da), wdb = dot3(w, db);
    const float sa = fminf(fmaxf(la * (ta + 0.5f * la - wda) / (la * la + 1e-6f), 0.0f), 1.0f) * la;
    const float sb = fminf(fmaxf(lb * (tb + 0.5f * lb + wdb) / (lb * lb + 1e-6f), 0.0f), 1.0f) * lb;
    const float da_ = sa - (ta + 0.5f * la), db_ = sb - (tb + 0.5f * lb);
    const bool first = 2.0f * da_ * wda + 2.0f * db_ * wdb + da_ * da_ - db_ * db_ < 0.0f;   // |na - cb|^2 < |ca - nb|^2
    const float fa = first ? ta + da_ : ta, fb = first ? tb : tb + db_;
    for (int t = 0; t < 3; t++) { p1[t] = am[t] + fa * da[t]; p2[t] = bm[t] + fb * db[t]; }
  } else if (c2) { ld3(p1, cw[0]); seg_point(b0, b1, cw[0], p2); }
  else if (c1) { ld3(p1, cw[1]); seg_point(a0, a1, cw[1], p2); }   // geoms ordered by type (mjx): the sphere is geom 1, the normal points from it to the capsule
  else { ld3(p1, cw[0]); ld3(p2, cw[1]); }
  {
    const bool swapped = c1 && !c2;
    const float r1 = m->foot_gsize[swapped ? 1 : 0][0], r2 = m->foot_gsize[swapped ? 0 : 1][0];
    float n[3];
    sub3(n, p2, p1);
    const float len = sqrtf(dot3(n, n));
    if (len < 1e-15f) { n[0] = 1.0f; n[1] = 0.0f; n[2] = 0.0f; } else { const float inv = 1.0f / len; n[0] *= inv; n[1] *= inv; n[2] *= inv; }
    const float dist = len - (r1 + r2);
    const float pos[3] = {p1[0] + n[0] * (r1 + 0.5f * dist), p1[1] + n[1] * (r1 + 0.5f * dist), p1[2] + n[2] * (r1 + 0.5f * dist)};
    put(8, dist, pos);
    for (int s = 1; s < 4; s++) put(8 + s, 1.0f, zero3);
    make_frame_dev(n, SCR + S::S_VF);
    // the Jacobian rows are frame . (v(foot 1) - v(foot 0)): with the geoms swapped that is (-frame) . (v(geom 2) - v(geom 1))
    if (swapped) for (int t = 0; t < 9; t++) SCR[S::S_VF + t] = -SCR[S::S_VF + t];
  }
}

// Sphere / capsule feet on a height-field floor: mjx hfield_sphere / hfield_capsule as the oracle restates them
// (oracle/odk_oracle.c: sphere_convex_at, capsule_convex_at, hfield_prim) -- the primitive against the prism of every cell under its
// bounding sphere, the deepest contact (sphere) / the two deepest (capsule) kept, ties to the lower candidate index.  Foot f = 16-lane
// row f of the env, lane = prism (two passes cover the 3 x 3 window).  Every face / edge of a prism is worked branch-free in unrolled
// loops with the result of the winning one kept by selects (a runtime face index would put the prism in scratch).  Rare path, out of line.
template <class S, int G>
__device__ __noinline__ void hfield_prim_floor(float* L, const DevModel* __restrict__ m, const float* __restrict__ hf, int lane) {
  constexpr int NB = S::NB;
  static_assert(G == 32, "height-field floors run 32 lanes per env");
  float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR; float* FR = L + S::O_SCR + S::S_FR;
  const float* XPOS = L + S::O_XPOS; const float* XQUAT = L + S::O_XQUAT; const float* QPOS = L + S::O_QPOS;
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  const int f = (lane >> 4) & 1, j = lane & 15;
  const float* Rh = m->floor_mat; const float ph[3] = {m->plane_pos[0], m->plane_pos[1], m->plane_pos[2]};
  const bool cap = m->foot_gtype[f] == 3;
  const float r = m->foot_gsize[f][0], hl = cap ? m->foot_gsize[f][1] : 0.0f;
  float cl[3], al[3];
  {
    const int fb = m->foot_body[f];
    float q[4], R[9], cw[3], aw[3];
    for (int t = 0; t < 4; t++) q[t] = XQUAT[t * NB + fb];
    q2mat(R, q);
    for (int t = 0; t < 3; t++) {
      cw[t] = XPOS[t * NB + fb] - ph[t] + R[3 * t] * m->foot_gpos[f][0] + R[3 * t + 1] * m->foot_gpos[f][1] + R[3 * t + 2] * m->foot_gpos[f][2];
      aw[t] = R[3 * t] * m->foot_gaxis[f][0] + R[3 * t + 1] * m->foot_gaxis[f][1] + R[3 * t + 2] * m->foot_gaxis[f][2];
    }
    for (int a = 0; a < 3; a++) { cl[a] = Rh[a] * cw[0] + Rh[3 + a] * cw[1] + Rh[6 + a] * cw[2]; al[a] = Rh[a] * aw[0] + Rh[3 + a] * aw[1] + Rh[6 + a] * aw[2]; }
  }
  const int nc = m->hfield_ncol, nr = m->hfield_nrow;
  const float sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2], base = m->hfield_size[3];
  const float dx = 2.0f * sx / (float)(nc - 1), dy = 2.0f * sy / (float)(nr - 1);
  const float rad = r + hl;
  int cmin = (int)floorf((cl[0] - rad + sx) / dx), cmax = (int)floorf((cl[0] + rad + sx) / dx);
  int rmin = (int)floorf((cl[1] - rad + sy) / dy), rmax = (int)floorf((cl[1] + rad + sy) / dy);
  cmin = cmin < 0 ? 0 : cmin; rmin = rmin < 0 ? 0 : rmin; cmax = cmax > nc - 2 ? nc - 2 : cmax; rmax = rmax > nr - 2 ? nr - 2 : rmax;
  int ncw = cmax - cmin + 1, nrw = rmax - rmin + 1;
  ncw = ncw > 3 ? 3 : ncw; nrw = nrw > 3 ? 3 : nrw;   // (the loader checks that the bounding sphere spans less than two cells)
  const float org[2] = {-sx + (float)cmin * dx, -sy + (float)rmin * dy};   // everything relative to the window's first grid corner (float32 digits)
  cl[0] -= org[0]; cl[1] -= org[1];
  const float ea[3] = {cl[0] - hl * al[0], cl[1] - hl * al[1], cl[2] - hl * al[2]}, eb[3] = {cl[0] + hl * al[0], cl[1] + hl * al[1], cl[2] + hl * al[2]};
  const float idiag = 1.0f / sqrtf(dx * dx + dy * dy);
  const int nprism = (ncw > 0 && nrw > 0) ? 2 * ncw * nrw : 0;
  // the lane's two best candidates so far, ordered by (dist, candidate index)
  float bd[2] = {3.0e38f, 3.0e38f}, bp[2][3] = {{0, 0, 0}, {0, 0, 0}}, bn[2][3] = {{0, 0, 1}, {0, 0, 1}};
  int bidx[2] = {0x7FFFFFFF, 0x7FFFFFFF};
  auto offer = [&](float d, const float* p, const float* n, int idx) {
    const bool lt0 = d < bd[0] || (d == bd[0] && idx < bidx[0]), lt1 = d < bd[1] || (d == bd[1] && idx < bidx[1]);
    for (int k = 0; k < 3; k++) { bp[1][k] = lt0 ? bp[0][k] : (lt1 ? p[k] : bp[1][k]); bn[1][k] = lt0 ? bn[0][k] : (lt1 ? n[k] : bn[1][k]); bp[0][k] = lt0 ? p[k] : bp[0][k]; bn[0][k] = lt0 ? n[k] : bn[0][k]; }
    bd[1] = lt0 ? bd[0] : (lt1 ? d : bd[1]); bidx[1] = lt0 ? bidx[0] : (lt1 ? idx : bidx[1]);
    bd[0] = lt0 ? d : bd[0]; bidx[0] = lt0 ? idx : bidx[0];
  };
  auto seg_point = [](const float* a, const float* b, const float* pt, float* out) {   // math.closest_segment_point
    float ab[3], t[3];
    sub3(ab, b, a); sub3(t, pt, a);
    float tt = dot3(t, ab) / (dot3(ab, ab) + 1e-6f);
    tt = fminf(fmaxf(tt, 0.0f), 1.0f);
    for (int i = 0; i < 3; i++) out[i] = a[i] + tt * ab[i];
  };
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    const int p = 16 * pass + j;
    const bool valid = p < nprism;
    if (__builtin_amdgcn_ballot_w64(valid) == 0) break;
    Prism P;
    {
      const int pp = valid ? p : 0, q = pp >> 1, tri = pp & 1;
      // q / ncw for q <= 8, ncw = 1 / 2 / 3 as a multiply and a shift (q 32 >> 5, q 16 >> 5, q 11 >> 5): the nested selects this replaces were compiled into
    // four basic blocks with exec-mask bookkeeping, once per pair-loop iteration and once per cull pass (round 6)
    const int ri = (q * (ncw == 1 ? 32 : (ncw == 2 ? 16 : 11))) >> 5;
      const int c = q - ri * ncw;
      const int cc[3] = {tri ? c + 1 : c, tri ? c : c + 1, tri ? c + 1 : c}, rr[3] = {tri ? ri + 1 : ri, tri ? ri + 1 : ri, tri ? ri : ri + 1};
      for (int k = 0; k < 3; k++) { P.x[k] = (float)cc[k] * dx; P.y[k] = (float)rr[k] * dy; P.z[k] = valid ? hf[(rmin + rr[k]) * nc + cmin + cc[k]] * sz : 0.0f; }
      P.base = base;
      const float e1[3] = {P.x[1] - P.x[0], P.y[1] - P.y[0], P.z[1] - P.z[0]}, e2[3] = {P.x[2] - P.x[0], P.y[2] - P.y[0], P.z[2] - P.z[0]};
      cross3(P.nt, e1, e2);
      const float inv = 1.0f / sqrtf(dot3(P.nt, P.nt));
      P.nt[0] *= inv; P.nt[1] *= inv; P.nt[2] *= inv;
      const float sg = tri ? -1.0f : 1.0f;
      P.ns[0][0] = 0.0f; P.ns[0][1] = -sg; P.ns[1][0] = sg * dy * idiag; P.ns[1][1] = sg * dx * idiag; P.ns[2][0] = -sg; P.ns[2][1] = 0.0f;
    }
    float V[6][3];
#pragma unroll
    for (int k = 0; k < 6; k++) prism_vert(P, k, V[k]);
    // ---- the face of least penetration among those the primitive is behind; its contact(s) computed for every face, the winner's kept
    float best = -3.0e38f;
    bool has_support = true;
    float fd[2] = {1.0f, 1.0f}, fp[2][3] = {{0, 0, 0}, {0, 0, 0}}, fn_[3] = {0, 0, 1};
#pragma unroll
    for (int fa = 0; fa < 5; fa++) {
      float N[3];
      prism_norm(P, fa, N);
      constexpr int CNT[5] = {3, 3, 4, 4, 4};
      const int cnt = CNT[fa];
      const float* v0 = V[PRISM_POLY[fa][1]];
      float ta[3], tb[3];
      sub3(ta, ea, v0); sub3(tb, eb, v0);
      const float sa = dot3(ta, N) - r, sb = dot3(tb, N) - r;
      float sup = fminf(sa, sb);
      has_support = has_support && sup < 0.0f;
      sup = sup >= 0.0f ? -1e12f : sup;
      float cd[2], cp[2][3];
      if (!cap) {   // (model-uniform per row: sphere)
        float pt[3];
        for (int k = 0; k < 3; k++) pt[k] = cl[k] - (sa + r) * N[k];   // the centre projected on the face plane
        bool inside = true; float dmin = 3.0e38f; float q0[3] = {0, 0, 0}, q1[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k >= cnt) break;
          const float* p0 = V[PRISM_POLY[fa][1 + (k + cnt - 1) % cnt]]; const float* p1 = V[PRISM_POLY[fa][1 + k]];
          float e[3], en[3], tp[3];
          sub3(e, p1, p0); cross3(en, e, N); sub3(tp, pt, p0);
          const float ed = dot3(tp, en);
          inside = inside && ed <= 0.0f;
          const bool degenerate = en[0] == 0.0f && en[1] == 0.0f && en[2] == 0.0f;
          const float val = (degenerate || ed < 0.0f) ? 1e12f : ed;
          if (val < dmin) { dmin = val; ld3(q0, p0); ld3(q1, p1); }
        }
        float qe[3];
        seg_point(q0, q1, pt, qe);
        for (int k = 0; k < 3; k++) pt[k] = inside ? pt[k] : qe[k];
        float nn[3];
        sub3(nn, pt, cl);
        const float d = sqrtf(dot3(nn, nn)), inv = 1.0f / (d + (d == 0.0f ? 1e-6f : 0.0f));
        cd[0] = d - r; cd[1] = 1.0f;
        for (int k = 0; k < 3; k++) { nn[k] *= inv; cp[0][k] = 0.5f * (pt[k] + cl[k] + nn[k] * r); cp[1][k] = 0.0f; }
        if (sup > best) { best = sup; fd[0] = cd[0]; fd[1] = 1.0f; for (int k = 0; k < 3; k++) { fp[0][k] = cp[0][k]; fn_[k] = -nn[k]; } }
      } else {
        // the capsule's axis clipped against the face's side planes (_clip_edge_to_planes in the parametric form of clip_edge_row)
        float d01[3];
        sub3(d01, eb, ea);
        const float dd = dot3(d01, d01);
        float b0 = -3.0e38f, b1 = -3.0e38f, t0b = 0.0f, t1b = 1.0f;
        bool both = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k >= cnt) break;
          const float* p0 = V[PRISM_POLY[fa][1 + (k + cnt - 1) % cnt]]; const float* p1 = V[PRISM_POLY[fa][1 + k]];
          float e[3], pn[3], t0[3];
          sub3(e, p1, p0); cross3(pn, e, N); sub3(t0, ea, p0);
          const float a0 = dot3(t0, pn), denom = dot3(pn, d01), a1 = a0 + denom;
          const bool f0 = a0 > 1e-6f, f1 = a1 > 1e-6f;
          both = both || (f0 && f1);
          float t = -a0 / (denom + (denom == 0.0f ? 1e-6f : 0.0f));
          t = fminf(fmaxf(t, 0.0f), 1.0f);
          const float s0 = f0 ? t * dd : 0.0f, s1 = f1 ? (1.0f - t) * dd : 0.0f;
          if (s0 > b0) { b0 = s0; t0b = f0 ? t : 0.0f; }
          if (s1 > b1) { b1 = s1; t1b = f1 ? t : 1.0f; }
        }
        const bool mask = !both && !(t0b > t1b);
#pragma unroll
        for (int e2 = 0; e2 < 2; e2++) {
          const float tt = !both ? (e2 ? t1b : t0b) : (e2 ? 1.0f : 0.0f);
          float cpt[3], tq[3];
          for (int k = 0; k < 3; k++) cpt[k] = (tt == 1.0f ? eb[k] : ea[k] + tt * d01[k]) - N[k] * r;   // the capsule's surface point under the clipped axis point
          sub3(tq, cpt, v0);
          const float off = dot3(tq, N);                                                           // its height over the face plane
          for (int k = 0; k < 3; k++) cp[e2][k] = cpt[k] - 0.5f * off * N[k];
          cd[e2] = mask ? off : 1.0f;   // dist = -penetration = -(face point - surface point) . N   (has_support enters below)
        }
        if (sup > best) { best = sup; for (int e2 = 0; e2 < 2; e2++) { fd[e2] = cd[e2]; for (int k = 0; k < 3; k++) fp[e2][k] = cp[e2][k]; } for (int k = 0; k < 3; k++) fn_[k] = N[k]; }
      }
    }
    float n0[3] = {fn_[0], fn_[1], fn_[2]};
    if (cap) {
      if (!has_support) { fd[0] = 1.0f; fd[1] = 1.0f; }
      // ---- a shallow edge contact: the prism edge closest to the capsule's axis
      float e_dist = 3.0e38f, e_ax[3] = {0, 0, 1}, e_pt[3] = {0, 0, 0}, c_pt[3] = {0, 0, 0};
      bool e_deg = true, e_front = false;
      float da[3];
      sub3(da, eb, ea);
      const float la2 = dot3(da, da), la = sqrtf(la2), ila = 1.0f / (la + (la == 0.0f ? 1e-6f : 0.0f));
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const float* a0 = V[PRISM_EDGE[k][0]]; const float* a1 = V[PRISM_EDGE[k][1]];
        // math.closest_segment_to_segment_points(edge, capsule axis), as the oracle states it
        float dir_a[3], dir_b[3], amid[3], bmid[3], diff[3];
        sub3(dir_a, a1, a0);
        const float lea = sqrtf(dot3(dir_a, dir_a)), ilea = 1.0f / (lea + (lea == 0.0f ? 1e-6f : 0.0f));
        for (int t = 0; t < 3; t++) { dir_a[t] *= ilea; dir_b[t] = da[t] * ila; amid[t] = 0.5f * (a0[t] + a1[t]); bmid[t] = 0.5f * (ea[t] + eb[t]); diff[t] = amid[t] - bmid[t]; }
        const float len_a = 0.5f * lea, len_b = 0.5f * la;
        const float dot_a = dot3(dir_a, diff), dot_b = dot3(dir_b, diff), dot_ab = dot3(dir_a, dir_b);
        const float denom = 1.0f - dot_ab * dot_ab;
        const float ota = (-dot_a + dot_ab * dot_b) / (denom + 1e-6f), otb = dot_b + ota * dot_ab;
        const float t_a = fminf(fmaxf(ota, -len_a), len_a), t_b = fminf(fmaxf(otb, -len_b), len_b);
        float ca[3], cb[3], na[3], nb[3], t1[3], t2[3];
        for (int t = 0; t < 3; t++) { ca[t] = amid[t] + t_a * dir_a[t]; cb[t] = bmid[t] + t_b * dir_b[t]; }
        seg_point(a0, a1, cb, na); seg_point(ea, eb, ca, nb);
        sub3(t1, na, cb); sub3(t2, ca, nb);
        const bool first = dot3(t1, t1) < dot3(t2, t2);
        float pe[3], pc[3], dir[3];
        for (int t = 0; t < 3; t++) { pe[t] = first ? na[t] : ca[t]; pc[t] = first ? cb[t] : nb[t]; dir[t] = pe[t] - pc[t]; }
        const float d2 = dot3(dir, dir), dd = sqrtf(d2);
        if (dd < e_dist) {
          e_dist = dd; e_deg = d2 < 1e-6f;
          const float inv = 1.0f / (dd + (dd == 0.0f ? 1e-6f : 0.0f));
          for (int t = 0; t < 3; t++) { e_ax[t] = dir[t] * inv; e_pt[t] = pe[t]; c_pt[t] = pc[t]; }
          float na_[3], nb_[3];
          prism_norm(P, PRISM_EDGE[k][2], na_); prism_norm(P, PRISM_EDGE[k][3], nb_);
          e_front = dot3(na_, e_ax) < 0.0f && dot3(nb_, e_ax) < 0.0f;
        }
      }
      const bool shallow = !e_deg && e_front;
      const float edge_pen = shallow ? r - e_dist : -1.0f;
      const bool parallel = fabsf(dot3(e_ax, fn_)) > 0.99f && has_support;
      const float min_face = fminf(-fd[0], -fd[1]);
      const bool has_edge = edge_pen > 0.0f && (min_face > 0.0f ? edge_pen < min_face : true) && !parallel;
      if (has_edge) {
        fd[0] = -edge_pen; fd[1] = 1.0f;
        for (int t = 0; t < 3; t++) { fp[0][t] = 0.5f * (e_pt[t] + c_pt[t] + e_ax[t] * r); n0[t] = -e_ax[t]; }
      }
    }
    const int ci = cap ? 2 * p : p;
    if (valid) { offer(fd[0], fp[0], n0, ci); if (cap) offer(fd[1], fp[1], fn_, ci + 1); }
  }
  // ---- the row's best one / two, then back to the world frame
  const int ncon = cap ? 2 : 1;
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    float vm;
    const unsigned kd = fkey(bd[0]), mn = rreduce_u<false>(kd);
    const unsigned wi = rreduce_u<false>(kd == mn ? (unsigned)bidx[0] : 0x7FFFFFFFu);   // lowest candidate index among the minima
    vm = fkey_inv(mn);
    const bool mine = kd == mn && (unsigned)bidx[0] == wi;
    const unsigned own = (unsigned)((__builtin_amdgcn_ballot_w64(mine) >> (threadIdx.x & 48u)) & 0xFFFFull);
    const int src = own ? __ffs((int)own) - 1 : 0;
    float wp[3], wn[3];
    for (int t = 0; t < 3; t++) { wp[t] = row_get(bp[0][t], src); wn[t] = row_get(bn[0][t], src); }
    const bool any = wi != 0x7FFFFFFFu && k < ncon;
    if (j == 0) {
      const int c = 4 * f + k;
      float pw[3], nw[3];
      for (int a = 0; a < 3; a++) {
        pw[a] = ph[a] + Rh[3 * a] * (wp[0] + org[0]) + Rh[3 * a + 1] * (wp[1] + org[1]) + Rh[3 * a + 2] * wp[2];
        nw[a] = any ? Rh[3 * a] * wn[0] + Rh[3 * a + 1] * wn[1] + Rh[3 * a + 2] * wn[2] : Rh[3 * a + 2];
      }
      CDIST[c] = any ? vm : 1.0f;
      for (int t = 0; t < 3; t++) CR[3 * c + t] = any ? pw[t] - ref[t] : 0.0f;
      make_frame_dev(nw, FR + 9 * c);
    }
    if (mine) {   // the winner's lane moves its second candidate up
      bd[0] = bd[1]; bidx[0] = bidx[1]; bd[1] = 3.0e38f; bidx[1] = 0x7FFFFFFF;
      for (int t = 0; t < 3; t++) { bp[0][t] = bp[1][t]; bn[0][t] = bn[1][t]; }
    }
  }
  if (j < 2) {   // slots 2, 3 of the pair stay empty
    const int c = 4 * f + 2 + j;
    const float up[3] = {Rh[2], Rh[5], Rh[8]};
    CDIST[c] = 1.0f; CR[3 * c] = 0.0f; CR[3 * c + 1] = 0.0f; CR[3 * c + 2] = 0.0f;
    make_frame_dev(up, FR + 9 * c);
  }
}

// HF: 0 = plane floor, 1 = height-field floor under the duck's mesh feet, 2 = height-field floor under sphere / capsule feet
// ---- elliptic cones (mjx solver / MuJoCo PrimalUpdateConstraint, HessianCone, PrimalEval as oracle/odk_oracle.c cone_eval restates them).
// One contact = rows (normal, tangent 1, tangent 2) with Jaref x, D = (Dn, Dt, Dt), Dt = Dn impratio, friction mu, mu_r = mu / sqrt(impratio).
// In the scaled space U = (mu_r x_n, mu x_1, mu x_2), N = U_0, T = |U_1..2|:  top zone (N >= mu_r T): no force;  bottom zone
// (mu_r N + T <= 0): the three rows are plain quadratic rows;  middle zone: cost 0.5 Dm (N - mu_r T)^2, Dm = Dn / (mu_r^2 (1 + mu_r^2)).
// A contact with Dn = 0 (not penetrating) has no rows: zone 0.
struct ConeEv { float N, T, U1, U2, Dm, NmT; int zone; };
// Dm = Dn / (mu_r^2 (1 + mu_r^2)): constant per contact and forward pass, formed once by the caller (cone_dm)
__device__ __forceinline__ float cone_dm(float Dn, float mur) { return Dn * __builtin_amdgcn_rcpf(fmaxf(mur * mur * (1.0f + mur * mur), MINVAL_F)); }
__device__ __forceinline__ ConeEv cone_ev(float Dn, float Dm, float mu, float mur, float xn, float x1, float x2) {
  ConeEv e;
  e.U1 = mu * x1; e.U2 = mu * x2; e.N = mur * xn;
  e.T = __builtin_amdgcn_sqrtf(e.U1 * e.U1 + e.U2 * e.U2);      // v_sqrt_f32 (1 ulp): the zone tests below compare against it as it is
  e.Dm = Dm;
  e.NmT = e.N - mur * e.T;
  const bool top = (e.N >= mur * e.T) || (e.T <= 0.0f && e.N >= 0.0f) || !(Dn > 0.0f);
  const bool bot = (mur * e.N + e.T <= 0.0f) || (e.T <= 0.0f && e.N < 0.0f);
  e.zone = top ? 0 : (bot ? 1 : 2);
  return e;
}
// cost of the contact at x
__device__ __forceinline__ float cone_cost(const ConeEv& e, float Dn, float Dt, float xn, float x1, float x2) {
  return e.zone == 0 ? 0.0f : (e.zone == 1 ? 0.5f * (Dn * xn * xn + Dt * (x1 * x1 + x2 * x2)) : 0.5f * e.Dm * e.NmT * e.NmT);
}
// force of row s (0 normal, 1 / 2 tangents) at x
__device__ __forceinline__ float cone_force(const ConeEv& e, float Dn, float Dt, float mu, float mur, float xn, float x1, float x2, int s) {
  if (e.zone == 0 || s > 2) return 0.0f;
  if (e.zone == 1) return s == 0 ? -Dn * xn : -Dt * (s == 1 ? x1 : x2);
  const float fn = -e.Dm * e.NmT * mur;
  return s == 0 ? fn : -fn * __builtin_amdgcn_rcpf(e.T) * (s == 1 ? e.U1 : e.U2) * mu;
}
// row s of the contact's 3 x 3 block d^2 cost / d x^2
__device__ __forceinline__ void cone_hess_row(const ConeEv& e, float Dn, float Dt, float mu, float mur, int s, float* C) {
  C[0] = C[1] = C[2] = 0.0f;
  if (e.zone == 0 || s > 2) return;
  if (e.zone == 1) { C[s] = s == 0 ? Dn : Dt; return; }
  const float iT = __builtin_amdgcn_rcpf(e.T);
  const float g[3] = {1.0f, -mur * e.U1 * iT, -mur * e.U2 * iT}, Sc[3] = {mur, mu, mu};
  // I_t / T - U_t U_t^T / T^3 = [U2^2, -U1 U2; -U1 U2, U1^2] / T^3: written without the difference; this lane's row by selects
  const float i3 = iT * iT * iT, p11 = e.U2 * e.U2 * i3, p12 = -e.U1 * e.U2 * i3, p22 = e.U1 * e.U1 * i3;
  const float P[3] = {0.0f, s == 1 ? p11 : (s == 2 ? p12 : 0.0f), s == 1 ? p12 : (s == 2 ? p22 : 0.0f)};
  const float gs = s == 0 ? g[0] : (s == 1 ? g[1] : g[2]), Ss = s == 0 ? mur : mu;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const float h = e.Dm * gs * g[b] - e.Dm * e.NmT * mur * P[b];
    C[b] = Ss * h * Sc[b];
  }
}
// cost, first and second derivative along x + alpha v (branch-free: the three zones as selects)
__device__ __forceinline__ void cone_line(float Dn, float Dt, float Dm, float mu, float mur, const float* x, const float* v, float& cost, float& d0, float& d1) {
  const ConeEv e = cone_ev(Dn, Dm, mu, mur, x[0], x[1], x[2]);
  const float c1 = 0.5f * (Dn * x[0] * x[0] + Dt * (x[1] * x[1] + x[2] * x[2]));
  const float a1 = Dn * x[0] * v[0] + Dt * (x[1] * v[1] + x[2] * v[2]);
  const float b1 = Dn * v[0] * v[0] + Dt * (v[1] * v[1] + v[2] * v[2]);
  const float V0 = v[0] * mur, V1 = v[1] * mu, V2 = v[2] * mu;
  const float UV = e.U1 * V1 + e.U2 * V2, iT = __builtin_amdgcn_rcpf(e.zone == 2 ? e.T : 1.0f);
  const float cr = e.U1 * V2 - e.U2 * V1;
  const float T1 = UV * iT, T2 = cr * cr * iT * iT * iT;       // d2T = VV / T - UV^2 / T^3 = (U x V)^2 / T^3 (Lagrange): no difference of two large terms
  const float g1 = V0 - mur * T1;
  const float c2 = 0.5f * e.Dm * e.NmT * e.NmT, a2 = e.Dm * e.NmT * g1, b2 = e.Dm * (g1 * g1 - e.NmT * mur * T2);
  cost = e.zone == 0 ? 0.0f : (e.zone == 1 ? c1 : c2);
  d0 = e.zone == 0 ? 0.0f : (e.zone == 1 ? a1 : a2);
  d1 = e.zone == 0 ? 0.0f : (e.zone == 1 ? b1 : b2);
}

template <class S, int G, int HF, bool PRE = false>
__device__ __forceinline__ void forward_env(float* L, const int* RT, const DevModel* __restrict__ m, const float* __restrict__ hfield, const Statics<S, G>& st, int lane, int flags, const HotSt& hot = HotSt()) {
  // RT: the packed reduced entries (DevModel::R_ent) in LDS, one copy per workgroup (load_shared): every substep reads them
  // twice (inertia, Hessian), and a table load from the L2-resident model right behind a phase hand-off is ~300 exposed cycles.
  // Behind them (SH_CT): the contact-row constants -- pair_mu[3] | pair_invweight[3] | pair_imp[3][9] | plane_frame[9].
  const float* SH = reinterpret_cast<const float*>(RT);
  const float* CT = SH + S::SH_CT;
  constexpr int NV = S::NV, NB = S::NB, NR = S::NVR;   // NR: reduced dofs = columns of CDOF / BUF6 / BUF6B
  using ST = Statics<S, G>;
  float* QPOS = L + S::O_QPOS; float* QVEL = L + S::O_QVEL; float* WARM = L + S::O_WARM; float* CTRL = L + S::O_CTRL;
  float* Q0 = L + S::O_Q0; float* MASS = L + S::O_MASS; float* ARM = L + S::O_ARM; float* FRL = L + S::O_FRL; float* KP = L + S::O_KP;
  float* XPOS = L + S::O_XPOS; float* XQUAT = L + S::O_XQUAT; float* CVEL = L + S::O_CVEL; float* CACC = L + S::O_CACC;
  float* CFRC = L + S::O_CFRC; float* CRB = L + S::O_CRB; float* SC = L + S::O_SC; float* CDOF = L + S::O_CDOF;
  float* BUF6 = L + S::O_BUF6; float* BUF6B = L + S::O_BUF6B; float* M = L + S::O_M; float* HL = L + S::O_HL;
  float* QFS = L + S::O_QFS; float* QAS = L + S::O_QAS; float* X = L + S::O_X; float* MA = L + S::O_MA; float* GRAD = L + S::O_GRAD;
  float* MV = L + S::O_MV; float* ED = L + S::O_D; float* AREF = L + S::O_AREF; float* JAR = L + S::O_JAR; float* JV = L + S::O_JV;
  float* W = L + S::O_W; float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR; float* SCR = L + S::O_SCR;
  float* SENS = L + S::O_SENS; float* ACTF = L + S::O_ACTF;
  const int nfl = m->nfl, nlim = m->nlim, r0c = nfl + nlim;
  const float dt = m->dt;
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  ODK_PROF_BEGIN();

  BodySt bs;
  load_body<S>(bs, m, lane);   // issued first: the loads complete under the sin/cos phase
  // ---------------- P0: half-angle sin/cos of every hinge (lane = joint)
  if (st.j_qadr >= 0) {
    float s, c;
    const float x = 0.5f * (QPOS[st.j_qadr] - Q0[st.j_qadr]);
    if (fabsf(x) <= 1.7f) {   // half joint angles are < 0.9 rad inside the joint ranges: Taylor to x^13 / x^14 in FMAs,
      const float x2 = x * x; // truncation < 2e-9, i.e. fp32 rounding only (sincosf's range reduction costs ~5x more)
      float ps = 1.6059043836821613e-10f, pc = -1.1470745597729725e-11f;
      ps = fmaf(ps, x2, -2.505210838544172e-08f); pc = fmaf(pc, x2, 2.08767569878681e-09f);
      ps = fmaf(ps, x2, 2.7557319223985893e-06f); pc = fmaf(pc, x2, -2.755731922398589e-07f);
      ps = fmaf(ps, x2, -0.0001984126984126984f); pc = fmaf(pc, x2, 2.48015873015873e-05f);
      ps = fmaf(ps, x2, 0.008333333333333333f); pc = fmaf(pc, x2, -0.001388888888888889f);
      ps = fmaf(ps, x2, -0.16666666666666666f); pc = fmaf(pc, x2, 0.041666666666666664f);
      ps = fmaf(ps, x2, 1.0f); pc = fmaf(pc, x2, -0.5f);
      s = ps * x; c = fmaf(pc, x2, 1.0f);
    } else {
      sincosf(x, &s, &c);
    }
    SC[2 * lane] = s; SC[2 * lane + 1] = c;
  }
  ODK_SYNC();
  ODK_PROF(0);
  // ---------------- P1: top-down level sweep (lane = body): pose, cdof, cvel, velocity part of cacc, local bias force
  {
    float p[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, cvel[6] = {0, 0, 0, 0, 0, 0}, cacc[6] = {0, 0, 0, 0, 0, 0};
    if (bs.level == -1) {  // world / static bodies
      for (int k = 0; k < 3; k++) p[k] = bs.pos[k];
      for (int k = 0; k < 4; k++) q[k] = bs.quat[k];
    }
    // bodies above the serial chains (world side of the tree: base, trunk): one tree level per step
    for (int lvl = 0; lvl <= m->max_nonpath_level; lvl++) {
      if (bs.level == lvl && !bs.is_path) {
        if (lvl == 0) {  // floating base: free joint (mj_comVel free-joint rule)
          for (int k = 0; k < 3; k++) p[k] = ref[k];
          for (int k = 0; k < 4; k++) q[k] = QPOS[3 + k];
          qnormalize(q);
          float R[9];
          q2mat(R, q);
          const float wl[3] = {QVEL[3], QVEL[4], QVEL[5]};
          float ww[3];
          for (int k = 0; k < 3; k++) ww[k] = R[3 * k] * wl[0] + R[3 * k + 1] * wl[1] + R[3 * k + 2] * wl[2];
          const float v[3] = {QVEL[0], QVEL[1], QVEL[2]};
          for (int k = 0; k < 3; k++) {
            for (int c = 0; c < 6; c++) CDOF[c * NR + k] = (c == 3 + k) ? 1.0f : 0.0f;
            CDOF[0 * NR + 3 + k] = R[k]; CDOF[1 * NR + 3 + k] = R[3 + k]; CDOF[2 * NR + 3 + k] = R[6 + k];
            CDOF[3 * NR + 3 + k] = 0; CDOF[4 * NR + 3 + k] = 0; CDOF[5 * NR + 3 + k] = 0;
          }
          float vxw[3];
          cross3(vxw, v, ww);
          cvel[0] = ww[0]; cvel[1] = ww[1]; cvel[2] = ww[2]; cvel[3] = v[0]; cvel[4] = v[1]; cvel[5] = v[2];
          cacc[3] = -m->gravity[0] + vxw[0]; cacc[4] = -m->gravity[1] + vxw[1]; cacc[5] = -m->gravity[2] + vxw[2];
        } else {
          const int pb = bs.parent;
          float pp[3], pq[4], t[3];
          for (int k = 0; k < 3; k++) pp[k] = XPOS[k * NB + pb];
          for (int k = 0; k < 4; k++) pq[k] = XQUAT[k * NB + pb];
          for (int k = 0; k < 6; k++) { cvel[k] = CVEL[k * NB + pb]; cacc[k] = CACC[k * NB + pb]; }
          qrot(t, pq, bs.pos);
          p[0] = pp[0] + t[0]; p[1] = pp[1] + t[1]; p[2] = pp[2] + t[2];
          qmul(q, pq, bs.quat);
#pragma unroll
          for (int jj = 0; jj < 2; jj++) {
            if (jj < bs.njnt) {  // hinge at the body origin (jnt_pos == 0, checked at load)
              float axw[3], cd[6], dot[6];
              qrot(axw, q, bs.ax[jj]);
              const float off[3] = {ref[0] - p[0], ref[1] - p[1], ref[2] - p[2]};
              cd[0] = axw[0]; cd[1] = axw[1]; cd[2] = axw[2];
              cross3(cd + 3, axw, off);
              const int d = bs.jd[jj], col = bs.jr[jj];
              if (col >= 0) {
#pragma unroll
                for (int k = 0; k < 6; k++) CDOF[k * NR + col] = cd[k];
              }
              const float qv = QVEL[d];
              cross_motion(dot, cvel, cd);
#pragma unroll
              for (int k = 0; k < 6; k++) { cacc[k] += dot[k] * qv; cvel[k] += cd[k] * qv; }
              const float s = SC[2 * bs.jj[jj]], c = SC[2 * bs.jj[jj] + 1];
              const float qj[4] = {c, s * bs.ax[jj][0], s * bs.ax[jj][1], s * bs.ax[jj][2]};
              qmul(q, q, qj);
            }
          }
          qnormalize(q);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) XPOS[k * NB + lane] = p[k];
#pragma unroll
        for (int k = 0; k < 4; k++) XQUAT[k * NB + lane] = q[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { CVEL[k * NB + lane] = cvel[k]; CACC[k * NB + lane] = cacc[k]; }
      }
      ODK_SYNC();
    }
    // serial body chains (legs, head): pose, velocity and bias acceleration as three prefix scans over neighbouring
    // lanes (Hillis-Steele from the chain head, 3 cross-lane steps each) instead of one dependent LDS round trip per
    // tree level.  Every lane runs the shuffles (uniform control flow); only chain bodies use the results.
    {
      const bool isp = bs.is_path != 0, head = bs.path_head != 0;
      // local transform of the body relative to its parent, and the frames in which its joint axes are given
      float ql[4] = {bs.quat[0], bs.quat[1], bs.quat[2], bs.quat[3]}, qb[2][4];
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
#pragma unroll
        for (int k = 0; k < 4; k++) qb[jj][k] = ql[k];
        if (jj < bs.njnt) {
          const float s = SC[2 * bs.jj[jj]], c = SC[2 * bs.jj[jj] + 1];
          const float qj[4] = {c, s * bs.ax[jj][0], s * bs.ax[jj][1], s * bs.ax[jj][2]};
          qmul(ql, ql, qj);
        }
      }
      float tp[3] = {bs.pos[0], bs.pos[1], bs.pos[2]}, tq[4] = {ql[0], ql[1], ql[2], ql[3]};
      float ppq[4] = {1, 0, 0, 0}, pcv[6] = {0, 0, 0, 0, 0, 0}, pca[6] = {0, 0, 0, 0, 0, 0};   // parent's world q, cvel, cacc
      if (isp && head) {   // chain head: its parent is one of the bodies done above
        const int pb = bs.parent;
        float pp[3], t[3];
        for (int k = 0; k < 3; k++) pp[k] = XPOS[k * NB + pb];
        for (int k = 0; k < 4; k++) ppq[k] = XQUAT[k * NB + pb];
        for (int k = 0; k < 6; k++) { pcv[k] = CVEL[k * NB + pb]; pca[k] = CACC[k * NB + pb]; }
        qrot(t, ppq, tp);
        tp[0] = pp[0] + t[0]; tp[1] = pp[1] + t[1]; tp[2] = pp[2] + t[2];
        qmul(tq, ppq, tq);
      }
#pragma unroll
      for (int si = 0; si < 3; si++) {   // world pose: T[b] <- T[b - 2^si] o T[b]
        float up[3], uq[4];
#pragma unroll
        for (int k = 0; k < 3; k++) up[k] = __shfl_up(tp[k], 1 << si, G);
#pragma unroll
        for (int k = 0; k < 4; k++) uq[k] = __shfl_up(tq[k], 1 << si, G);
        if ((bs.upmask >> si) & 1) {
          float t[3];
          qrot(t, uq, tp);
          tp[0] = up[0] + t[0]; tp[1] = up[1] + t[1]; tp[2] = up[2] + t[2];
          qmul(tq, uq, tq);
        }
      }
      {   // parent's world orientation: the lane above, except for chain heads (read from LDS above)
        float uq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) uq[k] = __shfl_up(tq[k], 1, G);
        if (isp && !head) { ppq[0] = uq[0]; ppq[1] = uq[1]; ppq[2] = uq[2]; ppq[3] = uq[3]; }
      }
      // joint twists about the base origin and the body's own velocity / acceleration increments
      float cd[2][6], qv[2] = {0, 0}, cinc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
#pragma unroll
        for (int k = 0; k < 6; k++) cd[jj][k] = 0.0f;
        if (isp && jj < bs.njnt) {
          float qw[4], axw[3];
          qmul(qw, ppq, qb[jj]);
          qrot(axw, qw, bs.ax[jj]);
          const float off[3] = {ref[0] - tp[0], ref[1] - tp[1], ref[2] - tp[2]};
          cd[jj][0] = axw[0]; cd[jj][1] = axw[1]; cd[jj][2] = axw[2];
          cross3(cd[jj] + 3, axw, off);
          const int d = bs.jd[jj], col = bs.jr[jj];
          if (col >= 0) {   // a twin's motion vector is its main dof's: one column
#pragma unroll
            for (int k = 0; k < 6; k++) CDOF[k * NR + col] = cd[jj][k];
          }
          qv[jj] = QVEL[d];
#pragma unroll
          for (int k = 0; k < 6; k++) cinc[k] += cd[jj][k] * qv[jj];
        }
      }
      float cv[6];
#pragma unroll
      for (int k = 0; k < 6; k++) cv[k] = cinc[k] + pcv[k];   // heads start from their parent's velocity
#pragma unroll
      for (int si = 0; si < 3; si++) {
        float u[6];
#pragma unroll
        for (int k = 0; k < 6; k++) u[k] = __shfl_up(cv[k], 1 << si, G);
        const bool ok = (bs.upmask >> si) & 1;
#pragma unroll
        for (int k = 0; k < 6; k++) cv[k] += ok ? u[k] : 0.0f;
      }
      {   // velocity of the parent (= velocity before this body's first joint)
        float u[6];
#pragma unroll
        for (int k = 0; k < 6; k++) u[k] = __shfl_up(cv[k], 1, G);
        if (isp && !head) {
#pragma unroll
          for (int k = 0; k < 6; k++) pcv[k] = u[k];
        }
      }
      float ca[6];
#pragma unroll
      for (int k = 0; k < 6; k++) ca[k] = pca[k];
      if (isp) {
        float vb[6] = {pcv[0], pcv[1], pcv[2], pcv[3], pcv[4], pcv[5]};
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          if (jj < bs.njnt) {
            float dot[6];
            cross_motion(dot, vb, cd[jj]);
#pragma unroll
            for (int k = 0; k < 6; k++) { ca[k] += dot[k] * qv[jj]; vb[k] += cd[jj][k] * qv[jj]; }
          }
        }
      }
#pragma unroll
      for (int si = 0; si < 3; si++) {
        float u[6];
#pragma unroll
        for (int k = 0; k < 6; k++) u[k] = __shfl_up(ca[k], 1 << si, G);
        const bool ok = (bs.upmask >> si) & 1;
#pragma unroll
        for (int k = 0; k < 6; k++) ca[k] += ok ? u[k] : 0.0f;
      }
      if (isp) {
        qnormalize(tq);
#pragma unroll
        for (int k = 0; k < 3; k++) { p[k] = tp[k]; XPOS[k * NB + lane] = tp[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) { q[k] = tq[k]; XQUAT[k * NB + lane] = tq[k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) { cvel[k] = cv[k]; cacc[k] = ca[k]; CVEL[k * NB + lane] = cv[k]; CACC[k * NB + lane] = ca[k]; }
      }
      ODK_SYNC();
    }
    float acc[16];   // cinert (10) | local bias force (6) of this lane's body, then their subtree sums
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.0f;
    if (lane < NB) {
      if (bs.level < 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) XPOS[k * NB + lane] = p[k];
#pragma unroll
        for (int k = 0; k < 4; k++) XQUAT[k * NB + lane] = q[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { CVEL[k * NB + lane] = 0; CACC[k * NB + lane] = 0; }
      }
      // cinert about the base origin, local bias force
      float R[9];
      q2mat(R, q);
      float ip[3] = {bs.ipos[0], bs.ipos[1], bs.ipos[2]};
      {   // (body 1's per-env centre of mass: read by every lane -- a broadcast -- and pinned; as an `if (lane == 1)` the three reads became three blocks under their own exec mask)
        const float i0 = L[S::O_IPOS1], i1 = L[S::O_IPOS1 + 1], i2 = L[S::O_IPOS1 + 2];
        asm volatile("" :: "v"(i0), "v"(i1), "v"(i2));
        ip[0] = lane == 1 ? i0 : ip[0]; ip[1] = lane == 1 ? i1 : ip[1]; ip[2] = lane == 1 ? i2 : ip[2];
      }
      float off[3];
      for (int k = 0; k < 3; k++) off[k] = p[k] + R[3 * k] * ip[0] + R[3 * k + 1] * ip[1] + R[3 * k + 2] * ip[2] - ref[k];
      const float* f = bs.inertia;
      const float Ib[9] = {f[0], f[3], f[4], f[3], f[1], f[5], f[4], f[5], f[2]};
      float T[9], Iw[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i] * Ib[j] + R[3 * i + 1] * Ib[3 + j] + R[3 * i + 2] * Ib[6 + j];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Iw[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
      const float mb = MASS[lane], o2 = dot3(off, off);
      float ci[10];
      ci[0] = Iw[0] + mb * (o2 - off[0] * off[0]);
      ci[1] = Iw[4] + mb * (o2 - off[1] * off[1]);
      ci[2] = Iw[8] + mb * (o2 - off[2] * off[2]);
      ci[3] = Iw[1] - mb * off[0] * off[1];
      ci[4] = Iw[2] - mb * off[0] * off[2];
      ci[5] = Iw[5] - mb * off[1] * off[2];
      ci[6] = mb * off[0]; ci[7] = mb * off[1]; ci[8] = mb * off[2];
      ci[9] = mb;
      float fr[6], t1[6], a3[3], c3[3];
      inert_mul(fr, ci, cacc);
      inert_mul(t1, ci, cvel);
      cross3(a3, cvel, t1);
      cross3(c3, cvel + 3, t1 + 3);
      fr[0] += a3[0] + c3[0]; fr[1] += a3[1] + c3[1]; fr[2] += a3[2] + c3[2];
      cross3(a3, cvel, t1 + 3);
      fr[3] += a3[0]; fr[4] += a3[1]; fr[5] += a3[2];
      const bool dyn = bs.level >= 0;
#pragma unroll
      for (int k = 0; k < 10; k++) acc[k] = dyn ? ci[k] : 0.0f;
#pragma unroll
      for (int k = 0; k < 6; k++) acc[10 + k] = dyn ? fr[k] : 0.0f;
    }
    ODK_PROF(1);
    // ---------------- P2: composite inertia and subtree bias force.  Serial body chains (legs, head): suffix sums
    // over neighbouring lanes, three cross-lane steps in registers; the few bodies above them: one level per step.
#pragma unroll
    for (int si = 0; si < 3; si++) {
      const bool ok = (bs.pathmask >> si) & 1;
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; k++) t[k] = __shfl_down(acc[k], 1 << si, G);
#pragma unroll
      for (int k = 0; k < 16; k++) acc[k] += ok ? t[k] : 0.0f;
    }
    if (lane < NB) {
#pragma unroll
      for (int k = 0; k < 10; k++) CRB[k * NB + lane] = acc[k];
#pragma unroll
      for (int k = 0; k < 6; k++) CFRC[k * NB + lane] = acc[10 + k];
    }
    ODK_SYNC();
  }
  // bodies above the serial chains (base, trunk): body i's subtree sum = its own value + the own values of the non-chain
  // bodies below it + the (final) sums of the chain heads below it -- a host-built source list (DevModel::np_src), so every
  // such body is folded in ONE step, 16 lanes per body (lane = component: cfrc | crb are contiguous, component q of body b
  // is at q * NB + b), instead of one dependent LDS round trip per tree level
  {
    float* ACC = CFRC;
    static_assert(S::O_CRB == S::O_CFRC + 6 * NB, "cfrc | crb must be contiguous");
    const int nnp = m->np_count;          // <= 2 * (G / 16), checked at load
    const int slot = lane >> 4, q = lane & 15;
    float t[2] = {0.0f, 0.0f};
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int i = pass * (G / 16) + slot;
      if (i < nnp) {   // (skipped wave-wide for pass 1 when there are at most G / 16 such bodies)
        const int ns = m->np_nsrc[i];
        int src[6];    // all list loads in flight together, then all LDS reads (np_nsrc <= 6, checked at load)
#pragma unroll
        for (int c = 0; c < 6; c++) src[c] = m->np_src[i][c];
        float x = 0.0f;
#pragma unroll
        for (int c = 0; c < 6; c++) { const float y = ACC[q * NB + src[c]]; x += c < ns ? y : 0.0f; }
        t[pass] = x;
      }
    }
    ODK_SYNC();   // all own values read before any is replaced by its sum
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int i = pass * (G / 16) + slot;
      if (i < nnp) ACC[q * NB + m->np_body[i]] = t[pass];
    }
    ODK_SYNC();
  }
  ODK_PROF(2);
  // ---------------- P3: per dof: crb*cdof, bias force, passive, actuation -> qfrc_smooth
  float qfs = 0.0f;
  if (st.d_on) {
    const int i = lane, b = st.d_body;
    float crb[10], cd[6], buf[6], bias = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) crb[k] = CRB[k * NB + b];
#pragma unroll
    for (int k = 0; k < 6; k++) { cd[k] = CDOF[k * NR + st.d_red]; bias += cd[k] * CFRC[k * NB + b]; }
    inert_mul(buf, crb, cd);
    if (st.d_tkind != 2) {
#pragma unroll
      for (int k = 0; k < 6; k++) BUF6[k * NR + st.d_red] = buf[k];
    }
    const float qv = QVEL[i];
    float frc = -st.d_damping * qv - bias;
    const int u = st.d_act;
    if (u >= 0) {
      ActSt as;
      if constexpr (PRE) as = hot.as; else load_act(as, m, u);
      float ctrl = CTRL[u];
      if (as.climited) ctrl = fminf(fmaxf(ctrl, as.clo), as.chi);
      const float kp = KP[u];
      float af = kp * ctrl - kp * QPOS[st.d_qadr] + as.bias2 * qv;
      if (as.flimited) af = fminf(fmaxf(af, as.flo), as.fhi);
      ACTF[u] = af;
      frc += af;
    }
    QFS[i] = frc;
    qfs = frc;
  }
  ODK_SYNC();
  ODK_PROF(3);
  // ---------------- P4: sparse reduced inertia entries (lane = entry): cdof_j . (crb cdof_i), armature on the unpaired
  // diagonals only (a pair's armatures enter the solves as the series term pair_einv, and M v as ARM[i] v_i)
  // Branch-free and batched: all table reads, then all operand reads, then the sums -- as a loop of guarded passes every pass
  // was two dependent LDS round trips behind the previous one's (lanes past the last entry redo entry 0 and store nothing).
  {
    int ee[ST::NME];
#pragma unroll
    for (int t = 0; t < ST::NME; t++) { const int pq = lane + t * G; ee[t] = RT[pq < S::NMR ? pq : 0]; }
    float vv[ST::NME];
#pragma unroll
    for (int t = 0; t < ST::NME; t++) {
      const int e = ee[t], i = e & 31, j = (e >> 5) & 31;
      const float arm = ARM[(e >> 16) & 31];
      float v = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) v += CDOF[k * NR + j] * BUF6[k * NR + i];
      vv[t] = v + ((((e >> 14) & 3) == 1) ? arm : 0.0f);   // diagonal of an unpaired dof
    }
#pragma unroll
    for (int t = 0; t < ST::NME; t++) { const int pq = lane + t * G; if (pq < S::NMR) M[pq] = vv[t]; }
  }
  ODK_SYNC();
  ODK_PROF(4);
  // ---------------- P5: qacc_smooth = M^-1 qfrc_smooth; dense symmetric row of M into registers
  float qas;
  if constexpr (S::PAIRED) {
    // M = P Mr P^T + diag(armature): reduced system with the pairs' series armature on the diagonal (HL), one chain solve
    {   // (branch-free and batched, as P4)
      int ee[ST::NME]; float mm[ST::NME];
#pragma unroll
      for (int t = 0; t < ST::NME; t++) { const int p = lane + t * G, pc = p < S::NMR ? p : 0; ee[t] = RT[pc]; mm[t] = M[pc]; }
#pragma unroll
      for (int t = 0; t < ST::NME; t++) {
        const float pe = pair_einv(ARM, nullptr, (ee[t] >> 16) & 31);
        mm[t] += ((ee[t] >> 15) & 1) ? pe : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < ST::NME; t++) { const int p = lane + t * G; if (p < S::NMR) HL[p] = mm[t]; }
    }
    if (st.r_on) GRAD[lane] = pair_rhs<S>(QFS, ARM, nullptr, m, lane);
    ODK_SYNC();
    ODK_PROF(5);
    chain_solve<S, G>(HL, GRAD, SCR + S::S_K, BUF6, st.cs_pk, lane);
    qas = pair_expand<S, G>(GRAD, QFS, qfs, ARM, nullptr, st, lane);
    if (st.d_on) QAS[lane] = qas;
  } else if constexpr (S::CL > 0) {
    if (st.d_on) QAS[lane] = qfs;
    ODK_SYNC();
    chain_solve<S, G>(M, QAS, SCR + S::S_K, BUF6, st.cs_pk, lane);
    ODK_PROF(5);
    qas = st.d_on ? QAS[lane] : 0.0f;
  } else {
#pragma unroll
    for (int t = 0; t < ST::NME; t++) { const int p = lane + t * G; if (p < S::NMR) HL[p] = M[p]; }
    ODK_SYNC();
    if (m->nrchain > 0 && m->rchain_first[0] == 6)   // floating base + serial chains: chain-parallel elimination
      factor_chains<G, S::DT, NV, S::DT - 5, 6>(HL, lane, st.r_on, st.r_depth, st.r_Madr, st.ch_first, st.ch_len, st.r_descmask, st.r_depth, st.r_Madr);
    else
      factor_rows<G, S::DT, NV>(HL, lane, st.r_on, st.r_depth, st.r_Madr, st.r_descmask, st.r_depth, st.r_Madr);
    ODK_PROF(5);
    qas = solve_rows<G, NV>(HL, qfs, lane, st.r_on, st.r_depth, st.r_Madr, st.r_ancmask, st.r_descmask, st.r_depth, st.r_Madr);
    if (st.d_on) QAS[lane] = qas;
  }
  // y = M v with the dense symmetric row of the reduced M in registers: gathered once per substep (unconditional LDS loads +
  // selects, at the start of the solver) and used by both products -- NR registers held through the solver, which the
  // column-wise chain solve left room for (gathering at each use cost ~120 VALU instructions per product).  VB = P^T v (LDS,
  // one entry per reduced dof; v itself without twins) is read with uniform addresses (broadcast reads on the LDS pipe) instead
  // of 2 x v_readlane + select per element on the VALU.  Twin dofs: y_i = (Mr P^T v)_r(i) + ARM[i] v_i, handed from the
  // reduced-dof lanes to the dof lanes through YB (LDS, NR floats); vi = v of this lane's dof.
  float mrow[NR];
  auto gather_M = [&]() {   // (addresses: Statics::m_adr, one unpack per entry)
    static_assert(S::CL > 0, "the zero entry of unrelated dofs is the floating base's CDOF[0][0]");
    static_assert(S::TOTAL * 4 < 65536, "16-bit byte offsets");
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const unsigned off = ((unsigned)st.m_adr[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      mrow[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(L) + off);
    }
  };
  auto mul_M = [&](const float* VB, float* YB, float vi) -> float {
    float acc0 = 0.0f, acc1 = 0.0f;   // two chains: a single accumulator is NR dependent FMAs
#pragma unroll
    for (int j = 0; j + 1 < NR; j += 2) { acc0 = fmaf(mrow[j], VB[j], acc0); acc1 = fmaf(mrow[j + 1], VB[j + 1], acc1); }
    if constexpr (NR & 1) acc0 = fmaf(mrow[NR - 1], VB[NR - 1], acc0);
    float acc = acc0 + acc1;
    if constexpr (S::PAIRED) {
      if (st.r_on) YB[lane] = acc;
      ODK_SYNC();
      acc = st.d_on ? YB[st.d_red] + (st.d_tkind != 0 ? ARM[lane] * vi : 0.0f) : 0.0f;
    }
    return acc;
  };
  ODK_SYNC();
  ODK_PROF(6);

  // ---------------- P7: collision.  Foot (convex mesh) vs plane: mjx collision_convex.plane_convex
  float fR[2][9], fP[2][3];
#pragma unroll
  for (int f = 0; f < 2; f++) {
    const int fb = m->foot_body[f];
    float q[4];
    for (int k = 0; k < 4; k++) q[k] = XQUAT[k * NB + fb];
    for (int k = 0; k < 3; k++) fP[f][k] = XPOS[k * NB + fb];
    q2mat(fR[f], q);
  }
  if constexpr (HF) {
    // height-field floor (rough terrain; its own kernel instantiation): prisms of the cells under each foot, out of line
    if constexpr (HF == 2) { hfield_prim_floor<S, G>(L, m, hfield, lane); ODK_SYNC(); prim_contacts<S, G>(L, m, lane, false); }   // floor, then foot against foot
    else hfield_contacts<S, G>(L, m, hfield, lane);
  } else if (m->foot_prim) {
    prim_contacts<S, G>(L, m, lane, true);   // sphere / capsule feet (model-uniform branch): floor and foot-foot contacts, out of line
  } else {
  const float pn0[3] = {m->plane_n[0], m->plane_n[1], m->plane_n[2]};
  {
    // both feet at once: foot f = 16-lane row f of the env's lanes (vertices 0..15), the 17th vertex rides along in every lane
    const int f = (lane >> 4) & 1, j = lane & 15;
    const int nvt = m->foot_nvert[f];
    const bool inrow = lane < 32;                    // G = 64: rows 2, 3 idle
    const bool has = inrow && j < nvt, has_e = inrow && nvt > 16;
    float pn[3] = {pn0[0], pn0[1], pn0[2]}, pp[3] = {m->plane_pos[0], m->plane_pos[1], m->plane_pos[2]};
    float Rf[9], Pf[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Rf[k] = f ? fR[1][k] : fR[0][k];
#pragma unroll
    for (int k = 0; k < 3; k++) Pf[k] = f ? fP[1][k] : fP[0][k];
    float w[3] = {0, 0, 0}, we[3] = {0, 0, 0}, sup = -3.0e38f, supe = -3.0e38f;
    {
      const int jv = has ? j : 0;
      float vb[3];
      if constexpr (PRE) { vb[0] = hot.vb[0]; vb[1] = hot.vb[1]; vb[2] = hot.vb[2]; } else { vb[0] = m->foot_vert[f][jv][0]; vb[1] = m->foot_vert[f][jv][1]; vb[2] = m->foot_vert[f][jv][2]; }
      const float ve[3] = {m->foot_vert[f][16][0], m->foot_vert[f][16][1], m->foot_vert[f][16][2]};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        w[k] = Pf[k] + Rf[3 * k] * vb[0] + Rf[3 * k + 1] * vb[1] + Rf[3 * k + 2] * vb[2];
        we[k] = Pf[k] + Rf[3 * k] * ve[0] + Rf[3 * k + 1] * ve[1] + Rf[3 * k + 2] * ve[2];
      }
      const float s0 = (pp[0] - w[0]) * pn[0] + (pp[1] - w[1]) * pn[1] + (pp[2] - w[2]) * pn[2];
      const float s1 = (pp[0] - we[0]) * pn[0] + (pp[1] - we[1]) * pn[1] + (pp[2] - we[2]) * pn[2];
      sup = has ? s0 : -3.0e38f; supe = has_e ? s1 : -3.0e38f;
    }
    int idx[4];
    select4_rows(w, has, sup, we, has_e, supe, nvt, pn, idx, lane);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      bool uniq = true;
      for (int q = 0; q < k; q++) uniq = uniq && (idx[q] != idx[k]);
      const bool ext = idx[k] == 16;
      if (inrow && j == (ext ? 0 : idx[k])) {   // the vertex's own lane writes (lane 0 of the row for the 17th vertex)
        const float sv = ext ? supe : sup;
        const float dist = uniq ? -sv : 1.0f;
        const int c = 4 * f + k;
        CDIST[c] = dist;
        for (int t = 0; t < 3; t++) CR[3 * c + t] = (ext ? we[t] : w[t]) - 0.5f * dist * pn[t] - ref[t];
      }
    }
  }
  }
  ODK_PROF(7);
  // foot-foot: bounding spheres first (a positive gap already means "inactive pair"); only when the spheres of some env
  // in the wave touch: oriented-box cull, 15 axes on 15 lanes (a positive box separation bounds the hulls' from below)
  if (!m->foot_prim) {
    float sep = -3.0e38f;
    float sph;
    {
      float dc[3];
      for (int k = 0; k < 3; k++) {
        const float a1 = fP[0][k] + fR[0][3 * k] * m->foot_obb_center[0][0] + fR[0][3 * k + 1] * m->foot_obb_center[0][1] + fR[0][3 * k + 2] * m->foot_obb_center[0][2];
        const float a2 = fP[1][k] + fR[1][3 * k] * m->foot_obb_center[1][0] + fR[1][3 * k + 1] * m->foot_obb_center[1][1] + fR[1][3 * k + 2] * m->foot_obb_center[1][2];
        dc[k] = a2 - a1;
      }
      sph = sqrtf(dot3(dc, dc)) - m->foot_sphere_r[0] - m->foot_sphere_r[1];
    }
    float boxsep = -3.0e38f;
    if (__builtin_amdgcn_ballot_w64(!(sph > 0.0f)) != 0) {
      if (lane < 15) {
        float c1[3], c2[3], A1[9], A2[9], tt[3];
  #pragma unroll
        for (int f = 0; f < 2; f++) {
          float* cc = f ? c2 : c1; float* AA = f ? A2 : A1;
          for (int k = 0; k < 3; k++) cc[k] = fP[f][k] + fR[f][3 * k] * m->foot_obb_center[f][0] + fR[f][3 * k + 1] * m->foot_obb_center[f][1] + fR[f][3 * k + 2] * m->foot_obb_center[f][2];
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) AA[3 * i + j] = fR[f][3 * i] * m->foot_obb_axes[f][j] + fR[f][3 * i + 1] * m->foot_obb_axes[f][3 + j] + fR[f][3 * i + 2] * m->foot_obb_axes[f][6 + j];
        }
        for (int k = 0; k < 3; k++) tt[k] = c2[k] - c1[k];
        // axis of this lane: 0-2 faces of box 1, 3-5 faces of box 2, 6-14 edge x edge.  Written without nested selects / if-else chains: those were
        // compiled into basic blocks with exec-mask bookkeeping (seven if / else pairs in this block alone: round 6)
        constexpr unsigned IA = 0u | 1u << 2 | 2u << 4 | 0u << 6 | 0u << 8 | 0u << 10 | 0u << 12 | 0u << 14 | 0u << 16 | 1u << 18 | 1u << 20 | 1u << 22 | 2u << 24 | 2u << 26 | 2u << 28;
        constexpr unsigned IB = 0u | 0u << 2 | 0u << 4 | 0u << 6 | 1u << 8 | 2u << 10 | 0u << 12 | 1u << 14 | 2u << 16 | 0u << 18 | 1u << 20 | 2u << 22 | 0u << 24 | 1u << 26 | 2u << 28;
        const int ia = (int)((IA >> (2 * lane)) & 3u), ib = (int)((IB >> (2 * lane)) & 3u);
        float e1[3], e2[3], ax[3];
  #pragma unroll
        for (int k = 0; k < 3; k++) {
          e1[k] = A1[3 * k + 2]; e1[k] = ia == 1 ? A1[3 * k + 1] : e1[k]; e1[k] = ia == 0 ? A1[3 * k] : e1[k];
          e2[k] = A2[3 * k + 2]; e2[k] = ib == 1 ? A2[3 * k + 1] : e2[k]; e2[k] = ib == 0 ? A2[3 * k] : e2[k];
        }
        cross3(ax, e1, e2);
        const float n = sqrtf(dot3(ax, ax));
        const bool edge = lane >= 6, ok = !edge | (n >= 1e-6f);
        const float inv = n >= 1e-6f ? 1.0f / n : 0.0f;
  #pragma unroll
        for (int k = 0; k < 3; k++) { ax[k] *= inv; ax[k] = lane < 6 ? e2[k] : ax[k]; ax[k] = lane < 3 ? e1[k] : ax[k]; }
        {
          float r1 = 0, r2 = 0;
  #pragma unroll
          for (int k = 0; k < 3; k++) {
            const float b1[3] = {A1[k], A1[3 + k], A1[6 + k]}, b2[3] = {A2[k], A2[3 + k], A2[6 + k]};
            r1 += m->foot_obb_half[0][k] * fabsf(dot3(ax, b1));
            r2 += m->foot_obb_half[1][k] * fabsf(dot3(ax, b2));
          }
          const float sv = fabsf(dot3(tt, ax)) - r1 - r2;
          sep = ok ? sv : sep;
        }
      }
      boxsep = gmax<G>(sep);   // (inside the wave-uniform branch: cross-lane ops stay in uniform control flow)
    }
    const float best = sph > 0.0f ? sph : boxsep;
    if (lane < 4) {
      const int c = 8 + lane;
      // separated spheres / boxes -> inactive pair
      CDIST[c] = (lane == 0 && best > 0) ? best : 1.0f;
      CR[3 * c] = 0; CR[3 * c + 1] = 0; CR[3 * c + 2] = 0;
      if (lane == 0) SCR[S::S_MISC] = best;
    }
    // Overlapping boxes (feet about to touch; never seen in random-action rollouts): full SAT manifold, kept out of line
    // so that the hot path's register allocation does not pay for it.  Wave-uniform branch.
    const bool overlap = !(best > 0.0f);
    if (__builtin_amdgcn_ballot_w64(overlap) != 0) foot_foot_sat<S, G>(L, m, lane, overlap);
  }
  ODK_SYNC();
  ODK_PROF(8);

  // ---------------- P8: constraint rows: D, aref, contact wrenches
  // friction-loss rows: lane = row; limit rows: lane = dof that owns the limit; contact rows: rc = lane + t G
  FlSt fs;
  if constexpr (PRE) fs = hot.fs; else load_fl(fs, m, lane);
  float fl_f = 0.0f, fl_rf = 0.0f;
  if (lane < nfl) {
    fl_f = FRL[fs.dof]; fl_rf = fs.R * fl_f;
    ED[lane] = fs.D;
    AREF[lane] = -fs.b * QVEL[fs.dof];
  }
  float lim_sgn = 0.0f;
  if (st.d_lim_on) {
    const int r = nfl + st.d_limrow;
    const float qv = QPOS[st.d_qadr];
    const float dlo = qv - st.d_lo, dhi = st.d_hi - qv;
    const float pos = fminf(dlo, dhi);
    lim_sgn = dlo < dhi ? 1.0f : -1.0f;
    float D = 0, aref = 0;
    if (pos < 0) row_params(m->lim_imp[st.d_limrow], pos, m->lim_invweight[st.d_limrow], lim_sgn * QVEL[lane], D, aref);
    ED[r] = D; AREF[r] = aref;
  }
  // ---- equality rows (<equality><joint>: DevModel::neq; shapes with S::EQ).  BOTH dof lanes of a row evaluate it -- residual, c = p'(x2),
  // D and aref come out of LDS-resident vectors, nothing is handed over -- and each applies its own Jacobian entry (1 / -c); the row's
  // cost and line-search terms are counted once, by the lane of joint1's dof.  Always active (an equality pushes and pulls).
  bool eq_on = false, eq_first = false;
  int eq_i = 0, eq_j = -1, eq_r = 0;
  float eq_D = 0.0f, eq_aref = 0.0f, eq_c = 0.0f;
  if constexpr (S::EQ) {
    if (m->neq > 0 && st.d_on) {
      const int r = m->dof_eqrow[lane];
      if (r >= 0) {
        eq_on = true; eq_r = r; eq_i = m->eq_dof1[r]; eq_j = m->eq_dof2[r]; eq_first = lane == eq_i;
        const int a1 = m->eq_qadr1[r];
        const float* pc = m->eq_poly[r];
        float pos = QPOS[a1] - Q0[a1], vel = QVEL[eq_i];
        if (eq_j >= 0) {
          const int a2 = m->eq_qadr2[r];
          const float xx = QPOS[a2] - Q0[a2];
          pos -= pc[0] + xx * (pc[1] + xx * (pc[2] + xx * (pc[3] + xx * pc[4])));
          eq_c = pc[1] + xx * (2.0f * pc[2] + xx * (3.0f * pc[3] + xx * 4.0f * pc[4]));
          vel -= eq_c * QVEL[eq_j];
        } else {
          pos -= pc[0];
        }
        row_params(m->eq_imp[r], pos, m->eq_invweight[r], vel, eq_D, eq_aref);
      }
    }
  }
  // ---- path rows (<equality><connect | weld> between two bodies of one root-to-leaf path, or a body and the world: DevModel::neqp; shapes
  // with S::EQ).  Lane = row builds the row's residual and its two wrenches (one per body: a point force e_k at the anchor, or for a weld's
  // rotation rows the moment A_k of the error quaternion's derivative); lane = dof turns them into ITS Jacobian entries
  // J_ri = m1 (w1 . cdof_i) - m2 (w2 . cdof_i), kept in registers (pj) and in LDS (the Hessian's entry lanes read both dofs'); every J v
  // product is one reduction over the lanes.  Always active, quadratic (mjx constraint._efc_equality_connect / _weld as the oracle's
  // make_equality restates them: the impedance of all rows of a constraint from the NORM of its residual).
  float pj[EQP_ROWS], pjar_s[EQP_ROWS], pjar_w[EQP_ROWS];
  int np_rows = 0;
  float* EQW = L + S::O_EQP;
  if constexpr (S::EQ) {
    np_rows = m->eqp_nrow;      // (wave-uniform: one model per launch)
#pragma unroll
    for (int r = 0; r < EQP_ROWS; r++) { pj[r] = 0.0f; pjar_s[r] = 0.0f; pjar_w[r] = 0.0f; }
    if (np_rows > 0) {
      auto row_c = [&](int r) { return (m->neqp > 1 && r >= m->eqp_row0[1]) ? 1 : 0; };
      if (lane < np_rows) {
        const int c = row_c(lane), k = lane - m->eqp_row0[c], b1 = m->eqp_b1[c], b2 = m->eqp_b2[c];
        float q1[4], q2[4], p1[3], p2[3], t[3];
        for (int a = 0; a < 4; a++) { q1[a] = XQUAT[a * NB + b1]; q2[a] = XQUAT[a * NB + b2]; }
        qrot(t, q1, m->eqp_a1[c]); for (int a = 0; a < 3; a++) p1[a] = XPOS[a * NB + b1] + t[a];
        qrot(t, q2, m->eqp_a2[c]); for (int a = 0; a < 3; a++) p2[a] = XPOS[a * NB + b2] + t[a];
        float w1[6] = {0, 0, 0, 0, 0, 0}, w2[6] = {0, 0, 0, 0, 0, 0}, cpos;
        if (k < 3) {
          const float e[3] = {k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f};
          const float r1[3] = {p1[0] - ref[0], p1[1] - ref[1], p1[2] - ref[2]}, r2[3] = {p2[0] - ref[0], p2[1] - ref[1], p2[2] - ref[2]};
          cross3(w1, r1, e); cross3(w2, r2, e);
          for (int a = 0; a < 3; a++) { w1[3 + a] = e[a]; w2[3 + a] = e[a]; }
          cpos = (k == 0 ? p1[0] - p2[0] : (k == 1 ? p1[1] - p2[1] : p1[2] - p2[2]));
        } else {      // weld, rotation row k - 3: error quaternion conj(q2) q1 relpose, its axis part times torquescale; d/dt = 0.5 conj(q2) (0, w1 - w2) q1 relpose
          const int kk = k - 3;
          const float ts = m->eqp_ts[c];
          float quat[4], q2c[4] = {q2[0], -q2[1], -q2[2], -q2[3]}, qe[4];
          qmul(quat, q1, m->eqp_relq[c]);
          qmul(qe, q2c, quat);
          cpos = (kk == 0 ? qe[1] : (kk == 1 ? qe[2] : qe[3])) * ts;
          for (int a = 0; a < 3; a++) {
            const float ax[4] = {0.0f, a == 0 ? 1.0f : 0.0f, a == 1 ? 1.0f : 0.0f, a == 2 ? 1.0f : 0.0f};
            float q3[4], q4[4];
            qmul(q3, q2c, ax); qmul(q4, q3, quat);
            const float v = 0.5f * ts * (kk == 0 ? q4[1] : (kk == 1 ? q4[2] : q4[3]));
            w1[a] = v; w2[a] = v;
          }
        }
        for (int a = 0; a < 6; a++) { EQW[S::EQP_W + 12 * lane + a] = w1[a]; EQW[S::EQP_W + 12 * lane + 6 + a] = w2[a]; }
        EQW[S::EQP_POS + lane] = cpos;
      }
      ODK_SYNC();
      float cdv[6];
#pragma unroll
      for (int k = 0; k < 6; k++) cdv[k] = st.d_on ? CDOF[k * NR + st.d_red] : 0.0f;
      const int pmask = st.d_on ? m->dof_eqp[lane] : 0;
      const float qv_l = st.d_on ? QVEL[lane] : 0.0f;
      float vr[EQP_ROWS];
#pragma unroll
      for (int r = 0; r < EQP_ROWS; r++) {
        float J = 0.0f;
        if (r < np_rows) {
          const int c = row_c(r);
          const float* w = EQW + S::EQP_W + 12 * r;
          float d1 = 0.0f, d2 = 0.0f;
#pragma unroll
          for (int k = 0; k < 6; k++) { d1 = fmaf(w[k], cdv[k], d1); d2 = fmaf(w[6 + k], cdv[k], d2); }
          J = (((pmask >> (2 * c)) & 1) ? d1 : 0.0f) - (((pmask >> (2 * c + 1)) & 1) ? d2 : 0.0f);
          if (st.d_on) EQW[S::EQP_J + r * NV + lane] = J;
        }
        pj[r] = J; vr[r] = J * qv_l;
      }
      gsum_n<G, EQP_ROWS>(vr);
      if (lane < np_rows) {
        const int c = row_c(lane), k = lane - m->eqp_row0[c];
        const int r0 = m->eqp_row0[c], r1 = (c + 1 < m->neqp) ? m->eqp_row0[c + 1] : np_rows;
        float nrm = 0.0f;
        for (int r = r0; r < r1; r++) { const float x = EQW[S::EQP_POS + r]; nrm = fmaf(x, x, nrm); }
        nrm = sqrtf(nrm);
        float vel = 0.0f;
#pragma unroll
        for (int r = 0; r < EQP_ROWS; r++) vel = r == lane ? vr[r] : vel;
        float D, aref;
        row_params_imp(m->eqp_imp[c], EQW[S::EQP_POS + lane], nrm, m->eqp_invw[c][k < 3 ? 0 : 1], vel, D, aref);
        EQW[S::EQP_D + lane] = D; EQW[S::EQP_AREF + lane] = aref;
      }
      ODK_SYNC();
    }
  }
  // foot-foot rows (32..47) are skipped wave-wide unless some env has a penetrating foot-foot contact (D = 0 rows
  // are never read again: the solver gates on D > 0 and on the same wave-uniform flag)
  const bool ff_rows = __builtin_amdgcn_ballot_w64(fminf(fminf(CDIST[8], CDIST[9]), fminf(CDIST[10], CDIST[11])) < 0.0f) != 0;
  const bool prim_feet = !HF && m->foot_prim != 0;   // sphere / capsule feet: per-contact frames in S_FR (prim_contacts)
  const bool ell = S::CONE || (S::ELL && m->cone != 0);   // (wave-uniform; a compile-time constant for the duck's cone instantiations) elliptic friction cones
  for (int rc = lane; rc < S::NCROW; rc += G) {
    const int r = r0c + rc, c = rc >> 2, s = rc & 3, pair = c >> 2;
    if (rc >= 32 && !ff_rows) { ED[r] = 0.0f; AREF[r] = 0.0f; continue; }
    const float dist = CDIST[c];
    const float mu = CT[pair];
    const float fs = (s & 1) ? -mu : mu;
    // foot-foot frame: left in S_VF by the SAT routine; height-field floor: one frame per contact left in S_FR by P7
    const float* fr = (pair == 2 && dist < 0) ? SCR + S::S_VF : ((pair < 2 && (HF || prim_feet)) ? SCR + S::S_FR + 9 * c : CT + 33);
    const int td = 3 * (1 + (s >> 1));
    float dir[3] = {fr[0] + fs * fr[td], fr[1] + fs * fr[td + 1], fr[2] + fs * fr[td + 2]};
    if constexpr (S::ELL) {
      if (ell) {   // elliptic cones: the row lanes of a contact are the frame's axes (normal, tangent 1, tangent 2); the fourth lane has no row
        const float on3 = s < 3 ? 1.0f : 0.0f;
        const int ax = s < 3 ? 3 * s : 0;
        dir[0] = on3 * fr[ax]; dir[1] = on3 * fr[ax + 1]; dir[2] = on3 * fr[ax + 2];
      }
    }
    const float rr[3] = {CR[3 * c], CR[3 * c + 1], CR[3 * c + 2]};
    float ang[3];
    cross3(ang, rr, dir);
    float* wr = W + 6 * rc;
    wr[0] = ang[0]; wr[1] = ang[1]; wr[2] = ang[2]; wr[3] = dir[0]; wr[4] = dir[1]; wr[5] = dir[2];
    float D = 0, aref = 0;
    if (dist < 0) {
      float vel = 0;
      const int b2 = m->foot_body[pair == 0 ? 0 : 1];  // geom2's body: left foot for pair 0, right foot otherwise
#pragma unroll
      for (int k = 0; k < 6; k++) vel += wr[k] * CVEL[k * NB + b2];
      if (pair == 2) {
        const int b1 = m->foot_body[0];
#pragma unroll
        for (int k = 0; k < 6; k++) vel -= wr[k] * CVEL[k * NB + b1];
      }
      row_params(CT + 6 + 9 * pair, dist, CT[3 + pair], vel, D, aref);
      if constexpr (S::ELL) {
        if (ell) {
          // (mjx constraint._efc_contact_elliptic as the oracle restates it: every row takes the NORMAL row's impedance -- from the contact
          // distance, with the two bodies' translational weights, which is what pair_invweight holds for a model with cone = 1 --; the
          // tangents have no position term and the regulariser R_t = R_n / impratio)
          if (s == 0) { /* D, aref as computed */ }
          else if (s < 3) { float Dn, an; row_params(CT + 6 + 9 * pair, dist, CT[3 + pair], 0.0f, Dn, an); D = Dn * m->impratio; aref = -CT[6 + 9 * pair + 1] * vel; }
          else { D = 0.0f; aref = 0.0f; }
        }
      }
    }
    ED[r] = D;
    AREF[r] = aref;
  }
  ODK_SYNC();
  ODK_PROF(9);

  // ---------------- P9: Newton solver, one iteration (mjx solver.solve)
  gather_M();
  const float warm = st.d_on ? WARM[lane] : 0.0f;
  // foot twists of both candidates: VF = sum_{d above foot} cdof[d] qacc_smooth[d], VF2 likewise for the warmstart.
  // NOTE: v_readlane (ubcast / bcast) must stay in uniform control flow -- inside a divergent branch the source
  // lanes may be inactive and hold stale (e.g. un-reloaded spill) registers.
  // Twin dofs: the columns are per reduced dof, so the vectors enter as P^T v (pair sums): P^T qacc_smooth -> X (free until
  // the line search ends), P^T warmstart -> MA (free until the forces of the chosen point).
  const float* VBQ = QAS; const float* VBW = WARM;
  if constexpr (S::PAIRED) {
    if (st.r_on) { X[lane] = pair_sum<S>(QAS, m, lane); MA[lane] = pair_sum<S>(WARM, m, lane); }
    ODK_SYNC();
    VBQ = X; VBW = MA;
  }
  // component k of foot f's twist for the reduced vector V (LDS): the dofs above a foot are the six base dofs and the foot's
  // own serial chain (DevModel::foot_rchain, checked at load), so the sum is 6 + chain-length terms with compile-time
  // offsets -- no per-dof foot mask (a v_readlane each), no predicate.  Generic trees keep the masked loop over all dofs.
  auto foot_twist = [&](const float* V, int k, int f) -> float {
    float s = 0;
    if constexpr (S::CL > 0) {
      const int c0 = f ? m->foot_rchain_first[1] : m->foot_rchain_first[0], cl = f ? m->foot_rchain_len[1] : m->foot_rchain_len[0];
      const float* cd = CDOF + k * NR;
#pragma unroll
      for (int t = 0; t < 6; t++) s += cd[t] * V[t];
#pragma unroll
      for (int t = 0; t < S::CL; t++) { const float a = cd[c0 + t] * V[c0 + t]; s += t < cl ? a : 0.0f; }   // in-bounds of the LDS image even past the chain
    } else {
#pragma unroll
      for (int d = 0; d < NR; d++) {
        const int fm = ubcast(st.r_foot, d);
        const float vd = V[d];
        if ((fm >> f) & 1) s += CDOF[k * NR + d] * vd;
      }
    }
    return s;
  };
  {
    const int which = (lane / 12) & 1, f = (lane / 6) & 1, k = lane % 6;
    const float s = foot_twist(which ? VBW : VBQ, k, f);
    if (lane < 24) SCR[(which ? S::S_VF2 : S::S_VF) + 6 * f + k] = s;
  }
  // M * warmstart from the register row
  const float ma_w = mul_M(VBW, MV, warm);
  const float gw = st.d_on ? (ma_w - qfs) * (warm - qas) : 0.0f;
  ODK_SYNC();
  auto contact_jx = [&](int rc, const float* VF) -> float {
    const int pair = rc >> 4;
    const float* wr = W + 6 * rc;
    const float* v2 = VF + (pair == 0 ? 0 : 6);
    float s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += wr[k] * v2[k];
    if (pair == 2) {
#pragma unroll
      for (int k = 0; k < 6; k++) s -= wr[k] * VF[k];
    }
    return s;
  };
  auto quad_cost = [](float D, float jar, float& force) -> float {
    if (D > 0 && jar < 0) { force = -D * jar; return 0.5f * D * jar * jar; }
    force = 0;
    return 0.0f;
  };
  auto fl_cost = [&](float jar, float& force) -> float {
    // select form of the three-zone cost (linear outside +-R*f): as branches the two candidates were packed into
    // v_pk_mul pairs whose constant halves were hoisted above the substep loop and parked in scratch
    const float aj = fabsf(jar);
    const bool lin = aj >= fl_rf;
    force = lin ? (jar > 0 ? -fl_f : fl_f) : -fs.D * jar;
    return lin ? fl_f * (aj - 0.5f * fl_rf) : 0.5f * fs.D * jar * jar;
  };
  // Jaref and cost of both candidates; rows owned by this lane: friction row `lane`, limit row of dof `lane`, contact rows
  float cost_s = 0, cost_w = 0, jar_fl_s = 0, jar_fl_w = 0, jar_lim_s = 0, jar_lim_w = 0, fo;
  if (lane < nfl) {
    const float ar = AREF[lane];
    jar_fl_s = QAS[fs.dof] - ar; jar_fl_w = WARM[fs.dof] - ar;
    cost_s += fl_cost(jar_fl_s, fo); cost_w += fl_cost(jar_fl_w, fo);
  }
  float lim_D = 0.0f;
  if (st.d_lim_on) {
    const int r = nfl + st.d_limrow;
    lim_D = ED[r];
    const float ar = AREF[r];
    jar_lim_s = lim_sgn * qas - ar; jar_lim_w = lim_sgn * warm - ar;
    cost_s += quad_cost(lim_D, jar_lim_s, fo); cost_w += quad_cost(lim_D, jar_lim_w, fo);
  }
  float jar_eq_s = 0.0f, jar_eq_w = 0.0f;
  if constexpr (S::EQ) {
    if (eq_on) {   // J x = x_i - c x_j for both candidates; the quadratic cost once per row
      jar_eq_s = QAS[eq_i] - eq_aref; jar_eq_w = WARM[eq_i] - eq_aref;
      if (eq_j >= 0) { jar_eq_s -= eq_c * QAS[eq_j]; jar_eq_w -= eq_c * WARM[eq_j]; }
      if (eq_first) { cost_s += 0.5f * eq_D * jar_eq_s * jar_eq_s; cost_w += 0.5f * eq_D * jar_eq_w * jar_eq_w; }
    }
  }
  if constexpr (S::EQ) {
    if (np_rows > 0) {   // path rows: J x for both candidates (one reduction per row and candidate); the row's cost counted by lane = row
      float a9[EQP_ROWS], b9[EQP_ROWS];
#pragma unroll
      for (int r = 0; r < EQP_ROWS; r++) { a9[r] = pj[r] * qas; b9[r] = pj[r] * warm; }
      gsum_n<G, EQP_ROWS>(a9); gsum_n<G, EQP_ROWS>(b9);
#pragma unroll
      for (int r = 0; r < EQP_ROWS; r++) {
        if (r < np_rows) {
          const float ar = EQW[S::EQP_AREF + r], D = EQW[S::EQP_D + r];
          pjar_s[r] = a9[r] - ar; pjar_w[r] = b9[r] - ar;
          if (lane == r) { cost_s += 0.5f * D * pjar_s[r] * pjar_s[r]; cost_w += 0.5f * D * pjar_w[r] * pjar_w[r]; }
        }
      }
    }
  }
  const bool c_act[3] = {fminf(fminf(CDIST[0], CDIST[1]), fminf(CDIST[2], CDIST[3])) < 0, fminf(fminf(CDIST[4], CDIST[5]), fminf(CDIST[6], CDIST[7])) < 0,
                         fminf(fminf(CDIST[8], CDIST[9]), fminf(CDIST[10], CDIST[11])) < 0};
  // wave-uniform: some env of the wave has a penetrating foot-foot contact.  Without one, contact rows 32-47 have D = 0 in
  // both envs and their lane slot (G = 32: the second one) is skipped in the force sums and in the line search.
  const bool any_ff = __builtin_amdgcn_ballot_w64(c_act[2]) != 0;
  // contact rows of this lane (rc = lane + t G): D and Jaref stay in registers from here to the end of the line search
  constexpr int NCL = (S::NCROW + G - 1) / G;
  float cD[NCL], cjar[NCL], cjv[NCL];   // cjv: the warmstart candidate's Jaref until the choice below, J search in the line search
#pragma unroll
  for (int t = 0; t < NCL; t++) {
    const int rc = lane + t * G;
    cD[t] = 0.0f; cjar[t] = 0.0f; cjv[t] = 0.0f;
    if (t * G >= 32 && !any_ff) continue;   // foot-foot slot, nothing active in the wave
    const bool on = rc < S::NCROW;
    const int rcl = on ? rc : 0, r = r0c + rcl;
    const float D = on ? ED[r] : 0.0f;
    cD[t] = D;
    if (D > 0) {
      const float ar = AREF[r];
      cjar[t] = contact_jx(rcl, SCR + S::S_VF) - ar; cjv[t] = contact_jx(rcl, SCR + S::S_VF2) - ar;
      if (!ell) { cost_s += quad_cost(D, cjar[t], fo); cost_w += quad_cost(D, cjv[t], fo); }
    }
  }
  // elliptic cones: the four row lanes of a contact are a DPP quad -- every lane of the quad gets the contact's three Jaref (and the normal
  // row's D) by quad_perm broadcasts (uniform control flow), lane 0 of the quad counts the contact
  const float ell_mur = ell ? __builtin_amdgcn_rsqf(m->impratio) : 0.0f;     // mu_r = mu sqrt(R_t / R_n) = mu / sqrt(impratio); times mu below
  auto quad3 = [](float v, float* o) { o[0] = ODK_DPP(v, 0x00, 0xF); o[1] = ODK_DPP(v, 0x55, 0xF); o[2] = ODK_DPP(v, 0xAA, 0xF); };
  if constexpr (S::ELL) {
    if (ell) {
#pragma unroll
      for (int t = 0; t < NCL; t++) {
        if (t * G >= 32 && !any_ff) continue;
        const int rc = lane + t * G;
        float xs[3], xw[3];
        quad3(cjar[t], xs); quad3(cjv[t], xw);
        const float Dn = ODK_DPP(cD[t], 0x00, 0xF), Dt = Dn * m->impratio;
        const float mu = CT[(rc < S::NCROW ? rc : 0) >> 4], mur = mu * ell_mur;
        const float one = ((rc & 3) == 0 && rc < S::NCROW) ? 1.0f : 0.0f;
        const float Dm = cone_dm(Dn, mur);
        const ConeEv es = cone_ev(Dn, Dm, mu, mur, xs[0], xs[1], xs[2]), ew = cone_ev(Dn, Dm, mu, mur, xw[0], xw[1], xw[2]);
        cost_s += one * cone_cost(es, Dn, Dt, xs[0], xs[1], xs[2]); cost_w += one * cone_cost(ew, Dn, Dt, xw[0], xw[1], xw[2]);
      }
    }
  }
  float gauss_w;
  { float r3[3] = {cost_s, gw, cost_w}; gsum_n<G, 3>(r3); cost_s = r3[0]; gauss_w = 0.5f * r3[1]; cost_w = r3[2] + gauss_w; }
  const bool use_warm = cost_w < cost_s;
  const float gauss = use_warm ? gauss_w : 0.0f;
  const float x = use_warm ? warm : qas;
  const float ma = use_warm ? ma_w : qfs;
  const float jar_fl = use_warm ? jar_fl_w : jar_fl_s, jar_lim = use_warm ? jar_lim_w : jar_lim_s;
  const float jar_eq = use_warm ? jar_eq_w : jar_eq_s;
  if constexpr (S::EQ) {
#pragma unroll
    for (int r = 0; r < EQP_ROWS; r++) pjar_s[r] = use_warm ? pjar_w[r] : pjar_s[r];      // (from here on: the chosen point's Jaref of the path rows)
  }
#pragma unroll
  for (int t = 0; t < NCL; t++) {
    cjar[t] = use_warm ? cjv[t] : cjar[t];
    const int rc = lane + t * G;
    if (rc < S::NCROW) JAR[r0c + rc] = cjar[t];   // debug image only
  }
  ODK_PROF(10);
  // forces of the chosen point: friction / limit rows stay in registers, contact forces -> JV
  float f_fl = 0, f_lim = 0;
  bool quad_fl = false;
  if (lane < nfl) { fl_cost(jar_fl, f_fl); quad_fl = (jar_fl > -fl_rf) && (jar_fl < fl_rf); }
  if (st.d_lim_on) quad_cost(lim_D, jar_lim, f_lim);
  // friction-row quantities are owned by lane = row; the dof that needs them is another lane: hand over through LDS
  if (lane < nfl) { MV[fs.dof] = f_fl; MA[fs.dof] = quad_fl ? fs.D : 0.0f; }
  // Foot wrench sums FF_f = sum_r w_r f_r and 6x6 blocks K_f = sum_r D_r [active] w_r w_r^T.  Contact row rc sits in
  // lane rc: rows 0-15 (left foot), 16-31 (right foot) and 32-47 (foot-foot) are exactly the 16-lane DPP rows, so
  // each sum is four DPP adds; lane 0 of a row stores its block.  (G = 32: the foot-foot rows are a second slot.)
  {
    constexpr int NSLOT = (S::NCROW + G - 1) / G;
#pragma unroll
    for (int t = 0; t < NSLOT; t++) {
      if (t * G >= 32 && !any_ff) continue;   // foot-foot slot with nothing active anywhere in the wave
      const int rc = lane + t * G;
      const bool on = rc < S::NCROW;
      const int rcl = on ? rc : 0, r = r0c + rcl;
      float w[6];
#pragma unroll
      for (int k = 0; k < 6; k++) w[k] = W[6 * rcl + k];
      const float D = cD[t], jar = cjar[t];
      const float act = (on && D > 0 && jar < 0) ? D : 0.0f;
      float fr = -act * jar;
      float v[27];
      float u[6];      // (J^T of this row's line of the contact's Hessian block: act w for a pyramid row)
#pragma unroll
      for (int k = 0; k < 6; k++) u[k] = act * w[k];
      if constexpr (S::ELL) {
        if (ell) {   // this lane's row s of its contact: force f_s, and u = sum_b C[s][b] w_b over the contact's three wrenches (LDS)
          float xs[3], C[3];
          quad3(jar, xs);
          const float Dn = ODK_DPP(D, 0x00, 0xF), Dt = Dn * m->impratio;
          const float mu = CT[rcl >> 4], mur = mu * ell_mur;
          const int sr = on ? (rcl & 3) : 3;
          const ConeEv e = cone_ev(Dn, cone_dm(Dn, mur), mu, mur, xs[0], xs[1], xs[2]);
          fr = cone_force(e, Dn, Dt, mu, mur, xs[0], xs[1], xs[2], sr);
          cone_hess_row(e, Dn, Dt, mu, mur, sr, C);
          const float* w0 = W + 6 * (rcl & ~3);
#pragma unroll
          for (int k = 0; k < 6; k++) u[k] = C[0] * w0[k] + C[1] * w0[6 + k] + C[2] * w0[12 + k];
        }
      }
#pragma unroll
      for (int k = 0; k < 6; k++) v[k] = fr * w[k];
      {
        int q = 6;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const float aw = act * w[a];
#pragma unroll
          for (int b2 = a; b2 < 6; b2++) {
            if constexpr (S::ELL) v[q++] = ell ? w[a] * u[b2] : aw * w[b2];      // (the pyramid rows keep their rounding: (D w_a) w_b)
            else v[q++] = aw * w[b2];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 27; q++) {
        float x = v[q];
        x += ODK_DPP(x, 0xB1, 0xF); x += ODK_DPP(x, 0x4E, 0xF); x += ODK_DPP(x, 0x141, 0xF); x += ODK_DPP(x, 0x140, 0xF);
        v[q] = x;
      }
      if (on && (lane & 15) == 0) {
        const int blk = rc >> 4;   // 0 left, 1 right, 2 foot-foot
        float* FFb = SCR + (blk < 2 ? S::S_FF + 6 * blk : S::S_FFX);
        float* Kb = SCR + S::S_K + 36 * blk;
#pragma unroll
        for (int k = 0; k < 6; k++) FFb[k] = v[k];
        int q = 6;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int b2 = a; b2 < 6; b2++) { Kb[6 * a + b2] = v[q]; Kb[6 * b2 + a] = v[q]; q++; }
      }
    }
    ODK_SYNC();
    if (!any_ff) {
      for (int t2 = lane; t2 < 36; t2 += G) SCR[S::S_K + 72 + t2] = 0.0f;
    } else {   // rows of the foot-foot pair act on both feet: K_L += K_X, K_R += K_X, FF_L -= FF_X, FF_R += FF_X
      for (int t2 = lane; t2 < 36; t2 += G) { const float kx = SCR[S::S_K + 72 + t2]; SCR[S::S_K + t2] += kx; SCR[S::S_K + 36 + t2] += kx; }
      if (lane < 6) { const float fx = SCR[S::S_FFX + lane]; SCR[S::S_FF + lane] -= fx; SCR[S::S_FF + 6 + lane] += fx; }
    }
  }
  ODK_SYNC();
  const bool ff_active = c_act[2] && (SCR[S::S_K + 72 + 21] != 0.0f || SCR[S::S_K + 72 + 28] != 0.0f || SCR[S::S_K + 72 + 35] != 0.0f);
  float grad = 0.0f;
  if (st.d_on) {
    const int i = lane;
    float qc = 0, hdiag_extra = 0.0f;
    if (st.d_flrow >= 0) { qc += MV[i]; hdiag_extra += MA[i]; }
    if (st.d_lim_on) { qc += lim_sgn * f_lim; if (lim_D > 0 && jar_lim < 0) hdiag_extra += lim_D; }
    if constexpr (S::EQ) {
      if (eq_on) {   // J^T f and J^T D J of the equality row: own entry 1 (joint1's dof) or -c (joint2's); the off-diagonal term through S_EQ
        const float je = eq_first ? 1.0f : -eq_c;
        qc += je * (-eq_D * jar_eq);
        hdiag_extra += eq_D * je * je;
        if (eq_first) SCR[S::S_EQ + eq_r] = -eq_D * eq_c;
      }
      if (np_rows > 0) {   // J^T f of the path rows (their J^T D J goes into the Hessian entries directly: hess_entry)
#pragma unroll
        for (int r = 0; r < EQP_ROWS; r++) if (r < np_rows) qc += pj[r] * (-EQW[S::EQP_D + r] * pjar_s[r]);
      }
    }
    float cd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cd[k] = CDOF[k * NR + st.d_red];
#pragma unroll
    for (int f = 0; f < 2; f++)
      if ((st.d_foot >> f) & 1) {
#pragma unroll
        for (int k = 0; k < 6; k++) qc += cd[k] * SCR[S::S_FF + 6 * f + k];
      }
    grad = ma - qfs - qc;
    // T_f[column] = K_f cdof[column] (a twin shares its main dof's column).  Floating base + chains: done below, one (foot,
    // column) pair per lane
    if (S::CL == 0 && st.d_tkind != 2) {
#pragma unroll
      for (int f = 0; f < 2; f++) {
        float* T = f ? BUF6B : BUF6;
        const bool on = ((st.d_foot >> f) & 1) && (c_act[f] || c_act[2]);
#pragma unroll
        for (int a = 0; a < 6; a++) {
          float s = 0;
          if (on) {
#pragma unroll
            for (int b2 = 0; b2 < 6; b2++) s += SCR[S::S_K + 36 * f + 6 * a + b2] * cd[b2];
          }
          T[a * NR + st.d_red] = s;
        }
      }
    }
    JV[i] = hdiag_extra;  // JV[0..NV) lies below the contact rows: free scratch for the per-dof diagonal addend
  }
  if constexpr (S::CL > 0) {
    // The columns the Hessian entries read are those of the dofs above a foot: the six base dofs and the foot's own chain,
    // 11 per foot at most.  Lane 11 f + idx computes column idx of foot f (36 FMAs) instead of every dof lane doing both
    // feet (72); the other columns of BUF6 / BUF6B are never read (hess_entry tests the foot masks).
    const int f = lane >= 6 + S::CL ? 1 : 0, idx = lane - (6 + S::CL) * f;
    const int c0 = f ? m->foot_rchain_first[1] : m->foot_rchain_first[0], cl = f ? m->foot_rchain_len[1] : m->foot_rchain_len[0];
    const bool lane_on = lane < 2 * (6 + S::CL) && idx - 6 < cl;
    const int col = lane_on ? (idx < 6 ? idx : c0 + idx - 6) : 0;
    const bool on = lane_on && ((f ? c_act[1] : c_act[0]) || c_act[2]);
    float cdc[6];
#pragma unroll
    for (int b2 = 0; b2 < 6; b2++) cdc[b2] = CDOF[b2 * NR + col];
    const float* Kf = SCR + S::S_K + 36 * f;
    float* T = f ? BUF6B : BUF6;
#pragma unroll
    for (int a = 0; a < 6; a++) {
      float s0 = 0.0f;
#pragma unroll
      for (int b2 = 0; b2 < 6; b2++) s0 = fmaf(Kf[6 * a + b2], cdc[b2], s0);
      if (lane_on) T[a * NR + col] = on ? s0 : 0.0f;
    }
  }
  ODK_SYNC();
  ODK_PROF(11);
  // wave-uniform: some env has an active foot-foot row -- or the model has a connect / weld between the two foot chains (a closed loop): both need
  // Hessian entries between the two legs, which the virtual tree's layout has
  const bool any_ffa = __builtin_amdgcn_ballot_w64(ff_active) != 0 || (S::EQ && m->eqp_cross != 0);
  // contact / diagonal terms of one reduced Hessian entry e (packed: DevModel::R_ent) on top of the inertia value v
  auto hess_entry = [&](int e, float v) -> float {
    const int i = e & 31, j = (e >> 5) & 31, fi = (e >> 10) & 3, fj = (e >> 12) & 3;
    if ((e >> 14) & 1) {   // diagonal: friction-loss / limit rows; a pair's two diagonals enter as their series term
      const int u = (e >> 16) & 31;
      v += (S::PAIRED && ((e >> 15) & 1)) ? pair_einv(ARM, JV, u) : JV[u];
    }
    if constexpr (S::EQ) {
      for (int r = 0; r < m->neq; r++) v += (e & 0x3FF) == m->eq_key[r] ? SCR[S::S_EQ + r] : 0.0f;
      for (int r = 0; r < np_rows; r++) v = fmaf(EQW[S::EQP_D + r] * EQW[S::EQP_J + r * NV + i], EQW[S::EQP_J + r * NV + j], v);      // path rows: D J_ri J_rj
    }
    const int both = fi & fj;
    if (both) {
      float cj[6];
#pragma unroll
      for (int k = 0; k < 6; k++) cj[k] = CDOF[k * NR + j];
      if (both & 1) {
#pragma unroll
        for (int k = 0; k < 6; k++) v += cj[k] * BUF6[k * NR + i];
      }
      if (both & 2) {
#pragma unroll
        for (int k = 0; k < 6; k++) v += cj[k] * BUF6B[k * NR + i];
      }
    }
    return v;
  };
  if (st.d_on) { MA[lane] = grad; GRAD[lane] = grad; }   // MA: kept for the debug image (gradient at the starting point)
  // right-hand side of the reduced system on the reduced-dof lanes (twin-free model: the gradient itself)
  float rhs = grad;
  if constexpr (S::PAIRED) {
    ODK_SYNC();
    rhs = pair_rhs<S>(GRAD, ARM, JV, m, lane);
  }
  float search;
  if (!any_ffa) {
    // ---- common case: no foot-foot coupling -> the Hessian has the reduced inertia's own tree pattern
    // hess_entry, branch-free and batched like the inertia entries (P4): table + M reads, then the operands, then the sums.
    // A one-foot entry takes its K cdof column from a per-lane buffer select; entries between two base dofs (both feet) are
    // the first 21 of the layout -- rows 0 .. 5 -- so the second foot's term exists in the first pass only.  Unused terms are
    // dropped by selects, never multiplied by zero (their operands may be unwritten).
    {
      static_assert(G >= 21, "base x base entries must sit in the first pass");
      int ee[ST::NME]; float mm[ST::NME], hv[ST::NME];
#pragma unroll
      for (int t = 0; t < ST::NME; t++) { const int p = lane + t * G, pc = p < S::NMR ? p : 0; ee[t] = RT[pc]; mm[t] = M[pc]; }
#pragma unroll
      for (int t = 0; t < ST::NME; t++) {
        const int e = ee[t], i = e & 31, j = (e >> 5) & 31, both = (e >> 10) & (e >> 12) & 3, u = (e >> 16) & 31;
        float dterm = JV[u];
        if constexpr (S::PAIRED) { const float pe = pair_einv(ARM, JV, u); dterm = ((e >> 15) & 1) ? pe : dterm; }
        float cj[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cj[k] = CDOF[k * NR + j];
        const float* T1 = (both & 1) ? BUF6 : BUF6B;
        float s1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) s1 = fmaf(cj[k], T1[k * NR + i], s1);
        float v = mm[t] + (((e >> 14) & 1) ? dterm : 0.0f) + (both ? s1 : 0.0f);
        if constexpr (S::EQ) {
          for (int r = 0; r < m->neq; r++) v += (e & 0x3FF) == m->eq_key[r] ? SCR[S::S_EQ + r] : 0.0f;
          for (int r = 0; r < np_rows; r++) v = fmaf(EQW[S::EQP_D + r] * EQW[S::EQP_J + r * NV + i], EQW[S::EQP_J + r * NV + j], v);
        }
        if (t == 0) {
          float s2 = 0.0f;
#pragma unroll
          for (int k = 0; k < 6; k++) s2 = fmaf(cj[k], BUF6B[k * NR + i], s2);
          v += both == 3 ? s2 : 0.0f;
        }
        hv[t] = v;
      }
#pragma unroll
      for (int t = 0; t < ST::NME; t++) { const int p = lane + t * G; if (p < S::NMR) HL[p] = hv[t]; }
    }
    if constexpr (S::PAIRED) { if (st.r_on) MV[lane] = rhs; }
    ODK_SYNC();
    ODK_PROF(12);
    if constexpr (S::PAIRED) {
      chain_solve<S, G>(HL, MV, SCR + S::S_K, BUF6, st.cs_pk, lane);
      ODK_PROF(13);
      search = -pair_expand<S, G>(MV, GRAD, grad, ARM, JV, st, lane);
    } else if constexpr (S::CL > 0) {
      chain_solve<S, G>(HL, GRAD, SCR + S::S_K, BUF6, st.cs_pk, lane);
      ODK_PROF(13);
      search = st.d_on ? -GRAD[lane] : 0.0f;
    } else {   // generic tree: the inertia's own (shallower) row layout instead of the virtual tree's
      if (m->nrchain > 0 && m->rchain_first[0] == 6)   // floating base + serial chains: chain-parallel elimination
        factor_chains<G, S::DT, NV, S::DT - 5, 6>(HL, lane, st.r_on, st.r_depth, st.r_Madr, st.ch_first, st.ch_len, st.r_descmask, st.r_depth, st.r_Madr);
      else
        factor_rows<G, S::DT, NV>(HL, lane, st.r_on, st.r_depth, st.r_Madr, st.r_descmask, st.r_depth, st.r_Madr);
      ODK_PROF(13);
      search = -solve_rows<G, NV>(HL, grad, lane, st.r_on, st.r_depth, st.r_Madr, st.r_ancmask, st.r_descmask, st.r_depth, st.r_Madr);
    }
    ODK_PROF(14);
  } else {
    // ---- an env of the wave has an active foot-foot row: Hessian on the reduced VIRTUAL tree (second leg below the first
    // foot), whose statics are fetched here (rare path: they do not occupy registers elsewhere)
    const int rl = st.r_on ? lane : 0;
    const int v_depth = st.r_on ? m->rv_depth[rl] : 0, v_Madr = m->rv_Madr[rl];
    const int v_ancmask = st.r_on ? m->rv_ancmask[rl] : 0, v_descmask = st.r_on ? m->rv_descmask[rl] : 0;
    int hent[ST::NHE];
#pragma unroll
    for (int t = 0; t < ST::NHE; t++) hent[t] = load_rhent(m, lane + t * G, S::NHR);
#pragma unroll
    for (int t = 0; t < ST::NHE; t++) {
      const int e = hent[t];
      if (e >= 0) {
        const int i = e & 31, j = (e >> 5) & 31, fi = (e >> 10) & 3, fj = (e >> 12) & 3, src = (e >> 21) - 1;
        float v = hess_entry(e, src >= 0 ? M[src] : 0.0f);
        if (ff_active) {  // cross terms -(A_R^T K_X A_L + A_L^T K_X A_R); rare
          const float wgt = (float)(((fi >> 1) & 1) && (fj & 1)) + (float)((fi & 1) && ((fj >> 1) & 1));
          if (wgt != 0.0f) {
            float sx = 0;
            for (int a2 = 0; a2 < 6; a2++)
              for (int b2 = 0; b2 < 6; b2++) sx += CDOF[a2 * NR + i] * SCR[S::S_K + 72 + 6 * a2 + b2] * CDOF[b2 * NR + j];
            v -= wgt * sx;
          }
        }
        HL[lane + t * G] = v;
      }
    }
    ODK_SYNC();
    ODK_PROF(12);
    factor_rows<G, S::DVR, NR>(HL, lane, st.r_on, v_depth, v_Madr, v_descmask, v_depth, v_Madr);
    ODK_PROF(13);
    const float z = solve_rows<G, NR>(HL, rhs, lane, st.r_on, v_depth, v_Madr, v_ancmask, v_descmask, v_depth, v_Madr);
    if constexpr (S::PAIRED) {
      if (st.r_on) MV[lane] = z;
      ODK_SYNC();
      search = -pair_expand<S, G>(MV, GRAD, grad, ARM, JV, st, lane);
    } else {
      search = -z;
    }
    ODK_PROF(14);
  }

  // ---- line search (mjx solver._linesearch)
  ODK_SYNC();   // every lane has read its GRAD / JV inputs
  if (st.d_on) GRAD[lane] = search;
  ODK_SYNC();
  const float* VBS = GRAD;   // P^T search (twin dofs: pair sums in X, free until the step is taken)
  if constexpr (S::PAIRED) {
    if (st.r_on) X[lane] = pair_sum<S>(GRAD, m, lane);
    ODK_SYNC();
    VBS = X;
  }
  const float mv = mul_M(VBS, MV, search);
  float sn = st.d_on ? search * search : 0.0f, qg1 = st.d_on ? search * (ma - qfs) : 0.0f, qg2 = st.d_on ? 0.5f * search * mv : 0.0f;
  { float r3[3] = {sn, qg1, qg2}; gsum_n<G, 3>(r3); sn = r3[0]; qg1 = r3[1]; qg2 = r3[2]; }
  ODK_SYNC();
  {  // foot twists of the search direction
    const float s = foot_twist(VBS, lane % 6, (lane / 6) & 1);
    if (lane < 12) SCR[S::S_VF + lane] = s;
  }
  ODK_SYNC();
  float jv_fl = 0, jv_lim = 0;
  if (lane < nfl) jv_fl = GRAD[fs.dof];
  if (st.d_lim_on) jv_lim = lim_sgn * search;
  float jv_eq = 0.0f, eq_Dw = 0.0f;      // the equality row's J search and its D on the lane that counts the row (0 elsewhere: exact zeros below)
  float pr_D = 0.0f, pr_jar = 0.0f, pr_jv = 0.0f;      // the path row counted by this lane (lane = row)
  if constexpr (S::EQ) {
    if (eq_on && eq_first) { jv_eq = GRAD[eq_i] - (eq_j >= 0 ? eq_c * GRAD[eq_j] : 0.0f); eq_Dw = eq_D; }
    if (np_rows > 0) {
      float s9[EQP_ROWS];
#pragma unroll
      for (int r = 0; r < EQP_ROWS; r++) s9[r] = st.d_on ? pj[r] * search : 0.0f;
      gsum_n<G, EQP_ROWS>(s9);
#pragma unroll
      for (int r = 0; r < EQP_ROWS; r++) if (r == lane && r < np_rows) { pr_D = EQW[S::EQP_D + r]; pr_jar = pjar_s[r]; pr_jv = s9[r]; }
    }
  }
  // J search of this lane's contact rows
#pragma unroll
  for (int t = 0; t < NCL; t++) {
    const int rc = lane + t * G;
    const bool on = rc < S::NCROW;
    cjv[t] = 0.0f;
    if (!(t * G >= 32 && !any_ff)) cjv[t] = (on && cD[t] > 0) ? contact_jx(rc, SCR + S::S_VF) : 0.0f;
    if (on) JV[r0c + rc] = cjv[t];  // debug image only
  }
  ODK_PROF(15);
  // elliptic cones: the contact's three Jaref and J search in every lane of its quad, once for the whole line search
  float ls_x[NCL][3], ls_v[NCL][3], ls_Dn[NCL], ls_Dt[NCL], ls_Dm[NCL], ls_mu[NCL], ls_mur[NCL], ls_on[NCL];
  if constexpr (S::ELL) {
    if (ell) {
#pragma unroll
      for (int t = 0; t < NCL; t++) {
        const int rc = lane + t * G;
        quad3(cjar[t], ls_x[t]); quad3(cjv[t], ls_v[t]);
        ls_Dn[t] = ODK_DPP(cD[t], 0x00, 0xF);
        ls_mu[t] = CT[(rc < S::NCROW ? rc : 0) >> 4];
        ls_mur[t] = ls_mu[t] * ell_mur; ls_Dt[t] = ls_Dn[t] * m->impratio; ls_Dm[t] = cone_dm(ls_Dn[t], ls_mur[t]);
        ls_on[t] = ((rc & 3) < 3 && rc < S::NCROW) ? 1.0f : 0.0f;      // the three lanes of a contact's quad that evaluate a step size each
      }
    }
  }
  const float gtol = m->tolerance * m->ls_tolerance * sqrtf(sn) * m->meaninertia * (float)(NV > 1 ? NV : 1);
  // Three step sizes at once.  The bracketing iterations only steer on the first and second derivative along the search, so
  // they evaluate those alone (COST = false: two sums per step size); the costs that pick the final step -- at lo, hi and 0 --
  // are one evaluation at the end (COST = true: one sum per step size, the quadratic combined per lane before the reduction).
  auto ls_eval = [&](auto cost_tag, const float* al, float* cost, float* d0, float* d1) {
    constexpr bool COST = decltype(cost_tag)::value;
    // branch-free: a lane without a row of some kind holds D = 0 or f = jar = jv = 0 there, so its terms are exact zeros (the
    // divergent `if`s around them cost more exec-mask traffic than the few idle multiplies); an active quadratic row enters
    // as weight 1 (one select + FMAs instead of a select + add per term)
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    {
      const float D = fs.D, jar = jar_fl, jv = jv_fl;   // lanes >= nfl: f = rf = jar = jv = 0 -> every piece below is 0
      const float q0 = 0.5f * D * jar * jar, q1 = D * jv * jar, q2 = 0.5f * D * jv * jv;
      const float lo0 = fl_f * (-0.5f * fl_rf - jar), lo1 = -fl_f * jv, hi0 = fl_f * (-0.5f * fl_rf + jar), hi1 = fl_f * jv;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float xx = jar + al[a] * jv;
        const bool lo = xx <= -fl_rf, hi = xx >= fl_rf;
        if constexpr (COST) acc[3 * a] = lo ? lo0 : (hi ? hi0 : q0);
        acc[3 * a + 1] = lo ? lo1 : (hi ? hi1 : q1); acc[3 * a + 2] = (lo || hi) ? 0.0f : q2;
      }
    }
    {
      const float D = lim_D, jar = jar_lim, jv = jv_lim;   // lanes without a limit row: D = 0
      const float q0 = 0.5f * D * jar * jar, q1 = D * jv * jar, q2 = 0.5f * D * jv * jv;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float w = jar + al[a] * jv < 0 ? 1.0f : 0.0f;
        if constexpr (COST) acc[3 * a] = fmaf(w, q0, acc[3 * a]);
        acc[3 * a + 1] = fmaf(w, q1, acc[3 * a + 1]); acc[3 * a + 2] = fmaf(w, q2, acc[3 * a + 2]);
      }
    }
    if constexpr (S::EQ) {   // equality row (joint coupling) and path row (connect / weld) of this lane: quadratic at every step size
      const float D = eq_Dw, jar = jar_eq, jv = jv_eq;
      const float q0 = 0.5f * D * jar * jar + 0.5f * pr_D * pr_jar * pr_jar, q1 = D * jv * jar + pr_D * pr_jv * pr_jar, q2 = 0.5f * D * jv * jv + 0.5f * pr_D * pr_jv * pr_jv;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        if constexpr (COST) acc[3 * a] += q0;
        acc[3 * a + 1] += q1; acc[3 * a + 2] += q2;
      }
    }
#pragma unroll
    for (int t = 0; t < NCL; t++) {
      if (t * G >= 32 && !any_ff) continue;
      if constexpr (S::ELL) {
        if (ell) {   // the contact's exact cost / derivatives at the three step sizes: lane s of the contact's quad evaluates step size s (one
                     // cone evaluation per lane instead of three; the fourth lane idles) and adds to the sums of ITS step size -- the reduction
                     // over the lanes does the rest.  The sums the callers form are d0 = 2 al t2 + t1 and d1 = 2 t2: a term (d0c, d1c) enters
                     // as t1 += d0c - al d1c, t2 += d1c / 2
          const int sa = lane & 3;
          const float als = sa == 0 ? al[0] : (sa == 1 ? al[1] : al[2]);
          const float xx[3] = {ls_x[t][0] + als * ls_v[t][0], ls_x[t][1] + als * ls_v[t][1], ls_x[t][2] + als * ls_v[t][2]};
          float cc, d0c, d1c;
          cone_line(ls_Dn[t], ls_Dt[t], ls_Dm[t], ls_mu[t], ls_mur[t], xx, ls_v[t], cc, d0c, d1c);
          const float t1c = d0c - als * d1c, t2c = 0.5f * d1c;
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const float w = sa == a ? ls_on[t] : 0.0f;
            if constexpr (COST) acc[3 * a] = fmaf(w, cc, acc[3 * a]);
            else { acc[3 * a + 1] = fmaf(w, t1c, acc[3 * a + 1]); acc[3 * a + 2] = fmaf(w, t2c, acc[3 * a + 2]); }
          }
          continue;
        }
      }
      const float D = cD[t], jar = cjar[t], jv = cjv[t];   // inactive rows: D = 0
      const float q0 = 0.5f * D * jar * jar, q1 = D * jv * jar, q2 = 0.5f * D * jv * jv;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float w = jar + al[a] * jv < 0 ? 1.0f : 0.0f;
        if constexpr (COST) acc[3 * a] = fmaf(w, q0, acc[3 * a]);
        acc[3 * a + 1] = fmaf(w, q1, acc[3 * a + 1]); acc[3 * a + 2] = fmaf(w, q2, acc[3 * a + 2]);
      }
    }
    if constexpr (COST) {
      float c[3];
#pragma unroll
      for (int a = 0; a < 3; a++) c[a] = fmaf(al[a], fmaf(al[a], acc[3 * a + 2], acc[3 * a + 1]), acc[3 * a]);
      gsum_n<G, 3>(c);
#pragma unroll
      for (int a = 0; a < 3; a++) cost[a] = c[a] + fmaf(al[a], fmaf(al[a], qg2, qg1), gauss);
    } else {
      float r[6] = {acc[1], acc[2], acc[4], acc[5], acc[7], acc[8]};
      gsum_n<G, 6>(r);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float t1 = r[2 * a] + qg1, t2 = r[2 * a + 1] + qg2;
        d0[a] = 2.0f * al[a] * t2 + t1;
        d1[a] = 2.0f * t2 + (t2 == 0.0f ? MINVAL_F : 0.0f);
      }
    }
  };
  using Deriv = std::integral_constant<bool, false>; using Cost = std::integral_constant<bool, true>;
  auto sdiv = [](float a, float b) { return b == 0.0f ? 0.0f : a * __builtin_amdgcn_rcpf(b); };   // Newton step of the 1-D search: 1 ulp is plenty
  float al[3] = {0, 0, 0}, cs[3], e0[3], e1[3];
  ls_eval(Deriv(), al, cs, e0, e1);   // three equal step sizes: the compiler folds them into one evaluation
  const float p0_d0 = e0[0], p0_d1 = e1[0];
  al[0] = -sdiv(p0_d0, p0_d1); al[1] = al[0]; al[2] = al[0];
  ls_eval(Deriv(), al, cs, e0, e1);
  float lo_a, lo_d0, lo_d1, hi_a, hi_d0, hi_d1;
  if (e0[0] < p0_d0) { lo_a = al[0]; lo_d0 = e0[0]; lo_d1 = e1[0]; hi_a = 0; hi_d0 = p0_d0; hi_d1 = p0_d1; }
  else { hi_a = al[0]; hi_d0 = e0[0]; hi_d1 = e1[0]; lo_a = 0; lo_d0 = p0_d0; lo_d1 = p0_d1; }
  bool swap = true;
  for (int it = 0; it < m->ls_iterations; it++) {
    bool done = !swap || (lo_d0 < 0 && lo_d0 > -gtol) || (hi_d0 > 0 && hi_d0 < gtol);
    if (done) break;
    al[0] = lo_a - sdiv(lo_d0, lo_d1); al[1] = hi_a - sdiv(hi_d0, hi_d1); al[2] = 0.5f * (lo_a + hi_a);
    ls_eval(Deriv(), al, cs, e0, e1);
    // lo_next = 0, hi_next = 1, mid = 2
    const bool s1 = (lo_d0 > 0) || (lo_d0 < e0[0]);
    if (s1) { lo_a = al[0]; lo_d0 = e0[0]; lo_d1 = e1[0]; }
    const bool s2 = (e0[2] < 0) && (lo_d0 < e0[2]);
    if (s2) { lo_a = al[2]; lo_d0 = e0[2]; lo_d1 = e1[2]; }
    const bool s3 = (e0[1] < 0) && (lo_d0 < e0[1]);
    if (s3) { lo_a = al[1]; lo_d0 = e0[1]; lo_d1 = e1[1]; }
    const bool s4 = (hi_d0 < 0) || (hi_d0 > e0[1]);
    if (s4) { hi_a = al[1]; hi_d0 = e0[1]; hi_d1 = e1[1]; }
    const bool s5 = (e0[2] > 0) && (hi_d0 > e0[2]);
    if (s5) { hi_a = al[2]; hi_d0 = e0[2]; hi_d1 = e1[2]; }
    const bool s6 = (e0[0] > 0) && (hi_d0 > e0[0]);
    if (s6) { hi_a = al[0]; hi_d0 = e0[0]; hi_d1 = e1[0]; }
    swap = s1 || s2 || s3 || s4 || s5 || s6;
  }
  al[0] = lo_a; al[1] = hi_a; al[2] = 0.0f;
  ls_eval(Cost(), al, cs, e0, e1);
  const float lo_c = cs[0], hi_c = cs[1], p0_cost = cs[2];
  const bool improved = (lo_c < p0_cost) || (hi_c < p0_cost);
  const float alpha = improved ? (lo_c < hi_c ? lo_a : hi_a) : 0.0f;
  if (st.d_on) {
    const float xa = x + alpha * search;
    X[lane] = xa;
    WARM[lane] = xa;
  }
  if (lane == 0) { SCR[S::S_MISC + 13] = qg1; SCR[S::S_MISC + 14] = qg2; SCR[S::S_MISC + 15] = gauss; }
  if (lane == 0) { SCR[S::S_MISC + 4] = p0_d0; SCR[S::S_MISC + 5] = p0_d1; SCR[S::S_MISC + 6] = p0_cost; SCR[S::S_MISC + 7] = gtol; }
  if (lane == 0) { SCR[S::S_MISC + 1] = alpha; SCR[S::S_MISC + 2] = use_warm ? 1.0f : 0.0f; SCR[S::S_MISC + 3] = use_warm ? cost_w : cost_s; }
  ODK_SYNC();
  ODK_PROF(16);

  // ---------------- P10: sensors (lane = sensor), only when requested
  if (flags & 1) {
    if (lane < m->nsensor) {
      const int s = lane, site = m->sensor_site[s], b = m->site_body[site], type = m->sensor_type[s];
      float R[9], Rs[9], sp[3], dif[3], qb[4];
      for (int k = 0; k < 4; k++) qb[k] = XQUAT[k * NB + b];
      q2mat(R, qb);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rs[3 * i + j] = R[3 * i] * m->site_mat[site][j] + R[3 * i + 1] * m->site_mat[site][3 + j] + R[3 * i + 2] * m->site_mat[site][6 + j];
      for (int k = 0; k < 3; k++) {
        sp[k] = XPOS[k * NB + b] + R[3 * k] * m->site_pos[site][0] + R[3 * k + 1] * m->site_pos[site][1] + R[3 * k + 2] * m->site_pos[site][2];
        dif[k] = sp[k] - ref[k];
      }
      float cv[6];
      for (int k = 0; k < 6; k++) cv[k] = CVEL[k * NB + b];
      float vang[3] = {cv[0], cv[1], cv[2]}, t[3], vlin[3];
      cross3(t, dif, vang);
      vlin[0] = cv[3] - t[0]; vlin[1] = cv[4] - t[1]; vlin[2] = cv[5] - t[2];
      float* out = SENS + m->sensor_adr[s];
      auto tmul = [&](float* o, const float* v) {  // Rs^T v
        o[0] = Rs[0] * v[0] + Rs[3] * v[1] + Rs[6] * v[2];
        o[1] = Rs[1] * v[0] + Rs[4] * v[1] + Rs[7] * v[2];
        o[2] = Rs[2] * v[0] + Rs[5] * v[1] + Rs[8] * v[2];
      };
      if (type == 0) tmul(out, vang);
      else if (type == 1) tmul(out, vlin);
      else if (type == 2) {  // accelerometer: site is on the floating base (asserted at load)
        float ca[6];
        for (int k = 0; k < 6; k++) ca[k] = CACC[k * NB + b];
        for (int d = 0; d < 6; d++) {
          const float qa = X[d];
          for (int k = 0; k < 6; k++) ca[k] += CDOF[k * NR + d] * qa;   // base dofs: reduced column = dof
        }
        float al3[3], wl[3], vl[3], corr[3], acc[3];
        cross3(t, dif, ca);
        al3[0] = ca[3] - t[0]; al3[1] = ca[4] - t[1]; al3[2] = ca[5] - t[2];
        tmul(acc, al3); tmul(wl, vang); tmul(vl, vlin);
        cross3(corr, wl, vl);
        out[0] = acc[0] + corr[0]; out[1] = acc[1] + corr[1]; out[2] = acc[2] + corr[2];
      } else if (type == 3) { out[0] = Rs[2]; out[1] = Rs[5]; out[2] = Rs[8]; }
      else if (type == 4) { out[0] = Rs[0]; out[1] = Rs[3]; out[2] = Rs[6]; }
      else if (type == 5) { out[0] = vlin[0]; out[1] = vlin[1]; out[2] = vlin[2]; }
      else if (type == 6) { out[0] = vang[0]; out[1] = vang[1]; out[2] = vang[2]; }
      else if (type == 7) { out[0] = sp[0]; out[1] = sp[1]; out[2] = sp[2]; }
      else if (type == 8) {  // framequat: body quaternion times site quaternion
        float qs[4];
        qmul(qs, qb, m->site_quat[site]);
        qnormalize(qs);
        out[0] = qs[0]; out[1] = qs[1]; out[2] = qs[2]; out[3] = qs[3];
      }
    }
    // feet site heights + imu site rotation for the env logic
    if (lane < 2) {
      const int site = m->site_feet[lane], b = m->site_body[site];
      float qb[4], R[9];
      for (int k = 0; k < 4; k++) qb[k] = XQUAT[k * NB + b];
      q2mat(R, qb);
      SCR[S::S_MISC + 8 + lane] = XPOS[2 * NB + b] + R[6] * m->site_pos[site][0] + R[7] * m->site_pos[site][1] + R[8] * m->site_pos[site][2];
    }
    if (lane >= 2 && lane < 5) {  // gravity = site_xmat[imu]^T (0,0,-1) = -(third row of the site rotation)
      const int site = m->site_imu, b = m->site_body[site], c = lane - 2;
      float qb[4], R[9];
      for (int k = 0; k < 4; k++) qb[k] = XQUAT[k * NB + b];
      q2mat(R, qb);
      float v = 0;
      for (int k = 0; k < 3; k++) v += R[6 + k] * m->site_mat[site][3 * k + c];
      SCR[S::S_MISC + 10 + c] = -v;
    }
    ODK_SYNC();
  }
  ODK_PROF(17);
}

// mjx forward.euler (eulerdamp disabled): qvel += dt qacc; qpos integrated with the NEW qvel
template <class S, int G>
__device__ __forceinline__ void euler_env(float* L, const DevModel* __restrict__ m, const Statics<S, G>& st, int lane) {
  float* QPOS = L + S::O_QPOS; float* QVEL = L + S::O_QVEL; const float* X = L + S::O_X;
  const float dt = m->dt;
  if (lane < S::NV) QVEL[lane] += dt * X[lane];
  ODK_SYNC();
  if (lane == 0) {
    for (int k = 0; k < 3; k++) QPOS[k] += dt * QVEL[k];
    float w[3] = {QVEL[3], QVEL[4], QVEL[5]};
    float n = sqrtf(dot3(w, w));
    if (n < MINVAL_F) { w[0] = 1; w[1] = 0; w[2] = 0; n = 0; } else { const float inv = 1.0f / n; w[0] *= inv; w[1] *= inv; w[2] *= inv; }
    float s, co;
    sincosf(0.5f * dt * n, &s, &co);
    float qr[4] = {co, s * w[0], s * w[1], s * w[2]}, q0[4] = {QPOS[3], QPOS[4], QPOS[5], QPOS[6]}, res[4];
    qmul(res, q0, qr);
    qnormalize(res);
    QPOS[3] = res[0]; QPOS[4] = res[1]; QPOS[5] = res[2]; QPOS[6] = res[3];
  } else if (st.j_qadr >= 0) {
    QPOS[st.j_qadr] += dt * QVEL[st.j_dadr];
  }
  ODK_SYNC();
}

}  // namespace odk
