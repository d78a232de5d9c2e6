// odk_kernels.h -- device code of the fused env step (included by odk_engine.hip only).
//
// Geometry: one workgroup = one wavefront (64 lanes) = 64/G environments, G lanes per env
// (G = 32 or 64).  Every per-env array lives in LDS for the whole env step; HBM is touched once at
// the start (state record, action) and once at the end (state record, obs, reward...).  Lanes map
// to bodies / dofs / sparse-matrix entries / constraint rows by phase; cross-lane reductions use
// sub-wave shuffles.  Physics follows mjx.step as restated in oracle/odk_oracle.c:
// the reference reaches it through mjx_env.step at playground/open_duck_mini_v2/joystick.py:420.
//
// Restructurings relative to the textbook pipeline (all exact in real arithmetic):
//  * spatial quantities are expressed about the floating-base origin instead of the subtree COM;
//  * contact Jacobian rows are never formed: J_r = w_r . cdof[d] for dofs d above the foot, with the
//    6-vector w_r = [r x dir; dir], so J x, J^T f and J^T D J collapse to 6-vector / 6x6 algebra;
//  * inertia and Newton Hessian share MuJoCo's tree-sparse qM layout and a fill-free L^T D L.
#pragma once
#include <hip/hip_runtime.h>

#include "odk_model.h"

namespace odk {

// One workgroup == one wavefront: DS (LDS) instructions of a wave are issued and serviced in order, so
// cross-lane hand-offs through LDS need no s_barrier and no s_waitcnt -- only a compiler barrier that keeps
// the LDS accesses in program order (and stops values being cached in registers across the hand-off).
#define ODK_SYNC() asm volatile("" ::: "memory")
// Phase timing (build with -DODK_PROFILE): lane 0 accumulates shader-clock deltas per phase into the
// scratch area, which the debug LDS image carries out.  Zero cost when the macro is off.
#ifdef ODK_PROFILE
#define ODK_PROF(i) do { if (lane == 0) { long long _t = clock64(); SCR[S::S_PROF + (i)] += (float)(_t - _tprev); _tprev = _t; } } while (0)
#define ODK_PROF_BEGIN() long long _tprev = clock64()
#else
#define ODK_PROF(i) do { } while (0)
#define ODK_PROF_BEGIN() do { } while (0)
#endif

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
};

// ------------------------------------------------------------------------------------------------
// LDS layout (floats), per environment.  Component-major (SoA) arrays: X[k * N + item].
template <int NQ_, int NV_, int NB_, int NU_, int NM_, int NH_, int NROW_>
struct Shape {
  static constexpr int NQ = NQ_, NV = NV_, NB = NB_, NU = NU_, NM = NM_, NH = NH_, NROW = NROW_;
  static constexpr int NCROW = 48;  // contact rows
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
  // position stage
  static constexpr int O_XPOS = O_IPOS1 + 3;         // [3][NB]
  static constexpr int O_XMAT = O_XPOS + 3 * NB;     // [9][NB]
  static constexpr int O_CINERT = O_XMAT + 9 * NB;   // [10][NB]
  static constexpr int O_CDOF = O_CINERT + 10 * NB;  // [6][NV]
  static constexpr int O_BUF6 = O_CDOF + 6 * NV;     // [6][NV]  crb*cdof -> cdof_dot*qvel -> K_L*cdof
  static constexpr int O_BODY = O_BUF6 + 6 * NV;     // [12][NB] cvel | cfrc_local -> K_R*cdof
  static constexpr int O_M = O_BODY + 12 * NB;       // [NM] sparse inertia
  static constexpr int O_HL = O_M + NM;              // [NH] L^T D L of M, then Hessian and its factor
  // dof vectors
  static constexpr int O_QFS = O_HL + NH;            // qfrc_smooth
  static constexpr int O_QAS = O_QFS + NV;           // qacc_smooth
  static constexpr int O_X = O_QAS + NV;             // current qacc iterate
  static constexpr int O_MA = O_X + NV;              // M * qacc
  static constexpr int O_GRAD = O_MA + NV;           // gradient, then search direction
  static constexpr int O_MV = O_GRAD + NV;           // M * search (scratch: M * warmstart)
  // constraint rows
  static constexpr int O_D = O_MV + NV;              // efc_D (0 = structurally inactive row)
  static constexpr int O_AREF = O_D + NROW;
  static constexpr int O_JAR = O_AREF + NROW;        // J qacc - aref
  static constexpr int O_JV = O_JAR + NROW;          // J search (scratch: candidate Jaref, forces)
  static constexpr int O_W = O_JV + NROW;            // [NCROW][6] contact row wrenches [r x dir; dir]
  static constexpr int O_CDIST = O_W + 6 * NCROW;    // [12]
  static constexpr int O_CR = O_CDIST + NCON;        // [12][3] contact position relative to the base origin
  static constexpr int O_SCR = O_CR + 3 * NCON;      // scratch: foot twists, wrenches, 6x6 blocks, sensor inputs
  static constexpr int N_SCR = 192;
  static constexpr int O_SENS = O_SCR + N_SCR;       // sensordata[46]
  static constexpr int O_ACTF = O_SENS + NSENSD;     // actuator_force
  static constexpr int O_QACC = O_X;                 // qacc of the last forward == final iterate
  static constexpr int TOTAL = ((O_ACTF + NU + 3) / 4) * 4;
  // scratch sub-offsets
  static constexpr int S_VF = 0;      // [2][6] foot twist of the current vector
  static constexpr int S_FF = 12;     // [2][6] foot wrench sums
  static constexpr int S_K = 24;      // [3][36] K_L, K_R, K_X
  static constexpr int S_SV = 132;    // [3][6] cvel of base, left foot, right foot (sensors)
  static constexpr int S_CA = 150;    // [6] velocity part of cacc[base]
  static constexpr int S_MISC = 156;  // misc scalars (16)
  static constexpr int S_PROF = 172;  // [20] per-phase cycle counters (ODK_PROFILE builds)
};

// ------------------------------------------------------------------------------------------------
template <int G> __device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}
template <int G> __device__ __forceinline__ float gmax(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
  return v;
}
// argmax, lowest index wins ties (jnp.argmax semantics); result broadcast from lane 0 of the group
template <int G> __device__ __forceinline__ int gargmax(float v, int i) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, G);
    int oi = __shfl_xor(i, o, G);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  return __shfl(i, 0, G);
}

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
__device__ __forceinline__ void qnormalize(float* q) {
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL_F) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  float inv = 1.0f / n;
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
// Sparse L^T D L (mj_factorM) on MuJoCo's qM layout.  Pairs (m, q), 1<=m<=q<=depth(k), update entry
// (anc_m(k), anc_q(k)) from row k; all pairs of one k are independent.
template <int G>
__device__ inline void factor_ld(float* A, int nv, const int* depth, const int* Madr, const int (*anc_adr)[MAXV], const int* tri_m,
                                 const int* tri_q, int lane) {
  for (int k = nv - 1; k > 0; k--) {
    int D = depth[k];
    if (D == 0) continue;
    int ak = Madr[k];
    float inv = 1.0f / A[ak];
    int np = D * (D + 1) / 2;
    for (int t = lane; t < np; t += G) {
      int mm = tri_m[t], q = tri_q[t];
      A[anc_adr[k][mm] + (q - mm)] -= A[ak + mm] * inv * A[ak + q];
    }
    ODK_SYNC();
    for (int mm = 1 + lane; mm <= D; mm += G) A[ak + mm] *= inv;
    ODK_SYNC();
  }
}
// x <- (L^T D L)^-1 x   (mj_solveLD)
template <int G>
__device__ inline void solve_ld(const float* A, float* x, int nv, const int* depth, const int* Madr, const int (*anc)[MAXV],
                                const int* ndesc, const int (*desc)[MAXV], const int (*desc_adr)[MAXV], int lane) {
  for (int k = nv - 1; k > 0; k--) {
    int D = depth[k];
    if (D == 0) continue;
    float xk = x[k];
    for (int mm = 1 + lane; mm <= D; mm += G) x[anc[k][mm]] -= A[Madr[k] + mm] * xk;
    ODK_SYNC();
  }
  for (int i = lane; i < nv; i += G) x[i] /= A[Madr[i]];
  ODK_SYNC();
  for (int j = 0; j < nv - 1; j++) {
    int nd = ndesc[j];
    if (nd == 0) continue;
    float xj = x[j];
    for (int t = lane; t < nd; t += G) x[desc[j][t]] -= A[desc_adr[j][t]] * xj;
    ODK_SYNC();
  }
}

// impedance / stiffness of one constraint row (mjx constraint._row); returns D = 1/R and aref
__device__ inline void row_params(const float* solref, const float* solimp, float dt, float pos, float invweight, float vel, float& D,
                                  float& aref) {
  float timeconst = fmaxf(solref[0], 2.0f * dt), dampratio = solref[1];
  float dmin = fminf(fmaxf(solimp[0], 0.0001f), 0.9999f), dmax = fminf(fmaxf(solimp[1], 0.0001f), 0.9999f);
  float width = fmaxf(solimp[2], MINVAL_F), mid = fminf(fmaxf(solimp[3], 0.0001f), 0.9999f), power = fmaxf(solimp[4], 1.0f);
  float k = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  float b = 2.0f / (dmax * timeconst);
  if (solref[0] <= 0) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  float x = fabsf(pos) / width, y;
  if (power == 2.0f) {
    y = x < mid ? x * x / mid : 1.0f - (1.0f - x) * (1.0f - x) / (1.0f - mid);
  } else {
    y = x < mid ? powf(x, power) / powf(mid, power - 1.0f) : 1.0f - powf(1.0f - x, power) / powf(1.0f - mid, power - 1.0f);
  }
  float imp = fminf(fmaxf(dmin + y * (dmax - dmin), dmin), dmax);
  if (x > 1.0f) imp = dmax;
  float R = fmaxf(invweight * (1.0f - imp) / imp, MINVAL_F);
  D = 1.0f / R;
  aref = -b * vel - k * imp * pos;
}

// ------------------------------------------------------------------------------------------------
// One mjx.forward for one env (all G lanes of the group call this together).
//   flags bit0: compute sensordata / debug outputs (last substep only)
template <class S, int G>
__device__ void forward_env(float* __restrict__ L, const DevModel* __restrict__ m, int lane, int flags) {
  constexpr int NV = S::NV, NB = S::NB, NU = S::NU, NM = S::NM, NH = S::NH, NROW = S::NROW;
  float* QPOS = L + S::O_QPOS; float* QVEL = L + S::O_QVEL; float* WARM = L + S::O_WARM; float* CTRL = L + S::O_CTRL;
  float* Q0 = L + S::O_Q0; float* MASS = L + S::O_MASS; float* ARM = L + S::O_ARM; float* FRL = L + S::O_FRL; float* KP = L + S::O_KP;
  float* XPOS = L + S::O_XPOS; float* XMAT = L + S::O_XMAT; float* CIN = L + S::O_CINERT; float* CDOF = L + S::O_CDOF;
  float* BUF6 = L + S::O_BUF6; float* BODY = L + S::O_BODY; float* M = L + S::O_M; float* HL = L + S::O_HL;
  float* QFS = L + S::O_QFS; float* QAS = L + S::O_QAS; float* X = L + S::O_X; float* MA = L + S::O_MA; float* GRAD = L + S::O_GRAD;
  float* MV = L + S::O_MV; float* ED = L + S::O_D; float* AREF = L + S::O_AREF; float* JAR = L + S::O_JAR; float* JV = L + S::O_JV;
  float* W = L + S::O_W; float* CDIST = L + S::O_CDIST; float* CR = L + S::O_CR; float* SCR = L + S::O_SCR;
  float* SENS = L + S::O_SENS; float* ACTF = L + S::O_ACTF;
  const int nfl = m->nfl, nlim = m->nlim, r0c = nfl + nlim;
  const float dt = m->dt;

  ODK_PROF_BEGIN();
  // ---------------- P1: kinematics + cinert + cdof (lane = body); spatial reference = base origin
  if (lane < NB) {
    const int b = lane;
    float p[3], q[4];
    const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
    if (!m->body_in_tree[b]) {
      for (int k = 0; k < 3; k++) p[k] = m->body_pos[b][k];
      for (int k = 0; k < 4; k++) q[k] = m->body_quat[b][k];
    } else {
      for (int k = 0; k < 3; k++) p[k] = ref[k];
      for (int k = 0; k < 4; k++) q[k] = QPOS[3 + k];
      qnormalize(q);
      if (b == m->base_body) {
        float R[9];
        q2mat(R, q);
        for (int k = 0; k < 3; k++) {
          for (int c = 0; c < 6; c++) { CDOF[c * NV + k] = (c == 3 + k) ? 1.0f : 0.0f; }
          CDOF[0 * NV + 3 + k] = R[k]; CDOF[1 * NV + 3 + k] = R[3 + k]; CDOF[2 * NV + 3 + k] = R[6 + k];
          CDOF[3 * NV + 3 + k] = 0; CDOF[4 * NV + 3 + k] = 0; CDOF[5 * NV + 3 + k] = 0;
        }
      }
      const int len = m->body_chain_len[b];
      for (int ci = 0; ci < len; ci++) {
        const int c = m->body_chain[b][ci];
        float t[3], bq[4] = {m->body_quat[c][0], m->body_quat[c][1], m->body_quat[c][2], m->body_quat[c][3]};
        float bp[3] = {m->body_pos[c][0], m->body_pos[c][1], m->body_pos[c][2]};
        qrot(t, q, bp);
        p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
        qmul(q, q, bq);
        const int ja = m->body_jntadr[c], jn = m->body_jntnum[c];
        for (int j = ja; j < ja + jn; j++) {
          float ax[3] = {m->jnt_axis[j][0], m->jnt_axis[j][1], m->jnt_axis[j][2]};
          float jp[3] = {m->jnt_pos[j][0], m->jnt_pos[j][1], m->jnt_pos[j][2]};
          float anchor[3], axw[3];
          qrot(t, q, jp);
          anchor[0] = p[0] + t[0]; anchor[1] = p[1] + t[1]; anchor[2] = p[2] + t[2];
          qrot(axw, q, ax);
          if (c == b) {
            const int d = m->jnt_dofadr[j];
            float off[3] = {ref[0] - anchor[0], ref[1] - anchor[1], ref[2] - anchor[2]}, lin[3];
            cross3(lin, axw, off);
            CDOF[0 * NV + d] = axw[0]; CDOF[1 * NV + d] = axw[1]; CDOF[2 * NV + d] = axw[2];
            CDOF[3 * NV + d] = lin[0]; CDOF[4 * NV + d] = lin[1]; CDOF[5 * NV + d] = lin[2];
          }
          const int qa = m->jnt_qposadr[j];
          float s, co;
          sincosf(0.5f * (QPOS[qa] - Q0[qa]), &s, &co);
          float qj[4] = {co, s * ax[0], s * ax[1], s * ax[2]};
          qmul(q, q, qj);
          qrot(t, q, jp);
          p[0] = anchor[0] - t[0]; p[1] = anchor[1] - t[1]; p[2] = anchor[2] - t[2];
        }
      }
      qnormalize(q);
    }
    float R[9];
    q2mat(R, q);
    for (int k = 0; k < 3; k++) XPOS[k * NB + b] = p[k];
    for (int k = 0; k < 9; k++) XMAT[k * NB + b] = R[k];
    // cinert about the base origin
    float ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
    if (b == 1) { ip[0] = L[S::O_IPOS1]; ip[1] = L[S::O_IPOS1 + 1]; ip[2] = L[S::O_IPOS1 + 2]; }
    float off[3];
    for (int k = 0; k < 3; k++) off[k] = p[k] + R[3 * k] * ip[0] + R[3 * k + 1] * ip[1] + R[3 * k + 2] * ip[2] - ref[k];
    const float* f = m->body_inertia[b];
    const float Ib[9] = {f[0], f[3], f[4], f[3], f[1], f[5], f[4], f[5], f[2]};
    float T[9], Iw[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i] * Ib[j] + R[3 * i + 1] * Ib[3 + j] + R[3 * i + 2] * Ib[6 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Iw[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
    const float mb = MASS[b], o2 = dot3(off, off);
    CIN[0 * NB + b] = Iw[0] + mb * (o2 - off[0] * off[0]);
    CIN[1 * NB + b] = Iw[4] + mb * (o2 - off[1] * off[1]);
    CIN[2 * NB + b] = Iw[8] + mb * (o2 - off[2] * off[2]);
    CIN[3 * NB + b] = Iw[1] - mb * off[0] * off[1];
    CIN[4 * NB + b] = Iw[2] - mb * off[0] * off[2];
    CIN[5 * NB + b] = Iw[5] - mb * off[1] * off[2];
    CIN[6 * NB + b] = mb * off[0]; CIN[7 * NB + b] = mb * off[1]; CIN[8 * NB + b] = mb * off[2];
    CIN[9 * NB + b] = mb;
  }
  ODK_SYNC();

  ODK_PROF(0);
  // ---------------- P2: composite inertia times cdof (lane = dof), velocity prefix / cdof_dot
  float dotv[6] = {0, 0, 0, 0, 0, 0};
  if (lane < NV) {
    const int i = lane, b = m->dof_body[i];
    float crb[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ns = m->body_nsub[b];
    for (int s = 0; s < ns; s++) {
      const int c = m->body_sub[b][s];
#pragma unroll
      for (int k = 0; k < 10; k++) crb[k] += CIN[k * NB + c];
    }
    float cd[6], buf[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cd[k] = CDOF[k * NV + i];
    inert_mul(buf, crb, cd);
#pragma unroll
    for (int k = 0; k < 6; k++) BUF6[k * NV + i] = buf[k];
    // mj_comVel prefix: velocity of the parent chain at the moment dof i is applied
    float pre[6] = {0, 0, 0, 0, 0, 0};
    const int np = m->dof_nprefix[i];
    for (int s = 0; s < np; s++) {
      const int e = m->dof_prefix[i][s];
      const float qv = QVEL[e];
#pragma unroll
      for (int k = 0; k < 6; k++) pre[k] += CDOF[k * NV + e] * qv;
    }
    if (i >= 3) {  // cdof_dot = cross_motion(prefix, cdof); stored pre-multiplied by qvel[i]
      float a[3], b2[3], c2[3];
      cross3(a, pre, cd);
      cross3(b2, pre, cd + 3);
      cross3(c2, pre + 3, cd);
      const float qv = QVEL[i];
      dotv[0] = a[0] * qv; dotv[1] = a[1] * qv; dotv[2] = a[2] * qv;
      dotv[3] = (b2[0] + c2[0]) * qv; dotv[4] = (b2[1] + c2[1]) * qv; dotv[5] = (b2[2] + c2[2]) * qv;
    }
  }
  ODK_SYNC();
  ODK_PROF(1);
  // ---------------- P3: sparse inertia entries (lane = entry)
  for (int p = lane; p < NM; p += G) {
    const int i = m->M_i[p], j = m->M_j[p];
    float v = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) v += CDOF[k * NV + j] * BUF6[k * NV + i];
    if (i == j) v += ARM[i];
    M[p] = v;
  }
  ODK_SYNC();
  if (lane < NV) {
#pragma unroll
    for (int k = 0; k < 6; k++) BUF6[k * NV + lane] = dotv[k];
  }
  ODK_SYNC();
  ODK_PROF(2);
  // ---------------- P4: body velocity, bias acceleration, local force (lane = body)
  if (lane < NB) {
    const int b = lane;
    float cvel[6] = {0, 0, 0, 0, 0, 0}, cacc[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
    const int na = m->body_nancdof[b];
    for (int s = 0; s < na; s++) {
      const int d = m->body_ancdof[b][s];
      const float qv = QVEL[d];
#pragma unroll
      for (int k = 0; k < 6; k++) { cvel[k] += CDOF[k * NV + d] * qv; cacc[k] += BUF6[k * NV + d]; }
    }
    float ci[10], t1[6], t2[6], fr[6];
#pragma unroll
    for (int k = 0; k < 10; k++) ci[k] = CIN[k * NB + b];
    inert_mul(fr, ci, cacc);
    inert_mul(t1, ci, cvel);
    // cross_force(cvel, t1)
    float a[3], c2[3];
    cross3(a, cvel, t1);
    cross3(c2, cvel + 3, t1 + 3);
    t2[0] = a[0] + c2[0]; t2[1] = a[1] + c2[1]; t2[2] = a[2] + c2[2];
    cross3(t2 + 3, cvel, t1 + 3);
#pragma unroll
    for (int k = 0; k < 6; k++) { BODY[k * NB + b] = cvel[k]; BODY[(6 + k) * NB + b] = fr[k] + t2[k]; }
    if (b == m->base_body) {
#pragma unroll
      for (int k = 0; k < 6; k++) { SCR[S::S_SV + k] = cvel[k]; SCR[S::S_CA + k] = cacc[k]; }
    }
    if (b == m->foot_body[0]) {
#pragma unroll
      for (int k = 0; k < 6; k++) SCR[S::S_SV + 6 + k] = cvel[k];
    }
    if (b == m->foot_body[1]) {
#pragma unroll
      for (int k = 0; k < 6; k++) SCR[S::S_SV + 12 + k] = cvel[k];
    }
  }
  ODK_SYNC();
  ODK_PROF(3);
  // ---------------- P5: bias force, passive, actuation -> qfrc_smooth (lane = dof)
  if (lane < NV) {
    const int i = lane, b = m->dof_body[i];
    float cf[6] = {0, 0, 0, 0, 0, 0};
    const int ns = m->body_nsub[b];
    for (int s = 0; s < ns; s++) {
      const int c = m->body_sub[b][s];
#pragma unroll
      for (int k = 0; k < 6; k++) cf[k] += BODY[(6 + k) * NB + c];
    }
    float bias = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) bias += CDOF[k * NV + i] * cf[k];
    const float qv = QVEL[i];
    float frc = -m->dof_damping[i] * qv - bias;
    const int u = m->dof_act[i];
    if (u >= 0) {
      float ctrl = CTRL[u];
      if (m->act_ctrllimited[u]) ctrl = fminf(fmaxf(ctrl, m->act_ctrlrange[u][0]), m->act_ctrlrange[u][1]);
      const float kp = KP[u];
      float af = kp * ctrl - kp * QPOS[m->act_qposadr[u]] + m->act_bias2[u] * qv;
      if (m->act_forcelimited[u]) af = fminf(fmaxf(af, m->act_forcerange[u][0]), m->act_forcerange[u][1]);
      ACTF[u] = af;
      frc += af;
    }
    QFS[i] = frc;
    QAS[i] = frc;
  }
  for (int p = lane; p < NM; p += G) HL[p] = M[p];
  ODK_SYNC();
  ODK_PROF(4);
  // ---------------- P6: qacc_smooth = M^-1 qfrc_smooth
  factor_ld<G>(HL, NV, m->dof_depth, m->dof_Madr, m->dof_anc_adr, m->tri_m, m->tri_q, lane);
  ODK_PROF(15);
  solve_ld<G>(HL, QAS, NV, m->dof_depth, m->dof_Madr, m->dof_anc, m->dof_ndesc, m->dof_desc, m->dof_desc_adr, lane);

  ODK_PROF(5);
  // ---------------- P7: collision.  Foot (convex mesh) vs plane: mjx collision_convex.plane_convex
  const float ref[3] = {QPOS[0], QPOS[1], QPOS[2]};
  for (int f = 0; f < 2; f++) {
    const int fb = m->foot_body[f], nvt = m->foot_nvert[f];
    const bool has = lane < nvt;
    float w[3] = {0, 0, 0}, sup = -3.0e38f;
    const float pn[3] = {m->plane_n[0], m->plane_n[1], m->plane_n[2]};
    if (has) {
      const float* vb = m->foot_vert[f][lane];
      for (int k = 0; k < 3; k++) w[k] = XPOS[k * NB + fb] + XMAT[(3 * k) * NB + fb] * vb[0] + XMAT[(3 * k + 1) * NB + fb] * vb[1] + XMAT[(3 * k + 2) * NB + fb] * vb[2];
      sup = (m->plane_pos[0] - w[0]) * pn[0] + (m->plane_pos[1] - w[1]) * pn[1] + (m->plane_pos[2] - w[2]) * pn[2];
    }
    const float smax = gmax<G>(sup);
    const float thr = fmaxf(0.0f, smax - 1e-3f);
    const float dm = has ? ((sup > thr) ? 0.0f : -1e6f) : -3.0e38f;
    int idx[4];
    idx[0] = gargmax<G>(dm, lane);
    float a[3] = {__shfl(w[0], idx[0], G), __shfl(w[1], idx[0], G), __shfl(w[2], idx[0], G)};
    float ap[3] = {a[0] - w[0], a[1] - w[1], a[2] - w[2]};
    idx[1] = gargmax<G>(has ? dot3(ap, ap) + dm : -3.0e38f, lane);
    float bq[3] = {__shfl(w[0], idx[1], G), __shfl(w[1], idx[1], G), __shfl(w[2], idx[1], G)};
    float amb[3] = {a[0] - bq[0], a[1] - bq[1], a[2] - bq[2]}, ab[3];
    cross3(ab, pn, amb);
    idx[2] = gargmax<G>(has ? fabsf(dot3(ap, ab)) + dm : -3.0e38f, lane);
    float cq[3] = {__shfl(w[0], idx[2], G), __shfl(w[1], idx[2], G), __shfl(w[2], idx[2], G)};
    float amc[3] = {a[0] - cq[0], a[1] - cq[1], a[2] - cq[2]}, bmc[3] = {bq[0] - cq[0], bq[1] - cq[1], bq[2] - cq[2]}, ac[3], bc[3];
    cross3(ac, pn, amc);
    cross3(bc, pn, bmc);
    float bp[3] = {bq[0] - w[0], bq[1] - w[1], bq[2] - w[2]};
    float v1 = fabsf(dot3(bp, bc)) + dm, v2 = fabsf(dot3(ap, ac)) + dm;
    float vv = v1; int vi = lane;
    if (v2 > v1) { vv = v2; vi = nvt + lane; }
    if (!has) { vv = -3.0e38f; vi = 2 * nvt + lane; }
    idx[3] = gargmax<G>(vv, vi);
    idx[3] = idx[3] >= nvt ? idx[3] - nvt : idx[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      bool uniq = true;
      for (int q = 0; q < k; q++) uniq = uniq && (idx[q] != idx[k]);
      if (lane == idx[k]) {
        const float dist = uniq ? -sup : 1.0f;
        const int c = 4 * f + k;
        CDIST[c] = dist;
        for (int t = 0; t < 3; t++) CR[3 * c + t] = w[t] - 0.5f * dist * pn[t] - ref[t];
      }
    }
  }
  ODK_PROF(6);
  // foot-foot: oriented-box cull (a positive separation of the boxes bounds the hulls' separation from below)
  {
    float c1[3], c2[3], A1[9], A2[9], tt[3];
    for (int f = 0; f < 2; f++) {
      const int fb = m->foot_body[f];
      float* cc = f ? c2 : c1; float* AA = f ? A2 : A1;
      float R[9];
      for (int k = 0; k < 9; k++) R[k] = XMAT[k * NB + fb];
      for (int k = 0; k < 3; k++) cc[k] = XPOS[k * NB + fb] + R[3 * k] * m->foot_obb_center[f][0] + R[3 * k + 1] * m->foot_obb_center[f][1] + R[3 * k + 2] * m->foot_obb_center[f][2];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) AA[3 * i + j] = R[3 * i] * m->foot_obb_axes[f][j] + R[3 * i + 1] * m->foot_obb_axes[f][3 + j] + R[3 * i + 2] * m->foot_obb_axes[f][6 + j];
    }
    for (int k = 0; k < 3; k++) tt[k] = c2[k] - c1[k];
    float best = -3.0e38f;
    float ax[15][3];
    int na = 0;
    for (int k = 0; k < 3; k++) { ax[na][0] = A1[k]; ax[na][1] = A1[3 + k]; ax[na][2] = A1[6 + k]; na++; }
    for (int k = 0; k < 3; k++) { ax[na][0] = A2[k]; ax[na][1] = A2[3 + k]; ax[na][2] = A2[6 + k]; na++; }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        float cr[3];
        cross3(cr, ax[i], ax[3 + j]);
        float n = sqrtf(dot3(cr, cr));
        if (n < 1e-6f) continue;
        ax[na][0] = cr[0] / n; ax[na][1] = cr[1] / n; ax[na][2] = cr[2] / n;
        na++;
      }
    for (int a = 0; a < na; a++) {
      float r1 = 0, r2 = 0;
      for (int k = 0; k < 3; k++) {
        r1 += m->foot_obb_half[0][k] * fabsf(dot3(ax[a], ax[k]));
        r2 += m->foot_obb_half[1][k] * fabsf(dot3(ax[a], ax[3 + k]));
      }
      best = fmaxf(best, fabsf(dot3(tt, ax[a])) - r1 - r2);
    }
    if (lane < 4) {
      const int c = 8 + lane;
      // separated boxes -> inactive pair.  Overlapping boxes (feet about to touch) need the full convex-convex
      // routine, which round 1 does not carry on the GPU: the pair is reported inactive and flagged.
      CDIST[c] = (lane == 0 && best > 0) ? best : 1.0f;
      CR[3 * c] = 0; CR[3 * c + 1] = 0; CR[3 * c + 2] = 0;
      if (lane == 0) SCR[S::S_MISC] = best;
    }
  }
  ODK_SYNC();

  ODK_PROF(7);
  // ---------------- P8: constraint rows (lane = row): D, aref, contact wrenches
  for (int r = lane; r < NROW; r += G) {
    float D = 0, aref = 0;
    if (r < nfl) {
      const int d = m->fl_dof[r];
      D = m->fl_D[r];
      aref = -m->fl_b[r] * QVEL[d];
    } else if (r < r0c) {
      const int j = m->lim_jnt[r - nfl];
      const float qv = QPOS[m->jnt_qposadr[j]];
      const float dlo = qv - m->jnt_range[j][0], dhi = m->jnt_range[j][1] - qv;
      const float pos = fminf(dlo, dhi);
      if (pos < 0) {
        const float sgn = dlo < dhi ? 1.0f : -1.0f;
        row_params(m->lim_solref[r - nfl], m->lim_solimp[r - nfl], dt, pos, m->lim_invweight[r - nfl], sgn * QVEL[m->jnt_dofadr[j]], D, aref);
      }
    } else {
      const int rc = r - r0c, c = rc >> 2, s = rc & 3, pair = c >> 2;
      const float dist = CDIST[c];
      const float mu = m->pair_mu[pair];
      const float fs = (s & 1) ? -mu : mu;
      const float* fr = m->plane_frame;  // pair 2 (foot-foot) frames come with the convex-convex routine
      const int td = 3 * (1 + (s >> 1));
      const float dir[3] = {fr[0] + fs * fr[td], fr[1] + fs * fr[td + 1], fr[2] + fs * fr[td + 2]};
      const float rr[3] = {CR[3 * c], CR[3 * c + 1], CR[3 * c + 2]};
      float ang[3];
      cross3(ang, rr, dir);
      float* wr = W + 6 * rc;
      wr[0] = ang[0]; wr[1] = ang[1]; wr[2] = ang[2]; wr[3] = dir[0]; wr[4] = dir[1]; wr[5] = dir[2];
      if (dist < 0) {
        float vel = 0;
        const float* v2 = SCR + S::S_SV + (pair == 0 ? 6 : 12);  // geom2's body: left foot for pair 0, right foot otherwise
#pragma unroll
        for (int k = 0; k < 6; k++) vel += wr[k] * v2[k];
        if (pair == 2) {
          const float* v1 = SCR + S::S_SV + 6;
#pragma unroll
          for (int k = 0; k < 6; k++) vel -= wr[k] * v1[k];
        }
        row_params(m->pair_solref[pair], m->pair_solimp[pair], dt, dist, m->pair_invweight[pair], vel, D, aref);
      }
    }
    ED[r] = D;
    AREF[r] = aref;
  }
  ODK_SYNC();

  ODK_PROF(8);
  // ---------------- P9: Newton solver, one iteration (mjx solver.solve)
  // helper lambdas -------------------------------------------------------------
  auto foot_twist = [&](const float* vec) {  // SCR[S_VF + 6 f + k] = sum_{d above foot f} cdof[k][d] vec[d]
    if (lane < 12) {
      const int f = lane / 6, k = lane % 6;
      float s = 0;
      for (int d = 0; d < NV; d++) s += m->foot_dofmask[f][d] ? CDOF[k * NV + d] * vec[d] : 0.0f;
      SCR[S::S_VF + lane] = s;
    }
  };
  auto row_jx = [&](int r, const float* vec) -> float {  // (J vec)[r]; needs foot_twist(vec) + sync
    if (r < nfl) return vec[m->fl_dof[r]];
    if (r < r0c) {
      const int j = m->lim_jnt[r - nfl];
      const float qv = QPOS[m->jnt_qposadr[j]];
      const float sgn = (qv - m->jnt_range[j][0]) < (m->jnt_range[j][1] - qv) ? 1.0f : -1.0f;
      return sgn * vec[m->jnt_dofadr[j]];
    }
    const int rc = r - r0c, pair = rc >> 4;
    const float* wr = W + 6 * rc;
    const float* v2 = SCR + S::S_VF + (pair == 0 ? 0 : 6);
    float s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += wr[k] * v2[k];
    if (pair == 2) {
      const float* v1 = SCR + S::S_VF;
#pragma unroll
      for (int k = 0; k < 6; k++) s -= wr[k] * v1[k];
    }
    return s;
  };
  auto row_cost = [&](int r, float jar, float& force, bool& quad) -> float {
    const float D = ED[r];
    if (r < nfl) {
      const float fl = FRL[m->fl_dof[r]], rf = m->fl_R[r] * fl;
      if (jar <= -rf) { force = fl; quad = false; return -0.5f * rf * fl - fl * jar; }
      if (jar >= rf) { force = -fl; quad = false; return -0.5f * rf * fl + fl * jar; }
      force = -D * jar; quad = true;
      return 0.5f * D * jar * jar;
    }
    if (D > 0 && jar < 0) { force = -D * jar; quad = true; return 0.5f * D * jar * jar; }
    force = 0; quad = false;
    return 0.0f;
  };
  auto mul_M = [&](int i, const float* vec) -> float {
    float s = 0;
    const int n = m->dof_nsym[i];
    for (int t = 0; t < n; t++) s += M[m->dof_sym_adr[i][t]] * vec[m->dof_sym_dof[i][t]];
    return s;
  };

  // candidate 1: qacc_smooth (Ma = qfrc_smooth, gauss = 0)
  foot_twist(QAS);
  ODK_SYNC();
  float cost_s = 0;
  for (int r = lane; r < NROW; r += G) {
    if (ED[r] == 0.0f && r >= nfl) { JAR[r] = 0; continue; }
    const float jar = row_jx(r, QAS) - AREF[r];
    JAR[r] = jar;
    float fo; bool qd;
    cost_s += row_cost(r, jar, fo, qd);
  }
  cost_s = gsum<G>(cost_s);
  ODK_SYNC();
  // candidate 2: warmstart
  foot_twist(WARM);
  float gw = 0;
  if (lane < NV) {
    const float ma = mul_M(lane, WARM);
    MV[lane] = ma;
    gw = (ma - QFS[lane]) * (WARM[lane] - QAS[lane]);
  }
  ODK_SYNC();
  float cost_w = 0;
  for (int r = lane; r < NROW; r += G) {
    if (ED[r] == 0.0f && r >= nfl) { JV[r] = 0; continue; }
    const float jar = row_jx(r, WARM) - AREF[r];
    JV[r] = jar;
    float fo; bool qd;
    cost_w += row_cost(r, jar, fo, qd);
  }
  cost_w = gsum<G>(cost_w) + 0.5f * gsum<G>(gw);
  const bool use_warm = cost_w < cost_s;
  float gauss = use_warm ? 0.5f * gsum<G>(gw) : 0.0f;
  ODK_SYNC();
  if (lane < NV) {
    X[lane] = use_warm ? WARM[lane] : QAS[lane];
    MA[lane] = use_warm ? MV[lane] : QFS[lane];
  }
  if (use_warm) for (int r = lane; r < NROW; r += G) JAR[r] = JV[r];
  ODK_SYNC();
  ODK_PROF(9);
  // forces of the chosen point -> JV (scratch), foot wrench sums, gradient
  for (int r = lane; r < NROW; r += G) {
    float fo = 0; bool qd = false;
    if (!(ED[r] == 0.0f && r >= nfl)) row_cost(r, JAR[r], fo, qd);
    JV[r] = fo;
    // mark rows that enter the Hessian by the sign bit trick: keep a separate flag in W? use AREF sign? -> store in MV-free slot below
  }
  ODK_SYNC();
  if (lane < 12) {  // FF[f][k] = sum over the foot's contact rows of w_r[k] f_r   (pair 2: +right, -left)
    const int f = lane / 6, k = lane % 6;
    float s = 0;
    const int rb = 16 * f;
    for (int rc = rb; rc < rb + 16; rc++) s += W[6 * rc + k] * JV[r0c + rc];
    for (int rc = 32; rc < 48; rc++) s += (f ? 1.0f : -1.0f) * W[6 * rc + k] * JV[r0c + rc];
    SCR[S::S_FF + lane] = s;
  }
  // K blocks: K_f[a][b] = sum_r D_r [quad] w_r[a] w_r[b]; rows of pair 2 add to both feet and form K_X
  for (int t = lane; t < 108; t += G) {
    const int blk = t / 36, a = (t % 36) / 6, b2 = t % 6;
    float s = 0;
    if (blk < 2) {
      for (int rc = 16 * blk; rc < 16 * blk + 16; rc++) {
        const int r = r0c + rc;
        const float act = (ED[r] > 0 && JAR[r] < 0) ? ED[r] : 0.0f;
        s += act * W[6 * rc + a] * W[6 * rc + b2];
      }
    }
    float sx = 0;
    for (int rc = 32; rc < 48; rc++) {
      const int r = r0c + rc;
      const float act = (ED[r] > 0 && JAR[r] < 0) ? ED[r] : 0.0f;
      sx += act * W[6 * rc + a] * W[6 * rc + b2];
    }
    SCR[S::S_K + t] = (blk < 2) ? s + sx : sx;
  }
  ODK_SYNC();
  const bool ff_active = SCR[S::S_K + 72 + 21] != 0.0f || SCR[S::S_K + 72 + 28] != 0.0f || SCR[S::S_K + 72 + 35] != 0.0f;  // diag(lin) of K_X
  if (lane < NV) {
    const int i = lane;
    float qc = 0;
    const int rf = m->dof_flrow[i], rl = m->dof_limrow[i];
    if (rf >= 0) qc += JV[rf];
    if (rl >= 0 && ED[nfl + rl] > 0) {
      const int j = m->lim_jnt[rl];
      const float qv = QPOS[m->jnt_qposadr[j]];
      const float sgn = (qv - m->jnt_range[j][0]) < (m->jnt_range[j][1] - qv) ? 1.0f : -1.0f;
      qc += sgn * JV[nfl + rl];
    }
    float cd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cd[k] = CDOF[k * NV + i];
    for (int f = 0; f < 2; f++)
      if (m->foot_dofmask[f][i]) {
#pragma unroll
        for (int k = 0; k < 6; k++) qc += cd[k] * SCR[S::S_FF + 6 * f + k];
      }
    GRAD[i] = MA[i] - QFS[i] - qc;
    // T_f[i] = K_f cdof[i]
    for (int f = 0; f < 2; f++) {
      float* T = f ? (BODY) : (BUF6);
      const int stride = NV;
#pragma unroll
      for (int a = 0; a < 6; a++) {
        float s = 0;
        if (m->foot_dofmask[f][i]) {
#pragma unroll
          for (int b2 = 0; b2 < 6; b2++) s += SCR[S::S_K + 36 * f + 6 * a + b2] * cd[b2];
        }
        T[a * stride + i] = s;
      }
    }
  }
  ODK_SYNC();
  ODK_PROF(10);
  // Hessian entries on the virtual-tree layout
  for (int p = lane; p < NH; p += G) {
    const int i = m->H_i[p], j = m->H_j[p], src = m->H_src[p];
    float v = src >= 0 ? M[src] : 0.0f;
    if (i == j) {
      const int rf = m->dof_flrow[i], rl = m->dof_limrow[i];
      if (rf >= 0) { const float fl = FRL[i], rfv = m->fl_R[rf] * fl, jar = JAR[rf]; if (jar > -rfv && jar < rfv) v += ED[rf]; }
      if (rl >= 0 && ED[nfl + rl] > 0 && JAR[nfl + rl] < 0) v += ED[nfl + rl];
    }
    float cj[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cj[k] = CDOF[k * NV + j];
    const int mLi = m->foot_dofmask[0][i], mLj = m->foot_dofmask[0][j], mRi = m->foot_dofmask[1][i], mRj = m->foot_dofmask[1][j];
    if (mLi && mLj) {
#pragma unroll
      for (int k = 0; k < 6; k++) v += cj[k] * BUF6[k * NV + i];
    }
    if (mRi && mRj) {
#pragma unroll
      for (int k = 0; k < 6; k++) v += cj[k] * BODY[k * NV + i];
    }
    if (ff_active) {  // cross terms -(A_R^T K_X A_L + A_L^T K_X A_R); rare
      const float wgt = (float)(mRi && mLj) + (float)(mLi && mRj);
      if (wgt != 0.0f) {
        float ci[6], s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) ci[k] = CDOF[k * NV + i];
        for (int a = 0; a < 6; a++)
          for (int b2 = 0; b2 < 6; b2++) s += ci[a] * SCR[S::S_K + 72 + 6 * a + b2] * cj[b2];
        v -= wgt * s;
      }
    }
    HL[p] = v;
  }
  ODK_SYNC();
  ODK_PROF(11);
  factor_ld<G>(HL, NV, m->vdof_depth, m->vdof_Madr, m->vdof_anc_adr, m->tri_m, m->tri_q, lane);
  ODK_PROF(16);
  solve_ld<G>(HL, GRAD, NV, m->vdof_depth, m->vdof_Madr, m->vdof_anc, m->vdof_ndesc, m->vdof_desc, m->vdof_desc_adr, lane);
  if (lane < NV) GRAD[lane] = -GRAD[lane];  // search = -H^-1 grad
  ODK_SYNC();

  ODK_PROF(12);
  // ---- line search (mjx solver._linesearch)
  foot_twist(GRAD);
  float sn = 0, qg1 = 0, qg2 = 0;
  if (lane < NV) {
    const float s = GRAD[lane], mv = mul_M(lane, GRAD);
    MV[lane] = mv;
    sn = s * s;
    qg1 = s * MA[lane] - s * QFS[lane];
    qg2 = 0.5f * s * mv;
  }
  sn = gsum<G>(sn); qg1 = gsum<G>(qg1); qg2 = gsum<G>(qg2);
  ODK_SYNC();
  for (int r = lane; r < NROW; r += G) JV[r] = (ED[r] == 0.0f && r >= nfl) ? 0.0f : row_jx(r, GRAD);
  ODK_SYNC();
  ODK_PROF(13);
  const float gtol = m->tolerance * m->ls_tolerance * sqrtf(sn) * m->meaninertia * (float)(NV > 1 ? NV : 1);
  // evaluate up to three step sizes at once: cost, first and second derivative along the search
  auto ls_eval3 = [&](const float* al, float* cost, float* d0, float* d1) {
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = lane; r < NROW; r += G) {
      const float D = ED[r];
      if (D == 0.0f && r >= nfl) continue;
      const float jar = JAR[r], jv = JV[r];
      const float q0 = 0.5f * D * jar * jar, q1 = D * jv * jar, q2 = 0.5f * D * jv * jv;
      if (r < nfl) {
        const float fl = FRL[m->fl_dof[r]], rf = m->fl_R[r] * fl;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float x = jar + al[a] * jv;
          if (x <= -rf) { acc[3 * a] += fl * (-0.5f * rf - jar); acc[3 * a + 1] += -fl * jv; }
          else if (x >= rf) { acc[3 * a] += fl * (-0.5f * rf + jar); acc[3 * a + 1] += fl * jv; }
          else { acc[3 * a] += q0; acc[3 * a + 1] += q1; acc[3 * a + 2] += q2; }
        }
      } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          if (jar + al[a] * jv < 0) { acc[3 * a] += q0; acc[3 * a + 1] += q1; acc[3 * a + 2] += q2; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = gsum<G>(acc[k]);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float t0 = acc[3 * a] + gauss, t1 = acc[3 * a + 1] + qg1, t2 = acc[3 * a + 2] + qg2;
      cost[a] = al[a] * al[a] * t2 + al[a] * t1 + t0;
      d0[a] = 2.0f * al[a] * t2 + t1;
      d1[a] = 2.0f * t2 + (t2 == 0.0f ? MINVAL_F : 0.0f);
    }
  };
  auto sdiv = [](float a, float b) { return b == 0.0f ? 0.0f : a / b; };
  float al[3] = {0, 0, 0}, cs[3], e0[3], e1[3];
  ls_eval3(al, cs, e0, e1);
  const float p0_cost = cs[0], p0_d0 = e0[0], p0_d1 = e1[0];
  al[0] = -sdiv(p0_d0, p0_d1); al[1] = al[0]; al[2] = al[0];
  ls_eval3(al, cs, e0, e1);
  float lo_a, lo_c, lo_d0, lo_d1, hi_a, hi_c, hi_d0, hi_d1;
  if (e0[0] < p0_d0) { lo_a = al[0]; lo_c = cs[0]; lo_d0 = e0[0]; lo_d1 = e1[0]; hi_a = 0; hi_c = p0_cost; hi_d0 = p0_d0; hi_d1 = p0_d1; }
  else { hi_a = al[0]; hi_c = cs[0]; hi_d0 = e0[0]; hi_d1 = e1[0]; lo_a = 0; lo_c = p0_cost; lo_d0 = p0_d0; lo_d1 = p0_d1; }
  bool swap = true;
  for (int it = 0; it < m->ls_iterations; it++) {
    bool done = !swap || (lo_d0 < 0 && lo_d0 > -gtol) || (hi_d0 > 0 && hi_d0 < gtol);
    if (done) break;
    al[0] = lo_a - sdiv(lo_d0, lo_d1); al[1] = hi_a - sdiv(hi_d0, hi_d1); al[2] = 0.5f * (lo_a + hi_a);
    ls_eval3(al, cs, e0, e1);
    // lo_next = 0, hi_next = 1, mid = 2
    const bool s1 = (lo_d0 > 0) || (lo_d0 < e0[0]);
    if (s1) { lo_a = al[0]; lo_c = cs[0]; lo_d0 = e0[0]; lo_d1 = e1[0]; }
    const bool s2 = (e0[2] < 0) && (lo_d0 < e0[2]);
    if (s2) { lo_a = al[2]; lo_c = cs[2]; lo_d0 = e0[2]; lo_d1 = e1[2]; }
    const bool s3 = (e0[1] < 0) && (lo_d0 < e0[1]);
    if (s3) { lo_a = al[1]; lo_c = cs[1]; lo_d0 = e0[1]; lo_d1 = e1[1]; }
    const bool s4 = (hi_d0 < 0) || (hi_d0 > e0[1]);
    if (s4) { hi_a = al[1]; hi_c = cs[1]; hi_d0 = e0[1]; hi_d1 = e1[1]; }
    const bool s5 = (e0[2] > 0) && (hi_d0 > e0[2]);
    if (s5) { hi_a = al[2]; hi_c = cs[2]; hi_d0 = e0[2]; hi_d1 = e1[2]; }
    const bool s6 = (e0[0] > 0) && (hi_d0 > e0[0]);
    if (s6) { hi_a = al[0]; hi_c = cs[0]; hi_d0 = e0[0]; hi_d1 = e1[0]; }
    swap = s1 || s2 || s3 || s4 || s5 || s6;
  }
  const bool improved = (lo_c < p0_cost) || (hi_c < p0_cost);
  const float alpha = improved ? (lo_c < hi_c ? lo_a : hi_a) : 0.0f;
  if (lane < NV) {
    const float xa = X[lane] + alpha * GRAD[lane];
    X[lane] = xa;
    WARM[lane] = xa;
  }
  if (lane == 0) { SCR[S::S_MISC + 1] = alpha; SCR[S::S_MISC + 2] = use_warm ? 1.0f : 0.0f; SCR[S::S_MISC + 3] = use_warm ? cost_w : cost_s; }
  ODK_SYNC();

  ODK_PROF(14);
  // ---------------- P10: sensors (lane = sensor), only when requested
  if (flags & 1) {
    if (lane < m->nsensor) {
      const int s = lane, site = m->sensor_site[s], b = m->site_body[site], type = m->sensor_type[s];
      float R[9], Rs[9], sp[3], dif[3];
      for (int k = 0; k < 9; k++) R[k] = XMAT[k * NB + b];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rs[3 * i + j] = R[3 * i] * m->site_mat[site][j] + R[3 * i + 1] * m->site_mat[site][3 + j] + R[3 * i + 2] * m->site_mat[site][6 + j];
      for (int k = 0; k < 3; k++) {
        sp[k] = XPOS[k * NB + b] + R[3 * k] * m->site_pos[site][0] + R[3 * k + 1] * m->site_pos[site][1] + R[3 * k + 2] * m->site_pos[site][2];
        dif[k] = sp[k] - ref[k];
      }
      const int slot = (b == m->base_body) ? 0 : (b == m->foot_body[0] ? 1 : 2);
      const float* cv = SCR + S::S_SV + 6 * slot;
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
        for (int k = 0; k < 6; k++) ca[k] = SCR[S::S_CA + k];
        for (int d = 0; d < 6; d++) {
          const float qa = X[d];
          for (int k = 0; k < 6; k++) ca[k] += CDOF[k * NV + d] * qa;
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
      else if (type == 8) {  // framequat on the base: qpos quaternion times site quaternion
        float qb[4] = {QPOS[3], QPOS[4], QPOS[5], QPOS[6]}, qs[4];
        qnormalize(qb);
        qmul(qs, qb, m->site_quat[site]);
        qnormalize(qs);
        out[0] = qs[0]; out[1] = qs[1]; out[2] = qs[2]; out[3] = qs[3];
      }
    }
    // feet site heights + imu site rotation for the env logic
    if (lane < 2) {
      const int site = m->site_feet[lane], b = m->site_body[site];
      SCR[S::S_MISC + 8 + lane] = XPOS[2 * NB + b] + XMAT[6 * NB + b] * m->site_pos[site][0] + XMAT[7 * NB + b] * m->site_pos[site][1] + XMAT[8 * NB + b] * m->site_pos[site][2];
    }
    if (lane >= 2 && lane < 5) {  // gravity = site_xmat[imu]^T (0,0,-1) = -(third row of the site rotation)
      const int site = m->site_imu, b = m->site_body[site], c = lane - 2;
      float v = 0;
      for (int k = 0; k < 3; k++) v += XMAT[(6 + k) * NB + b] * m->site_mat[site][3 * k + c];
      SCR[S::S_MISC + 10 + c] = -v;
    }
    ODK_SYNC();
  }
  ODK_PROF(17);
}

// mjx forward.euler (eulerdamp disabled): qvel += dt qacc; qpos integrated with the NEW qvel
template <class S, int G>
__device__ void euler_env(float* __restrict__ L, const DevModel* __restrict__ m, int lane) {
  float* QPOS = L + S::O_QPOS; float* QVEL = L + S::O_QVEL; const float* X = L + S::O_X;
  const float dt = m->dt;
  if (lane < S::NV) QVEL[lane] += dt * X[lane];
  ODK_SYNC();
  if (lane < m->nj) {
    const int j = lane;
    if (j == 0) {
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
    } else {
      QPOS[m->jnt_qposadr[j]] += dt * QVEL[m->jnt_dofadr[j]];
    }
  }
  ODK_SYNC();
}

}  // namespace odk
