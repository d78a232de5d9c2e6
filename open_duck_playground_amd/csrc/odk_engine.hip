// odk_engine.hip -- fused env-step kernels + the C-ABI of include/odk.h (libodk.so).
//
// Host side: blob -> DevModel, device buffers, launches on the caller's stream.  Device side:
// reset / step / physics-only kernels built from odk_kernels.h.  Env logic follows the reference
// playground/open_duck_mini_v2/joystick.py (line map next to each block) and the brax
// Episode/AutoReset wrappers.  gfx950 only; no CPU fallback of any kind.
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <string>
#include <vector>

#include "../../include/odk.h"
#include "odk_kernels.h"

using namespace odk;

// ================================================================================================
// per-env HBM records (floats; ints stored bit-exact in float slots).  The carried info of joystick.py:278-302 + the wrappers' additions,
// sized by the robot's actuator count nu (the duck: 14 -> the offsets of rounds 1-5: LAST 7, LAST2 21, ..., AHIST 69, IMU 111, NINFO 141)
struct RecLay {
  int CMD, LAST, LAST2, LAST3, MT, AIR, PEAK, PUSH, AHIST, IMU, EPSTEPS, TRUNC, DONE, EPSUM, EPLEN, EPMET, KEY0, KEY1, CTR, STEP, PSTEP, PINT, IMI, LCON, NINFO;
};
constexpr RecLay rec_lay(int nu) {
  RecLay r{};
  r.CMD = 0; r.LAST = 7; r.LAST2 = r.LAST + nu; r.LAST3 = r.LAST2 + nu; r.MT = r.LAST3 + nu; r.AIR = r.MT + nu; r.PEAK = r.AIR + 2; r.PUSH = r.PEAK + 2;
  r.AHIST = r.PUSH + 2; r.IMU = r.AHIST + 3 * nu; r.EPSTEPS = r.IMU + 9; r.TRUNC = r.EPSTEPS + 1; r.DONE = r.TRUNC + 1; r.EPSUM = r.DONE + 1; r.EPLEN = r.EPSUM + 1;
  r.EPMET = r.EPLEN + 1; r.KEY0 = r.EPMET + ODK_NMETRIC; r.KEY1 = r.KEY0 + 1; r.CTR = r.KEY1 + 1; r.STEP = r.CTR + 1; r.PSTEP = r.STEP + 1; r.PINT = r.PSTEP + 1;
  r.IMI = r.PINT + 1; r.LCON = r.IMI + 1; r.NINFO = r.LCON + 1;
  return r;
}
static_assert(rec_lay(14).AHIST == 69 && rec_lay(14).IMU == 111 && rec_lay(14).EPSTEPS == 120 && rec_lay(14).KEY0 == 133 && rec_lay(14).NINFO == 141, "the duck's record layout");

// Observation row strides (joystick.py:570-615 / standing.py:524-565) for a robot with nu actuators: the duck's 101 / 212 and 85 / 153
constexpr int obs_nobs(int nu, bool standing) { return standing ? 15 + 5 * nu : 17 + 6 * nu; }
constexpr int obs_npriv(int nu, bool standing) { return obs_nobs(nu, standing) + 26 + 3 * nu + (standing ? 0 : 43); }
static_assert(obs_nobs(14, false) == ODK_NOBS && obs_npriv(14, false) == ODK_NPRIV && obs_nobs(14, true) == ODK_NOBS_STANDING && obs_npriv(14, true) == ODK_NPRIV_STANDING, "include/odk.h");
// Random draws of an env step (stream definition shared with oracle/odk_oracle_env.c): 0 action delay | 2, 3 push | 4-6 gyro | 7-9 accelerometer |
// 10-12 gravity | 13 .. 12 + nu joint angles | 13 + nu .. 12 + 2 nu joint velocities | 13 + 2 nu .. 19 + 2 nu command | 20 + 2 nu zero-command
// (the duck: 13, 27, 41, 48).  Reset stream: 0-1 dxy | 2 yaw | 3 .. 2 + nu joint scale | 3 + nu .. 8 + nu base qvel | 9 + nu .. 15 + nu command |
// 16 + nu zero-command | 17 + nu push interval (the duck: 17, 23, 30, 31).
constexpr int draw_qvel(int nu) { return 13 + nu; }
constexpr int draw_cmd(int nu) { return 13 + 2 * nu; }
constexpr int draw_count(int nu) { return (17 + 2 * nu + 1) & ~1; }      // draws 4 .. 20 + 2 nu, rounded up to whole generator blocks (the duck: 46)

template <class S> struct Rec {
  static constexpr RecLay L = rec_lay(S::NU);
  static constexpr int NOBS = obs_nobs(S::NU, false), NPRIV = obs_npriv(S::NU, false);
  static constexpr int INFO = S::NQ + 2 * S::NV;
  static constexpr int SIZE = ((INFO + L.NINFO + 3) / 4) * 4;
  static constexpr int FOBS = S::NQ + 2 * S::NV;
  static constexpr int FSIZE = ((FOBS + NOBS + NPRIV + 3) / 4) * 4;
};

// extra LDS used by the env logic, placed after the physics arrays
template <class S> struct EnvL {
  static constexpr int O_INFO = S::TOTAL;               // [N_INFO] the carried info (Rec::L)
  static constexpr int O_ACT = O_INFO + S::N_INFO;      // [N_ACT] this step's action, then the imitation phase (2)
  static_assert(rec_lay(S::NU).NINFO <= S::N_INFO && S::NU + 2 <= S::N_ACT, "Shape::N_INFO / N_ACT");
  // epilogue only, on top of the motion-column buffers (dead after the last forward pass): this step's random draws and the
  // reference motion (evaluated in the epilogue: reward and privileged obs are its only readers)
  static constexpr int NDRAW = draw_count(S::NU);
  static constexpr int O_NZ = S::O_BUF6;                          // [NDRAW] draw_block
  static constexpr int O_REF = S::O_BUF6 + ((NDRAW + 3) / 4) * 4;   // [40] current_reference_motion
  static_assert(((NDRAW + 3) / 4) * 4 + 40 <= 6 * S::NVR && NDRAW <= 64, "draws + reference motion must fit in BUF6");
  static constexpr int O_PRIV = S::O_M;            // [NPRIV] aliases M|HL (dead after the last forward)
  static constexpr int TOTAL = O_ACT + S::N_ACT;
  static_assert(TOTAL == S::ENV_STRIDE, "Shape::ENV_STRIDE is the distance between the two env images of a workgroup");
  static_assert(S::NMR + S::NHR >= Rec<S>::NPRIV, "privileged obs must fit in the M|HL region");
  // per WORKGROUP, behind the envs' images: static tables shared by the envs of the workgroup
  static constexpr int SHARED = S::SHARED;       // DevModel::R_ent | contact-row constants (forward_env: RT, CT)
  static constexpr int wg_floats(int envs) { return envs * TOTAL + SHARED; }
};
// the workgroup's copy of the shared tables (call with all 64 lanes; followed by a hand-off barrier at the caller)
template <class S> __device__ __forceinline__ const int* load_shared(float* lds, int envs, const DevModel* m) {
  int* RT = reinterpret_cast<int*>(lds + envs * EnvL<S>::TOTAL);
  {
    constexpr int T = (S::NMR + 63) / 64;
    int v[T];
#pragma unroll
    for (int t = 0; t < T; t++) { const int k = threadIdx.x + 64 * t; v[t] = m->R_ent[k < S::NMR ? k : 0]; }
#pragma unroll
    for (int t = 0; t < T; t++) { const int k = threadIdx.x + 64 * t; if (k < S::NMR) RT[k] = v[t]; }
  }
  float* SH = reinterpret_cast<float*>(RT);
  float* CT = SH + S::SH_CT;
  const int k = threadIdx.x;
  if (k < 3) { CT[k] = m->pair_mu[k]; CT[3 + k] = m->pair_invweight[k]; }
  if (k < 27) CT[6 + k] = m->pair_imp[k / 9][k % 9];
  if (k < 9) CT[33 + k] = m->plane_frame[k];
  return RT;
}

using ShapeA = Shape<21, 20, 18, 14, 15, 145, 170, 76, 10, 15>;   // flat_terrain
using ShapeB = Shape<31, 30, 18, 14, 25, 285, 385, 86, 15, 25>;   // *_backlash
// the same two with the elliptic-cone code compiled in (Shape::ELL): launched for a duck model with <option cone="elliptic"> (plane floor, or
// the backlash model's height field), 32 lanes per env; the default kernels above stay the instruction streams they were
using ShapeAE = Shape<21, 20, 18, 14, 15, 145, 170, 76, 10, 15, true>;
using ShapeBE = Shape<31, 30, 18, 14, 25, 285, 385, 86, 15, 25, true>;
// A robot that is not the duck (tests/assets/tail_biped.xml: biped with a five-link tail, 21 dofs, 15 actuators, 19 bodies,
// box feet): reset / step / physics kernels -- the env kernels' task logic is joystick.py's with the robot's own tables (rec_lay, obs_nobs: sized
// by Shape::NU; actuators, default pose, sites and sensor addresses from the ModelBlob), imitation and Standing stay the duck's.  What adding it
// took: this line, the dispatch lines below that name it (tools/new_shape.py prints both for an XML), and nothing in odk_kernels.h beyond
// admitting nv = 21 to the chain solver.
using ShapeC = Shape<22, 21, 19, 15, 16, 156, 181, 78, 10, 15>;
// A second one (tests/assets/biped12.xml): a biped with SIX-dof legs (hip yaw / roll / pitch, knee, ankle pitch / roll), 18 dofs, 12 actuators,
// 16 bodies: serial chains of six (the chain solve's block size is the shape's CL), contact wrenches in their own floats (16 bodies' cfrc | crb
// region is too small for them).  Env kernels as for ShapeC (12 actions, observations 89 / 194 floats).
using ShapeD = Shape<19, 18, 16, 12, 13, 135, 171, 72, 12, 18, false, 6, true>;      // (chains of six; equality rows and elliptic cones compiled in, as ShapeC)
// The compiled model shapes, by the index odk_model carries: every per-shape dispatch of the host code below goes through this list, so a
// new robot is ONE `using` line above and ONE entry here (tools/new_shape.py <xml> prints both).  Entries 0 and 1 are the duck's two models
// (their cone / height-field / 64-lane instantiations are chosen in launch()); entries from 2 on run reset / step / physics kernels at 32
// lanes per env on a plane floor.
// Robots added without editing this file: `python tools/new_shape.py robot.xml --add` writes csrc/odk_shapes_user.h -- one `using ShapeU<k> = Shape<...>;`
// line per robot and `#define ODK_USER_SHAPES(X) X(4, ShapeU0) ...` -- and rebuilds the library.
#if __has_include("odk_shapes_user.h")
#include "odk_shapes_user.h"
#endif
#ifndef ODK_USER_SHAPES
#define ODK_USER_SHAPES(X)
#endif
#define ODK_SHAPES(X) X(0, ShapeA) X(1, ShapeB) X(2, ShapeC) X(3, ShapeD) ODK_USER_SHAPES(X)

struct KArgs {
  const DevModel* m;
  DevPRM prm;         // by value (232 bytes of kernel arguments): the grid searches read scalar registers, not 20 dependent loads
  const float* prm_table;
  float* recs;        // [nenv][Rec::SIZE]
  float* first;       // [nenv][Rec::FSIZE]
  const float* dr;    // [nenv][NDR] or null
  const float* action;  // [nenv][nu]
  const float* hfield;  // [nrow][ncol] height-field samples in [0, 1], or null (plane floor)
  float* obs; float* priv; float* reward; float* done; float* trunc; float* metrics;
  float* dbg_lds;     // [nenv][TOTAL] or null: LDS image after the last forward
  int nenv;
  uint32_t seed, env_offset;
  int n_substeps;
  EnvCfg cfg;
};

// DR buffer layout per env
template <class S> struct DRL {
  static constexpr int MASS = 0, IPOS = S::NB, FRL = S::NB + 3, ARM = FRL + S::NU, Q0 = ARM + S::NU, KP = Q0 + S::NU, SIZE = KP + S::NU;
};

__device__ __forceinline__ float i2f(int v) { return __int_as_float(v); }
__device__ __forceinline__ int f2i(float v) { return __float_as_int(v); }
__device__ __forceinline__ float nan_to_num(float x) {
  if (isnan(x)) return 0.0f;
  if (isinf(x)) return x > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return x;
}

// Debug image of the env's LDS (parity tests).  Out of line and rolled: unrolled in place, its 79 per-lane 64-bit store
// addresses were hoisted above the substep loop and parked in scratch (158 dwords per lane, ~400 MB of HBM per launch).
template <class S, int G>
__device__ __noinline__ void dump_lds(float* dbg, const float* L, int env, int lane) {
  float* o = dbg + (size_t)env * S::TOTAL;
#pragma unroll 1
  for (int k = lane; k < S::TOTAL; k += G) o[k] = L[k];
}

// Global -> LDS copies of the prologue.  Written as "all loads, then all stores" with compile-time trip counts: as
// `for (i = lane; i < N; i += G) dst[i] = src[i]` every trip was its own load -> s_waitcnt vmcnt(0) -> ds_write round trip
// (~25 serialised global round trips per env step: most of the 40 us a zero-substep launch took).
template <int N, int G> struct G2L {
  static constexpr int T = (N + G - 1) / G;
  float v[T];
  __device__ __forceinline__ void load(const float* __restrict__ src, int lane) {
#pragma unroll
    for (int t = 0; t < T; t++) { const int i = lane + t * G; v[t] = src[i < N ? i : 0]; }
  }
  __device__ __forceinline__ void store(float* dst, int lane) const {
#pragma unroll
    for (int t = 0; t < T; t++) { const int i = lane + t * G; if (i < N) dst[i] = v[t]; }
  }
};

// effective per-env model parameters -> LDS: nominal values from the model, domain-randomised ones (dr != null) on top
template <class S, int G>
struct ParamLoad {
  float q0, mass, arm, frl, kp, ipos, dq0, darm, dfrl;
  int qadr, dadr;
  __device__ __forceinline__ void load(const DevModel* __restrict__ m, const float* __restrict__ dr, int lane) {
    static_assert(S::NQ <= G && S::NV <= G && S::NB <= G && S::NU <= G, "one parameter of each kind per lane");
    const int iq = lane < S::NQ ? lane : 0, ib = lane < S::NB ? lane : 0, iv = lane < S::NV ? lane : 0, iu = lane < S::NU ? lane : 0, i3 = lane < 3 ? lane : 0;
    q0 = m->qpos0[iq];
    mass = dr ? dr[DRL<S>::MASS + ib] : m->body_mass[ib];
    arm = m->dof_armature[iv]; frl = m->dof_frictionloss[iv];
    kp = dr ? dr[DRL<S>::KP + iu] : m->act_kp[iu];
    ipos = dr ? dr[DRL<S>::IPOS + i3] : m->body_ipos[1][i3];
    qadr = m->act_qposadr[iu]; dadr = m->act_dofadr[iu];
    dq0 = dr ? dr[DRL<S>::Q0 + iu] : 0.0f; darm = dr ? dr[DRL<S>::ARM + iu] : 0.0f; dfrl = dr ? dr[DRL<S>::FRL + iu] : 0.0f;
  }
  __device__ __forceinline__ void store(float* L, bool has_dr, int lane) const {
    if (lane < S::NQ) L[S::O_Q0 + lane] = q0;
    if (lane < S::NB) L[S::O_MASS + lane] = mass;
    if (lane < S::NV) { L[S::O_ARM + lane] = arm; L[S::O_FRL + lane] = frl; }
    if (lane < S::NU) L[S::O_KP + lane] = kp;
    if (lane < 3) L[S::O_IPOS1 + lane] = ipos;
    ODK_SYNC();   // the actuated joints' randomised values go on top of the nominal ones written by other lanes
    if (has_dr && lane < S::NU) { L[S::O_Q0 + qadr] = dq0; L[S::O_ARM + dadr] = darm; L[S::O_FRL + dadr] = dfrl; }
    ODK_SYNC();
  }
};
template <class S, int G>
__device__ __forceinline__ void load_params(float* L, const DevModel* m, const float* dr, int lane) {
  ParamLoad<S, G> p;
  p.load(m, dr, lane);
  p.store(L, dr != nullptr, lane);
}

// PolyReferenceMotion.get_reference_motion (reference poly_reference_motion.py:148-168): float32 fma Horner
__device__ __forceinline__ int prm_nearest(const float* grid, int n, float v) {   // grid: kernel-argument array, fully unrolled
  int best = 0;
  float bd = fabsf(grid[0] - v);
#pragma unroll
  for (int i = 1; i < 16; i++) { const float d = fabsf(grid[i] - v); if (i < n && d < bd) { bd = d; best = i; } }
  return best;
}
template <int G>
__device__ __forceinline__ void prm_eval(const DevPRM* __restrict__ p, const float* table, float dx, float dy, float dth, int i, float* out, int lane) {
  const float x = fminf(fmaxf(dx, p->ranges[0]), p->ranges[1]);
  const float y = fminf(fmaxf(dy, p->ranges[2]), p->ranges[3]);
  const float t3 = fminf(fmaxf(dth, p->ranges[4]), p->ranges[5]);
  const int ix = prm_nearest(p->dxs, p->nx, x), iy = prm_nearest(p->dys, p->ny, y), it = prm_nearest(p->dths, p->nth, t3);
  float t = (float)(i % p->nsteps) / (float)p->nsteps;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float* c = table + ((size_t)((ix * p->ny + iy) * p->nth + it)) * 640;
  for (int k = lane; k < 40; k += G) {
    float yv = c[k * 16];
#pragma unroll
    for (int q = 1; q < 16; q++) yv = fmaf(yv, t, c[k * 16 + q]);
    out[k] = yv;
  }
}
// the same evaluation into two registers per lane (dims lane and lane + 32; G = 32 or 64): the step kernel evaluates the
// reference motion in its prologue, where the table loads overlap the state loads, and parks it in LDS only in the epilogue
template <int G>
__device__ __forceinline__ void prm_eval_regs(const DevPRM* p, const float* table, float dx, float dy, float dth, int i, float& r0, float& r1, int lane) {
  const float x = fminf(fmaxf(dx, p->ranges[0]), p->ranges[1]);
  const float y = fminf(fmaxf(dy, p->ranges[2]), p->ranges[3]);
  const float t3 = fminf(fmaxf(dth, p->ranges[4]), p->ranges[5]);
  const int ix = prm_nearest(p->dxs, p->nx, x), iy = prm_nearest(p->dys, p->ny, y), it = prm_nearest(p->dths, p->nth, t3);
  float t = (float)(i % p->nsteps) / (float)p->nsteps;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float* c = table + ((size_t)((ix * p->ny + iy) * p->nth + it)) * 640;
  const int k0 = lane < 40 ? lane : 0, k1 = lane + 32 < 40 ? lane + 32 : 0;
  float a = c[k0 * 16], b = c[k1 * 16];
#pragma unroll
  for (int q = 1; q < 16; q++) { a = fmaf(a, t, c[k0 * 16 + q]); b = fmaf(b, t, c[k1 * 16 + q]); }
  r0 = a; r1 = b;
}

// sample_command (joystick.py:671-725); draws base..base+7 of stream (k0,k1,ctr)
__device__ inline void sample_command(const EnvCfg& c, uint32_t k0, uint32_t k1, uint32_t ctr, uint32_t base, int k, float& out) {
  const float z = rng_uniform(k0, k1, ctr, base + 7);
  const float u = rng_uniform(k0, k1, ctr, base + k);
  out = (z < 0.1f) ? 0.0f : c.cmd_range[k][0] + u * (c.cmd_range[k][1] - c.cmd_range[k][0]);
}

// Draws 4 .. 3 + NDRAW of stream (k0, k1, ctr) in ONE threefry evaluation: lane l < NDRAW / 2 computes block l + 2, whose two words are
// draws 4 + 2 l and 5 + 2 l; NZ[i - 4] = draw i.  (The observation noise and the command resampling used to call the
// generator from inside divergent branches: ~8 serial threefry evaluations per env step.)
template <int NDRAW>
__device__ __forceinline__ void draw_block(uint32_t k0, uint32_t k1, uint32_t ctr, float* NZ, int lane) {
  uint32_t a, b;
  threefry2x32(k0, k1, ctr, (uint32_t)(lane + 2), a, b);
  if (lane < NDRAW / 2) { NZ[2 * lane] = (float)(a >> 8) * (1.0f / 16777216.0f); NZ[2 * lane + 1] = (float)(b >> 8) * (1.0f / 16777216.0f); }
  ODK_SYNC();
}

// _get_obs (joystick.py:487-620): builds privileged_state[212] (whose first 101 entries are `state`) in LDS; NZ: draw_block
// KIND (0 Joystick, 1 Standing) and the element loop are compile-time: element ks of pass `it` is lane + G it, so every
// pass keeps only the few layout segments its 32 / 64 elements can fall into (as one runtime loop over c.npriv, each of the
// seven passes walked all 23 segments' divergent branches).
template <class S, int G, int KIND>
__device__ __forceinline__ void build_obs_kind(float* L, const DevModel* m, const EnvCfg& c, const float* contact, const float* NZ,
                          int imitation_i, const float* phase, int lane) {
  using E = EnvL<S>;
  float* P = L + E::O_PRIV; float* INFO = L + E::O_INFO; const float* SENS = L + S::O_SENS; const float* SCR = L + S::O_SCR;
  const float* QPOS = L + S::O_QPOS; const float* QVEL = L + S::O_QVEL;
  const float lvl = c.noise_level;
  constexpr int NU = S::NU;
  constexpr RecLay RL = rec_lay(NU);
  constexpr int NOBS = obs_nobs(NU, false), DQV = draw_qvel(NU) - 4;   // (draw i sits at NZ[i - 4])
  const float con0 = contact[0], con1 = contact[1], ph0 = phase[0], ph1 = phase[1];
  const int adr_gyro = m->adr_gyro, adr_acc = m->adr_accelerometer, adr_lin = m->adr_local_linvel, adr_ang = m->adr_global_angvel;   // (read once, up front)
  // imu history ring (noisy gravity, never emitted: joystick.py:522-530)
  float ng = 0;
  if (lane < 3) ng = SCR[S::S_MISC + 10 + lane] + (2.0f * NZ[10 - 4 + lane] - 1.0f) * lvl * c.noise_gravity;
  float h0 = 0, h1 = 0;
  if (lane < 3) { h0 = INFO[RL.IMU + lane]; h1 = INFO[RL.IMU + 3 + lane]; }
  ODK_SYNC();
  if (lane < 3) { INFO[RL.IMU + lane] = ng; INFO[RL.IMU + 3 + lane] = h0; INFO[RL.IMU + 6 + lane] = h1; }
  // Standing (standing.py:524-565) = the Joystick layout minus motor_targets, imitation phase, reference motion, imitation_i
  constexpr bool standing = KIND != 0;
  constexpr int NP = obs_npriv(NU, standing);
#pragma unroll
  for (int it = 0; it < (NP + G - 1) / G; it++) {
    const int ks = lane + it * G;
    __builtin_assume(ks >= it * G && ks < it * G + G);
    if (ks >= NP) continue;
    const int k = !standing ? ks : (ks < 13 + 5 * NU ? ks : (ks < 15 + 5 * NU ? ks + NU : ks + NOBS - (15 + 5 * NU)));
    // This element's reads of the model's tables, ALL AT ONCE and before the case analysis (indices clamped: a lane outside a case reads a valid slot and drops the
    // value; cases that cannot occur in this unrolled iteration lose their reads with them).  Inside the cases every read was a global load followed by its own
    // wait -- ~40 exposed round trips per env step in this routine (round 6).
    const int q = k - NOBS;   // privileged tail (joystick.py:596-615)
    const int uA = min(max(k - 13, 0), NU - 1), uV = min(max(k - 13 - NU, 0), NU - 1), uQ = min(max(q - 15, 0), NU - 1), uQV = min(max(q - 15 - NU, 0), NU - 1);
    const int tF = min(max(q - 18 - 3 * NU, 0), 5);
    const int A_bq = m->act_backlash_qposadr[uA], A_aq = m->act_qposadr[uA], V_ad = m->act_dofadr[uV];
    const float A_kc = m->key_ctrl[uA], A_ns = c.qpos_noise_scale[uA];
    const int Q_bq = m->act_backlash_qposadr[uQ], Q_aq = m->act_qposadr[uQ], QV_ad = m->act_dofadr[uQV], F_adr = m->adr_foot_linvel[tF >= 3 ? 1 : 0];
    const float Q_kc = m->key_ctrl[uQ];
    {   // (the optimiser sinks a read into the one case that uses it, which is where it came from: the values a case of THIS unrolled iteration can use are pinned here,
        // all in flight together; `it` is a constant after unrolling, so the tests below cost nothing)
      const int k_lo = standing ? 0 : it * G, k_hi = standing ? NP + NOBS : it * G + G;      // (Standing re-maps ks: every case stays possible)
      auto hits = [&](int lo, int hi) { return k_lo < hi && k_hi > lo; };
      if (hits(13, 13 + NU)) asm volatile("" :: "v"(A_bq), "v"(A_aq), "v"(A_kc), "v"(A_ns));
      if (hits(13 + NU, 13 + 2 * NU)) asm volatile("" :: "v"(V_ad));
      if (hits(NOBS + 15, NOBS + 15 + NU)) asm volatile("" :: "v"(Q_bq), "v"(Q_aq), "v"(Q_kc));
      if (hits(NOBS + 15 + NU, NOBS + 15 + 2 * NU)) asm volatile("" :: "v"(QV_ad));
      if (hits(NOBS + 18 + 3 * NU, NOBS + 24 + 3 * NU)) asm volatile("" :: "v"(F_adr));
    }
    float v = 0;
    if (k < 3) v = SENS[adr_gyro + k] + (2.0f * NZ[k] - 1.0f) * lvl * c.noise_gyro;
    else if (k < 6) v = SENS[adr_acc + k - 3] + (2.0f * NZ[k] - 1.0f) * lvl * c.noise_accelerometer;
    else if (k < 13) v = INFO[RL.CMD + k - 6];
    else if (k < 13 + NU) {
      const int u = k - 13;
      const float ja = QPOS[A_aq] + (A_bq >= 0 ? QPOS[A_bq] : 0.0f);
      v = ja + (2.0f * NZ[13 - 4 + u] - 1.0f) * lvl * A_ns - A_kc;
    } else if (k < 13 + 2 * NU) {
      const int u = k - 13 - NU;
      v = (QVEL[V_ad] + (2.0f * NZ[DQV + u] - 1.0f) * lvl * c.noise_joint_vel) * c.dof_vel_scale;
    } else if (k < 13 + 3 * NU) v = INFO[RL.LAST + k - 13 - 2 * NU];
    else if (k < 13 + 4 * NU) v = INFO[RL.LAST2 + k - 13 - 3 * NU];
    else if (k < 13 + 5 * NU) v = INFO[RL.LAST3 + k - 13 - 4 * NU];
    else if (k < 13 + 6 * NU) v = INFO[RL.MT + k - 13 - 5 * NU];
    else if (k < 15 + 6 * NU) v = (k - 13 - 6 * NU) ? con1 : con0;   // (scalars + selects: a runtime index parks the two-element arrays in scratch)
    else if (k < 17 + 6 * NU) v = (k - 15 - 6 * NU) ? ph1 : ph0;
    else {
      if (q < 3) v = SENS[adr_gyro + q];
      else if (q < 6) v = SENS[adr_acc + q - 3];
      else if (q < 9) v = SCR[S::S_MISC + 10 + q - 6];
      else if (q < 12) v = SENS[adr_lin + q - 9];
      else if (q < 15) v = SENS[adr_ang + q - 12];
      else if (q < 15 + NU) v = QPOS[Q_aq] + (Q_bq >= 0 ? QPOS[Q_bq] : 0.0f) - Q_kc;
      else if (q < 15 + 2 * NU) v = QVEL[QV_ad];
      else if (q == 15 + 2 * NU) v = QPOS[2];
      else if (q < 16 + 3 * NU) v = L[S::O_ACTF + q - 16 - 2 * NU];
      else if (q < 18 + 3 * NU) v = (q - 16 - 3 * NU) ? con1 : con0;
      else if (q < 24 + 3 * NU) { const int t = q - 18 - 3 * NU; v = SENS[F_adr + (t >= 3 ? t - 3 : t)]; }
      else if (q < 26 + 3 * NU) v = INFO[RL.AIR + q - 24 - 3 * NU];
      else if (q < 66 + 3 * NU) v = L[E::O_REF + q - 26 - 3 * NU];
      else if (q == 66 + 3 * NU) v = (float)imitation_i;
      else v = (q - 67 - 3 * NU) ? ph1 : ph0;
    }
    P[ks] = v;
  }
  ODK_SYNC();
}
template <class S, int G>
__device__ __forceinline__ void build_obs(float* L, const DevModel* m, const EnvCfg& c, const float* contact, const float* NZ,
                          int imitation_i, const float* phase, int lane) {
  if (c.kind == 0) build_obs_kind<S, G, 0>(L, m, c, contact, NZ, imitation_i, phase, lane);
  else build_obs_kind<S, G, 1>(L, m, c, contact, NZ, imitation_i, phase, lane);
}

__device__ __forceinline__ void foot_contact_flags(const float* CDIST, float* contact) {
  for (int f = 0; f < 2; f++) {
    float md = 1e4f;
    for (int k = 0; k < 4; k++) md = fminf(md, CDIST[4 * f + k]);
    contact[f] = md < 0 ? 1.0f : 0.0f;
  }
}

template <class S, int G>
__device__ __forceinline__ void write_outputs(const KArgs& a, const float* L, int env, float reward, float done, float trunc, const float* metrics, int lane) {
  using E = EnvL<S>;
  const float* P = L + E::O_PRIV;
  const int nobs = a.cfg.nobs, npriv = a.cfg.npriv;
  if (a.obs) for (int k = lane; k < nobs; k += G) a.obs[(size_t)env * nobs + k] = P[k];
  if (a.priv) for (int k = lane; k < npriv; k += G) a.priv[(size_t)env * npriv + k] = P[k];
  if (lane == 0) {
    if (a.reward) a.reward[env] = reward;
    if (a.done) a.done[env] = done;
    if (a.trunc) a.trunc[env] = trunc;
  }
  if (a.metrics && lane < ODK_NMETRIC) a.metrics[(size_t)env * ODK_NMETRIC + lane] = metrics[lane];
}

// ================================================================================================
// Joystick.reset (joystick.py:206-321) + Episode/AutoReset wrapper resets
template <class S, int G, int HF>
__global__ void __launch_bounds__(64) reset_kernel(KArgs a) {
  extern __shared__ float lds[];
  using E = EnvL<S>; using R = Rec<S>;
  constexpr int NU = S::NU;
  constexpr RecLay RL = rec_lay(NU);
  const int slot = threadIdx.x / G, lane = threadIdx.x % G;
  const int env = blockIdx.x * (64 / G) + slot;
  const bool live = env < a.nenv;
  const int e = live ? env : a.nenv - 1;
  float* L = lds + slot * E::TOTAL;
  const int* RT = load_shared<S>(lds, 64 / G, a.m);   // ordered before its first use by the ODK_SYNCs below
#ifdef ODK_POISON_LDS   // debug build: every read of LDS that was not written by this launch surfaces as NaN
  for (int k = lane; k < E::TOTAL; k += G) L[k] = __int_as_float(0x7fc00000);
  ODK_SYNC();
#endif
  const DevModel* m = a.m;
  const EnvCfg& c = a.cfg;
  float* INFO = L + E::O_INFO;
  load_params<S, G>(L, m, a.dr ? a.dr + (size_t)e * DRL<S>::SIZE : nullptr, lane);
  Statics<S, G> st;
  load_statics<S, G>(st, m, lane);
  uint32_t k0, k1;
  threefry2x32(a.seed, 0x4F444B31u, a.env_offset + (uint32_t)e, 0u, k0, k1);
  const uint32_t kr = k1 ^ 0x52535421u;
  for (int i = lane; i < S::NQ; i += G) L[S::O_QPOS + i] = m->key_qpos[i];
  for (int i = lane; i < S::NV; i += G) { L[S::O_QVEL + i] = 0; L[S::O_WARM + i] = 0; }
  for (int i = lane; i < S::N_INFO; i += G) INFO[i] = 0;
  ODK_SYNC();
  if (lane < 2) L[S::O_QPOS + lane] += -0.05f + rng_uniform(k0, kr, 0, lane) * 0.1f;
  if (lane == 2) {
    const float yaw = -3.14f + rng_uniform(k0, kr, 0, 2) * 6.28f;
    float s, co;
    sincosf(0.5f * yaw, &s, &co);
    float q0[4] = {L[S::O_QPOS + 3], L[S::O_QPOS + 4], L[S::O_QPOS + 5], L[S::O_QPOS + 6]}, qz[4] = {co, 0, 0, s}, r[4];
    qmul(r, q0, qz);
    for (int k = 0; k < 4; k++) L[S::O_QPOS + 3 + k] = r[k];
  }
  if (lane >= 3 && lane < 9) L[S::O_QVEL + lane - 3] = -c.reset_base_qvel + rng_uniform(k0, kr, 0, 3 + NU + lane - 3) * (2.0f * c.reset_base_qvel);
  for (int u = lane; u < S::NU; u += G) {
    const float v = L[S::O_QPOS + m->act_qposadr[u]] * (0.5f + rng_uniform(k0, kr, 0, 3 + u));
    L[S::O_QPOS + m->act_qposadr[u]] = v;
    L[S::O_CTRL + u] = v;
    INFO[RL.MT + u] = c.kind != 0 ? 0.0f : m->key_ctrl[u];   // standing.py:279 starts from zeros
  }
  if (lane < 7) sample_command(c, k0, kr, 0, 9 + NU, lane, INFO[RL.CMD + lane]);
  ODK_SYNC();
  forward_env<S, G, HF>(L, RT, m, a.hfield, st, lane, 1);
  if (a.dbg_lds && live) dump_lds<S, G>(a.dbg_lds, L, env, lane);
  const float pint = c.push_interval_range[0] + rng_uniform(k0, kr, 0, 17 + NU) * (c.push_interval_range[1] - c.push_interval_range[0]);
  const int push_interval_steps = (int)rintf(pint / c.ctrl_dt);
  if (c.use_imitation) prm_eval<G>(&a.prm, a.prm_table, INFO[RL.CMD], INFO[RL.CMD + 1], INFO[RL.CMD + 2], 0, L + E::O_REF, lane);
  else for (int k = lane; k < 40; k += G) L[E::O_REF + k] = 0;
  ODK_SYNC();
  float contact[2];
  foot_contact_flags(L + S::O_CDIST, contact);
  const float phase[2] = {0, 0};
  // stash state before the obs overwrites the M|HL region?  (qpos/qvel/warm live elsewhere: safe)
  draw_block<E::NDRAW>(k0, k1, 0u, L + E::O_NZ, lane);   // the motion-column buffers are dead after the forward pass
  build_obs<S, G>(L, m, c, contact, L + E::O_NZ, 0, phase, lane);
  if (lane == 0) {
    INFO[RL.KEY0] = i2f((int)k0); INFO[RL.KEY1] = i2f((int)k1); INFO[RL.CTR] = i2f(1);
    INFO[RL.STEP] = i2f(0); INFO[RL.PSTEP] = i2f(0); INFO[RL.PINT] = i2f(push_interval_steps);
    INFO[RL.IMI] = i2f(0); INFO[RL.LCON] = i2f(0);
  }
  ODK_SYNC();
  if (live) {
    float* rc = a.recs + (size_t)env * R::SIZE;
    float* fs = a.first + (size_t)env * R::FSIZE;
    for (int i = lane; i < S::NQ; i += G) { rc[i] = L[S::O_QPOS + i]; fs[i] = L[S::O_QPOS + i]; }
    for (int i = lane; i < S::NV; i += G) {
      rc[S::NQ + i] = L[S::O_QVEL + i]; fs[S::NQ + i] = L[S::O_QVEL + i];
      rc[S::NQ + S::NV + i] = L[S::O_WARM + i]; fs[S::NQ + S::NV + i] = L[S::O_WARM + i];
    }
    for (int k = lane; k < R::NPRIV; k += G) {
      if (k < R::NOBS) fs[R::FOBS + k] = L[E::O_PRIV + k];
      fs[R::FOBS + R::NOBS + k] = L[E::O_PRIV + k];
    }
    for (int k = lane; k < RL.NINFO; k += G) rc[R::INFO + k] = INFO[k];
    float metrics[ODK_NMETRIC] = {0, 0, 0, 0, 0, 0, 0, 0};
    write_outputs<S, G>(a, L, env, 0.0f, 0.0f, 0.0f, metrics, lane);
  }
}

// ================================================================================================
// AutoReset.step -> Episode.step -> Joystick.step (joystick.py:323-481), all substeps fused
#ifndef ODK_STEP_WAVES
#define ODK_STEP_WAVES 2     // waves per SIMD the register allocation is held to (experiment builds: 4 = <= 128 VGPRs; profiles/r5/NOTES.md)
#endif
template <class S, int G, int HF>
__global__ void __launch_bounds__(64, ODK_STEP_WAVES) step_kernel(KArgs a) {
  extern __shared__ float lds[];
  using E = EnvL<S>; using R = Rec<S>;
  constexpr int NU = S::NU;
  constexpr RecLay RL = rec_lay(NU);
  const int slot = threadIdx.x / G, lane = threadIdx.x % G;
  const int env = blockIdx.x * (64 / G) + slot;
  const bool live = env < a.nenv;
  const int e = live ? env : a.nenv - 1;
  float* L = lds + slot * E::TOTAL;
#ifdef ODK_PROFILE
  const long long t_k0 = clock64();   // kernel-level stamps (profile build): prologue / substeps / epilogue pieces -> S_PROF slots 18, 19 + dbg tail
#endif
  const int* RT = load_shared<S>(lds, 64 / G, a.m);   // ordered before its first use by the ODK_SYNCs below
#ifdef ODK_POISON_LDS   // debug build: every read of LDS that was not written by this launch surfaces as NaN
  for (int k = lane; k < E::TOTAL; k += G) L[k] = __int_as_float(0x7fc00000);
  ODK_SYNC();
#endif
  const DevModel* m = a.m;
  const EnvCfg& c = a.cfg;
  float* INFO = L + E::O_INFO; float* ACT = L + E::O_ACT; float* CTRL = L + S::O_CTRL;
  float* rc = a.recs + (size_t)e * R::SIZE;
  // ---- state record, action, parameters, per-lane statics: ONE batch of global loads (coalesced: the group's lanes read
  // consecutive floats), then the LDS stores
  {
    G2L<S::NQ + 2 * S::NV, G> g_state;   // qpos|qvel|warm are contiguous in LDS too
    G2L<RL.NINFO, G> g_info;
    G2L<NU, G> g_act;
    ParamLoad<S, G> g_par;
    const float* drp = a.dr ? a.dr + (size_t)e * DRL<S>::SIZE : nullptr;
    g_state.load(rc, lane); g_info.load(rc + R::INFO, lane); g_act.load(a.action + (size_t)e * NU, lane);
    g_par.load(m, drp, lane);
    g_state.store(L + S::O_QPOS, lane); g_info.store(INFO, lane); g_act.store(ACT, lane);
    g_par.store(L, drp != nullptr, lane);   // syncs
  }
#ifdef ODK_PROFILE
  for (int k = lane; k < 36; k += G) L[S::O_SCR + S::S_PROF + k] = 0;
#endif
  Statics<S, G> st;
  load_statics<S, G>(st, m, lane);
  const uint32_t k0 = (uint32_t)f2i(INFO[RL.KEY0]), k1 = (uint32_t)f2i(INFO[RL.KEY1]), ctr = (uint32_t)f2i(INFO[RL.CTR]);
  int step = f2i(INFO[RL.STEP]), push_step = f2i(INFO[RL.PSTEP]);
  const int push_int = f2i(INFO[RL.PINT]);
  int imi = f2i(INFO[RL.IMI]);
  const int lcon = f2i(INFO[RL.LCON]);
  const float prev_done = INFO[RL.DONE];
  float ep_steps = prev_done != 0.0f ? 0.0f : INFO[RL.EPSTEPS];  // AutoReset.step prologue
  const float dt = c.ctrl_dt;
  // ---- imitation phase + reference motion (:325-355)
  float phase[2] = {0, 0};
  float ref0 = 0.0f, ref1 = 0.0f;   // current_reference_motion[lane], [lane + 32]: two registers across the substeps
  if (c.use_imitation) {
    imi = (imi + 1) % a.prm.nsteps;
    const float ph = ((float)imi / (float)a.prm.nsteps) * 2.0f * PI_F;
    phase[0] = cosf(ph); phase[1] = sinf(ph);
    prm_eval_regs<G>(&a.prm, a.prm_table, INFO[RL.CMD], INFO[RL.CMD + 1], INFO[RL.CMD + 2], imi, ref0, ref1, lane);   // (:347-353)
  } else {
    imi = 0;
  }
  // ---- action delay ring (:362-376): roll by nu, newest first
  float h0 = 0, h1 = 0;
  for (int u = lane; u < NU; u += G) { h0 = INFO[RL.AHIST + u]; h1 = INFO[RL.AHIST + NU + u]; }
  ODK_SYNC();
  for (int u = lane; u < NU; u += G) { INFO[RL.AHIST + u] = ACT[u]; INFO[RL.AHIST + NU + u] = h0; INFO[RL.AHIST + 2 * NU + u] = h1; }
  ODK_SYNC();
  uint32_t w0, w1, w2, w3;   // draws 0 | 1 (unused) and 2 | 3: two generator blocks
  threefry2x32(k0, k1, ctr, 0u, w0, w1);
  threefry2x32(k0, k1, ctr, 1u, w2, w3);
  const int aidx = randint3((float)(w0 >> 8) * (1.0f / 16777216.0f));
  // ---- push (:381-398)
  const float theta = (float)(w2 >> 8) * (1.0f / 16777216.0f) * (2.0f * PI_F);
  const float mag = c.push_magnitude_range[0] + (float)(w3 >> 8) * (1.0f / 16777216.0f) * (c.push_magnitude_range[1] - c.push_magnitude_range[0]);
  const float gate = (((push_step + 1) % push_int) == 0 ? 1.0f : 0.0f) * c.push_enable;
  const float push[2] = {cosf(theta) * gate, sinf(theta) * gate};
  if (lane < 2) L[S::O_QVEL + lane] += push[lane] * mag;
  // values only the epilogue needs go back to LDS now instead of riding through the substep loop in scratch:
  // info["push"], the imitation counter, its phase (two spare floats behind the action), the episode step counter
  if (lane == 0) {
    INFO[RL.PUSH] = push[0]; INFO[RL.PUSH + 1] = push[1];
    INFO[RL.IMI] = i2f(imi);
    ACT[NU] = phase[0]; ACT[NU + 1] = phase[1];
    INFO[RL.EPSTEPS] = ep_steps;
  }
  // ---- motor targets with speed limit (:404-417)
  for (int u = lane; u < NU; u += G) {
    float mt = m->key_ctrl[u] + INFO[RL.AHIST + aidx * NU + u] * c.action_scale;
    if (c.use_motor_speed_limits) {
      const float prev = INFO[RL.MT + u], lim = c.max_motor_velocity * dt;
      mt = fminf(fmaxf(mt, prev - lim), prev + lim);
    }
    CTRL[u] = mt;
  }
  ODK_SYNC();
#ifdef ODK_PROFILE
  const long long t_k1 = clock64();
  if (lane == 0) L[S::O_SCR + S::S_PROF + 18] = (float)(t_k1 - t_k0);
#endif
  // ---- mjx_env.step: n_substeps x (forward + Euler)   (:420)
  HotSt hot;
  if constexpr (HF == 0) load_hot<S, G>(hot, m, st, lane);
  for (int s = 0; s < a.n_substeps; s++) {
    const bool last = s == a.n_substeps - 1;
    // The model pointer is made opaque once per substep: the per-lane 64-bit table addresses (and loop-invariant table
    // loads) would otherwise be hoisted out of the loop and, with 256 VGPRs taken, parked in scratch -- ~150 dwords per
    // lane, private per wave, evicted to HBM (hundreds of MB per launch) -- while recomputing an address is one VALU op
    // and the tables themselves are 60 KB shared by every wave (L1 / L2 resident).  (Still a win with ~60 VGPRs free: -6 % without
    // the opaque pointer, -1 % without the opaque lane id -- the hoisted values lengthen live ranges, the loads they save are covered.)
    size_t opaque0 = 0;
    asm volatile("" : "+s"(opaque0));   // an offset, not the pointer itself: the address space (global) stays known
    const DevModel* ms = reinterpret_cast<const DevModel*>(reinterpret_cast<const char*>(m) + opaque0);
    // ... and so is the lane id: every lane-derived LDS address and predicate is a handful of VALU ops to rebuild, while
    // hoisted above the loop they sat in scratch (the range assumption keeps the 24-bit multiply / known-bits folds)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    __builtin_assume(lane_s >= 0 && lane_s < 64);
    if constexpr (HF == 0) forward_env<S, G, HF, true>(L, RT, ms, a.hfield, st, lane_s, last ? 1 : 0, hot);
    else forward_env<S, G, HF>(L, RT, ms, a.hfield, st, lane_s, last ? 1 : 0);
    if (last && a.dbg_lds && live) dump_lds<S, G>(a.dbg_lds, L, env, lane);
    euler_env<S, G>(L, ms, st, lane_s);
  }
#ifdef ODK_PROFILE
  const long long t_k2 = clock64();
#endif
  // same trick for the epilogue: its table addresses would otherwise be shared (CSE) with the prologue's and carried
  // across the substep loop in scratch
  size_t opaque1 = 0;
  asm volatile("" : "+s"(opaque1));
  const DevModel* mp = reinterpret_cast<const DevModel*>(reinterpret_cast<const char*>(m) + opaque1);
  // ... and for the RNG key / counter: everything derived from them (threefry key schedules of the observation-noise
  // draws) is recomputed here instead of riding through the loop in scratch
  const uint32_t k0e = (uint32_t)f2i(INFO[RL.KEY0]), k1e = (uint32_t)f2i(INFO[RL.KEY1]), ctre = (uint32_t)f2i(INFO[RL.CTR]);
  const int imi_e = f2i(INFO[RL.IMI]);
  const float phase_e[2] = {ACT[NU], ACT[NU + 1]};
  int step_e = f2i(INFO[RL.STEP]), push_step_e = f2i(INFO[RL.PSTEP]);
  float ep_steps_e = INFO[RL.EPSTEPS];
  const float prev_done_e = INFO[RL.DONE];
  for (int u = lane; u < NU; u += G) INFO[RL.MT + u] = CTRL[u];  // info["motor_targets"] (:422)
  draw_block<E::NDRAW>(k0e, k1e, ctre, L + E::O_NZ, lane);   // the motion-column buffers are dead after the last forward pass
  // reference motion of this step: evaluated in the prologue, parked here (reward and privileged obs are its only readers)
  if (lane < 40) L[E::O_REF + lane] = ref0;
  if (lane < 8) L[E::O_REF + 32 + lane] = ref1;
  ODK_SYNC();
  // ---- contacts, air time, swing peak (:424-435)
  float contact[2];
  foot_contact_flags(L + S::O_CDIST, contact);
  float air[2], peak[2];
  for (int f = 0; f < 2; f++) {
    air[f] = INFO[RL.AIR + f] + dt;
    peak[f] = fmaxf(INFO[RL.PEAK + f], L[S::O_SCR + S::S_MISC + 8 + f]);
  }
  ODK_SYNC();
  if (lane < 2) INFO[RL.AIR + lane] = air[lane];
  ODK_SYNC();
  // ---- termination (:483-485)
  float nanflag = 0;
  for (int i = lane; i < S::NQ + S::NV; i += G) nanflag += isnan(L[S::O_QPOS + i]) ? 1.0f : 0.0f;
  nanflag = gsum<G>(nanflag);
  const bool done_env = (L[S::O_SENS + mp->adr_upvector + 2] < 0.0f) || nanflag > 0;
  // ---- rewards (:622-669, :440-447); lanes 0..NU-1 hold the per-actuator terms
  float t_tq = 0, t_ar = 0, t_pose = 0, t_vel = 0, t_jp = 0, t_jv = 0;
  const float* REF = L + E::O_REF;
  if (lane < NU) {
    const int u = lane;
    const float jq = L[S::O_QPOS + mp->act_qposadr[u]], jv = L[S::O_QVEL + mp->act_dofadr[u]];
    const float af = L[S::O_ACTF + u];
    t_tq = af * af;
    const float da = ACT[u] - INFO[RL.LAST + u];
    t_ar = da * da;
    const bool leg = u < 5 || u >= 9;
    const bool counted = c.kind == 0 || leg;   // Standing: cost_stand_still(..., ignore_head=True) (standing.py:590-597)
    t_pose = counted ? fabsf(jq - mp->key_ctrl[u]) : 0.0f;
    t_vel = counted ? fabsf(jv) : 0.0f;
    if (c.kind != 0 && !leg) { const float dh = jq - INFO[RL.CMD + 3 + (u - 5)]; t_jp = dh * dh; }   // cost_head_pos (rewards.py:131-147)
    if (c.kind == 0 && leg) {  // joints[:5] ++ joints[9:] vs ref[:5] ++ ref[11:16]  (custom_rewards.py:80-88)
      const int ri = u < 5 ? u : u + 2;
      const float dp = jq - REF[ri], dv = jv - REF[16 + ri];
      t_jp = dp * dp; t_jv = dv * dv;
    }
  }
  t_tq = gsum<G>(t_tq); t_ar = gsum<G>(t_ar); t_pose = gsum<G>(t_pose); t_vel = gsum<G>(t_vel); t_jp = gsum<G>(t_jp); t_jv = gsum<G>(t_jv);
  float rew[7];
  {
    const float* cmd = INFO + RL.CMD;
    const float* lv = L + S::O_SENS + mp->adr_local_linvel;
    const float* gy = L + S::O_SENS + mp->adr_gyro;
    const float ex = (cmd[0] - lv[0]) * (cmd[0] - lv[0]);
    const float ey = fmaxf(fabsf(lv[1] - cmd[1]) - 0.1f, 0.0f);
    rew[0] = nan_to_num(expf(-(ex + ey * ey) / c.tracking_sigma));
    const float ea = (cmd[2] - gy[2]) * (cmd[2] - gy[2]);
    rew[1] = nan_to_num(expf(-ea / c.tracking_sigma));
    rew[2] = nan_to_num(t_tq);
    rew[3] = nan_to_num(t_ar);
    const float cn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
    if (c.kind != 0) {   // standing.py:585-606: cost_orientation(upvector), cost_head_pos (gated by the MOVE command norm)
      const float* up = L + S::O_SENS + mp->adr_upvector;
      rew[0] = nan_to_num(up[0] * up[0] + up[1] * up[1]);
      rew[1] = nan_to_num(t_jp) * (cn > 0.01f ? 1.0f : 0.0f);
    }
    rew[4] = nan_to_num(t_pose + t_vel) * (cn < 0.01f ? 1.0f : 0.0f);
    rew[5] = 1.0f;
    rew[6] = 0.0f;
    if (c.use_imitation) {  // custom_rewards.py:4-148
      const float* bv = L + S::O_QVEL;
      const float lin_xy = expf(-8.0f * ((bv[0] - REF[34]) * (bv[0] - REF[34]) + (bv[1] - REF[35]) * (bv[1] - REF[35])));
      const float lin_z = expf(-8.0f * (bv[2] - REF[36]) * (bv[2] - REF[36]));
      const float ang_xy = expf(-2.0f * ((bv[3] - REF[37]) * (bv[3] - REF[37]) + (bv[4] - REF[38]) * (bv[4] - REF[38]))) * 0.5f;
      const float ang_z = expf(-2.0f * (bv[5] - REF[39]) * (bv[5] - REF[39])) * 0.5f;
      float crew = 0;
      for (int f = 0; f < 2; f++) crew += (contact[f] == (REF[32 + f] > 0.5f ? 1.0f : 0.0f)) ? 1.0f : 0.0f;
      float r = lin_xy + lin_z + ang_xy + ang_z - t_jp * 15.0f - t_jv * 1.0e-3f + crew;
      r *= (cn > 0.01f) ? 1.0f : 0.0f;
      rew[6] = nan_to_num(r);
    }
  }
  float total = 0;
  for (int k = 0; k < 7; k++) { rew[k] *= c.reward_scales[k]; total += rew[k]; }
  const float reward = fminf(fmaxf(total * dt, 0.0f), 10000.0f);
  // ---- obs (uses the pre-shift last_act and the post-increment air time; :437)
  const float* NZ = L + E::O_NZ;   // this step's draws 4 .. 49 (drawn above, before the reward block)
  build_obs<S, G>(L, mp, c, contact, NZ, imi_e, phase_e, lane);
  // ---- info updates (:449-469)
  step_e += 1; push_step_e += 1;
  float la = 0, lla = 0;
  for (int u = lane; u < NU; u += G) { la = INFO[RL.LAST + u]; lla = INFO[RL.LAST2 + u]; }
  ODK_SYNC();
  for (int u = lane; u < NU; u += G) { INFO[RL.LAST3 + u] = lla; INFO[RL.LAST2 + u] = la; INFO[RL.LAST + u] = ACT[u]; }
  if (step_e > 500 && lane < 7) {   // sample_command (joystick.py:671-725) on draws 13 + 2 nu .. 20 + 2 nu of this step (the duck: 41 .. 48)
    constexpr int DC = draw_cmd(NU) - 4;
    const float z = NZ[DC + 7], u = NZ[DC + lane];
    INFO[RL.CMD + lane] = (z < 0.1f) ? 0.0f : c.cmd_range[lane][0] + u * (c.cmd_range[lane][1] - c.cmd_range[lane][0]);
  }
  if (done_env || step_e > 500) step_e = 0;
  int lcon_new = 0;
  for (int f = 0; f < 2; f++) {
    if (contact[f] != 0.0f) { air[f] = 0; peak[f] = 0; lcon_new |= (1 << f); }
  }
  (void)lcon;
  float metrics[ODK_NMETRIC];
  for (int k = 0; k < 7; k++) metrics[k] = c.reward_scales[k] > 0 ? rew[k] : -rew[k];
  metrics[7] = 0.5f * (peak[0] + peak[1]);
  // ---- EpisodeWrapper.step
  ep_steps_e += 1.0f;
  float done_f = done_env ? 1.0f : 0.0f, trunc = 0.0f;
  if (ep_steps_e >= (float)c.episode_length) { trunc = 1.0f - done_f; done_f = 1.0f; }
  const float keep = 1.0f - prev_done_e;  // info['episode_done'] of the previous step
  ODK_SYNC();
  if (lane == 0) {
    INFO[RL.AIR] = air[0]; INFO[RL.AIR + 1] = air[1]; INFO[RL.PEAK] = peak[0]; INFO[RL.PEAK + 1] = peak[1];
    INFO[RL.EPSTEPS] = ep_steps_e; INFO[RL.TRUNC] = trunc; INFO[RL.DONE] = done_f;
    INFO[RL.EPSUM] = (INFO[RL.EPSUM] + reward) * keep; INFO[RL.EPLEN] = (INFO[RL.EPLEN] + 1.0f) * keep;
    for (int k = 0; k < ODK_NMETRIC; k++) INFO[RL.EPMET + k] = (INFO[RL.EPMET + k] + metrics[k]) * keep;
    INFO[RL.CTR] = i2f((int)(ctre + 1)); INFO[RL.STEP] = i2f(step_e); INFO[RL.PSTEP] = i2f(push_step_e);
    INFO[RL.LCON] = i2f(lcon_new);
  }
  ODK_SYNC();
  // ---- AutoReset.step epilogue: data, obs <- first_* where done (info is NOT reset)
  if (done_f != 0.0f && c.autoreset) {
    const float* fs = a.first + (size_t)e * R::FSIZE;
    for (int i = lane; i < S::NQ + 2 * S::NV; i += G) L[S::O_QPOS + i] = fs[i];
    for (int k = lane; k < R::NPRIV; k += G) L[E::O_PRIV + k] = fs[R::FOBS + R::NOBS + k];
    // first_obs["state"] == first_priv[:101] by construction; every lane re-reads only what it wrote: no barrier
  }
  if (live) {
    // record store addresses recomputed here (opaque offset) instead of being shared with the prologue's loads and
    // carried across the substep loop: 69 64-bit per-lane pointers = 138 scratch dwords otherwise
    // (the env index is made opaque as well: the 64-bit record offset is then one multiply-add here instead of a value the
    // register allocator carries from the prologue -- in scratch, in the height-field kernel)
    int e_out = e;
    asm volatile("" : "+v"(e_out));
    float* rco = reinterpret_cast<float*>(reinterpret_cast<char*>(a.recs + (size_t)e_out * R::SIZE) + opaque1);
    for (int i = lane; i < S::NQ + 2 * S::NV; i += G) rco[i] = L[S::O_QPOS + i];
    for (int k = lane; k < RL.NINFO; k += G) rco[R::INFO + k] = INFO[k];
    write_outputs<S, G>(a, L, env, reward, done_f, trunc, metrics, lane);
  }
#ifdef ODK_PROFILE
  if (a.dbg_lds && live && lane == 0) a.dbg_lds[(size_t)env * S::TOTAL + S::O_SCR + S::S_PROF + 19] = (float)(clock64() - t_k2);
#endif
}

// mjx_env.step alone: ctrl = action buffer, no env logic (parity tests)
template <class S, int G, int HF>
__global__ void __launch_bounds__(64) physics_kernel(KArgs a) {
  extern __shared__ float lds[];
  using E = EnvL<S>; using R = Rec<S>;
  const int slot = threadIdx.x / G, lane = threadIdx.x % G;
  const int env = blockIdx.x * (64 / G) + slot;
  const bool live = env < a.nenv;
  const int e = live ? env : a.nenv - 1;
  float* L = lds + slot * E::TOTAL;
  const int* RT = load_shared<S>(lds, 64 / G, a.m);   // ordered before its first use by the ODK_SYNCs below
#ifdef ODK_POISON_LDS   // debug build: every read of LDS that was not written by this launch surfaces as NaN
  for (int k = lane; k < E::TOTAL; k += G) L[k] = __int_as_float(0x7fc00000);
  ODK_SYNC();
#endif
  float* rc = a.recs + (size_t)e * R::SIZE;
  for (int i = lane; i < S::NQ + 2 * S::NV; i += G) L[S::O_QPOS + i] = rc[i];
  for (int u = lane; u < S::NU; u += G) L[S::O_CTRL + u] = a.action[(size_t)e * S::NU + u];
  load_params<S, G>(L, a.m, a.dr ? a.dr + (size_t)e * DRL<S>::SIZE : nullptr, lane);
#ifdef ODK_PROFILE
  for (int k = lane; k < 36; k += G) L[S::O_SCR + S::S_PROF + k] = 0;
#endif
  ODK_SYNC();
  Statics<S, G> st;
  load_statics<S, G>(st, a.m, lane);
  for (int s = 0; s < a.n_substeps; s++) {
    const bool last = s == a.n_substeps - 1;
    forward_env<S, G, HF>(L, RT, a.m, a.hfield, st, lane, last ? 1 : 0);
    if (last && a.dbg_lds && live) dump_lds<S, G>(a.dbg_lds, L, env, lane);
    euler_env<S, G>(L, a.m, st, lane);
  }
  if (live) for (int i = lane; i < S::NQ + 2 * S::NV; i += G) rc[i] = L[S::O_QPOS + i];
}

// ================================================================================================
// host side
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int odk_fail_(int code, const char* msg) { return fail(code, "%s", msg); }   // for odk_learner.hip
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(ODK_ERR_HIP, "%s: %s", #x, hipGetErrorString(_e)); } while (0)

struct odk_model { DevModel h; int shape; std::vector<float> hfield; };  // shape: 0 = A, 1 = B, 2 = C (physics only); hfield: [nrow][ncol] in [0, 1]

struct odk_batch {
  odk_model model;
  int nenv, device, G;
  odk_env_config cfg;
  DevModel* d_model = nullptr; DevPRM h_prm; float* d_table = nullptr;
  float* d_recs = nullptr; float* d_first = nullptr; float* d_dr = nullptr; float* d_dbg = nullptr; float* d_hfield = nullptr;
  std::vector<float> h_dr; bool dr_enabled = false;
  int rec_size, frec_size, lds_total, dr_size, env_lds;
  static constexpr size_t ODK_TIMING_EVENT_PAIRS = 1024;
  int timing = 0; size_t timing_count = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> events; size_t ev_used = 0;   // timing: 0 off, n: every n-th launch
};

extern "C" const char* odk_last_error(void) { return g_err.c_str(); }

extern "C" void odk_default_config(odk_env_config* c) {
  memset(c, 0, sizeof(*c));
  c->ctrl_dt = 0.02f; c->action_scale = 0.25f; c->dof_vel_scale = 0.05f; c->max_motor_velocity = 5.24f;
  c->noise_level = 1.0f; c->noise_gyro = 0.1f; c->noise_accelerometer = 0.05f; c->noise_gravity = 0.1f; c->noise_joint_vel = 2.5f;
  const float s10[10] = {0.03f, 0.03f, 0.03f, 0.05f, 0.08f, 0.03f, 0.03f, 0.03f, 0.05f, 0.08f};  // BUG-COMPAT joystick.py:184-200
  for (int i = 0; i < 10; i++) c->qpos_noise_scale[i] = s10[i];
  const float rs[7] = {2.5f, 6.0f, -1.0e-3f, -0.5f, -0.2f, 20.0f, 1.0f};
  memcpy(c->reward_scales, rs, sizeof(rs));
  c->tracking_sigma = 0.01f;
  c->push_enable = 1.0f; c->push_interval_range[0] = 5.0f; c->push_interval_range[1] = 10.0f;
  c->push_magnitude_range[0] = 0.1f; c->push_magnitude_range[1] = 1.0f;
  const float cr[7][2] = {{-0.15f, 0.15f}, {-0.2f, 0.2f}, {-1.0f, 1.0f}, {-0.34f, 1.1f}, {-0.78f, 0.78f}, {-1.5f, 1.5f}, {-0.5f, 0.5f}};
  memcpy(c->cmd_range, cr, sizeof(cr));
  c->use_imitation = 1; c->use_motor_speed_limits = 1; c->autoreset = 1; c->episode_length = 1000; c->n_substeps = 10; c->lanes_per_env = 0;
  c->env_kind = ODK_ENV_JOYSTICK; c->reset_base_qvel = 0.05f; c->hfield_up_normals_only = 0;
}
extern "C" void odk_default_config_standing(odk_env_config* c) {   // reference standing.py:44-100
  odk_default_config(c);
  c->env_kind = ODK_ENV_STANDING; c->reset_base_qvel = 0.5f; c->hfield_up_normals_only = 0;
  c->max_motor_velocity = 0.0f;   // standing.py has no speed limit (and no such config key)
  c->noise_gyro = 0.05f; c->noise_accelerometer = 0.005f;
  const float rs[7] = {-0.5f, -2.0f, -1.0e-3f, -0.375f, -0.3f, 20.0f, 0.0f};   // orientation, head_pos, torques, action_rate, stand_still, alive
  memcpy(c->reward_scales, rs, sizeof(rs));
  for (int k = 0; k < 3; k++) c->cmd_range[k][0] = c->cmd_range[k][1] = 0.0f;   // standing.py:652-654
  c->cmd_range[5][0] = -2.7f; c->cmd_range[5][1] = 2.7f;
  c->use_imitation = 0; c->use_motor_speed_limits = 0;
}
extern "C" void odk_obs_sizes(int env_kind, int* nobs, int* npriv) {
  if (nobs) *nobs = env_kind == ODK_ENV_STANDING ? ODK_NOBS_STANDING : ODK_NOBS;
  if (npriv) *npriv = env_kind == ODK_ENV_STANDING ? ODK_NPRIV_STANDING : ODK_NPRIV;
}
static void obs_sizes_nu(int nu, int env_kind, int* nobs, int* npriv) {
  if (nobs) *nobs = obs_nobs(nu, env_kind == ODK_ENV_STANDING);
  if (npriv) *npriv = obs_npriv(nu, env_kind == ODK_ENV_STANDING);
}

// ---- blob parsing
struct RecHdr { char name[32]; uint32_t dtype, ndim, shape[4]; uint64_t nbytes; };
static const unsigned char* find_rec(const unsigned char* b, uint64_t len, const char* name, RecHdr* h) {
  uint32_t n;
  memcpy(&n, b + 8, 4);
  uint64_t off = 16;
  for (uint32_t i = 0; i < n && off + 64 <= len; i++) {
    memcpy(h, b + off, 64);
    off += 64;
    if (strncmp(h->name, name, 32) == 0) return b + off;
    off += h->nbytes + ((8 - (h->nbytes & 7)) & 7);
  }
  return nullptr;
}
struct Blob {
  const unsigned char* b; uint64_t len; bool ok = true; std::string missing;
  int F(const char* name, float* dst, int maxc) {
    RecHdr h; const unsigned char* p = find_rec(b, len, name, &h);
    if (!p || h.dtype != 0) { ok = false; missing = name; return -1; }
    int cnt = (int)(h.nbytes / 8);
    if (cnt > maxc) { ok = false; missing = std::string(name) + " (too large)"; return -1; }
    for (int i = 0; i < cnt; i++) { double v; memcpy(&v, p + 8 * i, 8); dst[i] = (float)v; }
    return cnt;
  }
  int D(const char* name, double* dst, int maxc) {
    RecHdr h; const unsigned char* p = find_rec(b, len, name, &h);
    if (!p || h.dtype != 0) { ok = false; missing = name; return -1; }
    int cnt = (int)(h.nbytes / 8);
    if (cnt > maxc) { ok = false; missing = std::string(name) + " (too large)"; return -1; }
    memcpy(dst, p, 8 * (size_t)cnt);
    return cnt;
  }
  int I(const char* name, int* dst, int maxc) {
    RecHdr h; const unsigned char* p = find_rec(b, len, name, &h);
    if (!p || h.dtype != 1) { ok = false; missing = name; return -1; }
    int cnt = (int)(h.nbytes / 4);
    if (cnt > maxc) { ok = false; missing = std::string(name) + " (too large)"; return -1; }
    memcpy(dst, p, 4 * (size_t)cnt);
    return cnt;
  }
  // 2D int table [rows][srccols] -> dst[rows][dstcols]
  void I2(const char* name, int* dst, int rows_max, int dstcols) {
    RecHdr h; const unsigned char* p = find_rec(b, len, name, &h);
    if (!p || h.dtype != 1 || h.ndim != 2) { ok = false; missing = name; return; }
    int rows = (int)h.shape[0], cols = (int)h.shape[1];
    if (rows > rows_max || cols > dstcols) { ok = false; missing = std::string(name) + " (shape)"; return; }
    for (int r = 0; r < rows; r++)
      for (int c2 = 0; c2 < cols; c2++) memcpy(&dst[r * dstcols + c2], p + 4 * ((size_t)r * cols + c2), 4);
  }
};

static void quat2mat(const double* q, double* m) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void make_frame_h(const double* n, float* frame) {
  double a[3] = {n[0], n[1], n[2]}, b[3] = {0, 0, 0}, c[3];
  double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  for (int k = 0; k < 3; k++) a[k] /= na;
  if (fabs(a[1]) < 0.5) b[1] = 1; else b[2] = 1;
  double dt = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  for (int k = 0; k < 3; k++) b[k] -= a[k] * dt;
  double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  for (int k = 0; k < 3; k++) b[k] /= nb;
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
  for (int k = 0; k < 3; k++) { frame[k] = (float)a[k]; frame[3 + k] = (float)b[k]; frame[6 + k] = (float)c[k]; }
}
// constant impedance of a row at pos = 0 (friction loss): returns R, b
static void row_consts(const double* solref, const double* solimp, double dt, double invweight, double* R, double* bb) {
  double timeconst = fmax(solref[0], 2 * dt), dmin = fmin(fmax(solimp[0], 0.0001), 0.9999), dmax = fmin(fmax(solimp[1], 0.0001), 0.9999);
  double b = 2.0 / (dmax * timeconst);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  double imp = dmin;  // imp_x = 0 -> imp_y = 0
  *R = fmax(invweight * (1 - imp) / imp, 1e-15);
  *bb = b;
}

// mju impedance constants of one constraint row (the same float arithmetic the kernels used to repeat per row and substep)
static void pack_imp(const float* solref, const float* solimp, float dt, float* P) {
  const float timeconst = fmaxf(solref[0], 2.0f * dt), dampratio = solref[1];
  const float dmin = fminf(fmaxf(solimp[0], 0.0001f), 0.9999f), dmax = fminf(fmaxf(solimp[1], 0.0001f), 0.9999f);
  const float width = fmaxf(solimp[2], 1e-15f), mid = fminf(fmaxf(solimp[3], 0.0001f), 0.9999f), power = fmaxf(solimp[4], 1.0f);
  float k = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  float b = 2.0f / (dmax * timeconst);
  if (solref[0] <= 0) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  P[0] = k; P[1] = b; P[2] = dmin; P[3] = dmax; P[4] = 1.0f / width; P[5] = mid; P[6] = power;
  P[7] = 1.0f / powf(mid, power - 1.0f); P[8] = 1.0f / powf(1.0f - mid, power - 1.0f);
}

// Twin-dof detection and the reduced (twins merged) tree layouts -- see DevModel::paired.  Called after the dof / joint /
// foot tables are in place.  Returns false when the reduced tree is not "floating base + up to three serial chains of <= 5
// dofs" (the form chain_solve is built for).
namespace {
struct SparseLayout { int depth[MAXV], adr[MAXV], ancmask[MAXV], descmask[MAXV], anc_at[MAXV][MAXV], nnz; };
// rows in dof order; row i = entries for i's ancestors by depth (c = depth[i]: the diagonal) -- tables.py _sparse_layout
void sparse_layout(const int* parent, int n, SparseLayout& L) {
  L.nnz = 0;
  for (int i = 0; i < n; i++) {
    L.depth[i] = parent[i] < 0 ? 0 : L.depth[parent[i]] + 1;
    L.adr[i] = L.nnz; L.nnz += L.depth[i] + 1;
    L.ancmask[i] = 0; L.descmask[i] = 0;
  }
  for (int i = 0; i < n; i++) {
    int a = i;
    for (int c = L.depth[i]; c >= 0; c--, a = parent[a]) {
      L.anc_at[i][c] = a;
      if (a != i) { L.ancmask[i] |= 1 << a; L.descmask[a] |= 1 << i; }
    }
  }
}
}  // namespace
// DevModel::body_st: what forward_env's sweeps need of each body, flattened (index MAXB: the record of a lane without a body)
static void fill_body_st(DevModel& m) {
  auto fill = [&](BodySt& b, int bi, bool in) {
    memset(&b, 0, sizeof(b));
    b.level = in ? m.body_level[bi] : -2;
    b.parent = m.body_parent[bi];
    b.nchild = in ? m.body_nchild[bi] : 0;
    b.pathmask = in ? m.body_pathmask[bi] : 0;
    b.is_path = in ? m.body_is_path[bi] : 0;
    b.upmask = in ? m.body_upmask[bi] : 0;
    b.path_head = in ? m.body_path_head[bi] : 0;
    for (int k = 0; k < 3; k++) b.child[k] = m.body_children[bi][k];
    b.njnt = (in && b.level > 0) ? m.body_jntnum[bi] : 0;
    for (int k = 0; k < 2; k++) {
      const bool on = k < b.njnt;
      const int j = on ? m.body_jntadr[bi] + k : 0;
      b.jj[k] = j;
      b.jd[k] = m.jnt_dofadr[j];
      b.jr[k] = (m.paired && m.dof_tkind[b.jd[k]] == 2) ? -1 : m.dof_red[b.jd[k]];
      for (int c = 0; c < 3; c++) b.ax[k][c] = on ? m.jnt_axis[j][c] : 0.0f;
    }
    for (int c = 0; c < 3; c++) { b.pos[c] = m.body_pos[bi][c]; b.ipos[c] = m.body_ipos[bi][c]; }
    for (int c = 0; c < 4; c++) b.quat[c] = m.body_quat[bi][c];
    for (int c = 0; c < 6; c++) b.inertia[c] = m.body_inertia[bi][c];
  };
  for (int bi = 0; bi < m.nb; bi++) fill(m.body_st[bi], bi, true);
  fill(m.body_st[MAXB], 0, false);
}

static bool build_reduced_tables(DevModel& m) {
  m.paired = 0; m.nrchain = 0;
  int ntwin = 0;
  for (int d = 0; d < MAXV; d++) { m.dof_tkind[d] = 0; m.dof_red[d] = 0; m.red_main[d] = 0; m.red_twin[d] = -1; }
  for (int v = 7; v < m.nv; v++) {
    const int u = v - 1, ju = m.dof_jnt[u], jv = m.dof_jnt[v];
    if (ju < 0 || jv < 0 || m.dof_tkind[u] != 0) continue;
    const bool same = m.dof_body[u] == m.dof_body[v] && m.dof_anc[v][1] == u && m.jnt_axis[ju][0] == m.jnt_axis[jv][0] &&
                      m.jnt_axis[ju][1] == m.jnt_axis[jv][1] && m.jnt_axis[ju][2] == m.jnt_axis[jv][2];   // jnt_pos == 0 for every hinge (checked by the caller)
    if (!same) continue;
    m.dof_tkind[u] = 1; m.dof_tkind[v] = 2; ntwin++;
  }
  m.paired = ntwin > 0;
  int nr = 0;
  for (int d = 0; d < m.nv; d++) {
    if (m.dof_tkind[d] == 2) { m.dof_red[d] = m.dof_red[d - 1]; continue; }
    m.dof_red[d] = nr; m.red_main[nr] = d; m.red_twin[nr] = m.dof_tkind[d] == 1 ? d + 1 : -1; nr++;
  }
  m.nvr = nr;
  int rparent[MAXV], rvparent[MAXV];
  for (int r = 0; r < nr; r++) {
    const int p = m.dof_anc[m.red_main[r]][1];   // -1 at the root
    rparent[r] = p < 0 ? -1 : m.dof_red[p];
    rvparent[r] = rparent[r];
    const int u = m.red_main[r];
    m.red_foot[r] = m.foot_dofmask[0][u] | (m.foot_dofmask[1][u] << 1);
  }
  // virtual tree (tables.py): the second leg hangs below the first foot's last dof
  {
    int l_last = -1, r_first = -1;
    for (int r = 0; r < nr; r++) {
      if (m.red_foot[r] & 1) l_last = r;
      if ((m.red_foot[r] & 2) && !(m.red_foot[r] & 1) && r_first < 0) r_first = r;
    }
    if (l_last >= 0 && r_first > l_last) rvparent[r_first] = l_last;
  }
  static SparseLayout T, V;   // model loading is not re-entrant anyway (thread-local error string aside)
  sparse_layout(rparent, nr, T);
  sparse_layout(rvparent, nr, V);
  if (T.nnz > MAXNZ || V.nnz > MAXNZ) return false;
  m.nMr = T.nnz; m.nHr = V.nnz;
  for (int r = 0; r < nr; r++) {
    m.red_depth[r] = T.depth[r]; m.red_Madr[r] = T.adr[r]; m.red_ancmask[r] = T.ancmask[r]; m.red_descmask[r] = T.descmask[r];
    m.rv_depth[r] = V.depth[r]; m.rv_Madr[r] = V.adr[r]; m.rv_ancmask[r] = V.ancmask[r]; m.rv_descmask[r] = V.descmask[r];
  }
  auto pack = [&](int r, int a) {
    const bool diag = a == r, pair = diag && m.red_twin[r] >= 0;
    return r | (a << 5) | (m.red_foot[r] << 10) | (m.red_foot[a] << 12) | ((int)diag << 14) | ((int)pair << 15) | (m.red_main[r] << 16);
  };
  for (int r = 0; r < nr; r++) {
    for (int c = 0; c <= T.depth[r]; c++) m.R_ent[T.adr[r] + c] = pack(r, T.anc_at[r][c]);
    for (int c = 0; c <= V.depth[r]; c++) {
      const int a = V.anc_at[r][c];
      int src = -1;   // address of (r, a) in the true reduced layout, if a is a true ancestor (or r itself)
      if (a == r || ((T.ancmask[r] >> a) & 1)) src = T.adr[r] + T.depth[a];
      m.RH_ent[V.adr[r] + c] = pack(r, a) | ((src + 1) << 21);
    }
  }
  // reduced chains below the floating base
  int d = 6;
  bool ok = nr > 6;
  for (int r = 0; r < 6 && ok; r++) ok = rparent[r] == r - 1;
  while (ok && d < nr) {
    if (rparent[d] != 5 || m.nrchain == 3) { ok = false; break; }
    int e = d;
    while (e + 1 < nr && rparent[e + 1] == e) e++;
    m.rchain_first[m.nrchain] = d; m.rchain_len[m.nrchain] = e - d + 1; m.nrchain++;
    if (e - d + 1 > 6) ok = false;      // (the chosen shape's own chain length is checked once the shape is known)
    d = e + 1;
  }
  // the reduced dofs above a foot must be exactly the six base dofs + one whole chain (foot_twist in odk_kernels.h)
  for (int f = 0; f < 2 && ok; f++) {
    int c = -1;
    for (int k = 0; k < m.nrchain; k++) if ((m.red_foot[m.rchain_first[k]] >> f) & 1) c = k;
    ok = c >= 0;
    for (int r = 0; r < nr && ok; r++) {
      const bool want = r < 6 || (r >= m.rchain_first[c] && r < m.rchain_first[c] + m.rchain_len[c]);
      ok = (((m.red_foot[r] >> f) & 1) != 0) == want;
    }
    if (ok) { m.foot_rchain_first[f] = m.rchain_first[c]; m.foot_rchain_len[f] = m.rchain_len[c]; }
  }
  if (ok && !m.paired) {   // a model without twins reduces to itself: the tables above must be the blob's own (tables.py)
    ok = m.nMr == m.nM && m.nHr == m.nH;
    for (int r = 0; r < nr && ok; r++)
      ok = m.red_depth[r] == m.dof_depth[r] && m.red_Madr[r] == m.dof_Madr[r] && m.red_ancmask[r] == m.dof_ancmask[r] && m.red_descmask[r] == m.dof_descmask[r] &&
           m.rv_depth[r] == m.vdof_depth[r] && m.rv_Madr[r] == m.vdof_Madr[r] && m.rv_ancmask[r] == m.vdof_ancmask[r] && m.rv_descmask[r] == m.vdof_descmask[r];
    for (int p = 0; p < m.nM && ok; p++) ok = (m.R_ent[p] & 31) == m.M_i[p] && ((m.R_ent[p] >> 5) & 31) == m.M_j[p];
    for (int p = 0; p < m.nH && ok; p++) ok = (m.RH_ent[p] & 31) == m.H_i[p] && ((m.RH_ent[p] >> 5) & 31) == m.H_j[p] && (m.RH_ent[p] >> 21) - 1 == m.H_src[p];
  }
  return ok;
}

// Face polygons and unique edges of a convex hull given as outward triangles (what mjx mesh.py prepares for collision_convex:
// coplanar facets merged, edges with their two faces).  Triangles that share an edge and a plane (normals within 1e-6) become one
// polygon (at most a quad here: larger merges are refused).  Order matters downstream (first-index tie-breaks): faces in order of
// their first triangle, polygons start at the first boundary edge, edges in face order with va < vb.
static bool build_convex_tables(const double (*v)[3], int nv, const int (*tri)[3], int nt, int* npoly, int (*poly)[5], float (*fnorm)[3], int* nedge,
                                int (*edge)[4], float* centroid, int maxf, int maxe) {
  std::vector<std::array<double, 3>> tn(nt);
  std::vector<int> grp(nt);
  double c[3] = {0, 0, 0};
  for (int i = 0; i < nv; i++) for (int k = 0; k < 3; k++) c[k] += v[i][k] / nv;
  for (int k = 0; k < 3; k++) centroid[k] = (float)c[k];
  for (int t = 0; t < nt; t++) {
    double e1[3], e2[3], n[3];
    for (int k = 0; k < 3; k++) { e1[k] = v[tri[t][1]][k] - v[tri[t][0]][k]; e2[k] = v[tri[t][2]][k] - v[tri[t][0]][k]; }
    n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    const double l = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (l == 0) return false;
    tn[t] = {n[0] / l, n[1] / l, n[2] / l};
    grp[t] = t;
  }
  for (int it = 0; it < nt; it++)
    for (int a = 0; a < nt; a++)
      for (int b = a + 1; b < nt; b++) {
        if (grp[a] == grp[b]) continue;
        int shared = 0;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (tri[a][i] == tri[b][j]) shared++;
        if (shared == 2 && fabs(tn[a][0] - tn[b][0]) < 1e-6 && fabs(tn[a][1] - tn[b][1]) < 1e-6 && fabs(tn[a][2] - tn[b][2]) < 1e-6) {
          const int ga = grp[a], gb = grp[b], lo = ga < gb ? ga : gb;
          for (int t = 0; t < nt; t++) if (grp[t] == ga || grp[t] == gb) grp[t] = lo;
        }
      }
  int nf = 0;
  for (int g = 0; g < nt; g++) {
    std::vector<int> ea, eb;
    bool any = false;
    for (int t = 0; t < nt; t++) {
      if (grp[t] != g) continue;
      any = true;
      for (int i = 0; i < 3; i++) {
        const int a = tri[t][i], b = tri[t][(i + 1) % 3];
        bool inner = false;
        for (int u = 0; u < nt && !inner; u++) {
          if (grp[u] != g || u == t) continue;
          for (int j = 0; j < 3; j++) if (tri[u][j] == b && tri[u][(j + 1) % 3] == a) inner = true;
        }
        if (!inner) { ea.push_back(a); eb.push_back(b); }
      }
    }
    if (!any) continue;
    if (nf >= maxf || ea.size() > 4) return false;
    int cur = ea[0], cnt = 0;
    for (size_t step = 0; step < ea.size(); step++) {
      poly[nf][1 + cnt++] = cur;
      int nxt = -1;
      for (size_t k = 0; k < ea.size(); k++) if (ea[k] == cur) { nxt = eb[k]; break; }
      cur = nxt;
      if (cur == ea[0] || cur < 0) break;
    }
    poly[nf][0] = cnt;
    for (int k = cnt; k < 4; k++) poly[nf][1 + k] = poly[nf][1];
    for (int k = 0; k < 3; k++) fnorm[nf][k] = (float)tn[g][k];
    nf++;
  }
  int ne = 0;
  for (int f = 0; f < nf; f++)
    for (int i = 0; i < poly[f][0]; i++) {
      const int a = poly[f][1 + i], b = poly[f][1 + (i + 1) % poly[f][0]];
      if (a > b) continue;
      if (ne >= maxe) return false;
      edge[ne][0] = a; edge[ne][1] = b; edge[ne][2] = f; edge[ne][3] = -1; ne++;
    }
  for (int f = 0; f < nf; f++)
    for (int i = 0; i < poly[f][0]; i++) {
      const int a = poly[f][1 + i], b = poly[f][1 + (i + 1) % poly[f][0]];
      if (a < b) continue;
      bool found = false;
      for (int k = 0; k < ne; k++) if (edge[k][0] == b && edge[k][1] == a) { edge[k][3] = f; found = true; }
      if (!found) return false;   // open surface
    }
  for (int k = 0; k < ne; k++) if (edge[k][2] < 0 || edge[k][3] < 0) return false;
  *npoly = nf; *nedge = ne;
  return true;
}

extern "C" int odk_model_load(const void* blob, uint64_t len, odk_model** out) {
  if (!blob || !out || len < 16 || memcmp(blob, "ODKM", 4) != 0) return fail(ODK_ERR_INVALID, "odk_model_load: not an ODKM blob");
  Blob B{(const unsigned char*)blob, len};
  odk_model* mo = new odk_model();
  DevModel& m = mo->h;
  memset(&m, 0, sizeof(m));
  int one[1];
  B.I("nq", one, 1); m.nq = one[0]; B.I("nv", one, 1); m.nv = one[0]; B.I("nu", one, 1); m.nu = one[0];
  B.I("nbody", one, 1); m.nb = one[0]; B.I("njnt", one, 1); m.nj = one[0]; B.I("nsite", one, 1); m.nsite = one[0];
  if (!B.ok) { delete mo; return fail(ODK_ERR_INVALID, "odk_model_load: missing %s", B.missing.c_str()); }
  if (m.nq > MAXQ || m.nv > MAXV || m.nu > MAXU || m.nb > MAXB || m.nj > MAXJ || m.nsite > MAXSITE) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "model too large"); }
  {   // <option cone="elliptic">: zones, cone Hessian and exact line search (odk_kernels.h "elliptic cones") are accepted for EVERY compiled shape
      // with hull feet, at 32 lanes per env: the third and fourth shapes carry the code as a runtime switch (Shape::ELL), the duck's two shapes have
      // instantiations of their own with it (ShapeAE / ShapeBE; launch()).  The only refusal is sphere / capsule feet (below, by name).
    RecHdr ch;
    if (find_rec((const unsigned char*)blob, len, "opt_cone", &ch)) {
      int cone[1] = {0};
      Blob Cn{(const unsigned char*)blob, len};
      Cn.I("opt_cone", cone, 1);
      m.cone = cone[0] != 0;
    }
  }
  // <equality> (mjcf.py compiles joint / connect / weld; the float64 oracle builds all their rows).  The kernels model <equality><joint>
  // rows between two hinges of one serial chain (odk_kernels.h "equality rows": shapes with S::EQ, at most EQ_MAX rows, a dof in at most
  // one); every other ACTIVE equality is refused by name instead of being stepped without it.  Collected here, finished below once the
  // reduced layout and the shape are known.
  int eq_n = 0, eq_type[16], eq_active[16], eq_o1[16], eq_o2[16];
  double eq_data[16 * 11], eq_solref[16 * 2], eq_solimp[16 * 5];
  {
    RecHdr eh;
    if (find_rec((const unsigned char*)blob, len, "eq_type", &eh) && eh.nbytes > 0) {
      Blob E{(const unsigned char*)blob, len};
      eq_n = E.I("eq_type", eq_type, 16);
      if (eq_n < 0 || E.I("eq_active", eq_active, 16) != eq_n || E.I("eq_obj1id", eq_o1, 16) != eq_n || E.I("eq_obj2id", eq_o2, 16) != eq_n ||
          E.D("eq_data", eq_data, 16 * 11) != 11 * eq_n || E.D("eq_solref", eq_solref, 32) != 2 * eq_n || E.D("eq_solimp", eq_solimp, 80) != 5 * eq_n) {
        delete mo; return fail(ODK_ERR_UNSUPPORTED, "equality constraints: more than 16, or eq_* records incomplete");
      }
      for (int k = 0; k < eq_n; k++)
        if (eq_active[k] && (eq_type[k] < 0 || eq_type[k] > 2)) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "<equality> constraint %d: type %d (connect, weld and joint are modelled)", k, eq_type[k]); }
    }
  }
  double dtv[1], g3[3], t1[1];
  B.D("opt_timestep", dtv, 1); m.dt = (float)dtv[0];
  B.D("opt_gravity", g3, 3); for (int k = 0; k < 3; k++) m.gravity[k] = (float)g3[k];
  B.D("opt_tolerance", t1, 1); m.tolerance = (float)t1[0]; B.D("opt_ls_tolerance", t1, 1); m.ls_tolerance = (float)t1[0];
  B.D("opt_impratio", t1, 1); m.impratio = (float)t1[0]; B.D("stat_meaninertia", t1, 1); m.meaninertia = (float)t1[0];
  B.I("opt_iterations", &m.iterations, 1); B.I("opt_ls_iterations", &m.ls_iterations, 1);
  int eulerdamp = 0; B.I("opt_eulerdamp", &eulerdamp, 1);
  // bodies
  B.I("k_base_body", &m.base_body, 1); B.I("k_body_in_tree", m.body_in_tree, MAXB); B.I("body_parentid", m.body_parent, MAXB);
  B.I("body_jntadr", m.body_jntadr, MAXB); B.I("body_jntnum", m.body_jntnum, MAXB);
  B.I2("k_body_chain", &m.body_chain[0][0], MAXB, MAXCHAIN); B.I("k_body_chain_len", m.body_chain_len, MAXB);
  B.I2("k_body_ancdof", &m.body_ancdof[0][0], MAXB, MAXV); B.I("k_body_nancdof", m.body_nancdof, MAXB);
  B.I2("k_body_sub", &m.body_sub[0][0], MAXB, MAXB); B.I("k_body_nsub", m.body_nsub, MAXB);
  B.I("k_max_level", &m.max_level, 1); B.I("k_body_level", m.body_level, MAXB); B.I2("k_body_children", &m.body_children[0][0], MAXB, 3);
  B.I("k_body_nchild", m.body_nchild, MAXB);
  B.I("k_max_nonpath_level", &m.max_nonpath_level, 1); B.I("k_body_pathmask", m.body_pathmask, MAXB); B.I("k_body_is_path", m.body_is_path, MAXB);
  B.I("k_body_upmask", m.body_upmask, MAXB); B.I("k_body_path_head", m.body_path_head, MAXB);
  B.F("body_pos", &m.body_pos[0][0], MAXB * 3); B.F("body_quat", &m.body_quat[0][0], MAXB * 4); B.F("body_ipos", &m.body_ipos[0][0], MAXB * 3);
  B.F("body_mass", m.body_mass, MAXB); B.F("body_inertia_full", &m.body_inertia[0][0], MAXB * 6);
  // joints
  B.I("jnt_qposadr", m.jnt_qposadr, MAXJ); B.I("jnt_dofadr", m.jnt_dofadr, MAXJ); B.I("jnt_bodyid", m.jnt_bodyid, MAXJ);
  B.F("jnt_axis", &m.jnt_axis[0][0], MAXJ * 3); B.F("jnt_pos", &m.jnt_pos[0][0], MAXJ * 3); B.F("jnt_range", &m.jnt_range[0][0], MAXJ * 2);
  B.F("qpos0", m.qpos0, MAXQ); B.F("key_qpos", m.key_qpos, MAXQ); B.F("key_ctrl", m.key_ctrl, MAXU);
  // dofs
  B.I("dof_bodyid", m.dof_body, MAXV); B.I("k_dof_depth", m.dof_depth, MAXV); B.I2("k_dof_anc", &m.dof_anc[0][0], MAXV, MAXV);
  B.I("k_dof_Madr", m.dof_Madr, MAXV); B.I2("k_dof_anc_adr", &m.dof_anc_adr[0][0], MAXV, MAXV);
  B.I("k_dof_ndesc", m.dof_ndesc, MAXV); B.I2("k_dof_desc", &m.dof_desc[0][0], MAXV, MAXV); B.I2("k_dof_desc_adr", &m.dof_desc_adr[0][0], MAXV, MAXV);
  B.I("k_dof_nprefix", m.dof_nprefix, MAXV); B.I2("k_dof_prefix", &m.dof_prefix[0][0], MAXV, MAXV);
  B.I("k_dof_nsym", m.dof_nsym, MAXV); B.I2("k_dof_sym_dof", &m.dof_sym_dof[0][0], MAXV, MAXV); B.I2("k_dof_sym_adr", &m.dof_sym_adr[0][0], MAXV, MAXV);
  B.I("k_dof_act", m.dof_act, MAXV); B.I("k_dof_flrow", m.dof_flrow, MAXV); B.I("k_dof_limrow", m.dof_limrow, MAXV);
  B.I("k_dof_ancmask", m.dof_ancmask, MAXV); B.I("k_dof_descmask", m.dof_descmask, MAXV);
  B.I("k_vdof_ancmask", m.vdof_ancmask, MAXV); B.I("k_vdof_descmask", m.vdof_descmask, MAXV);
  B.F("dof_armature", m.dof_armature, MAXV); B.F("dof_damping", m.dof_damping, MAXV); B.F("dof_frictionloss", m.dof_frictionloss, MAXV);
  B.F("dof_invweight0", m.dof_invweight0, MAXV);
  B.I("k_nM", &m.nM, 1); B.I("k_M_i", m.M_i, MAXNZ); B.I("k_M_j", m.M_j, MAXNZ);
  B.I("k_nchain", &m.nchain, 1); B.I("k_chain_first", m.chain_first, 3); B.I("k_chain_len", m.chain_len, 3);
  for (int d = 0; d < MAXV; d++) { m.dof_qadr[d] = -1; m.dof_jnt[d] = -1; }
  for (int j = 1; j < m.nj; j++) {
    const int d = m.jnt_dofadr[j];
    m.dof_qadr[d] = m.jnt_qposadr[j]; m.dof_jnt[d] = j;
    m.dof_range[d][0] = m.jnt_range[j][0]; m.dof_range[d][1] = m.jnt_range[j][1];
    if (m.jnt_pos[j][0] != 0.0f || m.jnt_pos[j][1] != 0.0f || m.jnt_pos[j][2] != 0.0f) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "hinge joints must sit at their body origin (jnt_pos == 0)"); }
  }
  for (int b2 = 0; b2 < m.nb; b2++) if (m.body_jntnum[b2] > 2) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "more than two joints on one body"); }
  B.I("k_vdof_depth", m.vdof_depth, MAXV); B.I2("k_vdof_anc", &m.vdof_anc[0][0], MAXV, MAXV); B.I("k_vdof_Madr", m.vdof_Madr, MAXV);
  B.I2("k_vdof_anc_adr", &m.vdof_anc_adr[0][0], MAXV, MAXV); B.I("k_vdof_ndesc", m.vdof_ndesc, MAXV);
  B.I2("k_vdof_desc", &m.vdof_desc[0][0], MAXV, MAXV); B.I2("k_vdof_desc_adr", &m.vdof_desc_adr[0][0], MAXV, MAXV);
  B.I("k_nH", &m.nH, 1); B.I("k_H_i", m.H_i, MAXNZ); B.I("k_H_j", m.H_j, MAXNZ); B.I("k_H_src", m.H_src, MAXNZ);
  B.I("k_tri_m", m.tri_m, MAXNZ); B.I("k_tri_q", m.tri_q, MAXNZ);
  // actuators
  B.I("k_act_qposadr", m.act_qposadr, MAXU); B.I("k_act_dofadr", m.act_dofadr, MAXU); B.I("k_act_backlash_qposadr", m.act_backlash_qposadr, MAXU);
  B.F("actuator_gainprm0", m.act_kp, MAXU);
  {
    float bias[MAXU * 3], gear[MAXU];
    B.F("actuator_biasprm", bias, MAXU * 3); B.F("actuator_gear", gear, MAXU);
    for (int u = 0; u < m.nu; u++) {
      m.act_bias1[u] = bias[3 * u + 1]; m.act_bias2[u] = bias[3 * u + 2];
      if (bias[3 * u] != 0.0f || gear[u] != 1.0f || fabsf(bias[3 * u + 1] + m.act_kp[u]) > 1e-6f) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "only gear-1 position actuators are supported"); }
    }
  }
  B.F("actuator_ctrlrange", &m.act_ctrlrange[0][0], MAXU * 2); B.F("actuator_forcerange", &m.act_forcerange[0][0], MAXU * 2);
  B.I("actuator_ctrllimited", m.act_ctrllimited, MAXU); B.I("actuator_forcelimited", m.act_forcelimited, MAXU);
  // rows
  m.nfl = B.I("k_fl_dof", m.fl_dof, MAXV); m.nlim = B.I("k_lim_jnt", m.lim_jnt, MAXJ);
  if (!B.ok) { delete mo; return fail(ODK_ERR_INVALID, "odk_model_load: missing %s", B.missing.c_str()); }
  m.nrow = m.nfl + m.nlim + 48;
  double dof_solref[MAXV * 2], dof_solimp[MAXV * 5], dof_iw[MAXV], jnt_solref[MAXJ * 2], jnt_solimp[MAXJ * 5], jnt_margin[MAXJ];
  B.D("dof_solref", dof_solref, MAXV * 2); B.D("dof_solimp", dof_solimp, MAXV * 5); B.D("dof_invweight0", dof_iw, MAXV);
  B.D("jnt_solref", jnt_solref, MAXJ * 2); B.D("jnt_solimp", jnt_solimp, MAXJ * 5); B.D("jnt_margin", jnt_margin, MAXJ);
  for (int r = 0; r < m.nfl; r++) {
    int d = m.fl_dof[r];
    double R, bb;
    row_consts(dof_solref + 2 * d, dof_solimp + 5 * d, dtv[0], dof_iw[d], &R, &bb);
    m.fl_R[r] = (float)R; m.fl_D[r] = (float)(1.0 / R); m.fl_b[r] = (float)bb;
  }
  for (int r = 0; r < m.nlim; r++) {
    int j = m.lim_jnt[r];
    if (jnt_margin[j] != 0) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "joint margin"); }
    for (int k = 0; k < 2; k++) m.lim_solref[r][k] = (float)jnt_solref[2 * j + k];
    for (int k = 0; k < 5; k++) m.lim_solimp[r][k] = (float)jnt_solimp[5 * j + k];
    pack_imp(m.lim_solref[r], m.lim_solimp[r], m.dt, m.lim_imp[r]);
    m.lim_invweight[r] = (float)dof_iw[m.jnt_dofadr[j]];
  }
  // geoms: feet + floor
  int foot_cg[2], floor_cg[1], cg_type[4], cg_body[4], cg_prio[4], cg_vadr[4], cg_vnum[4], cg_fadr[4], cg_fnum[4], cg_condim[4];
  double cg_pos[12], cg_quat[16], cg_fric[12], cg_solref[8], cg_solimp[20], cg_solmix[4], hv[64 * 3], biw[MAXB * 2];
  int hf[128 * 3];
  B.I("k_foot_cgeom", foot_cg, 2); B.I("k_floor_cgeom", floor_cg, 1); int ncg = B.I("cgeom_type", cg_type, 4);
  B.I("cgeom_bodyid", cg_body, 4); B.I("cgeom_priority", cg_prio, 4); B.I("cgeom_condim", cg_condim, 4);
  B.I("cgeom_vertadr", cg_vadr, 4); B.I("cgeom_vertnum", cg_vnum, 4); B.I("cgeom_faceadr", cg_fadr, 4); B.I("cgeom_facenum", cg_fnum, 4);
  B.D("cgeom_pos", cg_pos, 12); B.D("cgeom_quat", cg_quat, 16); B.D("cgeom_friction", cg_fric, 12);
  B.D("cgeom_solref", cg_solref, 8); B.D("cgeom_solimp", cg_solimp, 20); B.D("cgeom_solmix", cg_solmix, 4);
  double cg_size[12] = {0};
  { const bool was_ok = B.ok; const std::string miss = B.missing; B.D("cgeom_size", cg_size, 12); B.ok = was_ok; B.missing = miss; }   // optional: absent in blobs without primitive colliders
  double cg_margin[4] = {0};
  { const bool was_ok = B.ok; const std::string miss = B.missing; B.D("cgeom_margin", cg_margin, 4); B.ok = was_ok; B.missing = miss; }     // optional: blobs of rounds 1-3 have none (= 0)
  int nhv = B.D("hull_vert", hv, 64 * 3) / 3; int nhf = B.I("hull_face", hf, 128 * 3) / 3;
  B.D("body_invweight0", biw, MAXB * 2);
  B.I("k_foot_body", m.foot_body, 2); B.I2("k_foot_dofmask", &m.foot_dofmask[0][0], 2, MAXV);
  for (int p = 0; p < m.nM && p < MAXNZ; p++) {
    const int i = m.M_i[p], j = m.M_j[p];
    const int fi = m.foot_dofmask[0][i] | (m.foot_dofmask[1][i] << 1), fj = m.foot_dofmask[0][j] | (m.foot_dofmask[1][j] << 1);
    m.M_ent[p] = i | (j << 5) | (fi << 10) | (fj << 12);
  }
  if (!B.ok || ncg != 3) { delete mo; return fail(ODK_ERR_INVALID, "odk_model_load: missing %s", B.missing.c_str()); }
  (void)nhv; (void)nhf;
  {   // the kernels' pair structure is fixed: floor x left foot, floor x right foot, left foot x right foot.  MuJoCo collides geoms g1, g2 when
      // (contype1 & conaffinity2) || (contype2 & conaffinity1): a model whose masks leave one of the three out would get a pair it does not have
    int ct[4] = {1, 1, 1, 1}, ca[4] = {1, 1, 1, 1};
    { const bool was_ok = B.ok; const std::string miss = B.missing; B.I("cgeom_contype", ct, 4); B.I("cgeom_conaffinity", ca, 4); B.ok = was_ok; B.missing = miss; }
    auto collide = [&](int g1, int g2) { return ((ct[g1] & ca[g2]) | (ct[g2] & ca[g1])) != 0; };
    if (!collide(floor_cg[0], foot_cg[0]) || !collide(floor_cg[0], foot_cg[1]) || !collide(foot_cg[0], foot_cg[1])) {
      delete mo;
      return fail(ODK_ERR_UNSUPPORTED, "contype / conaffinity exclude one of the three geom pairs the kernels collide (floor x each foot, foot x foot)");
    }
  }
  for (int g = 0; g < 3; g++)   // the culls (height-field prisms, foot-foot boxes) drop every pair with a positive gap: only valid at margin 0
    if (cg_margin[g] != 0) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "collision geom %d has margin %g: contacts are detected at distance 0", g, cg_margin[g]); }
  m.foot_prim = 0;
  for (int f = 0; f < 2; f++) {
    int g = foot_cg[f];
    if (cg_condim[g] != 3) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "foot collider condim %d: the contact rows are pyramidal condim-3", cg_condim[g]); }
    if (cg_vnum[g] > HULL_MAXV || cg_fnum[g] > MAXHF) {
      delete mo; return fail(ODK_ERR_UNSUPPORTED, "foot hull with %d vertices / %d triangles: the kernels hold <= %d vertices and <= %d merged faces", cg_vnum[g], cg_fnum[g], HULL_MAXV, HULL_MAXF);
    }
    double gm[9];
    quat2mat(cg_quat + 4 * g, gm);
    m.foot_gtype[f] = cg_type[g];
    for (int k = 0; k < 3; k++) { m.foot_gpos[f][k] = (float)cg_pos[3 * g + k]; m.foot_gaxis[f][k] = (float)gm[3 * k + 2]; m.foot_gsize[f][k] = (float)cg_size[3 * g + k]; }
    if (cg_type[g] == 2 || cg_type[g] == 3) {   // sphere / capsule foot: no hull; bounding box for the records only
      if (!(cg_size[3 * g] > 0) || (cg_type[g] == 3 && !(cg_size[3 * g + 1] > 0))) { delete mo; return fail(ODK_ERR_INVALID, "primitive foot collider without a size"); }
      m.foot_prim = 1;
      m.foot_nvert[f] = 0; m.foot_nface[f] = 0; m.foot_npoly[f] = 0; m.foot_nedge[f] = 0;
      const double hz = cg_type[g] == 3 ? cg_size[3 * g] + cg_size[3 * g + 1] : cg_size[3 * g];
      m.foot_obb_half[f][0] = m.foot_obb_half[f][1] = (float)cg_size[3 * g]; m.foot_obb_half[f][2] = (float)hz;
      for (int k = 0; k < 3; k++) { m.foot_obb_center[f][k] = (float)cg_pos[3 * g + k]; m.foot_centroid[f][k] = (float)cg_pos[3 * g + k]; }
      for (int k = 0; k < 9; k++) m.foot_obb_axes[f][k] = (float)gm[k];
      continue;
    }
    if (cg_type[g] != 7) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "foot collider must be a convex mesh / box hull, a sphere or a capsule"); }
    m.foot_nvert[f] = cg_vnum[g]; m.foot_nface[f] = cg_fnum[g];
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    for (int v = 0; v < cg_vnum[g]; v++) {
      const double* p = hv + 3 * (cg_vadr[g] + v);
      for (int k = 0; k < 3; k++) {
        m.foot_vert[f][v][k] = (float)(cg_pos[3 * g + k] + gm[3 * k] * p[0] + gm[3 * k + 1] * p[1] + gm[3 * k + 2] * p[2]);
        lo[k] = fmin(lo[k], p[k]); hi[k] = fmax(hi[k], p[k]);
      }
    }
    for (int t = 0; t < cg_fnum[g]; t++) for (int k = 0; k < 3; k++) m.foot_face[f][t][k] = hf[3 * (cg_fadr[g] + t) + k];
    double cl[3];
    for (int k = 0; k < 3; k++) { cl[k] = 0.5 * (lo[k] + hi[k]); m.foot_obb_half[f][k] = (float)(0.5 * (hi[k] - lo[k])); }
    for (int k = 0; k < 3; k++) m.foot_obb_center[f][k] = (float)(cg_pos[3 * g + k] + gm[3 * k] * cl[0] + gm[3 * k + 1] * cl[1] + gm[3 * k + 2] * cl[2]);
    for (int k = 0; k < 9; k++) m.foot_obb_axes[f][k] = (float)gm[k];
    {   // polygons / edges / normals of the hull in the body frame (double precision, then rounded)
      double bv[MAXHV][3];
      int tr[MAXHF][3];
      for (int v = 0; v < cg_vnum[g]; v++) {
        const double* p = hv + 3 * (cg_vadr[g] + v);
        for (int k = 0; k < 3; k++) bv[v][k] = cg_pos[3 * g + k] + gm[3 * k] * p[0] + gm[3 * k + 1] * p[1] + gm[3 * k + 2] * p[2];
      }
      for (int t = 0; t < cg_fnum[g]; t++) for (int k = 0; k < 3; k++) tr[t][k] = hf[3 * (cg_fadr[g] + t) + k];
      if (!build_convex_tables(bv, cg_vnum[g], tr, cg_fnum[g], &m.foot_npoly[f], m.foot_poly[f], m.foot_fnorm[f], &m.foot_nedge[f], m.foot_edge[f],
                               m.foot_centroid[f], HULL_MAXF, HULL_MAXE)) {
        delete mo; return fail(ODK_ERR_UNSUPPORTED, "foot hull: not a closed polytope with <= 4-vertex faces, <= %d merged faces and <= %d edges", HULL_MAXF, HULL_MAXE);
      }
      for (int j = 0; j < 16; j++) {   // what a row lane keeps in registers over the height-field pair loop, as ONE 32-byte record
        for (int sl = 0; sl < 3; sl++) {
          const int jb = j + 16 * sl; const bool on = jb < m.foot_nedge[f]; const int* e = m.foot_edge[f][on ? jb : 0];
          m.foot_lane_rec[f][j][sl] = (int)((unsigned)e[2] | (unsigned)e[3] << 8 | (unsigned)e[0] << 16 | (unsigned)e[1] << 24 | (on ? 0u : 0x80000000u));
        }
        for (int sl = 0; sl < 2; sl++) {
          const int t = j + 16 * sl; const bool on = t < m.foot_npoly[f]; const int* pl = m.foot_poly[f][on ? t : 0];
          m.foot_lane_rec[f][j][3 + sl] = (int)((unsigned)pl[0] | (unsigned)pl[1] << 3 | (unsigned)pl[2] << 8 | (unsigned)pl[3] << 13 | (unsigned)pl[4] << 18 | (on ? 0u : 0x80000000u));
        }
        for (int sl = 5; sl < 8; sl++) m.foot_lane_rec[f][j][sl] = 0;
      }
      for (int t = 0; t < m.foot_npoly[f]; t++) {
        const double* v0 = bv[m.foot_poly[f][t][1]];
        m.foot_foff[f][t] = (float)(m.foot_fnorm[f][t][0] * v0[0] + m.foot_fnorm[f][t][1] * v0[1] + m.foot_fnorm[f][t][2] * v0[2]);
      }
    }
  }
  for (int f = 0; f < 2; f++) m.foot_sphere_r[f] = sqrtf(m.foot_obb_half[f][0] * m.foot_obb_half[f][0] + m.foot_obb_half[f][1] * m.foot_obb_half[f][1] + m.foot_obb_half[f][2] * m.foot_obb_half[f][2]);
  {   // a height-field prism's topology: the kernels' compile-time tables (odk_model.h) against this file's table builder
    const double pv[6][3] = {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 0, 0}, {1, 0, 0}, {0, 1, 0}};
    const int ptri[8][3] = {{0, 1, 2}, {3, 5, 4}, {0, 3, 4}, {0, 4, 1}, {1, 4, 5}, {1, 5, 2}, {2, 5, 3}, {2, 3, 0}};
    int np = 0, ne = 0, ppoly[5][5], pedge[9][4]; float fn[5][3], cc[3];
    bool same = build_convex_tables(pv, 6, ptri, 8, &np, ppoly, fn, &ne, pedge, cc, 5, 9) && np == 5 && ne == 9;
    for (int f = 0; same && f < 5; f++) for (int k = 0; k < 5; k++) same = same && ppoly[f][k] == PRISM_POLY[f][k];
    for (int k = 0; same && k < 9; k++) for (int t = 0; t < 4; t++) same = same && pedge[k][t] == PRISM_EDGE[k][t];
    if (!same) { delete mo; return fail(ODK_ERR_INVALID, "the kernels' compile-time prism tables disagree with build_convex_tables"); }
  }
  {
    int g = floor_cg[0];
    m.floor_is_plane = cg_type[g] == 0;
    double pm[9];
    quat2mat(cg_quat + 4 * g, pm);
    // floor body is static at the world origin in every reference scene
    double n[3] = {pm[2], pm[5], pm[8]};
    for (int k = 0; k < 3; k++) { m.plane_pos[k] = (float)cg_pos[3 * g + k]; m.plane_n[k] = (float)n[k]; }
    make_frame_h(n, m.plane_frame);
    for (int k = 0; k < 9; k++) m.floor_mat[k] = (float)pm[k];
    if (!m.floor_is_plane) {   // height field samples + size (scene_rough_terrain_backlash.xml:22)
      RecHdr hh;
      const unsigned char* hp = find_rec((const unsigned char*)blob, len, "hfield_data", &hh);
      double hs[4];
      if (!hp || hh.dtype != 0 || hh.ndim != 2 || B.D("hfield_size", hs, 4) != 4) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "height-field floor without hfield_data / hfield_size"); }
      m.hfield_nrow = (int)hh.shape[0]; m.hfield_ncol = (int)hh.shape[1];
      for (int k = 0; k < 4; k++) m.hfield_size[k] = (float)hs[k];
      mo->hfield.resize((size_t)m.hfield_nrow * m.hfield_ncol);
      for (size_t i = 0; i < mo->hfield.size(); i++) { double v; memcpy(&v, hp + 8 * i, 8); mo->hfield[i] = (float)v; }
      // hfield_contacts works on a window of <= 3 x 3 cells under the hull's oriented box (18 prisms per foot: the LIST region):
      // whatever the foot's orientation, its box must span less than two cells per axis (MJX sizes its sub-grid from the same
      // ratio at trace time; a finer field or a larger foot needs a larger window here, not silently dropped cells)
      if (m.hfield_nrow < 2 || m.hfield_ncol < 2) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "height field smaller than 2 x 2 samples"); }
      if (m.foot_prim && !((m.foot_gtype[0] == 2 || m.foot_gtype[0] == 3) && (m.foot_gtype[1] == 2 || m.foot_gtype[1] == 3))) {
        delete mo; return fail(ODK_ERR_UNSUPPORTED, "height-field floor: both feet are hulls (hfield_convex) or both are spheres / capsules (hfield_sphere / hfield_capsule)");
      }
      const double cell = fmin(2.0 * hs[0] / (m.hfield_ncol - 1), 2.0 * hs[1] / (m.hfield_nrow - 1));
      for (int f = 0; f < 2; f++) {
        const float* hh2 = m.foot_obb_half[f];
        const double diag = 2.0 * sqrt((double)hh2[0] * hh2[0] + (double)hh2[1] * hh2[1] + (double)hh2[2] * hh2[2]);
        if (!(diag < 2.0 * cell)) {
          delete mo; return fail(ODK_ERR_UNSUPPORTED, "foot %d spans %.4f m, the height field's cells are %.4f m: the prism window holds feet smaller than two cells", f, diag, cell);
        }
      }
    }
    // contact parameter mixing (mj_contactParam): pairs 0,1 = floor vs foot, pair 2 = foot vs foot
    for (int pr = 0; pr < 3; pr++) {
      int g1 = pr < 2 ? g : foot_cg[0], g2 = pr < 2 ? foot_cg[pr] : foot_cg[1];
      double mix;
      if (cg_prio[g1] > cg_prio[g2]) mix = 1; else if (cg_prio[g2] > cg_prio[g1]) mix = 0;
      else { double s1 = cg_solmix[g1], s2 = cg_solmix[g2]; mix = (s1 >= 1e-15 && s2 >= 1e-15) ? s1 / (s1 + s2) : ((s1 < 1e-15 && s2 < 1e-15) ? 0.5 : (s1 < 1e-15 ? 0.0 : 1.0)); }
      for (int k = 0; k < 2; k++) m.pair_solref[pr][k] = (float)(mix * cg_solref[2 * g1 + k] + (1 - mix) * cg_solref[2 * g2 + k]);
      for (int k = 0; k < 5; k++) m.pair_solimp[pr][k] = (float)(mix * cg_solimp[5 * g1 + k] + (1 - mix) * cg_solimp[5 * g2 + k]);
      pack_imp(m.pair_solref[pr], m.pair_solimp[pr], m.dt, m.pair_imp[pr]);
      double mu = cg_prio[g1] > cg_prio[g2] ? cg_fric[3 * g1] : (cg_prio[g2] > cg_prio[g1] ? cg_fric[3 * g2] : fmax(cg_fric[3 * g1], cg_fric[3 * g2]));
      m.pair_mu[pr] = (float)mu;
      double t = biw[2 * cg_body[g1]] + biw[2 * cg_body[g2]];
      // pyramidal rows: the pyramid edge's weight; elliptic cones: the two bodies' translational weights (the normal row's; the kernels scale the tangents)
      m.pair_invweight[pr] = m.cone ? (float)t : (float)((t + mu * mu * t) * 2 * mu * mu / (double)m.impratio);
    }
  }
  // sites / sensors
  B.I("site_bodyid", m.site_body, MAXSITE); B.F("site_pos", &m.site_pos[0][0], MAXSITE * 3); B.F("site_quat", &m.site_quat[0][0], MAXSITE * 4);
  {
    double sq[MAXSITE * 4];
    B.D("site_quat", sq, MAXSITE * 4);
    for (int s = 0; s < m.nsite; s++) { double mm[9]; quat2mat(sq + 4 * s, mm); for (int k = 0; k < 9; k++) m.site_mat[s][k] = (float)mm[k]; }
  }
  B.I("k_site_imu", &m.site_imu, 1); B.I("k_site_feet", m.site_feet, 2);
  m.nsensor = B.I("sensor_type", m.sensor_type, MAXSENS); B.I("sensor_objid", m.sensor_site, MAXSENS); B.I("sensor_adr", m.sensor_adr, MAXSENS);
  int adr[7];
  B.I("k_adr", adr, 7);
  m.adr_gyro = adr[0]; m.adr_local_linvel = adr[1]; m.adr_accelerometer = adr[2]; m.adr_upvector = adr[3]; m.adr_global_angvel = adr[4];
  m.adr_foot_linvel[0] = adr[5]; m.adr_foot_linvel[1] = adr[6];
  int nsd[1]; B.I("nsensordata", nsd, 1);
  if (!B.ok) { delete mo; return fail(ODK_ERR_INVALID, "odk_model_load: missing %s", B.missing.c_str()); }
  if (nsd[0] != NSENSD) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "sensordata size %d != %d", nsd[0], NSENSD); }
  if (eulerdamp != 0 || m.iterations != 1) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "kernels implement iterations=1, eulerdamp=disable (open_duck_mini_v2.xml:6-8)"); }
  for (int s = 0; s < m.nsensor; s++) {
    int b = m.site_body[m.sensor_site[s]];
    bool ok = (b == m.base_body) || (b == m.foot_body[0]) || (b == m.foot_body[1]);
    if (!ok || ((m.sensor_type[s] == 2 || m.sensor_type[s] == 8) && b != m.base_body)) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "sensor %d placement", s); }
  }
  // bodies above the serial chains that have children: flattened source lists for the one-step subtree fold (P2)
  m.np_count = 0;
  for (int b2 = 0; b2 < m.nb; b2++) {
    if (!(m.body_level[b2] >= 0 && m.body_nchild[b2] > 0 && !m.body_is_path[b2])) continue;
    if (m.np_count == 4) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "more than four branching bodies above the serial chains"); }
    const int i = m.np_count++;
    m.np_body[i] = b2; m.np_nsrc[i] = 0;
    for (int c2 = 0; c2 < m.nb; c2++) {   // c2 in the subtree of b2 (or b2 itself) and either not a chain body, or a chain head
      bool below = false;
      for (int a2 = c2; a2 > 0; a2 = m.body_parent[a2]) if (a2 == b2) { below = true; break; }
      if (!below || m.body_level[c2] < 0) continue;
      if (!m.body_is_path[c2] || m.body_path_head[c2]) {
        if (m.np_nsrc[i] == 6) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "more than six sources in a subtree fold"); }
        m.np_src[i][m.np_nsrc[i]++] = c2;
      }
    }
  }
  if (!build_reduced_tables(m)) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "dof tree is not a floating base with up to three serial chains of <= 5 (twin-merged) dofs"); }
  fill_body_st(m);
  int dt_max = 0, dv_max = 0;
  for (int d = 0; d < m.nv; d++) { dt_max = m.dof_depth[d] > dt_max ? m.dof_depth[d] : dt_max; dv_max = m.vdof_depth[d] > dv_max ? m.vdof_depth[d] : dv_max; }
  mo->shape = -1;
#define X(i, S) if (mo->shape < 0 && m.nq == S::NQ && m.nv == S::NV && m.nb == S::NB && m.nu == S::NU && m.nj == S::NJ && m.nM == S::NM && m.nH == S::NH && m.nrow == S::NROW && \
                    dt_max <= S::DT && dv_max <= S::DV) mo->shape = i;
  ODK_SHAPES(X)
#undef X
  if (mo->shape < 0) {
    const int dtm = dt_max, dvm = dv_max;
    delete mo;
    return fail(ODK_ERR_UNSUPPORTED, "model shape nq=%d nv=%d nb=%d nu=%d nj=%d nM=%d nH=%d nrow=%d depth=%d vdepth=%d has no compiled kernel (tools/new_shape.py <xml> prints the two lines to add to odk_engine.hip)",
                m.nq, m.nv, m.nb, m.nu, m.nj, m.nM, m.nH, m.nrow, dtm, dvm);
  }
  int shape_cl = 5; bool shape_eq = false;
#define X(i, S) if (mo->shape == i) { shape_cl = S::CL; shape_eq = S::EQ; }
  ODK_SHAPES(X)
#undef X
  for (int c = 0; c < m.nrchain; c++)
    if (m.rchain_len[c] > shape_cl) { const int len = m.rchain_len[c]; delete mo; return fail(ODK_ERR_UNSUPPORTED, "a serial chain of %d (twin-merged) dofs: the kernels of this model shape solve chains of <= %d", len, shape_cl); }
  if (!m.floor_is_plane && mo->shape != 1) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "height-field floors are built for the backlash model only"); }

  if (m.cone && m.foot_prim != 0) {
    delete mo; return fail(ODK_ERR_UNSUPPORTED, "<option cone=\"elliptic\">: the elliptic-cone kernels are built for convex (box / mesh) feet, not sphere / capsule feet");
  }
  {   // equality rows of the kernels: joint couplings inside one serial chain of a shape compiled with them
    m.neq = 0;
    for (int d = 0; d < MAXV; d++) m.dof_eqrow[d] = -1;
    // <equality><connect | weld>: "path rows" (odk_kernels.h) -- the two bodies on ONE root-to-leaf path of the tree, or body2 = the world,
    // so that the rows' J^T D J only touches entries the tree layout has; at most EQP_MAX constraints / EQP_ROWS rows.  In MJX's row order:
    // connects first, then welds.
    m.neqp = 0; m.eqp_nrow = 0; m.eqp_cross = 0;
    for (int d = 0; d < MAXV; d++) m.dof_eqp[d] = 0;
    for (int pass = 0; pass < 2; pass++)
      for (int k = 0; k < eq_n; k++) {
        if (!eq_active[k] || eq_type[k] != pass) continue;
        const char* kind = pass == 0 ? "connect" : "weld";
        const bool shape_ok = shape_eq && !m.paired;      // shapes compiled with the optional constraint code (Shape::EQ)
        const int nrow = pass == 0 ? 3 : 6;
        if (!shape_ok || m.neqp == EQP_MAX || m.eqp_nrow + nrow > EQP_ROWS) {
          delete mo;
          return fail(ODK_ERR_UNSUPPORTED, "<equality><%s> (constraint %d) is active: %s", kind, k,
                      shape_ok ? "the kernels hold at most two connect / weld constraints with nine rows in total" : "equality rows are compiled into the third and fourth model shapes only (the duck's kernels carry none)");
        }
        const int c = m.neqp, b1 = eq_o1[k], b2 = eq_o2[k] < 0 ? 0 : eq_o2[k];
        if (b1 < 1 || b1 >= m.nb || b2 >= m.nb) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "<equality><%s> (constraint %d): bad body ids", kind, k); }
        // dofs above a body: the dofs of the body itself and of its ancestors
        unsigned above[2] = {0u, 0u};
        for (int s2 = 0; s2 < 2; s2++)
          for (int b = s2 ? b2 : b1; b > 0; b = m.body_parent[b])
            for (int d = 0; d < m.nv; d++) if (m.dof_body[d] == b) above[s2] |= 1u << d;
        if ((above[0] & above[1]) != above[0] && (above[0] & above[1]) != above[1]) {
          // two chains: a closed loop.  The virtual tree (the Hessian layout of an active foot-foot contact: second leg below the first foot)
          // has an entry for every pair of dofs of base + the two foot chains -- a loop between exactly those is taken, on that layout
          unsigned legs = 0x3Fu;
          for (int f = 0; f < 2; f++) for (int t = 0; t < m.foot_rchain_len[f]; t++) legs |= 1u << (m.foot_rchain_first[f] + t);
          if (((above[0] | above[1]) & ~legs) != 0u || m.paired) {
            delete mo;
            return fail(ODK_ERR_UNSUPPORTED, "<equality><%s> (constraint %d): the two bodies must lie on one root-to-leaf path of the tree (or body2 be the world), or on the two foot chains: another loop has no entries in the Hessian's layouts", kind, k);
          }
          m.eqp_cross = 1;
        }
        for (int d = 0; d < m.nv; d++) m.dof_eqp[d] |= (((above[0] >> d) & 1) << (2 * c)) | (((above[1] >> d) & 1) << (2 * c + 1));
        const double* da = eq_data + 11 * k;
        for (int a = 0; a < 3; a++) { m.eqp_a1[c][a] = (float)(pass == 0 ? da[a] : da[3 + a]); m.eqp_a2[c][a] = (float)(pass == 0 ? da[3 + a] : da[a]); }
        for (int a = 0; a < 4; a++) m.eqp_relq[c][a] = pass == 0 ? (a == 0 ? 1.0f : 0.0f) : (float)da[6 + a];
        m.eqp_ts[c] = pass == 0 ? 0.0f : (float)da[10];
        float sr[2] = {(float)eq_solref[2 * k], (float)eq_solref[2 * k + 1]}, si[5];
        for (int a = 0; a < 5; a++) si[a] = (float)eq_solimp[5 * k + a];
        pack_imp(sr, si, m.dt, m.eqp_imp[c]);
        m.eqp_invw[c][0] = (float)(biw[2 * b1] + biw[2 * b2]); m.eqp_invw[c][1] = (float)(biw[2 * b1 + 1] + biw[2 * b2 + 1]);
        m.eqp_type[c] = pass; m.eqp_b1[c] = b1; m.eqp_b2[c] = b2; m.eqp_row0[c] = m.eqp_nrow;
        m.eqp_nrow += nrow; m.neqp++;
      }
    for (int k = 0; k < eq_n; k++) {
      if (!eq_active[k] || eq_type[k] != 2) continue;
      const bool shape_ok = shape_eq && !m.paired;      // shapes compiled with the optional constraint code (Shape::EQ)
      if (!shape_ok || m.neq == EQ_MAX) {
        delete mo;
        return fail(ODK_ERR_UNSUPPORTED, "<equality><joint> (constraint %d) is active: %s", k,
                    shape_ok ? "the kernels hold at most two equality rows" : "equality rows are compiled into the third and fourth model shapes only (the duck's kernels carry none)");
      }
      const int r = m.neq, j1 = eq_o1[k], j2 = eq_o2[k];
      if (j1 < 1 || j1 >= m.nj || j2 >= m.nj || j2 == 0 || j2 == j1) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "<equality><joint> (constraint %d): hinge joints expected", k); }
      const int d1 = m.jnt_dofadr[j1], d2 = j2 > 0 ? m.jnt_dofadr[j2] : -1;
      if (m.dof_eqrow[d1] >= 0 || (d2 >= 0 && m.dof_eqrow[d2] >= 0)) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "<equality><joint> (constraint %d): a joint takes part in at most one equality row", k); }
      m.eq_dof1[r] = d1; m.eq_dof2[r] = d2; m.eq_qadr1[r] = m.jnt_qposadr[j1]; m.eq_qadr2[r] = j2 > 0 ? m.jnt_qposadr[j2] : 0;
      m.eq_key[r] = -1;
      if (d2 >= 0) {      // the Hessian entry (d1, d2) must exist in the reduced tree layout: same serial chain
        for (int p = 0; p < m.nMr; p++) {
          const int e = m.R_ent[p], i = e & 31, j = (e >> 5) & 31;
          if ((i == d1 && j == d2) || (i == d2 && j == d1)) m.eq_key[r] = e & 0x3FF;
        }
        if (m.eq_key[r] < 0) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "<equality><joint> (constraint %d): the two joints must lie on one serial chain (the coupling's Hessian term needs an entry of the tree layout)", k); }
      }
      for (int c = 0; c < 5; c++) m.eq_poly[r][c] = (float)eq_data[11 * k + c];
      float sr[2] = {(float)eq_solref[2 * k], (float)eq_solref[2 * k + 1]}, si[5];
      for (int c = 0; c < 5; c++) si[c] = (float)eq_solimp[5 * k + c];
      pack_imp(sr, si, m.dt, m.eq_imp[r]);
      m.eq_invweight[r] = (float)(dof_iw[d1] + (d2 >= 0 ? dof_iw[d2] : 0.0));
      m.dof_eqrow[d1] = r; if (d2 >= 0) m.dof_eqrow[d2] = r;
      m.neq++;
    }
  }
  if (mo->shape == 1 && !(m.paired && m.nvr == ShapeB::NVR && m.nMr == ShapeB::NMR && m.nHr == ShapeB::NHR)) {
    delete mo;
    return fail(ODK_ERR_UNSUPPORTED, "the 30-dof kernels expect backlash twins (same body, anchor and axis as their joint) over the 20-dof tree");
  }
  if (mo->shape != 1 && m.paired) { delete mo; return fail(ODK_ERR_UNSUPPORTED, "twin dofs in a model of the 20-dof shape"); }
  for (int lane = 0; lane < 64; lane++) {   // per-lane statics of the kernels (LaneSt)
    memset(&m.lane_st[lane], 0, sizeof(LaneSt));
#define X(i, S) if (mo->shape == i) compute_statics<S>(m.lane_st[lane], &m, lane);
    ODK_SHAPES(X)
#undef X
  }
  *out = mo;
  return ODK_OK;
}
extern "C" void odk_model_free(odk_model* m) { delete m; }
extern "C" int odk_model_dims(const odk_model* m, int* nq, int* nv, int* nu, int* nbody) {
  if (!m) return fail(ODK_ERR_INVALID, "null model");
  if (nq) *nq = m->h.nq; if (nv) *nv = m->h.nv; if (nu) *nu = m->h.nu; if (nbody) *nbody = m->h.nb;
  return ODK_OK;
}
extern "C" int odk_model_obs_sizes(const odk_model* m, int env_kind, int* nobs, int* npriv) {
  if (!m) return fail(ODK_ERR_INVALID, "null model");
  obs_sizes_nu(m->h.nu, env_kind, nobs, npriv);
  return ODK_OK;
}

// occupancy by construction: 2 waves / SIMD = 8 single-wave workgroups per CU need <= 160 KiB / 8 of LDS per workgroup (2 envs)
#ifndef ODK_PROFILE   // (the phase-timing build carries 20 extra floats per env and may run 7 workgroups per CU)
static_assert(EnvL<ShapeA>::wg_floats(2) * sizeof(float) <= 20480, "shape A: LDS image too large for 8 workgroups per CU");
static_assert(EnvL<ShapeB>::wg_floats(2) * sizeof(float) <= 20480, "shape B: LDS image too large for 8 workgroups per CU");
#endif
extern "C" int odk_model_reduced(const odk_model* m, int* paired, int* nvr, int* nMr, int* nHr, int* red_main, int* red_twin) {
  if (!m) return fail(ODK_ERR_INVALID, "null model");
  if (paired) *paired = m->h.paired; if (nvr) *nvr = m->h.nvr; if (nMr) *nMr = m->h.nMr; if (nHr) *nHr = m->h.nHr;
  for (int r = 0; r < m->h.nvr; r++) { if (red_main) red_main[r] = m->h.red_main[r]; if (red_twin) red_twin[r] = m->h.red_twin[r]; }
  return ODK_OK;
}
extern "C" int odk_model_env_lds_floats(const odk_model* m) {
  if (!m) return -1;
#define X(i, S) if (m->shape == i) return EnvL<S>::TOTAL;
  ODK_SHAPES(X)
#undef X
  return -1;
}

template <class S> static void fill_sizes(odk_batch* b) {
  b->rec_size = Rec<S>::SIZE; b->frec_size = Rec<S>::FSIZE; b->lds_total = S::TOTAL; b->dr_size = DRL<S>::SIZE; b->env_lds = EnvL<S>::TOTAL;
}

static void to_dev_cfg(const odk_env_config& c, EnvCfg& d, int nu) {
  d.ctrl_dt = c.ctrl_dt; d.action_scale = c.action_scale; d.dof_vel_scale = c.dof_vel_scale; d.max_motor_velocity = c.max_motor_velocity;
  d.noise_level = c.noise_level; d.noise_gyro = c.noise_gyro; d.noise_accelerometer = c.noise_accelerometer; d.noise_gravity = c.noise_gravity;
  d.noise_joint_vel = c.noise_joint_vel;
  memcpy(d.qpos_noise_scale, c.qpos_noise_scale, sizeof(d.qpos_noise_scale)); memcpy(d.reward_scales, c.reward_scales, sizeof(d.reward_scales));
  d.tracking_sigma = c.tracking_sigma; d.push_enable = c.push_enable;
  memcpy(d.push_interval_range, c.push_interval_range, 8); memcpy(d.push_magnitude_range, c.push_magnitude_range, 8);
  memcpy(d.cmd_range, c.cmd_range, sizeof(d.cmd_range));
  d.use_imitation = c.use_imitation; d.use_motor_speed_limits = c.use_motor_speed_limits; d.autoreset = c.autoreset;
  d.episode_length = c.episode_length; d.n_substeps = c.n_substeps;
  d.kind = c.env_kind; d.reset_base_qvel = c.reset_base_qvel;
  obs_sizes_nu(nu, c.env_kind, &d.nobs, &d.npriv);
}

extern "C" int odk_batch_create(const odk_model* m, const odk_env_config* cfg, int nenv, int device, const float* prm_table, const double* dxs, int nx,
                                const double* dys, int ny, const double* dths, int nth, const double* ranges6, int nsteps, odk_batch** out) {
  if (!m || !cfg || !out || nenv <= 0 || !prm_table || nx > 16 || ny > 16 || nth > 16) return fail(ODK_ERR_INVALID, "odk_batch_create: bad arguments");
  HIPCHK(hipSetDevice(device));
  odk_batch* b = new odk_batch();
  b->model = *m; b->nenv = nenv; b->device = device; b->cfg = *cfg;
  if (cfg->lanes_per_env != 0 && cfg->lanes_per_env != 32 && cfg->lanes_per_env != 64) { delete b; return fail(ODK_ERR_INVALID, "lanes_per_env must be 0, 32 or 64"); }
  // lanes_per_env is a geometry HINT (one env per wave finishes a small batch's step sooner): the instantiations that exist at 32 lanes per env only
  // -- elliptic cones (whatever set the model's opt_cone: the XML's <option cone> or a config switch), height-field floors, robots that are not the
  // duck -- run there whatever was asked (odk_batch_lanes reports it)
  b->G = (cfg->lanes_per_env == 64 && !m->h.cone && m->h.floor_is_plane && m->shape < 2) ? 64 : 32;
#define X(i, S) if (m->shape == i) fill_sizes<S>(b);
  ODK_SHAPES(X)
#undef X
  DevPRM hp;
  memset(&hp, 0, sizeof(hp));
  hp.nx = nx; hp.ny = ny; hp.nth = nth; hp.nsteps = nsteps;
  for (int i = 0; i < nx; i++) hp.dxs[i] = (float)dxs[i];
  for (int i = 0; i < ny; i++) hp.dys[i] = (float)dys[i];
  for (int i = 0; i < nth; i++) hp.dths[i] = (float)dths[i];
  for (int i = 0; i < 6; i++) hp.ranges[i] = (float)ranges6[i];
  size_t tbytes = (size_t)nx * ny * nth * 640 * sizeof(float);
  HIPCHK(hipMalloc(&b->d_model, sizeof(DevModel)));
  { DevModel hm = m->h; hm.hfield_filter = cfg->hfield_up_normals_only ? 3 : 0;
#ifdef ODK_HF_KNOCK   // timing experiment (make libodk_knock.so; WRONG results): parts of the height-field routine switched off by bit
    if (const char* e = getenv("ODK_HF_KNOCK")) hm.hfield_filter |= atoi(e) << 8;
#endif
    HIPCHK(hipMemcpy(b->d_model, &hm, sizeof(DevModel), hipMemcpyHostToDevice)); }   // (the batch's own copy: the filter is a batch setting)
  b->h_prm = hp;
  HIPCHK(hipMalloc(&b->d_table, tbytes)); HIPCHK(hipMemcpy(b->d_table, prm_table, tbytes, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&b->d_recs, (size_t)nenv * b->rec_size * sizeof(float))); HIPCHK(hipMemset(b->d_recs, 0, (size_t)nenv * b->rec_size * sizeof(float)));
  HIPCHK(hipMalloc(&b->d_first, (size_t)nenv * b->frec_size * sizeof(float))); HIPCHK(hipMemset(b->d_first, 0, (size_t)nenv * b->frec_size * sizeof(float)));
  HIPCHK(hipMalloc(&b->d_dbg, (size_t)nenv * b->lds_total * sizeof(float))); HIPCHK(hipMemset(b->d_dbg, 0, (size_t)nenv * b->lds_total * sizeof(float)));
  if (!m->hfield.empty()) {   // shared by all envs, L2-resident (256 KB)
    HIPCHK(hipMalloc(&b->d_hfield, m->hfield.size() * sizeof(float)));
    HIPCHK(hipMemcpy(b->d_hfield, m->hfield.data(), m->hfield.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = b;
  return ODK_OK;
}
extern "C" void odk_batch_destroy(odk_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void* p : {(void*)b->d_model, (void*)b->d_table, (void*)b->d_recs, (void*)b->d_first, (void*)b->d_dr, (void*)b->d_dbg, (void*)b->d_hfield}) (void)hipFree(p);
  for (auto& ev : b->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete b;
}
extern "C" int odk_batch_set_config(odk_batch* b, const odk_env_config* cfg) {
  if (!b || !cfg) return fail(ODK_ERR_INVALID, "null");
  if (cfg->env_kind != b->cfg.env_kind) return fail(ODK_ERR_INVALID, "odk_batch_set_config: env_kind is fixed at creation (it sets the output strides)");
  int g = b->cfg.lanes_per_env;
  if ((cfg->hfield_up_normals_only != 0) != (b->cfg.hfield_up_normals_only != 0)) {   // lives in the batch's device model (outside any step)
    const int v = cfg->hfield_up_normals_only ? 3 : 0;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(reinterpret_cast<char*>(b->d_model) + offsetof(DevModel, hfield_filter), &v, sizeof(int), hipMemcpyHostToDevice));
  }
  b->cfg = *cfg;
  b->cfg.lanes_per_env = g;
  return ODK_OK;
}

extern "C" int odk_batch_set_param(odk_batch* b, int param, const float* v, int count) {
  if (!b || !v) return fail(ODK_ERR_INVALID, "null");
  const DevModel& m = b->model.h;
  const int nb = m.nb, nu = m.nu;
  const int MASS = 0, IPOS = nb, FRL = nb + 3, ARM = FRL + nu, Q0 = ARM + nu, KP = Q0 + nu, SIZE = KP + nu;
  if (SIZE != b->dr_size) return fail(ODK_ERR_INVALID, "dr layout");
  if (!b->dr_enabled) {  // start from the nominal model
    b->h_dr.resize((size_t)b->nenv * SIZE);
    for (int e = 0; e < b->nenv; e++) {
      float* d = &b->h_dr[(size_t)e * SIZE];
      for (int i = 0; i < nb; i++) d[MASS + i] = m.body_mass[i];
      for (int k = 0; k < 3; k++) d[IPOS + k] = m.body_ipos[1][k];
      for (int u = 0; u < nu; u++) { d[FRL + u] = m.dof_frictionloss[m.act_dofadr[u]]; d[ARM + u] = m.dof_armature[m.act_dofadr[u]]; d[Q0 + u] = m.qpos0[m.act_qposadr[u]]; d[KP + u] = m.act_kp[u]; }
    }
    b->dr_enabled = true;
  }
  int off, n;
  switch (param) {
    case ODK_PARAM_BODY_MASS: off = MASS; n = nb; break;
    case ODK_PARAM_BODY_IPOS_TORSO: off = IPOS; n = 3; break;
    case ODK_PARAM_DOF_FRICTIONLOSS: off = FRL; n = nu; break;
    case ODK_PARAM_DOF_ARMATURE: off = ARM; n = nu; break;
    case ODK_PARAM_QPOS0: off = Q0; n = nu; break;
    case ODK_PARAM_KP: off = KP; n = nu; break;
    default: return fail(ODK_ERR_INVALID, "unknown param %d", param);
  }
  if (count != n) return fail(ODK_ERR_INVALID, "param %d expects %d values per env, got %d", param, n, count);
  for (int e = 0; e < b->nenv; e++) memcpy(&b->h_dr[(size_t)e * SIZE + off], v + (size_t)e * n, n * sizeof(float));
  HIPCHK(hipSetDevice(b->device));
  if (!b->d_dr) HIPCHK(hipMalloc(&b->d_dr, b->h_dr.size() * sizeof(float)));
  HIPCHK(hipMemcpy(b->d_dr, b->h_dr.data(), b->h_dr.size() * sizeof(float), hipMemcpyHostToDevice));
  return ODK_OK;
}

enum { K_RESET = 0, K_STEP = 1, K_PHYS = 2 };

template <class S, int G, int HF> static hipError_t launch_sg(int which, const KArgs& a, hipStream_t st) {
  const int per_block = 64 / G;
  const int grid = (a.nenv + per_block - 1) / per_block;
  size_t lds = (size_t)EnvL<S>::wg_floats(per_block) * sizeof(float);
#ifdef ODK_OCC_EXPERIMENT   // occupancy experiment (make libodk_occ.so): extra dynamic LDS per workgroup -> fewer workgroups per CU
  static const size_t pad = getenv("ODK_LDS_PAD") ? (size_t)atol(getenv("ODK_LDS_PAD")) : 0;
  lds += pad;
#endif
  if (which == K_RESET) hipLaunchKernelGGL((reset_kernel<S, G, HF>), dim3(grid), dim3(64), lds, st, a);
  else if (which == K_STEP) hipLaunchKernelGGL((step_kernel<S, G, HF>), dim3(grid), dim3(64), lds, st, a);
  else hipLaunchKernelGGL((physics_kernel<S, G, HF>), dim3(grid), dim3(64), lds, st, a);
  return hipGetLastError();
}
static hipError_t launch(odk_batch* b, int which, const KArgs& a, hipStream_t st) {
#if defined(ODK_DEV_C32)   // development build: the tail_biped shape alone
  if (b->model.shape == 2 && b->G == 32) return launch_sg<ShapeC, 32, 0>(which, a, st);
  return hipErrorNotSupported;
#elif defined(ODK_DEV_D32)   // development build: the six-dof biped's shape alone
  if (b->model.shape == 3 && b->G == 32) return launch_sg<ShapeD, 32, 0>(which, a, st);
  return hipErrorNotSupported;
#elif defined(ODK_DEV_B32)   // development builds: one instantiation only (make libodk_devB.so / libodk_devA.so: ~25 s instead of 2 min)
  if (b->model.shape == 1 && b->model.h.floor_is_plane && b->G == 32) return launch_sg<ShapeB, 32, 0>(which, a, st);
  return hipErrorNotSupported;
#elif defined(ODK_DEV_A32)
  if (b->model.shape == 0 && b->G == 32) return launch_sg<ShapeA, 32, 0>(which, a, st);
  return hipErrorNotSupported;
#elif defined(ODK_DEV_HF)
  if (!b->model.h.floor_is_plane) return launch_sg<ShapeB, 32, 1>(which, a, st);
  return hipErrorNotSupported;
#endif
  // height-field floors exist only with the backlash model (scene_rough_terrain_backlash.xml) and run 32 lanes per env; sphere / capsule
  // feet on one are their own instantiation (HF = 2): its out-of-line calls must not enter the duck kernel's register allocation
  if (!b->model.h.floor_is_plane) {
    if (b->model.h.cone) return launch_sg<ShapeBE, 32, 1>(which, a, st);      // (hull feet: checked at load)
    return b->model.h.foot_prim ? launch_sg<ShapeB, 32, 2>(which, a, st) : launch_sg<ShapeB, 32, 1>(which, a, st);
  }
  // robots that are not the duck (reference README.md:74-85): reset / step / physics kernels of their own shape, 32 lanes per env
#define X(i, S) if (i >= 2 && b->model.shape == i) return b->G == 32 ? launch_sg<S, 32, 0>(which, a, st) : hipErrorNotSupported;
  ODK_SHAPES(X)
#undef X
  if (b->model.h.cone) {      // the duck with elliptic cones (plane floor, checked at load): 32 lanes per env
    if (b->G != 32) return hipErrorNotSupported;
    return b->model.shape == 0 ? launch_sg<ShapeAE, 32, 0>(which, a, st) : launch_sg<ShapeBE, 32, 0>(which, a, st);
  }
  if (b->model.shape == 0) return b->G == 64 ? launch_sg<ShapeA, 64, 0>(which, a, st) : launch_sg<ShapeA, 32, 0>(which, a, st);
  return b->G == 64 ? launch_sg<ShapeB, 64, 0>(which, a, st) : launch_sg<ShapeB, 32, 0>(which, a, st);
}

static void base_args(odk_batch* b, KArgs& a, const odk_outputs* o) {
  memset(&a, 0, sizeof(a));
  a.m = b->d_model; a.prm = b->h_prm; a.prm_table = b->d_table; a.recs = b->d_recs; a.first = b->d_first;
  a.hfield = b->d_hfield;
  a.dr = b->dr_enabled ? b->d_dr : nullptr; a.nenv = b->nenv; a.n_substeps = b->cfg.n_substeps;
  a.dbg_lds = nullptr;
  if (o) { a.obs = o->obs_dev; a.priv = o->priv_dev; a.reward = o->reward_dev; a.done = o->done_dev; a.trunc = o->truncation_dev; a.metrics = o->metrics_dev; }
  to_dev_cfg(b->cfg, a.cfg, b->model.h.nu);
}

// The env kernels' task logic is joystick.py's with the robot's own tables (actuators, default pose, feet / imu sites, sensor addresses from the
// ModelBlob).  What stays the duck's: the imitation reward (its reference-motion table and joint map: custom_rewards.py:80-88) and the Standing
// task's head joints (standing.py:590-597, rewards.py:131-147 index the duck's actuators 5..8) -- refused by name for another robot's
// odk_reset / odk_step (odk_physics_step has no task logic)
static int env_logic_ok(const odk_batch* b) {
  if (b->model.shape >= 2 && (b->cfg.use_imitation || b->cfg.env_kind != ODK_ENV_JOYSTICK))
    return fail(ODK_ERR_UNSUPPORTED, "%s on a robot that is not the duck: the reference-motion table / the head joints are open_duck_mini_v2's (set use_imitation = 0, env_kind = ODK_ENV_JOYSTICK)",
                b->cfg.use_imitation ? "use_imitation" : "the Standing task");
  return ODK_OK;
}

static int g_debug_dump = 0;
extern "C" void odk_set_debug_dump(int on) { g_debug_dump = on; }

extern "C" int odk_reset(odk_batch* b, uint32_t seed, uint32_t env_id_offset, const odk_outputs* outs, void* stream) {
  if (!b) return fail(ODK_ERR_INVALID, "null batch");
  if (int rc = env_logic_ok(b)) return rc;
  HIPCHK(hipSetDevice(b->device));
  KArgs a;
  base_args(b, a, outs);
  a.seed = seed; a.env_offset = env_id_offset;
  a.dbg_lds = b->d_dbg;
  HIPCHK(launch(b, K_RESET, a, (hipStream_t)stream));
  return ODK_OK;
}

extern "C" int odk_step(odk_batch* b, const float* action_dev, const odk_outputs* outs, void* stream) {
  if (!b || !action_dev) return fail(ODK_ERR_INVALID, "null argument");
  if (int rc = env_logic_ok(b)) return rc;
  HIPCHK(hipSetDevice(b->device));
  KArgs a;
  base_args(b, a, outs);
  a.action = action_dev;
  a.dbg_lds = g_debug_dump ? b->d_dbg : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // (event pairs are created by odk_batch_timing, never here: a step allocates nothing; when the pool is used up the
  // remaining launches of the window simply go untimed)
  if (b->timing > 0 && (b->timing_count++ % (size_t)b->timing) == 0 && b->ev_used < b->events.size()) {
    HIPCHK(hipEventRecord(b->events[b->ev_used].first, st));
    HIPCHK(launch(b, K_STEP, a, st));
    HIPCHK(hipEventRecord(b->events[b->ev_used].second, st));
    b->ev_used++;
  } else {
    HIPCHK(launch(b, K_STEP, a, st));
  }
  return ODK_OK;
}

extern "C" int odk_physics_step(odk_batch* b, const float* ctrl_dev, int n_substeps, void* stream) {
  if (!b || !ctrl_dev) return fail(ODK_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  KArgs a;
  base_args(b, a, nullptr);
  a.action = ctrl_dev; a.n_substeps = n_substeps;
  a.dbg_lds = b->d_dbg;
  HIPCHK(launch(b, K_PHYS, a, (hipStream_t)stream));
  return ODK_OK;
}

extern "C" int odk_batch_timing(odk_batch* b, int enable, float* avg_ms, int* launches) {
  if (!b) return fail(ODK_ERR_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  double tot = 0;
  for (size_t i = 0; i < b->ev_used; i++) {
    float ms = 0;
    HIPCHK(hipEventSynchronize(b->events[i].second));
    HIPCHK(hipEventElapsedTime(&ms, b->events[i].first, b->events[i].second));
    tot += ms;
  }
  if (avg_ms) *avg_ms = b->ev_used ? (float)(tot / b->ev_used) : 0.0f;
  if (launches) *launches = (int)b->ev_used;
  b->ev_used = 0; b->timing_count = 0;
  b->timing = enable > 0 ? enable : 0;
  while (b->timing > 0 && b->events.size() < odk_batch::ODK_TIMING_EVENT_PAIRS) {
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    b->events.push_back({e0, e1});
  }
  return ODK_OK;
}

extern "C" int odk_batch_lanes(const odk_batch* b) { return b ? b->G : -1; }
extern "C" int odk_batch_record_size(const odk_batch* b) { return b ? b->rec_size : -1; }
extern "C" int odk_batch_lds_size(const odk_batch* b) { return b ? b->lds_total : -1; }
extern "C" int odk_record_field(const odk_batch* b, const char* name, int* offset, int* count, int* kind) {
  if (!b || !name) return fail(ODK_ERR_INVALID, "null argument");
  const int nq = b->model.h.nq, nv = b->model.h.nv, I = nq + 2 * nv, nu = b->model.h.nu;
  const RecLay RL = rec_lay(nu);
  struct F { const char* name; int off, n, kind; };
  const F tab[] = {
    {"qpos", 0, nq, 0}, {"qvel", nq, nv, 0}, {"qacc_warmstart", nq + nv, nv, 0},
    {"command", I + RL.CMD, 7, 0}, {"last_act", I + RL.LAST, nu, 0}, {"last_last_act", I + RL.LAST2, nu, 0},
    {"last_last_last_act", I + RL.LAST3, nu, 0}, {"motor_targets", I + RL.MT, nu, 0}, {"feet_air_time", I + RL.AIR, 2, 0},
    {"swing_peak", I + RL.PEAK, 2, 0}, {"push", I + RL.PUSH, 2, 0}, {"action_history", I + RL.AHIST, 3 * nu, 0},
    {"imu_history", I + RL.IMU, 9, 0}, {"steps", I + RL.EPSTEPS, 1, 0}, {"truncation", I + RL.TRUNC, 1, 0},
    {"episode_done", I + RL.DONE, 1, 0}, {"episode_metrics/sum_reward", I + RL.EPSUM, 1, 0}, {"episode_metrics/length", I + RL.EPLEN, 1, 0},
    {"episode_metrics/reward_terms", I + RL.EPMET, 8, 0}, {"rng", I + RL.KEY0, 3, 1}, {"step", I + RL.STEP, 1, 1},
    {"push_step", I + RL.PSTEP, 1, 1}, {"push_interval_steps", I + RL.PINT, 1, 1}, {"imitation_i", I + RL.IMI, 1, 1},
    {"last_contact", I + RL.LCON, 1, 2},
  };
  for (const F& f : tab)
    if (!strcmp(name, f.name)) {
      if (offset) *offset = f.off;
      if (count) *count = f.n;
      if (kind) *kind = f.kind;
      return ODK_OK;
    }
  return fail(ODK_ERR_INVALID, "unknown record field '%s'", name);
}
extern "C" int odk_batch_get_records(odk_batch* b, float* host) {
  if (!b || !host) return fail(ODK_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, b->d_recs, (size_t)b->nenv * b->rec_size * sizeof(float), hipMemcpyDeviceToHost));
  return ODK_OK;
}
extern "C" int odk_batch_set_records(odk_batch* b, const float* host) {   // checkpoint restore / test presets of the carried info
  if (!b || !host) return fail(ODK_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_recs, host, (size_t)b->nenv * b->rec_size * sizeof(float), hipMemcpyHostToDevice));
  return ODK_OK;
}
extern "C" int odk_batch_get_lds(odk_batch* b, float* host) {  // debug image of the last forward pass (reset / physics_step / step with dump on)
  if (!b || !host) return fail(ODK_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, b->d_dbg, (size_t)b->nenv * b->lds_total * sizeof(float), hipMemcpyDeviceToHost));
  return ODK_OK;
}
// named offsets into the LDS image for tests
template <class S> static int lds_off(const char* name) {
  if (!strcmp(name, "qpos")) return S::O_QPOS;
  if (!strcmp(name, "qvel")) return S::O_QVEL;
  if (!strcmp(name, "warm")) return S::O_WARM;
  if (!strcmp(name, "ctrl")) return S::O_CTRL;
  if (!strcmp(name, "xpos")) return S::O_XPOS;
  if (!strcmp(name, "xquat")) return S::O_XQUAT;
  if (!strcmp(name, "crb")) return S::O_CRB;
  if (!strcmp(name, "cdof")) return S::O_CDOF;
  if (!strcmp(name, "M")) return S::O_M;
  if (!strcmp(name, "HL")) return S::O_HL;
  if (!strcmp(name, "qfrc_smooth")) return S::O_QFS;
  if (!strcmp(name, "qacc_smooth")) return S::O_QAS;
  if (!strcmp(name, "x")) return S::O_X;
  if (!strcmp(name, "Ma")) return S::O_MA;
  if (!strcmp(name, "search")) return S::O_GRAD;
  if (!strcmp(name, "mv")) return S::O_MV;
  if (!strcmp(name, "efc_D")) return S::O_D;
  if (!strcmp(name, "efc_aref")) return S::O_AREF;
  if (!strcmp(name, "jar")) return S::O_JAR;
  if (!strcmp(name, "jv")) return S::O_JV;
  if (!strcmp(name, "W")) return S::O_W;
  if (!strcmp(name, "contact_dist")) return S::O_CDIST;
  if (!strcmp(name, "contact_r")) return S::O_CR;
  if (!strcmp(name, "scr")) return S::O_SCR;
  if (!strcmp(name, "sensordata")) return S::O_SENS;
  if (!strcmp(name, "actuator_force")) return S::O_ACTF;
  if (!strcmp(name, "qacc")) return S::O_QACC;
  return -1;
}
extern "C" int odk_lds_offset(const odk_batch* b, const char* name) {
  if (!b || !name) return -1;
#define X(i, S) if (b->model.shape == i) return lds_off<S>(name);
  ODK_SHAPES(X)
#undef X
  return -1;
}

extern "C" int odk_batch_get_state(odk_batch* b, float* qpos, float* qvel, float* warm) {
  if (!b) return fail(ODK_ERR_INVALID, "null");
  std::vector<float> h((size_t)b->nenv * b->rec_size);
  int rc = odk_batch_get_records(b, h.data());
  if (rc) return rc;
  const int nq = b->model.h.nq, nv = b->model.h.nv;
  for (int e = 0; e < b->nenv; e++) {
    const float* r = &h[(size_t)e * b->rec_size];
    if (qpos) memcpy(qpos + (size_t)e * nq, r, nq * sizeof(float));
    if (qvel) memcpy(qvel + (size_t)e * nv, r + nq, nv * sizeof(float));
    if (warm) memcpy(warm + (size_t)e * nv, r + nq + nv, nv * sizeof(float));
  }
  return ODK_OK;
}
extern "C" int odk_batch_set_state(odk_batch* b, const float* qpos, const float* qvel, const float* warm) {
  if (!b) return fail(ODK_ERR_INVALID, "null");
  std::vector<float> h((size_t)b->nenv * b->rec_size);
  int rc = odk_batch_get_records(b, h.data());
  if (rc) return rc;
  const int nq = b->model.h.nq, nv = b->model.h.nv;
  for (int e = 0; e < b->nenv; e++) {
    float* r = &h[(size_t)e * b->rec_size];
    if (qpos) memcpy(r, qpos + (size_t)e * nq, nq * sizeof(float));
    if (qvel) memcpy(r + nq, qvel + (size_t)e * nv, nv * sizeof(float));
    if (warm) memcpy(r + nq + nv, warm + (size_t)e * nv, nv * sizeof(float));
  }
  HIPCHK(hipMemcpy(b->d_recs, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return ODK_OK;
}
extern "C" int odk_batch_get_debug(odk_batch* b, float* sensordata, float* actuator_force, float* contact_dist, float* qacc) {
  if (!b) return fail(ODK_ERR_INVALID, "null");
  std::vector<float> h((size_t)b->nenv * b->lds_total);
  int rc = odk_batch_get_lds(b, h.data());
  if (rc) return rc;
  const int nv = b->model.h.nv, nu = b->model.h.nu;
  const int o_s = odk_lds_offset(b, "sensordata"), o_a = odk_lds_offset(b, "actuator_force"), o_c = odk_lds_offset(b, "contact_dist"), o_q = odk_lds_offset(b, "qacc");
  for (int e = 0; e < b->nenv; e++) {
    const float* r = &h[(size_t)e * b->lds_total];
    if (sensordata) memcpy(sensordata + (size_t)e * NSENSD, r + o_s, NSENSD * sizeof(float));
    if (actuator_force) memcpy(actuator_force + (size_t)e * nu, r + o_a, nu * sizeof(float));
    if (contact_dist) memcpy(contact_dist + (size_t)e * NCON, r + o_c, NCON * sizeof(float));
    if (qacc) memcpy(qacc + (size_t)e * nv, r + o_q, nv * sizeof(float));
  }
  return ODK_OK;
}
