// odk_learner.hip -- learner-side kernels of the PPO training step (reference common/runner.py:104-118 ->
// brax ppo.losses.compute_ppo_loss / compute_gae, optax clip_by_global_norm + adam), gfx950 only.
//
// These are the element-wise halves of one minibatch step: everything around the policy/value GEMMs
// (which stay on hipBLASLt / MFMA).  Each replaces a chain of 50-150 tiny launches by one launch:
//   gae_kernel        GAE + advantage normalisation statistics          (1 workgroup)
//   ppo_head_kernel   tanh-normal log-prob, clipped surrogate, value loss, sampled entropy AND their
//                     gradients w.r.t. the network outputs              (16 lanes per sample)
//   sqnorm_kernel     per-block partial sums of the squared gradient norm (flat gradient buffer; folded by adam_kernel)
//   adam_kernel       clip_by_global_norm + Adam on the flat parameter buffer
// All launches are stream-ordered on the caller's stream and contain no host synchronisation, so they can be
// captured into a HIP graph together with the GEMMs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <utility>
#include <vector>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "../../include/odk.h"

int odk_fail_(int code, const char* msg);   // odk_engine.hip
int odk_func_lds_attr_(const void* fn, int slot, int bytes);   // odk_mlp.hip

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {   // sum over an aligned 16-lane group
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ float block_sum(float v, float* sh) {   // blockDim.x multiple of 64, <= 1024
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.0f;
  for (int i = 0; i < nw; i++) t += sh[i];
  return t;
}
__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// One step (time t, walking backwards) of brax compute_gae for one trajectory, with the roundings PINNED (explicit fused / unfused
// operations): four kernels restate this recursion (generic, register-staged, LDS-staged, fused with the loss head) and the compiler's
// choice of which multiply-add to contract depends on the surrounding loop -- as plain expressions two of them differed in the last bit.
// mask = 1 - truncation, nt = 1 - termination.  Returns the advantage; updates acc, v_next, vs_next; vs_t out.
__device__ __forceinline__ float gae_step(float r, float v, float mask, float nt, float discount, float lambda_, float& acc, float& v_next, float& vs_next,
                                          float& vs_t) {
  const float dn = __fmul_rn(discount, nt);
  const float delta = __fmul_rn(__fsub_rn(__fmaf_rn(dn, v_next, r), v), mask);
  acc = __fmaf_rn(__fmul_rn(__fmul_rn(dn, mask), lambda_), acc, delta);
  vs_t = __fadd_rn(acc, v);
  const float a = __fmul_rn(__fsub_rn(__fmaf_rn(dn, vs_next, r), v), mask);
  v_next = v; vs_next = vs_t;
  return a;
}

// GAE over row-major [B, T]; one thread per trajectory, serial in time (brax compute_gae); truncation / termination are
// FLAGS (any non-zero value counts as 1, in all three kernels); then the
// population mean / std of the advantages (brax: (adv - mean) / (std + 1e-8), jnp.std => ddof 0).
// stats[0] = mean, stats[1] = 1 / (std + 1e-8).  Single workgroup.
__global__ void gae_kernel(const float* __restrict__ trunc, const float* __restrict__ term, const float* __restrict__ rew,
                           const float* __restrict__ val, const float* __restrict__ boot, float* __restrict__ vs,
                           float* __restrict__ adv, float* __restrict__ stats, int B, int T, float lambda_, float discount) {
  __shared__ float sh[16];
  float s = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const size_t o = (size_t)b * T;
    float acc = 0.0f, v_next = boot[b], vs_next = boot[b];
    for (int t = T - 1; t >= 0; t--) {
      const float mask = trunc[o + t] != 0.0f ? 0.0f : 1.0f, nt = term[o + t] != 0.0f ? 0.0f : 1.0f, v = val[o + t], r = rew[o + t];
      float vs_t;
      const float a = gae_step(r, v, mask, nt, discount, lambda_, acc, v_next, vs_next, vs_t);
      adv[o + t] = a;
      vs[o + t] = vs_t;
      s = __fadd_rn(s, a);
    }
  }
  if (!stats) return;
  const float n = (float)B * (float)T;
  const float mean = block_sum(s, sh) / n;
  float q = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x)
    for (int t = 0; t < T; t++) { const float d = adv[(size_t)b * T + t] - mean; q = __fmaf_rn(d, d, q); }   // own writes: visible
  const float var = block_sum(q, sh) / n;
  if (threadIdx.x == 0) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
}

// Same arithmetic, in the same order, for T <= TT and B <= blockDim.x: the recursion is serial in time, so in the
// kernel above every one of its T steps waits for four strided global loads (20 us for 256 x 20).  Here a thread first
// issues all 4 T loads of its trajectory (independent, all in flight together), then runs the recursion and the
// variance pass out of registers.
template <int TT>
__global__ void gae_kernel_reg(const float* __restrict__ trunc, const float* __restrict__ term, const float* __restrict__ rew,
                               const float* __restrict__ val, const float* __restrict__ boot, float* __restrict__ vs,
                               float* __restrict__ adv, float* __restrict__ stats, int B, int T, float lambda_, float discount) {
  __shared__ float sh[16];
  const int b = threadIdx.x;
  const bool live = b < B;
  const size_t o = (size_t)(live ? b : 0) * T;
  float mk[TT], nt[TT], v[TT], r[TT];
#pragma unroll
  for (int t = 0; t < TT; t++) {
    const bool on = live && t < T;
    const size_t i = on ? o + t : 0;
    mk[t] = trunc[i] != 0.0f ? 0.0f : 1.0f; nt[t] = term[i] != 0.0f ? 0.0f : 1.0f; v[t] = val[i]; r[t] = rew[i];
  }
  float acc = 0.0f, v_next = live ? boot[b] : 0.0f, vs_next = v_next, s = 0.0f;
#pragma unroll
  for (int t = TT - 1; t >= 0; t--) {
    if (live && t < T) {
      float vs_t;
      const float a = gae_step(r[t], v[t], mk[t], nt[t], discount, lambda_, acc, v_next, vs_next, vs_t);
      adv[o + t] = a;
      vs[o + t] = vs_t;
      s = __fadd_rn(s, a);
      r[t] = a;   // kept for the variance pass
    }
  }
  if (!stats) return;
  const float n = (float)B * (float)T;
  const float mean = block_sum(s, sh) / n;
  float q = 0.0f;
#pragma unroll
  for (int t = 0; t < TT; t++)
    if (live && t < T) { const float d = r[t] - mean; q = __fmaf_rn(d, d, q); }
  const float var = block_sum(q, sh) / n;
  if (threadIdx.x == 0) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
}

// Same arithmetic once more for B * T <= GAE_LDS_N: the [B, T] inputs are row-major, so a thread walking its own trajectory
// touches a different cache line per lane in every load (26 us for 256 x 20 through one CU's L1).  Here the workgroup
// copies the inputs to LDS with coalesced loads, runs the recursion out of LDS, parks adv / vs in the slots of rew / val
// and writes them back coalesced.
constexpr int GAE_LDS_N = 5120;
__global__ void __launch_bounds__(1024) gae_kernel_lds(const float* __restrict__ trunc, const float* __restrict__ term,
                                                       const float* __restrict__ rew, const float* __restrict__ val,
                                                       const float* __restrict__ boot, float* __restrict__ vs, float* __restrict__ adv,
                                                       float* __restrict__ stats, int B, int T, float lambda_, float discount) {
  __shared__ float s_r[GAE_LDS_N], s_v[GAE_LDS_N], s_f[GAE_LDS_N];   // s_f = trunc + 2 * term (both are 0 / 1 flags)
  __shared__ float sh[16];
  const int N = B * T;
  for (int i = threadIdx.x; i < N; i += blockDim.x) { s_r[i] = rew[i]; s_v[i] = val[i]; s_f[i] = (trunc[i] != 0.0f ? 1.0f : 0.0f) + (term[i] != 0.0f ? 2.0f : 0.0f); }
  __syncthreads();
  float s = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int o = b * T;
    float acc = 0.0f, v_next = boot[b], vs_next = v_next;
    for (int t = T - 1; t >= 0; t--) {
      const float f = s_f[o + t], te = f >= 2.0f ? 1.0f : 0.0f, tr = f - 2.0f * te;
      const float mask = 1.0f - tr, nt = 1.0f - te, v = s_v[o + t], r = s_r[o + t];
      float vs_t;
      const float a = gae_step(r, v, mask, nt, discount, lambda_, acc, v_next, vs_next, vs_t);
      s_r[o + t] = a;
      s_v[o + t] = vs_t;
      s = __fadd_rn(s, a);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) { adv[i] = s_r[i]; vs[i] = s_v[i]; }
  if (!stats) return;
  const float n = (float)B * (float)T;
  const float mean = block_sum(s, sh) / n;
  float q = 0.0f;
  for (int b = threadIdx.x; b < B; b += blockDim.x)
    for (int t = 0; t < T; t++) { const float d = s_r[b * T + t] - mean; q = __fmaf_rn(d, d, q); }
  const float var = block_sum(q, sh) / n;
  if (threadIdx.x == 0) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
}

// The loss head of ONE sample on its 16-lane row (lane j < A = action dimension j): tanh-normal log-prob, clipped surrogate, value
// loss, sampled entropy and their gradients w.r.t. the network outputs (brax ppo.losses.compute_ppo_loss).  Shared by the stand-alone
// head launch and the fused GAE + head launch: one arithmetic, two callers.  `advn`: the (normalised) advantage.
struct HeadOut { float dloc, dscale_raw, dbaseline, lp, lv, le; };
__device__ __forceinline__ HeadOut head_sample(bool on, float loc, float raw, float a, float z, float advn, float old_lp, float vs_s, float base_s, float inv_n,
                                               float eps, float entropy_cost, float grad_scale) {
  const float HALF_LOG_2PI = 0.91893853320467274f, LOG2 = 0.69314718055994531f;
  const float scale = softplus(raw) + 0.001f, inv_scale = 1.0f / scale, lscale = __logf(scale);
  const float u = (a - loc) * inv_scale;
  const float ldj_a = 2.0f * (LOG2 - a - softplus(-2.0f * a));
  const float lp_j = on ? (-0.5f * u * u - lscale - HALF_LOG_2PI - ldj_a) : 0.0f;
  const float x = loc + scale * z;
  const float ldj_x = 2.0f * (LOG2 - x - softplus(-2.0f * x));
  const float ent_j = on ? (0.5f + HALF_LOG_2PI + lscale + ldj_x) : 0.0f;
  const float logp = row16_sum(lp_j), ent = row16_sum(ent_j);
  const float rho = __expf(logp - old_lp);
  const float rc = fminf(fmaxf(rho, 1.0f - eps), 1.0f + eps);
  const float s1 = rho * advn, s2 = rc * advn;
  const bool inside = rho >= 1.0f - eps && rho <= 1.0f + eps;
  const float dmin_drho = (inside || s1 < s2) ? advn : 0.0f;   // d min(s1, s2) / d rho (ties split evenly, both branches -> rho)
  const float dL_dlogp = -dmin_drho * rho * inv_n;
  const float verr = vs_s - base_s;
  const float th = tanhf(x);
  const float ce = -entropy_cost * inv_n;
  HeadOut o;
  o.dloc = grad_scale * (dL_dlogp * u * inv_scale + ce * (-2.0f * th));
  o.dscale_raw = grad_scale * (dL_dlogp * (u * u - 1.0f) * inv_scale + ce * (inv_scale - 2.0f * th * z)) * sigmoid(raw);
  o.dbaseline = grad_scale * (-0.5f * verr * inv_n);
  o.lp = -fminf(s1, s2) * inv_n; o.lv = 0.25f * verr * verr * inv_n; o.le = -entropy_cost * ent * inv_n;
  return o;
}

// One 16-lane row per sample, lane j < A = action dimension j.  logits [n, 2A] = (loc | raw_scale).
// losses[0..3] += (total, policy, value, entropy) contributions (caller zeroes them).
__global__ void ppo_head_kernel(const float* __restrict__ logits, const float* __restrict__ raw_action, const float* __restrict__ old_logp,
                                const float* __restrict__ adv, const float* __restrict__ stats, const float* __restrict__ vs,
                                const float* __restrict__ baseline, const float* __restrict__ noise, float* __restrict__ dlogits,
                                float* __restrict__ dbaseline, float* __restrict__ losses, int n, int A, float eps, float entropy_cost,
                                float grad_scale) {
  const int lane = threadIdx.x & 15;
  const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = s < n, on = live && lane < A;
  const int sc = live ? s : n - 1;
  float loc = 0.f, raw = 0.f, a = 0.f, z = 0.f;
  if (on) { loc = logits[(size_t)sc * 2 * A + lane]; raw = logits[(size_t)sc * 2 * A + A + lane]; a = raw_action[(size_t)sc * A + lane]; z = noise[(size_t)sc * A + lane]; }
  float advn = adv[sc];
  if (stats) advn = (advn - stats[0]) * stats[1];
  const HeadOut h = head_sample(on, loc, raw, a, z, advn, old_logp[sc], vs[sc], baseline[sc], 1.0f / (float)n, eps, entropy_cost, grad_scale);
  if (on) {
    dlogits[(size_t)s * 2 * A + lane] = h.dloc;
    dlogits[(size_t)s * 2 * A + A + lane] = h.dscale_raw;
  }
  float lp = 0.f, lv = 0.f, le = 0.f;
  if (live && lane == 0) { dbaseline[s] = h.dbaseline; lp = h.lp; lv = h.lv; le = h.le; }
  // block totals first: one atomic per loss and workgroup (1 280 wave-level float atomics on four addresses serialised
  // into ~50 us; these sums are reporting only, the gradients above do not depend on them)
  __shared__ float sh[16];
  lp = block_sum(lp, sh); lv = block_sum(lv, sh); le = block_sum(le, sh);
  if (threadIdx.x == 0) {
    atomicAdd(&losses[0], lp + lv + le); atomicAdd(&losses[1], lp); atomicAdd(&losses[2], lv); atomicAdd(&losses[3], le);
  }
}

// GAE + advantage statistics + the loss head in ONE launch, the rollout read through the minibatch's trajectory indices
// (include/odk.h: odk_ppo_gae_head).  The per-step chain was gather (6.6 us) -> ... -> GAE (7.8 us, one workgroup) -> head (10.8 us):
// three launches of dependent latency.  The advantage statistics are a global quantity, so instead of a launch boundary every
// workgroup redoes the whole B x T recursion in its own LDS -- same arithmetic and order as gae_kernel_lds, hence the same bits in
// every workgroup (and on every data-parallel replica) -- and then works its own 64 samples with head_sample().
struct GaeHead {
  const float *logits, *values, *raw_action, *old_logp, *reward, *term, *trunc, *noise;
  const long long* idx; const int* cursor;
  float *dlogits, *dvalues, *losses, *loss_partials, *adv_out, *vs_out, *stats_out;
  int B, T, A, n_traj, normalize;
  float lambda_, discount, eps, entropy_cost, grad_scale;
};
// Workgroup = 512 threads = 32 samples (ODK_GAE_HEAD_SAMPLES): 160 workgroups for the reference minibatch, one per CU, two waves per SIMD
// for the head's transcendental arithmetic (the first version -- 1024 threads, 64 samples, 80 workgroups -- put 4 waves of it on every
// SIMD of 80 CUs and left 176 idle: 19.6 us; its 4 x 80 float atomics on four addresses were another ~4 us of serialisation).
constexpr int GH_THREADS = 16 * ODK_GAE_HEAD_SAMPLES;
__device__ __forceinline__ void block_sum3(float& x, float& y, float& z, float (*sh3)[16]) {   // three block sums for the price of one (same order as block_sum)
  x = wave_sum(x); y = wave_sum(y); z = wave_sum(z);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sh3[0][w] = x; sh3[1][w] = y; sh3[2][w] = z; }
  __syncthreads();
  float tx = 0.0f, ty = 0.0f, tz = 0.0f;
  for (int i = 0; i < nw; i++) { tx += sh3[0][i]; ty += sh3[1][i]; tz += sh3[2][i]; }
  x = tx; y = ty; z = tz;
}
__global__ void __launch_bounds__(GH_THREADS) ppo_gae_head_kernel(GaeHead a) {
  __shared__ float s_r[GAE_LDS_N], s_v[GAE_LDS_N];   // reward -> advantage, value -> vs
  __shared__ unsigned char s_f[GAE_LDS_N];           // trunc + 2 * term
  __shared__ int s_j[1024];                          // the minibatch's trajectory numbers, -1 = outside the rollout
  __shared__ float sh[16], sh3[3][16];
  const int B = a.B, T = a.T, N = B * T, A = a.A;
  const int k = a.cursor ? *a.cursor : 0;
  const long long* idx = a.idx + (size_t)k * B;
  const float nanv = __builtin_nanf("");
  for (int b = threadIdx.x; b < B; b += GH_THREADS) { const long long j = idx[b]; s_j[b] = (j >= 0 && j < a.n_traj) ? (int)j : -1; }
  __syncthreads();
  // staging: element i = (trajectory b = i / T, step t); the elements are independent: unrolled so that their loads are in flight together
  // (a trajectory outside the rollout is never read: it becomes NaN and the losses say so)
#pragma unroll 5
  for (int i = threadIdx.x; i < N; i += GH_THREADS) {
    const int b = i / T, t = i - b * T;
    const int j = s_j[b];
    const size_t o = (size_t)(j >= 0 ? j : 0) * T + t;
    const float r = a.reward[o], tr = a.trunc[o], te = a.term[o];
    s_r[i] = j >= 0 ? r : nanv; s_v[i] = a.values[i];
    s_f[i] = (unsigned char)((tr != 0.0f ? 1 : 0) + (te != 0.0f ? 2 : 0));
  }
  // this workgroup's own samples: everything the head needs that does not depend on the recursion, fetched BEFORE it
  const int lane = threadIdx.x & 15;
  const int smp = (blockIdx.x * GH_THREADS + threadIdx.x) >> 4;
  const bool live = smp < N, on = live && lane < A;
  const int sc = live ? smp : N - 1;
  const int hb = sc / T, ht = sc - hb * T;
  const int hj = s_j[hb];
  const bool hok = hj >= 0;
  const size_t ho = (size_t)(hok ? hj : 0) * T + ht;
  const float* noise = a.noise + (size_t)k * N * A;
  float loc = 0.f, raw = 0.f, act = 0.f, z = 0.f;
  if (on) { loc = a.logits[(size_t)sc * 2 * A + lane]; raw = a.logits[(size_t)sc * 2 * A + A + lane]; act = hok ? a.raw_action[ho * A + lane] : nanv; z = noise[(size_t)sc * A + lane]; }
  const float old_lp = hok ? a.old_logp[ho] : nanv, base_s = a.values[sc];
  __syncthreads();
  float s = 0.0f;
  for (int b = threadIdx.x; b < B; b += GH_THREADS) {
    const int o = b * T;
    float acc = 0.0f, v_next = a.values[N + b], vs_next = v_next;
#pragma unroll 4
    for (int t = T - 1; t >= 0; t--) {
      const int f = s_f[o + t];
      const float te = (f & 2) ? 1.0f : 0.0f, tr = (f & 1) ? 1.0f : 0.0f;
      const float mask = 1.0f - tr, nt = 1.0f - te, v = s_v[o + t], r = s_r[o + t];
      float vs_t;
      const float ad = gae_step(r, v, mask, nt, a.discount, a.lambda_, acc, v_next, vs_next, vs_t);
      s_r[o + t] = ad;
      s_v[o + t] = vs_t;
      s = __fadd_rn(s, ad);
    }
  }
  const float n = (float)B * (float)T;
  const float mean = block_sum(s, sh) / n;          // (its barriers also publish the recursion's LDS writes)
  float q = 0.0f;
  for (int b = threadIdx.x; b < B; b += GH_THREADS) {
#pragma unroll 4
    for (int t = 0; t < T; t++) { const float d = s_r[b * T + t] - mean; q = __fmaf_rn(d, d, q); }
  }
  const float var = block_sum(q, sh) / n;
  const float rstd = 1.0f / (sqrtf(var) + 1e-8f);
  if (blockIdx.x == 0) {     // optional copies of the intermediate results (tests, debugging)
    if (a.adv_out) for (int i = threadIdx.x; i < N; i += GH_THREADS) a.adv_out[i] = s_r[i];
    if (a.vs_out) for (int i = threadIdx.x; i < N; i += GH_THREADS) a.vs_out[i] = s_v[i];
    if (a.stats_out && threadIdx.x == 0) { a.stats_out[0] = mean; a.stats_out[1] = rstd; }
  }
  // ---- the head on this workgroup's samples
  float advn = s_r[sc];
  if (a.normalize) advn = (advn - mean) * rstd;
  const HeadOut h = head_sample(on, loc, raw, act, z, advn, old_lp, s_v[sc], base_s, 1.0f / n, a.eps, a.entropy_cost, a.grad_scale);
  if (on) {
    a.dlogits[(size_t)smp * 2 * A + lane] = h.dloc;
    a.dlogits[(size_t)smp * 2 * A + A + lane] = h.dscale_raw;
  }
  float lp = 0.f, lv = 0.f, le = 0.f;
  if (live && lane == 0) { a.dvalues[smp] = h.dbaseline; lp = h.lp; lv = h.lv; le = h.le; }
  block_sum3(lp, lv, le, sh3);
  if (threadIdx.x == 0) {
    if (a.loss_partials) {
      float* o = a.loss_partials + 4 * blockIdx.x;
      o[0] = lp + lv + le; o[1] = lp; o[2] = lv; o[3] = le;
    } else {
      atomicAdd(&a.losses[0], lp + lv + le); atomicAdd(&a.losses[1], lp); atomicAdd(&a.losses[2], lv); atomicAdd(&a.losses[3], le);
    }
  }
}

// Rollout side of the same distribution: raw = loc + scale z, action = tanh(raw), log_prob = sum_j log N(raw; loc, scale) -
// log |d tanh / d raw| (brax NormalTanhDistribution; scale = softplus(raw_scale) + 0.001).  One 16-lane row per sample.
__global__ void policy_sample_kernel(const float* __restrict__ logits, const float* __restrict__ noise, float* __restrict__ raw_out,
                                     float* __restrict__ action_out, float* __restrict__ logp_out, int n, int A) {
  const int lane = threadIdx.x & 15;
  const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = s < n, on = live && lane < A;
  const int sc = live ? s : n - 1;
  const float HALF_LOG_2PI = 0.91893853320467274f, LOG2 = 0.69314718055994531f;
  float loc = 0.f, rs = 0.f, z = 0.f;
  if (on) { loc = logits[(size_t)sc * 2 * A + lane]; rs = logits[(size_t)sc * 2 * A + A + lane]; z = noise[(size_t)sc * A + lane]; }
  const float scale = softplus(rs) + 0.001f;
  const float raw = loc + scale * z;
  const float u = (raw - loc) / scale;
  const float ldj = 2.0f * (LOG2 - raw - softplus(-2.0f * raw));
  const float lp_j = on ? (-0.5f * u * u - __logf(scale) - HALF_LOG_2PI - ldj) : 0.0f;
  const float logp = row16_sum(lp_j);
  if (on) { raw_out[(size_t)s * A + lane] = raw; action_out[(size_t)s * A + lane] = tanhf(raw); }
  if (live && lane == 0) logp_out[s] = logp;
}

// Deterministic global gradient norm (data-parallel replicas must apply bit-identical updates, so no float atomics):
// block b writes its partial sum to acc[2 + b]; one block then folds the partials in a fixed order into acc[0] and
// advances the step counter acc[1].
__global__ void sqnorm_kernel(const float* __restrict__ g, float* __restrict__ acc, int64_t n) {
  __shared__ float sh[16];
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const float v = g[i]; s += v * v; }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) { acc[2 + blockIdx.x] = s; if (blockIdx.x == 0) acc[1] += 1.0f; }   // block 0 also advances the step counter
}

// optax.chain(clip_by_global_norm(max_norm), adam(lr)): g *= max_norm / norm when norm >= max_norm;
// m, v moments with bias correction 1 - b^t, p -= lr * mhat / (sqrt(vhat) + eps).
// Every block folds the per-block partial sums of the squared norm itself, in the same fixed order (so every block -- and
// every data-parallel replica -- gets the same bits): one launch less than a separate final-reduction kernel.  Block 0
// leaves the total in acc[0] for the caller.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ acc, int nblocks, int64_t n, float lr, float b1, float b2, float eps, float max_norm) {
  __shared__ float sh[16];
  float sq = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) sq += acc[2 + i];
  sq = block_sum(sq, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) acc[0] = sq;
  const float norm = sqrtf(sq), t = acc[1];
  const float clip = (max_norm > 0.0f && !(norm < max_norm)) ? max_norm / norm : 1.0f;
  const float c1 = 1.0f / (1.0f - powf(b1, t)), c2 = 1.0f / (1.0f - powf(b2, t));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr * (mi * c1) / (sqrtf(vi * c2) + eps);
  }
}

// dz = dh * silu'(z) and per-tile column sums of dz in one pass (bias gradient).  Tile = 64 rows x 64 columns, 256
// threads = 64 columns x 4 row phases (coalesced 256 B rows); partial[tile_row][c]; colsum_final_kernel folds the
// partials in a fixed order (again 64 columns x 4 phases per workgroup).
constexpr int SB_ROWS = 64;
__global__ void silu_bwd_colsum_kernel(const float* __restrict__ dh, const float* __restrict__ z, float* __restrict__ dz,
                                       float* __restrict__ partial, int n, int w) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * SB_ROWS, r1 = min(n, r0 + SB_ROWS);
  float s = 0.0f;
  if (c < w) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 4) {
      const size_t o = (size_t)r * w + c;
      const float x = z[o], sg = 1.0f / (1.0f + __expf(-x));
      const float g = dh[o] * sg * (1.0f + x * (1.0f - sg));   // d/dx x sigmoid(x)
      dz[o] = g;
      s += g;
    }
  }
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < w) partial[(size_t)blockIdx.y * w + c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}
// Tile sums of a plain [n, w] matrix in the same layout (the top layer's bias gradient: dz needs no activation derivative)
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int n, int w) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * SB_ROWS, r1 = min(n, r0 + SB_ROWS);
  float s = 0.0f;
  if (c < w)
    for (int r = r0 + ty; r < r1; r += 4) s += x[(size_t)r * w + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < w) partial[(size_t)blockIdx.y * w + c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int w) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.0f;
  if (c < w)
    for (int b = ty; b < nblk; b += 4) s += partial[(size_t)b * w + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < w) out[c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}

// The same fold for up to 8 layers in one launch (blockIdx.y = layer): the bias gradients are only needed by the clip + Adam
// step, so the per-layer finalisations leave the dz -> dh -> dz chain of the backward pass.
struct ColsumArgs { const float* partial[8]; float* out[8]; int w[8]; };
__global__ void colsum_final_multi_kernel(ColsumArgs a, int nblk) {
  __shared__ float sh[4][64];
  const int f = blockIdx.y, w = a.w[f];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const float* partial = a.partial[f];
  float s = 0.0f;
  if (c < w)
    for (int b = ty; b < nblk; b += 4) s += partial[(size_t)b * w + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < w) a.out[f][c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}

// Minibatch gather: row idx[b] (or, for a "direct" field, row base + b) of up to 10 row-major [N, row_floats] sources -> row b
// of the matching static buffers, one launch: blockIdx.x = destination row, .y = field, .z = 1024-float chunk of the row
// (16-byte copies when the row length is a multiple of 4: every field of the PPO minibatch is).
constexpr int GATHER_MAX = 10;
struct GatherArgs { const float* src[GATHER_MAX]; float* dst[GATHER_MAX]; int row[GATHER_MAX]; long long base[GATHER_MAX]; int direct[GATHER_MAX]; int nfields; };
__global__ void gather_rows_kernel(GatherArgs a, const long long* __restrict__ idx, long long src_rows) {
  const int f = blockIdx.y;
  const int n = a.row[f];
  const int c0 = blockIdx.z * 1024;
  if (c0 >= n) return;
  const int c1 = min(n, c0 + 1024);
  const bool direct = a.direct[f] != 0;
  const long long j = direct ? a.base[f] + blockIdx.x : idx[blockIdx.x];
  float* d = a.dst[f] + (size_t)blockIdx.x * n;
  if (!direct && (j < 0 || j >= src_rows)) {   // never read out of bounds: the row becomes NaN and the step's losses say so
    for (int i = c0 + threadIdx.x; i < c1; i += blockDim.x) d[i] = __builtin_nanf("");
    return;
  }
  const float* s = a.src[f] + (size_t)j * n;
  if ((n & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s); float4* d4 = reinterpret_cast<float4*>(d);
    for (int i = (c0 >> 2) + threadIdx.x; i < (c1 >> 2); i += blockDim.x) d4[i] = s4[i];
  } else {
    for (int i = c0 + threadIdx.x; i < c1; i += blockDim.x) d[i] = s[i];
  }
}

// ---- weight gradients of an MLP: dW_l = dz_l^T h_{l-1} for up to 8 layers (two networks) in ONE launch, on the f32 matrix cores.
// These are the learner's worst-shaped GEMMs (K = 5 120 minibatch rows deep, outputs as small as 28 x 128): the library runs
// them at ~30 TFLOP/s.  Both operands arrive in the quad-row layout the fused network kernels write (csrc/odk_mlp.hip:
// [rows / 4][width][4], four consecutive rows of one column = 16 bytes), so that a lane's piece of four reduction indices is ONE
// global_load_dwordx4 and the wave reads two contiguous 512-byte runs per operand block: lane l (r = l / 32, c = l % 32) of
// row group G takes rows 8 G + 4 r + {0..3} of column i0 + c (dz) / j0 + c (h), and the group's MFMAs j = 0..3 use component j
// of both (v_mfma_f32_32x32x2_f32: A[m = c][k = r], B[k = r][n = c]).  No LDS, no transposes, 4 loads per 16 MFMAs (with one
// 4-byte load per lane and MFMA -- the first version -- the CU's address unit was the limit: ~30 % of the matrix pipe).
// One wave = one 128 x 64 output tile (4 x 2 MFMA blocks, 128 accumulator registers: 6 operand loads per 32 MFMAs) over one slice
// of the rows (split-K);
// partial tiles go to a workspace laid out like the flat gradient buffer, and dw_reduce_kernel folds the slices in a fixed
// order (data-parallel replicas must stay bit-identical: no float atomics).  Block b runs on XCD b % 8: the blocks of an XCD
// share the same row slices, so each XCD's L2 reads its part of dz / h once and serves all tiles from it.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DW_MAX = 8;
#ifndef DW_AHEAD
#define DW_AHEAD 2
#endif
struct DwLayer { const float* dz; const float* h; int n_out, n_in, tj, tile0, ngroups; long long out_off; };   // ngroups = rows / 8
// Schedule (round 4).  A wave alone on its SIMD runs a full tile slice in ~35 us; the first version launched ntiles x kslices = 68 x 16
// = 1 088 single-wave workgroups on the chip's 1 024 SIMDs, so 64 SIMDs carried two and the launch lasted two wave-times (69 us,
// matrix pipe 40 % busy).  Now the host packs the (tile, slice) items of ONE XCD -- every tile x the kslices / 8 slices that XCD owns,
// weighted by the MFMA blocks the tile really has (edge tiles skip their missing blocks) x the slice's row groups -- into at most
// DW_SLOTS = SIMDs per XCD wave slots (longest item first, always onto the least-loaded slot), block b = slot (b >> 3) of XCD (b & 7)
// works its slot's items one after the other, and a dynamic-LDS reservation of a quarter CU per workgroup keeps the dispatcher
// from stacking more than four waves on a CU.  For the reference networks: 136 items -> 128 slots, 120 of them one full tile
// slice, 8 a half tile + a quarter tile; the longest slot is one full slice.
// Two waves per workgroup (round 4, second step): the two halves of a workspace slice's rows go to the two waves of ONE workgroup,
// wave 1 hands its accumulators over through LDS and wave 0 adds and stores -- half the split-K workspace (8 slices instead of 16:
// the finishing launch reads 16 MB instead of 32) for one 32 KB LDS round trip per tile.  A slot is now a workgroup slot: 64 per XCD.
constexpr int DW_SLOTS = 64, DW_SLOT_ITEMS = 4, DW_WAVES = 2;
struct DwArgs { DwLayer L[DW_MAX]; int nlayers, ntiles, kslices, nslots; float* ws; long long ws_stride; long long* prof; unsigned short item[DW_SLOTS][DW_SLOT_ITEMS]; };

#ifndef DW_MI
#define DW_MI 4      // output tile of a wave: DW_MI x DW_MJ blocks of 32 x 32 (rows = n_out side, columns = n_in side)
#define DW_MJ 2
#endif
// One (tile, slice) item with NI x NJ blocks of 32 x 32 that exist (compile time: an edge tile of the matrix runs its own
// instantiation -- loads and MFMAs for its blocks only; as one runtime-guarded loop the edge tiles ran 2.5x longer per MFMA than
// full ones and their slots were the launch's tail).
template <int NI, int NJ>
__device__ __forceinline__ void dw_tile(const DwArgs& a, const DwLayer& Ly, const int i0, const int j0, const int slice, float* fold) {
  const int n_out = Ly.n_out, n_in = Ly.n_in;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 5, c = lane & 31;
  const int s_lo = (int)((long long)slice * Ly.ngroups / a.kslices), s_hi = (int)((long long)(slice + 1) * Ly.ngroups / a.kslices);
  const int s_mid = s_lo + (s_hi - s_lo + 1) / 2;
  const int g_lo = wv ? s_mid : s_lo, g_hi = wv ? s_hi : s_mid;     // this wave's half of the slice
  // Out-of-range columns read column 0 of the tile (always valid) and are NOT zeroed: entry (i, j) depends on column i of dz and
  // column j of h only, so whatever the clamped loads bring into the padding never reaches a stored element.
  const f32x4* pa[NI]; const f32x4* pb[NJ];
  bool ma[NI], mb[NJ];
#pragma unroll
  for (int k = 0; k < NI; k++) { ma[k] = i0 + 32 * k + c < n_out; pa[k] = reinterpret_cast<const f32x4*>(Ly.dz) + (size_t)r * n_out + (ma[k] ? i0 + 32 * k + c : i0); }
#pragma unroll
  for (int k = 0; k < NJ; k++) { mb[k] = j0 + 32 * k + c < n_in; pb[k] = reinterpret_cast<const f32x4*>(Ly.h) + (size_t)r * n_in + (mb[k] ? j0 + 32 * k + c : j0); }
  const size_t sa = (size_t)2 * n_out, sb = (size_t)2 * n_in;   // one row group = two row quads
  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[i][j][v] = 0.0f;
  struct Ops { f32x4 a[NI], b[NJ]; };
  auto fetch = [&](int g, Ops& x) {
    const size_t o = (size_t)(g < g_hi ? g : g_lo);   // past the slice: re-read its first group (in bounds, unused)
#pragma unroll
    for (int k = 0; k < NI; k++) x.a[k] = pa[k][o * sa];
#pragma unroll
    for (int k = 0; k < NJ; k++) x.b[k] = pb[k][o * sb];
  };
  // the loads of the next AH groups are in flight under a group's MFMAs (a ring of register sets, the trip unrolled so that the
  // sets swap roles without copies); sched_barrier keeps the scheduler from sinking the loads down to their MFMAs.  Small tiles
  // have fewer MFMAs per group to cover a load's latency with: their ring is deeper.
  constexpr int AH = NI * NJ >= 8 ? DW_AHEAD : (NI * NJ >= 4 ? 2 * DW_AHEAD : 4 * DW_AHEAD);
  Ops ring[AH + 1];
#pragma unroll
  for (int k = 0; k < AH; k++) fetch(g_lo + k, ring[k]);
  for (int g = g_lo; g < g_hi; g += AH + 1) {
#pragma unroll
    for (int k = 0; k <= AH; k++) {
      fetch(g + k + AH, ring[(k + AH) % (AH + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      if (g + k < g_hi) {
        const Ops& x = ring[k];
#pragma unroll
        for (int j4 = 0; j4 < 4; j4++)
#pragma unroll
          for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.a[i][j4], x.b[j][j4], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the two halves of the slice: wave 1's accumulators through LDS ([register][lane]: conflict-free), wave 0 adds (wave 0 + wave 1:
  // a fixed order) and stores
  f32x4* fold4 = reinterpret_cast<f32x4*>(fold);      // [block][quarter][lane]: 16-byte pieces, lane-contiguous (ds_write_b128 / ds_read_b128)
  if (wv == 1) {
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++)
          fold4[((i * NJ + j) * 4 + q4) * 64 + lane] = f32x4{acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
  }
  __syncthreads();
  if (wv == 0) {
    // C/D fragment: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* w = a.ws + (size_t)slice * a.ws_stride + Ly.out_off;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        f32x4 other[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) other[j] = fold4[((i * NJ + j) * 4 + q4) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int v = 4 * q4 + t, row = i0 + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * r;
#pragma unroll
          for (int j = 0; j < NJ; j++)
            if (row < n_out && mb[j]) w[(size_t)row * n_in + j0 + 32 * j + c] = acc[i][j][v] + other[j][t];
        }
      }
  }
  __syncthreads();     // the fold buffer is free again (the slot's next item)
}
__device__ __forceinline__ void dw_item(const DwArgs& a, const int tile, const int slice, float* fold) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < DW_MAX; k++) if (k < a.nlayers && tile >= a.L[k].tile0) l = k;
  const DwLayer& Ly = a.L[l];
  const int t = tile - Ly.tile0, i0 = (t / Ly.tj) * (32 * DW_MI), j0 = (t % Ly.tj) * (32 * DW_MJ);
  const int ni = min(DW_MI, (Ly.n_out - i0 + 31) / 32), nj = min(DW_MJ, (Ly.n_in - j0 + 31) / 32);   // wave-uniform: blocks of an edge tile that exist
  static_assert(DW_MI == 4 && DW_MJ == 2, "the dispatch below lists the shapes of a 4 x 2 tile");
  switch (ni * 2 + nj - 3) {     // (ni, nj) -> 0 .. 7
    case 7: dw_tile<4, 2>(a, Ly, i0, j0, slice, fold); break;
    case 6: dw_tile<4, 1>(a, Ly, i0, j0, slice, fold); break;
    case 5: dw_tile<3, 2>(a, Ly, i0, j0, slice, fold); break;
    case 4: dw_tile<3, 1>(a, Ly, i0, j0, slice, fold); break;
    case 3: dw_tile<2, 2>(a, Ly, i0, j0, slice, fold); break;
    case 2: dw_tile<2, 1>(a, Ly, i0, j0, slice, fold); break;
    case 1: dw_tile<1, 2>(a, Ly, i0, j0, slice, fold); break;
    default: dw_tile<1, 1>(a, Ly, i0, j0, slice, fold); break;
  }
}
__global__ void __launch_bounds__(64 * DW_WAVES) dw_gemm_kernel(DwArgs a) {
  extern __shared__ float dw_fold[];     // [DW_MI * DW_MJ * 16][64] floats = 32 KB (the launch reserves half a CU's LDS per workgroup)
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  const int per_xcd = a.kslices >> 3;                 // kslices is a multiple of 8
  long long t0 = 0, c0 = 0;
  if (a.prof) { t0 = wall_clock64(); c0 = clock64(); }
#pragma unroll 1
  for (int k = 0; k < DW_SLOT_ITEMS; k++) {
    const int it = a.item[slot][k];
    if (it == 0xFFFF) break;
    dw_item(a, it / per_xcd, xcd * per_xcd + it % per_xcd, dw_fold);
  }
  if (a.prof && threadIdx.x == 0) {   // tools/gpu_dw_profile.py: when and where each wave ran (100 MHz wall clock; HW_ID, XCC_ID)
    a.prof[4 * b] = t0; a.prof[4 * b + 1] = wall_clock64();
    a.prof[4 * b + 2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
    a.prof[4 * b + 3] = clock64() - c0;      // shader-clock cycles of the wave: / its wall time = the clock the SIMD really ran at
  }
}
// out[off + e] = sum over the row slices, in slice order, for the weight ranges of the layers (e < count); four consecutive
// elements per thread (offsets, counts and the slice stride are multiples of 4: checked by the host).  The same launch can
// finish the bias gradients (blocks past the first nb_reduce: 16 columns x 16 row phases each fold the per-tile column sums
// of one layer, fixed order) and leave per-block partial sums of the squared gradient norm for the clip + Adam launch
// (sq != null; block 0 then also advances the step counter) -- two launches less per SGD step.
struct DwRanges { long long off[DW_MAX], count[DW_MAX]; int n; };
struct FinishArgs { const float* partial[8]; float* out[8]; int w[8], nblk[8], blk0[8]; int nbias, nb_reduce; float* sq; float* counter; };
__global__ void __launch_bounds__(256) grad_finish_kernel(const float* __restrict__ ws, long long ws_stride, int kslices, DwRanges rg, float* __restrict__ out,
                                                         FinishArgs fa) {
  __shared__ float sh[16][17];
  __shared__ float red[16];
  float sq = 0.0f;
  if ((int)blockIdx.x < fa.nb_reduce) {
    long long total = 0;
    for (int k = 0; k < rg.n; k++) total += rg.count[k];
    for (long long i = 4 * ((long long)blockIdx.x * blockDim.x + threadIdx.x); i < total; i += 4 * (long long)fa.nb_reduce * blockDim.x) {
      long long e = i, off = rg.off[0];
#pragma unroll
      for (int k = 0; k < DW_MAX - 1; k++) if (k + 1 < rg.n && e >= rg.count[k]) { e -= rg.count[k]; off = rg.off[k + 1]; } else break;
      // all slices' pieces first, then the sum in slice order: as a loop of load -> add the launch was a chain of kslices exposed
      // memory round trips per thread (9.6 us whether it folded 16 slices or 8)
      constexpr int KMAX = 16;
      float4 v[KMAX];
      const float* src = ws + off + e;
#pragma unroll
      for (int sl = 0; sl < KMAX; sl++) v[sl] = sl < kslices ? *reinterpret_cast<const float4*>(src + (size_t)sl * ws_stride) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4 s = v[0];
#pragma unroll
      for (int sl = 1; sl < KMAX; sl++) if (sl < kslices) { s.x += v[sl].x; s.y += v[sl].y; s.z += v[sl].z; s.w += v[sl].w; }
      for (int sl = KMAX; sl < kslices; sl++) {     // (more slices than the batch holds: the tail one by one)
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)sl * ws_stride);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      *reinterpret_cast<float4*>(out + off + e) = s;
      sq += (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
    }
  } else {
    const int fb = (int)blockIdx.x - fa.nb_reduce;
    int f = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) if (k < fa.nbias && fb >= fa.blk0[k]) f = k;
    const int w = fa.w[f], nblk = fa.nblk[f];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = (fb - fa.blk0[f]) * 16 + tx;
    const float* partial = fa.partial[f];
    float s = 0.0f;
    if (c < w) {     // (eight loads in flight, summed in the same ascending order as one by one)
      int bk = ty;
      for (; bk + 16 * 7 < nblk; bk += 16 * 8) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v8[u] = partial[(size_t)(bk + 16 * u) * w + c];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v8[u];
      }
      for (; bk < nblk; bk += 16) s += partial[(size_t)bk * w + c];
    }
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < w) {
      float t = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; k++) t += sh[k][tx];
      fa.out[f][c] = t;
      sq = t * t;
    }
  }
  if (fa.sq) {   // (uniform)
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) { fa.sq[blockIdx.x] = sq; if (blockIdx.x == 0 && fa.counter) *fa.counter += 1.0f; }
  }
}

// column sums and sums of squares of x [rows, w], float64 accumulators: block = 64 columns x 4 row phases (256-byte coalesced row reads), grid =
// (column tiles, row slices); slice s walks rows s, s + slices, ... -- with 4 phases: rows s + slices (4 k + phase)
__global__ void __launch_bounds__(256) col_moments_kernel(const float* __restrict__ x, long long rows, int w, int slices, double* __restrict__ partial) {
  __shared__ double sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx, sl = blockIdx.y;
  double s = 0.0, s2 = 0.0;
  if (c < w) {
    long long r = (long long)sl + (long long)slices * ty;
    const long long step = 4ll * slices;
    // eight rows in flight per thread (the launch is a latency-bound stream: bytes in flight per CU are what sets its rate)
    for (; r + 7 * step < rows; r += 8 * step) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = x[(r + u * step) * w + c];
#pragma unroll
      for (int u = 0; u < 8; u++) { s += (double)v[u]; s2 += (double)v[u] * (double)v[u]; }
    }
    for (; r < rows; r += step) { const float v = x[r * w + c]; s += (double)v; s2 += (double)v * (double)v; }
  }
  sh[0][ty][tx] = s; sh[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < w) {
    double* o = partial + (size_t)sl * 2 * w;
    o[c] = (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]);
    o[w + c] = (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]);
  }
}

// running_statistics.update from the slice moments: block = 64 columns x 4 slice phases; every thread reads the OLD count before the barrier and the
// one thread that owns it writes the new one after it (single workgroup per 64 columns: blocks other than 0 never write the count)
__global__ void __launch_bounds__(256) moments_update_kernel(const double* __restrict__ partial, int slices, int w, double rows, double* __restrict__ count,
                                                            float* __restrict__ mean, float* __restrict__ sv, float* __restrict__ sd, float std_min, float std_max) {
  __shared__ double sh[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const double cnt0 = *count;
  double s = 0.0, s2 = 0.0;
  if (c < w)
    for (int sl = ty; sl < slices; sl += 4) { const double* p = partial + (size_t)sl * 2 * w; s += p[c]; s2 += p[w + c]; }
  sh[0][ty][tx] = s; sh[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < w) {
    s = (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]);
    s2 = (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]);
    const double cnt = cnt0 + rows, m0 = (double)mean[c];
    const double m1 = m0 + (s / rows - m0) * (rows / cnt);
    const double v1 = (double)sv[c] + (s2 - s * (m0 + m1) + rows * m0 * m1);
    mean[c] = (float)m1; sv[c] = (float)v1;
    const float sdv = (float)sqrt(fmax(v1 / cnt, 0.0));
    sd[c] = fminf(fmaxf(sdv, std_min), std_max);
  }
  // the count moves once every reader has it (grid-wide: by the NEXT launch on the stream -- see the host wrapper: its own tiny launch)
}
__global__ void add_count_kernel(double* count, double rows) { *count += rows; }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return odk_fail_(ODK_ERR_HIP, what);
  return ODK_OK;
}

}  // namespace

extern "C" int odk_moments_update(const double* partial_dev, int slices, int w, long long rows, double* count_dev, float* mean_dev, float* summed_variance_dev,
                                  float* std_dev, float std_min, float std_max, void* stream) {
  if (!partial_dev || !count_dev || !mean_dev || !summed_variance_dev || !std_dev || slices <= 0 || w <= 0 || rows <= 0)
    return odk_fail_(ODK_ERR_INVALID, "odk_moments_update: bad arguments");
  hipLaunchKernelGGL(moments_update_kernel, dim3((w + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial_dev, slices, w, (double)rows, count_dev, mean_dev,
                     summed_variance_dev, std_dev, std_min, std_max);
  hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, count_dev, (double)rows);      // after every block of the update has read the old count
  return check_launch("odk_moments_update: launch failed");
}

extern "C" int odk_col_moments(const float* x_dev, long long rows, int w, int slices, double* partial_dev, void* stream) {
  if (!x_dev || !partial_dev || rows <= 0 || w <= 0 || slices <= 0 || slices > 1024) return odk_fail_(ODK_ERR_INVALID, "odk_col_moments: bad arguments (slices 1..1024)");
  hipLaunchKernelGGL(col_moments_kernel, dim3((w + 63) / 64, slices), dim3(256), 0, (hipStream_t)stream, x_dev, rows, w, slices, partial_dev);
  return check_launch("odk_col_moments: launch failed");
}

extern "C" int odk_gae(const float* truncation_dev, const float* termination_dev, const float* rewards_dev, const float* values_dev,
                       const float* bootstrap_dev, float* vs_dev, float* adv_dev, float* adv_stats_dev, int B, int T, float lambda_,
                       float discount, void* stream) {
  if (!truncation_dev || !termination_dev || !rewards_dev || !values_dev || !bootstrap_dev || !vs_dev || !adv_dev || B <= 0 || T <= 0)
    return odk_fail_(ODK_ERR_INVALID, "odk_gae: bad arguments");
  const int threads = B >= 1024 ? 1024 : ((B + 63) / 64) * 64;
  if ((long long)B * T <= GAE_LDS_N)   // 1024 threads: the staging loops are latency-bound (5 trips of 4 loads instead of 20), the recursion uses B of them
    hipLaunchKernelGGL(gae_kernel_lds, dim3(1), dim3(1024), 0, (hipStream_t)stream, truncation_dev, termination_dev, rewards_dev,
                       values_dev, bootstrap_dev, vs_dev, adv_dev, adv_stats_dev, B, T, lambda_, discount);
  else if (B <= 1024 && T <= 32)
    hipLaunchKernelGGL(gae_kernel_reg<32>, dim3(1), dim3(threads), 0, (hipStream_t)stream, truncation_dev, termination_dev, rewards_dev,
                       values_dev, bootstrap_dev, vs_dev, adv_dev, adv_stats_dev, B, T, lambda_, discount);
  else
    hipLaunchKernelGGL(gae_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, truncation_dev, termination_dev, rewards_dev, values_dev,
                       bootstrap_dev, vs_dev, adv_dev, adv_stats_dev, B, T, lambda_, discount);
  return check_launch("odk_gae: launch failed");
}

extern "C" int odk_ppo_head(const float* logits_dev, const float* raw_action_dev, const float* old_log_prob_dev, const float* adv_dev,
                            const float* adv_stats_dev, const float* vs_dev, const float* baseline_dev, const float* noise_dev,
                            float* dlogits_dev, float* dbaseline_dev, float* losses_dev, int n, int action_size, float clipping_epsilon,
                            float entropy_cost, float grad_scale, void* stream) {
  if (!logits_dev || !raw_action_dev || !old_log_prob_dev || !adv_dev || !vs_dev || !baseline_dev || !noise_dev || !dlogits_dev ||
      !dbaseline_dev || !losses_dev || n <= 0 || action_size <= 0 || action_size > 16)
    return odk_fail_(ODK_ERR_INVALID, "odk_ppo_head: bad arguments (action_size must be 1..16)");
  const int threads = 1024, per_block = threads / 16;   // 64 samples per workgroup: 4 loss atomics per 64 samples
  hipLaunchKernelGGL(ppo_head_kernel, dim3((n + per_block - 1) / per_block), dim3(threads), 0, (hipStream_t)stream, logits_dev, raw_action_dev,
                     old_log_prob_dev, adv_dev, adv_stats_dev, vs_dev, baseline_dev, noise_dev, dlogits_dev, dbaseline_dev, losses_dev, n,
                     action_size, clipping_epsilon, entropy_cost, grad_scale);
  return check_launch("odk_ppo_head: launch failed");
}

extern "C" int odk_ppo_gae_head(const odk_gae_head_args* g, void* stream) {
  if (!g || !g->logits || !g->values || !g->raw_action || !g->old_log_prob || !g->reward || !g->termination || !g->truncation || !g->noise || !g->row_idx ||
      !g->dlogits || !g->dvalues || (!g->losses && !g->loss_partials) || g->B <= 0 || g->T <= 0 || g->n_traj <= 0 || g->action_size <= 0 || g->action_size > 16)
    return odk_fail_(ODK_ERR_INVALID, "odk_ppo_gae_head: bad arguments (action_size must be 1..16)");
  if ((long long)g->B * g->T > GAE_LDS_N || g->B > 1024)
    return odk_fail_(ODK_ERR_INVALID, "odk_ppo_gae_head: B * T <= 5120 and B <= 1024 (larger minibatches: odk_gather_rows + odk_gae + odk_ppo_head)");
  GaeHead a;
  a.logits = g->logits; a.values = g->values; a.raw_action = g->raw_action; a.old_logp = g->old_log_prob; a.reward = g->reward; a.term = g->termination;
  a.trunc = g->truncation; a.noise = g->noise; a.idx = g->row_idx; a.cursor = g->cursor; a.dlogits = g->dlogits; a.dvalues = g->dvalues; a.losses = g->losses; a.loss_partials = g->loss_partials;
  a.adv_out = g->adv_out; a.vs_out = g->vs_out; a.stats_out = g->stats_out; a.B = g->B; a.T = g->T; a.A = g->action_size; a.n_traj = g->n_traj;
  a.normalize = g->normalize_advantage; a.lambda_ = g->gae_lambda; a.discount = g->discount; a.eps = g->clipping_epsilon; a.entropy_cost = g->entropy_cost;
  a.grad_scale = g->grad_scale;
  const int n = g->B * g->T;
  hipLaunchKernelGGL(ppo_gae_head_kernel, dim3((n + ODK_GAE_HEAD_SAMPLES - 1) / ODK_GAE_HEAD_SAMPLES), dim3(GH_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("odk_ppo_gae_head: launch failed");
}

extern "C" int odk_policy_sample(const float* logits_dev, const float* noise_dev, float* raw_action_dev, float* action_dev, float* log_prob_dev,
                                 int n, int action_size, void* stream) {
  if (!logits_dev || !noise_dev || !raw_action_dev || !action_dev || !log_prob_dev || n <= 0 || action_size <= 0 || action_size > 16)
    return odk_fail_(ODK_ERR_INVALID, "odk_policy_sample: bad arguments (action_size must be 1..16)");
  const int threads = 256, per_block = threads / 16;
  hipLaunchKernelGGL(policy_sample_kernel, dim3((n + per_block - 1) / per_block), dim3(threads), 0, (hipStream_t)stream, logits_dev, noise_dev,
                     raw_action_dev, action_dev, log_prob_dev, n, action_size);
  return check_launch("odk_policy_sample: launch failed");
}

extern "C" int odk_adam_clip(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr,
                             float b1, float b2, float eps, float max_grad_norm, void* stream) {
  if (!params_dev || !grads_dev || !m_dev || !v_dev || !acc_dev || n <= 0) return odk_fail_(ODK_ERR_INVALID, "odk_adam_clip: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // kernels only (no hipMemsetAsync): a memset NODE in the middle of a captured graph raced with its neighbours on
  // ROCm 7.2 (intermittent non-finite updates, 3 of 12 runs; 0 of 36 without it)
  const int threads = 256;
  int blocks = (int)((n + threads * 4 - 1) / (threads * 4));
  if (blocks > ODK_ADAM_MAX_PARTIALS) blocks = ODK_ADAM_MAX_PARTIALS;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(threads), 0, st, grads_dev, acc_dev, (int64_t)n);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(threads), 0, st, params_dev, grads_dev, m_dev, v_dev, acc_dev, blocks, (int64_t)n, lr, b1, b2, eps,
                     max_grad_norm);
  return check_launch("odk_adam_clip: launch failed");
}

extern "C" int odk_silu_bwd_colsum(const float* dh_dev, const float* z_dev, float* dz_dev, float* colsum_dev, float* partial_dev, int n,
                                   int w, void* stream) {
  if (!dh_dev || !z_dev || !dz_dev || !partial_dev || n <= 0 || w <= 0) return odk_fail_(ODK_ERR_INVALID, "odk_silu_bwd_colsum: bad arguments");
  const int nblk = (n + SB_ROWS - 1) / SB_ROWS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(silu_bwd_colsum_kernel, dim3((w + 63) / 64, nblk), dim3(256), 0, st, dh_dev, z_dev, dz_dev, partial_dev, n, w);
  if (colsum_dev)   // null: the caller folds the partials later with odk_colsum_finalize
    hipLaunchKernelGGL(colsum_final_kernel, dim3((w + 63) / 64), dim3(256), 0, st, partial_dev, colsum_dev, nblk, w);
  return check_launch("odk_silu_bwd_colsum: launch failed");
}

extern "C" int odk_colsum_partial(const float* x_dev, float* partial_dev, int n, int w, void* stream) {
  if (!x_dev || !partial_dev || n <= 0 || w <= 0) return odk_fail_(ODK_ERR_INVALID, "odk_colsum_partial: bad arguments");
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((w + 63) / 64, (n + SB_ROWS - 1) / SB_ROWS), dim3(256), 0, (hipStream_t)stream, x_dev, partial_dev, n, w);
  return check_launch("odk_colsum_partial: launch failed");
}

extern "C" int odk_colsum_finalize(const float* const* partial_dev, float* const* colsum_dev, const int* widths, int count, int n, void* stream) {
  if (!partial_dev || !colsum_dev || !widths || count <= 0 || count > 8 || n <= 0) return odk_fail_(ODK_ERR_INVALID, "odk_colsum_finalize: bad arguments");
  ColsumArgs a;
  int wmax = 0;
  for (int f = 0; f < 8; f++) {
    a.partial[f] = f < count ? partial_dev[f] : nullptr; a.out[f] = f < count ? colsum_dev[f] : nullptr; a.w[f] = f < count ? widths[f] : 0;
    if (f < count && (!partial_dev[f] || !colsum_dev[f] || widths[f] <= 0)) return odk_fail_(ODK_ERR_INVALID, "odk_colsum_finalize: bad layer arguments");
    if (a.w[f] > wmax) wmax = a.w[f];
  }
  hipLaunchKernelGGL(colsum_final_multi_kernel, dim3((wmax + 63) / 64, count), dim3(256), 0, (hipStream_t)stream, a, (n + SB_ROWS - 1) / SB_ROWS);
  return check_launch("odk_colsum_finalize: launch failed");
}

extern "C" int odk_gather_rows(const float* const* src_dev, float* const* dst_dev, const int* row_floats, const long long* direct_base, int nfields,
                               const long long* idx_dev, int nrows, long long src_rows, void* stream) {
  if (!src_dev || !dst_dev || !row_floats || !idx_dev || nfields <= 0 || nfields > GATHER_MAX || nrows <= 0 || src_rows <= 0)
    return odk_fail_(ODK_ERR_INVALID, "odk_gather_rows: bad arguments");
  GatherArgs a;
  a.nfields = nfields;
  int maxrow = 0;
  for (int f = 0; f < GATHER_MAX; f++) {
    a.src[f] = f < nfields ? src_dev[f] : nullptr; a.dst[f] = f < nfields ? dst_dev[f] : nullptr; a.row[f] = f < nfields ? row_floats[f] : 0;
    a.direct[f] = (f < nfields && direct_base && direct_base[f] >= 0) ? 1 : 0;
    a.base[f] = a.direct[f] ? direct_base[f] : 0;
    if (a.row[f] > maxrow) maxrow = a.row[f];
    if (f < nfields && (a.row[f] & 3) == 0 && ((((uintptr_t)a.src[f]) | ((uintptr_t)a.dst[f])) & 15) != 0) return odk_fail_(ODK_ERR_INVALID, "odk_gather_rows: 16-byte aligned buffers expected");
  }
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows, nfields, (maxrow + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, a, idx_dev, src_rows);
  return check_launch("odk_gather_rows: launch failed");
}

static long long* g_dw_prof = nullptr;
// tools: device buffer of 4 x (blocks of the weight-gradient launch) int64 receiving each wave's start / end (100 MHz clock), HW_ID, XCC_ID
extern "C" void odk_dw_set_profile(long long* dev) { g_dw_prof = dev; }

extern "C" int odk_dw_gemm(const float* const* dz_dev, const float* const* h_dev, const int* n_out, const int* n_in, const long long* out_off,
                           int nlayers, const int* nrows, int kslices, float* ws_dev, long long ws_stride, float* out_dev, odk_grad_finish* finish,
                           void* stream) {
  if (!dz_dev || !h_dev || !n_out || !n_in || !out_off || !ws_dev || !out_dev || nlayers <= 0 || nlayers > DW_MAX || !nrows || kslices <= 0 ||
      (kslices & 7) != 0 || (ws_stride & 3) != 0 || ((uintptr_t)ws_dev & 15) != 0 || ((uintptr_t)out_dev & 15) != 0)
    return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: bad arguments (1..8 layers, kslices a multiple of 8, ws_stride a multiple of 4, 16-byte aligned buffers)");
  DwArgs a;
  DwRanges rg;
  a.prof = g_dw_prof;
  a.nlayers = nlayers; a.kslices = kslices; a.ws = ws_dev; a.ws_stride = ws_stride; a.ntiles = 0; rg.n = nlayers;
  for (int l = 0; l < DW_MAX; l++) {
    DwLayer& L = a.L[l];
    if (l < nlayers) {
      if (!dz_dev[l] || !h_dev[l] || n_out[l] <= 0 || n_in[l] <= 0 || nrows[l] <= 0 || (nrows[l] & 7) != 0 || nrows[l] / 8 < 2 * kslices || out_off[l] < 0 || out_off[l] + (long long)n_out[l] * n_in[l] > ws_stride ||
          (out_off[l] & 3) != 0 || (((long long)n_out[l] * n_in[l]) & 3) != 0 || ((((uintptr_t)dz_dev[l]) | ((uintptr_t)h_dev[l])) & 15) != 0)
        return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: bad layer (rows a multiple of 8 and >= 16 * kslices, offset and element count multiples of 4, "
                                          "16-byte aligned operands)");
      L.dz = dz_dev[l]; L.h = h_dev[l]; L.n_out = n_out[l]; L.n_in = n_in[l]; L.out_off = out_off[l]; L.ngroups = nrows[l] / 8;
      L.tj = (n_in[l] + 32 * DW_MJ - 1) / (32 * DW_MJ); L.tile0 = a.ntiles;
      a.ntiles += ((n_out[l] + 32 * DW_MI - 1) / (32 * DW_MI)) * L.tj;
      rg.off[l] = out_off[l]; rg.count[l] = (long long)n_out[l] * n_in[l];
    } else {
      L.dz = L.h = nullptr; L.n_out = L.n_in = L.tj = L.ngroups = 0; L.tile0 = 1 << 30; L.out_off = 0; rg.off[l] = 0; rg.count[l] = 0;
    }
  }
  // ---- the per-XCD schedule: items (tile, slice-of-this-XCD) by decreasing work onto the least-loaded wave slot
  {
    const int per_xcd = kslices >> 3, nitems = a.ntiles * per_xcd;
    if (nitems >= 0xFFFF) return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: too many tiles");
    std::vector<std::pair<long long, int>> items;      // (work, item id = tile * per_xcd + k)
    for (int l = 0; l < nlayers; l++) {
      const DwLayer& L = a.L[l];
      const int ti = (L.n_out + 32 * DW_MI - 1) / (32 * DW_MI);
      for (int t = 0; t < ti * L.tj; t++) {
        const int i0 = (t / L.tj) * 32 * DW_MI, j0 = (t % L.tj) * 32 * DW_MJ;
        const int ni = std::min(DW_MI, (L.n_out - i0 + 31) / 32), nj = std::min(DW_MJ, (L.n_in - j0 + 31) / 32);
        for (int k = 0; k < per_xcd; k++) {
          // (slices of XCD 0 stand for all: the groups of a slice differ by at most one between XCDs)
          const long long g = (long long)(k + 1) * L.ngroups / kslices - (long long)k * L.ngroups / kslices;
          items.push_back({(long long)ni * nj * g + 8, (L.tile0 + t) * per_xcd + k});      // + 8: a tile's fixed cost (prologue, stores)
        }
      }
    }
    std::stable_sort(items.begin(), items.end(), [](const std::pair<long long, int>& x, const std::pair<long long, int>& y) { return x.first > y.first; });
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    a.nslots = std::max(1, std::min(std::min(DW_SLOTS, nitems), cus / 8 * 4 / DW_WAVES));      // workgroup slots of one XCD = its SIMDs / 2 (MI355X: 32 CUs x 4 / 2)
    long long load[DW_SLOTS] = {0}; int cnt[DW_SLOTS] = {0};
    for (int sidx = 0; sidx < DW_SLOTS; sidx++) for (int k = 0; k < DW_SLOT_ITEMS; k++) a.item[sidx][k] = 0xFFFF;
    for (const auto& itw : items) {
      int best = -1;
      for (int sidx = 0; sidx < a.nslots; sidx++) if (cnt[sidx] < DW_SLOT_ITEMS && (best < 0 || load[sidx] < load[best])) best = sidx;
      if (best < 0) return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: more than DW_SLOTS x DW_SLOT_ITEMS tile slices per XCD");
      a.item[best][cnt[best]++] = (unsigned short)itw.second; load[best] += itw.first;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  // 80 KB of LDS per two-wave workgroup (32 KB used by the fold): at most two of them per CU, i.e. one wave per SIMD when the XCD's
  // slots are all taken
  if (odk_func_lds_attr_((const void*)dw_gemm_kernel, 0, 80 * 1024)) return odk_fail_(ODK_ERR_HIP, "odk_dw_gemm: the device refuses 80 KB of dynamic LDS per workgroup");
  hipLaunchKernelGGL(dw_gemm_kernel, dim3(a.nslots * 8), dim3(64 * DW_WAVES), 80 * 1024, st, a);
  long long total = 0;
  for (int l = 0; l < nlayers; l++) total += rg.count[l];
  int blocks = (int)((total / 4 + 255) / 256);
  FinishArgs fa;
  memset(&fa, 0, sizeof(fa));
  int nfold = 0;
  if (finish) {
    if (finish->nbias < 0 || finish->nbias > 8) return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: finish: at most 8 bias gradients");
    fa.nbias = finish->nbias;
    for (int f = 0; f < 8; f++) {
      const bool on = f < finish->nbias;
      if (on && (!finish->bias_partial[f] || !finish->bias_grad[f] || finish->width[f] <= 0 || finish->nblk[f] <= 0))
        return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: finish: bad bias-gradient entry");
      fa.partial[f] = on ? finish->bias_partial[f] : nullptr; fa.out[f] = on ? finish->bias_grad[f] : nullptr;
      fa.w[f] = on ? finish->width[f] : 0; fa.nblk[f] = on ? finish->nblk[f] : 0; fa.blk0[f] = on ? nfold : (1 << 30);
      if (on) nfold += (finish->width[f] + 15) / 16;
    }
    fa.sq = finish->sq_partials_dev; fa.counter = finish->step_counter_dev;
  }
  const int cap = ODK_ADAM_MAX_PARTIALS - nfold;     // the partial sums of the norm must fit the Adam scratch
  if (blocks > cap) blocks = cap;
  if (blocks < 1) return odk_fail_(ODK_ERR_INVALID, "odk_dw_gemm: finish: too many bias columns");
  fa.nb_reduce = blocks;
  if (finish) finish->nblocks = blocks + nfold;
  hipLaunchKernelGGL(grad_finish_kernel, dim3(blocks + nfold), dim3(256), 0, st, ws_dev, ws_stride, kslices, rg, out_dev, fa);
  return check_launch("odk_dw_gemm: launch failed");
}
