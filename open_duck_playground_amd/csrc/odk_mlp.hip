// odk_mlp.hip -- the policy / value networks of the PPO learner as two launches: every layer of a swish MLP
// (in -> 512 -> 256 -> 128 -> out, brax ppo.networks as configured by reference common/runner.py:86-118) forward in ONE
// kernel, and the whole backward-data chain in ONE kernel, on the f32 matrix cores (v_mfma_f32_16x16x4_f32; the text below tells the story from the
// first, 32-sample / 32x32x2 version on -- the shipped tile is ODK_MLP_TILE = 16 samples, see "16-sample tiles").  gfx950 only.
//
// Why: a minibatch is 5 120 samples and the layers are at most 512 wide, so as separate GEMMs every layer is a ~10 us
// launch that under-fills the chip, with a 5 us element-wise launch (bias + swish, swish' + bias-gradient sums) between two
// of them: ~25 launches per network and minibatch step, 128 steps per training step.  Here a workgroup owns a tile of ODK_MLP_TILE = 16
// samples (32 in the first version) for ALL layers: the tile's activations stay in LDS, the weights stream from the L2 (both networks together are
// 2 MB: resident in every XCD's 4 MB L2), and the epilogues (bias, swish, swish', the bias gradients' tile sums) run on the
// accumulator registers.
//
// Operands.  A wave-wide load instruction costs the address unit ~16 cycles whatever its width, and a 32x32x2 MFMA takes 64
// cycles per SIMD: with one 4-byte load per lane and MFMA the four SIMDs of a CU saturate the address unit (the first version
// of these kernels ran at 25 % of the matrix pipe that way).  So both operands come in 16-byte pieces holding FOUR consecutive
// reduction indices: lane l (r = l / 32, c = l % 32) of k-group G takes k = 8 G + 4 r + {0, 1, 2, 3},
//   A = act[sample c][k .. k + 3]   one ds_read_b128 from the tile in LDS (row pitch = 4 mod 8 floats: conflict-free),
//   B = Wp[k / 4][col0 + c][0 .. 3]  one global_load_dwordx4 from a PACKED copy of the weights, [K / 4][N][4]: the wave reads two
//                                   contiguous 512-byte runs,
// and the group's four MFMAs use component j of both (MFMA j: lanes r = 0 / 1 supply k = 8 G + j and 8 G + 4 + j).  The packed
// copies -- one with the input index as reduction index for the forward pass, one with the output index for the backward
// pass, zero-padded to a multiple of 8 -- live next to the torch-layout parameters and are kept current by the Adam launch
// (odk_adam_clip_packed; odk_pack_weights rebuilds them).
// Accumulator fragment: col = c, row = (v & 3) + 8 (v >> 2) + 4 r for register v.
//
// Forward, per workgroup (4 waves): layer 1 is produced in four 128-column chunks (wave w: one 32 x 32 block per chunk) and
// each chunk is consumed at once as a K-slice of layer 2 (wave w: 64 columns, accumulators live across the chunks), so the
// 512-wide activation never exists in LDS: (first version, 32-sample tiles: X 29 KB + chunk 16.5 KB + layer-2 output 32.5 KB = 78 KB => two
// workgroups per CU; shipped, 16-sample tiles: <= 40 KB, static_assert below => FOUR workgroups per
// CU (all 328 tiles of the two networks resident at once).  Layer 3: one block per wave; output layer (<= 32 columns): K
// split over the four waves, partial blocks folded through LDS.
// Backward: dz_top (from the loss head) -> dh3 = dz_top W4 -> dz3 = dh3 * swish'(z3) -> ... -> dz1, each dz written once to
// global memory (the weight-gradient launch reads it) and kept in LDS as the next layer's A operand; per-tile column sums
// of every dz (the bias gradients' first half, folded by odk_colsum_fold in a fixed order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/odk.h"

int odk_fail_(int code, const char* msg);   // odk_engine.hip

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef ODK_FWD_DBUF
#define ODK_FWD_DBUF 0
#endif
constexpr int H1 = ODK_MLP_H1, H2 = ODK_MLP_H2, H3 = ODK_MLP_H3;
constexpr int TM = ODK_MLP_TILE;       // samples per workgroup
constexpr int CH = 128;                // layer-1 chunk width = K-slice of layer 2
constexpr int KIN_MAX = ODK_MLP_MAX_IN, NOUT_MAX = 32;
constexpr int PC = CH + 4, P2 = H2 + 4, P3 = H3 + 4, P4 = NOUT_MAX + 4;   // LDS row pitches: multiples of 4 (16-byte reads), = 4 mod 8
constexpr int PX_MAX = KIN_MAX + 4;
constexpr int SLACK = 64;              // the k loops fetch (never use) up to two batches past a row's end
static_assert(TM == 16 && KIN_MAX % 16 == 0, "tile = one 16-row MFMA block, padded input width");
// forward LDS (floats): X | chunk | layer-2 output | slack;  the layer-3 output aliases X, the output layer's partial blocks alias the chunk
constexpr int NCBUF = ODK_FWD_DBUF ? 2 : 1;
constexpr int F_X = 0, F_C = F_X + TM * PX_MAX, F_H2 = F_C + NCBUF * TM * PC, F_TOTAL = F_H2 + TM * P2 + SLACK;
static_assert(TM * P3 <= TM * PX_MAX, "layer-3 output must fit in the X region");
static_assert(3 * 2 * 4 * 64 <= TM * PC, "output-layer partial blocks must fit in the chunk region");
static_assert(F_TOTAL * 4 <= (ODK_FWD_DBUF ? 53 : 40) * 1024, "four (three with two chunk buffers) forward workgroups per CU");
// backward LDS: dz_top | dz3 | dz2 | slack
constexpr int B_D4 = 0, B_D3 = B_D4 + TM * P4, B_D2 = B_D3 + TM * P3, B_TOTAL = B_D2 + TM * P2 + SLACK;

__host__ __device__ constexpr int pad16(int k) { return (k + 15) & ~15; }

struct Net {
  const float* x; const float* in_mean; const float* in_std; const float* wf[4]; const float* wb[4]; const float* b[4];
  float* h[3]; float* g[3]; float* out; float* xp;
  const float* dout; float* dz[3]; float* doutp; float* bias_partial[4];
  int n, n_in, n_out, tile0;
  const long long* row_idx; const int* cursor; const float* x_tail; int traj_len, n_main, n_traj;     // odk_mlp_desc: row sources of the forward pass
};
// Block -> (network, tile).  All workgroups of a launch are resident at once (<= 4 per CU) and the dispatcher hands block b to CU
// b mod R (R = CUs; measured: tools/gpu_mlp_wg_profile.py finds exactly the predicted mixes), so with T = 656 tiles on 256 CUs
// 144 CUs run three tiles and 112 two, and a CU's time is the sum of its tiles' matrix work: the launch lasts as long as the
// costliest three-tile mix.  The value network's tiles (wider input) cost 1.24x the policy's; in network order the mixes were
// PPV x 64, PVV x 80, PV x 112.  `Map` deals the costly network's tiles to the two-tile CUs first and spreads the rest one per
// three-tile CU: PPV x 112, PPP x 32, VV x 112 -- the longest CU 7 % shorter.  Only speed depends on the dispatch order.
struct Map { int R, rounds, rem, vin2, pin2, v3, costly, on; };
struct Args { Net net[2]; int nnets; int diag; long long* prof; long long* wgprof; Map map; };
__device__ __forceinline__ void map_block(const Args& a, int b, int& net, int& tile) {
  const Map& m = a.map;
  if (!m.on) { net = (a.nnets > 1 && b >= a.net[1].tile0) ? 1 : 0; tile = b - a.net[net].tile0; return; }
  const int k = b % m.R, r = b / m.R;
  bool costly; int idx;
  if (k >= m.rem) { const int s2 = (k - m.rem) * m.rounds + r; costly = s2 < m.vin2; idx = costly ? s2 : s2 - m.vin2; }
  else { const int c = r * m.rem + k; costly = c < m.v3; idx = costly ? m.vin2 + c : m.pin2 + (c - m.v3); }
  net = costly ? m.costly : 1 - m.costly; tile = idx;
}   // prof: phase timestamps of workgroup 0, wave 0; wgprof: start / end / place of every workgroup (tools only)
#define ODK_WG_BEGIN() long long wg_t0 = 0, wg_c0 = 0; if (a.wgprof) { wg_t0 = wall_clock64(); wg_c0 = clock64(); }
#define ODK_WG_END() do { if (a.wgprof && threadIdx.x == 0) { long long* o = a.wgprof + 4 * blockIdx.x; o[0] = wg_t0; o[1] = wall_clock64(); \
    o[2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32) | ((long long)net_i << 40); \
    o[3] = clock64() - wg_c0; } } while (0)
#define ODK_STAMP(i) do { if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[i] = clock64(); } while (0)

// One GEMM phase of a wave: acc[blk] += A (16 samples x K, LDS) * Bp (packed [K / 4][N][4], columns col0 + 16 blk + c), over ng
// k-groups of 16 (lane (q, c): k = 16 G + 4 q + {0..3}; the group's MFMAs j = 0..3 use component j of both pieces).  U groups
// per batch, register double-buffered: while a batch's MFMAs run (4 U NBLK MFMAs of 32 cycles) the next batch's loads are in
// flight.  The FIRST batch of weights is fetched by prefetch(), which the caller issues before the previous phase's epilogue
// and barrier (weights do not depend on them), so a phase starts with its operands already on the way; A comes from LDS after
// the barrier.  __builtin_amdgcn_sched_barrier pins the order: left alone, the scheduler sinks every load down to its MFMA
// (minimum register pressure = one exposed L2 round trip per MFMA).  Groups >= ng are fetched from group 0 (in bounds) and
// skipped.  NBLK >= 2 everywhere: a 16x16x4 MFMA has 40 cycles of dependent latency for 32 of issue.
#define ODK_PIN() __builtin_amdgcn_sched_barrier(0)
#ifndef ODK_MLP_WG_PER_CU
#define ODK_MLP_WG_PER_CU 4       // resident workgroups per CU the register allocation is held to (4: <= 128 VGPRs, 3: <= 168)
#endif
#ifndef ODK_BWD_PREFETCH_G
#define ODK_BWD_PREFETCH_G 0      // backward: a phase's swish' pieces are fetched one phase ahead instead of behind its MFMAs
#endif
#ifndef ODK_FWD_DBUF
#define ODK_FWD_DBUF 0            // forward: two chunk buffers alternate, one barrier per chunk instead of two (three workgroups per CU)
#endif
#ifndef ODK_MLP_SETPRIO
#define ODK_MLP_SETPRIO 0         // raise the wave's priority inside the MFMA loops
#endif
template <int NBLK, int U, bool RING = false>      // RING: the diagnostic build's forward kernel (weights from LDS: diag bits 5 / 6)
struct Phase {
  f32x4 fb[U][NBLK];
  const f32x4* B; unsigned lane_off, blk_off; int N4, ng;
#ifdef ODK_MLP_DIAG    // diagnostic build (make libodk_mlpdiag.so; tools/gpu_mlp_wg_profile.py): bit 0 = every group re-reads group 0 (L1 hits), bit 1 = no MFMAs,
                       // bit 2 = no operand loads inside the loops, bit 3 = no activation stores, bit 4 = half of the weight loads,
                       // bit 5 (round 6: what would a SHARED WEIGHT RING IN LDS buy at best?) = the weights come from LDS instead of global memory -- every B piece
                       // is one ds_read_b128 from an 8 KB region behind the workgroup's image, with the real layout's bank pattern, no fill and no synchronisation --,
                       // bit 6 = with bit 5, every third k-group is ALSO fetched from global memory and written to that region (a ring filled once per CU by the
                       // three tile groups it serves: each group's share of the fill traffic)
  int diag = 0;
  float* ring = nullptr;
#else
  static constexpr int diag = 0;
#endif
  __device__ __forceinline__ void load_b(int G0, f32x4 (*xb)[NBLK]) const {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int G = G0 + u;
      const f32x4* bp = B + (size_t)((G < ng && !(diag & 1)) ? G : 0) * N4;   // wave-uniform base, 32-bit lane offset (diag 1, tools: every group re-reads group 0 -> L1 hits)
#pragma unroll
      for (int k = 0; k < NBLK; k++) {
        if ((diag & 16) && (k & 1)) xb[u][k] = xb[u][k - 1];       // (diag 16, tools: half of the weight loads: what the launch would cost with twice the reuse per piece)
#ifdef ODK_MLP_DIAG
        else if (RING && (diag & 32)) {
          f32x4* slot = reinterpret_cast<f32x4*>(ring) + ((G & 1) << 8);
          const unsigned pi = (lane_off + blk_off * k) & 255u;
          if ((diag & 64) && G % 3 == 0) slot[pi] = bp[lane_off + blk_off * k];
          xb[u][k] = slot[pi];
        }
#endif
        else xb[u][k] = bp[lane_off + blk_off * k];
      }
    }
  }
  // Bp: the packed weight, ncols columns; this lane's column in block k: col + blk * k (blk = 16; the output layer, narrower
  // than its blocks, passes clamped columns); ng groups starting at group g0
  __device__ __forceinline__ void prefetch(const float* __restrict__ Bp, int ncols, int col, int q, int g0, int ng_, unsigned blk = 16) {
    B = reinterpret_cast<const f32x4*>(Bp) + (size_t)(4 * g0) * ncols;
    lane_off = (unsigned)(q * ncols + col); blk_off = blk; N4 = 4 * ncols; ng = ng_;
    load_b(0, fb);
    ODK_PIN();
  }
  // A: LDS address of act[sample c][4 q] (16-byte aligned)
  __device__ __forceinline__ void run(f32x4 (&acc)[NBLK], const float* A) {
    f32x4 fa[U], ga[U], gb[U][NBLK];
#if ODK_MLP_SETPRIO
    __builtin_amdgcn_s_setprio(ODK_MLP_SETPRIO);
#endif
    auto load_a = [&](int G0, f32x4* xa) {
#pragma unroll
      for (int u = 0; u < U; u++) xa[u] = *reinterpret_cast<const f32x4*>(A + 16 * (G0 + u));   // in bounds of the LDS image (SLACK)
    };
    auto mma = [&](int G0, const f32x4* xa, const f32x4 (*xb)[NBLK]) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (G0 + u < ng && !(diag & 2)) {     // (diag 2, tools: no MFMAs)
#pragma unroll
          for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < NBLK; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][j], xb[u][k][j], acc[k], 0, 0, 0);
        }
      }
    };
    load_a(0, fa);
    ODK_PIN();
    for (int G0 = 0; G0 < ng; G0 += 2 * U) {
      if (!(diag & 4)) { load_b(G0 + U, gb); load_a(G0 + U, ga); }        // (diag 4, tools: no operand loads inside the loop: what the MFMAs alone cost)
      else if (G0 == 0) {
#pragma unroll
        for (int u = 0; u < U; u++) { ga[u] = fa[u];
#pragma unroll
          for (int k = 0; k < NBLK; k++) gb[u][k] = fb[u][k]; }
      }
      ODK_PIN();
      mma(G0, fa, fb);
      ODK_PIN();
      if (!(diag & 4)) { load_b(G0 + 2 * U, fb); load_a(G0 + 2 * U, fa); }
      ODK_PIN();
      mma(G0 + U, ga, gb);
      ODK_PIN();
    }
#if ODK_MLP_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
};

template <int N> __device__ __forceinline__ void zero(f32x4 (&a)[N]) {
#pragma unroll
  for (int k = 0; k < N; k++) a[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// bias + swish on NBLK accumulator blocks (columns col0 + 16 k; accumulator register v = row 4 q + v): h -> LDS (next layer's A
// operand, row-major) and, when the buffers exist, h and swish'(z) -> global memory in the quad-row layout [rows / 4][width][4]
// (a lane's four rows are one 16-byte store; rows past the end of the batch are written as zeros, so the weight-gradient launch
// may read whole tiles).  hq / gq: this lane's first column in ITS row quad.
template <int NBLK>
__device__ __forceinline__ void fwd_epilogue(const f32x4 (&acc)[NBLK], const float* bias, float* Ls, int P, float* hq, float* gq, int q, int rows_valid) {
#pragma unroll
  for (int k = 0; k < NBLK; k++) {
    f32x4 hv, gv;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int row = 4 * q + t;
      const float z = acc[k][t] + bias[k];
      const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
      const float h = z * sg;
      Ls[row * P + 16 * k] = h;
      const bool live = row < rows_valid;
      hv[t] = live ? h : 0.0f;
      gv[t] = live ? sg + (h - h * sg) : 0.0f;   // swish'(z) = s + z s (1 - s)
    }
    if (hq) {
      *reinterpret_cast<f32x4*>(hq + 64 * k) = hv;
      *reinterpret_cast<f32x4*>(gq + 64 * k) = gv;
    }
  }
}

__global__ void __launch_bounds__(256, ODK_MLP_WG_PER_CU) mlp_fwd_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int net_i, tile;
  map_block(a, blockIdx.x, net_i, tile);
  const Net& N = a.net[net_i];
  const int m0 = tile * TM;
  const int rows_valid = N.n - m0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane >> 4, c = lane & 15;
  const int kin = N.n_in, k16 = pad16(kin), PX = k16 + 4;
  float* X = lds + F_X; float* C1 = lds + F_C; float* H2s = lds + F_H2; float* H3s = lds + F_X;
#ifdef ODK_MLP_DIAG
  Phase<2, 2, true> p1; Phase<4, 1, true> p2;
#else
  Phase<2, 2> p1;     // layer 1 (32 columns of the current chunk), layer 3, output layer
  Phase<4, 1> p2;     // layer 2, K-slice = the chunk, 64 columns
#endif
#ifdef ODK_MLP_DIAG
  p1.diag = p2.diag = a.diag;
  p1.ring = p2.ring = lds + ((F_TOTAL + 3) & ~3);
  if (a.diag & 32) { for (int i = threadIdx.x; i < 2048; i += 256) p1.ring[i] = 0.01f; __syncthreads(); }
#endif
  ODK_WG_BEGIN();
  ODK_STAMP(0);
  p1.prefetch(N.wf[0], H1, w * 32 + c, q, 0, k16 >> 4);
  const int nout = N.n_out;
  int colc[2];
#pragma unroll
  for (int k = 0; k < 2; k++) colc[k] = 16 * k + c < nout ? 16 * k + c : nout - 1;
  // this lane's biases: fetched now, used by the epilogues (a load there is an exposed round trip per block)
  float bias1[H1 / CH][2], bias2[4], bias3[2], bias4[2];
#pragma unroll
  for (int ch = 0; ch < H1 / CH; ch++)
#pragma unroll
    for (int k = 0; k < 2; k++) bias1[ch][k] = N.b[0][ch * CH + w * 32 + 16 * k + c];
#pragma unroll
  for (int k = 0; k < 4; k++) bias2[k] = N.b[1][w * 64 + 16 * k + c];
#pragma unroll
  for (int k = 0; k < 2; k++) { bias3[k] = N.b[2][w * 32 + 16 * k + c]; bias4[k] = N.b[3][colc[k]]; }
  // ---- the tile's input rows, zero-padded to a multiple of 16 columns (rows past the end repeat the last one): all loads
  // first, then the LDS stores (one exposed round trip instead of one per row)
  {
    constexpr int T = (KIN_MAX + 63) / 64;
    float xv[TM / 4][T];
#pragma unroll
    for (int i = 0; i < TM / 4; i++) {
      const int rr = w + 4 * i, row = m0 + rr < N.n ? m0 + rr : N.n - 1;
      const float* src = N.x + (size_t)row * kin;
      bool ok = true;
      if (N.row_idx) {      // the minibatch gather folded into the load (odk_mlp_desc.row_idx): sample -> (trajectory of the schedule, step)
        const int TL = N.traj_len, Bm = N.n_main / TL;
        const long long* idx = N.row_idx + (size_t)(N.cursor ? *N.cursor : 0) * Bm;
        const bool main_row = row < N.n_main;
        const int b = main_row ? row / TL : row - N.n_main;
        const long long j = idx[b];
        ok = j >= 0 && j < N.n_traj;
        const long long jj = ok ? j : 0;
        src = main_row ? N.x + (size_t)(jj * TL + (row - b * TL)) * kin : N.x_tail + (size_t)jj * kin;
      }
#pragma unroll
      for (int t = 0; t < T; t++) { const int k = lane + 64 * t; const float v = src[k < kin ? k : 0]; xv[i][t] = ok ? v : __builtin_nanf(""); }
    }
    if (N.in_mean) {   // observation normaliser folded into the load: (x - mean) / std, IEEE division (bit-identical to the torch ops it replaces)
      float mu[T], sd[T];
#pragma unroll
      for (int t = 0; t < T; t++) { const int k = lane + 64 * t; mu[t] = N.in_mean[k < kin ? k : 0]; sd[t] = N.in_std[k < kin ? k : 0]; }
#pragma unroll
      for (int i = 0; i < TM / 4; i++)
#pragma unroll
        for (int t = 0; t < T; t++) xv[i][t] = (xv[i][t] - mu[t]) / sd[t];
    }
    ODK_PIN();
#pragma unroll
    for (int i = 0; i < TM / 4; i++)
#pragma unroll
      for (int t = 0; t < T; t++) { const int k = lane + 64 * t; if (k < k16) X[(w + 4 * i) * PX + k] = k < kin ? xv[i][t] : 0.0f; }
  }
  __syncthreads();
  ODK_STAMP(1);
#ifdef ODK_MLP_DIAG
  const bool store = N.h[0] != nullptr && !(a.diag & 8);      // (diag 8, tools: no stores of the activations)
#else
  const bool store = N.h[0] != nullptr;
#endif
  const size_t q0 = (size_t)(m0 >> 2);   // the tile's first row quad
  if (store) {   // quad-row copy of the input (the weight-gradient launch's operand)
    for (int k = threadIdx.x; k < kin; k += 256)
#pragma unroll
      for (int qq = 0; qq < TM / 4; qq++) {
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = X[(4 * qq + t) * PX + k];
        *reinterpret_cast<f32x4*>(N.xp + ((q0 + qq) * kin + k) * 4) = v;
      }
  }
  const size_t ql = q0 + q;              // this lane's row quad
  f32x4 acc2[4];
  zero(acc2);
#pragma unroll
  for (int ch = 0; ch < H1 / CH; ch++) {
    f32x4 acc1[2];
    zero(acc1);
    const int col = ch * CH + w * 32 + c;
    float* Cc = C1 + (ODK_FWD_DBUF ? (ch & 1) * TM * PC : 0);    // this chunk's buffer
    p1.run(acc1, X + c * PX + 4 * q);
    ODK_STAMP(2 + 5 * ch);
    p2.prefetch(N.wf[1], H2, w * 64 + c, q, ch * (CH / 16), CH / 16);
    fwd_epilogue<2>(acc1, bias1[ch], Cc + w * 32 + c, PC, store ? N.h[0] + (ql * H1 + col) * 4 : nullptr, store ? N.g[0] + (ql * H1 + col) * 4 : nullptr, q,
                    rows_valid);
    ODK_STAMP(3 + 5 * ch);
    __syncthreads();
    ODK_STAMP(4 + 5 * ch);
    p2.run(acc2, Cc + c * PC + 4 * q);
    ODK_STAMP(5 + 5 * ch);
    if (ch + 1 < H1 / CH) p1.prefetch(N.wf[0], H1, col + CH, q, 0, k16 >> 4);
    else p1.prefetch(N.wf[2], H3, w * 32 + c, q, 0, H2 / 16);
    // one buffer: nobody may overwrite the chunk before every wave has read it.  Two buffers: the next chunk goes to the other one, and the
    // chunk after that is written behind the next chunk's barrier, which a wave passes only after this p2.run -- no barrier here (the last
    // chunk keeps it: the layer-2 epilogue below writes H2s, which nobody reads before the barrier that follows it, but the output layer's
    // partial blocks alias the chunk region)
    if (!ODK_FWD_DBUF || ch + 1 == H1 / CH) __syncthreads();
    ODK_STAMP(6 + 5 * ch);
  }
  {
    const int col = w * 64 + c;
    fwd_epilogue<4>(acc2, bias2, H2s + col, P2, store ? N.h[1] + (ql * H2 + col) * 4 : nullptr, store ? N.g[1] + (ql * H2 + col) * 4 : nullptr, q, rows_valid);
  }
  ODK_STAMP(22);
  __syncthreads();
  ODK_STAMP(23);
  {  // layer 3, columns w * 32 .. + 31
    f32x4 acc3[2];
    zero(acc3);
    const int col = w * 32 + c;
    p1.run(acc3, H2s + c * P2 + 4 * q);
    ODK_STAMP(24);
    // output layer (<= 32 columns): wave w reduces k in [32 w, 32 w + 32); a lane whose column is past n_out reads the last
    // column instead (its results are never stored)
    p1.prefetch(N.wf[3], nout, colc[0], q, w * (H3 / 64), H3 / 64, (unsigned)(colc[1] - colc[0]));
    fwd_epilogue<2>(acc3, bias3, H3s + col, P3, store ? N.h[2] + (ql * H3 + col) * 4 : nullptr, store ? N.g[2] + (ql * H3 + col) * 4 : nullptr, q,
                    rows_valid);   // X is dead: every wave passed two barriers since its last read
  }
  ODK_STAMP(25);
  __syncthreads();
  ODK_STAMP(26);
  {  // partial blocks of the output layer folded by wave 0
    f32x4 acc4[2];
    zero(acc4);
    p1.run(acc4, H3s + c * P3 + 32 * w + 4 * q);
    ODK_STAMP(27);
    float* R = C1;
    if (w > 0) {
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int t = 0; t < 4; t++) R[((w - 1) * 8 + 4 * k + t) * 64 + lane] = acc4[k][t];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int row = 4 * q + t, e = 4 * k + t;
          const float z = ((acc4[k][t] + R[e * 64 + lane]) + (R[(8 + e) * 64 + lane] + R[(16 + e) * 64 + lane])) + bias4[k];
          if (16 * k + c < nout && row < rows_valid) N.out[(size_t)(m0 + row) * nout + 16 * k + c] = z;
        }
    }
  }
  ODK_STAMP(28);
  ODK_WG_END();
}

// dz = dh * swish'(z) on NBLK accumulator blocks (columns col0 + 16 k): -> global in the quad-row layout (the weight-gradient
// launch reads it), -> LDS (next layer's A operand, row-major; may be null), and each block's column sums over the tile's 16
// rows -> partial[col].  swish' is zero in the rows past the end of the batch (the forward pass wrote it so), hence so is dz.
// gq / dzq: this lane's first column in ITS row quad.  All swish' loads are issued before the first use.
template <int NBLK>
__device__ __forceinline__ void load_g(f32x4 (&gv)[NBLK], const float* __restrict__ gq) {
#pragma unroll
  for (int k = 0; k < NBLK; k++) gv[k] = *reinterpret_cast<const f32x4*>(gq + 64 * k);
  ODK_PIN();
}
template <int NBLK>
__device__ __forceinline__ void bwd_epilogue(const f32x4 (&acc)[NBLK], const f32x4 (&gv)[NBLK], float* dzq, float* Ls, int P, float* partial, int q) {
#pragma unroll
  for (int k = 0; k < NBLK; k++) {
    f32x4 dv;
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float dz = acc[k][t] * gv[k][t];
      dv[t] = dz;
      if (Ls) Ls[(4 * q + t) * P + 16 * k] = dz;
      s += dz;
    }
    *reinterpret_cast<f32x4*>(dzq + 64 * k) = dv;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) partial[16 * k] = s;
  }
}

__global__ void __launch_bounds__(256, ODK_MLP_WG_PER_CU) mlp_bwd_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int net_i, tile;
  map_block(a, blockIdx.x, net_i, tile);
  const Net& N = a.net[net_i];
  const int m0 = tile * TM;
  const int rows_valid = N.n - m0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, q = lane >> 4, c = lane & 15;
  const int nout = N.n_out;
  float* D4 = lds + B_D4; float* D3 = lds + B_D3; float* D2 = lds + B_D2;
  Phase<2, 2> p3;     // dh3 = dout W4: columns w * 32 .. + 31, K = n_out
  Phase<4, 1> p2;     // dh2 = dz3 W3: columns w * 64 .. + 63, K = 128
  Phase<8, 1> p1;     // dh1 = dz2 W2: columns w * 128 .. + 127, K = 256
#ifdef ODK_MLP_DIAG
  p1.diag = p2.diag = p3.diag = a.diag;
#endif
  ODK_WG_BEGIN();
  p3.prefetch(N.wb[3], H3, w * 32 + c, q, 0, pad16(nout) >> 4);
  // ---- the tile of dLoss/dout, zero beyond the tile's rows / the layer's columns
  for (int e = threadIdx.x; e < TM * NOUT_MAX; e += 256) {
    const int rr = e >> 5, k = e & 31;
    D4[rr * P4 + k] = (rr < rows_valid && k < nout) ? N.dout[(size_t)(m0 + rr) * nout + k] : 0.0f;
  }
  __syncthreads();
  const size_t q0 = (size_t)(m0 >> 2), ql = q0 + q;   // the tile's first row quad, this lane's row quad
  if (threadIdx.x < nout) {   // output layer's bias gradient: tile sums of dout
    float s = 0.0f;
    for (int rr = 0; rr < TM; rr++) s += D4[rr * P4 + threadIdx.x];
    N.bias_partial[3][(size_t)tile * nout + threadIdx.x] = s;
  }
  for (int e = threadIdx.x; e < (TM / 4) * nout; e += 256) {   // quad-row copy of dout (the weight-gradient launch's operand)
    const int qq = e / nout, k = e - qq * nout;
    f32x4 v;
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = D4[(4 * qq + t) * P4 + k];
    *reinterpret_cast<f32x4*>(N.doutp + ((q0 + qq) * nout + k) * 4) = v;
  }
  f32x4 gv3[2], gv2[4], gv1[8];
#if ODK_BWD_PREFETCH_G
  load_g<2>(gv3, N.g[2] + (ql * H3 + w * 32 + c) * 4);       // (the row quads' swish' pieces do not depend on anything computed here)
  load_g<4>(gv2, N.g[1] + (ql * H2 + w * 64 + c) * 4);
#endif
  {
    f32x4 acc[2];
    zero(acc);
    const int col = w * 32 + c;
    p3.run(acc, D4 + c * P4 + 4 * q);
    p2.prefetch(N.wb[2], H2, w * 64 + c, q, 0, H3 / 16);
#if !ODK_BWD_PREFETCH_G
    load_g<2>(gv3, N.g[2] + (ql * H3 + col) * 4);
#endif
    bwd_epilogue<2>(acc, gv3, N.dz[2] + (ql * H3 + col) * 4, D3 + col, P3, N.bias_partial[2] + (size_t)tile * H3 + col, q);
  }
  __syncthreads();
#if ODK_BWD_PREFETCH_G
  load_g<8>(gv1, N.g[0] + (ql * H1 + w * 128 + c) * 4);
#endif
  {
    f32x4 acc[4];
    zero(acc);
    const int col = w * 64 + c;
    p2.run(acc, D3 + c * P3 + 4 * q);
    p1.prefetch(N.wb[1], H1, w * 128 + c, q, 0, H2 / 16);
#if !ODK_BWD_PREFETCH_G
    load_g<4>(gv2, N.g[1] + (ql * H2 + col) * 4);
#endif
    bwd_epilogue<4>(acc, gv2, N.dz[1] + (ql * H2 + col) * 4, D2 + col, P2, N.bias_partial[1] + (size_t)tile * H2 + col, q);
  }
  __syncthreads();
  {
    f32x4 acc[8];
    zero(acc);
    const int col = w * 128 + c;
    p1.run(acc, D2 + c * P2 + 4 * q);
#if !ODK_BWD_PREFETCH_G
    load_g<8>(gv1, N.g[0] + (ql * H1 + col) * 4);
#endif
    bwd_epilogue<8>(acc, gv1, N.dz[0] + (ql * H1 + col) * 4, nullptr, 0, N.bias_partial[0] + (size_t)tile * H1 + col, q);
  }
  ODK_WG_END();
}

// ---- parameters -> packed weight copies.  Weight k: params[off .. off + rows * cols), torch layout [rows = n_out][cols = n_in];
// forward copy at fwd_off: [pad16(cols) / 4][rows][4] (reduction over the input index), backward copy at bwd_off (< 0: none):
// [pad16(rows) / 4][cols][4] (reduction over the output index).  Padding elements are never written: the caller zeroes the
// buffers once.
struct WeightTable { int off[8], fwd[8], bwd[8], rows[8], cols[8]; int n; };   // (32-bit: the buffers are far below 2^31 floats; checked by the host)

__device__ __forceinline__ void packed_store(const WeightTable& t, int i, float v, float* __restrict__ pf, float* __restrict__ pb) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k < t.n) {
      const unsigned e = (unsigned)(i - t.off[k]);     // i < off wraps to a huge value
      if (e < (unsigned)(t.rows[k] * t.cols[k])) {
        const unsigned o = e / (unsigned)t.cols[k], in = e - o * (unsigned)t.cols[k];
        pf[t.fwd[k] + ((in >> 2) * t.rows[k] + o) * 4 + (in & 3)] = v;
        if (t.bwd[k] >= 0) pb[t.bwd[k] + ((o >> 2) * t.cols[k] + in) * 4 + (o & 3)] = v;
        return;
      }
    }
  }
}

__global__ void pack_weights_kernel(const float* __restrict__ p, float* __restrict__ pf, float* __restrict__ pb, long long n, WeightTable t) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) packed_store(t, (int)i, p[i], pf, pb);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.0f;
  for (int i = 0; i < nw; i++) t += sh[i];
  return t;
}
// end-of-step duties riding on the clip + Adam launch (odk_step_tail): block 0 advances the schedule's cursor, the LAST block folds the
// loss head's per-workgroup sums into the running losses, in workgroup order
struct Tail { int* cursor; const float* partials; int npartials; float* losses; };
__device__ __forceinline__ void step_tail(const Tail& t) {
  if (t.cursor && blockIdx.x == 0 && threadIdx.x == 0) *t.cursor += 1;     // nobody in THIS launch reads it: the next step's launches do
  if (t.partials && blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {
    // lane = (slice q = lane / 4, component c = lane % 4): slice q sums workgroups q, q + 16, ... in ascending order, then the 16 slices fold
    // in a fixed butterfly -- all loads of a lane are independent (as one serial loop over 160 partials the launch grew by 12 us)
    const int c = threadIdx.x & 3, q = threadIdx.x >> 2;
    float s = 0.0f;
    for (int w = q; w < t.npartials; w += 16) s += t.partials[4 * w + c];
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) s += __shfl_xor(s, o);
    if (threadIdx.x < 4) t.losses[c] += s;
  }
}
// adam_kernel of odk_learner.hip (same arithmetic, same fixed-order norm fold) that also writes every updated weight to its
// places in the packed copies
__global__ void adam_packed_kernel(float* __restrict__ p, float* __restrict__ pf, float* __restrict__ pb, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, float* __restrict__ acc, int nblocks, int64_t n, float lr, float b1, float b2, float eps,
                                   float max_norm, WeightTable t_, Tail tail) {
  __shared__ float sh[16];
  step_tail(tail);
  float sq = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) sq += acc[2 + i];
  sq = block_sum(sq, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) acc[0] = sq;
  const float norm = sqrtf(sq), tt = acc[1];
  const float clip = (max_norm > 0.0f && !(norm < max_norm)) ? max_norm / norm : 1.0f;
  const float c1 = 1.0f / (1.0f - powf(b1, tt)), c2 = 1.0f / (1.0f - powf(b2, tt));
  // four consecutive parameters per thread as 16-byte pieces (one trip for the reference networks: 493 469 parameters on
  // 482 x 256 threads); the tail element by element
  const int64_t n4 = n >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const float4 g4 = reinterpret_cast<const float4*>(g)[q], m4 = reinterpret_cast<const float4*>(m)[q], v4 = reinterpret_cast<const float4*>(v)[q],
                 p4 = reinterpret_cast<const float4*>(p)[q];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
    float mo[4], vo[4], po[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float gi = gg[t] * clip;
      mo[t] = b1 * mm[t] + (1.0f - b1) * gi; vo[t] = b2 * vv[t] + (1.0f - b2) * gi * gi;
      po[t] = pp[t] - lr * (mo[t] * c1) / (sqrtf(vo[t] * c2) + eps);
    }
    reinterpret_cast<float4*>(m)[q] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    reinterpret_cast<float4*>(v)[q] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    reinterpret_cast<float4*>(p)[q] = make_float4(po[0], po[1], po[2], po[3]);
#pragma unroll
    for (int t = 0; t < 4; t++) packed_store(t_, (int)(4 * q + t), po[t], pf, pb);
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pi = p[i] - lr * (mi * c1) / (sqrtf(vi * c2) + eps);
    p[i] = pi;
    packed_store(t_, (int)i, pi, pf, pb);
  }
}
// The same update with the weight matrices walked in TILES of 16 output rows x 64 input columns, so that every global access is a run
// of >= 256 bytes: the torch-layout streams (p, g, m, v) row by row, then -- through a 16 x 64 tile of the new weights in LDS -- the
// forward-packed copy ([in / 4][out][4]: 16 consecutive outputs of one input quad = 256 bytes) and the backward-packed copy
// ([out / 4][in][4]: 64 consecutive inputs of one output quad = 1 KB).  Element by element (adam_packed_kernel) a wave's packed
// stores went to 64 different cache lines, 16 bytes each: 8 such instructions per 256 parameters made the launch twice as long as
// its 18 MB of traffic.  Blocks past the tiles take the parameters outside the weight matrices (the biases) linearly.
struct AdamTiles { int tile0[9], tj[8], ntiles, ngap; long long gap0[9], gapn[9]; };
__global__ void __launch_bounds__(256) adam_tiled_kernel(float* __restrict__ p, float* __restrict__ pf, float* __restrict__ pb, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, float* __restrict__ acc, int nblocks, float lr, float b1,
                                                         float b2, float eps, float max_norm, WeightTable t_, AdamTiles at, Tail tail) {
  __shared__ float sh[16];
  __shared__ float tile[16][65];
  step_tail(tail);
  float sq = 0.0f;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) sq += acc[2 + i];
  sq = block_sum(sq, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) acc[0] = sq;
  const float norm = sqrtf(sq), tt = acc[1];
  const float clip = (max_norm > 0.0f && !(norm < max_norm)) ? max_norm / norm : 1.0f;
  const float c1 = 1.0f / (1.0f - powf(b1, tt)), c2 = 1.0f / (1.0f - powf(b2, tt));
  auto update = [&](long long i) -> float {
    const float gi = g[i] * clip;
    const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pi = p[i] - lr * (mi * c1) / (sqrtf(vi * c2) + eps);
    p[i] = pi;
    return pi;
  };
  const int b = blockIdx.x;
  if (b >= at.ntiles) {     // outside the weight matrices
    const int gb = b - at.ntiles, ngb = gridDim.x - at.ntiles;
    for (int k = 0; k < at.ngap; k++)
      for (long long e = (long long)gb * blockDim.x + threadIdx.x; e < at.gapn[k]; e += (long long)ngb * blockDim.x) (void)update(at.gap0[k] + e);
    return;
  }
  int k = 0;
#pragma unroll
  for (int q = 1; q < 8; q++) if (q < t_.n && b >= at.tile0[q]) k = q;
  const int tloc = b - at.tile0[k], tjn = at.tj[k];
  const int o0 = (tloc / tjn) * 16, in0 = (tloc % tjn) * 64;
  const int rows = t_.rows[k], cols = t_.cols[k];
  const long long off = t_.off[k];
  {   // torch layout: lane = input column, four output rows per pass
    const int col = threadIdx.x & 63, rr = threadIdx.x >> 6;
    const bool cin = in0 + col < cols;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int r = rr + 4 * pass, o = o0 + r;
      float pi = 0.0f;
      if (cin && o < rows) pi = update(off + (long long)o * cols + in0 + col);
      tile[r][col] = pi;      // zeros outside the matrix: the packed copies' padding is zero
    }
  }
  __syncthreads();
  {   // forward-packed copy: [in / 4][rows][4]; thread = (input quad iq, output o): 16 consecutive outputs = 256 bytes
    const int iq = threadIdx.x >> 4, o = threadIdx.x & 15;
    if (in0 + 4 * iq < cols && o0 + o < rows)
      *reinterpret_cast<f32x4*>(pf + t_.fwd[k] + ((long long)((in0 >> 2) + iq) * rows + o0 + o) * 4) = f32x4{tile[o][4 * iq], tile[o][4 * iq + 1], tile[o][4 * iq + 2], tile[o][4 * iq + 3]};
  }
  if (t_.bwd[k] >= 0) {   // backward-packed copy: [rows / 4][cols][4]; thread = (output quad oq, input in): 64 consecutive inputs = 1 KB
    const int oq = threadIdx.x >> 6, in = threadIdx.x & 63;
    if (in0 + in < cols && o0 + 4 * oq < rows)
      *reinterpret_cast<f32x4*>(pb + t_.bwd[k] + ((long long)((o0 >> 2) + oq) * cols + in0 + in) * 4) = f32x4{tile[4 * oq][in], tile[4 * oq + 1][in], tile[4 * oq + 2][in], tile[4 * oq + 3][in]};
  }
}
__global__ void sqnorm_p_kernel(const float* __restrict__ g, float* __restrict__ acc, int64_t n) {
  __shared__ float sh[16];
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const float v = g[i]; s += v * v; }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) { acc[2 + blockIdx.x] = s; if (blockIdx.x == 0) acc[1] += 1.0f; }
}

// colsum[f][c] = sum over nblk[f] tile rows of partial[f][tile, c], up to 8 layers in one launch, fixed order
struct FoldArgs { const float* partial[8]; float* out[8]; int w[8], nblk[8]; };
__global__ void __launch_bounds__(1024) colsum_fold_kernel(FoldArgs a) {   // 64 columns x 16 row phases per workgroup
  __shared__ float sh[16][64];
  const int f = blockIdx.y, w = a.w[f], nblk = a.nblk[f];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  if (blockIdx.x * 64 >= w) return;
  const float* partial = a.partial[f];
  float s = 0.0f;
  if (c < w)
    for (int b = ty; b < nblk; b += 16) s += partial[(size_t)b * w + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < w) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += sh[k][tx];
    a.out[f][c] = t;
  }
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return odk_fail_(ODK_ERR_HIP, what);
  return ODK_OK;
}

long long* g_prof = nullptr;
long long* g_wgprof = nullptr;
int g_diag = 0;

int fill_args(Args& a, const odk_mlp_desc* nets, int count, bool backward, int& tiles, const char*& err) {
  a.nnets = count; tiles = 0; a.prof = backward ? nullptr : g_prof; a.wgprof = g_wgprof; a.diag = g_diag;
  for (int k = 0; k < 2; k++) {
    Net& N = a.net[k];
    if (k >= count) { N = a.net[0]; N.tile0 = 1 << 30; continue; }
    const odk_mlp_desc& d = nets[k];
    if (d.n <= 0 || d.n_in <= 0 || d.n_in > KIN_MAX || d.n_out <= 0 || d.n_out > NOUT_MAX) { err = "n_in must be 1..ODK_MLP_MAX_IN, n_out 1..32, n > 0"; return 1; }
    for (int l = 0; l < 4; l++) {
      const float* wp = backward ? d.wb[l] : d.wf[l];
      if ((!backward && (!d.b[l] || !wp)) || (backward && l > 0 && !wp)) { err = "missing packed weights / biases"; return 1; }
      if (((uintptr_t)wp & 15) != 0) { err = "packed weights must be 16-byte aligned"; return 1; }
    }
    const bool has_act = d.h[0] && d.h[1] && d.h[2] && d.g[0] && d.g[1] && d.g[2] && d.xp;
    if (!backward) {
      if (!d.x || !d.out) { err = "missing x / out"; return 1; }
      if ((d.in_mean != nullptr) != (d.in_std != nullptr)) { err = "in_mean / in_std: both or none"; return 1; }
      if (!has_act && (d.h[0] || d.h[1] || d.h[2] || d.g[0] || d.g[1] || d.g[2] || d.xp)) { err = "xp / h / g buffers: all seven or none"; return 1; }
    } else {
      if (!d.dout || !d.doutp || !d.g[0] || !d.g[1] || !d.g[2] || !d.dz[0] || !d.dz[1] || !d.dz[2] || !d.bias_partial[0] || !d.bias_partial[1] ||
          !d.bias_partial[2] || !d.bias_partial[3]) { err = "backward needs dout, doutp, g, dz and bias_partial"; return 1; }
    }
    if (!backward && d.row_idx) {
      const int tail = d.n - d.n_main;
      if (d.traj_len <= 0 || d.n_main <= 0 || d.n_main % d.traj_len != 0 || tail < 0 || d.n_traj <= 0 || (tail > 0 && (!d.x_tail || tail > d.n_main / d.traj_len))) {
        err = "row_idx: n_main a positive multiple of traj_len and <= n; n_traj > 0; rows past n_main need x_tail and number at most n_main / traj_len"; return 1;
      }
    }
    N.row_idx = d.row_idx; N.cursor = d.cursor; N.x_tail = d.x_tail; N.traj_len = d.traj_len; N.n_main = d.n_main; N.n_traj = d.n_traj;
    N.x = d.x; N.in_mean = d.in_mean; N.in_std = d.in_std; N.out = d.out; N.dout = d.dout; N.xp = d.xp; N.doutp = d.doutp; N.n = d.n; N.n_in = d.n_in; N.n_out = d.n_out; N.tile0 = tiles;
    for (int l = 0; l < 4; l++) { N.wf[l] = d.wf[l]; N.wb[l] = d.wb[l]; N.b[l] = d.b[l]; N.bias_partial[l] = d.bias_partial[l]; }
    for (int l = 0; l < 3; l++) { N.h[l] = has_act ? d.h[l] : nullptr; N.g[l] = d.g[l]; N.dz[l] = d.dz[l]; }
    if (!backward && !has_act) for (int l = 0; l < 3; l++) N.g[l] = nullptr;
    tiles += (d.n + TM - 1) / TM;
  }
  // the tile deal (struct Map): two networks, more tiles than CUs, fewer than 4 per CU (all resident at once)
  Map& m = a.map;
  memset(&m, 0, sizeof(m));
  int cus = 0, dev = 0;     // (per call: the current device's, not the first one's)
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (count == 2 && tiles > cus && tiles <= 4 * cus && tiles % cus != 0 && !getenv("ODK_MLP_NO_DEAL")) {
    const int nt[2] = {a.net[1].tile0, tiles - a.net[1].tile0};
    m.R = cus; m.rounds = tiles / cus; m.rem = tiles % cus;
    m.costly = nets[1].n_in >= nets[0].n_in ? 1 : 0;
    const int n2slots = (cus - m.rem) * m.rounds, nv = nt[m.costly];
    m.vin2 = nv < n2slots ? nv : n2slots; m.pin2 = n2slots - m.vin2; m.v3 = nv - m.vin2;
    m.on = 1;
  }
  return 0;
}

int fill_table(WeightTable& t, const odk_weight_table* h, long long n, long long nf, long long nb) {
  if (!h || h->count < 0 || h->count > 8 || n >= (1ll << 30) || nf >= (1ll << 30) || nb >= (1ll << 30)) return 1;
  t.n = h->count;
  for (int k = 0; k < 8; k++) {
    const bool on = k < h->count;
    t.off[k] = on ? (int)h->off[k] : 0; t.fwd[k] = on ? (int)h->fwd_off[k] : 0; t.bwd[k] = on ? (int)h->bwd_off[k] : -1;
    t.rows[k] = on ? h->rows[k] : 0; t.cols[k] = on ? h->cols[k] : 1;
    if (!on) continue;
    const long long rc = (long long)h->rows[k] * h->cols[k];
    if (h->off[k] < 0 || t.rows[k] <= 0 || t.cols[k] <= 0 || h->off[k] + rc > n) return 1;
    if (h->fwd_off[k] < 0 || (h->fwd_off[k] & 3) || h->fwd_off[k] + (long long)pad16(t.cols[k]) * t.rows[k] > nf) return 1;
    if (h->bwd_off[k] >= 0 && ((h->bwd_off[k] & 3) || h->bwd_off[k] + (long long)pad16(t.rows[k]) * t.cols[k] > nb)) return 1;
  }
  return 0;
}

}  // namespace

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set once per (kernel slot, device ordinal) -- a process that drives a
// second device must set it there too -- under a lock, return code checked.  (Once, not per call: the first call of a kernel is a warm-up
// launch outside any stream capture.)  Returns non-zero on failure.
#include <mutex>
int odk_func_lds_attr_(const void* fn, int slot, int bytes) {
  static std::mutex mu;
  static bool done[4][64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || slot < 0 || slot >= 4) return 1;
  std::lock_guard<std::mutex> lock(mu);
  if (done[slot][dev]) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  done[slot][dev] = true;
  return 0;
}

// tools: device buffer of 32 int64 that receives the forward kernel's phase timestamps (workgroup 0); NULL switches it off
extern "C" void odk_mlp_set_profile(long long* stamps_dev) { g_prof = stamps_dev; }
// tools: device buffer of 4 x (workgroups of a network launch) int64: start / end (100 MHz wall clock), HW_ID | XCC_ID << 32, shader cycles
extern "C" void odk_mlp_set_wg_profile(long long* dev) { g_wgprof = dev; }
// tools: diagnostic variants of the network launches (WRONG results): bit 0 = every weight load re-reads the phase's first group (L1 hits), bit 1 = no MFMAs
extern "C" void odk_mlp_set_diag(int bits) { g_diag = bits; }

extern "C" int odk_mlp_forward(const odk_mlp_desc* nets, int count, void* stream) {
  if (!nets || count < 1 || count > 2) return odk_fail_(ODK_ERR_INVALID, "odk_mlp_forward: 1 or 2 networks");
  Args a; int tiles; const char* err = nullptr;
  if (fill_args(a, nets, count, false, tiles, err)) return odk_fail_(ODK_ERR_INVALID, err);
#ifdef ODK_MLP_DIAG
  const int fwd_lds = ((F_TOTAL + 3) & ~3) * 4 + 8192;      // (+ the stand-in for a weight ring: diag bits 5 / 6)
#else
  const int fwd_lds = F_TOTAL * 4;
#endif
  if (odk_func_lds_attr_((const void*)mlp_fwd_kernel, 1, fwd_lds)) return odk_fail_(ODK_ERR_HIP, "odk_mlp_forward: the device refuses the kernel's dynamic LDS size");
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(tiles), dim3(256), fwd_lds, (hipStream_t)stream, a);
  return check_launch("odk_mlp_forward: launch failed");
}

extern "C" int odk_mlp_backward(const odk_mlp_desc* nets, int count, void* stream) {
  if (!nets || count < 1 || count > 2) return odk_fail_(ODK_ERR_INVALID, "odk_mlp_backward: 1 or 2 networks");
  Args a; int tiles; const char* err = nullptr;
  if (fill_args(a, nets, count, true, tiles, err)) return odk_fail_(ODK_ERR_INVALID, err);
  if (odk_func_lds_attr_((const void*)mlp_bwd_kernel, 2, B_TOTAL * 4)) return odk_fail_(ODK_ERR_HIP, "odk_mlp_backward: the device refuses the kernel's dynamic LDS size");
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(tiles), dim3(256), B_TOTAL * 4, (hipStream_t)stream, a);
  return check_launch("odk_mlp_backward: launch failed");
}

extern "C" int odk_pack_weights(const float* params_dev, long long n, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                                const odk_weight_table* table, void* stream) {
  WeightTable t;
  if (!params_dev || !fwd_packed_dev || !bwd_packed_dev || n <= 0 || fill_table(t, table, n, n_fwd, n_bwd))
    return odk_fail_(ODK_ERR_INVALID, "odk_pack_weights: bad arguments (at most 8 weights inside the buffers, packed offsets multiples of 4)");
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params_dev, fwd_packed_dev, bwd_packed_dev, n, t);
  return check_launch("odk_pack_weights: launch failed");
}

extern "C" int odk_adam_clip_packed_tail(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr, float b1,
                                         float b2, float eps, float max_grad_norm, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                                         const odk_weight_table* table, int norm_blocks, const odk_step_tail* tail_, void* stream) {
  Tail tail = {nullptr, nullptr, 0, nullptr};
  if (tail_) {
    if (tail_->loss_partials_dev && (!tail_->losses_dev || tail_->n_loss_partials <= 0)) return odk_fail_(ODK_ERR_INVALID, "odk_adam_clip_packed_tail: loss partials need a count and losses_dev");
    tail.cursor = tail_->cursor_dev; tail.partials = tail_->loss_partials_dev; tail.npartials = tail_->n_loss_partials; tail.losses = tail_->losses_dev;
  }
  WeightTable t;
  if (!params_dev || !fwd_packed_dev || !bwd_packed_dev || !grads_dev || !m_dev || !v_dev || !acc_dev || n <= 0 || fill_table(t, table, n, n_fwd, n_bwd))
    return odk_fail_(ODK_ERR_INVALID, "odk_adam_clip_packed: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int threads = 256;
  int blocks = (int)((n + threads * 4 - 1) / (threads * 4));
  if (blocks > ODK_ADAM_MAX_PARTIALS) blocks = ODK_ADAM_MAX_PARTIALS;
  if (norm_blocks < 0 || norm_blocks > ODK_ADAM_MAX_PARTIALS) return odk_fail_(ODK_ERR_INVALID, "odk_adam_clip_packed: norm_blocks out of range");
  if (norm_blocks == 0) hipLaunchKernelGGL(sqnorm_p_kernel, dim3(blocks), dim3(threads), 0, st, grads_dev, acc_dev, (int64_t)n);
  // tiles of the weight matrices + the gaps between them (biases)
  AdamTiles at;
  memset(&at, 0, sizeof(at));
  bool tiled = t.n > 0 && !getenv("ODK_ADAM_LINEAR");
  long long covered = 0;
  for (int k = 0; k < t.n && tiled; k++) {
    at.tile0[k] = at.ntiles; at.tj[k] = (t.cols[k] + 63) / 64;
    at.ntiles += ((t.rows[k] + 15) / 16) * at.tj[k];
    if ((t.fwd[k] & 3) || (t.bwd[k] >= 0 && (t.bwd[k] & 3)) || t.off[k] < covered) tiled = false;      // (16-byte packed pieces; weights in ascending, disjoint order)
    if (tiled) {
      if (t.off[k] > covered) { at.gap0[at.ngap] = covered; at.gapn[at.ngap] = t.off[k] - covered; at.ngap++; }
      covered = (long long)t.off[k] + (long long)t.rows[k] * t.cols[k];
    }
  }
  if (tiled) {
    for (int k = t.n; k < 9; k++) at.tile0[k] = 1 << 30;
    if (covered < n) { at.gap0[at.ngap] = covered; at.gapn[at.ngap] = n - covered; at.ngap++; }
    long long gap_total = 0;
    for (int k = 0; k < at.ngap; k++) gap_total += at.gapn[k];
    int gap_blocks = (int)((gap_total + 1023) / 1024);
    if (gap_blocks < 1) gap_blocks = 1;
    if (gap_blocks > 64) gap_blocks = 64;
    if ((((uintptr_t)fwd_packed_dev) | ((uintptr_t)bwd_packed_dev)) & 15) tiled = false;
    if (tiled)
      hipLaunchKernelGGL(adam_tiled_kernel, dim3(at.ntiles + gap_blocks), dim3(256), 0, st, params_dev, fwd_packed_dev, bwd_packed_dev, grads_dev, m_dev, v_dev,
                         acc_dev, norm_blocks > 0 ? norm_blocks : blocks, lr, b1, b2, eps, max_grad_norm, t, at, tail);
  }
  if (!tiled)
    hipLaunchKernelGGL(adam_packed_kernel, dim3(blocks), dim3(threads), 0, st, params_dev, fwd_packed_dev, bwd_packed_dev, grads_dev, m_dev, v_dev, acc_dev,
                       norm_blocks > 0 ? norm_blocks : blocks, (int64_t)n, lr, b1, b2, eps, max_grad_norm, t, tail);
  return check_launch("odk_adam_clip_packed: launch failed");
}

extern "C" int odk_adam_clip_packed(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr, float b1,
                                    float b2, float eps, float max_grad_norm, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                                    const odk_weight_table* table, int norm_blocks, void* stream) {
  return odk_adam_clip_packed_tail(params_dev, grads_dev, m_dev, v_dev, acc_dev, n, lr, b1, b2, eps, max_grad_norm, fwd_packed_dev, n_fwd, bwd_packed_dev, n_bwd,
                                   table, norm_blocks, nullptr, stream);
}

extern "C" int odk_colsum_fold(const float* const* partial_dev, float* const* colsum_dev, const int* widths, const int* nblk, int count, void* stream) {
  if (!partial_dev || !colsum_dev || !widths || !nblk || count <= 0 || count > 8) return odk_fail_(ODK_ERR_INVALID, "odk_colsum_fold: bad arguments");
  FoldArgs a;
  int wmax = 0;
  for (int f = 0; f < 8; f++) {
    a.partial[f] = f < count ? partial_dev[f] : nullptr; a.out[f] = f < count ? colsum_dev[f] : nullptr; a.w[f] = f < count ? widths[f] : 0; a.nblk[f] = f < count ? nblk[f] : 0;
    if (f < count && (!partial_dev[f] || !colsum_dev[f] || widths[f] <= 0 || nblk[f] <= 0)) return odk_fail_(ODK_ERR_INVALID, "odk_colsum_fold: bad layer arguments");
    if (a.w[f] > wmax) wmax = a.w[f];
  }
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((wmax + 63) / 64, count), dim3(1024), 0, (hipStream_t)stream, a);
  return check_launch("odk_colsum_fold: launch failed");
}
