// valu_issue.hip -- issue-cost microbenchmark for the instruction kinds the fused env step is made of (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_issue tools/ubench/valu_issue.hip && tools/ubench/valu_issue
//
// One workgroup = one wavefront (like step_kernel); the grid puts W = 1 or 2 waves on every SIMD of every CU
// (256 CUs x 4 SIMDs x W).  Every wave runs ITER x 64 copies of one instruction, as 8 independent chains
// ("indep") or as one dependent chain ("dep").  Reported: shader cycles per instruction per SIMD (wall time x
// clock / instructions issued on one SIMD), i.e. the issue cost the env kernel pays per instruction of that kind.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
  __shared__ float lds[256];
  const int lane = threadIdx.x;
  lds[lane] = lane; lds[lane + 64] = lane; lds[lane + 128] = 1.0f; lds[lane + 192] = 2.0f;
  float a[8];
  float2v p[8];
  for (int i = 0; i < 8; i++) { a[i] = 1.0f + 0.001f * (lane + i); p[i] = float2v{a[i], a[i] + 1.0f}; }
  const float b = 0.999f, c = 0.001f;
  const float2v pb{b, b}, pc{c, c};
  int idx = (lane * 4) & 255;
  int sel = lane & 1;
  unsigned sgp = 0;
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {         // v_fma_f32, 8 independent chains
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (KIND == 1) {  // v_fma_f32, one dependent chain
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (KIND == 2) {  // v_pk_fma_f32, independent
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      REP64(X)
#undef X
    } else if (KIND == 3) {  // v_pk_fma_f32, dependent
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pb), "v"(pc));
      REP64(X)
#undef X
    } else if (KIND == 4) {  // v_add_f32_dpp row_mirror, independent
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == 5) {  // v_add_f32_dpp, dependent (a butterfly reduction is exactly this)
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[0]));
      REP64(X)
#undef X
    } else if (KIND == 6) {  // v_cndmask_b32 (VCC select), independent
      asm volatile("v_cmp_eq_u32 vcc, 1, %0" :: "v"(sel) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if (KIND == 7) {  // v_readlane_b32 -> SGPR -> v_add with the SGPR (the bcast idiom), independent
#define X(i) asm volatile("v_readlane_b32 %1, %0, 3\n\tv_add_f32 %0, %1, %0" : "+v"(a[i]), "=s"(sgp));
      REP64(X)
#undef X
    } else if (KIND == 8) {  // ds_read_b32, uniform address (broadcast read), results consumed once per 8
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(a[i]) : "v"(0));
      REP64(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 9) {  // ds_read_b32, per-lane address (conflict-free)
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(a[i]) : "v"(idx));
      REP64(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 10) { // ds_bpermute_b32
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(idx));
      REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 11) { // v_permlane16_swap
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
      REP64(X)
#undef X
    } else if (KIND == 12) { // v_rcp_f32 (transcendental pipe), independent
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == 13) { // v_mul_f32 independent
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP64(X)
#undef X
    } else if (KIND == 14) { // ds_read dependent chain: read -> use as address -> read (latency)
#define X(i) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(idx));
      REP64(X)
#undef X
    } else if (KIND == 15) { // ds_write_b32 + ds_read_b32 hand-off (write, read other lane's slot, wait)
#define X(i) asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]) : "v"(idx));
      REP64(X)
#undef X
    } else if (KIND == 16) { // 3 fma : 1 uniform ds_read mix (typical phase body), independent
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b), "v"(c));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (KIND == 17) { // v_mov_b32_dpp (unfused permute), independent
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == 18) { // v_pk_mul_f32 independent
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      REP64(X)
#undef X
    } else if (KIND == 19) { // v_add_u32 (address arithmetic), independent
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(idx) : "v"(sel));
      REP64(X)
#undef X
    } else if (KIND == 20) { // s_nop 0 (hazard padding)
#define X(i) asm volatile("s_nop 0");
      REP64(X)
#undef X
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 64 + lane] = s + idx + sgp;
}

struct Kind { const char* name; int per_iter; void (*fn)(float*, int); };

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int clk_khz = 0;
  CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("device %s, %d CUs, clock attribute %.0f MHz\n", prop.name, cus, clk_khz / 1e3);
  float* out;
  CHECK(hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
  const Kind kinds[] = {
      {"v_fma_f32 indep", 64, k<0>}, {"v_fma_f32 dep", 64, k<1>}, {"v_pk_fma_f32 indep", 64, k<2>}, {"v_pk_fma_f32 dep", 64, k<3>},
      {"v_add_f32_dpp indep", 64, k<4>}, {"v_add_f32_dpp dep", 64, k<5>}, {"v_cndmask_b32 indep", 64, k<6>},
      {"v_readlane+v_add(sgpr)", 128, k<7>}, {"ds_read_b32 uniform addr", 64, k<8>}, {"ds_read_b32 per-lane", 64, k<9>},
      {"ds_bpermute_b32", 64, k<10>}, {"v_permlane16_swap", 64, k<11>}, {"v_rcp_f32 indep", 64, k<12>}, {"v_mul_f32 indep", 64, k<13>},
      {"ds_read dep chain (latency)", 64, k<14>}, {"ds_write+ds_read+wait hand-off", 64, k<15>}, {"v_fma x2 pairs", 64, k<16>},
      {"v_mov_b32_dpp indep", 64, k<17>}, {"v_pk_mul_f32 indep", 64, k<18>}, {"v_add_u32 dep", 64, k<19>}, {"s_nop 0", 64, k<20>},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-34s %14s %14s %14s\n", "instruction", "cyc/inst W=1", "cyc/inst W=2", "cyc/inst W=4");
  for (const Kind& kd : kinds) {
    double cyc[3];
    int wi = 0;
    for (int W : {1, 2, 4}) {
      const int grid = cus * 4 * W;
      hipLaunchKernelGGL(kd.fn, dim3(grid), dim3(64), 0, 0, out, 10);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kd.fn, dim3(grid), dim3(64), 0, 0, out, ITER);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double insts_per_simd = (double)W * ITER * kd.per_iter;
      cyc[wi++] = ms * 1e-3 * (clk_khz * 1e3) / insts_per_simd;
    }
    printf("%-34s %14.2f %14.2f %14.2f\n", kd.name, cyc[0], cyc[1], cyc[2]);
  }
  return 0;
}
