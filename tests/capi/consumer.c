/* A C caller of libodk.so with no Python and no torch in the process: the drop-in boundary of include/odk.h used the way a
 * non-Python host would (tests/test_gpu_capi_consumer.py builds and runs it, then compares its outputs bit for bit with the Python host's).
 *   consumer <blob> <prm_table.f32> <prm_grids.f64> <nx> <ny> <nth> <nsteps_in_period> <nenv> <steps> <seed> <out.f32>
 * prm_grids.f64 = dxs[nx] | dys[ny] | dthetas[nth] | ranges[6].  Actions: a fixed integer hash of (step, env, actuator) mapped to [-1, 1).
 * out.f32 = after the last step: obs [nenv, 101] | reward [nenv] | done [nenv] | qpos [nenv, nq]. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/odk.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, odk_last_error()); return 2; } } while (0)
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

static void* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(4); }
  fseek(f, 0, SEEK_END); *len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  void* p = malloc(*len);
  if (fread(p, 1, *len, f) != *len) { perror("read"); exit(4); }
  fclose(f);
  return p;
}

float consumer_action(uint32_t step, uint32_t env, uint32_t act) {   /* same formula in the Python test */
  uint32_t h = step * 2654435761u ^ env * 40503u ^ act * 2246822519u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

int main(int argc, char** argv) {
  if (argc != 12) { fprintf(stderr, "usage: see the header of consumer.c\n"); return 1; }
  size_t blob_len, tl, gl;
  void* blob = slurp(argv[1], &blob_len);
  float* table = (float*)slurp(argv[2], &tl);
  double* grids = (double*)slurp(argv[3], &gl);
  const int nx = atoi(argv[4]), ny = atoi(argv[5]), nth = atoi(argv[6]), period = atoi(argv[7]), nenv = atoi(argv[8]), steps = atoi(argv[9]);
  const uint32_t seed = (uint32_t)strtoul(argv[10], NULL, 10);
  odk_model* m = NULL; odk_batch* b = NULL;
  CHECK(odk_model_load(blob, blob_len, &m));
  int nq, nv, nu, nbody;
  CHECK(odk_model_dims(m, &nq, &nv, &nu, &nbody));
  odk_env_config cfg;
  odk_default_config(&cfg);
  CHECK(odk_batch_create(m, &cfg, nenv, 0, table, grids, nx, grids + nx, ny, grids + nx + ny, nth, grids + nx + ny + nth, period, &b));
  odk_outputs o;
  HIPOK(hipMalloc((void**)&o.obs_dev, sizeof(float) * nenv * ODK_NOBS));
  HIPOK(hipMalloc((void**)&o.priv_dev, sizeof(float) * nenv * ODK_NPRIV));
  HIPOK(hipMalloc((void**)&o.reward_dev, sizeof(float) * nenv));
  HIPOK(hipMalloc((void**)&o.done_dev, sizeof(float) * nenv));
  HIPOK(hipMalloc((void**)&o.truncation_dev, sizeof(float) * nenv));
  HIPOK(hipMalloc((void**)&o.metrics_dev, sizeof(float) * nenv * ODK_NMETRIC));
  float* act_dev; float* act = (float*)malloc(sizeof(float) * nenv * nu);
  HIPOK(hipMalloc((void**)&act_dev, sizeof(float) * nenv * nu));
  hipStream_t st;
  HIPOK(hipStreamCreate(&st));
  CHECK(odk_reset(b, seed, 0, &o, st));
  for (int t = 0; t < steps; t++) {
    for (int e = 0; e < nenv; e++) for (int a = 0; a < nu; a++) act[e * nu + a] = consumer_action((uint32_t)t, (uint32_t)e, (uint32_t)a);
    HIPOK(hipMemcpyAsync(act_dev, act, sizeof(float) * nenv * nu, hipMemcpyHostToDevice, st));
    HIPOK(hipStreamSynchronize(st));   /* (the host buffer is reused next step) */
    CHECK(odk_step(b, act_dev, &o, st));
  }
  HIPOK(hipStreamSynchronize(st));
  const size_t n_out = (size_t)nenv * (ODK_NOBS + 2 + nq);
  float* out = (float*)malloc(sizeof(float) * n_out); float* qvel = (float*)malloc(sizeof(float) * nenv * nv); float* warm = (float*)malloc(sizeof(float) * nenv * nv);
  HIPOK(hipMemcpy(out, o.obs_dev, sizeof(float) * nenv * ODK_NOBS, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(out + (size_t)nenv * ODK_NOBS, o.reward_dev, sizeof(float) * nenv, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(out + (size_t)nenv * (ODK_NOBS + 1), o.done_dev, sizeof(float) * nenv, hipMemcpyDeviceToHost));
  CHECK(odk_batch_get_state(b, out + (size_t)nenv * (ODK_NOBS + 2), qvel, warm));
  FILE* f = fopen(argv[11], "wb");
  if (!f || fwrite(out, sizeof(float), n_out, f) != n_out) { perror(argv[11]); return 4; }
  fclose(f);
  odk_batch_destroy(b); odk_model_free(m);
  printf("consumer: %d envs x %d steps, nq %d nv %d nu %d\n", nenv, steps, nq, nv, nu);
  return 0;
}
