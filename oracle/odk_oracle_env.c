/* odk_oracle_env.c -- CPU restatement of the Joystick env logic.  TEST INFRASTRUCTURE ONLY
 * (see odk_oracle_env.h for the reference line map). */
#include "odk_oracle_env.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI_F 3.14159265358979323846f

/* ------------------------------------------------------------------ RNG (build-defined stream)
 * threefry2x32, 20 rounds (Salmon et al. 2011; the same block function JAX uses).
 *   env key      = TF(key=(seed, 'ODK1'), ctr=(env_id, 0))
 *   uniform(idx) = word[idx&1] of TF(key, ctr=(rng_ctr, idx>>1)), mapped to [0,1) by (w>>8)*2^-24
 * step draw ids (rng_ctr = 0 at reset's observation, then 1,2,...), nu = the robot's actuators (the duck: 14):
 *   0 action-delay index | 2 push theta | 3 push magnitude | 4-6 gyro | 7-9 accelerometer |
 *   10-12 gravity (10 also drives the IMU-delay index: the reference reuses that key,
 *   joystick.py:513-529) | 13 .. 12+nu joint angles | 13+nu .. 12+2nu joint velocities | 13+2nu .. 19+2nu command |
 *   20+2nu zero-command                                              (the duck: 13-26 | 27-40 | 41-47 | 48)
 * reset draw ids (key1 ^ 'RST!', ctr 0): 0-1 dxy | 2 yaw | 3 .. 2+nu joint scale | 3+nu .. 8+nu base qvel |
 *   9+nu .. 15+nu command | 16+nu zero-command | 17+nu push interval  (the duck: 3-16 | 17-22 | 23-29 | 30 | 31) */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0, uint32_t* o1) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int blk = 0; blk < 5; blk++) {
    for (int r = 0; r < 4; r++) {
      x0 += x1;
      x1 = rotl32(x1, R[(blk & 1) * 4 + r]);
      x1 ^= x0;
    }
    x0 += ks[(blk + 1) % 3];
    x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  *o0 = x0; *o1 = x1;
}
void odko_env_key(uint32_t seed, uint32_t env_id, uint32_t* key2) { threefry2x32(seed, 0x4F444B31u, env_id, 0u, &key2[0], &key2[1]); }
float odko_rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr, uint32_t idx) {
  uint32_t a, b;
  threefry2x32(k0, k1, ctr, idx >> 1, &a, &b);
  return (float)((idx & 1 ? b : a) >> 8) * (1.0f / 16777216.0f);
}
static real U(const odko_env* e, uint32_t idx) { return (real)odko_rng_uniform(e->key[0], e->key[1], e->rng_ctr, idx); }
static real UR(const odko_env* e, uint32_t idx) { return (real)odko_rng_uniform(e->key[0], e->key[1] ^ 0x52535421u, 0u, idx); }
static int randint3(float u) { int i = (int)(u * 3.0f); return i > 2 ? 2 : i; } /* jax.random.randint(key,(1,),0,3) */

static real nan_to_num(real x) {
  if (isnan(x)) return 0;
  if (isinf(x)) return x > 0 ? 3.4028234663852886e38 : -3.4028234663852886e38;
  return x;
}

/* ------------------------------------------------------------------ reference motion */
odko_prm* odko_prm_new(const float* table, const double* dxs, int nx, const double* dys, int ny, const double* dths, int nth,
                       const double* ranges, int nsteps) {
  odko_prm* p = (odko_prm*)calloc(1, sizeof(odko_prm));
  p->nx = nx; p->ny = ny; p->nth = nth; p->nsteps = nsteps; p->table = table;
  memcpy(p->dxs, dxs, nx * sizeof(double)); memcpy(p->dys, dys, ny * sizeof(double)); memcpy(p->dths, dths, nth * sizeof(double));
  memcpy(p->ranges, ranges, 6 * sizeof(double));
  return p;
}
void odko_prm_set_table64(odko_prm* p, const double* t) { p->table64 = t; }
void odko_prm_free(odko_prm* p) { free(p); }

/* vel_to_index (poly_reference_motion.py:148-158): clip to range, nearest grid point, first index on ties.
 * Evaluated in float32 like the reference's jnp arrays so that ties (e.g. dy = 0) break identically. */
static int nearest(const double* grid, int n, float v) {
  int best = 0;
  float bd = fabsf((float)grid[0] - v);
  for (int i = 1; i < n; i++) { float dd = fabsf((float)grid[i] - v); if (dd < bd) { bd = dd; best = i; } }
  return best;
}
static float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
void odko_prm_index(const odko_prm* p, real dx, real dy, real dth, int* idx) {
  float x = clipf((float)dx, (float)p->ranges[0], (float)p->ranges[1]);
  float y = clipf((float)dy, (float)p->ranges[2], (float)p->ranges[3]);
  float t = clipf((float)dth, (float)p->ranges[4], (float)p->ranges[5]);
  idx[0] = nearest(p->dxs, p->nx, x); idx[1] = nearest(p->dys, p->ny, y); idx[2] = nearest(p->dths, p->nth, t);
}
void odko_prm_eval(const odko_prm* p, real dx, real dy, real dth, int i, real* out) {
  int idx[3];
  odko_prm_index(p, dx, dy, dth, idx);
  float t = (float)(i % p->nsteps) / (float)p->nsteps;
  t = clipf(t, 0.0f, 1.0f);
  const float* c = p->table + ((size_t)((idx[0] * p->ny + idx[1]) * p->nth + idx[2])) * 40 * 16;
  for (int k = 0; k < 40; k++) {
    float y = c[k * 16];
    for (int q = 1; q < 16; q++) y = fmaf(y, t, c[k * 16 + q]);
    out[k] = (real)y;
  }
}
void odko_prm_eval64(const odko_prm* p, double dx, double dy, double dth, int i, double* out) {
  /* float64 variant used to check against the reference's numpy mirror
     (poly_reference_motion_numpy.py): float64 clip / argmin / polyval */
  double v[3] = {dx, dy, dth};
  const double* grids[3] = {p->dxs, p->dys, p->dths};
  int ns[3] = {p->nx, p->ny, p->nth}, idx[3];
  for (int a = 0; a < 3; a++) {
    double x = v[a] < p->ranges[2 * a] ? p->ranges[2 * a] : (v[a] > p->ranges[2 * a + 1] ? p->ranges[2 * a + 1] : v[a]);
    int best = 0; double bd = fabs(grids[a][0] - x);
    for (int k = 1; k < ns[a]; k++) { double dd = fabs(grids[a][k] - x); if (dd < bd) { bd = dd; best = k; } }
    idx[a] = best;
  }
  double t = (double)(i % p->nsteps) / (double)p->nsteps;
  t = t < 0 ? 0 : (t > 1 ? 1 : t);
  const double* c = p->table64 + ((size_t)((idx[0] * p->ny + idx[1]) * p->nth + idx[2])) * 40 * 16;
  for (int k = 0; k < 40; k++) {
    double y = c[k * 16];
    for (int q = 1; q < 16; q++) y = y * t + c[k * 16 + q];
    out[k] = y;
  }
}

/* ------------------------------------------------------------------ rewards */
real odko_reward_tracking_lin_vel(const real* cmd, const real* lv, real sigma) { /* rewards.py:11-22 */
  real y_tol = 0.1;
  real ex = (cmd[0] - lv[0]) * (cmd[0] - lv[0]);
  real ey = fabs(lv[1] - cmd[1]) - y_tol;
  if (ey < 0) ey = 0;
  return nan_to_num(exp(-(ex + ey * ey) / sigma));
}
real odko_reward_tracking_ang_vel(const real* cmd, const real* av, real sigma) { /* rewards.py:25-31 */
  real e = (cmd[2] - av[2]) * (cmd[2] - av[2]);
  return nan_to_num(exp(-e / sigma));
}
real odko_cost_torques(const real* t, int n) { /* rewards.py:68-69 */
  real s = 0; for (int i = 0; i < n; i++) s += t[i] * t[i];
  return nan_to_num(s);
}
real odko_cost_action_rate(const real* a, const real* b, int n) { /* rewards.py:77-79 */
  real s = 0; for (int i = 0; i < n; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
  return nan_to_num(s);
}
real odko_cost_stand_still(const real* cmd, const real* qpos, const real* qvel, const real* def, int n) { /* rewards.py:93-117, ignore_head=False */
  real cn = sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
  real pc = 0, vc = 0;
  for (int i = 0; i < n; i++) { pc += fabs(qpos[i] - def[i]); vc += fabs(qvel[i]); }
  return nan_to_num(pc + vc) * (cn < 0.01 ? 1.0 : 0.0);
}
real odko_cost_stand_still_legs(const real* cmd, const real* qpos, const real* qvel, const real* def, int n) { /* rewards.py:105-117, ignore_head=True */
  real cn = sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
  real pc = 0, vc = 0;
  for (int i = 0; i < n; i++) if (i < 5 || i >= 9) { pc += fabs(qpos[i] - def[i]); vc += fabs(qvel[i]); }
  return nan_to_num(pc + vc) * (cn < 0.01 ? 1.0 : 0.0);
}
real odko_cost_orientation(const real* z) { return nan_to_num(z[0] * z[0] + z[1] * z[1]); } /* rewards.py:45-46 */
real odko_cost_head_pos(const real* jq, const real* cmd) { /* rewards.py:131-147: gated by the MOVE command norm (> 0.01) */
  real cn = sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
  real e = 0;
  for (int k = 0; k < 4; k++) e += (jq[5 + k] - cmd[3 + k]) * (jq[5 + k] - cmd[3 + k]);
  return nan_to_num(e) * (cn > 0.01 ? 1.0 : 0.0);
}
real odko_reward_imitation(const real* base_qpos, const real* base_qvel, const real* jq, const real* jv, const real* contacts,
                           const real* ref, const real* cmd) { /* custom_rewards.py:4-148 */
  real cn = sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
  const real w_lin_vel_xy = 1.0, w_lin_vel_z = 1.0, w_ang_vel_xy = 0.5, w_ang_vel_z = 0.5, w_joint_pos = 15.0, w_joint_vel = 1.0e-3, w_contact = 1.0;
  const real* ref_lin = ref + 34; const real* ref_ang = ref + 37;
  const real* lin = base_qvel; const real* ang = base_qvel + 3;
  (void)base_qpos; /* torso orientation term is computed but not summed in the reference (:104-107,145) */
  real jp_err = 0, jv_err = 0;
  for (int k = 0; k < 10; k++) {
    int ri = k < 5 ? k : k + 6;  /* ref[:5] ++ ref[11:16] */
    int ji = k < 5 ? k : k + 4;  /* joints[:5] ++ joints[9:14] */
    real dp = jq[ji] - ref[ri], dv = jv[ji] - ref[16 + ri];
    jp_err += dp * dp; jv_err += dv * dv;
  }
  real lin_xy = exp(-8.0 * ((lin[0] - ref_lin[0]) * (lin[0] - ref_lin[0]) + (lin[1] - ref_lin[1]) * (lin[1] - ref_lin[1]))) * w_lin_vel_xy;
  real lin_z = exp(-8.0 * (lin[2] - ref_lin[2]) * (lin[2] - ref_lin[2])) * w_lin_vel_z;
  real ang_xy = exp(-2.0 * ((ang[0] - ref_ang[0]) * (ang[0] - ref_ang[0]) + (ang[1] - ref_ang[1]) * (ang[1] - ref_ang[1]))) * w_ang_vel_xy;
  real ang_z = exp(-2.0 * (ang[2] - ref_ang[2]) * (ang[2] - ref_ang[2])) * w_ang_vel_z;
  real contact_rew = 0;
  for (int f = 0; f < 2; f++) {
    real rc = ref[32 + f] > 0.5 ? 1.0 : 0.0;
    contact_rew += (contacts[f] == rc) ? 1.0 : 0.0;
  }
  real reward = lin_xy + lin_z + ang_xy + ang_z - jp_err * w_joint_pos - jv_err * w_joint_vel + contact_rew * w_contact;
  reward *= (cn > 0.01) ? 1.0 : 0.0;
  return nan_to_num(reward);
}

/* ------------------------------------------------------------------ env */
static void default_cfg(odko_env_cfg* c, int nu) {
  memset(c, 0, sizeof(*c));
  c->ctrl_dt = 0.02; c->action_scale = 0.25; c->dof_vel_scale = 0.05; c->max_motor_velocity = 5.24;
  c->noise_level = 1.0; c->noise_gyro = 0.1; c->noise_accelerometer = 0.05; c->noise_gravity = 0.1; c->noise_joint_vel = 2.5;
  /* BUG-COMPAT (joystick.py:184-200): indices from the 10-entry JOINTS_ORDER_NO_HEAD written into the
     nu-entry actuator-ordered array: hips {0,1,2,5,6,7}=0.03, knees {3,8}=0.05, ankles {4,9}=0.08 */
  static const real s10[10] = {0.03, 0.03, 0.03, 0.05, 0.08, 0.03, 0.03, 0.03, 0.05, 0.08};
  for (int i = 0; i < nu && i < 10; i++) c->qpos_noise_scale[i] = s10[i];
  /* order: tracking_lin_vel, tracking_ang_vel, torques, action_rate, stand_still, alive, imitation (joystick.py:78-86) */
  static const real rs[7] = {2.5, 6.0, -1.0e-3, -0.5, -0.2, 20.0, 1.0};
  memcpy(c->reward_scales, rs, sizeof(rs));
  c->tracking_sigma = 0.01;
  c->push_enable = 1; c->push_interval_range[0] = 5.0; c->push_interval_range[1] = 10.0;
  c->push_magnitude_range[0] = 0.1; c->push_magnitude_range[1] = 1.0;
  static const real cr[7][2] = {{-0.15, 0.15}, {-0.2, 0.2}, {-1.0, 1.0}, {-0.34, 1.1}, {-0.78, 0.78}, {-1.5, 1.5}, {-0.5, 0.5}};
  memcpy(c->cmd_range, cr, sizeof(cr));
  c->use_imitation = 1; c->use_motor_speed_limits = 1; c->autoreset = 1; c->episode_length = 1000; c->n_substeps = 10;
  c->env_kind = 0; c->reset_base_qvel = 0.05;
}
/* standing.py:44-100 default_config; reward slots: orientation, head_pos, torques, action_rate, stand_still, alive, (unused) */
void odko_env_set_standing(odko_env* e) {
  odko_env_cfg* c = &e->cfg;
  c->env_kind = 1; c->reset_base_qvel = 0.5;
  c->noise_gyro = 0.05; c->noise_accelerometer = 0.005;
  static const real rs[7] = {-0.5, -2.0, -1.0e-3, -0.375, -0.3, 20.0, 0.0};
  memcpy(c->reward_scales, rs, sizeof(rs));
  for (int k = 0; k < 3; k++) c->cmd_range[k][0] = c->cmd_range[k][1] = 0.0; /* standing.py:652-654: no move command */
  c->cmd_range[5][0] = -2.7; c->cmd_range[5][1] = 2.7;
  c->use_imitation = 0; c->use_motor_speed_limits = 0;
}
int odko_env_nobs(const odko_env* e) { return e->cfg.env_kind != 0 ? 15 + 5 * e->m->nu : 17 + 6 * e->m->nu; }
int odko_env_npriv(const odko_env* e) { return odko_env_nobs(e) + 26 + 3 * e->m->nu + (e->cfg.env_kind != 0 ? 0 : 43); }

odko_env* odko_env_new(const odko_model* m, const odko_prm* prm, const odko_env_cfg* cfg) {
  odko_env* e = (odko_env*)calloc(1, sizeof(odko_env));
  e->m = m; e->prm = prm;
  if (cfg) e->cfg = *cfg; else default_cfg(&e->cfg, m->nu);
  /* actuator joints / backlash twins (base.py:63-125): a backlash joint is the hinge that follows an
     actuated joint on the same body */
  int is_act[ODKO_MAXJ] = {0};
  for (int u = 0; u < m->nu; u++) is_act[m->actuator_trnid[u]] = 1;
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_trnid[u];
    e->act_qposadr[u] = m->jnt_qposadr[j];
    e->act_dofadr[u] = m->jnt_dofadr[j];
    e->backlash_qposadr[u] = -1;
    if (j + 1 < m->njnt && !is_act[j + 1] && m->jnt_type[j + 1] == ODKO_JNT_HINGE && m->jnt_bodyid[j + 1] == m->jnt_bodyid[j])
      e->backlash_qposadr[u] = m->jnt_qposadr[j + 1];
  }
  /* named objects: set by the caller through odko_env_int (defaults match the shipped models) */
  e->imu_site = 0; e->feet_site[0] = 2; e->feet_site[1] = 4;
  e->feet_cgeom[0] = 0; e->feet_cgeom[1] = 1; e->floor_cgeom = 2;
  e->adr_gyro = 0; e->adr_local_linvel = 3; e->adr_accelerometer = 6; e->adr_upvector = 9; e->adr_global_angvel = 18;
  e->adr_foot_linvel[0] = 31; e->adr_foot_linvel[1] = 28;
  return e;
}
void odko_env_free(odko_env* e) { free(e); }
odko_env* odko_env_clone(const odko_env* e) { /* flat struct (model / table pointers are shared): tests step copies from one state */
  odko_env* c = (odko_env*)malloc(sizeof(odko_env));
  memcpy(c, e, sizeof(odko_env));
  return c;
}
odko_data* odko_env_data(odko_env* e) { return &e->d; }

#define CF(nm, cnt) if (!strcmp(name, #nm)) { *count = (cnt); return (real*)&e->cfg.nm; }
real* odko_env_config(odko_env* e, const char* name, int* count) {
  CF(ctrl_dt, 1) CF(action_scale, 1) CF(dof_vel_scale, 1) CF(max_motor_velocity, 1) CF(noise_level, 1) CF(noise_gyro, 1)
  CF(noise_accelerometer, 1) CF(noise_gravity, 1) CF(noise_joint_vel, 1) CF(qpos_noise_scale, ODKO_MAXU) CF(reward_scales, 7)
  CF(tracking_sigma, 1) CF(push_enable, 1) CF(push_interval_range, 2) CF(push_magnitude_range, 2) CF(cmd_range, 14)
  CF(use_imitation, 1) CF(use_motor_speed_limits, 1) CF(autoreset, 1) CF(episode_length, 1) CF(n_substeps, 1)
  CF(env_kind, 1) CF(reset_base_qvel, 1)
  *count = 0; return NULL;
}
#define EF(nm, cnt) if (!strcmp(name, #nm)) { *count = (cnt); return (real*)e->nm; }
#define EF1(nm) if (!strcmp(name, #nm)) { *count = 1; return (real*)&e->nm; }
real* odko_env_field(odko_env* e, const char* name, int* count) {
  EF(command, 7) EF(last_act, ODKO_MAXU) EF(last_last_act, ODKO_MAXU) EF(last_last_last_act, ODKO_MAXU) EF(motor_targets, ODKO_MAXU)
  EF(feet_air_time, 2) EF(swing_peak, 2) EF(push, 2) EF(action_history, 3 * ODKO_MAXU) EF(imu_history, 9)
  EF(current_reference_motion, 40) EF(imitation_phase, 2) EF(ep_metrics, ODKO_NMETRIC)
  EF(first_qpos, ODKO_MAXQ) EF(first_qvel, ODKO_MAXV) EF(first_warmstart, ODKO_MAXV) EF(first_obs, ODKO_NOBS) EF(first_priv, ODKO_NPRIV)
  EF(obs, ODKO_NOBS) EF(priv, ODKO_NPRIV) EF(metrics, ODKO_NMETRIC) EF(contact, 2)
  EF1(reward) EF1(done) EF1(ep_steps) EF1(truncation) EF1(episode_done) EF1(ep_sum_reward) EF1(ep_length)
  *count = 0; return NULL;
}
#define EI(nm, cnt) if (!strcmp(name, #nm)) { *count = (cnt); return (int*)e->nm; }
#define EI1(nm) if (!strcmp(name, #nm)) { *count = 1; return (int*)&e->nm; }
int* odko_env_int(odko_env* e, const char* name, int* count) {
  EI(act_qposadr, ODKO_MAXU) EI(act_dofadr, ODKO_MAXU) EI(backlash_qposadr, ODKO_MAXU) EI(feet_site, 2) EI(feet_cgeom, 2)
  EI(adr_foot_linvel, 2) EI(last_contact, 2) EI(key, 2)
  EI1(imu_site) EI1(floor_cgeom) EI1(adr_gyro) EI1(adr_local_linvel) EI1(adr_accelerometer) EI1(adr_upvector) EI1(adr_global_angvel)
  EI1(step) EI1(push_step) EI1(push_interval_steps) EI1(imitation_i) EI1(rng_ctr)
  *count = 0; return NULL;
}

/* sample_command (joystick.py:671-725); `base` = first draw id, reset = use the reset stream */
static void sample_command(const odko_env* e, int reset, uint32_t base, real* cmd) {
  float z = (float)(reset ? UR(e, base + 7) : U(e, base + 7));
  for (int k = 0; k < 7; k++) {
    real u = reset ? UR(e, base + k) : U(e, base + k);
    cmd[k] = e->cfg.cmd_range[k][0] + u * (e->cfg.cmd_range[k][1] - e->cfg.cmd_range[k][0]);
  }
  if (z < 0.1f) for (int k = 0; k < 7; k++) cmd[k] = 0; /* bernoulli(p=0.1) -> all zero */
}

/* geoms_colliding(data, foot, floor): min dist over that pair's contacts < 0 (mujoco_playground collision.py) */
static void foot_contacts(const odko_env* e, real* contact) {
  const odko_data* d = &e->d;
  for (int f = 0; f < 2; f++) {
    real md = 1e4;
    int found = 0;
    for (int c = 0; c < d->ncon; c++) {
      int a = d->contact_geom1[c], b = d->contact_geom2[c];
      if ((a == e->floor_cgeom && b == e->feet_cgeom[f]) || (b == e->floor_cgeom && a == e->feet_cgeom[f])) {
        if (d->contact_dist[c] < md) md = d->contact_dist[c];
        found = 1;
      }
    }
    contact[f] = (found && md < 0) ? 1.0 : 0.0;
  }
}

/* _get_obs (joystick.py:487-620) */
static void get_obs(odko_env* e, const real* contact) {
  const odko_model* m = e->m;
  const odko_data* d = &e->d;
  const odko_env_cfg* c = &e->cfg;
  int nu = m->nu;
  real lvl = c->noise_level;
  const real* gyro = d->sensordata + e->adr_gyro;
  const real* acc = d->sensordata + e->adr_accelerometer;
  real ngyro[3], nacc[3], gravity[3], ngrav[3];
  for (int k = 0; k < 3; k++) ngyro[k] = gyro[k] + (2 * U(e, 4 + k) - 1) * lvl * c->noise_gyro;
  /* BUG-COMPAT: `accelerometer.at[0].set(...)` result is discarded (:502) -> no +1.3 offset */
  for (int k = 0; k < 3; k++) nacc[k] = acc[k] + (2 * U(e, 7 + k) - 1) * lvl * c->noise_accelerometer;
  const real* R = d->site_xmat[e->imu_site];
  gravity[0] = -R[6]; gravity[1] = -R[7]; gravity[2] = -R[8]; /* site_xmat^T (0,0,-1) */
  for (int k = 0; k < 3; k++) ngrav[k] = gravity[k] + (2 * U(e, 10 + k) - 1) * lvl * c->noise_gravity;
  /* IMU delay ring: roll by 3, newest first; delayed sample is computed but never emitted (:522-530) */
  for (int k = 8; k >= 3; k--) e->imu_history[k] = e->imu_history[k - 3];
  for (int k = 0; k < 3; k++) e->imu_history[k] = ngrav[k];
  int imu_idx = randint3(odko_rng_uniform(e->key[0], e->key[1], e->rng_ctr, 10));
  (void)imu_idx;
  real ja[ODKO_MAXU], jv[ODKO_MAXU], nja[ODKO_MAXU], njv[ODKO_MAXU];
  for (int u = 0; u < nu; u++) {
    real bl = e->backlash_qposadr[u] >= 0 ? d->qpos[e->backlash_qposadr[u]] : 0.0;
    ja[u] = d->qpos[e->act_qposadr[u]] + bl; /* joint_angles + joint_backlash (zeros where no twin) */
    jv[u] = d->qvel[e->act_dofadr[u]];
    nja[u] = ja[u] + (2.0 * U(e, 13 + u) - 1.0) * lvl * c->qpos_noise_scale[u];
    njv[u] = jv[u] + (2.0 * U(e, 13 + nu + u) - 1.0) * lvl * c->noise_joint_vel;
  }
  real* o = e->obs;
  int p = 0;
  for (int k = 0; k < 3; k++) o[p++] = ngyro[k];
  for (int k = 0; k < 3; k++) o[p++] = nacc[k];
  for (int k = 0; k < 7; k++) o[p++] = e->command[k];
  for (int u = 0; u < nu; u++) o[p++] = nja[u] - m->key_ctrl[u];
  for (int u = 0; u < nu; u++) o[p++] = njv[u] * c->dof_vel_scale;
  for (int u = 0; u < nu; u++) o[p++] = e->last_act[u];
  for (int u = 0; u < nu; u++) o[p++] = e->last_last_act[u];
  for (int u = 0; u < nu; u++) o[p++] = e->last_last_last_act[u];
  const int standing = c->env_kind != 0; /* standing.py:524-540: no motor_targets, empty current_reference_motion */
  if (!standing) for (int u = 0; u < nu; u++) o[p++] = e->motor_targets[u];
  for (int k = 0; k < 2; k++) o[p++] = contact[k];
  if (!standing) for (int k = 0; k < 2; k++) o[p++] = e->imitation_phase[k];
  real* q = e->priv;
  int s = 0;
  for (int k = 0; k < p; k++) q[s++] = o[k];
  for (int k = 0; k < 3; k++) q[s++] = gyro[k];
  for (int k = 0; k < 3; k++) q[s++] = acc[k];
  for (int k = 0; k < 3; k++) q[s++] = gravity[k];
  for (int k = 0; k < 3; k++) q[s++] = d->sensordata[e->adr_local_linvel + k];
  for (int k = 0; k < 3; k++) q[s++] = d->sensordata[e->adr_global_angvel + k];
  for (int u = 0; u < nu; u++) q[s++] = ja[u] - m->key_ctrl[u];
  for (int u = 0; u < nu; u++) q[s++] = jv[u];
  q[s++] = d->qpos[2]; /* root height */
  for (int u = 0; u < nu; u++) q[s++] = d->actuator_force[u];
  for (int k = 0; k < 2; k++) q[s++] = contact[k];
  for (int f = 0; f < 2; f++) for (int k = 0; k < 3; k++) q[s++] = d->sensordata[e->adr_foot_linvel[f] + k];
  for (int k = 0; k < 2; k++) q[s++] = e->feet_air_time[k];
  if (!standing) {
    for (int k = 0; k < 40; k++) q[s++] = e->current_reference_motion[k];
    q[s++] = (real)e->imitation_i;
    for (int k = 0; k < 2; k++) q[s++] = e->imitation_phase[k];
  }
  for (int k = p; k < ODKO_NOBS; k++) o[k] = 0;
  for (int k = s; k < ODKO_NPRIV; k++) q[k] = 0;
}

static void quat_mul_local(real* r, const real* a, const real* b) {
  r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

/* Joystick.reset (joystick.py:206-321) + wrapper resets */
void odko_env_reset(odko_env* e, uint32_t seed, uint32_t env_id) {
  const odko_model* m = e->m;
  odko_data* d = &e->d;
  int nu = m->nu;
  odko_env_key(seed, env_id, e->key);
  e->rng_ctr = 0;
  odko_make_data(m, d);
  for (int i = 0; i < m->nq; i++) d->qpos[i] = m->key_qpos[i];
  d->qpos[0] += -0.05 + UR(e, 0) * 0.1;
  d->qpos[1] += -0.05 + UR(e, 1) * 0.1;
  real yaw = -3.14 + UR(e, 2) * 6.28;
  real qz[4] = {cos(0.5 * yaw), 0, 0, sin(0.5 * yaw)}, nq[4];
  quat_mul_local(nq, d->qpos + 3, qz);
  memcpy(d->qpos + 3, nq, sizeof(nq));
  for (int u = 0; u < nu; u++) d->qpos[e->act_qposadr[u]] *= 0.5 + UR(e, 3 + u) * 1.0;
  for (int k = 0; k < 6; k++) d->qvel[k] = -e->cfg.reset_base_qvel + UR(e, 3 + nu + k) * (2 * e->cfg.reset_base_qvel);
  for (int u = 0; u < nu; u++) d->ctrl[u] = d->qpos[e->act_qposadr[u]];
  odko_forward(m, d);
  sample_command(e, 1, 9 + nu, e->command);
  real push_interval = e->cfg.push_interval_range[0] + UR(e, 17 + nu) * (e->cfg.push_interval_range[1] - e->cfg.push_interval_range[0]);
  e->push_interval_steps = (int)rint(push_interval / e->cfg.ctrl_dt);
  if (e->cfg.use_imitation) odko_prm_eval(e->prm, e->command[0], e->command[1], e->command[2], 0, e->current_reference_motion);
  else memset(e->current_reference_motion, 0, sizeof(e->current_reference_motion));
  e->step = 0; e->push_step = 0; e->imitation_i = 0;
  memset(e->last_act, 0, sizeof(e->last_act)); memset(e->last_last_act, 0, sizeof(e->last_last_act));
  memset(e->last_last_last_act, 0, sizeof(e->last_last_last_act));
  for (int u = 0; u < nu; u++) e->motor_targets[u] = e->cfg.env_kind != 0 ? 0.0 : m->key_ctrl[u]; /* standing.py:279 zeros */
  memset(e->feet_air_time, 0, sizeof(e->feet_air_time)); memset(e->swing_peak, 0, sizeof(e->swing_peak));
  memset(e->push, 0, sizeof(e->push)); memset(e->action_history, 0, sizeof(e->action_history));
  memset(e->imu_history, 0, sizeof(e->imu_history)); memset(e->imitation_phase, 0, sizeof(e->imitation_phase));
  e->last_contact[0] = e->last_contact[1] = 0;
  memset(e->metrics, 0, sizeof(e->metrics));
  foot_contacts(e, e->contact);
  get_obs(e, e->contact);
  e->reward = 0; e->done = 0;
  /* wrappers: Episode (steps/truncation/episode metrics) + AutoReset (first_state/first_obs) */
  e->ep_steps = 0; e->truncation = 0; e->episode_done = 0; e->ep_sum_reward = 0; e->ep_length = 0;
  memset(e->ep_metrics, 0, sizeof(e->ep_metrics));
  memcpy(e->first_qpos, d->qpos, sizeof(e->first_qpos)); memcpy(e->first_qvel, d->qvel, sizeof(e->first_qvel));
  memcpy(e->first_warmstart, d->qacc_warmstart, sizeof(e->first_warmstart));
  memcpy(e->first_obs, e->obs, sizeof(e->first_obs)); memcpy(e->first_priv, e->priv, sizeof(e->first_priv));
  e->rng_ctr = 1;
}

/* AutoReset.step -> Episode.step -> Joystick.step (joystick.py:323-481) */
void odko_env_step(odko_env* e, const real* action) {
  const odko_model* m = e->m;
  odko_data* d = &e->d;
  const odko_env_cfg* c = &e->cfg;
  int nu = m->nu;
  real dt = c->ctrl_dt;
  /* AutoReset.step prologue */
  if (e->done != 0) e->ep_steps = 0;
  /* ---- Joystick.step */
  if (c->use_imitation) {
    e->imitation_i = (e->imitation_i + 1) % e->prm->nsteps;
    float ph = ((float)e->imitation_i / (float)e->prm->nsteps) * 2.0f * PI_F;
    e->imitation_phase[0] = cosf(ph); e->imitation_phase[1] = sinf(ph);
    odko_prm_eval(e->prm, e->command[0], e->command[1], e->command[2], e->imitation_i, e->current_reference_motion);
  } else {
    e->imitation_i = 0;
  }
  /* action delay ring (:362-376) */
  for (int k = 3 * nu - 1; k >= nu; k--) e->action_history[k] = e->action_history[k - nu];
  for (int u = 0; u < nu; u++) e->action_history[u] = action[u];
  int aidx = randint3(odko_rng_uniform(e->key[0], e->key[1], e->rng_ctr, 0));
  const real* awd = e->action_history + aidx * nu;
  /* push (:381-398) */
  real theta = U(e, 2) * (real)(2.0f * PI_F);
  real mag = c->push_magnitude_range[0] + U(e, 3) * (c->push_magnitude_range[1] - c->push_magnitude_range[0]);
  real gate = (((e->push_step + 1) % e->push_interval_steps) == 0 ? 1.0 : 0.0) * c->push_enable;
  real push[2] = {cos(theta) * gate, sin(theta) * gate};
  d->qvel[0] += push[0] * mag; d->qvel[1] += push[1] * mag;
  /* motor targets with speed limit (:404-417) */
  real mt[ODKO_MAXU];
  for (int u = 0; u < nu; u++) {
    mt[u] = m->key_ctrl[u] + awd[u] * c->action_scale;
    if (c->use_motor_speed_limits) {
      real lo = e->motor_targets[u] - c->max_motor_velocity * dt, hi = e->motor_targets[u] + c->max_motor_velocity * dt;
      mt[u] = mt[u] < lo ? lo : (mt[u] > hi ? hi : mt[u]);
    }
  }
  odko_env_physics_step(m, d, mt, (int)c->n_substeps);
  for (int u = 0; u < nu; u++) e->motor_targets[u] = mt[u];
  /* contacts & air time (:424-435) */
  real contact[2], first_contact[2];
  foot_contacts(e, contact);
  for (int f = 0; f < 2; f++) {
    int cf = (contact[f] != 0) || e->last_contact[f];
    first_contact[f] = (e->feet_air_time[f] > 0.0 ? 1.0 : 0.0) * cf;
    e->feet_air_time[f] += dt;
    real pz = d->site_xpos[e->feet_site[f]][2];
    if (pz > e->swing_peak[f]) e->swing_peak[f] = pz;
  }
  (void)first_contact;
  get_obs(e, contact);
  /* termination (:483-485) */
  int done = d->sensordata[e->adr_upvector + 2] < 0.0;
  for (int i = 0; i < m->nq; i++) if (isnan(d->qpos[i])) done = 1;
  for (int i = 0; i < m->nv; i++) if (isnan(d->qvel[i])) done = 1;
  /* rewards (:622-669, :440-447) */
  real jq[ODKO_MAXU], jv[ODKO_MAXU], rew[7];
  for (int u = 0; u < nu; u++) { jq[u] = d->qpos[e->act_qposadr[u]]; jv[u] = d->qvel[e->act_dofadr[u]]; }
  if (c->env_kind != 0) { /* standing.py:585-606 */
    rew[0] = odko_cost_orientation(d->sensordata + e->adr_upvector);
    rew[1] = odko_cost_head_pos(jq, e->command);
  } else {
    rew[0] = odko_reward_tracking_lin_vel(e->command, d->sensordata + e->adr_local_linvel, c->tracking_sigma);
    rew[1] = odko_reward_tracking_ang_vel(e->command, d->sensordata + e->adr_gyro, c->tracking_sigma);
  }
  rew[2] = odko_cost_torques(d->actuator_force, nu);
  rew[3] = odko_cost_action_rate(action, e->last_act, nu);
  rew[4] = c->env_kind != 0 ? odko_cost_stand_still_legs(e->command, jq, jv, m->key_ctrl, nu) : odko_cost_stand_still(e->command, jq, jv, m->key_ctrl, nu);
  rew[5] = 1.0;
  rew[6] = c->use_imitation ? odko_reward_imitation(d->qpos, d->qvel, jq, jv, contact, e->current_reference_motion, e->command) : 0.0;
  real total = 0;
  for (int k = 0; k < 7; k++) { rew[k] *= c->reward_scales[k]; total += rew[k]; }
  real reward = total * dt;
  reward = reward < 0 ? 0 : (reward > 10000.0 ? 10000.0 : reward);
  /* info updates (:449-469) */
  e->push[0] = push[0]; e->push[1] = push[1];
  e->step += 1; e->push_step += 1;
  memcpy(e->last_last_last_act, e->last_last_act, sizeof(e->last_act));
  memcpy(e->last_last_act, e->last_act, sizeof(e->last_act));
  for (int u = 0; u < nu; u++) e->last_act[u] = action[u];
  if (e->step > 500) sample_command(e, 0, 13 + 2 * nu, e->command);
  if (done || e->step > 500) e->step = 0;
  for (int f = 0; f < 2; f++) {
    if (contact[f] != 0) { e->feet_air_time[f] = 0; e->swing_peak[f] = 0; }
    e->last_contact[f] = contact[f] != 0;
  }
  /* metrics (:470-477): reward/<k> = v, cost/<k> = -v */
  for (int k = 0; k < 7; k++) e->metrics[k] = c->reward_scales[k] > 0 ? rew[k] : -rew[k];
  e->metrics[7] = 0.5 * (e->swing_peak[0] + e->swing_peak[1]);
  e->contact[0] = contact[0]; e->contact[1] = contact[1];
  e->reward = reward;
  /* ---- EpisodeWrapper.step */
  e->ep_steps += 1;
  real done_f = done ? 1.0 : 0.0;
  if (e->ep_steps >= c->episode_length) { e->truncation = 1 - done_f; done_f = 1.0; } else e->truncation = 0;
  real keep = 1 - e->episode_done;
  e->ep_sum_reward = (e->ep_sum_reward + reward) * keep;
  e->ep_length = (e->ep_length + 1) * keep;
  for (int k = 0; k < ODKO_NMETRIC; k++) e->ep_metrics[k] = (e->ep_metrics[k] + e->metrics[k]) * keep;
  e->episode_done = done_f;
  e->done = done_f;
  e->rng_ctr += 1;
  /* ---- AutoReset.step epilogue: data, obs <- first_* where done (info is NOT reset) */
  if (done_f != 0 && c->autoreset) {
    memcpy(d->qpos, e->first_qpos, sizeof(e->first_qpos)); memcpy(d->qvel, e->first_qvel, sizeof(e->first_qvel));
    memcpy(d->qacc_warmstart, e->first_warmstart, sizeof(e->first_warmstart));
    memcpy(e->obs, e->first_obs, sizeof(e->first_obs)); memcpy(e->priv, e->first_priv, sizeof(e->first_priv));
  }
}

/* ------------------------------------------------------------------ multi-threaded CPU baseline */
typedef struct { const odko_model* m; const odko_prm* prm; int e0, e1, nsteps, nwarm; uint32_t seed; pthread_barrier_t* bar; } mt_arg;

static void* mt_worker(void* p) {
  mt_arg* a = (mt_arg*)p;
  int n = a->e1 - a->e0;
  odko_env** envs = (odko_env**)malloc(sizeof(odko_env*) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    envs[i] = odko_env_new(a->m, a->prm, NULL);
    envs[i]->cfg.noise_level = 0; envs[i]->cfg.push_enable = 0; /* BASELINE.md protocol */
    odko_env_reset(envs[i], a->seed, (uint32_t)(a->e0 + i));
  }
  for (int t = 0; t < a->nwarm + a->nsteps; t++) {
    if (t == a->nwarm) pthread_barrier_wait(a->bar);
    for (int i = 0; i < n; i++) {
      real act[ODKO_MAXU];
      for (int u = 0; u < a->m->nu; u++)
        act[u] = 2 * (real)odko_rng_uniform(envs[i]->key[0] ^ 0x41435431u, envs[i]->key[1], (uint32_t)t, (uint32_t)u) - 1;
      odko_env_step(envs[i], act);
    }
  }
  pthread_barrier_wait(a->bar);
  for (int i = 0; i < n; i++) odko_env_free(envs[i]);
  free(envs);
  return NULL;
}

double odko_rollout_mt(const odko_model* m, const odko_prm* prm, int nenv, int nsteps, int nwarm, int nthreads, uint32_t seed) {
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  mt_arg* args = (mt_arg*)malloc(sizeof(mt_arg) * nthreads);
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, nthreads + 1);
  for (int t = 0; t < nthreads; t++) {
    args[t].m = m; args[t].prm = prm; args[t].nsteps = nsteps; args[t].nwarm = nwarm; args[t].seed = seed; args[t].bar = &bar;
    args[t].e0 = (int)((long)nenv * t / nthreads); args[t].e1 = (int)((long)nenv * (t + 1) / nthreads);
    pthread_create(&th[t], NULL, mt_worker, &args[t]);
  }
  struct timespec t0, t1;
  pthread_barrier_wait(&bar); /* all warmed up */
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_barrier_wait(&bar); /* all done */
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&bar);
  free(th); free(args);
  double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  return (double)nenv * nsteps / sec;
}

/* ------------------------------------------------------------------ a vector of envs stepped by a pool of threads (tools: the
 * height-field hypothesis sweep trains a policy against the ORACLE -- tools/hfield_variants.py -- which needs the brax-style batched
 * reset / step surface; float32 in and out whatever `real` is) */
struct odko_vec { const odko_model* m; const odko_prm* prm; int n; odko_env** e; };
typedef struct { odko_vec* v; int e0, e1, reset; uint32_t seed, offset; const float* act; float *obs, *priv, *rew, *done, *trunc, *met; } vec_arg;

odko_vec* odko_vec_new(const odko_model* m, const odko_prm* prm, int n, int standing) {
  odko_vec* v = (odko_vec*)malloc(sizeof(odko_vec));
  v->m = m; v->prm = prm; v->n = n; v->e = (odko_env**)malloc(sizeof(odko_env*) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { v->e[i] = odko_env_new(m, prm, NULL); if (standing) odko_env_set_standing(v->e[i]); }
  return v;
}
void odko_vec_free(odko_vec* v) { if (!v) return; for (int i = 0; i < v->n; i++) odko_env_free(v->e[i]); free(v->e); free(v); }
odko_env* odko_vec_env(odko_vec* v, int i) { return (i >= 0 && i < v->n) ? v->e[i] : NULL; }
static void* vec_worker(void* p) {
  vec_arg* a = (vec_arg*)p;
  const int nu = a->v->m->nu;
  for (int i = a->e0; i < a->e1; i++) {
    odko_env* e = a->v->e[i];
    if (a->reset) odko_env_reset(e, a->seed, a->offset + (uint32_t)i);
    else { real act[ODKO_MAXU]; for (int u = 0; u < nu; u++) act[u] = (real)a->act[(size_t)i * nu + u]; odko_env_step(e, act); }
    const int no = odko_env_nobs(e), np = odko_env_npriv(e);
    for (int k = 0; k < no; k++) a->obs[(size_t)i * no + k] = (float)e->obs[k];
    for (int k = 0; k < np; k++) a->priv[(size_t)i * np + k] = (float)e->priv[k];
    if (a->rew) a->rew[i] = (float)e->reward;
    if (a->done) a->done[i] = (float)e->done;
    if (a->trunc) a->trunc[i] = (float)e->truncation;
    if (a->met) for (int k = 0; k < ODKO_NMETRIC; k++) a->met[(size_t)i * ODKO_NMETRIC + k] = (float)e->metrics[k];
  }
  return NULL;
}
static void vec_run(vec_arg proto, int nthreads) {
  const int n = proto.v->n;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n > 0 ? n : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
  vec_arg* args = (vec_arg*)malloc(sizeof(vec_arg) * nthreads);
  for (int t = 0; t < nthreads; t++) {
    args[t] = proto; args[t].e0 = (int)((long)n * t / nthreads); args[t].e1 = (int)((long)n * (t + 1) / nthreads);
    pthread_create(&th[t], NULL, vec_worker, &args[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(args);
}
void odko_vec_reset(odko_vec* v, uint32_t seed, uint32_t env_offset, int nthreads, float* obs, float* priv) {
  vec_arg a; memset(&a, 0, sizeof(a));
  a.v = v; a.reset = 1; a.seed = seed; a.offset = env_offset; a.obs = obs; a.priv = priv;
  vec_run(a, nthreads);
}
void odko_vec_step(odko_vec* v, const float* actions, int nthreads, float* obs, float* priv, float* reward, float* done, float* trunc, float* metrics) {
  vec_arg a; memset(&a, 0, sizeof(a));
  a.v = v; a.act = actions; a.obs = obs; a.priv = priv; a.rew = reward; a.done = done; a.trunc = trunc; a.met = metrics;
  vec_run(a, nthreads);
}
