/* odk_oracle_env.h -- CPU restatement of the Joystick (and Standing) task logic.  TEST INFRASTRUCTURE ONLY.
 * Follows reference playground/open_duck_mini_v2/joystick.py (reset :206-321, step :323-481,
 * termination :483-485, obs :487-620, reward :622-669, command :671-725), base.py index tables
 * (:63-125,154-231), common/rewards.py, open_duck_mini_v2/custom_rewards.py and
 * common/poly_reference_motion.py, plus the brax EpisodeWrapper / AutoResetWrapper semantics
 * that wrap it during training ([UPSTREAM-MEMORY]).
 * env_kind = 1 selects the Standing task (reference standing.py: reset :200-314, step :316-430, obs :436-567,
 * reward :569-608, command :610-660): same skeleton; obs without motor_targets / imitation phase (85 / 153 floats),
 * rewards orientation / head_pos in slots 0 / 1, stand_still with ignore_head, base-velocity reset range 0.5.
 *
 * Random numbers: the reference uses JAX threefry keys carried in info["rng"]; exact stream parity
 * is unpinned, so the build defines its own counter-based stream (threefry2x32
 * keyed per env, counter = env-step index, draw ids listed in odk_oracle_env.c) that oracle and HIP
 * kernel share bit-for-bit.
 */
#ifndef ODK_ORACLE_ENV_H
#define ODK_ORACLE_ENV_H
#include "odk_oracle.h"

/* capacity of the observation arrays: 17 + 6 nu / + 69 + 3 nu floats for nu <= ODKO_MAXU = 16 actuators (the duck, nu = 14: 101 / 212;
 * odko_env_nobs / odko_env_npriv give an env's own sizes, the rest of the arrays is zero) */
#define ODKO_NOBS 113
#define ODKO_NPRIV 230
#define ODKO_NMETRIC 8 /* tracking_lin_vel, tracking_ang_vel, torques, action_rate, stand_still, alive, imitation, swing_peak */

typedef struct {
  int nx, ny, nth, nsteps;
  double dxs[16], dys[16], dths[16], ranges[6];
  const float* table;    /* [nx][ny][nth][40][16], highest power first */
  const double* table64; /* optional float64 copy (golden-vector checks) */
} odko_prm;

typedef struct {
  /* configuration (reference joystick.py:49-102 default_config) */
  real ctrl_dt, action_scale, dof_vel_scale, max_motor_velocity;
  real noise_level, noise_gyro, noise_accelerometer, noise_gravity, noise_joint_vel, qpos_noise_scale[ODKO_MAXU];
  real reward_scales[7], tracking_sigma;
  real push_enable, push_interval_range[2], push_magnitude_range[2];
  real cmd_range[7][2];
  real use_imitation, use_motor_speed_limits, autoreset, episode_length, n_substeps;
  real env_kind;        /* 0 Joystick, 1 Standing */
  real reset_base_qvel; /* half-range of the base velocity at reset: joystick.py:253 0.05, standing.py:247 0.5 */
} odko_env_cfg;

typedef struct odko_env {
  const odko_model* m;
  const odko_prm* prm;
  odko_data d;
  odko_env_cfg cfg;
  /* index tables (reference base.py:63-125) */
  int act_qposadr[ODKO_MAXU], act_dofadr[ODKO_MAXU], backlash_qposadr[ODKO_MAXU];
  int imu_site, feet_site[2], floor_cgeom, feet_cgeom[2];
  int adr_gyro, adr_local_linvel, adr_accelerometer, adr_upvector, adr_global_angvel, adr_foot_linvel[2];
  /* info dict (joystick.py:278-302) */
  uint32_t key[2], rng_ctr;
  int step, push_step, push_interval_steps, imitation_i, last_contact[2];
  real command[7], last_act[ODKO_MAXU], last_last_act[ODKO_MAXU], last_last_last_act[ODKO_MAXU], motor_targets[ODKO_MAXU];
  real feet_air_time[2], swing_peak[2], push[2], action_history[3 * ODKO_MAXU], imu_history[9];
  real current_reference_motion[40], imitation_phase[2];
  /* wrappers */
  real ep_steps, truncation, episode_done, ep_sum_reward, ep_length, ep_metrics[ODKO_NMETRIC];
  real first_qpos[ODKO_MAXQ], first_qvel[ODKO_MAXV], first_warmstart[ODKO_MAXV], first_obs[ODKO_NOBS], first_priv[ODKO_NPRIV];
  /* outputs of the last reset/step */
  real obs[ODKO_NOBS], priv[ODKO_NPRIV], reward, done, metrics[ODKO_NMETRIC], contact[2];
  real motor_targets_out[ODKO_MAXU];
} odko_env;

#ifdef __cplusplus
extern "C" {
#endif
odko_prm* odko_prm_new(const float* table, const double* dxs, int nx, const double* dys, int ny, const double* dths, int nth,
                       const double* ranges, int nsteps);
void odko_prm_set_table64(odko_prm* p, const double* table64);
void odko_prm_free(odko_prm* p);
void odko_prm_index(const odko_prm* p, real dx, real dy, real dth, int* idx3);
void odko_prm_eval(const odko_prm* p, real dx, real dy, real dth, int i, real* out40);     /* float32 fma Horner */
void odko_prm_eval64(const odko_prm* p, double dx, double dy, double dth, int i, double* out40);

odko_env* odko_env_new(const odko_model* m, const odko_prm* prm, const odko_env_cfg* cfg);
void odko_env_free(odko_env* e);
odko_env* odko_env_clone(const odko_env* e);
real* odko_env_config(odko_env* e, const char* name, int* count);
real* odko_env_field(odko_env* e, const char* name, int* count);
int* odko_env_int(odko_env* e, const char* name, int* count);
odko_data* odko_env_data(odko_env* e);
void odko_env_reset(odko_env* e, uint32_t seed, uint32_t env_id);
void odko_env_step(odko_env* e, const real* action);

/* rng */
void odko_env_key(uint32_t seed, uint32_t env_id, uint32_t* key2);
float odko_rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr, uint32_t idx);

/* rewards (reference common/rewards.py, custom_rewards.py) */
real odko_reward_tracking_lin_vel(const real* cmd, const real* local_vel, real sigma);
real odko_reward_tracking_ang_vel(const real* cmd, const real* ang_vel, real sigma);
real odko_cost_torques(const real* torques, int n);
real odko_cost_action_rate(const real* act, const real* last_act, int n);
real odko_cost_stand_still(const real* cmd, const real* qpos, const real* qvel, const real* default_pose, int n);
real odko_cost_stand_still_legs(const real* cmd, const real* qpos, const real* qvel, const real* default_pose, int n); /* ignore_head=True */
real odko_cost_orientation(const real* torso_zaxis);
real odko_cost_head_pos(const real* joints_qpos, const real* cmd);
void odko_env_set_standing(odko_env* e); /* standing.py default_config on top of the defaults */
int odko_env_nobs(const odko_env* e);
int odko_env_npriv(const odko_env* e);
real odko_reward_imitation(const real* base_qpos, const real* base_qvel, const real* joints_qpos, const real* joints_qvel,
                           const real* contacts, const real* ref, const real* cmd);

/* multi-threaded random-action rollout for the CPU baseline: returns env-steps/sec */
double odko_rollout_mt(const odko_model* m, const odko_prm* prm, int nenv, int nsteps, int nwarm, int nthreads, uint32_t seed);
/* a vector of envs behind the batched reset / step surface, stepped by nthreads pthreads (tools only: tools/hfield_variants.py) */
typedef struct odko_vec odko_vec;
odko_vec* odko_vec_new(const odko_model* m, const odko_prm* prm, int n, int standing);
void odko_vec_free(odko_vec* v);
odko_env* odko_vec_env(odko_vec* v, int i);
void odko_vec_reset(odko_vec* v, uint32_t seed, uint32_t env_offset, int nthreads, float* obs, float* priv);
void odko_vec_step(odko_vec* v, const float* actions, int nthreads, float* obs, float* priv, float* reward, float* done, float* trunc, float* metrics);

#ifdef __cplusplus
}
#endif
#endif
