/* odk.h -- C-ABI of the MI355X-native Open Duck env engine (libodk.so).
 *
 * Drop-in boundary for the physics + Joystick-task hot path of apirrone/Open_Duck_Playground
 * (SURVEY.md section 8b).  The reference has no FFI: its boundary is the Python/JAX API
 *     mjx.put_model(mj_model)                                  playground/open_duck_mini_v2/base.py:61
 *     Joystick.reset(rng) -> State                              playground/open_duck_mini_v2/joystick.py:206
 *     Joystick.step(State, action) -> State                     playground/open_duck_mini_v2/joystick.py:323
 *     Standing.reset / Standing.step (env_kind = ODK_ENV_STANDING) playground/open_duck_mini_v2/standing.py:200,316
 *       (which calls mjx_env.init :258 and mjx_env.step(model, data, motor_targets, n_substeps) :420)
 *     randomize.domain_randomize(model, rng) -> batched fields  playground/common/randomize.py:26-146
 *     wrapper.wrap_for_brax_training (Vmap/Episode/AutoReset)   playground/common/runner.py:117
 * Each entry point below names the call it replaces.  Plain pointers and sizes only; all
 * `*_dev` pointers are device (HIP) addresses owned by the caller; calls are ordered on the caller's
 * stream (`hipStream_t` passed as void*), never synchronise the host, and never allocate in step.
 *
 * Every function returns 0 on success or a negative odk_status; odk_last_error() gives the
 * thread-local message.  Numerical failure inside an env is NOT an error: it yields NaN ->
 * done = 1 for that env (reference joystick.py:483-485).
 */
#ifndef ODK_H
#define ODK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct odk_model odk_model;
typedef struct odk_batch odk_batch;

enum odk_status {
  ODK_OK = 0,
  ODK_ERR_INVALID = -1,      /* bad argument / malformed blob */
  ODK_ERR_UNSUPPORTED = -2,  /* model shape or option the kernels were not built for */
  ODK_ERR_HIP = -3,          /* HIP runtime failure */
  ODK_ERR_NOMEM = -4
};

/* Observation row strides of the duck (nu = 14 actuators).  For a robot with nu actuators (odk_model_obs_sizes): joystick.py:570-615
 * state 17 + 6 nu, privileged_state = state + 69 + 3 nu; standing.py:524-565 state 15 + 5 nu, privileged_state = state + 26 + 3 nu. */
#define ODK_NOBS 101     /* obs["state"]            joystick.py:570-589 */
#define ODK_NPRIV 212    /* obs["privileged_state"] joystick.py:596-615 */
#define ODK_NOBS_STANDING 85    /* standing.py:524-540 */
#define ODK_NPRIV_STANDING 153  /* standing.py:548-565 */
#define ODK_ENV_JOYSTICK 0
#define ODK_ENV_STANDING 1
#define ODK_NMETRIC 8    /* reward/cost terms (7) + swing_peak, joystick.py:304-311 */
#define ODK_NU 14        /* the duck's actuators; odk_model_dims reports a model's own count */

/* Environment configuration == default_config() of the reference (joystick.py:49-102). */
typedef struct {
  float ctrl_dt, action_scale, dof_vel_scale, max_motor_velocity;
  float noise_level, noise_gyro, noise_accelerometer, noise_gravity, noise_joint_vel;
  float qpos_noise_scale[16];
  float reward_scales[7];  /* tracking_lin_vel, tracking_ang_vel, torques, action_rate, stand_still, alive, imitation;
                              Standing: orientation, head_pos, torques, action_rate, stand_still, alive, (unused) */
  float tracking_sigma;
  float push_enable, push_interval_range[2], push_magnitude_range[2];
  float cmd_range[7][2];   /* lin_vel_x, lin_vel_y, ang_vel_yaw, neck_pitch, head_pitch, head_yaw, head_roll */
  int32_t use_imitation, use_motor_speed_limits;
  int32_t autoreset;       /* BraxAutoResetWrapper on/off */
  int32_t episode_length;  /* EpisodeWrapper */
  int32_t n_substeps;      /* ctrl_dt / sim_dt */
  int32_t lanes_per_env;   /* kernel geometry hint: 32 or 64 (0 = default); see odk_batch_lanes */
  int32_t env_kind;        /* ODK_ENV_JOYSTICK (joystick.py) or ODK_ENV_STANDING (standing.py): selects the obs layout
                              (101/212 vs 85/153 floats per env -- the output row strides) and the reward table */
  float reset_base_qvel;   /* half-range of the base velocity noise at reset: joystick.py:253 0.05, standing.py:247 0.5 */
  int32_t hfield_up_normals_only; /* BUILD-DEFINED opt-in, default 0 = the prism algorithm as recalled from MJX.  1: on a height-field
                              floor a prism pair's contacts count only when their normal points up (n_z > 0.5 in the field's frame): drops the
                              sideways contacts of prism side faces.  The reading under which the reference's rough-terrain task trains
                              (profiles/r4/hfield_variants.json); parity against the oracle's hfield_mode 3 */
} odk_env_config;

/* Caller-owned device outputs of reset/step (any pointer may be NULL to skip it). */
typedef struct {
  float* obs_dev;         /* [nenv, nobs]   (the duck: 101; odk_model_obs_sizes) */
  float* priv_dev;        /* [nenv, npriv]  (the duck: 212) */
  float* reward_dev;      /* [nenv] */
  float* done_dev;        /* [nenv] */
  float* truncation_dev;  /* [nenv] */
  float* metrics_dev;     /* [nenv, 8] */
} odk_outputs;

/* Per-env randomised model fields == the 8 fields of randomize.py:119-144 (geom_friction is a
 * visual geom in the reference and therefore has no physical effect; it is not taken). */
enum odk_param {
  ODK_PARAM_BODY_MASS = 0,        /* [nenv, nbody] */
  ODK_PARAM_BODY_IPOS_TORSO = 1,  /* [nenv, 3]   body_ipos[TORSO_BODY_ID=1] */
  ODK_PARAM_DOF_FRICTIONLOSS = 2, /* [nenv, nu]  actuated dofs */
  ODK_PARAM_DOF_ARMATURE = 3,     /* [nenv, nu] */
  ODK_PARAM_QPOS0 = 4,            /* [nenv, nu]  actuated joints */
  ODK_PARAM_KP = 5                /* [nenv, nu]  gainprm[:,0]; biasprm[:,1] = -kp */
};

const char* odk_last_error(void);
void odk_default_config(odk_env_config* cfg);
/* default_config() of reference standing.py:44-100 (incl. USE_IMITATION_REWARD = False, no motor speed limit) */
void odk_default_config_standing(odk_env_config* cfg);
/* row strides of the obs / privileged_state outputs for an env kind -- of the duck (14 actuators) */
void odk_obs_sizes(int env_kind, int* nobs, int* npriv);

/* mjx.put_model: parse a ModelBlob (open_duck_playground_amd/model.py; written by the MJCF compiler mjcf.py).
 * Colliders the kernels take: two feet -- convex meshes or boxes (cgeom_type 7: hull vertices + outward triangles), or spheres /
 * capsules (cgeom_type 2 / 3 with the optional record cgeom_size[ncgeom][3]: radius, half length) -- and one floor, a plane (0) or a
 * height field (1: hfield_data + hfield_size).  ODK_ERR_UNSUPPORTED for anything else (other foot types, a hull foot beside a
 * primitive one on a height field, feet wider than two height-field cells, hulls with more than 17 vertices / 30 faces / 48 edges
 * or faces of more than four vertices).
 * Solver options read from the blob: opt_iterations / opt_ls_iterations, opt_impratio, and the optional opt_cone (0 pyramidal, 1 elliptic:
 * the elliptic-cone instantiations of the kernels -- hull feet, 32 lanes per env; sphere / capsule feet refuse it).  The optional eq_*
 * records (<equality>): joint couplings between two hinges of one serial chain, and connect / weld constraints whose two bodies lie on one
 * root-to-leaf path of the tree (or body2 = the world) or on the two foot chains (a closed loop) -- at most two with nine rows --, are
 * taken for the third and fourth model shapes; every other
 * ACTIVE equality is refused by name. */
int odk_model_load(const void* blob, uint64_t len, odk_model** out);
void odk_model_free(odk_model* m);
int odk_model_dims(const odk_model* m, int* nq, int* nv, int* nu, int* nbody);
/* row strides of the obs / privileged_state outputs of THIS model's env kernels (what `observation_size` of the reference's env reports,
 * base.py:277-291 / joystick.py:570-615, for a robot with the model's actuator count): the duck 101 / 212 (Standing 85 / 153), a robot
 * with 12 actuators 89 / 194.  The per-env layout is the reference's with nu in place of 14. */
int odk_model_obs_sizes(const odk_model* m, int env_kind, int* nobs, int* npriv);

/* Twin dofs (backlash joints: a hinge declared right after another hinge on the same body, same anchor and axis) share
 * their motion column, so the kernels keep the inertia / Newton Hessian on the REDUCED tree with the twins merged
 * (csrc/odk_model.h DevModel::paired).  Reports that reduction: paired (0 / 1), reduced dof count, entries of the reduced
 * tree layout and of its virtual (Hessian) tree, and per reduced dof the main dof and the twin dof (-1: none); arrays of
 * >= 32 ints, any pointer may be NULL.  Host-only (no GPU needed). */
int odk_model_reduced(const odk_model* m, int* paired, int* nvr, int* nMr, int* nHr, int* red_main, int* red_twin);
/* floats of LDS one env occupies in the fused step kernel (8 single-wave workgroups of two envs per CU need <= 2560) */
int odk_model_env_lds_floats(const odk_model* m);

/* One batch of `nenv` environments resident on HIP device `device`.  `prm_table` is the host
 * [nx,ny,nth,40,16] float32 reference-motion table (poly_reference_motion.py), grids in float64. */
int odk_batch_create(const odk_model* m, const odk_env_config* cfg, int nenv, int device, const float* prm_table,
                     const double* dxs, int nx, const double* dys, int ny, const double* dths, int nth, const double* ranges6,
                     int nsteps_in_period, odk_batch** out);
void odk_batch_destroy(odk_batch* b);
int odk_batch_set_config(odk_batch* b, const odk_env_config* cfg);

/* domain_randomize: per-env model fields, host pointer, copied synchronously */
int odk_batch_set_param(odk_batch* b, int param, const float* host_values, int count_per_env);

/* Joystick.reset (vmapped) + wrapper resets.  Env e uses key(seed, env_id_offset + e). */
int odk_reset(odk_batch* b, uint32_t seed, uint32_t env_id_offset, const odk_outputs* outs, void* stream);

/* AutoReset.step -> Episode.step -> Joystick.step for all envs; action_dev is [nenv, nu]. */
int odk_step(odk_batch* b, const float* action_dev, const odk_outputs* outs, void* stream);

/* mjx_env.step alone (physics only, n_substeps, ctrl = ctrl_dev [nenv, nu]); for parity tests */
int odk_physics_step(odk_batch* b, const float* ctrl_dev, int n_substeps, void* stream);

/* state access (host pointers, synchronous): qpos [nenv,nq], qvel [nenv,nv], qacc_warmstart [nenv,nv] */
int odk_batch_get_state(odk_batch* b, float* qpos, float* qvel, float* warm);
int odk_batch_set_state(odk_batch* b, const float* qpos, const float* qvel, const float* warm);
/* debug read-back of the last forward pass of every env: sensordata [nenv,46], actuator_force [nenv,14],
 * contact_dist [nenv,12], qacc [nenv,nv]  (any may be NULL) */
int odk_batch_get_debug(odk_batch* b, float* sensordata, float* actuator_force, float* contact_dist, float* qacc);
/* debug: LDS image (floats) of every env's last forward pass, taken at reset / physics_step and, when
 * odk_set_debug_dump(1), at step; odk_lds_offset names the arrays inside it (csrc/odk_kernels.h Shape) */
void odk_set_debug_dump(int on);
int odk_batch_lds_size(const odk_batch* b);
int odk_batch_get_lds(odk_batch* b, float* host_image);
int odk_lds_offset(const odk_batch* b, const char* name);
/* lanes per env the batch's kernels really run (odk_env_config.lanes_per_env is a hint: elliptic cones, height-field floors and robots that are
 * not the duck exist at 32 lanes per env only) */
int odk_batch_lanes(const odk_batch* b);
/* raw per-env info record (floats, layout in csrc/odk_engine.hip) for tests */
int odk_batch_record_size(const odk_batch* b);
int odk_batch_get_records(odk_batch* b, float* host_records);
/* writes the records back (synchronous): restores a saved batch, or presets carried `info` fields -- e.g. info["step"] = 500 so
 * that the next step resamples the command (joystick.py:456-466) */
int odk_batch_set_records(odk_batch* b, const float* host_records);
/* where a field of the carried state lives inside a record: names are the keys of the reference's `info` dict (joystick.py:278-302:
 * "rng", "step", "command", "last_act", "last_last_act", "last_last_last_act", "motor_targets", "feet_air_time", "last_contact",
 * "swing_peak", "push", "push_step", "push_interval_steps", "action_history", "imu_history", "imitation_i"), the wrapper's additions
 * ("steps", "truncation", "episode_done", "episode_metrics/sum_reward", "episode_metrics/length", "episode_metrics/reward_terms")
 * and the physics state ("qpos", "qvel", "qacc_warmstart").  *offset / *count in 4-byte words from the start of the record;
 * *kind = 0 float32, 1 int32 / uint32, 2 bit mask in one int32 (last_contact: bit f = foot f).  `current_reference_motion` and
 * `imitation_phase` are functions of imitation_i and the command and are not carried.  Returns ODK_ERR_INVALID for an unknown name. */
int odk_record_field(const odk_batch* b, const char* name, int* offset, int* count, int* kind);

/* ---- learner-side kernels (csrc/odk_learner.hip): the element-wise halves of one PPO minibatch step.  The
 * reference reaches them through brax ppo.train (common/runner.py:104-118): ppo.losses.compute_gae /
 * compute_ppo_loss and optax.chain(clip_by_global_norm, adam).  All stream-ordered, graph-capturable. ---- */

/* GAE over row-major [B, T] device arrays: vs and advantages out; truncation / termination are flags (0 = clear, any
 * other value = set);
 * bootstrap is [B].  adv_stats (may be NULL) receives {mean, 1/(std+1e-8)} of the advantages (ddof 0). */
int odk_gae(const float* truncation_dev, const float* termination_dev, const float* rewards_dev, const float* values_dev,
            const float* bootstrap_dev, float* vs_dev, float* adv_dev, float* adv_stats_dev, int B, int T, float lambda_,
            float discount, void* stream);

/* PPO loss head, forward and backward in one launch.  logits [n, 2*action_size] = (loc | raw scale) of the
 * tanh-normal policy, noise [n, action_size] ~ N(0,1) for the sampled entropy term, adv_stats from odk_gae (NULL:
 * no advantage normalisation).  Writes grad_scale * dLoss/dlogits and grad_scale * dLoss/dbaseline, and ADDS
 * (total, policy, value, entropy) loss to losses[0..3]: the caller zeroes them, per step or -- for the mean over an
 * epoch of steps -- once per epoch. */
int odk_ppo_head(const float* logits_dev, const float* raw_action_dev, const float* old_log_prob_dev, const float* adv_dev,
                 const float* adv_stats_dev, const float* vs_dev, const float* baseline_dev, const float* noise_dev,
                 float* dlogits_dev, float* dbaseline_dev, float* losses_dev, int n, int action_size, float clipping_epsilon,
                 float entropy_cost, float grad_scale, void* stream);

/* clip_by_global_norm(max_grad_norm; <= 0 disables) + Adam on flat buffers of n floats.  acc_dev[ODK_ADAM_ACC_FLOATS]
 * is scratch owned by the caller: acc[0] = squared gradient norm of this call, acc[1] = step count (zero it once),
 * acc[2..] = per-block partial sums (the norm is reduced in a fixed order: data-parallel replicas stay bit-identical). */
/* Rollout sampling of the tanh-normal policy (brax NormalTanhDistribution, as odk_ppo_head): logits [n, 2A] = (loc | raw_scale),
 * noise [n, A] standard normal (zeros: the mode).  raw_action = loc + (softplus(raw_scale) + 0.001) noise, action = tanh(raw_action),
 * log_prob[n] = log density of the action.  One launch instead of ~15 element-wise ones per rollout step. */
int odk_policy_sample(const float* logits_dev, const float* noise_dev, float* raw_action_dev, float* action_dev, float* log_prob_dev, int n,
                      int action_size, void* stream);

#define ODK_ADAM_MAX_PARTIALS 1024
#define ODK_ADAM_ACC_FLOATS (2 + ODK_ADAM_MAX_PARTIALS)
int odk_adam_clip(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr,
                  float b1, float b2, float eps, float max_grad_norm, void* stream);

/* dz = dh * silu'(z) over row-major [n, w] and colsum[c] = sum_r dz[r, c] (the bias gradient of the layer below), fixed
 * summation order.  partial_dev: scratch of ceil(n / 64) * w floats.  colsum_dev may be NULL: the per-tile partial sums
 * then stay in partial_dev for odk_colsum_finalize. */
int odk_silu_bwd_colsum(const float* dh_dev, const float* z_dev, float* dz_dev, float* colsum_dev, float* partial_dev, int n, int w,
                        void* stream);

/* partial[tile, c] = sum of x[r, c] over the 64 rows of the tile (x row-major [n, w]): the first half of a column sum whose
 * second half is odk_colsum_finalize. */
int odk_colsum_partial(const float* x_dev, float* partial_dev, int n, int w, void* stream);

/* colsum[f][c] = sum over the ceil(n / 64) tile rows of partial[f][tile, c] for up to 8 layers in ONE launch (same fixed
 * order as odk_silu_bwd_colsum's own fold).  partial_dev / colsum_dev / widths are HOST arrays. */
int odk_colsum_finalize(const float* const* partial_dev, float* const* colsum_dev, const int* widths, int count, int n, void* stream);

/* Weight gradients of up to 8 dense layers in one launch on the f32 matrix cores (v_mfma_f32_32x32x2_f32):
 *   out[out_off[l] + i * n_in[l] + j] = sum over the nrows[l] minibatch rows s of dz[l](s, i) * h[l](s, j)      (= dz^T h, torch Linear weight layout)
 * dz[l] / h[l] are in the QUAD-ROW layout the fused network kernels write: [nrows / 4][width][4], element (s, f) at
 * ((s / 4) * width + f) * 4 + s % 4 (16-byte aligned; nrows[l] a multiple of 8 and >= 16 * kslices; rows past the batch hold zeros
 * in at least one of the two operands).  out_dev is the flat gradient buffer.  The rows are split into kslices slices (a
 * multiple of 8); ws_dev is a workspace of kslices * ws_stride floats laid out like out_dev (ws_stride >= every out_off +
 * n_out * n_in, a multiple of 4; every out_off and n_out * n_in a multiple of 4; ws_dev and out_dev 16-byte aligned); the slices
 * are folded in a fixed order, so the result is bit-reproducible.  dz_dev / h_dev / n_out / n_in / out_off / nrows are HOST arrays. */
/* Optional extra work of odk_dw_gemm's slice-fold launch (the launch that finishes the gradient): the bias gradients
 * bias_grad[f][c] = sum over the nblk[f] tile rows of bias_partial[f][tile, c] (what odk_colsum_fold does), and -- when
 * sq_partials_dev is not NULL -- per-block partial sums of the squared norm of everything the launch wrote (all weight and bias
 * gradients), for odk_adam_clip_packed(norm_blocks = nblocks): sq_partials_dev[0 .. nblocks), nblocks <= ODK_ADAM_MAX_PARTIALS is
 * returned in the struct; step_counter_dev (may be NULL) is incremented by 1.  Host struct; the pointers inside are device pointers. */
typedef struct odk_grad_finish {
  const float* bias_partial[8];
  float* bias_grad[8];
  int width[8], nblk[8];
  int nbias;
  float* sq_partials_dev;
  float* step_counter_dev;
  int nblocks;               /* out */
} odk_grad_finish;
int odk_dw_gemm(const float* const* dz_dev, const float* const* h_dev, const int* n_out, const int* n_in, const long long* out_off, int nlayers,
                const int* nrows, int kslices, float* ws_dev, long long ws_stride, float* out_dev, odk_grad_finish* finish /* may be NULL */,
                void* stream);

/* ---- fused policy / value networks (csrc/odk_mlp.hip): a swish MLP  n_in -> 512 -> 256 -> 128 -> n_out  (brax ppo.networks as
 * configured by the reference, common/runner.py:86-118) forward in ONE launch and its backward-data chain in ONE launch, on the
 * f32 matrix cores; a workgroup keeps a tile of 16 samples in LDS for all layers.  One or two networks per launch (policy and
 * value side by side).  Everything row-major float32 on the device.
 * The kernels read the weights from PACKED copies (16-byte pieces of four consecutive reduction indices per column):
 *   forward copy of W [n_out, n_in]:  [pad16(n_in) / 4][n_out][4],  element (o, i) at ((i / 4) * n_out + o) * 4 + i % 4
 *   backward copy:                    [pad16(n_out) / 4][n_in][4],  element (o, i) at ((o / 4) * n_in + i) * 4 + o % 4
 * with pad16(k) = k rounded up to a multiple of 16 and the padding zero (the caller zeroes the buffers once; odk_pack_weights /
 * odk_adam_clip_packed write the weights' elements only). */
#define ODK_MLP_H1 512
#define ODK_MLP_H2 256
#define ODK_MLP_H3 128
#define ODK_MLP_MAX_IN 224
#define ODK_MLP_TILE 16        /* samples per workgroup */
typedef struct odk_mlp_desc {
  const float* x;            /* [n, n_in] network input */
  const float* in_mean;      /* forward, optional (both or none): the input is normalised on load, x <- (x - in_mean) / in_std, per column */
  const float* in_std;
  const float* wf[4];        /* forward-packed weights of the four layers (16-byte aligned) */
  const float* wb[4];        /* backward-packed weights (wb[0] is not read: the input's gradient is never formed) */
  const float* b[4];         /* biases */
  /* training buffers, all in the QUAD-ROW layout [np / 4][width][4] with np = n rounded up to a multiple of ODK_MLP_TILE = 16 (the kernels work on 16-sample tiles and write ceil(n / 16) tile rows of
   * bias_partial; element (s, f) at
   * ((s / 4) * width + f) * 4 + s % 4; the operand layout of odk_dw_gemm); rows n .. np - 1 are written as zeros in h / g / dz / doutp */
  float* xp;                 /* forward out: quad-row copy of x, width n_in (rows past n: copies of row n - 1) */
  float* h[3];               /* forward out: swish(z_l), width H_l; all of xp / h / g NULL: inference only, nothing but `out` is written */
  float* g[3];               /* forward out / backward in: swish'(z_l) */
  float* out;                /* forward out: [n, n_out] row-major */
  const float* dout;         /* backward in: dLoss/dout [n, n_out] row-major */
  float* doutp;              /* backward out: quad-row copy of dout, width n_out */
  float* dz[3];              /* backward out: dLoss/dz_l, width H_l */
  float* bias_partial[4];    /* backward out: per-tile column sums of dz_l (l = 3: of dout), [ceil(n / 16), width_l]; odk_colsum_fold finishes them */
  int n, n_in, n_out;        /* n_in <= ODK_MLP_MAX_IN, n_out <= 32 */
  /* Optional row sources of the forward pass (row_idx NULL: row r of the input is x[r], as above) -- the minibatch gather of brax's
   * sgd_step (jnp.take of the shuffled trajectories; reference common/runner.py:104-118 -> brax ppo.train) folded into the load, so that
   * no gathered copy of the observations is ever written:
   *   rows r <  n_main:  x[row_idx[k B + r / traj_len] * traj_len + r % traj_len]     x = the WHOLE rollout, [n_traj * traj_len, n_in]
   *   rows r >= n_main:  x_tail[row_idx[k B + r - n_main]]                              x_tail = [n_traj, n_in] (the bootstrap observations)
   * with B = n_main / traj_len trajectories per minibatch and k = *cursor (device int: which minibatch of the schedule row_idx holds;
   * NULL: 0).  An index outside [0, n_traj) is never dereferenced: that row reads as NaN and the step's losses say so. */
  const long long* row_idx;
  const int* cursor;
  const float* x_tail;
  int traj_len, n_main, n_traj;
} odk_mlp_desc;
int odk_mlp_forward(const odk_mlp_desc* nets, int count, void* stream);
int odk_mlp_backward(const odk_mlp_desc* nets, int count, void* stream);
/* tools: device buffer of 4 x 1024 int64 receiving, per single-wave workgroup of odk_dw_gemm's matrix launch, its start / end on the
 * 100 MHz wall clock and the HW_ID / XCC_ID registers (where it ran); NULL: off */
void odk_dw_set_profile(long long* dev);
/* tools: device buffer of 32 int64 receiving the forward kernel's phase timestamps (shader clock, workgroup 0); NULL: off */
void odk_mlp_set_profile(long long* stamps_dev);
/* tools: device buffer of 4 x (workgroups of a network launch) int64 receiving every workgroup's start / end on the 100 MHz wall
 * clock, HW_ID | XCC_ID << 32 and its shader-clock cycles (forward and backward launches alike); NULL: off */
void odk_mlp_set_wg_profile(long long* dev);
/* tools: diagnostic variants of the network launches (results are WRONG): bit 0 = every weight load re-reads its phase's first group
 * (what the launch costs when the weights come from the L1), bit 1 = the MFMAs are skipped (what the operand traffic alone costs), bit 2 = no
 * operand loads inside the loops (what the MFMAs alone cost), bit 3 = no activation stores, bit 4 = every second weight load skipped.  Only the
 * diagnostic build (make libodk_mlpdiag.so) honours them */
void odk_mlp_set_diag(int bits);
/* Where up to 8 weight matrices sit in a flat parameter buffer (float offset `off`, torch layout [rows = n_out, cols = n_in]) and
 * in the packed buffers (float offsets, multiples of 4; bwd_off < 0: no backward copy of that weight). */
typedef struct odk_weight_table {
  int count;
  long long off[8];
  int rows[8], cols[8];
  long long fwd_off[8], bwd_off[8];
} odk_weight_table;
/* (re)builds the packed copies from the parameters */
int odk_pack_weights(const float* params_dev, long long n, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                     const odk_weight_table* table, void* stream);
/* odk_adam_clip that also keeps the packed copies current (every updated weight is written to all of its places).
 * norm_blocks > 0: acc[2 .. 2 + norm_blocks) already hold the partial sums of the squared gradient norm and acc[1] the advanced step
 * count (odk_dw_gemm with an odk_grad_finish whose sq_partials_dev = acc + 2, step_counter_dev = acc + 1): no norm launch of its own. */
int odk_adam_clip_packed(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr, float b1,
                         float b2, float eps, float max_grad_norm, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                         const odk_weight_table* table, int norm_blocks, void* stream);
/* The same launch with the END-OF-STEP duties of an indexed minibatch step folded in (all optional, NULL / 0 = off), so that a step is
 * six launches and nothing else:
 *   cursor_dev:        *cursor_dev += 1 -- the next step's launches read the next minibatch of the schedule (odk_mlp_desc.cursor,
 *                      odk_ppo_gae_head) without any host-side call between two graph replays;
 *   loss_partials_dev: [n_loss_partials][4] per-workgroup sums of (total, policy, value, entropy) left by odk_ppo_gae_head, folded in
 *                      workgroup order into losses_dev[0..3] += ... (a fixed order; as float atomics the 4 x 320 additions on four
 *                      addresses cost the head launch ~4 us of serialisation). */
typedef struct odk_step_tail {
  int* cursor_dev;
  const float* loss_partials_dev;
  int n_loss_partials;
  float* losses_dev;
} odk_step_tail;
int odk_adam_clip_packed_tail(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, float* acc_dev, long long n, float lr, float b1,
                              float b2, float eps, float max_grad_norm, float* fwd_packed_dev, long long n_fwd, float* bwd_packed_dev, long long n_bwd,
                              const odk_weight_table* table, int norm_blocks, const odk_step_tail* tail /* may be NULL */, void* stream);

/* GAE + advantage statistics + the PPO loss head in ONE launch, reading the rollout through the minibatch's trajectory indices (what
 * odk_gather_rows + odk_gae + odk_ppo_head do as three launches on gathered copies; same arithmetic, same summation orders; brax
 * ppo.losses.compute_gae / compute_ppo_loss).  Every workgroup recomputes the B x T recursion in LDS (B * T <= 5120, B <= 1024) -- the
 * advantage statistics are a global quantity and a launch boundary costs more than 80 redundant copies of a 100 KFLOP scan -- then
 * works its own 64 samples.  Host struct; every pointer inside is a device pointer.
 *   sample s = b T + t of minibatch k = *cursor (NULL: 0) is step t of trajectory j = row_idx[k B + b] of the rollout;
 *   logits [n, 2A], values [n + B] (baselines, then the B bootstrap values), n = B T: the network outputs of THIS minibatch;
 *   raw_action [n_traj, T, A], old_log_prob / reward / termination / truncation [n_traj, T]: the whole rollout;
 *   noise: the entropy sample of minibatch k is noise[k n A ..];  dlogits [n, 2A], dvalues [n] out;
 *   losses: loss_partials != NULL: workgroup w writes its sums of (total, policy, value, entropy) to loss_partials[4 w ..] (ODK_GAE_HEAD_SAMPLES
 *   samples per workgroup: ceil(n / ODK_GAE_HEAD_SAMPLES) entries; odk_adam_clip_packed_tail folds them); otherwise losses[4] += ... by atomics;
 *   adv_out / vs_out [n], stats_out[2] (mean, 1 / (std + 1e-8)): optional outputs (written by workgroup 0), may be NULL. */
#define ODK_GAE_HEAD_SAMPLES 32
typedef struct odk_gae_head_args {
  const float *logits, *values, *raw_action, *old_log_prob, *reward, *termination, *truncation, *noise;
  const long long* row_idx;
  const int* cursor;
  float *dlogits, *dvalues, *losses, *loss_partials, *adv_out, *vs_out, *stats_out;
  int B, T, action_size, n_traj, normalize_advantage;
  float gae_lambda, discount, clipping_epsilon, entropy_cost, grad_scale;
} odk_gae_head_args;
int odk_ppo_gae_head(const odk_gae_head_args* args, void* stream);

/* Column moments of a row-major float32 matrix x [rows, w] in ONE pass, accumulated in float64: partial_dev [slices][2][w] doubles receives, per row
 * slice, the column sums and the column sums of squares (slice s takes rows s, s + slices, ...; fixed order inside a slice); the caller folds the
 * slices (any fixed-order sum: they are few).  The observation normaliser's batch statistics (brax running_statistics.update as reached through
 * reference common/runner.py:104-118: normalize_observations=True) without a float64 copy of the 10^7-element rollout.  slices <= 1024. */
int odk_col_moments(const float* x_dev, long long rows, int w, int slices, double* partial_dev, void* stream);
/* brax running_statistics.update on those moments, one launch: folds the slices of partial_dev (in slice order) into the batch's column sums s
 * and sums of squares s2 over `rows` rows, then, in float64,
 *   count' = count + rows;  mean' = mean + (s / rows - mean) rows / count';  summed_variance' = summed_variance + (s2 - s (mean + mean') + rows mean mean');
 *   std = clamp(sqrt(max(summed_variance' / count', 0)), std_min, std_max)
 * count_dev: one double; mean / summed_variance / std: float32 [w], updated in place. */
int odk_moments_update(const double* partial_dev, int slices, int w, long long rows, double* count_dev, float* mean_dev, float* summed_variance_dev,
                       float* std_dev, float std_min, float std_max, void* stream);

/* colsum[f][c] = sum over the nblk[f] tile rows of partial[f][tile, c] for up to 8 layers in one launch (fixed order);
 * partial_dev / colsum_dev / widths / nblk are HOST arrays */
int odk_colsum_fold(const float* const* partial_dev, float* const* colsum_dev, const int* widths, const int* nblk, int count, void* stream);

/* dst[f][b, :] = src[f][idx[b], :] for up to 10 row-major float fields in one launch (minibatch gather of the rollout).
 * src_dev / dst_dev / row_floats are HOST arrays of device pointers / row lengths; idx_dev is int64 on the device, nrows
 * entries; every indexed source has src_rows rows.  An index outside [0, src_rows) is never dereferenced: its destination
 * rows are filled with NaN.  direct_base (HOST array, may be NULL): direct_base[f] >= 0 makes field f a plain block copy,
 * dst[f][b, :] = src[f][direct_base[f] + b, :] (no index; e.g. this step's slice of a noise pool) -- the caller guarantees
 * the range.  Buffers of fields whose row length is a multiple of 4 must be 16-byte aligned. */
int odk_gather_rows(const float* const* src_dev, float* const* dst_dev, const int* row_floats, const long long* direct_base, int nfields,
                    const long long* idx_dev, int nrows, long long src_rows, void* stream);

/* live timing of odk_step launches with HIP events on the launch stream: returns the average milliseconds per TIMED launch since
 * the last call (and resets the window).  enable: 0 = off, n > 0 = an event pair around every n-th launch from now on (the pair costs
 * the stream ~7 us of serialisation per timed launch: 1.2 % of an 8192-env step when every launch is timed) */
int odk_batch_timing(odk_batch* b, int enable, float* avg_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif
