/* odk_oracle.h -- CPU restatement of the Open Duck hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a plain C, one-env-at-a-time, float64 restatement of
 *   - mjx.forward / mjx.step (third-party `mujoco-mjx`, UNPINNED in the reference:
 *     pyproject.toml:7-19 has lower bounds only) as called from
 *     reference playground/open_duck_mini_v2/joystick.py:258 (mjx_env.init) and :420
 *     (mjx_env.step, n_substeps=10), and
 *   - Joystick.reset / Joystick.step / _get_obs / _get_reward / sample_command
 *     (joystick.py:206-725), plus the brax Episode/AutoReset wrapper semantics
 *     ([UPSTREAM-MEMORY]).
 *
 * PARITY UNPINNED for the physics: no MuJoCo / MJX / JAX install exists in the build
 * container and the reference holds no golden vectors, so the physics half of this oracle is
 * pinned only by analytic invariants (tests/test_oracle_physics.py).  The reward and
 * reference-motion halves ARE pinned against fixtures generated from the reference's numpy
 * mirrors (tests/golden/, tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (open_duck_playground_amd/csrc) never links or calls it.
 */
#ifndef ODK_ORACLE_H
#define ODK_ORACLE_H

#include <stdint.h>

#ifndef ODKO_REAL
#define ODKO_REAL double
#endif
typedef ODKO_REAL real;

#define ODKO_MAXB 20   /* bodies */
#define ODKO_MAXJ 26   /* joints */
#define ODKO_MAXQ 32   /* nq */
#define ODKO_MAXV 32   /* nv */
#define ODKO_MAXU 16   /* actuators */
#define ODKO_MAXS 8    /* sites */
#define ODKO_MAXSENS 16
#define ODKO_MAXSD 48  /* sensordata floats */
#define ODKO_MAXG 4    /* collision geoms */
#define ODKO_MAXHV 64  /* hull vertices (all meshes) */
#define ODKO_MAXHF 128 /* hull faces */
#define ODKO_MAXCON 12 /* contacts: 3 geom pairs x 4 */
#define ODKO_MAXEFC 96 /* constraint rows */
#define ODKO_MAXEQ 8   /* equality constraints (<= 6 rows each) */

enum { ODKO_JNT_FREE = 0, ODKO_JNT_HINGE = 3 };
enum { ODKO_EQ_CONNECT = 0, ODKO_EQ_WELD = 1, ODKO_EQ_JOINT = 2 };   /* mjtEq */
enum { ODKO_GEOM_PLANE = 0, ODKO_GEOM_HFIELD = 1, ODKO_GEOM_SPHERE = 2, ODKO_GEOM_CAPSULE = 3, ODKO_GEOM_MESH = 7 };   /* mjtGeom */
#define ODKO_MAXHFIELD (256 * 256)
enum { ODKO_S_GYRO = 0, ODKO_S_VELOCIMETER, ODKO_S_ACCELEROMETER, ODKO_S_FRAMEZAXIS, ODKO_S_FRAMEXAXIS,
       ODKO_S_FRAMELINVEL, ODKO_S_FRAMEANGVEL, ODKO_S_FRAMEPOS, ODKO_S_FRAMEQUAT };

/* a convex polytope: vertices, face polygons, unique edges with their two faces (odk_oracle_convex.inc) */
#define CV_MAXV 32
#define CV_MAXF 64
#define CV_MAXP 10  /* vertices of one face polygon */
#define CV_MAXE 96
#define CV_MAXCLIP (4 * CV_MAXP)

typedef struct {
  int nv, nf, ne;
  real v[CV_MAXV][3];
  int fcnt[CV_MAXF], fidx[CV_MAXF][CV_MAXP]; /* face polygons, counter-clockwise seen from outside */
  real fnorm[CV_MAXF][3];                    /* outward unit normals */
  int e[CV_MAXE][2], ef[CV_MAXE][2];         /* unique edges a -> b; ef[0] = face that runs a -> b, ef[1] = face that runs b -> a */
  real c[3];                                 /* mean of the vertices (interior point) */
  int top_only;                              /* height-field hypothesis sweep (hfield_mode 2): of THIS polytope only face 0 and the edges of face 0 give axes / incident faces */
} odko_convex;


typedef struct {
  int nq, nv, nu, nbody, njnt, nsite, nsensor, nsensordata, ncgeom, nhullvert, nhullface;
  real timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations, eulerdamp;
  int cone;   /* mjtCone: 0 pyramidal (4 rows per condim-3 contact), 1 elliptic (3 rows: normal, two tangents; odk_oracle.c "elliptic cones") */
  /* bodies */
  int body_parentid[ODKO_MAXB], body_rootid[ODKO_MAXB], body_weldid[ODKO_MAXB];
  int body_jntadr[ODKO_MAXB], body_jntnum[ODKO_MAXB], body_dofadr[ODKO_MAXB], body_dofnum[ODKO_MAXB];
  real body_pos[ODKO_MAXB][3], body_quat[ODKO_MAXB][4], body_mass[ODKO_MAXB], body_ipos[ODKO_MAXB][3];
  real body_inertia_full[ODKO_MAXB][6], body_invweight0[ODKO_MAXB][2];
  /* joints / dofs */
  int jnt_type[ODKO_MAXJ], jnt_bodyid[ODKO_MAXJ], jnt_qposadr[ODKO_MAXJ], jnt_dofadr[ODKO_MAXJ], jnt_limited[ODKO_MAXJ];
  real jnt_pos[ODKO_MAXJ][3], jnt_axis[ODKO_MAXJ][3], jnt_range[ODKO_MAXJ][2], jnt_solref[ODKO_MAXJ][2],
      jnt_solimp[ODKO_MAXJ][5], jnt_margin[ODKO_MAXJ];
  int dof_bodyid[ODKO_MAXV], dof_jntid[ODKO_MAXV], dof_parentid[ODKO_MAXV];
  real dof_armature[ODKO_MAXV], dof_damping[ODKO_MAXV], dof_frictionloss[ODKO_MAXV], dof_invweight0[ODKO_MAXV];
  real dof_solref[ODKO_MAXV][2], dof_solimp[ODKO_MAXV][5];
  real qpos0[ODKO_MAXQ], key_qpos[ODKO_MAXQ], key_ctrl[ODKO_MAXU];
  /* actuators */
  int actuator_trnid[ODKO_MAXU], actuator_ctrllimited[ODKO_MAXU], actuator_forcelimited[ODKO_MAXU];
  real actuator_gainprm0[ODKO_MAXU], actuator_biasprm[ODKO_MAXU][3], actuator_ctrlrange[ODKO_MAXU][2],
      actuator_forcerange[ODKO_MAXU][2], actuator_gear[ODKO_MAXU];
  /* sites, sensors */
  int site_bodyid[ODKO_MAXS];
  real site_pos[ODKO_MAXS][3], site_quat[ODKO_MAXS][4];
  int sensor_type[ODKO_MAXSENS], sensor_objid[ODKO_MAXSENS], sensor_adr[ODKO_MAXSENS], sensor_dim[ODKO_MAXSENS];
  /* collision geoms */
  int cgeom_id[ODKO_MAXG], cgeom_type[ODKO_MAXG], cgeom_bodyid[ODKO_MAXG], cgeom_priority[ODKO_MAXG],
      cgeom_condim[ODKO_MAXG], cgeom_contype[ODKO_MAXG], cgeom_conaffinity[ODKO_MAXG];
  int cgeom_vertadr[ODKO_MAXG], cgeom_vertnum[ODKO_MAXG], cgeom_faceadr[ODKO_MAXG], cgeom_facenum[ODKO_MAXG];
  real cgeom_size[ODKO_MAXG][3];   /* sphere: radius; capsule: radius, half length (along the geom frame's z axis) */
  real cgeom_pos[ODKO_MAXG][3], cgeom_quat[ODKO_MAXG][4], cgeom_friction[ODKO_MAXG][3], cgeom_solref[ODKO_MAXG][2],
      cgeom_solimp[ODKO_MAXG][5], cgeom_solmix[ODKO_MAXG];
  real hull_vert[ODKO_MAXHV][3];
  int hull_face[ODKO_MAXHF][3];
  /* height field of the floor geom (rough terrain): data[row][col] in [0, 1], row <-> y, col <-> x */
  int hfield_nrow, hfield_ncol;
  real hfield_size[4];                 /* x half-extent, y half-extent, elevation scale, base thickness */
  real hfield_data[ODKO_MAXHFIELD];
  /* equality constraints (mjModel eq_*): obj = body ids (connect / weld) or joint ids (joint; obj2 = -1: none); data as MuJoCo's compiler
   * leaves it -- connect: anchor in body1 [0:3], in body2 [3:6]; weld: anchor in body2 [0:3], in body1 [3:6], relpose quaternion [6:10],
   * torquescale [10]; joint: polycoef [0:5] */
  int neq, eq_type[ODKO_MAXEQ], eq_obj1id[ODKO_MAXEQ], eq_obj2id[ODKO_MAXEQ], eq_active[ODKO_MAXEQ];
  real eq_data[ODKO_MAXEQ][11], eq_solref[ODKO_MAXEQ][2], eq_solimp[ODKO_MAXEQ][5];
  /* derived: contact pair list, face polygons / edges of the mesh geoms (geom frame) */
  int npair, pair_g1[3], pair_g2[3];
  odko_convex cgeom_convex[ODKO_MAXG];
  int hfield_mode;   /* 0: prisms of the cells under the geom (MJX hfield_convex as recalled; what the kernels run); 1: round-2's one-triangle plane;
                      * hypothesis sweep (tools/hfield_variants.py; oracle only): 2 = a prism's side / bottom faces and vertical edges give no axis and are never
                      * incident (only its top triangle collides), 3 = mode 0 but a contact is kept only when its normal points up (n_z > 0.5 in the field's
                      * frame), 4 = one contact per prism (its deepest), the four deepest of those kept (MuJoCo-C's mjc_ConvexHField gives one per prism),
                      * 5 = the four contacts chosen by the plane-convex manifold heuristic over all prisms' active candidates with their mean normal (round 6) */
} odko_model;

typedef struct {
  /* state */
  real qpos[ODKO_MAXQ], qvel[ODKO_MAXV], qacc_warmstart[ODKO_MAXV], ctrl[ODKO_MAXU], time;
  /* position stage */
  real xpos[ODKO_MAXB][3], xquat[ODKO_MAXB][4], xmat[ODKO_MAXB][9], xipos[ODKO_MAXB][3];
  real xanchor[ODKO_MAXJ][3], xaxis[ODKO_MAXJ][3];
  real site_xpos[ODKO_MAXS][3], site_xmat[ODKO_MAXS][9];
  real geom_xpos[ODKO_MAXG][3], geom_xmat[ODKO_MAXG][9];
  real subtree_com[ODKO_MAXB][3], cinert[ODKO_MAXB][10], crb[ODKO_MAXB][10], cdof[ODKO_MAXV][6];
  real qM[ODKO_MAXV * ODKO_MAXV], qL[ODKO_MAXV * ODKO_MAXV]; /* dense nv x nv (row stride nv) and its Cholesky */
  /* contacts */
  int ncon;
  real contact_dist[ODKO_MAXCON], contact_pos[ODKO_MAXCON][3], contact_frame[ODKO_MAXCON][9], contact_friction[ODKO_MAXCON];
  int contact_geom1[ODKO_MAXCON], contact_geom2[ODKO_MAXCON]; /* cgeom indices */
  /* constraints */
  int nefc, ne, nf, nl, nc;
  real efc_J[ODKO_MAXEFC * ODKO_MAXV], efc_pos[ODKO_MAXEFC], efc_D[ODKO_MAXEFC], efc_R[ODKO_MAXEFC], efc_aref[ODKO_MAXEFC],
      efc_frictionloss[ODKO_MAXEFC], efc_force[ODKO_MAXEFC], efc_invweight[ODKO_MAXEFC], efc_b[ODKO_MAXEFC], efc_k[ODKO_MAXEFC],
      efc_imp[ODKO_MAXEFC];
  /* elliptic cones: per contact, the first of its 3 rows, the friction coefficient and the regularised cone's mu = friction sqrt(R_t / R_n) */
  int contact_efc[ODKO_MAXCON];
  real contact_mu_reg[ODKO_MAXCON];
  /* velocity stage */
  real cvel[ODKO_MAXB][6], cdof_dot[ODKO_MAXV][6], cacc[ODKO_MAXB][6];
  real qfrc_bias[ODKO_MAXV], qfrc_passive[ODKO_MAXV], qfrc_actuator[ODKO_MAXV], actuator_force[ODKO_MAXU];
  real qfrc_smooth[ODKO_MAXV], qacc_smooth[ODKO_MAXV], qacc[ODKO_MAXV], qfrc_constraint[ODKO_MAXV];
  real sensordata[ODKO_MAXSD];
  /* solver diagnostics */
  real solver_cost0, solver_cost1, ls_alpha;
  int warm_used, ls_iters;
  /* smallest gap between the branch taken and its runner-up over the discrete decisions since the caller last set these to 1e30
   * (odk_oracle.c "decision margins"): [0] collision lengths (m), [1] normal cosines, [2] clipping-plane distances (m), [3] manifold
   * arg-max steps (relative), [4] warm-start pick (relative cost difference) */
  real decision_margin[5];
} odko_data;

#ifdef __cplusplus
extern "C" {
#endif

/* model */
odko_model* odko_model_load(const void* blob, uint64_t len);
void odko_model_free(odko_model* m);
odko_model* odko_model_copy(const odko_model* m);
/* tests' referee: inside a band of eps (lengths in m, cosines; eps_rel for the manifold's relative gaps) around a tie, the collision
 * decisions whose class bit is set in mask take the runner-up (odk_oracle.c "Tie bias"); mask 0 switches it off.  Per thread. */
void odko_set_tie_bias(int mask, real eps, real eps_rel);
void odko_set_tie_bias_window(int mask, real eps, real eps_rel, int first_pass, int last_pass); /* only the collision passes first .. last after this call */
void odko_model_jitter_hulls(odko_model* m, unsigned seed, real rel); /* relative noise on the hull vertices + rebuild of the convex tables (tests' referee) */
/* named access for tests / domain randomisation: returns pointer + element count, NULL if unknown */
real* odko_model_field(odko_model* m, const char* name, int* count);
int odko_model_int(const odko_model* m, const char* name);
int odko_model_set_int(odko_model* m, const char* name, int value); /* "iterations" / "ls_iterations" / "hfield_mode" */
int odko_model_eq_set_active(odko_model* m, int e, int on);          /* mjData.eq_active: equality constraint e on / off */
/* tests: the solver's cost, gradient (nv) and Newton Hessian (nv x nv, row-major) at an arbitrary qacc, for the constraint rows of the last
 * odko_forward on d (d is not modified) */
void odko_solver_probe(const odko_model* m, const odko_data* d, const real* qacc, real* cost, real* grad, real* hess);
int odko_convex_pair(const real* va, int nva, const int* ta, int nta, const real* pa, const real* ma, const real* vb, int nvb, const int* tb,
                     int ntb, const real* pb, const real* mb, real* dist4, real* pos12, real* normal3, real* sat3);
int odko_model_convex_counts(const odko_model* m, int g, int* nv, int* nf, int* ne);

/* physics */
odko_data* odko_data_new(void);
void odko_data_free(odko_data* d);
void odko_make_data(const odko_model* m, odko_data* d);          /* mjx.make_data: qpos0, zeros */
void odko_forward(const odko_model* m, odko_data* d);            /* mjx.forward */
void odko_step(const odko_model* m, odko_data* d);               /* mjx.step = forward + Euler */
void odko_env_physics_step(const odko_model* m, odko_data* d, const real* ctrl, int n_substeps); /* mjx_env.step */
real* odko_data_field(odko_data* d, const char* name, int* count);
int odko_data_int(const odko_data* d, const char* name);

#ifdef __cplusplus
}
#endif
#endif
