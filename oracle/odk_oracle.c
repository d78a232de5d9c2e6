/* odk_oracle.c -- float64 CPU restatement of mjx.forward / mjx.step.  TEST INFRASTRUCTURE ONLY
 * (see odk_oracle.h).  PARITY UNPINNED: written from the published MuJoCo "Computation"
 * chapter and recalled MJX sources ([UPSTREAM-MEMORY]); the reference
 * calls this code at playground/open_duck_mini_v2/joystick.py:258 (init -> forward) and :420
 * (mjx_env.step -> 10 x mjx.step).
 *
 * Readability over speed: dense matrices, explicit loops, one env at a time.
 */
#include "odk_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------ small vector math */
static void v3_zero(real* a) { a[0] = a[1] = a[2] = 0; }
static void v3_copy(real* a, const real* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static real v3_dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3_cross(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void v3_addscl(real* r, const real* a, const real* b, real s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static void v3_sub(real* r, const real* a, const real* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static real v3_normalize(real* a) {
  real n = sqrt(v3_dot(a, a));
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
static void quat_mul(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mat_mulvec(real* r, const real* m, const real* v) { /* r = M v */
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
       z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mat_tmulvec(real* r, const real* m, const real* v) { /* r = M^T v */
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
       z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mat_mul(real* r, const real* a, const real* b) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof(t));
}

/* spatial (6D, [angular; linear]) helpers, MuJoCo conventions */
static void inert_mul(real* res, const real* i, const real* v) { /* mju_mulInertVec */
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(real* res, const real* vel, const real* v) { /* mju_crossMotion */
  real a[3], b[3];
  v3_cross(res, vel, v);
  v3_cross(a, vel, v + 3);
  v3_cross(b, vel + 3, v);
  res[3] = a[0] + b[0]; res[4] = a[1] + b[1]; res[5] = a[2] + b[2];
}
static void cross_force(real* res, const real* vel, const real* f) { /* mju_crossForce */
  real a[3], b[3];
  v3_cross(a, vel, f);
  v3_cross(b, vel + 3, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  v3_cross(res + 3, vel, f + 3);
}

/* ------------------------------------------------------------------ blob loader */
typedef struct { char name[32]; uint32_t dtype, ndim, shape[4]; uint64_t nbytes; } rec_hdr;

static const unsigned char* find_rec(const unsigned char* blob, uint64_t len, const char* name, rec_hdr* h) {
  uint32_t n;
  memcpy(&n, blob + 8, 4);
  uint64_t off = 16;
  for (uint32_t i = 0; i < n && off + 64 <= len; i++) {
    memcpy(h->name, blob + off, 32);
    memcpy(&h->dtype, blob + off + 32, 4);
    memcpy(&h->ndim, blob + off + 36, 4);
    memcpy(h->shape, blob + off + 40, 16);
    memcpy(&h->nbytes, blob + off + 56, 8);
    off += 64;
    if (strncmp(h->name, name, 32) == 0) return blob + off;
    off += h->nbytes + ((8 - (h->nbytes & 7)) & 7);
  }
  return NULL;
}
static int load_f(const unsigned char* blob, uint64_t len, const char* name, real* dst, int maxcount) {
  rec_hdr h;
  const unsigned char* p = find_rec(blob, len, name, &h);
  if (!p || h.dtype != 0) return -1;
  int cnt = (int)(h.nbytes / 8);
  if (cnt > maxcount) { fprintf(stderr, "odko: field %s too large (%d > %d)\n", name, cnt, maxcount); return -1; }
  for (int i = 0; i < cnt; i++) { double v; memcpy(&v, p + 8 * i, 8); dst[i] = (real)v; }
  return cnt;
}
static int load_i(const unsigned char* blob, uint64_t len, const char* name, int* dst, int maxcount) {
  rec_hdr h;
  const unsigned char* p = find_rec(blob, len, name, &h);
  if (!p || h.dtype != 1) return -1;
  int cnt = (int)(h.nbytes / 4);
  if (cnt > maxcount) { fprintf(stderr, "odko: field %s too large (%d > %d)\n", name, cnt, maxcount); return -1; }
  for (int i = 0; i < cnt; i++) { int32_t v; memcpy(&v, p + 4 * i, 4); dst[i] = v; }
  return cnt;
}

#define LF(name, dst, max) if (load_f(b, len, name, (real*)(dst), max) < 0) { fprintf(stderr, "odko: missing %s\n", name); ok = 0; }
#define LI(name, dst, max) if (load_i(b, len, name, (int*)(dst), max) < 0) { fprintf(stderr, "odko: missing %s\n", name); ok = 0; }

static void build_mesh_convex(odko_model* m);
odko_model* odko_model_load(const void* blob, uint64_t len) {
  const unsigned char* b = (const unsigned char*)blob;
  if (len < 16 || memcmp(b, "ODKM", 4) != 0) return NULL;
  odko_model* m = (odko_model*)calloc(1, sizeof(odko_model));
  int ok = 1;
  LI("nq", &m->nq, 1); LI("nv", &m->nv, 1); LI("nu", &m->nu, 1); LI("nbody", &m->nbody, 1); LI("njnt", &m->njnt, 1);
  LI("nsite", &m->nsite, 1); LI("nsensordata", &m->nsensordata, 1);
  if (!ok || m->nq > ODKO_MAXQ || m->nv > ODKO_MAXV || m->nu > ODKO_MAXU || m->nbody > ODKO_MAXB || m->njnt > ODKO_MAXJ ||
      m->nsite > ODKO_MAXS) { free(m); return NULL; }
  LF("opt_timestep", &m->timestep, 1); LF("opt_gravity", m->gravity, 3); LF("opt_tolerance", &m->tolerance, 1);
  LF("opt_ls_tolerance", &m->ls_tolerance, 1); LF("opt_impratio", &m->impratio, 1); LF("stat_meaninertia", &m->meaninertia, 1);
  LI("opt_iterations", &m->iterations, 1); LI("opt_ls_iterations", &m->ls_iterations, 1); LI("opt_eulerdamp", &m->eulerdamp, 1);
  { rec_hdr chh; if (find_rec(b, len, "opt_cone", &chh)) load_i(b, len, "opt_cone", &m->cone, 1); }   /* optional: blobs before round 5 have none (= pyramidal) */
  LI("body_parentid", m->body_parentid, ODKO_MAXB); LI("body_rootid", m->body_rootid, ODKO_MAXB); LI("body_weldid", m->body_weldid, ODKO_MAXB);
  LI("body_jntadr", m->body_jntadr, ODKO_MAXB); LI("body_jntnum", m->body_jntnum, ODKO_MAXB);
  LI("body_dofadr", m->body_dofadr, ODKO_MAXB); LI("body_dofnum", m->body_dofnum, ODKO_MAXB);
  LF("body_pos", m->body_pos, ODKO_MAXB * 3); LF("body_quat", m->body_quat, ODKO_MAXB * 4); LF("body_mass", m->body_mass, ODKO_MAXB);
  LF("body_ipos", m->body_ipos, ODKO_MAXB * 3); LF("body_inertia_full", m->body_inertia_full, ODKO_MAXB * 6);
  LF("body_invweight0", m->body_invweight0, ODKO_MAXB * 2);
  LI("jnt_type", m->jnt_type, ODKO_MAXJ); LI("jnt_bodyid", m->jnt_bodyid, ODKO_MAXJ); LI("jnt_qposadr", m->jnt_qposadr, ODKO_MAXJ);
  LI("jnt_dofadr", m->jnt_dofadr, ODKO_MAXJ); LI("jnt_limited", m->jnt_limited, ODKO_MAXJ);
  LF("jnt_pos", m->jnt_pos, ODKO_MAXJ * 3); LF("jnt_axis", m->jnt_axis, ODKO_MAXJ * 3); LF("jnt_range", m->jnt_range, ODKO_MAXJ * 2);
  LF("jnt_solref", m->jnt_solref, ODKO_MAXJ * 2); LF("jnt_solimp", m->jnt_solimp, ODKO_MAXJ * 5); LF("jnt_margin", m->jnt_margin, ODKO_MAXJ);
  LI("dof_bodyid", m->dof_bodyid, ODKO_MAXV); LI("dof_jntid", m->dof_jntid, ODKO_MAXV); LI("dof_parentid", m->dof_parentid, ODKO_MAXV);
  LF("dof_armature", m->dof_armature, ODKO_MAXV); LF("dof_damping", m->dof_damping, ODKO_MAXV);
  LF("dof_frictionloss", m->dof_frictionloss, ODKO_MAXV); LF("dof_invweight0", m->dof_invweight0, ODKO_MAXV);
  LF("dof_solref", m->dof_solref, ODKO_MAXV * 2); LF("dof_solimp", m->dof_solimp, ODKO_MAXV * 5);
  LF("qpos0", m->qpos0, ODKO_MAXQ); LF("key_qpos", m->key_qpos, ODKO_MAXQ); LF("key_ctrl", m->key_ctrl, ODKO_MAXU);
  LI("actuator_trnid", m->actuator_trnid, ODKO_MAXU); LI("actuator_ctrllimited", m->actuator_ctrllimited, ODKO_MAXU);
  LI("actuator_forcelimited", m->actuator_forcelimited, ODKO_MAXU);
  LF("actuator_gainprm0", m->actuator_gainprm0, ODKO_MAXU); LF("actuator_biasprm", m->actuator_biasprm, ODKO_MAXU * 3);
  LF("actuator_ctrlrange", m->actuator_ctrlrange, ODKO_MAXU * 2); LF("actuator_forcerange", m->actuator_forcerange, ODKO_MAXU * 2);
  LF("actuator_gear", m->actuator_gear, ODKO_MAXU);
  LI("site_bodyid", m->site_bodyid, ODKO_MAXS); LF("site_pos", m->site_pos, ODKO_MAXS * 3); LF("site_quat", m->site_quat, ODKO_MAXS * 4);
  m->nsensor = load_i(b, len, "sensor_type", m->sensor_type, ODKO_MAXSENS);
  LI("sensor_objid", m->sensor_objid, ODKO_MAXSENS); LI("sensor_adr", m->sensor_adr, ODKO_MAXSENS); LI("sensor_dim", m->sensor_dim, ODKO_MAXSENS);
  m->ncgeom = load_i(b, len, "cgeom_type", m->cgeom_type, ODKO_MAXG);
  LI("cgeom_id", m->cgeom_id, ODKO_MAXG); LI("cgeom_bodyid", m->cgeom_bodyid, ODKO_MAXG); LI("cgeom_priority", m->cgeom_priority, ODKO_MAXG);
  LI("cgeom_condim", m->cgeom_condim, ODKO_MAXG); LI("cgeom_contype", m->cgeom_contype, ODKO_MAXG);
  LI("cgeom_conaffinity", m->cgeom_conaffinity, ODKO_MAXG);
  LI("cgeom_vertadr", m->cgeom_vertadr, ODKO_MAXG); LI("cgeom_vertnum", m->cgeom_vertnum, ODKO_MAXG);
  LI("cgeom_faceadr", m->cgeom_faceadr, ODKO_MAXG); LI("cgeom_facenum", m->cgeom_facenum, ODKO_MAXG);
  LF("cgeom_pos", m->cgeom_pos, ODKO_MAXG * 3); LF("cgeom_quat", m->cgeom_quat, ODKO_MAXG * 4);
  load_f(b, len, "cgeom_size", (real*)m->cgeom_size, ODKO_MAXG * 3); /* optional: blobs of models without primitive colliders predate it */
  LF("cgeom_friction", m->cgeom_friction, ODKO_MAXG * 3); LF("cgeom_solref", m->cgeom_solref, ODKO_MAXG * 2);
  LF("cgeom_solimp", m->cgeom_solimp, ODKO_MAXG * 5); LF("cgeom_solmix", m->cgeom_solmix, ODKO_MAXG);
  m->nhullvert = load_f(b, len, "hull_vert", (real*)m->hull_vert, ODKO_MAXHV * 3) / 3;
  m->nhullface = load_i(b, len, "hull_face", (int*)m->hull_face, ODKO_MAXHF * 3) / 3;
  if (!ok || m->nsensor < 0 || m->ncgeom < 0) { free(m); return NULL; }
  { /* optional equality constraints (mjcf.py: <equality> joint / connect / weld) */
    rec_hdr eh;
    if (find_rec(b, len, "eq_type", &eh) && eh.nbytes > 0) {
      m->neq = load_i(b, len, "eq_type", m->eq_type, ODKO_MAXEQ);
      if (m->neq < 0 || load_i(b, len, "eq_obj1id", m->eq_obj1id, ODKO_MAXEQ) != m->neq || load_i(b, len, "eq_obj2id", m->eq_obj2id, ODKO_MAXEQ) != m->neq ||
          load_i(b, len, "eq_active", m->eq_active, ODKO_MAXEQ) != m->neq || load_f(b, len, "eq_data", (real*)m->eq_data, ODKO_MAXEQ * 11) != 11 * m->neq ||
          load_f(b, len, "eq_solref", (real*)m->eq_solref, ODKO_MAXEQ * 2) != 2 * m->neq || load_f(b, len, "eq_solimp", (real*)m->eq_solimp, ODKO_MAXEQ * 5) != 5 * m->neq) {
        free(m); return NULL;
      }
    }
  }
  { /* optional height field (scene_rough_terrain_backlash.xml:22) */
    rec_hdr hh;
    { const char* hm = getenv("ODK_ORACLE_HFIELD_MODE"); if (hm) m->hfield_mode = atoi(hm); }   /* hypothesis sweep: the parity tests against a variant kernel build */
    if (find_rec(b, len, "hfield_data", &hh)) {
      m->hfield_nrow = (int)hh.shape[0]; m->hfield_ncol = (int)hh.shape[1];
      if (m->hfield_nrow * m->hfield_ncol > ODKO_MAXHFIELD || load_f(b, len, "hfield_data", m->hfield_data, ODKO_MAXHFIELD) < 0 ||
          load_f(b, len, "hfield_size", m->hfield_size, 4) < 0) { free(m); return NULL; }
    }
  }
  /* contact pairs: contype/conaffinity filter, same-weld-body exclusion, parent-child exclusion (MuJoCo's filterparent: the
     welded parents, the world excepted); order: the floor's pairs (plane / hfield against anything) first, by geom id, then the
     pairs between body geoms (MJX groups by type pair, and the floor types are the lowest) */
  m->npair = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int i = 0; i < m->ncgeom; i++)
      for (int j = i + 1; j < m->ncgeom; j++) {
        int ti = m->cgeom_type[i], tj = m->cgeom_type[j];
        int mm = !(ti == ODKO_GEOM_PLANE || ti == ODKO_GEOM_HFIELD || tj == ODKO_GEOM_PLANE || tj == ODKO_GEOM_HFIELD);
        if ((pass == 0) == mm) continue;
        if (!((m->cgeom_contype[i] & m->cgeom_conaffinity[j]) || (m->cgeom_contype[j] & m->cgeom_conaffinity[i]))) continue;
        int b1 = m->body_weldid[m->cgeom_bodyid[i]], b2 = m->body_weldid[m->cgeom_bodyid[j]];
        if (b1 == b2) continue;
        if (b1 != 0 && b2 != 0 && (m->body_weldid[m->body_parentid[b1]] == b2 || m->body_weldid[m->body_parentid[b2]] == b1)) continue;
        if (m->npair < 3) {
          int first = i, second = j; /* geom1 = lower type (plane/hfield first) */
          if (ti > tj) { first = j; second = i; }
          m->pair_g1[m->npair] = first; m->pair_g2[m->npair] = second; m->npair++;
        }
      }
  build_mesh_convex(m);
  return m;
}
void odko_model_free(odko_model* m) { free(m); }
odko_model* odko_model_copy(const odko_model* m) {
  odko_model* c = (odko_model*)malloc(sizeof(odko_model));
  memcpy(c, m, sizeof(odko_model));
  return c;
}

static void build_mesh_convex(odko_model* m);
/* Moves every hull vertex by a relative `rel` (uniform in +-rel x max(|coordinate|, 1 cm), counter-hashed from `seed`) and rebuilds
 * the polygons / normals / edges: what the collision geometry looks like to an implementation that carries its constants in
 * another precision.  The parity tests' float32 referee uses it (rel ~ 1e-7) to tell near-ties between hull features -- decided
 * by the last bits of the face normals, and blind to any perturbation of the STATE -- from real disagreements. */
void odko_model_jitter_hulls(odko_model* m, unsigned seed, real rel) {
  unsigned x = seed * 2654435761u + 12345u;
  for (int i = 0; i < ODKO_MAXHV; i++)
    for (int k = 0; k < 3; k++) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      real u = ((real)(x & 0xFFFFFF) / (real)0x800000) - 1.0, v = m->hull_vert[i][k];
      m->hull_vert[i][k] = v + u * rel * (fabs(v) > 0.01 ? fabs(v) : 0.01);
    }
  build_mesh_convex(m);
}

#define MF(nm, cnt) if (!strcmp(name, #nm)) { *count = (cnt); return (real*)m->nm; }
real* odko_model_field(odko_model* m, const char* name, int* count) {
  MF(body_mass, m->nbody) MF(body_ipos, m->nbody * 3) MF(body_pos, m->nbody * 3) MF(body_quat, m->nbody * 4)
  MF(body_inertia_full, m->nbody * 6) MF(body_invweight0, m->nbody * 2) MF(cgeom_size, m->ncgeom * 3)
  MF(dof_frictionloss, m->nv) MF(dof_armature, m->nv) MF(dof_damping, m->nv) MF(dof_invweight0, m->nv)
  MF(qpos0, m->nq) MF(key_qpos, m->nq) MF(key_ctrl, m->nu)
  MF(actuator_gainprm0, m->nu) MF(actuator_biasprm, m->nu * 3) MF(actuator_ctrlrange, m->nu * 2) MF(actuator_forcerange, m->nu * 2)
  MF(cgeom_friction, m->ncgeom * 3) MF(jnt_range, m->njnt * 2) MF(gravity, 3) MF(hull_vert, m->nhullvert * 3)
  MF(eq_data, m->neq * 11) MF(eq_solref, m->neq * 2) MF(eq_solimp, m->neq * 5)
  if (!strcmp(name, "timestep")) { *count = 1; return &m->timestep; }
  if (!strcmp(name, "meaninertia")) { *count = 1; return &m->meaninertia; }
  *count = 0;
  return NULL;
}
#define MI(nm) if (!strcmp(name, #nm)) return m->nm;
int odko_model_int(const odko_model* m, const char* name) {
  MI(nq) MI(nv) MI(nu) MI(nbody) MI(njnt) MI(nsite) MI(nsensor) MI(nsensordata) MI(ncgeom) MI(npair) MI(iterations) MI(ls_iterations) MI(neq) MI(cone)
  return -1;
}

/* solver options only (tests drive the Newton solver to convergence to check its fixed point) */
int odko_model_set_int(odko_model* m, const char* name, int value) {
  if (!strcmp(name, "iterations")) { m->iterations = value; return 0; }
  if (!strcmp(name, "ls_iterations")) { m->ls_iterations = value; return 0; }
  if (!strcmp(name, "hfield_mode")) { m->hfield_mode = value; return 0; }
  if (!strcmp(name, "cone")) { m->cone = value != 0; return 0; }
  return -1;
}

/* mjData.eq_active counterpart: switches equality constraint e on / off (tests compare a run with and without it) */
int odko_model_eq_set_active(odko_model* m, int e, int on) {
  if (e < 0 || e >= m->neq) return -1;
  m->eq_active[e] = on != 0;
  return 0;
}

odko_data* odko_data_new(void) { return (odko_data*)calloc(1, sizeof(odko_data)); }
void odko_data_free(odko_data* d) { free(d); }

#define DF(nm, cnt) if (!strcmp(name, #nm)) { *count = (cnt); return (real*)d->nm; }
real* odko_data_field(odko_data* d, const char* name, int* count) {
  DF(qpos, ODKO_MAXQ) DF(qvel, ODKO_MAXV) DF(qacc_warmstart, ODKO_MAXV) DF(ctrl, ODKO_MAXU)
  DF(xpos, ODKO_MAXB * 3) DF(xquat, ODKO_MAXB * 4) DF(xmat, ODKO_MAXB * 9) DF(xipos, ODKO_MAXB * 3)
  DF(xanchor, ODKO_MAXJ * 3) DF(xaxis, ODKO_MAXJ * 3) DF(site_xpos, ODKO_MAXS * 3) DF(site_xmat, ODKO_MAXS * 9)
  DF(geom_xpos, ODKO_MAXG * 3) DF(geom_xmat, ODKO_MAXG * 9) DF(subtree_com, ODKO_MAXB * 3) DF(cinert, ODKO_MAXB * 10)
  DF(cdof, ODKO_MAXV * 6) DF(qM, ODKO_MAXV * ODKO_MAXV) DF(contact_dist, ODKO_MAXCON) DF(contact_pos, ODKO_MAXCON * 3)
  DF(contact_frame, ODKO_MAXCON * 9) DF(contact_friction, ODKO_MAXCON)
  DF(efc_J, ODKO_MAXEFC * ODKO_MAXV) DF(efc_pos, ODKO_MAXEFC) DF(efc_D, ODKO_MAXEFC) DF(efc_R, ODKO_MAXEFC) DF(efc_aref, ODKO_MAXEFC)
  DF(efc_frictionloss, ODKO_MAXEFC) DF(efc_force, ODKO_MAXEFC) DF(efc_imp, ODKO_MAXEFC)
  DF(cvel, ODKO_MAXB * 6) DF(cdof_dot, ODKO_MAXV * 6) DF(cacc, ODKO_MAXB * 6)
  DF(qfrc_bias, ODKO_MAXV) DF(qfrc_passive, ODKO_MAXV) DF(qfrc_actuator, ODKO_MAXV) DF(actuator_force, ODKO_MAXU)
  DF(qfrc_smooth, ODKO_MAXV) DF(qacc_smooth, ODKO_MAXV) DF(qacc, ODKO_MAXV) DF(qfrc_constraint, ODKO_MAXV) DF(sensordata, ODKO_MAXSD)
  DF(decision_margin, 5)
  if (!strcmp(name, "time")) { *count = 1; return &d->time; }
  if (!strcmp(name, "solver_cost0")) { *count = 1; return &d->solver_cost0; }
  if (!strcmp(name, "solver_cost1")) { *count = 1; return &d->solver_cost1; }
  if (!strcmp(name, "ls_alpha")) { *count = 1; return &d->ls_alpha; }
  *count = 0;
  return NULL;
}
#define DI(nm) if (!strcmp(name, #nm)) return d->nm;
int odko_data_int(const odko_data* d, const char* name) {
  DI(ncon) DI(nefc) DI(ne) DI(nf) DI(nl) DI(nc) DI(warm_used) DI(ls_iters)
  return -1;
}

void odko_make_data(const odko_model* m, odko_data* d) {
  memset(d, 0, sizeof(*d));
  for (int k = 0; k < 5; k++) d->decision_margin[k] = 1e30;
  for (int i = 0; i < m->nq; i++) d->qpos[i] = m->qpos0[i];
}

/* ------------------------------------------------------------------ fwd_position */
/* mjx smooth.kinematics */
static void kinematics(const odko_model* m, odko_data* d) {
  v3_zero(d->xpos[0]);
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat_to_mat(d->xmat[0], d->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b], jn = m->body_jntnum[b], ja = m->body_jntadr[b];
    real pos[3], quat[4];
    if (jn == 1 && m->jnt_type[ja] == ODKO_JNT_FREE) {
      int a = m->jnt_qposadr[ja];
      v3_copy(pos, d->qpos + a);
      memcpy(quat, d->qpos + a + 3, 4 * sizeof(real));
      quat_normalize(quat);
      v3_copy(d->xanchor[ja], pos);
      d->xaxis[ja][0] = 0; d->xaxis[ja][1] = 0; d->xaxis[ja][2] = 1;
    } else {
      real t[3];
      mat_mulvec(t, d->xmat[p], m->body_pos[b]);
      v3_addscl(pos, d->xpos[p], t, 1);
      quat_mul(quat, d->xquat[p], m->body_quat[b]);
      for (int j = ja; j < ja + jn; j++) {
        real R[9], qj[4], ang, s;
        int a = m->jnt_qposadr[j];
        quat_to_mat(R, quat);
        mat_mulvec(t, R, m->jnt_pos[j]);
        v3_addscl(d->xanchor[j], pos, t, 1);
        mat_mulvec(d->xaxis[j], R, m->jnt_axis[j]);
        ang = d->qpos[a] - m->qpos0[a]; /* hinge angle is relative to qpos0 */
        s = sin(0.5 * ang);
        qj[0] = cos(0.5 * ang); qj[1] = s * m->jnt_axis[j][0]; qj[2] = s * m->jnt_axis[j][1]; qj[3] = s * m->jnt_axis[j][2];
        quat_mul(quat, quat, qj);
        quat_to_mat(R, quat); /* off-centre rotation correction */
        mat_mulvec(t, R, m->jnt_pos[j]);
        v3_sub(pos, d->xanchor[j], t);
      }
    }
    quat_normalize(quat);
    v3_copy(d->xpos[b], pos);
    memcpy(d->xquat[b], quat, sizeof(quat));
    quat_to_mat(d->xmat[b], quat);
  }
  for (int b = 0; b < m->nbody; b++) {
    real t[3];
    mat_mulvec(t, d->xmat[b], m->body_ipos[b]);
    v3_addscl(d->xipos[b], d->xpos[b], t, 1);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    real t[3], R[9];
    mat_mulvec(t, d->xmat[b], m->site_pos[s]);
    v3_addscl(d->site_xpos[s], d->xpos[b], t, 1);
    quat_to_mat(R, m->site_quat[s]);
    mat_mul(d->site_xmat[s], d->xmat[b], R);
  }
  for (int g = 0; g < m->ncgeom; g++) {
    int b = m->cgeom_bodyid[g];
    real t[3], R[9];
    mat_mulvec(t, d->xmat[b], m->cgeom_pos[g]);
    v3_addscl(d->geom_xpos[g], d->xpos[b], t, 1);
    quat_to_mat(R, m->cgeom_quat[g]);
    mat_mul(d->geom_xmat[g], d->xmat[b], R);
  }
}

/* mjx smooth.com_pos: subtree COM, cinert, cdof */
static void com_pos(const odko_model* m, odko_data* d) {
  real mass[ODKO_MAXB];
  for (int b = 0; b < m->nbody; b++) {
    mass[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[b][k] = m->body_mass[b] * d->xipos[b][k];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    mass[p] += mass[b];
    for (int k = 0; k < 3; k++) d->subtree_com[p][k] += d->subtree_com[b][k];
  }
  for (int b = 0; b < m->nbody; b++) {
    if (mass[b] < MINVAL) v3_copy(d->subtree_com[b], d->xipos[b]);
    else for (int k = 0; k < 3; k++) d->subtree_com[b][k] /= mass[b];
  }
  for (int b = 0; b < m->nbody; b++) {
    const real* f = m->body_inertia_full[b];
    real Ib[9] = {f[0], f[3], f[4], f[3], f[1], f[5], f[4], f[5], f[2]}, T[9], Iw[9], Rt[9], off[3];
    const real* R = d->xmat[b];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
    mat_mul(T, R, Ib);
    mat_mul(Iw, T, Rt);
    v3_sub(off, d->xipos[b], d->subtree_com[m->body_rootid[b]]);
    real mb = m->body_mass[b], o2 = v3_dot(off, off);
    real* c = d->cinert[b];
    c[0] = Iw[0] + mb * (o2 - off[0] * off[0]);
    c[1] = Iw[4] + mb * (o2 - off[1] * off[1]);
    c[2] = Iw[8] + mb * (o2 - off[2] * off[2]);
    c[3] = Iw[1] - mb * off[0] * off[1];
    c[4] = Iw[2] - mb * off[0] * off[2];
    c[5] = Iw[5] - mb * off[1] * off[2];
    c[6] = mb * off[0]; c[7] = mb * off[1]; c[8] = mb * off[2];
    c[9] = mb;
  }
  for (int j = 0; j < m->njnt; j++) {
    int da = m->jnt_dofadr[j], b = m->jnt_bodyid[j];
    real off[3];
    v3_sub(off, d->subtree_com[m->body_rootid[b]], d->xanchor[j]);
    if (m->jnt_type[j] == ODKO_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        real* c = d->cdof[da + k];
        memset(c, 0, 6 * sizeof(real));
        c[3 + k] = 1;
        real* r = d->cdof[da + 3 + k];
        real ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        v3_copy(r, ax);
        v3_cross(r + 3, ax, off);
      }
    } else {
      real* c = d->cdof[da];
      v3_copy(c, d->xaxis[j]);
      v3_cross(c + 3, d->xaxis[j], off);
    }
  }
}

/* mjx smooth.crb + dense qM */
static void crb(const odko_model* m, odko_data* d) {
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++) memcpy(d->crb[b], d->cinert[b], 10 * sizeof(real));
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[p][k] += d->crb[b][k];
  }
  memset(d->qM, 0, sizeof(d->qM));
  for (int i = 0; i < nv; i++) {
    real buf[6];
    inert_mul(buf, d->crb[m->dof_bodyid[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real v = 0;
      for (int k = 0; k < 6; k++) v += d->cdof[j][k] * buf[k];
      d->qM[i * nv + j] = v;
      d->qM[j * nv + i] = v;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
}

/* dense Cholesky A = L L^T (lower); returns 0 on failure */
static int cholesky(real* L, const real* A, int n) {
  memcpy(L, A, (size_t)n * n * sizeof(real));
  for (int j = 0; j < n; j++) {
    real s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) s = MINVAL;
    s = sqrt(s);
    L[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      real t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / s;
    }
  }
  return 1;
}
static void chol_solve(real* x, const real* L, const real* b, int n) {
  for (int i = 0; i < n; i++) {
    real t = b[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}
static void mul_m(const odko_model* m, const odko_data* d, real* res, const real* v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int j = 0; j < nv; j++) s += d->qM[i * nv + j] * v[j];
    res[i] = s;
  }
}

/* ---- collision ---- */
/* mju_makeFrame / mjx math.make_frame: rows = normal, tangent1, tangent2 */
static void make_frame(real* frame, const real* n) {
  real a[3] = {n[0], n[1], n[2]}, b[3] = {0, 0, 0}, c[3];
  v3_normalize(a);
  if (fabs(a[1]) < 0.5) b[1] = 1; else b[2] = 1;
  real dt = v3_dot(a, b);
  v3_addscl(b, b, a, -dt);
  v3_normalize(b);
  v3_cross(c, a, b);
  v3_copy(frame, a); v3_copy(frame + 3, b); v3_copy(frame + 6, c);
}

/* mjx collision_convex._manifold_points: 4 points of approximately maximal area among masked.
 * [BUILD-DEFINED] AREA0: an area measure below 1e-7 m^2 counts as zero.  When the masked candidates are collinear or coincide
 * (a clipped sliver, one surviving point: common between a foot and a height-field prism) every measure is zero in exact
 * arithmetic and jp.argmax takes the first index; in floating point the zeros come out as +-1e-10 residues and the pick -- and
 * with it how many of the four slots repeat which point, i.e. the weight of each contact -- would be rounding noise, different
 * in float32 and float64.  With the threshold both resolve the tie like exact arithmetic does. */
#define AREA0(v) area0(v)
#define AREA_TIE 1e-7
/* ---- decision margins (test infrastructure of the test infrastructure).  The collision routines take discrete decisions -- which
 * axis separates best, face or edge contact, which face is incident, which side of a clipping plane, which candidates span the
 * largest area, which four contacts are the deepest -- and a float32 evaluation of the SAME algorithm takes the other branch
 * whenever the two alternatives are closer than its rounding error, however well-conditioned the state is otherwise (moving
 * the state by 1e-6 moves both alternatives alike: the perturbation check of the parity tests cannot see such a tie).
 * g_margin[k] keeps, per forward pass, the smallest gap between the alternative taken and the runner-up over the decisions of
 * every colliding pair: [0] lengths in m (separations, depths), [1] cosines between unit normals, [2] signed distances to
 * clipping planes in m, [3] relative gaps of the manifold's arg-max steps (squared lengths / areas).  Identical alternatives
 * (the same point twice in a candidate list) do not count.  Read through odko_data_field "decision_margin". */
static __thread real g_margin[4] = {1e30, 1e30, 1e30, 1e30};
/* Tie bias (odko_set_tie_bias): inside a band of `eps` around a tie, the decision classes whose bit is set in `mask` take the
 * RUNNER-UP instead -- consistently, for as long as the bias is on.  Bits: 1 which face of a polytope separates best, 2 whose face
 * is the reference, 4 edge contact or face contact (the EDGE_TOL threshold), 8 which face is incident, 16 side of a clipping
 * plane, 32 the manifold's arg-max steps (band in relative units: eps_rel), 64 the cut behind the fourth-deepest height-field
 * contact, 128 (the one solver decision) the Newton solver's starting point, warm start or unconstrained acceleration, when their
 * costs are within eps_rel of each other; 256 / 512: the line search (see ls_search); 1024: the AREA0 cut of the manifold's area measures (area0).  This is how the parity tests ask "does the kernel's answer follow from the oracle's own algorithm when THIS class
 * of near-ties falls the other way": an implementation that carries its constants in another precision resolves a tie between
 * two hull features the same way substep after substep, which neither a single flip nor noise on the state reproduces. */
static __thread int g_bias_mask = 0, g_bias_request = 0, g_bias_first = 0, g_bias_last = 1 << 30, g_bias_pass = 0;
static __thread real g_bias_eps = 0, g_bias_eps_rel = 0;
/* the bias applies to the collision passes first .. last (0-based, counted from this call: one pass per mjx.step / forward): a
 * foot that ROTATES THROUGH a tie within one env step crosses it in one substep only, a tie between two hull features persists */
void odko_set_tie_bias_window(int mask, real eps, real eps_rel, int first, int last) {
  g_bias_request = mask; g_bias_mask = 0; g_bias_eps = eps; g_bias_eps_rel = eps_rel; g_bias_first = first; g_bias_last = last; g_bias_pass = 0;
}
void odko_set_tie_bias(int mask, real eps, real eps_rel) { odko_set_tie_bias_window(mask, eps, eps_rel, 0, 1 << 30); }
#define BIAS(bit, gap) ((g_bias_mask & (bit)) && fabs(gap) < g_bias_eps)
/* the AREA0 cut itself is a decision: an area measure within rounding of the 1e-7 m^2 threshold counts as zero in one precision and as itself in the
 * other.  Tie class 1024: inside a band of eps x 1 cm around the threshold (a length error of eps on a lever of a centimetre) the cut falls the other
 * way.  (Found by the rollout-state test of round 6: a fallen robot's foot lying on its side in the terrain, a sliver manifold whose area measure
 * sat at 1.0e-7.) */
static real area0(real v) {
  int small = v < 1e-7;
  if ((g_bias_mask & 1024) && fabs(v - 1e-7) < g_bias_eps * 1e-2) small = !small;
  return small ? 0.0 : v;
}
static void margin_reset(void) { for (int k = 0; k < 4; k++) g_margin[k] = 1e30; }
static void margin_note(int k, real gap) { gap = fabs(gap); if (gap < g_margin[k]) g_margin[k] = gap; }
/* gap between the largest and the runner-up of v[0..n) among entries whose POINT differs from the winner's */
static int margin_argmax(int cat, const real* v, const real (*pt)[3], int n, int win, int relative) {   /* returns the index to use */
  real second = -1e30; int si = -1;
  for (int i = 0; i < n; i++) {
    if (i == win || v[i] < -1e5) continue;
    real t[3]; v3_sub(t, pt[i], pt[win]);
    if (v3_dot(t, t) < 1e-18) continue;
    if (v[i] > second) { second = v[i]; si = i; }
  }
  if (si < 0) return win;
  real gap = relative ? (v[win] - second) / (fabs(v[win]) > 1e-12 ? fabs(v[win]) : 1e-12) : v[win] - second;
  margin_note(cat, gap);
  if ((g_bias_mask & 32) && fabs(gap) < g_bias_eps_rel) return si;
  return win;
}
static void manifold_points(const real (*poly)[3], const int* mask, int n, const real* norm, int* idx) {
  real dm[ODKO_MAXHV];
  int ai = 0, bi = 0, ci = 0, di = 0;
  real best;
  for (int i = 0; i < n; i++) dm[i] = mask[i] ? 0.0 : -1e6;
  best = -1e30; for (int i = 0; i < n; i++) if (dm[i] > best) { best = dm[i]; ai = i; }
  const real* a = poly[ai];
  real vv[2 * ODKO_MAXHV];
  best = -1e30;
  for (int i = 0; i < n; i++) {
    real t[3]; v3_sub(t, a, poly[i]);
    real v = v3_dot(t, t) + dm[i];
    vv[i] = v;
    if (v > best) { best = v; bi = i; }
  }
  bi = margin_argmax(3, vv, poly, n, bi, 1);
  const real* b = poly[bi];
  real ab[3], t[3];
  v3_sub(t, a, b); v3_cross(ab, norm, t);
  best = -1e30;
  for (int i = 0; i < n; i++) {
    real ap[3]; v3_sub(ap, a, poly[i]);
    real v = AREA0(fabs(v3_dot(ap, ab))) + dm[i];
    vv[i] = v;
    if (v > best) { best = v; ci = i; }
  }
  ci = margin_argmax(3, vv, poly, n, ci, 1);
  const real* c = poly[ci];
  real ac[3], bc[3];
  v3_sub(t, a, c); v3_cross(ac, norm, t);
  v3_sub(t, b, c); v3_cross(bc, norm, t);
  /* argmax over concat([dist_bp, dist_ap]) % n : first half (bp) wins ties.  For a triangle of candidates (a, b, c) the two largest
   * entries -- p = a in the first half, p = b in the second -- are both twice its area: equal in exact arithmetic, apart by rounding
   * residue in floating point (in float64 as in float32: which of the two points ends up in two of the four slots, i.e. carries twice
   * the weight, would be a coin flip).  Values within AREA_TIE of the maximum count as the maximum; the lowest index wins. */
  real vd[2 * ODKO_MAXHV];
  best = -1e30;
  for (int i = 0; i < n; i++) {
    real bp[3]; v3_sub(bp, b, poly[i]);
    vd[i] = AREA0(fabs(v3_dot(bp, bc))) + dm[i];
    if (vd[i] > best) best = vd[i];
  }
  for (int i = 0; i < n; i++) {
    real ap[3]; v3_sub(ap, a, poly[i]);
    vd[n + i] = AREA0(fabs(v3_dot(ap, ac))) + dm[i];
    if (vd[n + i] > best) best = vd[n + i];
  }
  for (int i = 0; i < 2 * n; i++) if (vd[i] >= best - AREA_TIE) { di = i % n; break; }
  {   /* the tie rule makes values within AREA_TIE of the maximum equal: the margin is how far the nearest OTHER point's value is from that band */
    real second = -1e30;
    for (int i = 0; i < 2 * n; i++) {
      if (vd[i] < -1e5 || vd[i] >= best - AREA_TIE) continue;
      real t[3]; v3_sub(t, poly[i % n], poly[di]);
      if (v3_dot(t, t) < 1e-18) continue;
      if (vd[i] > second) second = vd[i];
    }
    int alt = -1; real altgap = 1e30;
    if (second > -1e29) {
      real g = ((best - AREA_TIE) - second) / (fabs(best) > 1e-12 ? fabs(best) : 1e-12);
      margin_note(3, g);
      if (g < altgap) { altgap = g; for (int i = 0; i < 2 * n; i++) if (vd[i] == second) { real t[3]; v3_sub(t, poly[i % n], poly[di]); if (v3_dot(t, t) >= 1e-18) { alt = i % n; break; } } }
    }
    for (int i = 0; i < 2 * n; i++) {   /* and a different point INSIDE the band with a lower index would have won */
      if (vd[i] < best - AREA_TIE || i % n == di) continue;
      real t[3]; v3_sub(t, poly[i % n], poly[di]);
      if (v3_dot(t, t) >= 1e-18) {
        real g = (vd[i] - (best - AREA_TIE)) / (fabs(best) > 1e-12 ? fabs(best) : 1e-12);
        margin_note(3, g);
        if (g < altgap) { altgap = g; alt = i % n; }
      }
    }
    if ((g_bias_mask & 32) && alt >= 0 && altgap < g_bias_eps_rel) di = alt;
  }
  idx[0] = ai; idx[1] = bi; idx[2] = ci; idx[3] = di;
}

/* mjx collision_convex.plane_convex against the plane (pos_w, pn_w); writes 4 contacts starting at slot c0 */
static void plane_convex_at(const odko_model* m, odko_data* d, const real* pos_w, const real* pn_w, int gc, int c0) {
  const real (*vert)[3] = &m->hull_vert[m->cgeom_vertadr[gc]];
  int n = m->cgeom_vertnum[gc];
  const real* cm = d->geom_xmat[gc];
  real rel[3], plane_pos[3], nl[3];
  v3_sub(rel, pos_w, d->geom_xpos[gc]);
  mat_tmulvec(plane_pos, cm, rel);
  mat_tmulvec(nl, cm, pn_w);
  real support[ODKO_MAXHV], smax = -1e30;
  int mask[ODKO_MAXHV], idx[4];
  for (int i = 0; i < n; i++) {
    real t[3]; v3_sub(t, plane_pos, vert[i]);
    support[i] = v3_dot(t, nl);
    if (support[i] > smax) smax = support[i];
  }
  real thr = smax - 1e-3; if (thr < 0) thr = 0;
  for (int i = 0; i < n; i++) mask[i] = support[i] > thr;
  manifold_points(vert, mask, n, nl, idx);
  real frame[9];
  make_frame(frame, pn_w);
  for (int k = 0; k < 4; k++) {
    /* unique = tril(idx == idx[:,None]).sum(axis=1) == 1 : first occurrence only */
    int unique = 1;
    for (int q = 0; q < k; q++) if (idx[q] == idx[k]) unique = 0;
    real dist = unique ? -support[idx[k]] : 1.0;
    real pw[3];
    mat_mulvec(pw, cm, vert[idx[k]]);
    v3_addscl(pw, pw, d->geom_xpos[gc], 1);
    v3_addscl(pw, pw, pn_w, -0.5 * dist);
    d->contact_dist[c0 + k] = dist;
    v3_copy(d->contact_pos[c0 + k], pw);
    memcpy(d->contact_frame[c0 + k], frame, sizeof(frame));
  }
}
static void plane_convex(const odko_model* m, odko_data* d, int gp, int gc, int c0) {
  real pn_w[3] = {d->geom_xmat[gp][2], d->geom_xmat[gp][5], d->geom_xmat[gp][8]};
  plane_convex_at(m, d, d->geom_xpos[gp], pn_w, gc, c0);
}

#include "odk_oracle_convex.inc"

static void hull_aabb(const odko_model* m, int g, real* c, real* h);
/* Round-2 approximation of the height-field floor, kept behind m->hfield_mode = 1 so that its difference to the prism algorithm
 * below can be measured (tests/test_oracle_physics.py): the terrain under the foot is replaced by the plane
 * of the height-field triangle below the hull's centre (cell (c, r) is split along the (c+1, r)-(c, r+1) diagonal)
 * and plane_convex runs against that plane.  Exact on flat patches; the terrain's slopes are <= 1 cm per 7.8 cm cell. */
static void hfield_plane(const odko_model* m, const odko_data* d, int gh, const real* point_w, real* pos_w, real* n_w) {
  const real* R = d->geom_xmat[gh];
  real rel[3], p[3];
  v3_sub(rel, point_w, d->geom_xpos[gh]);
  mat_tmulvec(p, R, rel);
  int nc = m->hfield_ncol, nr = m->hfield_nrow;
  real sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
  real dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
  real fx = (p[0] + sx) / dx, fy = (p[1] + sy) / dy;
  int c = (int)floor(fx), r = (int)floor(fy);
  if (c < 0) c = 0;
  if (c > nc - 2) c = nc - 2;
  if (r < 0) r = 0;
  if (r > nr - 2) r = nr - 2;
  real tx = fx - c, ty = fy - r;
  real x0 = -sx + c * dx, y0 = -sy + r * dy;
  real z00 = m->hfield_data[r * nc + c] * sz, z10 = m->hfield_data[r * nc + c + 1] * sz;
  real z01 = m->hfield_data[(r + 1) * nc + c] * sz, z11 = m->hfield_data[(r + 1) * nc + c + 1] * sz;
  real a[3], e1[3], e2[3], nl[3];
  if (tx + ty <= 1.0) { a[0] = x0; a[1] = y0; a[2] = z00; e1[0] = dx; e1[1] = 0; e1[2] = z10 - z00; e2[0] = 0; e2[1] = dy; e2[2] = z01 - z00; }
  else { a[0] = x0 + dx; a[1] = y0 + dy; a[2] = z11; e1[0] = -dx; e1[1] = 0; e1[2] = z01 - z11; e2[0] = 0; e2[1] = -dy; e2[2] = z10 - z11; }
  v3_cross(nl, e1, e2);
  v3_normalize(nl); /* e1 x e2 points up in both cases */
  mat_mulvec(n_w, R, nl);
  mat_mulvec(pos_w, R, a);
  v3_addscl(pos_w, pos_w, d->geom_xpos[gh], 1);
}
static void hfield_convex_one_triangle(const odko_model* m, odko_data* d, int gh, int gc, int c0) {
  real c[3], h[3], cw[3], pos_w[3], n_w[3];
  hull_aabb(m, gc, c, h);
  mat_mulvec(cw, d->geom_xmat[gc], c);
  v3_addscl(cw, cw, d->geom_xpos[gc], 1);
  hfield_plane(m, d, gh, cw, pos_w, n_w);
  plane_convex_at(m, d, pos_w, n_w, gc, c0);
}

/* mjx collision_convex.hfield_convex -> _hfield_collision (see odk_oracle_convex.inc): the foot against the prisms of every cell
 * under its bounding sphere, the four deepest contacts of all prisms kept (ties: lower candidate index, as lax.top_k), each with
 * the normal of its own prism test.  Candidate order: rows, then columns, then the two triangles of the cell, then the prism's
 * four manifold slots. */
#define HF_MAXCAND 256
static void hfield_convex(const odko_model* m, odko_data* d, int gh, int gc, int c0) {
  if (m->hfield_mode == 1) { hfield_convex_one_triangle(m, d, gh, gc, c0); return; }
  const real* Rh = d->geom_xmat[gh]; const real* ph = d->geom_xpos[gh];
  odko_convex Fw, F;
  mesh_convex_world(m, d, gc, &Fw);
  F = Fw; /* foot in the height field's frame */
  for (int i = 0; i < Fw.nv; i++) { real t[3]; v3_sub(t, Fw.v[i], ph); mat_tmulvec(F.v[i], Rh, t); }
  for (int f = 0; f < Fw.nf; f++) mat_tmulvec(F.fnorm[f], Rh, Fw.fnorm[f]);
  { real t[3]; v3_sub(t, Fw.c, ph); mat_tmulvec(F.c, Rh, t); }
  real bc[3], bh[3], cw[3], cl[3];
  hull_aabb(m, gc, bc, bh);
  mat_mulvec(cw, d->geom_xmat[gc], bc); v3_addscl(cw, cw, d->geom_xpos[gc], 1);
  { real t[3]; v3_sub(t, cw, ph); mat_tmulvec(cl, Rh, t); }
  real rad = sqrt(v3_dot(bh, bh));
  int nc = m->hfield_ncol, nr = m->hfield_nrow;
  real sx = m->hfield_size[0], sy = m->hfield_size[1];
  real dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
  int cmin = (int)floor((cl[0] - rad + sx) / dx), cmax = (int)floor((cl[0] + rad + sx) / dx);
  int rmin = (int)floor((cl[1] - rad + sy) / dy), rmax = (int)floor((cl[1] + rad + sy) / dy);
  if (cmin < 0) cmin = 0;
  if (rmin < 0) rmin = 0;
  if (cmax > nc - 2) cmax = nc - 2;
  if (rmax > nr - 2) rmax = nr - 2;
  real cd[HF_MAXCAND], cp[HF_MAXCAND][3], cn[HF_MAXCAND][3];
  int ncand = 0;
  for (int r = rmin; r <= rmax; r++)
    for (int c = cmin; c <= cmax; c++)
      for (int tri = 0; tri < 2; tri++) {
        if (ncand + 4 > HF_MAXCAND) continue;
        odko_convex P;
        hfield_prism(m, c, r, tri, &P);
        P.top_only = m->hfield_mode == 2;
        real dist4[4], pos4[4][3], nrm[3];
        real keep[4]; for (int k = 0; k < 4; k++) keep[k] = g_margin[k];
        margin_reset();
        convex_convex_sat(&P, &F, dist4, pos4, nrm, NULL);
        int live = 0;
        for (int k = 0; k < 4; k++) live |= dist4[k] < 0;
        for (int k = 0; k < 4; k++) g_margin[k] = (live && g_margin[k] < keep[k]) ? g_margin[k] : keep[k];     /* a separated prism decides nothing */
        if (m->hfield_mode == 3 && !(nrm[2] > 0.5)) for (int k = 0; k < 4; k++) dist4[k] = 1.0;              /* hypothesis sweep: upward normals only */
        if (m->hfield_mode == 4) {                                                                          /* hypothesis sweep: the prism's deepest contact alone */
          int kb = 0;
          for (int k = 1; k < 4; k++) if (dist4[k] < dist4[kb]) kb = k;
          for (int k = 0; k < 4; k++) if (k != kb) dist4[k] = 1.0;
        }
        for (int k = 0; k < 4; k++) { cd[ncand] = dist4[k]; v3_copy(cp[ncand], pos4[k]); v3_copy(cn[ncand], nrm); ncand++; }
      }
  int used[HF_MAXCAND] = {0};
  /* hypothesis sweep, mode 5 (the judge's unverified recollection of MJX's _hfield_collision): the pair's four contacts are chosen by
   * the plane-convex manifold heuristic over ALL prisms' active candidates with their mean normal (first active point, the farthest from it, the
   * farthest from that line, the farthest from that triangle: `manifold_points`) instead of the four deepest; a point picked twice counts once
   * (plane_convex marks duplicates inactive) */
  int sel5[4] = {-1, -1, -1, -1};
  if (m->hfield_mode == 5) {
    int act[ODKO_MAXHV], na = 0, mask[ODKO_MAXHV];
    real nm[3] = {0, 0, 0}, poly[ODKO_MAXHV][3];
    for (int i = 0; i < ncand && na < ODKO_MAXHV; i++) if (cd[i] < 0) { act[na] = i; mask[na] = 1; v3_copy(poly[na], cp[i]); v3_addscl(nm, nm, cn[i], 1); na++; }
    if (na > 0) {
      real l = sqrt(v3_dot(nm, nm));
      if (l > 1e-12) { nm[0] /= l; nm[1] /= l; nm[2] /= l; } else { nm[0] = 0; nm[1] = 0; nm[2] = 1; }
      int idx[4];
      manifold_points((const real (*)[3])poly, mask, na, nm, idx);
      for (int k = 0; k < 4; k++) { int dup = 0; for (int q = 0; q < k; q++) dup |= idx[q] == idx[k]; sel5[k] = dup ? -1 : act[idx[k]]; }
    }
  }
  for (int k = 0; k < 4; k++) {
    int bi = -1;
    if (m->hfield_mode == 5) bi = sel5[k];
    else for (int i = 0; i < ncand; i++) if (!used[i] && (bi < 0 || cd[i] < cd[bi])) bi = i;
    if (m->hfield_mode != 5 && bi >= 0 && cd[bi] < 0 && (k == 3 || (g_bias_mask & 64))) {
      /* the cut behind the fourth: the nearest active candidate that is a different contact and stays out.  Under the tie bias
       * (class 64) EVERY pick prefers a different contact within the band: two one-point manifolds of equal depth from neighbouring
       * prisms fill the last two slots with copies of one OR the other */
      int alt = -1;
      for (int i = 0; i < ncand; i++) {
        if (used[i] || i == bi || cd[i] >= 0) continue;
        real t[3], tn[3]; v3_sub(t, cp[i], cp[bi]); v3_sub(tn, cn[i], cn[bi]);   /* (the same point from two prisms is two contacts when their normals differ) */
        if (v3_dot(t, t) > 1e-18 || fabs(cd[i] - cd[bi]) > 0 || v3_dot(tn, tn) > 1e-12) { if (k == 3) margin_note(0, cd[i] - cd[bi]); if (alt < 0 || cd[i] < cd[alt]) alt = i; }
      }
      if (alt >= 0 && BIAS(64, cd[alt] - cd[bi])) bi = alt;
    }
    real nw[3] = {Rh[2], Rh[5], Rh[8]}, pw[3] = {0, 0, 0};
    real dist = 1.0;
    if (bi >= 0) {
      used[bi] = 1; dist = cd[bi];
      mat_mulvec(nw, Rh, cn[bi]);
      mat_mulvec(pw, Rh, cp[bi]); v3_addscl(pw, pw, ph, 1);
    }
    d->contact_dist[c0 + k] = dist;
    v3_copy(d->contact_pos[c0 + k], pw);
    make_frame(d->contact_frame[c0 + k], nw);
  }
}

/* Oriented-bounding-box cull (15-axis SAT on the hull AABBs expressed in each geom frame).  A positive
 * return value is a lower bound on the true separation of the two hulls: the pair is then inactive
 * in MJX as well (all dist > 0), so culling is parity-safe. */
static void hull_aabb(const odko_model* m, int g, real* c, real* h) { /* (declared above) */
  real lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
  for (int i = 0; i < m->cgeom_vertnum[g]; i++)
    for (int k = 0; k < 3; k++) {
      real v = m->hull_vert[m->cgeom_vertadr[g] + i][k];
      if (v < lo[k]) lo[k] = v;
      if (v > hi[k]) hi[k] = v;
    }
  for (int k = 0; k < 3; k++) { c[k] = 0.5 * (lo[k] + hi[k]); h[k] = 0.5 * (hi[k] - lo[k]); }
}
static real obb_separation(const odko_model* m, const odko_data* d, int g1, int g2) {
  real c1[3], h1[3], c2[3], h2[3], w1[3], w2[3], t[3], best = -1e30;
  hull_aabb(m, g1, c1, h1); hull_aabb(m, g2, c2, h2);
  const real *R1 = d->geom_xmat[g1], *R2 = d->geom_xmat[g2];
  mat_mulvec(w1, R1, c1); v3_addscl(w1, w1, d->geom_xpos[g1], 1);
  mat_mulvec(w2, R2, c2); v3_addscl(w2, w2, d->geom_xpos[g2], 1);
  v3_sub(t, w2, w1);
  { /* bounding spheres first: a positive gap already makes the pair inactive (and is what the kernels report) */
    real sph = sqrt(v3_dot(t, t)) - sqrt(v3_dot(h1, h1)) - sqrt(v3_dot(h2, h2));
    if (sph > 0) return sph;
  }
  real ax[15][3];
  int na = 0;
  for (int k = 0; k < 3; k++) { ax[na][0] = R1[k]; ax[na][1] = R1[3 + k]; ax[na][2] = R1[6 + k]; na++; }
  for (int k = 0; k < 3; k++) { ax[na][0] = R2[k]; ax[na][1] = R2[3 + k]; ax[na][2] = R2[6 + k]; na++; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      v3_cross(ax[na], ax[i], ax[3 + j]);
      real n = sqrt(v3_dot(ax[na], ax[na]));
      if (n < 1e-6) continue; /* parallel edges: covered by the face axes */
      ax[na][0] /= n; ax[na][1] /= n; ax[na][2] /= n;
      na++;
    }
  for (int a = 0; a < na; a++) {
    real r1 = 0, r2 = 0;
    for (int k = 0; k < 3; k++) { r1 += h1[k] * fabs(v3_dot(ax[a], ax[k])); r2 += h2[k] * fabs(v3_dot(ax[a], ax[3 + k])); }
    real sep = fabs(v3_dot(t, ax[a])) - r1 - r2;
    if (sep > best) best = sep;
  }
  return best;
}

/* convex-convex (foot vs foot): mjx collision_convex.convex_convex (odk_oracle_convex.inc).  Bounding spheres / boxes with a
 * positive gap are culled first: MJX would report dist > 0 on all four slots, which adds nothing to the dynamics (inactive rows
 * have J = 0); slot 0 then carries the gap. */
static void convex_convex(const odko_model* m, odko_data* d, int g1, int g2, int c0) {
  real sep = obb_separation(m, d, g1, g2);
  if (sep > 0) {
    real axis[3] = {0, 0, 1}, frame[9];
    make_frame(frame, axis);
    for (int k = 0; k < 4; k++) {
      d->contact_dist[c0 + k] = (k == 0) ? sep : 1.0;
      for (int q = 0; q < 3; q++) d->contact_pos[c0 + k][q] = 0.5 * (d->geom_xpos[g1][q] + d->geom_xpos[g2][q]);
      memcpy(d->contact_frame[c0 + k], frame, sizeof(frame));
    }
    return;
  }
  odko_convex A, B;
  mesh_convex_world(m, d, g1, &A); mesh_convex_world(m, d, g2, &B);
  real dist4[4], pos4[4][3], nrm[3], frame[9];
  convex_convex_sat(&A, &B, dist4, pos4, nrm, NULL);
  make_frame(frame, nrm);
  for (int k = 0; k < 4; k++) {
    d->contact_dist[c0 + k] = dist4[k];
    v3_copy(d->contact_pos[c0 + k], pos4[k]);
    memcpy(d->contact_frame[c0 + k], frame, sizeof(frame));
  }
}

/* test entry: two polytopes given as vertices + outward triangles and poses (pos[3], row-major mat[9]); out = dist[4], pos[12],
 * normal[3], sat[3] = (best face separation of A, of B, best Minkowski-edge separation), kind (0 reference A, 1 reference B, 2 edge) */
int odko_convex_pair(const real* va, int nva, const int* ta, int nta, const real* pa, const real* ma, const real* vb, int nvb, const int* tb,
                     int ntb, const real* pb, const real* mb, real* dist4, real* pos12, real* normal3, real* sat3) {
  if (nva > CV_MAXV || nvb > CV_MAXV || nta > CV_MAXF || ntb > CV_MAXF) return -1;
  odko_convex LA, LB, A, B;
  convex_from_tris(&LA, (const real (*)[3])va, nva, (const int (*)[3])ta, nta); convex_from_tris(&LB, (const real (*)[3])vb, nvb, (const int (*)[3])tb, ntb);
  convex_transform(&A, &LA, pa, ma); convex_transform(&B, &LB, pb, mb);
  odko_sat S;
  convex_convex_sat(&A, &B, dist4, (real (*)[3])pos12, normal3, &S);
  sat3[0] = S.sep_a; sat3[1] = S.sep_b; sat3[2] = S.sep_e;
  return S.kind;
}
/* test entry: face / edge counts of mesh geom g after the coplanar merge, and of a height-field prism */
int odko_model_convex_counts(const odko_model* m, int g, int* nv, int* nf, int* ne) {
  if (g < 0 || g >= m->ncgeom) return -1;
  *nv = m->cgeom_convex[g].nv; *nf = m->cgeom_convex[g].nf; *ne = m->cgeom_convex[g].ne;
  return 0;
}

/* ---- primitive colliders (mjx collision_primitive.py, restated from memory like the rest: plane_sphere, plane_capsule,
 * sphere_sphere, sphere_capsule, capsule_capsule).  One or two contacts; the unused slots of a pair get dist = 1. */
static void prim_fill(odko_data* d, int c0, int k, real dist, const real* pos, const real* frame) {
  d->contact_dist[c0 + k] = dist;
  v3_copy(d->contact_pos[c0 + k], pos);
  memcpy(d->contact_frame[c0 + k], frame, 9 * sizeof(real));
}
static void prim_pad(odko_data* d, int c0, int from, const real* frame) {
  real z[3] = {0, 0, 0};
  for (int k = from; k < 4; k++) prim_fill(d, c0, k, 1.0, z, frame);
}
static void plane_sphere_at(const real* n, const real* ppos, const real* spos, real radius, real* dist, real* pos) { /* _plane_sphere */
  real t[3]; v3_sub(t, spos, ppos);
  *dist = v3_dot(t, n) - radius;
  v3_addscl(pos, spos, n, -(radius + 0.5 * *dist));
}
static void plane_sphere(const odko_model* m, odko_data* d, int gp, int gs, int c0) {
  real n[3] = {d->geom_xmat[gp][2], d->geom_xmat[gp][5], d->geom_xmat[gp][8]}, dist, pos[3], frame[9];
  plane_sphere_at(n, d->geom_xpos[gp], d->geom_xpos[gs], m->cgeom_size[gs][0], &dist, pos);
  make_frame(frame, n);
  prim_fill(d, c0, 0, dist, pos, frame);
  prim_pad(d, c0, 1, frame);
}
static void plane_capsule(const odko_model* m, odko_data* d, int gp, int gc, int c0) {
  real n[3] = {d->geom_xmat[gp][2], d->geom_xmat[gp][5], d->geom_xmat[gp][8]};
  real axis[3] = {d->geom_xmat[gc][2], d->geom_xmat[gc][5], d->geom_xmat[gc][8]};
  /* contact frame aligned with the capsule axis: b = axis - n (n . axis), falling back to y / z when the capsule stands on end */
  real b[3], frame[9];
  v3_addscl(b, axis, n, -v3_dot(n, axis));
  real bn = sqrt(v3_dot(b, b));
  if (bn < 0.5) { b[0] = 0; b[1] = (-0.5 < n[1] && n[1] < 0.5) ? 1 : 0; b[2] = (-0.5 < n[1] && n[1] < 0.5) ? 0 : 1; }
  else { b[0] /= bn; b[1] /= bn; b[2] /= bn; }
  v3_copy(frame, n); v3_copy(frame + 3, b); v3_cross(frame + 6, n, b);
  for (int k = 0; k < 2; k++) {
    real end[3], dist, pos[3];
    v3_addscl(end, d->geom_xpos[gc], axis, (k == 0 ? 1.0 : -1.0) * m->cgeom_size[gc][1]);
    plane_sphere_at(n, d->geom_xpos[gp], end, m->cgeom_size[gc][0], &dist, pos);
    prim_fill(d, c0, k, dist, pos, frame);
  }
  prim_pad(d, c0, 2, frame);
}
static void sphere_sphere_at(odko_data* d, int c0, const real* p1, real r1, const real* p2, real r2) { /* _sphere_sphere */
  real n[3], frame[9], pos[3];
  v3_sub(n, p2, p1);
  real len = sqrt(v3_dot(n, n));
  if (len < MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; } else { n[0] /= len; n[1] /= len; n[2] /= len; }
  real dist = len - (r1 + r2);
  v3_addscl(pos, p1, n, r1 + 0.5 * dist);
  make_frame(frame, n);
  prim_fill(d, c0, 0, dist, pos, frame);
  prim_pad(d, c0, 1, frame);
}
static void closest_segment_point(real* out, const real* a, const real* b, const real* pt) { /* math.closest_segment_point */
  real ab[3], t[3];
  v3_sub(ab, b, a); v3_sub(t, pt, a);
  real tt = v3_dot(t, ab) / (v3_dot(ab, ab) + 1e-6);
  tt = tt < 0 ? 0 : (tt > 1 ? 1 : tt);
  v3_addscl(out, a, ab, tt);
}
static void closest_segment_to_segment(real* best_a, real* best_b, const real* a0, const real* a1, const real* b0, const real* b1) {
  /* math.closest_segment_to_segment_points: closest points of the two (infinite) lines, clamped to the segments, then each
   * re-projected on the other segment; the pair of the two candidates with the smaller distance */
  real dir_a[3], dir_b[3], half_a[3], half_b[3], amid[3], bmid[3];
  v3_sub(dir_a, a1, a0); real len_a = v3_normalize(dir_a) * 0.5; (void)half_a; (void)half_b;
  v3_sub(dir_b, b1, b0); real len_b = v3_normalize(dir_b) * 0.5;
  for (int k = 0; k < 3; k++) { amid[k] = 0.5 * (a0[k] + a1[k]); bmid[k] = 0.5 * (b0[k] + b1[k]); }
  real diff[3]; v3_sub(diff, amid, bmid);
  real dot_a = v3_dot(dir_a, diff), dot_b = v3_dot(dir_b, diff), dot_ab = v3_dot(dir_a, dir_b);
  real denom = 1.0 - dot_ab * dot_ab;
  real orig_t_a = (-dot_a + dot_ab * dot_b) / (denom + 1e-6);
  real orig_t_b = dot_b + orig_t_a * dot_ab;
  real t_a = orig_t_a < -len_a ? -len_a : (orig_t_a > len_a ? len_a : orig_t_a);
  real t_b = orig_t_b < -len_b ? -len_b : (orig_t_b > len_b ? len_b : orig_t_b);
  real ca[3], cb[3], new_a[3], new_b[3];
  v3_addscl(ca, amid, dir_a, t_a); v3_addscl(cb, bmid, dir_b, t_b);
  closest_segment_point(new_a, a0, a1, cb);     /* the clamping moved a point: re-project each on the other segment ... */
  closest_segment_point(new_b, b0, b1, ca);
  real t1[3], t2[3];
  v3_sub(t1, new_a, cb); v3_sub(t2, ca, new_b);
  if (v3_dot(t1, t1) < v3_dot(t2, t2)) { v3_copy(best_a, new_a); v3_copy(best_b, cb); } /* ... and keep the closer pair */
  else { v3_copy(best_a, ca); v3_copy(best_b, new_b); }
}
static void capsule_ends(const odko_model* m, const odko_data* d, int g, real* e0, real* e1) {
  real axis[3] = {d->geom_xmat[g][2], d->geom_xmat[g][5], d->geom_xmat[g][8]};
  v3_addscl(e0, d->geom_xpos[g], axis, -m->cgeom_size[g][1]); v3_addscl(e1, d->geom_xpos[g], axis, m->cgeom_size[g][1]);
}
static void sphere_capsule(const odko_model* m, odko_data* d, int gs, int gc, int c0, int flip) {
  real e0[3], e1[3], pt[3];
  capsule_ends(m, d, gc, e0, e1);
  closest_segment_point(pt, e0, e1, d->geom_xpos[gs]);
  if (!flip) sphere_sphere_at(d, c0, d->geom_xpos[gs], m->cgeom_size[gs][0], pt, m->cgeom_size[gc][0]);
  else sphere_sphere_at(d, c0, pt, m->cgeom_size[gc][0], d->geom_xpos[gs], m->cgeom_size[gs][0]);
}
static void capsule_capsule(const odko_model* m, odko_data* d, int g1, int g2, int c0) {
  real a0[3], a1[3], b0[3], b1[3], pa[3], pb[3];
  capsule_ends(m, d, g1, a0, a1); capsule_ends(m, d, g2, b0, b1);
  closest_segment_to_segment(pa, pb, a0, a1, b0, b1);
  sphere_sphere_at(d, c0, pa, m->cgeom_size[g1][0], pb, m->cgeom_size[g2][0]);
}

/* ---- sphere / capsule against a convex polytope (mjx collision_convex._sphere_convex, _capsule_convex: restated from memory,
 * PARITY UNPINNED like the rest of the file), both in the polytope's frame.  The normals returned here point from the polytope to
 * the primitive: the order of the pair (height field, primitive). */
static void sphere_convex_at(const odko_convex* C, const real* s, real r, real* dist, real* pos, real* n) {
  real best = -1e300; int bf = 0;
  for (int f = 0; f < C->nf; f++) { /* the face of least penetration among those the sphere is behind ("has support") */
    real t[3]; v3_sub(t, s, C->v[C->fidx[f][0]]);
    real sup = v3_dot(t, C->fnorm[f]) - r;
    if (sup >= 0) sup = -1e12;
    if (sup > best) { best = sup; bf = f; }
  }
  const real* N = C->fnorm[bf]; int cnt = C->fcnt[bf];
  real pt[3], t[3];
  v3_sub(t, s, C->v[C->fidx[bf][0]]);
  v3_addscl(pt, s, N, -v3_dot(t, N)); /* the centre projected on the face plane */
  int inside = 1, idx = 0; real dmin = 1e300;
  for (int k = 0; k < cnt; k++) { /* edge k runs from vertex k - 1 to vertex k (jp.roll(face, 1)) */
    const real* p0 = C->v[C->fidx[bf][(k + cnt - 1) % cnt]]; const real* p1 = C->v[C->fidx[bf][k]];
    real e[3], en[3], tp[3];
    v3_sub(e, p1, p0); v3_cross(en, e, N); v3_sub(tp, pt, p0);
    real ed = v3_dot(tp, en);
    if (!(ed <= 0)) inside = 0;
    int degenerate = en[0] == 0 && en[1] == 0 && en[2] == 0;
    real val = (degenerate || ed < 0) ? 1e12 : ed;
    if (val < dmin) { dmin = val; idx = k; }
  }
  if (!inside) { /* outside the polygon: the closest point of the nearest edge the projection is in front of */
    real q[3];
    closest_segment_point(q, C->v[C->fidx[bf][(idx + cnt - 1) % cnt]], C->v[C->fidx[bf][idx]], pt);
    v3_copy(pt, q);
  }
  real nn[3]; v3_sub(nn, pt, s);
  real d = sqrt(v3_dot(nn, nn)), inv = 1.0 / (d + (d == 0 ? 1e-6 : 0.0)); /* math.normalize_with_norm */
  for (int k = 0; k < 3; k++) nn[k] *= inv;
  *dist = d - r;
  for (int k = 0; k < 3; k++) { pos[k] = 0.5 * (pt[k] + s[k] + nn[k] * r); n[k] = -nn[k]; }
}
/* capsule = segment (a, b) with radius r; two contacts */
static void capsule_convex_at(const odko_convex* C, const real* a, const real* b, real r, real* dist2, real (*pos2)[3], real (*n2)[3]) {
  real best = -1e300; int bf = 0, has_support = 1;
  for (int f = 0; f < C->nf; f++) {
    real ta[3], tb[3]; v3_sub(ta, a, C->v[C->fidx[f][0]]); v3_sub(tb, b, C->v[C->fidx[f][0]]);
    real sa = v3_dot(ta, C->fnorm[f]) - r, sb = v3_dot(tb, C->fnorm[f]) - r, sup = sa < sb ? sa : sb;
    if (!(sup < 0)) has_support = 0;
    if (sup >= 0) sup = -1e12;
    if (sup > best) { best = sup; bf = f; }
  }
  const real* N = C->fnorm[bf]; int cnt = C->fcnt[bf];
  real ppt[CV_MAXP][3], pn[CV_MAXP][3], clipped[2][3];
  for (int k = 0; k < cnt; k++) {
    const real* p0 = C->v[C->fidx[bf][(k + cnt - 1) % cnt]]; const real* p1 = C->v[C->fidx[bf][k]];
    real e[3]; v3_sub(e, p1, p0); v3_cross(pn[k], e, N); v3_copy(ppt[k], p0);
  }
  int mask = clip_edge_to_planes(a, b, (const real (*)[3])ppt, (const real (*)[3])pn, cnt, clipped);
  real face_pen[2];
  for (int k = 0; k < 2; k++) {
    real cp[3], fp[3], t[3];
    v3_addscl(cp, clipped[k], N, -r);                       /* the capsule's surface point under the clipped axis point */
    v3_sub(t, cp, C->v[C->fidx[bf][0]]);
    v3_addscl(fp, cp, N, -v3_dot(t, N));                    /* projected on the face plane */
    for (int q = 0; q < 3; q++) { pos2[k][q] = 0.5 * (cp[q] + fp[q]); n2[k][q] = N[q]; }
    v3_sub(t, fp, cp);
    face_pen[k] = (mask && has_support) ? v3_dot(t, N) : -1.0;
  }
  /* a shallow edge contact: the polytope edge closest to the capsule's axis */
  real e_dist = 1e300, e_axis[3] = {0, 0, 1}, e_pt[3] = {0, 0, 0}, c_pt[3] = {0, 0, 0}; int e_deg = 1, e_idx = 0;
  for (int k = 0; k < C->ne; k++) {
    real pe[3], pc[3], dir[3];
    closest_segment_to_segment(pe, pc, C->v[C->e[k][0]], C->v[C->e[k][1]], a, b);
    v3_sub(dir, pe, pc);
    real d2 = v3_dot(dir, dir), dd = sqrt(d2);
    if (dd < e_dist) {
      e_dist = dd; e_idx = k; e_deg = d2 < 1e-6;
      real inv = 1.0 / (dd + (dd == 0 ? 1e-6 : 0.0));
      for (int q = 0; q < 3; q++) e_axis[q] = dir[q] * inv;
      v3_copy(e_pt, pe); v3_copy(c_pt, pc);
    }
  }
  int voronoi_front = v3_dot(C->fnorm[C->ef[e_idx][0]], e_axis) < 0 && v3_dot(C->fnorm[C->ef[e_idx][1]], e_axis) < 0;
  int shallow = !e_deg && voronoi_front;
  real edge_pen = shallow ? r - e_dist : -1.0;
  int parallel = fabs(v3_dot(e_axis, N)) > 0.99 && has_support;
  real min_face = face_pen[0] < face_pen[1] ? face_pen[0] : face_pen[1];
  int has_edge = edge_pen > 0 && (min_face > 0 ? edge_pen < min_face : 1) && !parallel;
  if (has_edge) {
    for (int q = 0; q < 3; q++) { pos2[0][q] = 0.5 * (e_pt[q] + c_pt[q] + e_axis[q] * r); n2[0][q] = -e_axis[q]; }
    face_pen[0] = edge_pen; face_pen[1] = -1.0;
  }
  dist2[0] = -face_pen[0]; dist2[1] = -face_pen[1];
}

/* mjx collision_convex.hfield_sphere / hfield_capsule -> _hfield_collision: the primitive against the prisms of every cell under its
 * bounding sphere; the deepest contact (sphere) / the two deepest (capsule) of all prisms kept, ties to the lower candidate index.
 * Candidate order as in hfield_convex: rows, columns, the cell's two triangles, the prism's slots. */
static void hfield_prim(const odko_model* m, odko_data* d, int gh, int g, int c0) {
  const real* Rh = d->geom_xmat[gh]; const real* ph = d->geom_xpos[gh];
  const int capsule = m->cgeom_type[g] == ODKO_GEOM_CAPSULE, ncon = capsule ? 2 : 1;
  const real r = m->cgeom_size[g][0], hl = capsule ? m->cgeom_size[g][1] : 0.0;
  real cl[3], al[3] = {0, 0, 1}, t[3];
  v3_sub(t, d->geom_xpos[g], ph); mat_tmulvec(cl, Rh, t);
  if (capsule) { real aw[3] = {d->geom_xmat[g][2], d->geom_xmat[g][5], d->geom_xmat[g][8]}; mat_tmulvec(al, Rh, aw); }
  real ea[3], eb[3];
  v3_addscl(ea, cl, al, -hl); v3_addscl(eb, cl, al, hl);
  const real rad = r + hl;
  int nc = m->hfield_ncol, nr = m->hfield_nrow;
  real sx = m->hfield_size[0], sy = m->hfield_size[1];
  real dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
  int cmin = (int)floor((cl[0] - rad + sx) / dx), cmax = (int)floor((cl[0] + rad + sx) / dx);
  int rmin = (int)floor((cl[1] - rad + sy) / dy), rmax = (int)floor((cl[1] + rad + sy) / dy);
  if (cmin < 0) cmin = 0;
  if (rmin < 0) rmin = 0;
  if (cmax > nc - 2) cmax = nc - 2;
  if (rmax > nr - 2) rmax = nr - 2;
  real cd[HF_MAXCAND], cp[HF_MAXCAND][3], cn[HF_MAXCAND][3];
  int ncand = 0;
  for (int rr = rmin; rr <= rmax; rr++)
    for (int cc = cmin; cc <= cmax; cc++)
      for (int tri = 0; tri < 2; tri++) {
        if (ncand + 2 > HF_MAXCAND) continue;
        odko_convex P;
        hfield_prism(m, cc, rr, tri, &P);
        if (capsule) {
          real d2[2], p2[2][3], n2[2][3];
          capsule_convex_at(&P, ea, eb, r, d2, p2, n2);
          for (int k = 0; k < 2; k++) { cd[ncand] = d2[k]; v3_copy(cp[ncand], p2[k]); v3_copy(cn[ncand], n2[k]); ncand++; }
        } else {
          sphere_convex_at(&P, cl, r, &cd[ncand], cp[ncand], cn[ncand]);
          ncand++;
        }
      }
  int used[HF_MAXCAND] = {0};
  for (int k = 0; k < 4; k++) {
    int bi = -1;
    if (k < ncon) for (int i = 0; i < ncand; i++) if (!used[i] && (bi < 0 || cd[i] < cd[bi])) bi = i;
    real nw[3] = {Rh[2], Rh[5], Rh[8]}, pw[3] = {0, 0, 0}, dist = 1.0;
    if (bi >= 0) {
      used[bi] = 1; dist = cd[bi];
      mat_mulvec(nw, Rh, cn[bi]);
      mat_mulvec(pw, Rh, cp[bi]); v3_addscl(pw, pw, ph, 1);
    }
    d->contact_dist[c0 + k] = dist;
    v3_copy(d->contact_pos[c0 + k], pw);
    make_frame(d->contact_frame[c0 + k], nw);
  }
}

static void collision_pairs(const odko_model* m, odko_data* d);
static void collision(const odko_model* m, odko_data* d) {
  margin_reset();
  g_bias_mask = (g_bias_pass >= g_bias_first && g_bias_pass <= g_bias_last) ? g_bias_request : 0;
  g_bias_pass++;
  collision_pairs(m, d);
  g_bias_mask = 0;
  for (int k = 0; k < 4; k++) if (g_margin[k] < d->decision_margin[k]) d->decision_margin[k] = g_margin[k];   /* min since the caller last reset it */
}
static void collision_pairs(const odko_model* m, odko_data* d) {
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_g1[p], g2 = m->pair_g2[p], c0 = d->ncon;
    const int t1 = m->cgeom_type[g1], t2 = m->cgeom_type[g2];
    if (t1 == ODKO_GEOM_PLANE && t2 == ODKO_GEOM_SPHERE) plane_sphere(m, d, g1, g2, c0);
    else if (t1 == ODKO_GEOM_PLANE && t2 == ODKO_GEOM_CAPSULE) plane_capsule(m, d, g1, g2, c0);
    else if (t1 == ODKO_GEOM_SPHERE && t2 == ODKO_GEOM_SPHERE) sphere_sphere_at(d, c0, d->geom_xpos[g1], m->cgeom_size[g1][0], d->geom_xpos[g2], m->cgeom_size[g2][0]);
    else if (t1 == ODKO_GEOM_SPHERE && t2 == ODKO_GEOM_CAPSULE) sphere_capsule(m, d, g1, g2, c0, 0);
    else if (t1 == ODKO_GEOM_CAPSULE && t2 == ODKO_GEOM_SPHERE) sphere_capsule(m, d, g2, g1, c0, 1);
    else if (t1 == ODKO_GEOM_CAPSULE && t2 == ODKO_GEOM_CAPSULE) capsule_capsule(m, d, g1, g2, c0);
    else if (m->cgeom_type[g1] == ODKO_GEOM_PLANE && m->cgeom_type[g2] == ODKO_GEOM_MESH) plane_convex(m, d, g1, g2, c0);
    else if (m->cgeom_type[g1] == ODKO_GEOM_MESH && m->cgeom_type[g2] == ODKO_GEOM_MESH) convex_convex(m, d, g1, g2, c0);
    else if (m->cgeom_type[g1] == ODKO_GEOM_HFIELD && m->cgeom_type[g2] == ODKO_GEOM_MESH && m->hfield_nrow > 1) hfield_convex(m, d, g1, g2, c0);
    else if (t1 == ODKO_GEOM_HFIELD && (t2 == ODKO_GEOM_SPHERE || t2 == ODKO_GEOM_CAPSULE) && m->hfield_nrow > 1) hfield_prim(m, d, g1, g2, c0);
    else { /* unsupported pair type: no contact */
      for (int k = 0; k < 4; k++) { d->contact_dist[c0 + k] = 1.0; v3_zero(d->contact_pos[c0 + k]); real z[3] = {0, 0, 1}; make_frame(d->contact_frame[c0 + k], z); }
    }
    /* friction: higher priority wins, else max (mj_contactParam) */
    real f1 = m->cgeom_friction[g1][0], f2 = m->cgeom_friction[g2][0], fr;
    if (m->cgeom_priority[g1] > m->cgeom_priority[g2]) fr = f1;
    else if (m->cgeom_priority[g2] > m->cgeom_priority[g1]) fr = f2;
    else fr = f1 > f2 ? f1 : f2;
    for (int k = 0; k < 4; k++) { d->contact_friction[c0 + k] = fr; d->contact_geom1[c0 + k] = g1; d->contact_geom2[c0 + k] = g2; }
    d->ncon += 4;
  }
}

/* ---- constraints (mjx constraint.make_constraint) ---- */
static void efc_row_params_imp(const odko_model* m, odko_data* d, int r, real pos, real pos_imp, real invweight, const real* solref, const real* solimp,
                               real vel, real frictionloss);
static void efc_row_params(const odko_model* m, odko_data* d, int r, real pos, real invweight, const real* solref, const real* solimp,
                           real vel, real frictionloss) {
  efc_row_params_imp(m, d, r, pos, pos, invweight, solref, solimp, vel, frictionloss);
}
/* pos: the row's own residual (enters aref); pos_imp: the distance the impedance is evaluated at -- the same number for scalar rows, the
 * NORM of the residual vector for the 3 rows of a connect / the 6 rows of a weld (MuJoCo getposdim; MJX constraint._row's pos_imp) */
static void efc_row_params_imp(const odko_model* m, odko_data* d, int r, real pos, real pos_imp, real invweight, const real* solref, const real* solimp,
                               real vel, real frictionloss) {
  real timeconst = solref[0], dampratio = solref[1];
  real dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (timeconst < 2 * m->timestep) timeconst = 2 * m->timestep; /* refsafe */
  dmin = fmin(fmax(dmin, MINIMP), MAXIMP); dmax = fmin(fmax(dmax, MINIMP), MAXIMP);
  width = fmax(width, MINVAL); mid = fmin(fmax(mid, MINIMP), MAXIMP); power = fmax(power, 1.0);
  real k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  real b = 2.0 / (dmax * timeconst);
  if (solref[0] <= 0) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  real imp_x = fabs(pos_imp) / width;
  real imp_a = (1.0 / pow(mid, power - 1)) * pow(imp_x, power);
  real imp_b = 1 - (1.0 / pow(1 - mid, power - 1)) * pow(1 - imp_x, power);
  real imp_y = imp_x < mid ? imp_a : imp_b;
  real imp = dmin + imp_y * (dmax - dmin);
  imp = fmin(fmax(imp, dmin), dmax);
  if (imp_x > 1.0) imp = dmax;
  real R = fmax(invweight * (1 - imp) / imp, MINVAL);
  d->efc_pos[r] = pos; d->efc_invweight[r] = invweight; d->efc_imp[r] = imp; d->efc_k[r] = k; d->efc_b[r] = b;
  d->efc_R[r] = R; d->efc_D[r] = 1.0 / R;
  d->efc_aref[r] = -b * vel - k * imp * pos;
  d->efc_frictionloss[r] = frictionloss;
}

static void contact_mix(const odko_model* m, int g1, int g2, real* solref, real* solimp) {
  /* mj_contactParam: priority wins; otherwise solmix-weighted average */
  real mix;
  if (m->cgeom_priority[g1] > m->cgeom_priority[g2]) mix = 1;
  else if (m->cgeom_priority[g2] > m->cgeom_priority[g1]) mix = 0;
  else {
    real s1 = m->cgeom_solmix[g1], s2 = m->cgeom_solmix[g2];
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
  }
  for (int i = 0; i < 2; i++) solref[i] = mix * m->cgeom_solref[g1][i] + (1 - mix) * m->cgeom_solref[g2][i];
  for (int i = 0; i < 5; i++) solimp[i] = mix * m->cgeom_solimp[g1][i] + (1 - mix) * m->cgeom_solimp[g2][i];
}

/* translational Jacobian of a world point attached to body b (support.jac) */
static void jac_point(const odko_model* m, const odko_data* d, int b, const real* point, real* jacp /* 3 x nv */) {
  int nv = m->nv;
  memset(jacp, 0, 3 * (size_t)nv * sizeof(real));
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b == 0) return;
  real off[3];
  v3_sub(off, point, d->subtree_com[m->body_rootid[b]]);
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    real t[3];
    v3_cross(t, d->cdof[i], off);
    for (int k = 0; k < 3; k++) jacp[k * nv + i] = d->cdof[i][3 + k] + t[k];
  }
}

/* rotational Jacobian of body b (support.jac's jacr): column i = the angular part of dof i's motion axis, for the dofs above b */
static void jac_rot(const odko_model* m, const odko_data* d, int b, real* jacr /* 3 x nv */) {
  int nv = m->nv;
  memset(jacr, 0, 3 * (size_t)nv * sizeof(real));
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b == 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i])
    for (int k = 0; k < 3; k++) jacr[k * nv + i] = d->cdof[i][k];
}

/* Equality rows (mjx constraint._efc_equality_connect / _weld / _joint, MuJoCo mj_instantiateEquality; [UPSTREAM-MEMORY] like the rest of
 * the physics: parity unpinned; tests/test_oracle_equality.py holds the rows to their own definitions -- Jacobian = derivative of the
 * residual, residual decays at the solref rate, a connect carries the weight).  Order as MJX builds them: connects, welds, joints.
 * Always active, quadratic cost.  Returns the next free row. */
static int make_equality(const odko_model* m, odko_data* d, int r) {
  int nv = m->nv;
  for (int pass = 0; pass < 3; pass++)
    for (int e = 0; e < m->neq; e++) {
      if (!m->eq_active[e] || m->eq_type[e] != pass) continue;
      const real* data = m->eq_data[e];
      if (pass == ODKO_EQ_JOINT) {
        int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
        int i1 = m->jnt_dofadr[j1], q1 = m->jnt_qposadr[j1];
        real pos = d->qpos[q1] - m->qpos0[q1], invw = m->dof_invweight0[i1], vel = d->qvel[i1];
        d->efc_J[r * nv + i1] = 1;
        if (j2 >= 0) {
          int i2 = m->jnt_dofadr[j2], q2 = m->jnt_qposadr[j2];
          real x = d->qpos[q2] - m->qpos0[q2];
          real poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
          real dpoly = data[1] + x * (2 * data[2] + x * (3 * data[3] + x * 4 * data[4]));
          pos -= poly;
          d->efc_J[r * nv + i2] = -dpoly;
          invw += m->dof_invweight0[i2];
          vel -= dpoly * d->qvel[i2];
        } else {
          pos -= data[0];
        }
        efc_row_params(m, d, r, pos, invw, m->eq_solref[e], m->eq_solimp[e], vel, 0.0);
        r++;
        continue;
      }
      int b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
      const real* a1 = pass == ODKO_EQ_CONNECT ? data : data + 3;     /* anchor in body1's frame */
      const real* a2 = pass == ODKO_EQ_CONNECT ? data + 3 : data;     /* anchor in body2's frame */
      real p1[3], p2[3], cpos[6], t[3];
      mat_mulvec(t, d->xmat[b1], a1); for (int k = 0; k < 3; k++) p1[k] = d->xpos[b1][k] + t[k];
      mat_mulvec(t, d->xmat[b2], a2); for (int k = 0; k < 3; k++) p2[k] = d->xpos[b2][k] + t[k];
      for (int k = 0; k < 3; k++) cpos[k] = p1[k] - p2[k];
      real j1[3 * ODKO_MAXV], j2[3 * ODKO_MAXV];
      jac_point(m, d, b1, p1, j1); jac_point(m, d, b2, p2, j2);
      int nrow = 3;
      for (int k = 0; k < 3; k++)
        for (int i = 0; i < nv; i++) d->efc_J[(r + k) * nv + i] = j1[k * nv + i] - j2[k * nv + i];
      if (pass == ODKO_EQ_WELD) {
        real ts = data[10], quat[4], q1n[4], q2[4], r1[3 * ODKO_MAXV], r2[3 * ODKO_MAXV];
        quat_mul(quat, d->xquat[b1], data + 6);                          /* q(body1) * relpose */
        q1n[0] = d->xquat[b2][0]; for (int k = 1; k < 4; k++) q1n[k] = -d->xquat[b2][k];
        quat_mul(q2, q1n, quat);                                         /* conj(q(body2)) * q(body1) * relpose: identity when welded */
        for (int k = 0; k < 3; k++) cpos[3 + k] = q2[1 + k] * ts;
        jac_rot(m, d, b1, r1); jac_rot(m, d, b2, r2);
        for (int i = 0; i < nv; i++) {                                   /* d/dt of the error quaternion's axis part: 0.5 conj(q2) (0, w1 - w2) q1 relpose */
          real ax[4] = {0, r1[i] - r2[i], r1[nv + i] - r2[nv + i], r1[2 * nv + i] - r2[2 * nv + i]}, q3[4], q4[4];
          quat_mul(q3, q1n, ax); quat_mul(q4, q3, quat);
          for (int k = 0; k < 3; k++) d->efc_J[(r + 3 + k) * nv + i] = 0.5 * q4[1 + k] * ts;
        }
        nrow = 6;
      }
      real nrm = 0;
      for (int k = 0; k < nrow; k++) nrm += cpos[k] * cpos[k];
      nrm = sqrt(nrm);
      for (int k = 0; k < nrow; k++) {
        real vel = 0;
        for (int i = 0; i < nv; i++) vel += d->efc_J[(r + k) * nv + i] * d->qvel[i];
        real invw = m->body_invweight0[b1][k < 3 ? 0 : 1] + m->body_invweight0[b2][k < 3 ? 0 : 1];
        efc_row_params_imp(m, d, r + k, cpos[k], nrm, invw, m->eq_solref[e], m->eq_solimp[e], vel, 0.0);
      }
      r += nrow;
    }
  return r;
}

static void make_constraint(const odko_model* m, odko_data* d) {
  int nv = m->nv, r = 0;
  memset(d->efc_J, 0, sizeof(d->efc_J));
  r = make_equality(m, d, 0);
  d->ne = r;
  /* friction loss rows: dofs with frictionloss > 0 */
  for (int i = 0; i < nv; i++) {
    if (m->dof_frictionloss[i] <= 0) continue;
    d->efc_J[r * nv + i] = 1;
    efc_row_params(m, d, r, 0.0, m->dof_invweight0[i], m->dof_solref[i], m->dof_solimp[i], d->qvel[i], m->dof_frictionloss[i]);
    r++;
  }
  d->nf = r - d->ne;
  /* joint limit rows (hinge) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] != ODKO_JNT_HINGE) continue;
    real q = d->qpos[m->jnt_qposadr[j]];
    real dmin = q - m->jnt_range[j][0], dmax = m->jnt_range[j][1] - q;
    real pos = (dmin < dmax ? dmin : dmax) - m->jnt_margin[j];
    int active = pos < 0;
    real sgn = (dmin < dmax) ? 1.0 : -1.0;
    int i = m->jnt_dofadr[j];
    d->efc_J[r * nv + i] = sgn * active;
    efc_row_params(m, d, r, pos, m->dof_invweight0[i], m->jnt_solref[j], m->jnt_solimp[j], sgn * active * d->qvel[i], 0.0);
    r++;
  }
  d->nl = r - d->nf - d->ne;
  /* contact rows: pyramidal condim 3 -> 4 rows per contact; elliptic condim 3 -> 3 rows (normal, two tangents) */
  for (int c = 0; c < d->ncon; c++) {
    int g1 = d->contact_geom1[c], g2 = d->contact_geom2[c];
    int b1 = m->cgeom_bodyid[g1], b2 = m->cgeom_bodyid[g2];
    real j1[3 * ODKO_MAXV], j2[3 * ODKO_MAXV], dif[3 * ODKO_MAXV], con[3 * ODKO_MAXV];
    real solref[2], solimp[5];
    contact_mix(m, g1, g2, solref, solimp);
    jac_point(m, d, b1, d->contact_pos[c], j1);
    jac_point(m, d, b2, d->contact_pos[c], j2);
    for (int i = 0; i < 3 * nv; i++) dif[i] = j2[i] - j1[i];
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < nv; i++)
        con[a * nv + i] = d->contact_frame[c][3 * a] * dif[i] + d->contact_frame[c][3 * a + 1] * dif[nv + i] + d->contact_frame[c][3 * a + 2] * dif[2 * nv + i];
    real t = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    real dist = d->contact_dist[c];
    int active = dist < 0;
    real mu = d->contact_friction[c];
    d->contact_efc[c] = r;
    if (m->cone) {
      /* Elliptic cone (mjx constraint._efc_contact_elliptic / MuJoCo mj_instantiateContact + mj_makeImpedance, [UPSTREAM-MEMORY]; the cost it
       * leads to is pinned to the documented dual cone program by tests/test_oracle_elliptic.py).  Rows = the contact frame's axes; only the
       * normal row carries a position (the tangents' pos = 0), all three share the normal's impedance, stiffness and damping; the normal's
       * regulariser R_n = max(MINVAL, invweight (1 - imp) / imp) with invweight = the two bodies' translational weights, the tangents'
       * R_t = R_n / impratio; the regularised cone's mu = friction sqrt(R_t / R_n) (update_constraint). */
      real vel[3] = {0, 0, 0};
      for (int a = 0; a < 3; a++)
        for (int i = 0; i < nv; i++) {
          real v = con[a * nv + i] * active;
          d->efc_J[(r + a) * nv + i] = v;
          vel[a] += v * d->qvel[i];
        }
      efc_row_params(m, d, r, dist, t, solref, solimp, vel[0], 0.0);
      real Rt = d->efc_R[r] / (m->impratio > MINVAL ? m->impratio : MINVAL);
      d->contact_mu_reg[c] = mu * sqrt(Rt / d->efc_R[r]);
      for (int a = 1; a < 3; a++) {
        d->efc_pos[r + a] = 0; d->efc_invweight[r + a] = t; d->efc_imp[r + a] = d->efc_imp[r]; d->efc_k[r + a] = d->efc_k[r]; d->efc_b[r + a] = d->efc_b[r];
        d->efc_R[r + a] = Rt; d->efc_D[r + a] = 1.0 / Rt;
        d->efc_aref[r + a] = -d->efc_b[r] * vel[a];
        d->efc_frictionloss[r + a] = 0;
      }
      r += 3;
      continue;
    }
    for (int tdir = 1; tdir <= 2; tdir++)
      for (int s = 0; s < 2; s++) {
        real f = s == 0 ? mu : -mu, vel = 0;
        for (int i = 0; i < nv; i++) {
          real v = (con[i] + con[tdir * nv + i] * f) * active;
          d->efc_J[r * nv + i] = v;
          vel += v * d->qvel[i];
        }
        real invw = (t + f * f * t) * 2 * f * f / m->impratio;
        efc_row_params(m, d, r, dist, invw, solref, solimp, vel, 0.0);
        r++;
      }
  }
  d->nc = r - d->nf - d->nl - d->ne;
  d->nefc = r;
}

static void fwd_position(const odko_model* m, odko_data* d) {
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  cholesky(d->qL, d->qM, m->nv);
  collision(m, d);
  make_constraint(m, d);
}

/* ------------------------------------------------------------------ fwd_velocity */
static void com_vel(const odko_model* m, odko_data* d) {
  memset(d->cvel[0], 0, 6 * sizeof(real));
  for (int b = 1; b < m->nbody; b++) {
    real cvel[6];
    memcpy(cvel, d->cvel[m->body_parentid[b]], sizeof(cvel));
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == ODKO_JNT_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(d->cdof_dot[da + k], 0, 6 * sizeof(real));
          for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da + k][q] * d->qvel[da + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot[da + k], cvel, d->cdof[da + k]);
        for (int k = 3; k < 6; k++) for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da + k][q] * d->qvel[da + k];
      } else {
        cross_motion(d->cdof_dot[da], cvel, d->cdof[da]);
        for (int q = 0; q < 6; q++) cvel[q] += d->cdof[da][q] * d->qvel[da];
      }
    }
    memcpy(d->cvel[b], cvel, sizeof(cvel));
  }
}

/* rne with optional acceleration term (flg_acc: rne_postconstraint's cacc) */
static void rne_cacc(const odko_model* m, odko_data* d, int flg_acc) {
  memset(d->cacc[0], 0, 6 * sizeof(real));
  for (int k = 0; k < 3; k++) d->cacc[0][3 + k] = -m->gravity[k];
  for (int b = 1; b < m->nbody; b++) {
    memcpy(d->cacc[b], d->cacc[m->body_parentid[b]], 6 * sizeof(real));
    for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int q = 0; q < 6; q++) {
        d->cacc[b][q] += d->cdof_dot[i][q] * d->qvel[i];
        if (flg_acc) d->cacc[b][q] += d->cdof[i][q] * d->qacc[i];
      }
  }
}
static void rne(const odko_model* m, odko_data* d) {
  real cfrc[ODKO_MAXB][6];
  rne_cacc(m, d, 0);
  for (int b = 0; b < m->nbody; b++) {
    real t1[6], t2[6];
    inert_mul(cfrc[b], d->cinert[b], d->cacc[b]);
    inert_mul(t1, d->cinert[b], d->cvel[b]);
    cross_force(t2, d->cvel[b], t1);
    for (int q = 0; q < 6; q++) cfrc[b][q] += t2[q];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int q = 0; q < 6; q++) cfrc[p][q] += cfrc[b][q];
  }
  for (int i = 0; i < m->nv; i++) {
    real s = 0;
    for (int q = 0; q < 6; q++) s += d->cdof[i][q] * cfrc[m->dof_bodyid[i]][q];
    d->qfrc_bias[i] = s;
  }
}

static void fwd_velocity(const odko_model* m, odko_data* d) {
  com_vel(m, d);
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  rne(m, d);
}

/* position actuators: force = kp*ctrl + bias0 + bias1*length + bias2*velocity */
static void fwd_actuation(const odko_model* m, odko_data* d) {
  memset(d->qfrc_actuator, 0, sizeof(d->qfrc_actuator));
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_trnid[u];
    real gear = m->actuator_gear[u];
    real len = d->qpos[m->jnt_qposadr[j]] * gear, vel = d->qvel[m->jnt_dofadr[j]] * gear;
    real ctrl = d->ctrl[u];
    if (m->actuator_ctrllimited[u]) ctrl = fmin(fmax(ctrl, m->actuator_ctrlrange[u][0]), m->actuator_ctrlrange[u][1]);
    real f = m->actuator_gainprm0[u] * ctrl + m->actuator_biasprm[u][0] + m->actuator_biasprm[u][1] * len + m->actuator_biasprm[u][2] * vel;
    if (m->actuator_forcelimited[u]) f = fmin(fmax(f, m->actuator_forcerange[u][0]), m->actuator_forcerange[u][1]);
    d->actuator_force[u] = f;
    d->qfrc_actuator[m->jnt_dofadr[j]] += gear * f;
  }
}

static void fwd_acceleration(const odko_model* m, odko_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  chol_solve(d->qacc_smooth, d->qL, d->qfrc_smooth, m->nv);
}

/* ------------------------------------------------------------------ solver (mjx solver.solve, Newton) */
typedef struct {
  real qacc[ODKO_MAXV], Ma[ODKO_MAXV], Jaref[ODKO_MAXEFC], force[ODKO_MAXEFC], qfrc_constraint[ODKO_MAXV];
  int active[ODKO_MAXEFC]; /* rows in the quadratic regime (enter the Hessian) */
  real gauss, cost;
  real grad[ODKO_MAXV], Mgrad[ODKO_MAXV], search[ODKO_MAXV];
  int cone_zone[ODKO_MAXCON];   /* elliptic cones: 0 top (no force), 1 bottom (all rows quadratic), 2 middle (on the cone) */
} solver_ctx;

static void ctx_init(const odko_model* m, const odko_data* d, solver_ctx* c, const real* qacc) {
  int nv = m->nv;
  memcpy(c->qacc, qacc, nv * sizeof(real));
  mul_m(m, d, c->Ma, qacc);
  for (int r = 0; r < d->nefc; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * qacc[i];
    c->Jaref[r] = s - d->efc_aref[r];
  }
}
/* ---- elliptic cones (mjx solver / MuJoCo engine_solver.c PrimalUpdateConstraint, HessianCone, PrimalEval; [UPSTREAM-MEMORY]).
 * One contact = 3 rows (normal, two tangents) with Jaref x = (x_n, x_1, x_2), friction mu, R = (R_n, R_t, R_t), R_t = R_n / impratio.
 * In the scaled space U = (mu_r x_n, mu x_1, mu x_2), mu_r = mu sqrt(R_t / R_n), N = U_0, T = |U_1..2|:
 *   top zone     N >= mu_r T (or T = 0, N >= 0):        no force, no cost;
 *   bottom zone  mu_r N + T <= 0 (or T = 0, N < 0):     every row quadratic, f_j = -D_j x_j;
 *   middle zone  otherwise:                             cost = 0.5 Dm (N - mu_r T)^2, Dm = D_n / (mu_r^2 (1 + mu_r^2)),
 *                                                       f_n = -Dm (N - mu_r T) mu_r, f_j = -f_n mu U_j / T.
 * This IS the documented dual problem  min 0.5 f^T R f + f^T x  over the cone f_n >= 0, |f_t| <= mu f_n  in closed form: bottom = the
 * unconstrained minimiser lies in the cone, top = x in the dual cone, middle = the minimiser on the cone's boundary, value
 * 0.5 (x_n - mu |x_t|)^2 / (R_n + mu^2 R_t) (tests/test_oracle_elliptic.py derives and checks that independently). */
typedef struct { real mu, mur, N, T, U[3], Dm, NmT; int zone; } cone_pt;
static cone_pt cone_eval(const odko_data* d, int c, const real* x /* Jaref of the contact's 3 rows */) {
  cone_pt p;
  int r = d->contact_efc[c];
  p.mu = d->contact_friction[c]; p.mur = d->contact_mu_reg[c];
  p.U[0] = x[0] * p.mur; p.U[1] = x[1] * p.mu; p.U[2] = x[2] * p.mu;
  p.N = p.U[0]; p.T = sqrt(p.U[1] * p.U[1] + p.U[2] * p.U[2]);
  real m2 = p.mur * p.mur * (1 + p.mur * p.mur);
  p.Dm = d->efc_D[r] / (m2 > MINVAL ? m2 : MINVAL);
  p.NmT = p.N - p.mur * p.T;
  if (p.N >= p.mur * p.T || (p.T <= 0 && p.N >= 0)) p.zone = 0;
  else if (p.mur * p.N + p.T <= 0 || (p.T <= 0 && p.N < 0)) p.zone = 1;
  else p.zone = 2;
  return p;
}
static int cone_contact_rows(const odko_model* m, const odko_data* d, int r) {   /* is row r one of an elliptic contact's rows? */
  return m->cone && r >= d->ne + d->nf + d->nl;
}

/* solver._update_constraint */
static void update_constraint(const odko_model* m, const odko_data* d, solver_ctx* c) {
  int nv = m->nv;
  real cost = 0;
  if (m->cone)
    for (int k = 0; k < d->ncon; k++) {
      int r = d->contact_efc[k];
      cone_pt p = cone_eval(d, k, c->Jaref + r);
      c->cone_zone[k] = p.zone;
      for (int a = 0; a < 3; a++) { c->force[r + a] = 0; c->active[r + a] = p.zone != 0; }
      if (p.zone == 1) {
        for (int a = 0; a < 3; a++) { c->force[r + a] = -d->efc_D[r + a] * c->Jaref[r + a]; cost += 0.5 * d->efc_D[r + a] * c->Jaref[r + a] * c->Jaref[r + a]; }
      } else if (p.zone == 2) {
        cost += 0.5 * p.Dm * p.NmT * p.NmT;
        c->force[r] = -p.Dm * p.NmT * p.mur;
        for (int a = 1; a < 3; a++) c->force[r + a] = -c->force[r] / p.T * p.U[a] * p.mu;
      }
    }
  for (int r = 0; r < d->nefc; r++) {
    if (cone_contact_rows(m, d, r)) break;   /* (the contact rows are last) */
    real jar = c->Jaref[r];
    if (r >= d->ne && r < d->ne + d->nf) {
      real f = d->efc_frictionloss[r], rf = d->efc_R[r] * f;
      if (jar <= -rf) { c->force[r] = f; c->active[r] = 0; cost += -0.5 * rf * f - f * jar; }
      else if (jar >= rf) { c->force[r] = -f; c->active[r] = 0; cost += -0.5 * rf * f + f * jar; }
      else { c->force[r] = -d->efc_D[r] * jar; c->active[r] = 1; cost += 0.5 * d->efc_D[r] * jar * jar; }
    } else {
      int act = (r < d->ne) || (jar < 0);
      c->active[r] = act;
      c->force[r] = act ? -d->efc_D[r] * jar : 0.0;
      if (act) cost += 0.5 * d->efc_D[r] * jar * jar;
    }
  }
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int r = 0; r < d->nefc; r++) s += d->efc_J[r * nv + i] * c->force[r];
    c->qfrc_constraint[i] = s;
  }
  real g = 0;
  for (int i = 0; i < nv; i++) g += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
  c->gauss = 0.5 * g;
  c->cost = cost + c->gauss;
}
/* H = M + J^T (d^2 cost / d Jaref^2) J: D on the diagonal for the active quadratic rows; for an elliptic contact in its middle zone the
 * 3 x 3 block C = S (Dm g g^T - Dm (N - mu_r T) mu_r (I_t / T - U_t U_t^T / T^3)) S with g = (1, -mu_r U_t / T), S = diag(mu_r, mu, mu) */
static void hessian(const odko_model* m, const odko_data* d, const solver_ctx* c, real* H) {
  int nv = m->nv;
  memcpy(H, d->qM, (size_t)nv * nv * sizeof(real));
  for (int r = 0; r < d->nefc; r++) {
    if (cone_contact_rows(m, d, r)) break;
    if (!c->active[r]) continue;
    const real* J = d->efc_J + r * nv;
    for (int i = 0; i < nv; i++) {
      if (J[i] == 0) continue;
      for (int j = 0; j < nv; j++) H[i * nv + j] += d->efc_D[r] * J[i] * J[j];
    }
  }
  if (!m->cone) return;
  for (int k = 0; k < d->ncon; k++) {
    int r = d->contact_efc[k];
    real C[3][3] = {{0}};
    if (c->cone_zone[k] == 0) continue;
    if (c->cone_zone[k] == 1) { for (int a = 0; a < 3; a++) C[a][a] = d->efc_D[r + a]; }
    else {
      cone_pt p = cone_eval(d, k, c->Jaref + r);
      real g[3] = {1, -p.mur * p.U[1] / p.T, -p.mur * p.U[2] / p.T}, S[3] = {p.mur, p.mu, p.mu};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          real h = p.Dm * g[a] * g[b];
          if (a > 0 && b > 0) h -= p.Dm * p.NmT * p.mur * ((a == b ? 1.0 : 0.0) / p.T - p.U[a] * p.U[b] / (p.T * p.T * p.T));
          C[a][b] = S[a] * h * S[b];
        }
    }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        if (C[a][b] == 0) continue;
        const real* Ja = d->efc_J + (r + a) * nv; const real* Jb = d->efc_J + (r + b) * nv;
        for (int i = 0; i < nv; i++) {
          if (Ja[i] == 0) continue;
          for (int j = 0; j < nv; j++) H[i * nv + j] += C[a][b] * Ja[i] * Jb[j];
        }
      }
  }
}

/* solver._update_gradient (Newton): H = M + J^T diag(D*active) J, Cholesky, Mgrad = H^-1 grad */
static void update_gradient(const odko_model* m, const odko_data* d, solver_ctx* c) {
  int nv = m->nv;
  real H[ODKO_MAXV * ODKO_MAXV], L[ODKO_MAXV * ODKO_MAXV];
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  hessian(m, d, c, H);
  cholesky(L, H, nv);
  chol_solve(c->Mgrad, L, c->grad, nv);
}

typedef struct { real alpha, cost, deriv0, deriv1; } ls_point;

typedef struct {
  const odko_model* m;
  const odko_data* d;
  const solver_ctx* c;
  real jv[ODKO_MAXEFC], quad[ODKO_MAXEFC][3], quad_gauss[3];
} ls_ctx;

/* solver._LSPoint.create */
static ls_point ls_eval(const ls_ctx* L, real alpha) {
  const odko_data* d = L->d;
  real q0 = L->quad_gauss[0], q1 = L->quad_gauss[1], q2 = L->quad_gauss[2];
  real cone_cost = 0, cone_d0 = 0, cone_d1 = 0;   /* elliptic contacts: cost and its first / second derivative along the search, exactly */
  int nrow = d->nefc;
  if (L->m->cone) {
    nrow = d->ne + d->nf + d->nl;
    for (int k = 0; k < d->ncon; k++) {
      int r = d->contact_efc[k];
      real x[3], v[3];
      for (int a = 0; a < 3; a++) { v[a] = L->jv[r + a]; x[a] = L->c->Jaref[r + a] + alpha * v[a]; }
      cone_pt p = cone_eval(d, k, x);
      if (p.zone == 1) {
        for (int a = 0; a < 3; a++) { cone_cost += 0.5 * d->efc_D[r + a] * x[a] * x[a]; cone_d0 += d->efc_D[r + a] * x[a] * v[a]; cone_d1 += d->efc_D[r + a] * v[a] * v[a]; }
      } else if (p.zone == 2) {
        real V[3] = {v[0] * p.mur, v[1] * p.mu, v[2] * p.mu};
        real UV = p.U[1] * V[1] + p.U[2] * V[2], VV = V[1] * V[1] + V[2] * V[2];
        real T1 = UV / p.T, T2 = VV / p.T - UV * UV / (p.T * p.T * p.T);        /* dT / d alpha, d2T / d alpha2 */
        real g1 = V[0] - p.mur * T1;                                            /* d (N - mu_r T) / d alpha */
        cone_cost += 0.5 * p.Dm * p.NmT * p.NmT;
        cone_d0 += p.Dm * p.NmT * g1;
        cone_d1 += p.Dm * (g1 * g1 - p.NmT * p.mur * T2);
      }
    }
  }
  for (int r = 0; r < nrow; r++) {
    real x = L->c->Jaref[r] + alpha * L->jv[r];
    if (r >= d->ne && r < d->ne + d->nf) {
      real f = d->efc_frictionloss[r], rf = d->efc_R[r] * f;
      if (x <= -rf) { q0 += f * (-0.5 * rf - L->c->Jaref[r]); q1 += -f * L->jv[r]; }
      else if (x >= rf) { q0 += f * (-0.5 * rf + L->c->Jaref[r]); q1 += f * L->jv[r]; }
      else { q0 += L->quad[r][0]; q1 += L->quad[r][1]; q2 += L->quad[r][2]; }
    } else if (r < d->ne || x < 0) {
      q0 += L->quad[r][0]; q1 += L->quad[r][1]; q2 += L->quad[r][2];
    }
  }
  ls_point p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0 + cone_cost;
  p.deriv0 = 2 * alpha * q2 + q1 + cone_d0;
  p.deriv1 = 2 * q2 + cone_d1;
  if (p.deriv1 == 0) p.deriv1 = MINVAL;
  return p;
}
static real safe_div(real a, real b) { return b == 0 ? 0.0 : a / b; }

/* solver._linesearch */
static void linesearch(const odko_model* m, odko_data* d, solver_ctx* c) {
  int nv = m->nv;
  ls_ctx L;
  real mv[ODKO_MAXV], snorm = 0;
  L.m = m; L.d = d; L.c = c;
  for (int i = 0; i < nv; i++) snorm += c->search[i] * c->search[i];
  snorm = sqrt(snorm);
  real smag = snorm * m->meaninertia * (nv > 1 ? nv : 1);
  real gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d, mv, c->search);
  for (int r = 0; r < d->nefc; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * c->search[i];
    L.jv[r] = s;
    L.quad[r][0] = 0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r];
    L.quad[r][1] = s * c->Jaref[r] * d->efc_D[r];
    L.quad[r][2] = 0.5 * s * s * d->efc_D[r];
  }
  real sMa = 0, sq = 0, smv = 0;
  for (int i = 0; i < nv; i++) { sMa += c->search[i] * c->Ma[i]; sq += c->search[i] * d->qfrc_smooth[i]; smv += c->search[i] * mv[i]; }
  L.quad_gauss[0] = c->gauss; L.quad_gauss[1] = sMa - sq; L.quad_gauss[2] = 0.5 * smv;

  /* tie bias, solver classes (odko_set_tie_bias): 256 = which end of the final bracket is returned when the two costs are within
   * eps_rel of each other (relative to the cost at 0) -- the two ends can be far apart in alpha --, 512 = every other comparison of
   * the bracketing (relative to the larger operand): the search has a fixed budget of iterations and each comparison steers it */
  const int ls_on = g_bias_pass - 1 >= g_bias_first && g_bias_pass - 1 <= g_bias_last;
#define LS_LT(a, b) (((g_bias_request & 512) && ls_on && fabs((a) - (b)) < g_bias_eps_rel * (fabs(a) > fabs(b) ? fabs(a) : fabs(b))) ? !((a) < (b)) : ((a) < (b)))
  ls_point p0 = ls_eval(&L, 0.0);
  ls_point lo_in = ls_eval(&L, -safe_div(p0.deriv0, p0.deriv1));
  int lo_less = LS_LT(lo_in.deriv0, p0.deriv0);
  ls_point lo = lo_less ? lo_in : p0, hi = lo_less ? p0 : lo_in;
  int swap = 1, it = 0;
  const real zero = 0;
  while (1) {
    int done = it >= m->ls_iterations;
    done |= !swap;
    done |= LS_LT(lo.deriv0, zero) && LS_LT(-gtol, lo.deriv0);
    done |= LS_LT(zero, hi.deriv0) && LS_LT(hi.deriv0, gtol);
    if (done) break;
    ls_point lo_next = ls_eval(&L, lo.alpha - safe_div(lo.deriv0, lo.deriv1));
    ls_point hi_next = ls_eval(&L, hi.alpha - safe_div(hi.deriv0, hi.deriv1));
    ls_point mid = ls_eval(&L, 0.5 * (lo.alpha + hi.alpha));
    int s1 = LS_LT(zero, lo.deriv0) || LS_LT(lo.deriv0, lo_next.deriv0);
    if (s1) lo = lo_next;
    int s2 = LS_LT(mid.deriv0, zero) && LS_LT(lo.deriv0, mid.deriv0);
    if (s2) lo = mid;
    int s3 = LS_LT(hi_next.deriv0, zero) && LS_LT(lo.deriv0, hi_next.deriv0);
    if (s3) lo = hi_next;
    int s4 = LS_LT(hi.deriv0, zero) || LS_LT(hi_next.deriv0, hi.deriv0);
    if (s4) hi = hi_next;
    int s5 = LS_LT(zero, mid.deriv0) && LS_LT(mid.deriv0, hi.deriv0);
    if (s5) hi = mid;
    int s6 = LS_LT(zero, lo_next.deriv0) && LS_LT(lo_next.deriv0, hi.deriv0);
    if (s6) hi = lo_next;
    swap = s1 | s2 | s3 | s4 | s5 | s6;
    it++;
  }
#undef LS_LT
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  int take_lo = lo.cost < hi.cost;
  if ((g_bias_request & 256) && ls_on && fabs(lo.cost - hi.cost) < g_bias_eps_rel * (fabs(p0.cost) > 1e-9 ? fabs(p0.cost) : 1e-9)) take_lo = !take_lo;
  real alpha = take_lo ? lo.alpha : hi.alpha;
  if (!improved) alpha = 0;
  d->ls_alpha = alpha; d->ls_iters = it;
  for (int i = 0; i < nv; i++) { c->qacc[i] += alpha * c->search[i]; c->Ma[i] += alpha * mv[i]; }
  for (int r = 0; r < d->nefc; r++) c->Jaref[r] += alpha * L.jv[r];
}

static void solve(const odko_model* m, odko_data* d) {
  int nv = m->nv;
  solver_ctx cw, cs, *c;
  /* warmstart: lower cost of qacc_warmstart / qacc_smooth */
  ctx_init(m, d, &cw, d->qacc_warmstart); update_constraint(m, d, &cw);
  ctx_init(m, d, &cs, d->qacc_smooth); update_constraint(m, d, &cs);
  d->warm_used = cw.cost < cs.cost;
  { real g = fabs(cw.cost - cs.cost) / (fabs(cs.cost) > 1e-9 ? fabs(cs.cost) : 1e-9); if (g < d->decision_margin[4]) d->decision_margin[4] = g;   /* warm-start pick */
    if ((g_bias_request & 128) && g_bias_pass - 1 >= g_bias_first && g_bias_pass - 1 <= g_bias_last && g < g_bias_eps_rel) d->warm_used = !d->warm_used; }   /* tie bias, class 128 */
  c = d->warm_used ? &cw : &cs;
  d->solver_cost0 = c->cost;
  update_gradient(m, d, c);
  for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i];
  /* iterations: the model sets iterations=1 -> body() exactly once */
  for (int it = 0; it < m->iterations; it++) {
    linesearch(m, d, c);
    update_constraint(m, d, c);
    if (it + 1 < m->iterations) {
      real prev_cost = d->solver_cost0;
      update_gradient(m, d, c);
      for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i];
      /* termination tests of mjx solver.cond (only reached when iterations > 1) */
      real scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1)), gn = 0;
      for (int i = 0; i < nv; i++) gn += c->grad[i] * c->grad[i];
      if (scale * (prev_cost - c->cost) < m->tolerance || scale * sqrt(gn) < m->tolerance) break;
      d->solver_cost0 = c->cost;
    }
  }
  d->solver_cost1 = c->cost;
  memcpy(d->qacc, c->qacc, nv * sizeof(real));
  memcpy(d->qacc_warmstart, c->qacc, nv * sizeof(real));
  memcpy(d->qfrc_constraint, c->qfrc_constraint, nv * sizeof(real));
  memcpy(d->efc_force, c->force, d->nefc * sizeof(real));
}

/* tests: cost, gradient and Newton Hessian of the solver's objective at an arbitrary qacc (rows of the last odko_forward) */
void odko_solver_probe(const odko_model* m, const odko_data* d, const real* qacc, real* cost, real* grad, real* hess) {
  solver_ctx c;
  ctx_init(m, d, &c, qacc);
  update_constraint(m, d, &c);
  for (int i = 0; i < m->nv; i++) c.grad[i] = c.Ma[i] - d->qfrc_smooth[i] - c.qfrc_constraint[i];
  if (cost) *cost = c.cost;
  if (grad) memcpy(grad, c.grad, m->nv * sizeof(real));
  if (hess) hessian(m, d, &c, hess);
}

/* ------------------------------------------------------------------ sensors (mjx sensor.py) */
static void sensors(const odko_model* m, odko_data* d) {
  rne_cacc(m, d, 1); /* rne_postconstraint's cacc (uses the solved qacc) */
  for (int s = 0; s < m->nsensor; s++) {
    int site = m->sensor_objid[s], b = m->site_bodyid[s >= 0 ? site : 0];
    real* out = d->sensordata + m->sensor_adr[s];
    const real* R = d->site_xmat[site];
    real dif[3], vang[3], vlin[3], t[3];
    v3_sub(dif, d->site_xpos[site], d->subtree_com[m->body_rootid[b]]);
    v3_copy(vang, d->cvel[b]);
    v3_cross(t, dif, d->cvel[b]);          /* lin at site = lin_ref - dif x ang */
    v3_sub(vlin, d->cvel[b] + 3, t);
    switch (m->sensor_type[s]) {
      case ODKO_S_GYRO: mat_tmulvec(out, R, vang); break;
      case ODKO_S_VELOCIMETER: mat_tmulvec(out, R, vlin); break;
      case ODKO_S_ACCELEROMETER: {
        real al[3], wl[3], vl[3], corr[3], acc[3];
        v3_cross(t, dif, d->cacc[b]);
        v3_sub(al, d->cacc[b] + 3, t);
        mat_tmulvec(acc, R, al);
        mat_tmulvec(wl, R, vang);
        mat_tmulvec(vl, R, vlin);
        v3_cross(corr, wl, vl);
        v3_addscl(out, acc, corr, 1);
        break;
      }
      case ODKO_S_FRAMEZAXIS: out[0] = R[2]; out[1] = R[5]; out[2] = R[8]; break;
      case ODKO_S_FRAMEXAXIS: out[0] = R[0]; out[1] = R[3]; out[2] = R[6]; break;
      case ODKO_S_FRAMELINVEL: v3_copy(out, vlin); break;
      case ODKO_S_FRAMEANGVEL: v3_copy(out, vang); break;
      case ODKO_S_FRAMEPOS: v3_copy(out, d->site_xpos[site]); break;
      case ODKO_S_FRAMEQUAT: {
        real q[4];
        quat_mul(q, d->xquat[b], m->site_quat[site]);
        quat_normalize(q);
        memcpy(out, q, sizeof(q));
        break;
      }
      default: break;
    }
  }
}

void odko_forward(const odko_model* m, odko_data* d) {
  fwd_position(m, d);
  fwd_velocity(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  if (d->nefc == 0) memcpy(d->qacc, d->qacc_smooth, m->nv * sizeof(real));
  else solve(m, d);
  sensors(m, d);
}

/* mjx forward.euler with eulerdamp disabled (open_duck_mini_v2.xml:6-8) */
static void euler(const odko_model* m, odko_data* d) {
  real dt = m->timestep;
  if (m->eulerdamp) {
    /* implicit-in-damping velocity update: (M + dt*diag(damping)) qacc' = qfrc_smooth + qfrc_constraint */
    int nv = m->nv;
    real H[ODKO_MAXV * ODKO_MAXV], L[ODKO_MAXV * ODKO_MAXV], rhs[ODKO_MAXV];
    memcpy(H, d->qM, (size_t)nv * nv * sizeof(real));
    for (int i = 0; i < nv; i++) { H[i * nv + i] += dt * m->dof_damping[i]; rhs[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    cholesky(L, H, nv);
    chol_solve(d->qacc, L, rhs, nv);
  }
  for (int i = 0; i < m->nv; i++) d->qvel[i] += dt * d->qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == ODKO_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      real w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real n = sqrt(v3_dot(w, w)), q[4], r[4];
      if (n < MINVAL) { w[0] = 1; w[1] = 0; w[2] = 0; n = 0; } else { w[0] /= n; w[1] /= n; w[2] /= n; } /* math.normalize_with_norm */
      real ang = dt * n, s = sin(0.5 * ang);
      q[0] = cos(0.5 * ang); q[1] = s * w[0]; q[2] = s * w[1]; q[3] = s * w[2];
      quat_mul(r, d->qpos + qa + 3, q);
      quat_normalize(r);
      memcpy(d->qpos + qa + 3, r, sizeof(r));
    } else {
      d->qpos[qa] += dt * d->qvel[da];
    }
  }
  d->time += dt;
}

void odko_step(const odko_model* m, odko_data* d) {
  odko_forward(m, d);
  euler(m, d);
}

void odko_env_physics_step(const odko_model* m, odko_data* d, const real* ctrl, int n_substeps) {
  for (int s = 0; s < n_substeps; s++) {
    for (int u = 0; u < m->nu; u++) d->ctrl[u] = ctrl[u];
    odko_step(m, d);
  }
}
