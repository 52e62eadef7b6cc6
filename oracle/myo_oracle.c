/* myo_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the physics that the reference reaches through third-party
 * MuJoCo 3.2.8 (README.md:35-41) at
 *     myosuite/physics/mj_sim_scene.py:55      (dm_control Physics.step -> mj_step)
 *     myosuite/physics/mjpy_sim_scene.py:177-179 (mj_step2; mj_step1)
 *     myosuite/mjx/play.py:41,47                 (mjx.step)
 * for the feature subset the MyoSuite config models use (SURVEY.md section 8a rows A3-A12,
 * Appendix B).  MuJoCo's source is NOT under /root/reference and not installed here, so
 * this file restates its published algorithms (MuJoCo documentation, "Computation" and
 * "Modeling" chapters) [3P].  PARITY PIN: the kinematics + tendon-wrapping part is pinned
 * against the MuJoCo-computed `lengthrange` values stored in the reference's own model file
 * (simhive/myo_sim/hand/assets/myohand_assets.xml:501-539; tests/test_oracle_golden.py).
 * The dynamics part has no golden vectors in the reference: "parity unpinned"
 * beyond physical invariants (energy, finite-difference Jacobians, M symmetry/positivity).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile (double: libmyo_oracle.so; float: libmyo_oracle_f32.so).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(MYOO_COUNT_FLOPS)      /* instrumented C++ build (make flops): `real` counts its own arithmetic, see flopcount.h */
#include "flopcount.h"
thread_local FlopCounters g_flops;
extern "C" {
#elif defined(MYOO_FLOAT)
typedef float real;
#else
typedef double real;
#endif

#define MINVAL ((real)1e-15)
#define MAXVAL ((real)1e10)
#define MINIMP ((real)0.0001)
#define MAXIMP ((real)0.9999)
#define NCON_MAX 128
#define GEOM_PRISM 100   /* internal shape of the height-field narrow phase: a triangular prism given by 6 vertices */

enum { GEOM_PLANE = 0, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH };
enum { JNT_FREE = 0, JNT_BALL, JNT_SLIDE, JNT_HINGE };
enum { WRAP_NONE = 0, WRAP_JOINT, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER };
enum { CT_EQUALITY = 0, CT_LIMIT_JOINT, CT_LIMIT_TENDON, CT_CONTACT, CT_FRICTION_DOF };

/* ------------------------------------------------------------------ model */
typedef struct {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, npair, nM;
  real timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations;
  int integrator;   /* 0: semi-implicit Euler with implicit joint damping (mj_Euler); 1: 4th-order Runge-Kutta (mj_RungeKutta) */
  int *body_parentid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_weldid, *body_rootid;
  real *body_pos, *body_quat, *body_mass, *body_ipos, *body_iquat, *body_inertia, *body_invweight0;
  real *body_subtreemass;
  int *body_lastdof;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  real *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp, *jnt_stiffness;
  real *qpos0, *qpos_spring;
  int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  real *dof_armature, *dof_damping, *dof_invweight0;
  real *dof_frictionloss, *dof_solref_fri, *dof_solimp_fri;   /* joint friction loss (mj_instantiateFriction rows); NULL-safe: zero when the blob lacks them */
  int *geom_meshadr, *geom_meshnum; real* mesh_vert;          /* convex-hull vertices of colliding mesh geoms (geom frame) */
  int *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority;
  real *geom_pos, *geom_quat, *geom_size, *geom_margin, *geom_gap, *geom_solmix, *geom_friction, *geom_solref,
      *geom_solimp, *geom_rbound;
  int *site_bodyid;
  real *site_pos;
  int *wrap_type, *wrap_objid;
  real *wrap_prm;
  int *tendon_adr, *tendon_num, *tendon_limited;
  real *tendon_range, *tendon_margin, *tendon_stiffness, *tendon_damping, *tendon_solref, *tendon_solimp,
      *tendon_invweight0;
  int *actuator_trnid, *actuator_trntype, *actuator_ctrllimited, *actuator_forcelimited, *actuator_kind;
  real* hfield_size;   /* x, y half-extents, z scale, base depth */
  int* hfield_dims;    /* nrow, ncol, geom id of the colliding height field (-1: none) */
  real *actuator_gear, *actuator_dynprm, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange,
      *actuator_forcerange, *actuator_lengthrange, *actuator_acc0;
  int *pair_geom, *pair_condim;
  int neq, *eq_obj1id, *eq_obj2id;
  real *eq_data, *eq_solref, *eq_solimp;
  int disable_contact, disable_limit, disable_ellipsoid; /* test switches */
  int disable_passive, disable_gravity, disable_actuation; /* mjDSBL_PASSIVE / GRAVITY / ACTUATION, used by the length-range simulation */
  void* storage[256];
  int nstorage;
} Model;

typedef struct {
  real dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2;
} Contact;

typedef struct {
  /* state */
  real *qpos, *qvel, *act, *ctrl, *qacc_warmstart, time;
  /* position stage */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos;
  real *subtree_com, *cdof, *cinert, *crb, *ten_length, *ten_J, *qM, *qLD, *qLDiagInv;
  real *actuator_length, *actuator_moment;
  Contact con[NCON_MAX];
  int ncon, ncon_dropped;
  float* hfield_data;   /* [nrow * ncol] elevation in [0, 1]-ish units of hfield_size[2], row-major (mjModel.hfield_data: per instance here, the
                           terrain envs rewrite it per episode) */
  /* velocity stage */
  real *cvel, *cdof_dot, *ten_velocity, *actuator_velocity, *qfrc_bias, *qfrc_passive;
  /* acceleration stage */
  real *actuator_force, *act_dot, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc, *qfrc_applied;
  /* constraints */
  int nefc, nefc_max;
  int *efc_type, *efc_id, *efc_state;
  real *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_aref, *efc_vel, *efc_force, *efc_b, *efc_frictionloss;
  /* solver diagnostics */
  int solver_iter, warning;
  real solver_improvement, solver_gradient;
  /* scratch */
  real *wk; /* >= 8*nv + nv*nv*2 + 8*nefc_max */
} Data;

/* ------------------------------------------------------------------ blob loading */
typedef struct {
  char name[32];
  uint32_t dtype, ndim, shape[4];
  uint64_t nbytes, offset;
} BlobRec;

static const BlobRec* blob_find(const uint8_t* blob, const char* name) {
  uint32_t n;
  memcpy(&n, blob + 8, 4);
  for (uint32_t i = 0; i < n; i++) {
    const BlobRec* r = (const BlobRec*)(blob + 16 + (size_t)i * sizeof(BlobRec));
    if (!strncmp(r->name, name, 32)) return r;
  }
  return NULL;
}

static real* load_f(Model* m, const uint8_t* blob, const char* name) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 0) { fprintf(stderr, "myo_oracle: missing f64 array %s\n", name); return NULL; }
  size_t n = r->nbytes / 8;
  real* out = (real*)malloc((n + 1) * sizeof(real));
  const double* src = (const double*)(blob + r->offset);
  for (size_t i = 0; i < n; i++) out[i] = (real)src[i];
  m->storage[m->nstorage++] = out;
  return out;
}

static int* load_i(Model* m, const uint8_t* blob, const char* name) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 1) { fprintf(stderr, "myo_oracle: missing i32 array %s\n", name); return NULL; }
  size_t n = r->nbytes / 4;
  int* out = (int*)malloc((n + 1) * sizeof(int));
  memcpy(out, blob + r->offset, n * 4);
  m->storage[m->nstorage++] = out;
  return out;
}

Model* myoo_load(const void* blobv, size_t nbytes) {
  const uint8_t* blob = (const uint8_t*)blobv;
  uint32_t ver;
  if (nbytes < 16 || memcmp(blob, "MYOB", 4)) return NULL;
  memcpy(&ver, blob + 4, 4);
  if (ver != 3) return NULL;
  Model* m = (Model*)calloc(1, sizeof(Model));
  int* sz = load_i(m, blob, "sizes");
  m->nq = sz[0]; m->nv = sz[1]; m->nu = sz[2]; m->na = sz[3]; m->nbody = sz[4]; m->njnt = sz[5];
  m->ngeom = sz[6]; m->nsite = sz[7]; m->ntendon = sz[8]; m->nwrap = sz[9]; m->npair = sz[10]; m->nM = sz[11];
  real* opt = load_f(m, blob, "opt");
  m->timestep = opt[0]; m->gravity[0] = opt[1]; m->gravity[1] = opt[2]; m->gravity[2] = opt[3];
  m->tolerance = opt[4]; m->iterations = (int)opt[5]; m->ls_iterations = (int)opt[6]; m->ls_tolerance = opt[7];
  m->impratio = opt[8]; m->meaninertia = opt[9];
#define LF(x) m->x = load_f(m, blob, #x)
#define LI(x) m->x = load_i(m, blob, #x)
  LI(body_parentid); LI(body_jntadr); LI(body_jntnum); LI(body_dofadr); LI(body_dofnum); LI(body_weldid);
  LI(body_rootid); LF(body_pos); LF(body_quat); LF(body_mass); LF(body_ipos); LF(body_iquat); LF(body_inertia);
  LF(body_invweight0);
  LI(jnt_type); LI(jnt_qposadr); LI(jnt_dofadr); LI(jnt_bodyid); LI(jnt_limited); LF(jnt_pos); LF(jnt_axis);
  LF(jnt_range); LF(jnt_margin); LF(jnt_solref); LF(jnt_solimp); LF(jnt_stiffness); LF(qpos0); LF(qpos_spring);
  LI(dof_bodyid); LI(dof_jntid); LI(dof_parentid); LI(dof_Madr); LF(dof_armature); LF(dof_damping);
  LF(dof_invweight0);
  LI(geom_type); LI(geom_bodyid); LI(geom_contype); LI(geom_conaffinity); LI(geom_condim); LI(geom_priority);
  LF(geom_pos); LF(geom_quat); LF(geom_size); LF(geom_margin); LF(geom_gap); LF(geom_solmix); LF(geom_friction);
  LF(geom_solref); LF(geom_solimp); LF(geom_rbound);
  LI(site_bodyid); LF(site_pos); LI(wrap_type); LI(wrap_objid); LF(wrap_prm);
  LI(tendon_adr); LI(tendon_num); LI(tendon_limited); LF(tendon_range); LF(tendon_margin); LF(tendon_stiffness);
  LF(tendon_damping); LF(tendon_solref); LF(tendon_solimp); LF(tendon_invweight0);
  LI(actuator_trnid); LI(actuator_trntype); LI(actuator_ctrllimited); LI(actuator_forcelimited); LI(actuator_kind);
  LF(hfield_size); LI(hfield_dims);
  LF(actuator_gear); LF(actuator_dynprm); LF(actuator_gainprm); LF(actuator_biasprm); LF(actuator_ctrlrange);
  LF(actuator_forcerange); LF(actuator_lengthrange); LF(actuator_acc0);
  LI(pair_geom); LI(pair_condim);
  /* optional tables (models compiled before they existed simply lack them) */
  if (blob_find(blob, "dof_frictionloss")) { LF(dof_frictionloss); LF(dof_solref_fri); LF(dof_solimp_fri); }
  else { m->dof_frictionloss = (real*)calloc(m->nv + 1, sizeof(real)); m->storage[m->nstorage++] = m->dof_frictionloss; }
  if (blob_find(blob, "geom_meshadr")) { LI(geom_meshadr); LI(geom_meshnum); LF(mesh_vert); }
  if (blob_find(blob, "integrator")) { int* ig = load_i(m, blob, "integrator"); m->integrator = ig[0]; }
  m->neq = sz[12];
  LI(eq_obj1id); LI(eq_obj2id); LF(eq_data); LF(eq_solref); LF(eq_solimp);
#undef LF
#undef LI
  /* derived: subtree mass, last dof on the chain of each body */
  m->body_subtreemass = (real*)calloc(m->nbody, sizeof(real));
  m->storage[m->nstorage++] = m->body_subtreemass;
  for (int i = 0; i < m->nbody; i++) m->body_subtreemass[i] = m->body_mass[i];
  for (int i = m->nbody - 1; i > 0; i--) m->body_subtreemass[m->body_parentid[i]] += m->body_subtreemass[i];
  m->body_lastdof = (int*)calloc(m->nbody, sizeof(int));
  m->storage[m->nstorage++] = m->body_lastdof;
  m->body_lastdof[0] = -1;
  for (int i = 1; i < m->nbody; i++)
    m->body_lastdof[i] = m->body_dofnum[i] ? m->body_dofadr[i] + m->body_dofnum[i] - 1
                                           : m->body_lastdof[m->body_parentid[i]];
  return m;
}

void myoo_free(Model* m) {
  if (!m) return;
  for (int i = 0; i < m->nstorage; i++) free(m->storage[i]);
  free(m);
}

void myoo_set_switch(Model* m, int disable_contact, int disable_limit, int disable_ellipsoid) {
  m->disable_contact = disable_contact; m->disable_limit = disable_limit; m->disable_ellipsoid = disable_ellipsoid;
}

int myoo_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ small math */
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline real norm3(const real* a) { return sqrt(dot3(a, a)); }
static inline real normalize3(real* a) {
  real n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static inline real clipr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline real maxr(real a, real b) { return a > b ? a : b; }
static inline real minr(real a, real b) { return a < b ? a : b; }
static void mul_quat(real* r, const real* a, const real* b) {
  real t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
               a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2mat(real* R, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
static void rot_vec_quat(real* r, const real* v, const real* q) {
  real R[9];
  quat2mat(R, q);
  real t[3] = {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
               R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
static void normalize4(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { for (int i = 0; i < 4; i++) q[i] /= n; }
}
static void mat_vec3(real* r, const real* R, const real* v) {
  real t[3] = {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
               R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
static void matT_vec3(real* r, const real* R, const real* v) {
  real t[3] = {R[0] * v[0] + R[3] * v[1] + R[6] * v[2], R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
               R[2] * v[0] + R[5] * v[1] + R[8] * v[2]};
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
static void mat_mul3(real* C, const real* A, const real* B) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}

/* ------------------------------------------------------------------ data */
Data* myoo_make_data(const Model* m) {
  Data* d = (Data*)calloc(1, sizeof(Data));
  int nv = m->nv, nb = m->nbody;
#define AL(x, n) d->x = (real*)calloc((size_t)(n) + 1, sizeof(real))
  AL(qpos, m->nq); AL(qvel, nv); AL(act, m->na); AL(ctrl, m->nu); AL(qacc_warmstart, nv);
  AL(xpos, 3 * nb); AL(xquat, 4 * nb); AL(xmat, 9 * nb); AL(xipos, 3 * nb); AL(ximat, 9 * nb);
  AL(xanchor, 3 * m->njnt); AL(xaxis, 3 * m->njnt); AL(geom_xpos, 3 * m->ngeom); AL(geom_xmat, 9 * m->ngeom);
  AL(site_xpos, 3 * m->nsite); AL(subtree_com, 3 * nb); AL(cdof, 6 * nv); AL(cinert, 10 * nb); AL(crb, 10 * nb);
  AL(ten_length, m->ntendon); AL(ten_J, m->ntendon * nv); AL(qM, m->nM); AL(qLD, m->nM); AL(qLDiagInv, nv);
  AL(actuator_length, m->nu); AL(actuator_moment, m->nu * nv);
  AL(cvel, 6 * nb); AL(cdof_dot, 6 * nv); AL(ten_velocity, m->ntendon); AL(actuator_velocity, m->nu);
  AL(qfrc_bias, nv); AL(qfrc_passive, nv);
  AL(actuator_force, m->nu); AL(act_dot, m->na); AL(qfrc_actuator, nv); AL(qfrc_smooth, nv); AL(qacc_smooth, nv);
  AL(qfrc_constraint, nv); AL(qacc, nv); AL(qfrc_applied, nv);
  d->nefc_max = m->neq + m->nv + 2 * m->njnt + 2 * m->ntendon + 10 * NCON_MAX;
  int ne = d->nefc_max;
  d->efc_type = (int*)calloc(ne, sizeof(int)); d->efc_id = (int*)calloc(ne, sizeof(int));
  d->efc_state = (int*)calloc(ne, sizeof(int));
  AL(efc_J, ne * nv); AL(efc_pos, ne); AL(efc_margin, ne); AL(efc_diagApprox, ne); AL(efc_R, ne); AL(efc_D, ne);
  AL(efc_aref, ne); AL(efc_vel, ne); AL(efc_force, ne); AL(efc_b, ne); AL(efc_frictionloss, ne);
  AL(wk, 32 * nv + 2 * nv * nv + 8 * ne + 12 * nb + 64);
#undef AL
  memcpy(d->qpos, m->qpos0, m->nq * sizeof(real));
  d->hfield_data = (float*)calloc((size_t)m->hfield_dims[0] * m->hfield_dims[1] + 1, sizeof(float));
  return d;
}

/* model edit between episodes, as ObjHoldRandomEnvV0.reset does to model.geom_size (obj_hold_v0.py:133-139): size and bounding radius of one
 * primitive geom; mass and inertia stay (the reference does not recompile either) */
void myoo_set_geom_size(Model* m, int g, const double* size) {
  for (int k = 0; k < 3; k++) m->geom_size[3 * g + k] = (real)size[k];
  real a = (real)size[0], b = (real)size[1], c = (real)size[2], rb = a;
  switch (m->geom_type[g]) {
    case GEOM_CAPSULE: rb = a + b; break;
    case GEOM_CYLINDER: rb = sqrt(a * a + b * b); break;
    case GEOM_ELLIPSOID: rb = maxr(a, maxr(b, c)); break;
    default: break;
  }
  m->geom_rbound[g] = rb;
}

void myoo_set_hfield(const Model* m, Data* d, const float* data) {
  memcpy(d->hfield_data, data, (size_t)m->hfield_dims[0] * m->hfield_dims[1] * sizeof(float));
}

void myoo_free_data(Data* d) {
  if (!d) return;
  free(d->hfield_data);
  real** f = (real**)&d->qpos;
  (void)f;
  free(d->qpos); free(d->qvel); free(d->act); free(d->ctrl); free(d->qacc_warmstart); free(d->xpos); free(d->xquat);
  free(d->xmat); free(d->xipos); free(d->ximat); free(d->xanchor); free(d->xaxis); free(d->geom_xpos);
  free(d->geom_xmat); free(d->site_xpos); free(d->subtree_com); free(d->cdof); free(d->cinert); free(d->crb);
  free(d->ten_length); free(d->ten_J); free(d->qM); free(d->qLD); free(d->qLDiagInv); free(d->actuator_length);
  free(d->actuator_moment); free(d->cvel); free(d->cdof_dot); free(d->ten_velocity); free(d->actuator_velocity);
  free(d->qfrc_bias); free(d->qfrc_passive); free(d->actuator_force); free(d->act_dot); free(d->qfrc_actuator);
  free(d->qfrc_smooth); free(d->qacc_smooth); free(d->qfrc_constraint); free(d->qacc); free(d->qfrc_applied); free(d->efc_type);
  free(d->efc_id); free(d->efc_state); free(d->efc_J); free(d->efc_pos); free(d->efc_margin);
  free(d->efc_diagApprox); free(d->efc_R); free(d->efc_D); free(d->efc_aref); free(d->efc_vel); free(d->efc_force);
  free(d->efc_b); free(d->efc_frictionloss); free(d->wk);
  free(d);
}

void myoo_reset(const Model* m, Data* d) { /* mj_resetData: robot.py:976 -> sim_scene.py:115-118 */
  memcpy(d->qpos, m->qpos0, m->nq * sizeof(real));
  memset(d->qvel, 0, m->nv * sizeof(real));
  memset(d->act, 0, m->na * sizeof(real));
  memset(d->ctrl, 0, m->nu * sizeof(real));
  memset(d->qacc_warmstart, 0, m->nv * sizeof(real));
  memset(d->qfrc_applied, 0, m->nv * sizeof(real));
  d->time = 0;
}

/* ------------------------------------------------------------------ position stage */
static void kinematics(const Model* m, Data* d) { /* mj_kinematics [3P] */
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  d->xipos[0] = d->xipos[1] = d->xipos[2] = 0;
  quat2mat(d->ximat, d->xquat);
  for (int i = 1; i < m->nbody; i++) {
    int pid = m->body_parentid[i], jadr = m->body_jntadr[i], jnum = m->body_jntnum[i];
    real xpos[3], xquat[4];
    if (jnum == 1 && m->jnt_type[jadr] == JNT_FREE) {
      int qa = m->jnt_qposadr[jadr];
      memcpy(xpos, d->qpos + qa, 3 * sizeof(real));
      memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(real));
      normalize4(xquat);
      memcpy(d->xanchor + 3 * jadr, xpos, 3 * sizeof(real));
      d->xaxis[3 * jadr] = 0; d->xaxis[3 * jadr + 1] = 0; d->xaxis[3 * jadr + 2] = 1;
    } else {
      mat_vec3(xpos, d->xmat + 9 * pid, m->body_pos + 3 * i);
      for (int k = 0; k < 3; k++) xpos[k] += d->xpos[3 * pid + k];
      mul_quat(xquat, d->xquat + 4 * pid, m->body_quat + 4 * i);
      for (int j = jadr; j < jadr + jnum; j++) {
        int qa = m->jnt_qposadr[j];
        real* xanchor = d->xanchor + 3 * j;
        real* xaxis = d->xaxis + 3 * j;
        rot_vec_quat(xaxis, m->jnt_axis + 3 * j, xquat);
        rot_vec_quat(xanchor, m->jnt_pos + 3 * j, xquat);
        for (int k = 0; k < 3; k++) xanchor[k] += xpos[k];
        real q = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == JNT_SLIDE) {
          for (int k = 0; k < 3; k++) xpos[k] += xaxis[k] * q;
        } else if (m->jnt_type[j] == JNT_HINGE) {
          real s = sin(q * (real)0.5), c = cos(q * (real)0.5);
          real qloc[4] = {c, m->jnt_axis[3 * j] * s, m->jnt_axis[3 * j + 1] * s, m->jnt_axis[3 * j + 2] * s};
          mul_quat(xquat, xquat, qloc);
          real vec[3];
          rot_vec_quat(vec, m->jnt_pos + 3 * j, xquat);
          for (int k = 0; k < 3; k++) xpos[k] = xanchor[k] - vec[k];
        }
      }
    }
    normalize4(xquat);
    memcpy(d->xpos + 3 * i, xpos, sizeof xpos);
    memcpy(d->xquat + 4 * i, xquat, sizeof xquat);
    quat2mat(d->xmat + 9 * i, xquat);
    real v[3], iq[4];
    mat_vec3(v, d->xmat + 9 * i, m->body_ipos + 3 * i);
    for (int k = 0; k < 3; k++) d->xipos[3 * i + k] = xpos[k] + v[k];
    mul_quat(iq, xquat, m->body_iquat + 4 * i);
    quat2mat(d->ximat + 9 * i, iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real v[3], R[9];
    mat_vec3(v, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + v[k];
    quat2mat(R, m->geom_quat + 4 * g);
    mat_mul3(d->geom_xmat + 9 * g, d->xmat + 9 * b, R);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    real v[3];
    mat_vec3(v, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + v[k];
  }
}

/* 10-element spatial inertia about the reference point: (Ixx Iyy Izz Ixy Ixz Iyz, m*dif, m) */
static void inert_com(real* res, const real* inert, const real* mat, const real* dif, real mass) {
  real t[9], I[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = mat[3 * i + j] * inert[j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) I[3 * i + j] = t[3 * i] * mat[3 * j] + t[3 * i + 1] * mat[3 * j + 1] + t[3 * i + 2] * mat[3 * j + 2];
  res[0] = I[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = I[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = I[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = I[1] - mass * dif[0] * dif[1];
  res[4] = I[2] - mass * dif[0] * dif[2];
  res[5] = I[5] - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
static void mul_inert_vec(real* res, const real* i, const real* v) {
  real r[6];
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
  memcpy(res, r, sizeof r);
}

static void com_pos(const Model* m, Data* d) { /* mj_comPos [3P] */
  int nb = m->nbody;
  for (int i = 0; i < nb; i++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] = m->body_mass[i] * d->xipos[3 * i + k];
  for (int i = nb - 1; i > 0; i--)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * m->body_parentid[i] + k] += d->subtree_com[3 * i + k];
  for (int i = 0; i < nb; i++) {
    if (m->body_subtreemass[i] < MINVAL) memcpy(d->subtree_com + 3 * i, d->xipos + 3 * i, 3 * sizeof(real));
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] /= m->body_subtreemass[i];
  }
  memset(d->cinert, 0, 10 * sizeof(real));
  for (int i = 1; i < nb; i++) {
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * i + k] - d->subtree_com[3 * m->body_rootid[i] + k];
    inert_com(d->cinert + 10 * i, m->body_inertia + 3 * i, d->ximat + 9 * i, off, m->body_mass[i]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[3 * m->body_rootid[b] + k] - d->xanchor[3 * j + k];
    real* c = d->cdof + 6 * da;
    if (m->jnt_type[j] == JNT_HINGE) {
      memcpy(c, d->xaxis + 3 * j, 3 * sizeof(real));
      cross3(c + 3, d->xaxis + 3 * j, off);
    } else if (m->jnt_type[j] == JNT_SLIDE) {
      c[0] = c[1] = c[2] = 0;
      memcpy(c + 3, d->xaxis + 3 * j, 3 * sizeof(real));
    } else if (m->jnt_type[j] == JNT_FREE) {
      memset(c, 0, 36 * sizeof(real));
      c[3] = 1; c[6 + 4] = 1; c[12 + 5] = 1;
      for (int r = 0; r < 3; r++) {
        real ax[3] = {d->xmat[9 * b + r], d->xmat[9 * b + 3 + r], d->xmat[9 * b + 6 + r]};
        memcpy(c + 18 + 6 * r, ax, sizeof ax);
        cross3(c + 18 + 6 * r + 3, ax, off);
      }
    }
  }
}

/* translational Jacobian (3 x nv, row-major) of a world point attached to body */
static void jac_point(const Model* m, const Data* d, real* jacp, const real* point, int body) {
  int nv = m->nv;
  memset(jacp, 0, 3 * nv * sizeof(real));
  int i = m->body_lastdof[body];
  while (i >= 0) {
    int j = m->dof_jntid[i];
    real col[3] = {0, 0, 0};
    if (m->jnt_type[j] == JNT_HINGE) {
      real r[3] = {point[0] - d->xanchor[3 * j], point[1] - d->xanchor[3 * j + 1], point[2] - d->xanchor[3 * j + 2]};
      cross3(col, d->xaxis + 3 * j, r);
    } else if (m->jnt_type[j] == JNT_SLIDE) {
      memcpy(col, d->xaxis + 3 * j, sizeof col);
    } else { /* free: use com-based cdof */
      const real* c = d->cdof + 6 * i;
      int b = m->dof_bodyid[i];
      real r[3];
      for (int k = 0; k < 3; k++) r[k] = point[k] - d->subtree_com[3 * m->body_rootid[b] + k];
      cross3(col, c, r);
      for (int k = 0; k < 3; k++) col[k] += c[3 + k];
    }
    jacp[i] = col[0]; jacp[nv + i] = col[1]; jacp[2 * nv + i] = col[2];
    i = m->dof_parentid[i];
  }
}

/* rotational Jacobian (3 x nv) of a body: hinge -> axis, slide -> 0, free joint's rotational dofs -> body axes (cdof angular part) */
static void jac_rot(const Model* m, const Data* d, real* jacr, int body) {
  int nv = m->nv;
  memset(jacr, 0, 3 * nv * sizeof(real));
  int i = m->body_lastdof[body];
  while (i >= 0) {
    int j = m->dof_jntid[i];
    if (m->jnt_type[j] == JNT_HINGE) { for (int k = 0; k < 3; k++) jacr[k * nv + i] = d->xaxis[3 * j + k]; }
    else if (m->jnt_type[j] == JNT_FREE) { for (int k = 0; k < 3; k++) jacr[k * nv + i] = d->cdof[6 * i + k]; }
    i = m->dof_parentid[i];
  }
}

/* ---- tendon wrapping: restatement of MuJoCo's mju_wrap / wrap_circle / wrap_inside [3P] */
static int is_intersect(const real* p1, const real* p2, const real* p3, const real* p4) {
  real det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1]);
  if (fabs(det) < MINVAL) return 0;
  real a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det;
  real b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det;
  return a >= 0 && a <= 1 && b >= 0 && b <= 1;
}

static real wrap_circle(real* pnt, const real* d, const real* sd, real rad) {
  real sq0 = d[0] * d[0] + d[1] * d[1], sq1 = d[2] * d[2] + d[3] * d[3], sqr = rad * rad;
  if (sq0 < sqr || sq1 < sqr || rad < MINVAL) return -1;
  real dif[2] = {d[2] - d[0], d[3] - d[1]};
  real dd = dif[0] * dif[0] + dif[1] * dif[1];
  if (dd < MINVAL) return -1;
  real a = -(dif[0] * d[0] + dif[1] * d[1]) / dd;
  a = clipr(a, 0, 1);
  real tmp[2] = {a * dif[0] + d[0], a * dif[1] + d[1]};
  if (tmp[0] * tmp[0] + tmp[1] * tmp[1] > sqr && (!sd || sd[0] * tmp[0] + sd[1] * tmp[1] >= 0)) return -1;
  real s0 = sqrt(sq0 - sqr), s1 = sqrt(sq1 - sqr);
  real sol[2][4], good[2];
  for (int i = 0; i < 2; i++) {
    real sgn = i == 0 ? 1 : -1;
    sol[i][0] = (d[0] * sqr + sgn * rad * d[1] * s0) / sq0;
    sol[i][1] = (d[1] * sqr - sgn * rad * d[0] * s0) / sq0;
    sol[i][2] = (d[2] * sqr - sgn * rad * d[3] * s1) / sq1;
    sol[i][3] = (d[3] * sqr + sgn * rad * d[2] * s1) / sq1;
    if (sd) {
      real t[2] = {sol[i][0] + sol[i][2], sol[i][1] + sol[i][3]};
#ifdef MYOO_FLOAT
      /* float build (mirrors the HIP kernel, csrc/myo_physics.h wrap_circle): nearly antipodal tangent points make the sum round-off in float32;
       * its direction is then taken from the chord between the tangent points (always perpendicular to the sum), the sum keeps only its sign */
      real c[2] = {sol[i][0] - sol[i][2], sol[i][1] - sol[i][3]};
      real n2 = t[0] * t[0] + t[1] * t[1], c2 = c[0] * c[0] + c[1] * c[1];
      if (n2 < c2) {
        real inv = 1 / sqrt(c2), sg = (t[1] * c[0] - t[0] * c[1]) < 0 ? -1 : 1;
        good[i] = sg * (c[0] * sd[1] - c[1] * sd[0]) * inv;
      } else
#endif
      {
      real n = sqrt(t[0] * t[0] + t[1] * t[1]);
      if (n > MINVAL) { t[0] /= n; t[1] /= n; }
      good[i] = t[0] * sd[0] + t[1] * sd[1];
      }
    } else {
      real t[2] = {sol[i][0] - sol[i][2], sol[i][1] - sol[i][3]};
      good[i] = -(t[0] * t[0] + t[1] * t[1]);
    }
#ifdef MYOO_FLOAT
    /* float build (mirrors the HIP kernel): a grazing solution (tangent points closer than 1e-3 rad)
     * makes the segment-intersection test meaningless in float; skip it there */
    real gz0 = sol[i][0] - sol[i][2], gz1 = sol[i][1] - sol[i][3];
    if (gz0 * gz0 + gz1 * gz1 < (real)1e-6 * sqr) continue;
#endif
    if (is_intersect(d, sol[i], d + 2, sol[i] + 2)) good[i] = -10000;
  }
  int i = good[0] > good[1] ? 0 : 1;
  memcpy(pnt, sol[i], 4 * sizeof(real));
#ifdef MYOO_FLOAT
  if ((pnt[0] - pnt[2]) * (pnt[0] - pnt[2]) + (pnt[1] - pnt[3]) * (pnt[1] - pnt[3]) >= (real)1e-6 * sqr)
#endif
  if (is_intersect(d, pnt, d + 2, pnt + 2)) return -1;
  return rad * acos(clipr((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqr, -1, 1));
}

static real wrap_inside(real* pnt, const real* d, real rad) {
  const int maxiter = 20;
  const real zinit = 1 - (real)1e-7, tolerance = (real)1e-6;
  real len0 = sqrt(d[0] * d[0] + d[1] * d[1]), len1 = sqrt(d[2] * d[2] + d[3] * d[3]);
  real dif[2] = {d[2] - d[0], d[3] - d[1]};
  real dd = dif[0] * dif[0] + dif[1] * dif[1];
  if (len0 <= rad || len1 <= rad || rad < MINVAL || len0 < MINVAL || len1 < MINVAL) return -1;
  if (dd > MINVAL) {
    real a = -(dif[0] * d[0] + dif[1] * d[1]) / dd;
    if (a > 0 && a < 1) {
      real t[2] = {a * dif[0] + d[0], a * dif[1] + d[1]};
      if (sqrt(t[0] * t[0] + t[1] * t[1]) <= rad) return -1;
    }
  }
  pnt[0] = (real)0.5 * (d[0] + d[2]); pnt[1] = (real)0.5 * (d[1] + d[3]);
  real n = sqrt(pnt[0] * pnt[0] + pnt[1] * pnt[1]);
  if (n > MINVAL) { pnt[0] *= rad / n; pnt[1] *= rad / n; }
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  real A = rad / len0, B = rad / len1;
  real cosG = (len0 * len0 + len1 * len1 - dd) / (2 * len0 * len1);
  if (cosG < -1 + MINVAL) return -1;
  if (cosG > 1 - MINVAL) return 0;
  real G = acos(cosG);
#ifdef MYOO_FLOAT
  /* float build: Newton on theta = asin(z) (same root, well conditioned near z -> 1); mirrors the HIP kernel */
  (void)zinit;
  real th = (real)1.5707963267948966 - (real)4.4721360e-4;
  real sn = sin(th), f = asin(A * sn) + asin(B * sn) - 2 * th + G;
  if (f > 0) return 0;
  for (int iter = 0; iter < maxiter && fabs(f) > tolerance; iter++) {
    real cs = cos(th);
    real df = A * cs / maxr(MINVAL, sqrt(1 - A * A * sn * sn)) + B * cs / maxr(MINVAL, sqrt(1 - B * B * sn * sn)) - 2;
    th = clipr(th - f / df, (real)1e-6, (real)1.5707963267948966);
    sn = sin(th);
    f = asin(A * sn) + asin(B * sn) - 2 * th + G;
  }
  real vec[2], ang;
  if (d[0] * d[3] - d[1] * d[2] > 0) { vec[0] = d[0] / len0; vec[1] = d[1] / len0; ang = th - asin(A * sn); }
  else { vec[0] = d[2] / len1; vec[1] = d[3] / len1; ang = th - asin(B * sn); }
#else
  real z = zinit;
  real f = asin(A * z) + asin(B * z) - 2 * asin(z) + G;
  if (f > 0) return 0;
  int iter;
  for (iter = 0; iter < maxiter && fabs(f) > tolerance; iter++) {
    real df = A / maxr(MINVAL, sqrt(1 - z * z * A * A)) + B / maxr(MINVAL, sqrt(1 - z * z * B * B)) -
              2 / maxr(MINVAL, sqrt(1 - z * z));
    if (df > -MINVAL) return 0;
    real z1 = z - f / df;
    if (z1 > z) return 0;
    z = z1;
    f = asin(A * z) + asin(B * z) - 2 * asin(z) + G;
    if (f > tolerance) return 0;
  }
  if (iter >= maxiter) return 0;
  real vec[2], ang;
  if (d[0] * d[3] - d[1] * d[2] > 0) { vec[0] = d[0] / len0; vec[1] = d[1] / len0; ang = asin(z) - asin(A * z); }
  else { vec[0] = d[2] / len1; vec[1] = d[3] / len1; ang = asin(z) - asin(B * z); }
#endif
  pnt[0] = rad * (cos(ang) * vec[0] - sin(ang) * vec[1]);
  pnt[1] = rad * (sin(ang) * vec[0] + cos(ang) * vec[1]);
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  return 0;
}

static real wrap_geom(real* wpnt, const real* x0, const real* x1, const real* xpos, const real* xmat, real radius,
                      int type, const real* side) {
  real p[6], s[3] = {0, 0, 0}, tmp[3], axis[6], d[4], sd[2], pnt[4], res[6];
  for (int k = 0; k < 3; k++) tmp[k] = x0[k] - xpos[k];
  matT_vec3(p, xmat, tmp);
  for (int k = 0; k < 3; k++) tmp[k] = x1[k] - xpos[k];
  matT_vec3(p + 3, xmat, tmp);
  if (norm3(p) < MINVAL || norm3(p + 3) < MINVAL) return -1;
  if (side) {
    for (int k = 0; k < 3; k++) tmp[k] = side[k] - xpos[k];
    matT_vec3(s, xmat, tmp);
  }
  if (type == WRAP_SPHERE) {
    memcpy(axis, p, 3 * sizeof(real));
    normalize3(axis);
    real nrmv[3];
    cross3(nrmv, p, p + 3);
    real nrm = norm3(nrmv);
    if (nrm < MINVAL) {
      int i = 0;
      if (fabs(axis[1]) > fabs(axis[0]) && fabs(axis[1]) > fabs(axis[2])) i = 1;
      if (fabs(axis[2]) > fabs(axis[0]) && fabs(axis[2]) > fabs(axis[1])) i = 2;
      real t[3] = {1, 1, 1};
      t[i] = 0;
      cross3(nrmv, axis, t);
      nrm = norm3(nrmv);
    }
    for (int k = 0; k < 3; k++) nrmv[k] /= nrm;
    cross3(axis + 3, nrmv, axis);
    normalize3(axis + 3);
    d[0] = dot3(p, axis); d[1] = dot3(p, axis + 3); d[2] = dot3(p + 3, axis); d[3] = dot3(p + 3, axis + 3);
    if (side) { sd[0] = dot3(s, axis); sd[1] = dot3(s, axis + 3); }
  } else {
    d[0] = p[0]; d[1] = p[1]; d[2] = p[3]; d[3] = p[4];
    if (side) { sd[0] = s[0]; sd[1] = s[1]; }
  }
  real wlen;
  if (side && sqrt(sd[0] * sd[0] + sd[1] * sd[1]) < radius) {
    wlen = wrap_inside(pnt, d, radius);
  } else {
    if (side) {
      real n = sqrt(sd[0] * sd[0] + sd[1] * sd[1]);
      if (n > MINVAL) { sd[0] /= n; sd[1] /= n; }
    }
    wlen = wrap_circle(pnt, d, side ? sd : NULL, radius);
  }
  if (wlen < 0) return -1;
  if (type == WRAP_SPHERE) {
    for (int k = 0; k < 3; k++) {
      res[k] = axis[k] * pnt[0] + axis[3 + k] * pnt[1];
      res[3 + k] = axis[k] * pnt[2] + axis[3 + k] * pnt[3];
    }
  } else {
    real L0 = sqrt((p[0] - pnt[0]) * (p[0] - pnt[0]) + (p[1] - pnt[1]) * (p[1] - pnt[1]));
    real L1 = sqrt((p[3] - pnt[2]) * (p[3] - pnt[2]) + (p[4] - pnt[3]) * (p[4] - pnt[3]));
    real tot = L0 + wlen + L1;
    res[0] = pnt[0]; res[1] = pnt[1]; res[3] = pnt[2]; res[4] = pnt[3];
    res[2] = p[2] + (p[5] - p[2]) * L0 / tot;
    res[5] = p[2] + (p[5] - p[2]) * (L0 + wlen) / tot;
    real h = res[5] - res[2];
    wlen = sqrt(wlen * wlen + h * h);
  }
  mat_vec3(wpnt, xmat, res);
  mat_vec3(wpnt + 3, xmat, res + 3);
  for (int k = 0; k < 3; k++) { wpnt[k] += xpos[k]; wpnt[3 + k] += xpos[k]; }
  return wlen;
}

static void tendon(const Model* m, Data* d) { /* mj_tendon, spatial tendons [3P] */
  int nv = m->nv;
  real* jac0 = d->wk;
  real* jac1 = d->wk + 3 * nv;
  memset(d->ten_J, 0, (size_t)m->ntendon * nv * sizeof(real));
  for (int t = 0; t < m->ntendon; t++) {
    int adr = m->tendon_adr[t], num = m->tendon_num[t];
    real divisor = 1, L = 0;
    real* J = d->ten_J + (size_t)t * nv;
    int j = 0;
    while (j < num - 1) {
      int type0 = m->wrap_type[adr + j], type1 = m->wrap_type[adr + j + 1];
      if (type0 == WRAP_PULLEY || type1 == WRAP_PULLEY) {
        if (type0 == WRAP_PULLEY) divisor = m->wrap_prm[adr + j];
        j++;
        continue;
      }
      int id0 = m->wrap_objid[adr + j], id1 = m->wrap_objid[adr + j + 1];
      real wpnt[12];
      int wbody[4], npnt;
      real wlen = -1;
      memcpy(wpnt, d->site_xpos + 3 * id0, 3 * sizeof(real));
      wbody[0] = m->site_bodyid[id0];
      int idg = -1;
      if (type1 == WRAP_SPHERE || type1 == WRAP_CYLINDER) {
        idg = id1;
        id1 = m->wrap_objid[adr + j + 2];
        int sideid = (int)lround((double)m->wrap_prm[adr + j + 1]);
        wlen = wrap_geom(wpnt + 3, d->site_xpos + 3 * id0, d->site_xpos + 3 * id1, d->geom_xpos + 3 * idg,
                         d->geom_xmat + 9 * idg, m->geom_size[3 * idg], type1,
                         sideid >= 0 ? d->site_xpos + 3 * sideid : NULL);
      }
      if (wlen < 0) {
        memcpy(wpnt + 3, d->site_xpos + 3 * id1, 3 * sizeof(real));
        wbody[1] = m->site_bodyid[id1];
        npnt = 2;
      } else {
        wbody[1] = wbody[2] = m->geom_bodyid[idg];
        memcpy(wpnt + 9, d->site_xpos + 3 * id1, 3 * sizeof(real));
        wbody[3] = m->site_bodyid[id1];
        npnt = 4;
      }
      for (int k = 0; k < npnt - 1; k++) {
        if (npnt == 4 && k == 1) { L += wlen / divisor; continue; }
        real dif[3] = {wpnt[3 * k + 3] - wpnt[3 * k], wpnt[3 * k + 4] - wpnt[3 * k + 1], wpnt[3 * k + 5] - wpnt[3 * k + 2]};
        real dist = norm3(dif);
        L += dist / divisor;
        if (wbody[k] != wbody[k + 1] && dist > MINVAL) {
          for (int c = 0; c < 3; c++) dif[c] /= dist;
          jac_point(m, d, jac0, wpnt + 3 * k, wbody[k]);
          jac_point(m, d, jac1, wpnt + 3 * k + 3, wbody[k + 1]);
          for (int i = 0; i < nv; i++)
            J[i] += (dif[0] * (jac1[i] - jac0[i]) + dif[1] * (jac1[nv + i] - jac0[nv + i]) +
                     dif[2] * (jac1[2 * nv + i] - jac0[2 * nv + i])) / divisor;
        }
      }
      j += idg >= 0 ? 2 : 1;
    }
    d->ten_length[t] = L;
  }
}

static void crb(const Model* m, Data* d) { /* mj_crb [3P] */
  int nv = m->nv, nb = m->nbody;
  memcpy(d->crb, d->cinert, 10 * nb * sizeof(real));
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * i + k];
  }
  memset(d->qM, 0, m->nM * sizeof(real));
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    real buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    int j = i;
    while (j >= 0) {
      real s = 0;
      for (int k = 0; k < 6; k++) s += d->cdof[6 * j + k] * buf[k];
      d->qM[adr++] = s;
      j = m->dof_parentid[j];
    }
    d->qM[m->dof_Madr[i]] += m->dof_armature[i];
  }
}

static void factor_m(const Model* m, const real* qM, real* qLD, real* diaginv) { /* mj_factorM: M = L' D L */
  int nv = m->nv;
  memcpy(qLD, qM, m->nM * sizeof(real));
  for (int k = nv - 1; k >= 0; k--) {
    int akk = m->dof_Madr[k];
    int i = m->dof_parentid[k], aki = akk + 1;
    while (i >= 0) {
      real tmp = qLD[aki] / qLD[akk];
      int aij = m->dof_Madr[i], akj = aki, j = i;
      while (j >= 0) { qLD[aij++] -= qLD[akj++] * tmp; j = m->dof_parentid[j]; }
      qLD[aki] = tmp;
      i = m->dof_parentid[i];
      aki++;
    }
    diaginv[k] = 1 / qLD[akk];
  }
}

static void solve_ld(const Model* m, real* x, const real* qLD, const real* diaginv) { /* mj_solveLD */
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    if (x[i] != 0) {
      int a = m->dof_Madr[i] + 1, j = m->dof_parentid[i];
      while (j >= 0) { x[j] -= qLD[a++] * x[i]; j = m->dof_parentid[j]; }
    }
  }
  for (int i = 0; i < nv; i++) x[i] *= diaginv[i];
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i] + 1, j = m->dof_parentid[i];
    while (j >= 0) { x[i] -= qLD[a++] * x[j]; j = m->dof_parentid[j]; }
  }
}

static void mul_m(const Model* m, const Data* d, real* res, const real* v) { /* res = M v (sparse) */
  int nv = m->nv;
  memset(res, 0, nv * sizeof(real));
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i];
    res[i] += d->qM[a] * v[i];
    int j = m->dof_parentid[i];
    a++;
    while (j >= 0) {
      res[i] += d->qM[a] * v[j];
      res[j] += d->qM[a] * v[i];
      a++;
      j = m->dof_parentid[j];
    }
  }
}

/* ------------------------------------------------------------------ collision */
static void make_frame(real* frame) { /* mju_makeFrame [3P] */
  normalize3(frame);
  real* y = frame + 3;
  if (norm3(y) < (real)0.5) {
    y[0] = y[1] = y[2] = 0;
    if (frame[1] < (real)0.5 && frame[1] > (real)-0.5) y[1] = 1; else y[2] = 1;
  }
  real t = dot3(frame, y);
  for (int k = 0; k < 3; k++) y[k] -= t * frame[k];
  normalize3(y);
  cross3(frame + 6, frame, y);
}

static int sphere_sphere(Contact* c, real margin, const real* p1, real r1, const real* p2, real r2) {
  real dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  real cd = norm3(dif);
  if (cd > margin + r1 + r2) return 0;
  if (cd < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; } else { for (int k = 0; k < 3; k++) dif[k] /= cd; }
  c->dist = cd - r1 - r2;
  for (int k = 0; k < 3; k++) c->pos[k] = p1[k] + dif[k] * (r1 + (real)0.5 * c->dist);
  memset(c->frame, 0, sizeof c->frame);
  memcpy(c->frame, dif, sizeof dif);
  return 1;
}

static int capsule_capsule(Contact* c, real margin, const real* pos1, const real* mat1, const real* size1,
                           const real* pos2, const real* mat2, const real* size2) { /* mjraw_CapsuleCapsule [3P] */
  real a1[3] = {mat1[2], mat1[5], mat1[8]}, a2[3] = {mat2[2], mat2[5], mat2[8]};
  real dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  real ma = 1, mb = -dot3(a1, a2), mc = 1, u = -dot3(a1, dif), v = dot3(a2, dif);
  real det = ma * mc - mb * mb;
  real x1, x2;
  if (fabs(det) >= MINVAL) {
    x1 = (mc * u - mb * v) / det;
    x2 = (ma * v - mb * u) / det;
    if (x1 > size1[1]) { x1 = size1[1]; x2 = (v - mb * size1[1]) / mc; }
    else if (x1 < -size1[1]) { x1 = -size1[1]; x2 = (v + mb * size1[1]) / mc; }
    if (x2 > size2[1]) { x2 = size2[1]; x1 = clipr((u - mb * size2[1]) / ma, -size1[1], size1[1]); }
    else if (x2 < -size2[1]) { x2 = -size2[1]; x1 = clipr((u + mb * size2[1]) / ma, -size1[1], size1[1]); }
  } else {
    /* parallel axes: nearest points from the segment mid-overlap (single contact) */
    x1 = clipr(u, -size1[1], size1[1]);
    x2 = clipr(v - mb * x1, -size2[1], size2[1]);
    x1 = clipr(u - mb * x2, -size1[1], size1[1]);
  }
  real v1[3], v2[3];
  for (int k = 0; k < 3; k++) { v1[k] = pos1[k] + a1[k] * x1; v2[k] = pos2[k] + a2[k] * x2; }
  return sphere_sphere(c, margin, v1, size1[0], v2, size2[0]);
}

/* ---- general convex pair (ellipsoid pads): margin-inflated MPR, see convex section below */
static int convex_pair(const Model* m, const Data* d, Contact* c, real margin, int g1, int g2);
static void convex_hfield(const Model* m, Data* d, real margin, real gap, int g1, int g2);

static void contact_params(const Model* m, Contact* c, int g1, int g2) { /* mj_contactParam [3P] */
  int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    c->dim = m->geom_condim[g];
    memcpy(c->solref, m->geom_solref + 2 * g, 2 * sizeof(real));
    memcpy(c->solimp, m->geom_solimp + 5 * g, 5 * sizeof(real));
    c->friction[0] = c->friction[1] = m->geom_friction[3 * g];
    c->friction[2] = m->geom_friction[3 * g + 1];
    c->friction[3] = c->friction[4] = m->geom_friction[3 * g + 2];
    return;
  }
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  real s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
  if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
  else if (s1 < MINVAL && s2 < MINVAL) mix = (real)0.5;
  else if (s1 < MINVAL) mix = 0;
  else mix = 1;
  const real *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
  if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
  else for (int k = 0; k < 2; k++) c->solref[k] = minr(r1[k], r2[k]);
  for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  real f[3];
  for (int k = 0; k < 3; k++) f[k] = maxr(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
  c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = c->friction[4] = f[2];
}

static int plane_sphere_dist(const real* ppos, const real* pmat, const real* x, real* nrm) {
  (void)ppos; (void)pmat; (void)x; (void)nrm;
  return 0;
}

static void collision(const Model* m, Data* d) { /* mj_collision over the compile-time pair table */
  d->ncon = 0;
  d->ncon_dropped = 0;
  if (m->disable_contact) return;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom[2 * p], g2 = m->pair_geom[2 * p + 1];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    real margin = maxr(m->geom_margin[g1], m->geom_margin[g2]);
    real gap = maxr(m->geom_gap[g1], m->geom_gap[g2]);
    const real *x1 = d->geom_xpos + 3 * g1, *x2 = d->geom_xpos + 3 * g2;
    /* bounding-sphere filter (planes: signed distance of the sphere to the plane) */
    if (t1 == GEOM_PLANE) {
      const real* R = d->geom_xmat + 9 * g1;
      real nrm[3] = {R[2], R[5], R[8]};
      real dif[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
      if (dot3(dif, nrm) > m->geom_rbound[g2] + margin) continue;
    } else {
      real dif[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
      real bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (dot3(dif, dif) > bound * bound) continue;
    }
    /* cylinder cap filter: a bounding sphere entirely beyond one of the cylinder's cap planes cannot touch it (the scene's pedestal is a
       tall cylinder whose own bounding sphere reaches far above its top) */
    if (t1 == GEOM_CYLINDER || t2 == GEOM_CYLINDER) {
      int gc = t1 == GEOM_CYLINDER ? g1 : g2, go = t1 == GEOM_CYLINDER ? g2 : g1;
      const real* Rc = d->geom_xmat + 9 * gc;
      real ax[3] = {Rc[2], Rc[5], Rc[8]};
      real dif[3] = {d->geom_xpos[3 * go] - d->geom_xpos[3 * gc], d->geom_xpos[3 * go + 1] - d->geom_xpos[3 * gc + 1],
                     d->geom_xpos[3 * go + 2] - d->geom_xpos[3 * gc + 2]};
      real h = dot3(dif, ax);
      if ((h < 0 ? -h : h) - m->geom_size[3 * gc + 1] - m->geom_rbound[go] > margin) continue;
    }
    if (t1 == GEOM_HFIELD) { convex_hfield(m, d, margin, gap, g1, g2); continue; }
    Contact c;
    memset(&c, 0, sizeof c);
    int hit = 0;
    if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) {
      hit = capsule_capsule(&c, margin, x1, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, x2, d->geom_xmat + 9 * g2,
                            m->geom_size + 3 * g2);
    } else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) {
      /* mjc_PlaneCapsule: the two end spheres against the plane */
      const real* R = d->geom_xmat + 9 * g1;
      real nrm[3] = {R[2], R[5], R[8]};
      const real* R2 = d->geom_xmat + 9 * g2;
      real ax[3] = {R2[2], R2[5], R2[8]};
      real r = m->geom_size[3 * g2], h = m->geom_size[3 * g2 + 1];
      for (int s = -1; s <= 1 && d->ncon < NCON_MAX; s += 2) {
        real cpos[3] = {x2[0] + s * h * ax[0], x2[1] + s * h * ax[1], x2[2] + s * h * ax[2]};
        real dif[3] = {cpos[0] - x1[0], cpos[1] - x1[1], cpos[2] - x1[2]};
        real dist = dot3(dif, nrm) - r;
        if (dist > margin) continue;
        Contact* cc = &d->con[d->ncon];
        memset(cc, 0, sizeof *cc);
        cc->dist = dist;
        for (int k = 0; k < 3; k++) cc->pos[k] = cpos[k] - nrm[k] * (r + (real)0.5 * dist);
        memcpy(cc->frame, nrm, sizeof nrm);
        /* frame y-axis along the capsule axis, as MuJoCo does for plane-capsule */
        memcpy(cc->frame + 3, ax, sizeof ax);
        real t = dot3(cc->frame, cc->frame + 3);
        for (int k = 0; k < 3; k++) cc->frame[3 + k] -= t * cc->frame[k];
        make_frame(cc->frame);
        contact_params(m, cc, g1, g2);
        cc->includemargin = margin - gap;
        cc->geom1 = g1; cc->geom2 = g2;
        d->ncon++;
      }
      continue;
    } else if ((t1 == GEOM_CAPSULE || t1 == GEOM_ELLIPSOID || t1 == GEOM_CYLINDER || t1 == GEOM_SPHERE || t1 == GEOM_BOX || (t1 == GEOM_MESH && m->mesh_vert)) &&
               (t2 == GEOM_CAPSULE || t2 == GEOM_ELLIPSOID || t2 == GEOM_CYLINDER || t2 == GEOM_SPHERE || t2 == GEOM_BOX || (t2 == GEOM_MESH && m->mesh_vert))) {
      /* general convex pair.  Boxes and convex-hull meshes (TrackEnv: table, object) go through the same MPR as the ellipsoid pads;
         MuJoCo uses analytic capsule-box / sphere-box routines (1-2 contacts) where this gives the single MPR contact [deviation, DESIGN 3] */
      if (m->disable_ellipsoid) continue;
      hit = convex_pair(m, d, &c, margin, g1, g2);
    } else if (t1 == GEOM_PLANE && (t2 == GEOM_BOX || (t2 == GEOM_MESH && m->mesh_vert))) {
      /* mjc_PlaneBox [3P]: every corner below the box centre and within the margin, at most 4.  Plane - mesh (mjc_PlaneConvex): the deepest
         hull vertex (MuJoCo adds up to three neighbouring vertices; the planes of the TrackEnv scene lie under the table) */
      const real* R = d->geom_xmat + 9 * g1;
      real nrm[3] = {R[2], R[5], R[8]};
      const real* R2 = d->geom_xmat + 9 * g2;
      const real* sz = m->geom_size + 3 * g2;
      real vec0[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
      real dist0 = dot3(vec0, nrm);
      real cdist[4], cpt[12];
      int nc = 0;
      if (t2 == GEOM_BOX) {
        for (int i = 0; i < 8 && nc < 4; i++) {
          real vec[3] = {(i & 1) ? sz[0] : -sz[0], (i & 2) ? sz[1] : -sz[1], (i & 4) ? sz[2] : -sz[2]}, corner[3];
          mat_vec3(corner, R2, vec);
          real ldist = dot3(nrm, corner);
          if (dist0 + ldist > margin || ldist > 0) continue;
          cdist[nc] = dist0 + ldist;
          for (int k = 0; k < 3; k++) cpt[3 * nc + k] = corner[k] + x2[k] - nrm[k] * cdist[nc] * (real)0.5;
          nc++;
        }
      } else {
        real nl[3], best = 0, pw[3];
        int bi = -1;
        matT_vec3(nl, R2, nrm);
        const real* V = m->mesh_vert + 3 * m->geom_meshadr[g2];
        for (int i = 0; i < m->geom_meshnum[g2]; i++) { real t = dot3(V + 3 * i, nl); if (bi < 0 || t < best) { best = t; bi = i; } }
        if (bi >= 0 && dist0 + best <= margin) {
          mat_vec3(pw, R2, V + 3 * bi);
          cdist[0] = dist0 + best;
          for (int k = 0; k < 3; k++) cpt[k] = x2[k] + pw[k] - nrm[k] * cdist[0] * (real)0.5;
          nc = 1;
        }
      }
      for (int q = 0; q < nc; q++) {
        if (d->ncon >= NCON_MAX) { d->ncon_dropped++; continue; }
        Contact* cc = &d->con[d->ncon];
        memset(cc, 0, sizeof *cc);
        cc->dist = cdist[q];
        memcpy(cc->pos, cpt + 3 * q, 3 * sizeof(real));
        memcpy(cc->frame, nrm, sizeof nrm);
        make_frame(cc->frame);
        contact_params(m, cc, g1, g2);
        cc->includemargin = margin - gap;
        cc->geom1 = g1; cc->geom2 = g2;
        d->ncon++;
      }
      continue;
    } else if (t1 == GEOM_PLANE && (t2 == GEOM_SPHERE || t2 == GEOM_ELLIPSOID || t2 == GEOM_CYLINDER)) {
      /* mjc_PlaneSphere / mjc_PlaneConvex (ellipsoid: deepest support point) / mjc_PlaneCylinder [3P] */
      const real* R = d->geom_xmat + 9 * g1;
      real nrm[3] = {R[2], R[5], R[8]};
      const real* R2 = d->geom_xmat + 9 * g2;
      const real* sz = m->geom_size + 3 * g2;
      real cdist[4], cpt[12];
      int nc = 0;
      real vec0[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
      real dist0 = dot3(vec0, nrm);
      if (t2 == GEOM_SPHERE) {
        real dist = dist0 - sz[0];
        if (dist <= margin) { cdist[0] = dist; for (int k = 0; k < 3; k++) cpt[k] = x2[k] - nrm[k] * (sz[0] + (real)0.5 * dist); nc = 1; }
      } else if (t2 == GEOM_ELLIPSOID) {
        real nl[3], sp[3], pw[3];
        matT_vec3(nl, R2, nrm);                           /* plane normal in the ellipsoid frame */
        real sv[3] = {sz[0] * nl[0], sz[1] * nl[1], sz[2] * nl[2]};
        real nn = norm3(sv);
        for (int k = 0; k < 3; k++) sp[k] = nn > MINVAL ? -sz[k] * sv[k] / nn : 0;   /* support point along -normal */
        mat_vec3(pw, R2, sp);
        real dist = dist0 + dot3(pw, nrm);
        if (dist <= margin) { cdist[0] = dist; for (int k = 0; k < 3; k++) cpt[k] = x2[k] + pw[k] - nrm[k] * (real)0.5 * dist; nc = 1; }
      } else {
        real axis[3] = {R2[2], R2[5], R2[8]};
        real prjaxis = dot3(nrm, axis);
        if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
        real vec[3];
        for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - nrm[k];
        real len_sqr = dot3(vec, vec);
        if (len_sqr >= MINVAL) { real sc = sz[0] / sqrt(len_sqr); for (int k = 0; k < 3; k++) vec[k] *= sc; }
        else { vec[0] = R2[0] * sz[0]; vec[1] = R2[3] * sz[0]; vec[2] = R2[6] * sz[0]; }
        real prjvec = dot3(vec, nrm);
        for (int k = 0; k < 3; k++) axis[k] *= sz[1];
        prjaxis *= sz[1];
        if (dist0 + prjaxis + prjvec <= margin) {
          cdist[0] = dist0 + prjaxis + prjvec;
          for (int k = 0; k < 3; k++) cpt[k] = x2[k] + vec[k] + axis[k] - nrm[k] * cdist[0] * (real)0.5;
          nc = 1;
          if (dist0 - prjaxis + prjvec <= margin) {
            cdist[nc] = dist0 - prjaxis + prjvec;
            for (int k = 0; k < 3; k++) cpt[3 * nc + k] = x2[k] + vec[k] - axis[k] - nrm[k] * cdist[nc] * (real)0.5;
            nc++;
          }
          real prjvec1 = -prjvec * (real)0.5;
          if (dist0 + prjaxis + prjvec1 <= margin) {
            real vec1[3];
            cross3(vec1, vec, axis);
            normalize3(vec1);
            for (int k = 0; k < 3; k++) vec1[k] *= sz[0] * (real)0.8660254037844386;
            for (int sgn = 1; sgn >= -1; sgn -= 2) {
              cdist[nc] = dist0 + prjaxis + prjvec1;
              for (int k = 0; k < 3; k++) cpt[3 * nc + k] = x2[k] + sgn * vec1[k] + axis[k] - vec[k] * (real)0.5 - nrm[k] * cdist[nc] * (real)0.5;
              nc++;
            }
          }
        }
      }
      for (int q = 0; q < nc; q++) {
        if (d->ncon >= NCON_MAX) { d->ncon_dropped++; continue; }
        Contact* cc = &d->con[d->ncon];
        memset(cc, 0, sizeof *cc);
        cc->dist = cdist[q];
        memcpy(cc->pos, cpt + 3 * q, 3 * sizeof(real));
        memcpy(cc->frame, nrm, sizeof nrm);
        make_frame(cc->frame);
        contact_params(m, cc, g1, g2);
        cc->includemargin = margin - gap;
        cc->geom1 = g1; cc->geom2 = g2;
        d->ncon++;
      }
      continue;
    } else {
      d->warning |= 4;   /* a pair type with no narrow phase here (mesh, box, hfield) came within bounding-sphere range: the model
                            compiler only admits such pairs when they are provably out of reach, so this is an error */
      continue;
    }
    if (!hit) continue;
    if (d->ncon >= NCON_MAX) { d->ncon_dropped++; continue; }
    make_frame(c.frame);
    contact_params(m, &c, g1, g2);
    if (m->pair_condim[p] > 0) c.dim = m->pair_condim[p];   /* explicit <contact><pair condim=...> */
    c.includemargin = margin - gap;
    c.geom1 = g1; c.geom2 = g2;
    d->con[d->ncon++] = c;
  }
  (void)plane_sphere_dist;
}

/* ---- convex support functions and MPR (Minkowski Portal Refinement) penetration.
 * MuJoCo 3.2.8 routes ellipsoid pairs through libccd's ccdMPRPenetration with both shapes
 * inflated by margin/2 (engine_collision_convex.c) [3P].  This is a restatement of the
 * published XenoCollide/MPR algorithm (G. Snethen, Game Programming Gems 7), with
 * tolerance 1e-6 and 50 iterations like MuJoCo's ccd_tolerance / ccd_iterations defaults. */
typedef struct { const real *pos, *mat, *size; int type; real margin; const real* verts; int nvert; } CObj;

static void support_local(const CObj* o, const real* dl, real* out) {
  switch (o->type) {
    case GEOM_SPHERE: { real n = norm3(dl); for (int k = 0; k < 3; k++) out[k] = n > MINVAL ? dl[k] / n * o->size[0] : 0; break; }
    case GEOM_CAPSULE: {
      real n = norm3(dl);
      for (int k = 0; k < 3; k++) out[k] = n > MINVAL ? dl[k] / n * o->size[0] : 0;
      out[2] += dl[2] >= 0 ? o->size[1] : -o->size[1];
      break;
    }
    case GEOM_ELLIPSOID: {
      real s[3] = {o->size[0] * dl[0], o->size[1] * dl[1], o->size[2] * dl[2]};
      real n = norm3(s);
      for (int k = 0; k < 3; k++) out[k] = n > MINVAL ? o->size[k] * s[k] / n : 0;
      break;
    }
    case GEOM_CYLINDER: {
      real n = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
      out[0] = n > MINVAL ? dl[0] / n * o->size[0] : 0;
      out[1] = n > MINVAL ? dl[1] / n * o->size[0] : 0;
      out[2] = dl[2] >= 0 ? o->size[1] : -o->size[1];
      break;
    }
    case GEOM_BOX: for (int k = 0; k < 3; k++) out[k] = dl[k] >= 0 ? o->size[k] : -o->size[k]; break;
    case GEOM_MESH: { /* convex hull: the vertex furthest along the direction (MuJoCo climbs the hull's vertex graph to the same vertex) */
      int best = 0;
      real bd = dot3(o->verts, dl);
      for (int i = 1; i < o->nvert; i++) { real t = dot3(o->verts + 3 * i, dl); if (t > bd) { bd = t; best = i; } }
      for (int k = 0; k < 3; k++) out[k] = o->verts[3 * best + k];
      break;
    }
    case GEOM_PRISM: { /* prism_support [3P]: only the bottom (0..2) or top (3..5) triangle can be extremal, by the sign of dir_z */
      const real* V = o->size;
      int i0 = dl[2] < 0 ? 0 : 3, best = i0;
      real bd = dot3(V + 3 * i0, dl);
      for (int i = i0 + 1; i < i0 + 3; i++) { real t = dot3(V + 3 * i, dl); if (t > bd) { bd = t; best = i; } }
      out[0] = V[3 * best]; out[1] = V[3 * best + 1]; out[2] = V[3 * best + 2];
      break;
    }
    default: out[0] = out[1] = out[2] = 0;
  }
}

static void support_world(const CObj* o, const real* dir, real* out) {
  real dl[3], pl[3];
  matT_vec3(dl, o->mat, dir);
  support_local(o, dl, pl);
  mat_vec3(out, o->mat, pl);
  real n = norm3(dir);
  for (int k = 0; k < 3; k++) out[k] += o->pos[k] + (n > MINVAL ? dir[k] / n * o->margin : 0);
}

typedef struct { real v[3], v1[3], v2[3]; } Sup;

static void mink_support(const CObj* a, const CObj* b, const real* dir, Sup* s) {
  real nd[3] = {-dir[0], -dir[1], -dir[2]};
  support_world(a, dir, s->v1);
  support_world(b, nd, s->v2);
  for (int k = 0; k < 3; k++) s->v[k] = s->v1[k] - s->v2[k];
}

static void portal_dir(const Sup* p, real* dir) { /* normal of the portal triangle p[1],p[2],p[3] */
  real a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = p[2].v[k] - p[1].v[k]; b[k] = p[3].v[k] - p[1].v[k]; }
  cross3(dir, a, b);
  normalize3(dir);
}

/* returns 1 if penetrating; fills depth, dir (from obj1 to obj2), pos */
static long g_mpr_hist[64];   /* developer statistics: refinement steps per MPR call */
void myoo_mpr_hist(long* out, int reset) { for (int k = 0; k < 64; k++) { out[k] = g_mpr_hist[k]; if (reset) g_mpr_hist[k] = 0; } }
/* Output convention of the penetration query (test hook; default 0).
 *   0: final support plane, refinement converged to 1e-11 (what the HIP path implements; DESIGN.md 3, deviation 1)
 *   1: libccd's own output as MuJoCo 3.2.8 receives it (third-party dependency, not vendored in /root/reference: libccd 2.1,
 *      src/mpr.c findPenetr + src/vec3.c ccdVec3PointTriDist2, restated from the published algorithm): refinement stopped at
 *      MuJoCo's ccd_tolerance 1e-6, depth = distance from the origin to the NEAREST POINT OF THE FINAL PORTAL TRIANGLE, direction =
 *      that point normalised.  tests/test_oracle_mpr_modes.py measures the gap between the two on seeded hand states. */
static int g_mpr_mode = 0;
void myoo_set_mpr_mode(int mode) { g_mpr_mode = mode; }
static real seg_dist2(const real* a, const real* b, real* w) {   /* origin to segment ab (libccd ccdVec3PointSegmentDist2 with P = 0) */
  real d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  real t = -(a[0] * d[0] + a[1] * d[1] + a[2] * d[2]) / (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (t <= 0) { for (int k = 0; k < 3; k++) w[k] = a[k]; }
  else if (t >= 1) { for (int k = 0; k < 3; k++) w[k] = b[k]; }
  else { for (int k = 0; k < 3; k++) w[k] = a[k] + t * d[k]; }
  return w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
}
static real tri_dist2(const real* x0, const real* B, const real* Cc, real* w) {  /* origin to triangle: interior solution, else the nearest edge */
  real d1[3], d2[3];
  for (int k = 0; k < 3; k++) { d1[k] = B[k] - x0[k]; d2[k] = Cc[k] - x0[k]; }
  real v = dot3(d1, d1), ww = dot3(d2, d2), pp = dot3(x0, d1), q = dot3(x0, d2), r = dot3(d1, d2);
  real sdiv = ww * v - r * r;
  real s = sdiv != 0 ? (q * r - ww * pp) / sdiv : (real)-1;
  real t = (-s * r - q) / ww;
  const real e = (real)2.2e-16;
  if (s >= -e && s <= 1 + e && t >= -e && t <= 1 + e && t + s <= 1 + e) {
    for (int k = 0; k < 3; k++) w[k] = x0[k] + s * d1[k] + t * d2[k];
    return dot3(w, w);
  }
  real w2[3], best = seg_dist2(x0, B, w), dd = seg_dist2(x0, Cc, w2);
  if (dd < best) { best = dd; for (int k = 0; k < 3; k++) w[k] = w2[k]; }
  dd = seg_dist2(B, Cc, w2);
  if (dd < best) { best = dd; for (int k = 0; k < 3; k++) w[k] = w2[k]; }
  return best;
}
static int mpr_penetration(const CObj* o1, const CObj* o2, real tol, int maxit, real* depth, real* dirout, real* posout) {
  if (g_mpr_mode == 1) { tol = (real)1e-6; maxit = 50; }
  Sup p[4];
  real dir[3], va[3], vb[3];
  /* v0: interior point = centre difference */
  for (int k = 0; k < 3; k++) { p[0].v1[k] = o1->pos[k]; p[0].v2[k] = o2->pos[k]; p[0].v[k] = o1->pos[k] - o2->pos[k]; }
  if (norm3(p[0].v) < MINVAL) p[0].v[0] += (real)1e-5;
  /* v1: support in direction of origin */
  for (int k = 0; k < 3; k++) dir[k] = -p[0].v[k];
  normalize3(dir);
  mink_support(o1, o2, dir, &p[1]);
  if (dot3(p[1].v, dir) < 0) return 0;
  cross3(dir, p[0].v, p[1].v);
  if (norm3(dir) < (real)1e-12) {
    /* origin on the v0-v1 segment: penetration along that line */
    real d1 = norm3(p[1].v);
    *depth = d1;
    for (int k = 0; k < 3; k++) { dirout[k] = p[1].v[k]; posout[k] = (real)0.5 * (p[1].v1[k] + p[1].v2[k]); }
    normalize3(dirout);
    return 1;
  }
  normalize3(dir);
  mink_support(o1, o2, dir, &p[2]);
  if (dot3(p[2].v, dir) < 0) return 0;
  for (int k = 0; k < 3; k++) { va[k] = p[1].v[k] - p[0].v[k]; vb[k] = p[2].v[k] - p[0].v[k]; }
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, p[0].v) > 0) { Sup t = p[1]; p[1] = p[2]; p[2] = t; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
  /* discover portal */
  for (int it = 0;; it++) {
    if (it > maxit) return 0;
    mink_support(o1, o2, dir, &p[3]);
    if (dot3(p[3].v, dir) < 0) return 0;
    int cont = 0;
    cross3(va, p[1].v, p[3].v);
    if (dot3(va, p[0].v) < -MINVAL) { p[2] = p[3]; cont = 1; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v);
      if (dot3(va, p[0].v) < -MINVAL) { p[1] = p[3]; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { va[k] = p[1].v[k] - p[0].v[k]; vb[k] = p[2].v[k] - p[0].v[k]; }
    cross3(dir, va, vb);
    normalize3(dir);
  }
  /* refine portal until the origin ray is enclosed (libccd refinePortal) */
  for (int it = 0;; it++) {
    if (it > maxit) return 0;
    portal_dir(p, dir);
    if (dot3(dir, p[1].v) >= 0) break; /* portal encapsulates the origin */
    Sup v4;
    mink_support(o1, o2, dir, &v4);
    real dv4 = dot3(v4.v, dir);
    real dmin = minr(minr(dv4 - dot3(p[1].v, dir), dv4 - dot3(p[2].v, dir)), dv4 - dot3(p[3].v, dir));
    if (dv4 < 0 || dmin <= tol) return 0;
    cross3(va, v4.v, p[0].v);
    if (dot3(p[1].v, va) > 0) { if (dot3(p[2].v, va) > 0) p[1] = v4; else p[3] = v4; }
    else { if (dot3(p[3].v, va) > 0) p[2] = v4; else p[1] = v4; }
  }
  /* push the portal to the surface of the Minkowski difference (libccd findPenetr) */
  Sup v4;
  for (int it = 0;; it++) {
    portal_dir(p, dir);
    mink_support(o1, o2, dir, &v4);
    real dv4 = dot3(v4.v, dir);
    real dmin = minr(minr(dv4 - dot3(p[1].v, dir), dv4 - dot3(p[2].v, dir)), dv4 - dot3(p[3].v, dir));
    if (dmin <= tol || it > maxit) { g_mpr_hist[it < 63 ? it : 63]++; break; }
    cross3(va, v4.v, p[0].v);
    if (dot3(p[1].v, va) > 0) { if (dot3(p[2].v, va) > 0) p[1] = v4; else p[3] = v4; }
    else { if (dot3(p[3].v, va) > 0) p[2] = v4; else p[1] = v4; }
  }
  /* Output from the final SUPPORT PLANE rather than libccd's nearest point on the portal triangle: for the
   * smooth shapes handled here the support point s(dir) has outward normal exactly `dir`, so (dir, dir.s(dir))
   * is an exact supporting plane of the inflated Minkowski difference.  libccd's triangle-nearest-point output
   * switches between face / edge / vertex cases when the overlap is tiny and is not a continuous function of
   * the pose ("ellipsoid contacts"). */
  *depth = dot3(v4.v, dir);
  /* contact position: barycentric coordinates of the origin in the tetrahedron (v0, portal), applied to the
   * witness points (libccd findPos) -- interpolates along flat / ruled parts such as a capsule's side */
  real bw[4], cr[3];
  cross3(cr, p[1].v, p[2].v); bw[0] = dot3(cr, p[3].v);
  cross3(cr, p[3].v, p[2].v); bw[1] = dot3(cr, p[0].v);
  cross3(cr, p[0].v, p[1].v); bw[2] = dot3(cr, p[3].v);
  cross3(cr, p[2].v, p[1].v); bw[3] = dot3(cr, p[0].v);
  real sum = bw[0] + bw[1] + bw[2] + bw[3];
  if (sum <= 0) {
    bw[0] = 0;
    cross3(cr, p[2].v, p[3].v); bw[1] = dot3(cr, dir);
    cross3(cr, p[3].v, p[1].v); bw[2] = dot3(cr, dir);
    cross3(cr, p[1].v, p[2].v); bw[3] = dot3(cr, dir);
    sum = bw[1] + bw[2] + bw[3];
  }
  real inv = 1 / sum;
  if (g_mpr_mode == 1) {
    real w[3];
    real d2 = tri_dist2(p[1].v, p[2].v, p[3].v, w);
    *depth = sqrt(d2);
    if (*depth > 0) for (int k = 0; k < 3; k++) dir[k] = w[k] / *depth;
  }
  for (int k = 0; k < 3; k++) {
    dirout[k] = dir[k];
    posout[k] = (real)0.5 * inv * (bw[0] * (p[0].v1[k] + p[0].v2[k]) + bw[1] * (p[1].v1[k] + p[1].v2[k]) +
                                   bw[2] * (p[2].v1[k] + p[2].v2[k]) + bw[3] * (p[3].v1[k] + p[3].v2[k]));
  }
  return 1;
}

static int convex_pair(const Model* m, const Data* d, Contact* c, real margin, int g1, int g2) {
  if (m->geom_type[g1] == GEOM_PLANE) return 0; /* plane vs non-capsule convex: pruned at compile time for config models */
  /* coordinates relative to geom1's centre; tolerance converged well below MuJoCo's ccd_tolerance (1e-6) so that
   * the result is the path-independent limit of the portal refinement ("ellipsoid contacts") */
  const real zero3[3] = {0, 0, 0};
  real rel[3];
  for (int k = 0; k < 3; k++) rel[k] = d->geom_xpos[3 * g2 + k] - d->geom_xpos[3 * g1 + k];
  CObj o1 = {zero3, d->geom_xmat + 9 * g1, m->geom_size + 3 * g1, m->geom_type[g1], margin * (real)0.5, NULL, 0};
  CObj o2 = {rel, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, m->geom_type[g2], margin * (real)0.5, NULL, 0};
  if (o1.type == GEOM_MESH) { o1.verts = m->mesh_vert + 3 * m->geom_meshadr[g1]; o1.nvert = m->geom_meshnum[g1]; }
  if (o2.type == GEOM_MESH) { o2.verts = m->mesh_vert + 3 * m->geom_meshadr[g2]; o2.nvert = m->geom_meshnum[g2]; }
  real depth, dir[3], pos[3];
#ifdef MYOO_FLOAT
  if (!mpr_penetration(&o1, &o2, (real)1e-8, 60, &depth, dir, pos)) return 0;
#else
  if (!mpr_penetration(&o1, &o2, (real)1e-11, 100, &depth, dir, pos)) return 0;
#endif
  for (int k = 0; k < 3; k++) pos[k] += d->geom_xpos[3 * g1 + k];
  c->dist = margin - depth;
  /* dir points from the origin to the nearest boundary point of (obj1 - obj2): translating obj2 by
   * depth*dir separates the pair, so dir is the contact normal from geom1 to geom2 */
  for (int k = 0; k < 3; k++) { c->frame[k] = dir[k]; c->pos[k] = pos[k]; }
  for (int k = 3; k < 9; k++) c->frame[k] = 0;
  return 1;
}

/* Height field against a convex primitive: mjc_ConvexHField (engine_collision_convex.c) [3P, restated from the documented algorithm].
 * In the height field's frame: bounding tests, the geom's AABB from six support queries, the sub-grid of cells under it; every row of
 * cells is walked as a zig-zag strip of vertices (r+1, c), (r, c), (r+1, c+1), (r, c+1), ... and each three consecutive vertices span
 * one triangular prism from the base depth up to the terrain surface; a prism whose top lies wholly below the geom is skipped, the
 * others go through the same MPR as every convex pair (prism centre = mean of its six vertices), one contact each, at most 50 per
 * pair (mjMAXCONPAIR).  Margin convention: as for the other convex pairs here (both shapes swept by margin / 2, dist = margin - depth);
 * MuJoCo's own bookkeeping of the margin in this routine is not reproduced from a source and this 1 mm-level detail is parity-unpinned.
 * The height field must be axis-aligned (checked by the model compiler). */
static void convex_hfield(const Model* m, Data* d, real margin, real gap, int g1, int g2) {
  const real* hs = m->hfield_size;
  const int nrow = m->hfield_dims[0], ncol = m->hfield_dims[1];
  const real* p1 = d->geom_xpos + 3 * g1;
  real pos[3];
  for (int k = 0; k < 3; k++) pos[k] = d->geom_xpos[3 * g2 + k] - p1[k];
  const real r2 = m->geom_rbound[g2];
  for (int i = 0; i < 2; i++) if (hs[i] < pos[i] - r2 - margin || -hs[i] > pos[i] + r2 + margin) return;
  if (hs[2] < pos[2] - r2 - margin || -hs[3] > pos[2] + r2 + margin) return;
  /* AABB of geom 2 in the height field frame */
  CObj og = {pos, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, m->geom_type[g2], 0, NULL, 0};
  real lo[3], hi[3];
  for (int a = 0; a < 3; a++) {
    real dir[3] = {0, 0, 0}, s[3];
    dir[a] = 1; support_world(&og, dir, s); hi[a] = s[a];
    dir[a] = -1; support_world(&og, dir, s); lo[a] = s[a];
  }
  if (lo[0] - margin > hs[0] || hi[0] + margin < -hs[0] || lo[1] - margin > hs[1] || hi[1] + margin < -hs[1] ||
      lo[2] - margin > hs[2] || hi[2] + margin < -hs[3]) return;
  int cmin = (int)floor((lo[0] + hs[0]) / (2 * hs[0]) * (ncol - 1)), cmax = (int)ceil((hi[0] + hs[0]) / (2 * hs[0]) * (ncol - 1));
  int rmin = (int)floor((lo[1] + hs[1]) / (2 * hs[1]) * (nrow - 1)), rmax = (int)ceil((hi[1] + hs[1]) / (2 * hs[1]) * (nrow - 1));
  if (cmin < 0) cmin = 0;
  if (rmin < 0) rmin = 0;
  if (cmax > ncol - 1) cmax = ncol - 1;
  if (rmax > nrow - 1) rmax = nrow - 1;
  const real dx = 2 * hs[0] / (ncol - 1), dy = 2 * hs[1] / (nrow - 1);
  const real ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero3[3] = {0, 0, 0};
  int cnt = 0;
  for (int r = rmin; r < rmax; r++) {
    real vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0}, vz[3] = {0, 0, 0};   /* the last three strip vertices (oldest first) */
    int nvert = 0;
    for (int c = cmin; c <= cmax; c++) {
      for (int i = 0; i < 2; i++) {
        const int rr = r + (i == 0 ? 1 : 0);
        vx[0] = vx[1]; vy[0] = vy[1]; vz[0] = vz[1];
        vx[1] = vx[2]; vy[1] = vy[2]; vz[1] = vz[2];
        vx[2] = dx * c - hs[0]; vy[2] = dy * rr - hs[1]; vz[2] = (real)d->hfield_data[rr * ncol + c] * hs[2];
        if (++nvert <= 2) continue;
        if (vz[0] < lo[2] - margin && vz[1] < lo[2] - margin && vz[2] < lo[2] - margin) continue;   /* prism top wholly below the geom */
        /* prism vertices relative to their mean */
        real V[18], cen[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
          V[3 * k] = vx[k]; V[3 * k + 1] = vy[k]; V[3 * k + 2] = -hs[3];
          V[9 + 3 * k] = vx[k]; V[9 + 3 * k + 1] = vy[k]; V[9 + 3 * k + 2] = vz[k];
        }
        for (int k = 0; k < 6; k++) for (int a = 0; a < 3; a++) cen[a] += V[3 * k + a] / 6;
        for (int k = 0; k < 6; k++) for (int a = 0; a < 3; a++) V[3 * k + a] -= cen[a];
        real rel[3] = {pos[0] - cen[0], pos[1] - cen[1], pos[2] - cen[2]};
        CObj o1 = {zero3, ident, V, GEOM_PRISM, margin * (real)0.5, NULL, 0};
        CObj o2 = {rel, d->geom_xmat + 9 * g2, m->geom_size + 3 * g2, m->geom_type[g2], margin * (real)0.5, NULL, 0};
        real depth, dir[3], cp[3];
#ifdef MYOO_FLOAT
        if (!mpr_penetration(&o1, &o2, (real)1e-8, 60, &depth, dir, cp)) continue;
#else
        if (!mpr_penetration(&o1, &o2, (real)1e-11, 100, &depth, dir, cp)) continue;
#endif
        if (d->ncon >= NCON_MAX) { d->ncon_dropped++; continue; }
        Contact* cc = &d->con[d->ncon];
        memset(cc, 0, sizeof *cc);
        cc->dist = margin - depth;
        for (int k = 0; k < 3; k++) { cc->pos[k] = cp[k] + cen[k] + p1[k]; cc->frame[k] = dir[k]; }
        make_frame(cc->frame);
        contact_params(m, cc, g1, g2);
        cc->includemargin = margin - gap;
        cc->geom1 = g1; cc->geom2 = g2;
        d->ncon++;
        if (++cnt >= 50) return;
      }
    }
  }
}

/* ------------------------------------------------------------------ constraints */
static void get_impedance(const real* solimp_in, real pos, real margin, real* imp) { /* getimpedance [3P] */
  real dmin = clipr(solimp_in[0], MINIMP, MAXIMP), dmax = clipr(solimp_in[1], MINIMP, MAXIMP);
  real width = maxr(MINVAL, solimp_in[2]), mid = clipr(solimp_in[3], MINIMP, MAXIMP), power = maxr(1, solimp_in[4]);
  if (dmin == dmax || width <= MINVAL) { *imp = (real)0.5 * (dmin + dmax); return; }
  real x = (pos - margin) / width;
  if (x < 0) x = -x;
  if (x >= 1) { *imp = dmax; return; }
  if (x == 0) { *imp = dmin; return; }
  real y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

static int add_row(Data* d, int nv, int type, int id, real pos, real margin, real diag) {
  int i = d->nefc;
  if (i >= d->nefc_max) return -1;
  memset(d->efc_J + (size_t)i * nv, 0, nv * sizeof(real));
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_diagApprox[i] = diag;
  d->nefc++;
  return i;
}

static void make_constraint(const Model* m, Data* d) { /* mj_makeConstraint: limits then contacts [3P] */
  int nv = m->nv;
  d->nefc = 0;
  /* equality: joint couplings q1 - q1_0 = poly(q2 - q2_0) (mj_instantiateEquality, mjEQ_JOINT) [3P] */
  for (int e = 0; e < m->neq; e++) {
    int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
    const real* a = m->eq_data + 5 * e;
    int q1 = m->jnt_qposadr[j1], d1 = m->jnt_dofadr[j1];
    real pos = d->qpos[q1] - m->qpos0[q1], deriv = 0, diag = m->dof_invweight0[d1];
    if (j2 >= 0) {
      int q2 = m->jnt_qposadr[j2];
      real x = d->qpos[q2] - m->qpos0[q2];
      pos -= a[0] + x * (a[1] + x * (a[2] + x * (a[3] + x * a[4])));
      deriv = a[1] + x * (2 * a[2] + x * (3 * a[3] + x * 4 * a[4]));
      diag += m->dof_invweight0[m->jnt_dofadr[j2]];
    } else {
      pos -= a[0];
    }
    int r = add_row(d, nv, CT_EQUALITY, e, pos, 0, diag);
    if (r >= 0) {
      d->efc_J[(size_t)r * nv + d1] = 1;
      if (j2 >= 0) d->efc_J[(size_t)r * nv + m->jnt_dofadr[j2]] = -deriv;
    }
  }
  /* dof friction loss (mj_instantiateFriction [3P]): one row per dof with frictionloss > 0, J = e_dof, pos = 0 */
  for (int i = 0; i < nv; i++) {
    if (!(m->dof_frictionloss[i] > 0)) continue;
    int r = add_row(d, nv, CT_FRICTION_DOF, i, 0, 0, m->dof_invweight0[i]);
    if (r >= 0) { d->efc_J[(size_t)r * nv + i] = 1; d->efc_frictionloss[r] = m->dof_frictionloss[i]; }
  }
  if (!m->disable_limit) {
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j]) continue;
      if (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE) continue;
      real q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
      for (int side = -1; side <= 1; side += 2) {
        real dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - q);
        if (dist < margin) {
          int da = m->jnt_dofadr[j];
          int r = add_row(d, nv, CT_LIMIT_JOINT, j, dist, margin, m->dof_invweight0[da]);
          if (r >= 0) d->efc_J[(size_t)r * nv + da] = -(real)side;
        }
      }
    }
    for (int t = 0; t < m->ntendon; t++) {
      if (!m->tendon_limited[t]) continue;
      real L = d->ten_length[t], margin = m->tendon_margin[t];
      for (int side = -1; side <= 1; side += 2) {
        real dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - L);
        if (dist < margin) {
          int r = add_row(d, nv, CT_LIMIT_TENDON, t, dist, margin, m->tendon_invweight0[t]);
          if (r >= 0) for (int i = 0; i < nv; i++) d->efc_J[(size_t)r * nv + i] = -(real)side * d->ten_J[(size_t)t * nv + i];
        }
      }
    }
  }
  real* jac1 = d->wk;
  real* jac2 = d->wk + 3 * nv;
  for (int ci = 0; ci < d->ncon; ci++) {
    Contact* c = &d->con[ci];
    if (c->dist >= c->includemargin) continue;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    jac_point(m, d, jac1, c->pos, b1);
    jac_point(m, d, jac2, c->pos, b2);
    real* jf = d->wk + 6 * nv; /* 6 x nv, contact frame: 3 translational rows, then 3 rotational ones (condim 4 / 6) */
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < nv; i++)
        jf[r * nv + i] = c->frame[3 * r] * (jac2[i] - jac1[i]) + c->frame[3 * r + 1] * (jac2[nv + i] - jac1[nv + i]) +
                         c->frame[3 * r + 2] * (jac2[2 * nv + i] - jac1[2 * nv + i]);
    if (c->dim > 3) {
      real* jr1 = d->wk + 12 * nv;
      real* jr2 = d->wk + 15 * nv;
      jac_rot(m, d, jr1, b1);
      jac_rot(m, d, jr2, b2);
      for (int r = 0; r < 3; r++)
        for (int i = 0; i < nv; i++)
          jf[(3 + r) * nv + i] = c->frame[3 * r] * (jr2[i] - jr1[i]) + c->frame[3 * r + 1] * (jr2[nv + i] - jr1[nv + i]) +
                                 c->frame[3 * r + 2] * (jr2[2 * nv + i] - jr1[2 * nv + i]);
    }
    real tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    real rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
    c->mu = c->friction[0]; /* * sqrt(impratio) with impratio = 1 */
    if (c->dim == 1) {
      int r = add_row(d, nv, CT_CONTACT, ci, c->dist, c->includemargin, tran);
      if (r >= 0) memcpy(d->efc_J + (size_t)r * nv, jf, nv * sizeof(real));
    } else {
      /* pyramidal cone (mj_instantiateContact [3P]): 2 (dim - 1) rows J_normal +- friction[k-1] J_k; k = 1, 2 tangents, 3 torsion (spin
         about the normal), 4, 5 rolling; diagApprox = tran + fri^2 (tran | rot) */
      for (int k = 1; k < c->dim; k++) {
        real fri = c->friction[k - 1];
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          int r = add_row(d, nv, CT_CONTACT, ci, c->dist, c->includemargin, tran + fri * fri * (k < 3 ? tran : rot));
          if (r >= 0) for (int i = 0; i < nv; i++) d->efc_J[(size_t)r * nv + i] = jf[i] + sgn * fri * jf[k * nv + i];
        }
      }
    }
  }
}

static void get_solparams(const Model* m, const Data* d, int i, const real** solref, const real** solimp) {
  int id = d->efc_id[i];
  switch (d->efc_type[i]) {
    case CT_EQUALITY: *solref = m->eq_solref + 2 * id; *solimp = m->eq_solimp + 5 * id; break;
    case CT_LIMIT_JOINT: *solref = m->jnt_solref + 2 * id; *solimp = m->jnt_solimp + 5 * id; break;
    case CT_LIMIT_TENDON: *solref = m->tendon_solref + 2 * id; *solimp = m->tendon_solimp + 5 * id; break;
    case CT_FRICTION_DOF: *solref = m->dof_solref_fri + 2 * id; *solimp = m->dof_solimp_fri + 5 * id; break;
    default: *solref = d->con[id].solref; *solimp = d->con[id].solimp; break;
  }
}

static void reference_constraint(const Model* m, Data* d) { /* mj_makeImpedance + mj_referenceConstraint [3P] */
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    const real *solref, *solimp;
    get_solparams(m, d, i, &solref, &solimp);
    real imp;
    get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp);
    d->efc_R[i] = maxr(MINVAL, (1 - imp) / imp * d->efc_diagApprox[i]);
    real dmax = clipr(solimp[1], MINIMP, MAXIMP), K, B;
    if (solref[0] > 0) {
      real tc = maxr(solref[0], 2 * m->timestep), dr = solref[1];
      K = 1 / maxr(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / maxr(MINVAL, dmax * tc);
    } else {
      K = -solref[0] / maxr(MINVAL, dmax * dmax);
      B = -solref[1] / maxr(MINVAL, dmax);
    }
    real vel = 0;
    for (int k = 0; k < nv; k++) vel += d->efc_J[(size_t)i * nv + k] * d->qvel[k];
    d->efc_vel[i] = vel;
    d->efc_aref[i] = -B * vel - K * imp * (d->efc_pos[i] - d->efc_margin[i]);
  }
  /* pyramidal contacts: all edges of one contact share Rpy = 2 mu^2 R(first edge) */
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_type[i] == CT_CONTACT && d->con[d->efc_id[i]].dim > 1) {
      const Contact* c = &d->con[d->efc_id[i]];
      int nrow = 2 * (c->dim - 1);
      real Rpy = 2 * c->mu * c->mu * d->efc_R[i];
      for (int k = 0; k < nrow; k++) d->efc_R[i + k] = maxr(MINVAL, Rpy);
      i += nrow - 1;
    }
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage */
static void cross_motion(real* r, const real* vel, const real* v) {
  real a[3], b[3];
  cross3(r, vel, v);
  cross3(a, vel, v + 3);
  cross3(b, vel + 3, v);
  for (int k = 0; k < 3; k++) r[3 + k] = a[k] + b[k];
}
static void cross_force(real* r, const real* vel, const real* f) {
  real a[3], b[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  for (int k = 0; k < 3; k++) r[k] = a[k] + b[k];
  cross3(r + 3, vel, f + 3);
}

static void com_vel(const Model* m, Data* d) { /* mj_comVel [3P] */
  memset(d->cvel, 0, 6 * sizeof(real));
  for (int i = 1; i < m->nbody; i++) {
    real cvel[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[i], sizeof cvel);
    int da = m->body_dofadr[i];
    for (int j = m->body_jntadr[i]; j < m->body_jntadr[i] + m->body_jntnum[i]; j++) {
      int a = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        memset(d->cdof_dot + 6 * a, 0, 18 * sizeof(real));
        for (int k = 0; k < 3; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (a + k) + c] * d->qvel[a + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (a + k), cvel, d->cdof + 6 * (a + k));
        for (int k = 3; k < 6; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (a + k) + c] * d->qvel[a + k];
      } else {
        cross_motion(d->cdof_dot + 6 * a, cvel, d->cdof + 6 * a);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * a + c] * d->qvel[a];
      }
    }
    (void)da;
    memcpy(d->cvel + 6 * i, cvel, sizeof cvel);
  }
}

static void passive(const Model* m, Data* d) { /* mj_passive: joint springs + dampers, tendon springs + dampers */
  int nv = m->nv;
  if (m->disable_passive) { memset(d->qfrc_passive, 0, nv * sizeof(real)); return; }
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == JNT_HINGE || m->jnt_type[j] == JNT_SLIDE) {
      int qa = m->jnt_qposadr[j];
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
    }
  }
  for (int t = 0; t < m->ntendon; t++) {
    real f = -m->tendon_damping[t] * d->ten_velocity[t];
    /* tendon stiffness is zero in all config models; springlength handling intentionally omitted */
    if (f != 0) for (int i = 0; i < nv; i++) d->qfrc_passive[i] += f * d->ten_J[(size_t)t * nv + i];
  }
}

static void rne(const Model* m, Data* d) { /* mj_rne with flg_acc = 0 [3P] */
  int nb = m->nbody, nv = m->nv;
  real* cacc = d->wk;           /* 6*nb */
  real* cfrc = d->wk + 6 * nb;  /* 6*nb */
  cacc[0] = cacc[1] = cacc[2] = 0;
  for (int k = 0; k < 3; k++) cacc[3 + k] = m->disable_gravity ? 0 : -m->gravity[k];
  memset(cfrc, 0, 6 * sizeof(real));
  for (int i = 1; i < nb; i++) {
    memcpy(cacc + 6 * i, cacc + 6 * m->body_parentid[i], 6 * sizeof(real));
    for (int a = m->body_dofadr[i]; a >= 0 && a < m->body_dofadr[i] + m->body_dofnum[i]; a++)
      for (int c = 0; c < 6; c++) cacc[6 * i + c] += d->cdof_dot[6 * a + c] * d->qvel[a];
    real t[6], t1[6];
    mul_inert_vec(cfrc + 6 * i, d->cinert + 10 * i, cacc + 6 * i);
    mul_inert_vec(t, d->cinert + 10 * i, d->cvel + 6 * i);
    cross_force(t1, d->cvel + 6 * i, t);
    for (int c = 0; c < 6; c++) cfrc[6 * i + c] += t1[c];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * i + c];
  }
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int c = 0; c < 6; c++) s += d->cdof[6 * i + c] * cfrc[6 * m->dof_bodyid[i] + c];
    d->qfrc_bias[i] = s;
  }
}

/* ------------------------------------------------------------------ actuation: MuJoCo muscle model [3P] */
static real muscle_gain_length(real L, real lmin, real lmax) {
  if (lmin <= L && L <= lmax) {
    real a = (real)0.5 * (lmin + 1), b = (real)0.5 * (1 + lmax), x;
    if (L <= a) { x = (L - lmin) / maxr(MINVAL, a - lmin); return (real)0.5 * x * x; }
    else if (L <= 1) { x = (1 - L) / maxr(MINVAL, 1 - a); return 1 - (real)0.5 * x * x; }
    else if (L <= b) { x = (L - 1) / maxr(MINVAL, b - 1); return 1 - (real)0.5 * x * x; }
    else { x = (lmax - L) / maxr(MINVAL, lmax - b); return (real)0.5 * x * x; }
  }
  return 0;
}
static real muscle_gain(real len, real vel, const real* lr, real acc0, const real* prm) {
  real r0 = prm[0], r1 = prm[1], force = prm[2], scale = prm[3], lmin = prm[4], lmax = prm[5], vmax = prm[6], fvmax = prm[8];
  if (force < 0) force = scale / maxr(MINVAL, acc0);
  real L0 = (lr[1] - lr[0]) / maxr(MINVAL, r1 - r0);
  real L = r0 + (len - lr[0]) / maxr(MINVAL, L0);
  real V = vel / maxr(MINVAL, L0 * vmax);
  real FL = muscle_gain_length(L, lmin, lmax), FV, y = fvmax - 1;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1) * (V + 1);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / maxr(MINVAL, y);
  else FV = fvmax;
  return -force * FL * FV;
}
static real muscle_bias(real len, const real* lr, real acc0, const real* prm) {
  real r0 = prm[0], r1 = prm[1], force = prm[2], scale = prm[3], lmax = prm[5], fpmax = prm[7];
  if (force < 0) force = scale / maxr(MINVAL, acc0);
  real L0 = (lr[1] - lr[0]) / maxr(MINVAL, r1 - r0);
  real L = r0 + (len - lr[0]) / maxr(MINVAL, L0);
  real b = (real)0.5 * (1 + lmax), x;
  if (L <= 1) return 0;
  else if (L <= b) { x = (L - 1) / maxr(MINVAL, b - 1); return -force * fpmax * (real)0.5 * x * x; }
  else { x = (L - b) / maxr(MINVAL, b - 1); return -force * fpmax * ((real)0.5 + x); }
}
static real muscle_dynamics(real ctrl, real act, const real* prm) {
  real cc = clipr(ctrl, 0, 1), ac = clipr(act, 0, 1);
  real tau_act = prm[0] * ((real)0.5 + (real)1.5 * ac), tau_deact = prm[1] / ((real)0.5 + (real)1.5 * ac);
  real dctrl = cc - act, tau;
  if (prm[2] < MINVAL) tau = dctrl > 0 ? tau_act : tau_deact;
  else { /* smooth switch (tausmooth > 0): quintic sigmoid between the two time constants */
    real x = clipr(dctrl / prm[2] + (real)0.5, 0, 1);
    real s = x * x * x * (3 * x * (2 * x - 5) + 10);
    tau = tau_deact + (tau_act - tau_deact) * s;
  }
  return dctrl / maxr(MINVAL, tau);
}

static void transmission(const Model* m, Data* d) { /* mj_transmission */
  int nv = m->nv;
  for (int i = 0; i < m->nu; i++) {
    real g = m->actuator_gear[i];
    real* mom = d->actuator_moment + (size_t)i * nv;
    if (m->actuator_trntype[i] == 1) {
      int t = m->actuator_trnid[i];
      d->actuator_length[i] = g * d->ten_length[t];
      for (int k = 0; k < nv; k++) mom[k] = g * d->ten_J[(size_t)t * nv + k];
    } else {
      int j = m->actuator_trnid[i];
      memset(mom, 0, nv * sizeof(real));
      d->actuator_length[i] = g * d->qpos[m->jnt_qposadr[j]];
      mom[m->jnt_dofadr[j]] = g;
    }
  }
}

static void fwd_velocity(const Model* m, Data* d) {
  int nv = m->nv;
  for (int t = 0; t < m->ntendon; t++) {
    real s = 0;
    for (int k = 0; k < nv; k++) s += d->ten_J[(size_t)t * nv + k] * d->qvel[k];
    d->ten_velocity[t] = s;
  }
  for (int i = 0; i < m->nu; i++) {
    real s = 0;
    for (int k = 0; k < nv; k++) s += d->actuator_moment[(size_t)i * nv + k] * d->qvel[k];
    d->actuator_velocity[i] = s;
  }
  com_vel(m, d);
  passive(m, d);
  reference_constraint(m, d);
  rne(m, d);
}

static void fwd_actuation(const Model* m, Data* d) { /* mj_fwdActuation, muscle actuators */
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, nv * sizeof(real));
  if (m->disable_actuation) { memset(d->actuator_force, 0, m->nu * sizeof(real)); memset(d->act_dot, 0, m->na * sizeof(real)); return; }
  for (int i = 0; i < m->nu; i++) {
    real ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i]) ctrl = clipr(ctrl, m->actuator_ctrlrange[2 * i], m->actuator_ctrlrange[2 * i + 1]);
    if (m->actuator_kind[i] == 1) { /* stateless affine actuator (motor / position / velocity / general with dyntype none): [3P] mj_fwdActuation,
                                       gaintype fixed: gain = gainprm[0]; biastype affine: bias = b0 + b1 * length + b2 * velocity; force = gain * ctrl + bias.
                                       The (unused) activation slot stays at zero */
      const real* gp = m->actuator_gainprm + 9 * i;
      const real* bp = m->actuator_biasprm + 9 * i;
      real fa = gp[0] * ctrl + bp[0] + bp[1] * d->actuator_length[i] + bp[2] * d->actuator_velocity[i];
      if (m->actuator_forcelimited[i]) fa = clipr(fa, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
      d->act_dot[i] = 0;
      d->actuator_force[i] = fa;
      for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += d->actuator_moment[(size_t)i * nv + k] * fa;
      continue;
    }
    d->act_dot[i] = muscle_dynamics(ctrl, d->act[i], m->actuator_dynprm + 3 * i);
    real gain = muscle_gain(d->actuator_length[i], d->actuator_velocity[i], m->actuator_lengthrange + 2 * i,
                            m->actuator_acc0[i], m->actuator_gainprm + 9 * i);
    real bias = muscle_bias(d->actuator_length[i], m->actuator_lengthrange + 2 * i, m->actuator_acc0[i],
                            m->actuator_biasprm + 9 * i);
    real f = gain * d->act[i] + bias;
    if (m->actuator_forcelimited[i]) f = clipr(f, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
    d->actuator_force[i] = f;
    for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += d->actuator_moment[(size_t)i * nv + k] * f;
  }
}

static void fwd_acceleration(const Model* m, Data* d) {
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i] + d->qfrc_applied[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  solve_ld(m, d->qacc_smooth, d->qLD, d->qLDiagInv);
}

/* ------------------------------------------------------------------ Newton solver (primal, pyramidal) [3P] */
static real constraint_update(const Model* m, Data* d, const real* jar, int set_force) {
  real cost = 0;
  int nv = m->nv;
  if (set_force) memset(d->qfrc_constraint, 0, nv * sizeof(real));
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_type[i] == CT_FRICTION_DOF) {
      /* friction-loss row (PrimalUpdateConstraint [3P]): quadratic while |jar| < R f, linear beyond: the force saturates at +-f */
      real f = d->efc_frictionloss[i], rf = d->efc_R[i] * f, x = jar[i], force;
      int quad = 0;
      if (x <= -rf) { force = f; cost += f * (-(real)0.5 * rf - x); }
      else if (x >= rf) { force = -f; cost += f * (-(real)0.5 * rf + x); }
      else { force = -d->efc_D[i] * x; cost += (real)0.5 * d->efc_D[i] * x * x; quad = 1; }
      if (set_force) {
        d->efc_state[i] = quad;        /* only the quadratic zone enters the Hessian */
        d->efc_force[i] = force;
        for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)i * nv + k] * force;
      }
      continue;
    }
    int active = d->efc_type[i] == CT_EQUALITY || jar[i] < 0;
    if (active) cost += (real)0.5 * d->efc_D[i] * jar[i] * jar[i];
    if (set_force) {
      d->efc_state[i] = active;
      d->efc_force[i] = active ? -d->efc_D[i] * jar[i] : 0;
      if (active) for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
    }
  }
  return cost;
}

/* developer statistics (not thread safe; single-threaded probes only): evaluations per exact line search */
#ifndef MYOO_LS_NOISE
#define MYOO_LS_NOISE 0
#endif
#ifndef MYOO_NEWTON_NOISE
#define MYOO_NEWTON_NOISE 0
#endif
#ifndef MYOO_GRAD_NOISE
#define MYOO_GRAD_NOISE 0
#endif
static long g_ls_hist[64];
static int g_ls_last;
void myoo_ls_hist(long* out, int reset) { for (int k = 0; k < 64; k++) { out[k] = g_ls_hist[k]; if (reset) g_ls_hist[k] = 0; } }

typedef struct { real g1, g2; const real *jar, *jv, *D, *R, *floss; const int* type; int nefc; } LSctx;
static real g_ls_mag;   /* magnitude of the terms that cancel in d1 at the last evaluation (round-off scale of the float build) */
static void ls_eval(const LSctx* c, real alpha, real* d1, real* d2) {
  real a = c->g1 + 2 * alpha * c->g2, b = 2 * c->g2, p = 0;
  for (int i = 0; i < c->nefc; i++) {
    real x = c->jar[i] + alpha * c->jv[i];
    if (c->type[i] == CT_FRICTION_DOF) {
      real f = c->floss[i], rf = c->R[i] * f;
      if (x <= -rf) p -= f * c->jv[i];
      else if (x >= rf) p += f * c->jv[i];
      else { p += c->D[i] * x * c->jv[i]; b += c->D[i] * c->jv[i] * c->jv[i]; }
      continue;
    }
    if (c->type[i] == CT_EQUALITY || x < 0) { p += c->D[i] * x * c->jv[i]; b += c->D[i] * c->jv[i] * c->jv[i]; }
  }
  g_ls_mag = fabs(c->g1) + fabs(2 * alpha * c->g2) + fabs(p);
  *d1 = a + p; *d2 = b;
}

static int cholesky(real* H, int n) { /* in place lower Cholesky, row-major; returns 0 on success */
  for (int j = 0; j < n; j++) {
    real s = H[j * n + j];
    for (int k = 0; k < j; k++) s -= H[j * n + k] * H[j * n + k];
    if (s < MINVAL) s = MINVAL;
    s = sqrt(s);
    H[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      real t = H[i * n + j];
      for (int k = 0; k < j; k++) t -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = t / s;
    }
  }
  return 0;
}
static void chol_solve(const real* L, int n, real* x) {
  for (int i = 0; i < n; i++) { real s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { real s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

static void fwd_constraint(const Model* m, Data* d) { /* mj_fwdConstraint + mj_solNewton [3P] */
  int nv = m->nv, nefc = d->nefc;
  d->solver_iter = 0; d->solver_improvement = 0; d->solver_gradient = 0;
  if (!nefc) {
    memcpy(d->qacc, d->qacc_smooth, nv * sizeof(real));
    memcpy(d->qacc_warmstart, d->qacc_smooth, nv * sizeof(real));
    memset(d->qfrc_constraint, 0, nv * sizeof(real));
    return;
  }
  real* Ma = d->wk;            real* jar = Ma + nv;          real* grad = jar + nefc;
  real* search = grad + nv;    real* Mv = search + nv;       real* jv = Mv + nv;
  real* H = jv + nefc;         real* tmpv = H + nv * nv;
  /* efc_b = J*qacc_smooth - aref */
  for (int i = 0; i < nefc; i++) {
    real s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k];
    d->efc_b[i] = s - d->efc_aref[i];
  }
  /* warmstart: better of qacc_warmstart and qacc_smooth */
  for (int i = 0; i < nefc; i++) {
    real s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
    jar[i] = s - d->efc_aref[i];
  }
  real cost_warm = constraint_update(m, d, jar, 0);
  mul_m(m, d, Ma, d->qacc_warmstart);
  for (int k = 0; k < nv; k++) cost_warm += (real)0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc_warmstart[k] - d->qacc_smooth[k]);
  real cost_smooth = constraint_update(m, d, d->efc_b, 0);
  if (cost_warm > cost_smooth) memcpy(d->qacc, d->qacc_smooth, nv * sizeof(real));
  else memcpy(d->qacc, d->qacc_warmstart, nv * sizeof(real));
  /* init */
  mul_m(m, d, Ma, d->qacc);
  for (int i = 0; i < nefc; i++) {
    real s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc[k];
    jar[i] = s - d->efc_aref[i];
  }
  real scale = 1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  real cost = constraint_update(m, d, jar, 1);
  for (int k = 0; k < nv; k++) cost += (real)0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
  int iter = 0;
  while (iter < m->iterations) {
    /* gradient and Newton direction */
    for (int k = 0; k < nv; k++) grad[k] = Ma[k] - d->qfrc_smooth[k] - d->qfrc_constraint[k];
    memset(H, 0, (size_t)nv * nv * sizeof(real));
    for (int i = 0; i < nv; i++) {
      int a = m->dof_Madr[i], j = i;
      while (j >= 0) { H[i * nv + j] = d->qM[a]; H[j * nv + i] = d->qM[a]; a++; j = m->dof_parentid[j]; }
    }
    for (int r = 0; r < nefc; r++) {
      if (!d->efc_state[r]) continue;
      const real* Jr = d->efc_J + (size_t)r * nv;
      for (int i = 0; i < nv; i++) {
        if (Jr[i] == 0) continue;
        real t = d->efc_D[r] * Jr[i];
        for (int j = 0; j <= i; j++) H[i * nv + j] += t * Jr[j];
      }
    }
    cholesky(H, nv);
    for (int k = 0; k < nv; k++) search[k] = -grad[k];
    chol_solve(H, nv, search);
    /* exact line search along `search` */
    real snorm = 0;
    for (int k = 0; k < nv; k++) snorm += search[k] * search[k];
    snorm = sqrt(snorm);
    if (snorm < MINVAL) break;
    mul_m(m, d, Mv, search);
    for (int i = 0; i < nefc; i++) {
      real s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * search[k];
      jv[i] = s;
    }
    LSctx ls;
    ls.g1 = 0; ls.g2 = 0;
    for (int k = 0; k < nv; k++) { ls.g1 += search[k] * (Ma[k] - d->qfrc_smooth[k]); ls.g2 += (real)0.5 * search[k] * Mv[k]; }
    ls.jar = jar; ls.jv = jv; ls.D = d->efc_D; ls.R = d->efc_R; ls.floss = d->efc_frictionloss; ls.type = d->efc_type; ls.nefc = nefc;
    real gtol = m->tolerance * m->ls_tolerance * snorm / scale;
    real lo = 0, hi = -1, dlo, d2lo, dhi = 0, d2hi = 0, alpha = 0, d1, d2;
    ls_eval(&ls, 0, &dlo, &d2lo);
    if (dlo >= 0 || d2lo <= 0) break; /* not a descent direction: converged to round-off */
    alpha = -dlo / d2lo;
    g_ls_last = 0;
    for (int lsit = 0; lsit < m->ls_iterations; lsit++) {
      ls_eval(&ls, alpha, &d1, &d2);
      g_ls_last = lsit + 1;
#ifdef MYOO_FLOAT
      if (fabs(d1) < gtol || fabs(d1) < MYOO_LS_NOISE * g_ls_mag) break;   /* slope below the float round-off of its own terms */
#else
      if (fabs(d1) < gtol) break;
#endif
      if (d1 < 0) { lo = alpha; dlo = d1; d2lo = d2; } else { hi = alpha; dhi = d1; d2hi = d2; }
      real cand = alpha - d1 / d2;
      if (hi < 0) { /* not bracketed yet: keep doing one-sided Newton steps to the right */
        if (!(cand > lo)) break;
        alpha = cand;
      } else {
        if (!(cand > lo && cand < hi)) { /* Newton from the other end, else bisect */
          real c2 = d1 < 0 ? hi - dhi / d2hi : lo - dlo / d2lo;
          cand = (c2 > lo && c2 < hi) ? c2 : (real)0.5 * (lo + hi);
        }
        if (cand == alpha || hi - lo <= (real)1e-15 * hi) break;
        alpha = cand;
      }
    }
    g_ls_hist[g_ls_last < 63 ? g_ls_last : 63]++;
    if (alpha <= 0) break;
    for (int k = 0; k < nv; k++) { d->qacc[k] += alpha * search[k]; Ma[k] += alpha * Mv[k]; }
    for (int i = 0; i < nefc; i++) jar[i] += alpha * jv[i];
    real oldcost = cost;
    cost = constraint_update(m, d, jar, 1);
    for (int k = 0; k < nv; k++) cost += (real)0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    real gn = 0, gmag = 0;
    for (int k = 0; k < nv; k++) {
      real g = Ma[k] - d->qfrc_smooth[k] - d->qfrc_constraint[k]; gn += g * g;
      real t = fabs(Ma[k]) + fabs(d->qfrc_smooth[k]) + fabs(d->qfrc_constraint[k]); gmag += t * t;   /* round-off scale of the gradient */
    }
    d->solver_improvement = scale * (oldcost - cost);
    d->solver_gradient = scale * sqrt(gn);
    iter++;
#ifdef MYOO_FLOAT
    if (d->solver_improvement < maxr(m->tolerance, MYOO_NEWTON_NOISE * scale * fabs(cost)) ||
        d->solver_gradient < maxr(m->tolerance, MYOO_GRAD_NOISE * scale * sqrt(gmag))) break;
#else
    if (d->solver_improvement < m->tolerance || d->solver_gradient < m->tolerance) break;
#endif
  }
  (void)tmpv;
  d->solver_iter = iter;
  memcpy(d->qacc_warmstart, d->qacc, nv * sizeof(real));
}

/* ------------------------------------------------------------------ drivers */
static int is_bad(const real* x, int n) {
  for (int i = 0; i < n; i++) if (!(x[i] == x[i]) || x[i] > MAXVAL || x[i] < -MAXVAL) return 1;
  return 0;
}

void myoo_fwd_position(const Model* m, Data* d) {
  kinematics(m, d);
  com_pos(m, d);
  tendon(m, d);
  crb(m, d);
  factor_m(m, d->qM, d->qLD, d->qLDiagInv);
  collision(m, d);
  make_constraint(m, d);
  transmission(m, d);
}

void myoo_forward(const Model* m, Data* d) { /* mj_forward */
  myoo_fwd_position(m, d);
  fwd_velocity(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  fwd_constraint(m, d);
}

static void euler(const Model* m, Data* d) { /* mj_Euler with implicit joint damping [3P] */
  int nv = m->nv;
  real dt = m->timestep;
  real* qacc = d->wk;
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) { damped = 1; break; }
  if (!damped) {
    memcpy(qacc, d->qacc, nv * sizeof(real));
  } else {
    real* qH = d->wk + nv;
    real* hinv = qH + m->nM;
    real* qHf = hinv + nv;
    memcpy(qH, d->qM, m->nM * sizeof(real));
    for (int i = 0; i < nv; i++) qH[m->dof_Madr[i]] += dt * m->dof_damping[i];
    factor_m(m, qH, qHf, hinv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    solve_ld(m, qacc, qHf, hinv);
  }
  for (int i = 0; i < m->na; i++) d->act[i] += dt * d->act_dot[i];
  for (int i = 0; i < nv; i++) d->qvel[i] += dt * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      real w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real ang = norm3(w) * dt;
      if (ang > MINVAL) {
        real n = norm3(w), s = sin(ang * (real)0.5);
        real dq[4] = {cos(ang * (real)0.5), w[0] / n * s, w[1] / n * s, w[2] / n * s};
        mul_quat(d->qpos + qa + 3, d->qpos + qa + 3, dq);
        normalize4(d->qpos + qa + 3);
      }
    } else {
      d->qpos[qa] += dt * d->qvel[da];
    }
  }
  d->time += dt;
}

/* mj_integratePos [3P]: qpos advanced by velocity `vel` over dt (hinge / slide: linear; free joint: position linear, quaternion by the
 * body-frame angular velocity) */
static void integrate_pos(const Model* m, real* qpos, const real* vel, real dt) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += dt * vel[da + k];
      real w[3] = {vel[da + 3], vel[da + 4], vel[da + 5]};
      real n = norm3(w), ang = n * dt;
      if (ang > MINVAL) {
        real s = sin(ang * (real)0.5);
        real dq[4] = {cos(ang * (real)0.5), w[0] / n * s, w[1] / n * s, w[2] / n * s};
        mul_quat(qpos + qa + 3, qpos + qa + 3, dq);
        normalize4(qpos + qa + 3);
      }
    } else {
      qpos[qa] += dt * vel[da];
    }
  }
}

/* mj_RungeKutta(m, d, 4) [3P]: classic RK4 on the state (qpos, qvel, act) with derivative (qvel, qacc, act_dot); the first derivative is
 * the forward pass mj_step has just done, each further stage sets the state to X0 + h a F[i-1] (a = 1/2, 1/2, 1), the time to
 * t0 + c h and runs the forward pass again; the result is X0 + h (F0 + 2 F1 + 2 F2 + F3) / 6.  No implicit joint damping (Euler only). */
static int runge_kutta4(const Model* m, Data* d) {
  int nq = m->nq, nv = m->nv, na = m->na;
  static const real A[3] = {(real)0.5, (real)0.5, (real)1}, Bw[4] = {(real)1 / 6, (real)1 / 3, (real)1 / 3, (real)1 / 6};
  real h = m->timestep, t0 = d->time;
  real* X0 = (real*)malloc((size_t)(nq + nv + na + 2 * nv + na + 8) * sizeof(real));
  real* Sv = X0 + nq + nv + na;     /* sum B_j qvel_j */
  real* Sa = Sv + nv;               /* sum B_j qacc_j */
  real* Sd = Sa + nv;               /* sum B_j act_dot_j */
  memcpy(X0, d->qpos, nq * sizeof(real)); memcpy(X0 + nq, d->qvel, nv * sizeof(real)); memcpy(X0 + nq + nv, d->act, na * sizeof(real));
  for (int k = 0; k < nv; k++) { Sv[k] = Bw[0] * d->qvel[k]; Sa[k] = Bw[0] * d->qacc[k]; }
  for (int k = 0; k < na; k++) Sd[k] = Bw[0] * d->act_dot[k];
  int rc = 0;
  for (int i = 1; i < 4 && !rc; i++) {
    real a = A[i - 1];
    real* dv = d->wk + 20 * nv;      /* a * F[i-1].qvel: velocity that advances qpos */
    for (int k = 0; k < nv; k++) dv[k] = a * d->qvel[k];
    for (int k = 0; k < nv; k++) d->qvel[k] = X0[nq + k] + h * a * d->qacc[k];
    for (int k = 0; k < na; k++) d->act[k] = X0[nq + nv + k] + h * a * d->act_dot[k];
    memcpy(d->qpos, X0, nq * sizeof(real));
    integrate_pos(m, d->qpos, dv, h);
    d->time = t0 + a * h;
    myoo_forward(m, d);
    if (d->warning & 4) { rc = 4; break; }
    if (is_bad(d->qacc, nv)) { rc = 2; break; }
    for (int k = 0; k < nv; k++) { Sv[k] += Bw[i] * d->qvel[k]; Sa[k] += Bw[i] * d->qacc[k]; }
    for (int k = 0; k < na; k++) Sd[k] += Bw[i] * d->act_dot[k];
  }
  if (!rc) {
    memcpy(d->qpos, X0, nq * sizeof(real));
    integrate_pos(m, d->qpos, Sv, h);
    for (int k = 0; k < nv; k++) d->qvel[k] = X0[nq + k] + h * Sa[k];
    for (int k = 0; k < na; k++) d->act[k] = X0[nq + nv + k] + h * Sd[k];
    d->time = t0 + h;
  }
  free(X0);
  return rc;
}

/* one mj_step; returns nonzero if the state went bad and was reset (mj_sim_scene.py:54-61) */
int myoo_step1(const Model* m, Data* d) {
  if (is_bad(d->qpos, m->nq) || is_bad(d->qvel, m->nv)) { myoo_reset(m, d); d->warning |= 1; return 1; }
  myoo_forward(m, d);
  if (d->warning & 4) return 4;
  if (is_bad(d->qacc, m->nv)) { myoo_reset(m, d); d->warning |= 2; return 2; }
  if (m->integrator == 1) {
    int rc = runge_kutta4(m, d);
    if (rc == 2) { myoo_reset(m, d); d->warning |= 2; }
    return rc;
  }
  euler(m, d);
  return 0;
}

int myoo_step(const Model* m, Data* d, int nsub) {
  for (int s = 0; s < nsub; s++) {
    int r = myoo_step1(m, d);
    if (r) return r;
  }
  return 0;
}

/* ------------------------------------------------------------------ muscle length ranges: mj_setLengthRange / evalAct [3P]
 * MuJoCo's compiler computes `lengthrange` by simulating the model with contacts, passive forces, gravity and actuation disabled while a
 * force of fixed acceleration magnitude pulls along the actuator's moment arm (side 0: shorten, side 1: lengthen), velocities damped by
 * exp(-dt / timeconst) per step; the range is the min (max) length over the last `interval` seconds of `inttotal`.  Defaults (mjLROpt):
 * accel 20, maxforce 0, timeconst 1, timestep 0.01, inttotal 10, interval 2, tolrange 0.05.  The reference's model files store the numbers
 * MuJoCo produced this way (myohand_assets.xml:501-539, myolegs_assets.xml:606-685): restating the procedure on top of this oracle's own
 * position / velocity / constraint / Euler stages turns every stored pair into a golden vector for those stages (tests/test_oracle.py).
 * out = {range lo, range hi, spread of side 0 over the interval, spread of side 1}; returns 0, or 1 if the state went bad. */
int myoo_lengthrange(Model* m, Data* d, int actuator, double accel, double maxforce, double timeconst, double timestep,
                     double inttotal, double interval, double* out) {
  int nv = m->nv, rc = 0;
  real save_dt = m->timestep;
  int sc = m->disable_contact, sp = m->disable_passive, sg = m->disable_gravity, sa = m->disable_actuation;
  m->timestep = (real)timestep;
  m->disable_contact = m->disable_passive = m->disable_gravity = m->disable_actuation = 1;
  real* moment = (real*)calloc(nv, sizeof(real));
  real* tmp = (real*)calloc(nv, sizeof(real));
  for (int side = 0; side < 2 && !rc; side++) {
    myoo_reset(m, d);
    real lmin = 0, lmax = 0;
    int updated = 0;
    while (d->time < inttotal) {
      myoo_fwd_position(m, d);
      fwd_velocity(m, d);
      memcpy(moment, d->actuator_moment + (size_t)actuator * nv, nv * sizeof(real));
      memcpy(tmp, moment, nv * sizeof(real));
      solve_ld(m, tmp, d->qLD, d->qLDiagInv);
      real nrm = 0;
      for (int i = 0; i < nv; i++) nrm += tmp[i] * moment[i];
      nrm = sqrt(nrm);
      real scl = (real)(2 * side - 1) * (real)accel / maxr(MINVAL, nrm);
      for (int i = 0; i < nv; i++) d->qfrc_applied[i] = scl * moment[i];
      if (maxforce > 0) {
        real fn = 0;
        for (int i = 0; i < nv; i++) fn += d->qfrc_applied[i] * d->qfrc_applied[i];
        fn = sqrt(fn);
        if (fn > maxforce) for (int i = 0; i < nv; i++) d->qfrc_applied[i] *= (real)maxforce / fn;
      }
      fwd_actuation(m, d);
      fwd_acceleration(m, d);
      fwd_constraint(m, d);
      if (is_bad(d->qacc, nv)) { rc = 1; break; }
      euler(m, d);
      real damp = exp(-(real)timestep / maxr((real)0.01, (real)timeconst));
      for (int i = 0; i < nv; i++) d->qvel[i] *= damp;
      real len = d->actuator_length[actuator];
      if (d->time > inttotal - interval) {
        if (len < lmin || !updated) lmin = len;
        if (len > lmax || !updated) lmax = len;
        updated = 1;
      }
    }
    out[side] = (double)(side == 0 ? lmin : lmax);
    out[2 + side] = (double)(lmax - lmin);
  }
  free(moment); free(tmp);
  m->timestep = save_dt;
  m->disable_contact = sc; m->disable_passive = sp; m->disable_gravity = sg; m->disable_actuation = sa;
  myoo_reset(m, d);
  return rc;
}

/* ------------------------------------------------------------------ field access for tests */
#define FIELD(nm, cnt) if (!strcmp(name, #nm)) { *n = (cnt); return d->nm; }
real* myoo_field(const Model* m, Data* d, const char* name, int* n) {
  int nv = m->nv, nb = m->nbody;
  FIELD(qpos, m->nq) FIELD(qvel, nv) FIELD(act, m->na) FIELD(ctrl, m->nu) FIELD(qacc_warmstart, nv)
  FIELD(xpos, 3 * nb) FIELD(xquat, 4 * nb) FIELD(xmat, 9 * nb) FIELD(xipos, 3 * nb) FIELD(ximat, 9 * nb)
  FIELD(xanchor, 3 * m->njnt) FIELD(xaxis, 3 * m->njnt) FIELD(geom_xpos, 3 * m->ngeom) FIELD(geom_xmat, 9 * m->ngeom)
  FIELD(site_xpos, 3 * m->nsite) FIELD(subtree_com, 3 * nb) FIELD(cdof, 6 * nv) FIELD(cinert, 10 * nb)
  FIELD(ten_length, m->ntendon) FIELD(ten_J, m->ntendon * nv) FIELD(qM, m->nM) FIELD(qLD, m->nM)
  FIELD(actuator_length, m->nu) FIELD(actuator_moment, m->nu * nv) FIELD(actuator_velocity, m->nu)
  FIELD(actuator_force, m->nu) FIELD(act_dot, m->na) FIELD(qfrc_bias, nv) FIELD(qfrc_passive, nv)
  FIELD(qfrc_actuator, nv) FIELD(qfrc_smooth, nv) FIELD(qacc_smooth, nv) FIELD(qfrc_constraint, nv) FIELD(qacc, nv)
  FIELD(cvel, 6 * nb) FIELD(efc_J, d->nefc * nv) FIELD(efc_pos, d->nefc) FIELD(efc_aref, d->nefc)
  FIELD(efc_D, d->nefc) FIELD(efc_R, d->nefc) FIELD(efc_force, d->nefc) FIELD(efc_vel, d->nefc)
  *n = 0;
  return NULL;
}
double myoo_get_time(const Data* d) { return (double)d->time; }
void myoo_set_time(Data* d, double t) { d->time = (real)t; }
int myoo_nefc(const Data* d) { return d->nefc; }
int myoo_ncon(const Data* d) { return d->ncon; }
int myoo_solver_iter(const Data* d) { return d->solver_iter; }
int myoo_warning(const Data* d) { return d->warning; }
void myoo_contact(const Data* d, int i, double* out /* dist,pos3,normal3,geom1,geom2 */) {
  const Contact* c = &d->con[i];
  out[0] = (double)c->dist;
  for (int k = 0; k < 3; k++) { out[1 + k] = (double)c->pos[k]; out[4 + k] = (double)c->frame[k]; }
  out[7] = c->geom1; out[8] = c->geom2;
}
/* dense mass matrix for tests */
void myoo_full_m(const Model* m, const Data* d, double* out) {
  int nv = m->nv;
  for (int i = 0; i < nv * nv; i++) out[i] = 0;
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i], j = i;
    while (j >= 0) { out[i * nv + j] = out[j * nv + i] = (double)d->qM[a++]; j = m->dof_parentid[j]; }
  }
}
/* kinetic + potential energy (for the conservation invariant) */
void myoo_energy(const Model* m, Data* d, double* out) {
  int nv = m->nv;
  real* Mv = d->wk;
  mul_m(m, d, Mv, d->qvel);
  double ke = 0, pe = 0;
  for (int i = 0; i < nv; i++) ke += 0.5 * (double)(Mv[i] * d->qvel[i]);
  for (int b = 1; b < m->nbody; b++)
    for (int k = 0; k < 3; k++) pe -= (double)(m->body_mass[b] * m->gravity[k] * d->xipos[3 * b + k]);
  out[0] = ke; out[1] = pe;
}

/* ------------------------------------------------------------------ threaded batch driver = CPU baseline */
typedef struct {
  const Model* m; int lo, hi, nsub;
  double *qpos, *qvel, *act, *warm, *time; const double* ctrl; int* flags;
} Job;

static void* batch_worker(void* arg) {
  Job* j = (Job*)arg;
  const Model* m = j->m;
  Data* d = myoo_make_data(m);
  for (int e = j->lo; e < j->hi; e++) {
    for (int k = 0; k < m->nq; k++) d->qpos[k] = (real)j->qpos[(size_t)e * m->nq + k];
    for (int k = 0; k < m->nv; k++) d->qvel[k] = (real)j->qvel[(size_t)e * m->nv + k];
    for (int k = 0; k < m->na; k++) d->act[k] = (real)j->act[(size_t)e * m->na + k];
    for (int k = 0; k < m->nv; k++) d->qacc_warmstart[k] = (real)j->warm[(size_t)e * m->nv + k];
    for (int k = 0; k < m->nu; k++) d->ctrl[k] = (real)j->ctrl[(size_t)e * m->nu + k];
    d->time = (real)j->time[e];
    d->warning = 0;
    myoo_step(m, d, j->nsub);
    for (int k = 0; k < m->nq; k++) j->qpos[(size_t)e * m->nq + k] = (double)d->qpos[k];
    for (int k = 0; k < m->nv; k++) j->qvel[(size_t)e * m->nv + k] = (double)d->qvel[k];
    for (int k = 0; k < m->na; k++) j->act[(size_t)e * m->na + k] = (double)d->act[k];
    for (int k = 0; k < m->nv; k++) j->warm[(size_t)e * m->nv + k] = (double)d->qacc_warmstart[k];
    j->time[e] = (double)d->time;
    if (j->flags) j->flags[e] = d->warning;
  }
  myoo_free_data(d);
  return NULL;
}

/* env-major double arrays [B][n]; steps every env nsub substeps with its own ctrl row */
void myoo_step_batch(const Model* m, int B, double* qpos, double* qvel, double* act, double* warm, double* time,
                     const double* ctrl, int nsub, int nthreads, int* flags) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  Job jobs[256];
  for (int t = 0; t < nthreads; t++) {
    Job j = {m, (int)((long)B * t / nthreads), (int)((long)B * (t + 1) / nthreads), nsub, qpos, qvel, act, warm, time, ctrl, flags};
    jobs[t] = j;
    pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

#if defined(MYOO_COUNT_FLOPS)
/* counters of the calling thread: add, mul, div, sqrt, special; reset = 1 clears them after the read */
void myoo_flops(uint64_t* out, int reset) {
  out[0] = g_flops.add; out[1] = g_flops.mul; out[2] = g_flops.div; out[3] = g_flops.sqrt_; out[4] = g_flops.special;
  if (reset) memset(&g_flops, 0, sizeof(g_flops));
}
}  /* extern "C" */
#endif
