/* myo_oracle_abi.c -- the C ABI of include/myo_hip.h on top of the float64 CPU oracle ("device = -1", SURVEY.md 8b last sentence).
 *
 * TEST INFRASTRUCTURE, like everything under oracle/: only tests/ load this library (tests/abi_backend.py), so that a test written against the
 * C ABI can run on either backend.  The product (myosuite_mjx_amd/capi.py) never loads it and has no CPU path.
 *
 * What is implemented is the physics surface of the ABI -- model load (the same MYOB blob), batch create, state fields, set_state, reset to
 * qpos0, myo_step with MYO_ACTMAP_NONE / MYO_ACTMAP_MUSCLE_SIGMOID, per-env fault flags with reset (mj_sim_scene.py:54-61), status --
 * with the library's conventions: 0 / negative error code, never throws, myo_last_error(), env-major float32 rows, "device" pointers are host
 * pointers here and the stream argument is ignored (every call is synchronous).  Every other entry point of the header is exported too and
 * returns MYO_E_UNSUPPORTED with a message: tasks (observations / rewards / task resets), conditions, the TrackEnv task, policies and the
 * bench / profiling calls exist on the GPU side only (their CPU checkers are the numpy / torch statements in tests/). */
#include "../include/myo_hip.h"
#include "myo_oracle.c"
#include <stdio.h>

struct myo_model { Model* m; };
struct myo_batch { const struct myo_model* model; int B; Data** d; float* ctrl; float* qacc; float* tenlen; float* actforce; int32_t* flags; int32_t* diag; };

static _Thread_local char g_err[256] = "";
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
static int unsupported(const char* what) { snprintf(g_err, sizeof g_err, "oracle backend (device -1): %s is not implemented on the CPU twin of the ABI", what); return MYO_E_UNSUPPORTED; }

const char* myo_last_error(void) { return g_err; }
int myo_version(void) { return 3; }

int myo_model_load(const void* blob, size_t nbytes, int device, myo_model** out) {
  if (!blob || !out) return fail(MYO_E_ARG, "myo_model_load: null argument");
  if (device != -1) return fail(MYO_E_UNSUPPORTED, "oracle backend: device must be -1");
  Model* m = myoo_load(blob, nbytes);
  if (!m) return fail(MYO_E_BLOB, "oracle backend: bad model blob");
  struct myo_model* h = (struct myo_model*)calloc(1, sizeof *h);
  if (!h) { myoo_free(m); return fail(MYO_E_NOMEM, "out of memory"); }
  h->m = m;
  *out = h;
  return MYO_OK;
}
void myo_model_free(myo_model* h) { if (h) { myoo_free(h->m); free(h); } }
int myo_model_dims(const myo_model* h, myo_dims* out) {
  if (!h || !out) return fail(MYO_E_ARG, "myo_model_dims: null argument");
  const Model* m = h->m;
  memset(out, 0, sizeof *out);
  out->nq = m->nq; out->nv = m->nv; out->nu = m->nu; out->na = m->na; out->nbody = m->nbody; out->ntendon = m->ntendon; out->nsite = m->nsite;
  out->ncon_max = NCON_MAX; out->timestep = (float)m->timestep;
  return MYO_OK;
}
int myo_model_set_switch(myo_model* h, int disable_contact, int disable_limit, int disable_ellipsoid) {
  if (!h) return fail(MYO_E_ARG, "myo_model_set_switch: null model");
  myoo_set_switch(h->m, disable_contact, disable_limit, disable_ellipsoid);
  return MYO_OK;
}

int myo_batch_create(const myo_model* h, int B, myo_batch** out) {
  if (!h || !out || B <= 0) return fail(MYO_E_ARG, "myo_batch_create: bad argument");
  const Model* m = h->m;
  struct myo_batch* b = (struct myo_batch*)calloc(1, sizeof *b);
  if (!b) return fail(MYO_E_NOMEM, "out of memory");
  b->model = h; b->B = B;
  b->d = (Data**)calloc((size_t)B, sizeof(Data*));
  b->ctrl = (float*)calloc((size_t)B * m->nu + 1, sizeof(float));
  b->qacc = (float*)calloc((size_t)B * m->nv + 1, sizeof(float));
  b->tenlen = (float*)calloc((size_t)B * m->nu + 1, sizeof(float));
  b->actforce = (float*)calloc((size_t)B * m->nu + 1, sizeof(float));
  b->flags = (int32_t*)calloc((size_t)B, sizeof(int32_t));
  b->diag = (int32_t*)calloc((size_t)B * 8, sizeof(int32_t));
  int ok = b->d && b->ctrl && b->qacc && b->tenlen && b->actforce && b->flags && b->diag;
  for (int e = 0; ok && e < B; e++) { b->d[e] = myoo_make_data(m); ok = b->d[e] != NULL; if (ok) myoo_reset(m, b->d[e]); }
  if (!ok) { myo_batch_free(b); return fail(MYO_E_NOMEM, "out of memory"); }
  *out = b;
  return MYO_OK;
}
void myo_batch_free(myo_batch* b) {
  if (!b) return;
  if (b->d) { for (int e = 0; e < b->B; e++) if (b->d[e]) myoo_free_data(b->d[e]); free(b->d); }
  free(b->ctrl); free(b->qacc); free(b->tenlen); free(b->actforce); free(b->flags); free(b->diag); free(b);
}
int myo_batch_size(const myo_batch* b) { return b ? b->B : 0; }

/* field -> (pointer to env e's real row | float row, width); is_real: the row lives in the oracle's Data in `real` */
static int field_row(myo_batch* b, int field, int e, real** rrow, float** frow, int32_t** irow, int* width) {
  const Model* m = b->model->m;
  Data* d = b->d[e];
  *rrow = NULL; *frow = NULL; *irow = NULL;
  switch (field) {
    case MYO_F_QPOS: *rrow = d->qpos; *width = m->nq; return 0;
    case MYO_F_QVEL: *rrow = d->qvel; *width = m->nv; return 0;
    case MYO_F_ACT: *rrow = d->act; *width = m->na; return 0;            /* (all-muscle models: na == nu) */
    case MYO_F_WARMSTART: *rrow = d->qacc_warmstart; *width = m->nv; return 0;
    case MYO_F_TIME: *rrow = &d->time; *width = 1; return 0;
    case MYO_F_CTRL: *frow = b->ctrl + (size_t)e * m->nu; *width = m->nu; return 0;
    case MYO_F_QACC: *frow = b->qacc + (size_t)e * m->nv; *width = m->nv; return 0;
    case MYO_F_TENLEN: *frow = b->tenlen + (size_t)e * m->nu; *width = m->nu; return 0;
    case MYO_F_ACTFORCE: *frow = b->actforce + (size_t)e * m->nu; *width = m->nu; return 0;
    case MYO_F_FLAGS: *irow = b->flags + e; *width = 1; return 0;
    case MYO_F_DIAG: *irow = b->diag + (size_t)e * 8; *width = 8; return 0;
    default: return -1;
  }
}
int myo_batch_read(myo_batch* b, int field, void* host, size_t nbytes) {
  if (!b || !host) return fail(MYO_E_ARG, "myo_batch_read: null argument");
  for (int e = 0; e < b->B; e++) {
    real* rr; float* fr; int32_t* ir; int w;
    if (field_row(b, field, e, &rr, &fr, &ir, &w)) return unsupported("this field");
    if (nbytes != (size_t)b->B * w * 4) return fail(MYO_E_ARG, "myo_batch_read: size mismatch");
    float* o = (float*)host + (size_t)e * w;
    if (rr) for (int k = 0; k < w; k++) o[k] = (float)rr[k];
    else if (fr) memcpy(o, fr, (size_t)w * 4);
    else memcpy(o, ir, (size_t)w * 4);
  }
  return MYO_OK;
}
int myo_batch_write(myo_batch* b, int field, const void* host, size_t nbytes) {
  if (!b || !host) return fail(MYO_E_ARG, "myo_batch_write: null argument");
  for (int e = 0; e < b->B; e++) {
    real* rr; float* fr; int32_t* ir; int w;
    if (field_row(b, field, e, &rr, &fr, &ir, &w)) return unsupported("this field");
    if (nbytes != (size_t)b->B * w * 4) return fail(MYO_E_ARG, "myo_batch_write: size mismatch");
    const float* s = (const float*)host + (size_t)e * w;
    if (rr) for (int k = 0; k < w; k++) rr[k] = (real)s[k];
    else if (fr) memcpy(fr, s, (size_t)w * 4);
    else memcpy(ir, s, (size_t)w * 4);
  }
  return MYO_OK;
}
int myo_batch_field(myo_batch* b, int field, void** dev_ptr, size_t* pitch, size_t* width) {
  (void)b; (void)field; (void)dev_ptr; (void)pitch; (void)width;
  return unsupported("myo_batch_field (zero-copy device rows; use myo_batch_read / myo_batch_write)");
}

int myo_set_state(myo_batch* b, const float* qpos, const float* qvel, const float* act, const float* time, void* stream) {
  (void)stream;
  if (!b) return fail(MYO_E_ARG, "myo_set_state: null batch");
  const Model* m = b->model->m;
  for (int e = 0; e < b->B; e++) {
    Data* d = b->d[e];
    if (qpos) for (int k = 0; k < m->nq; k++) d->qpos[k] = (real)qpos[(size_t)e * m->nq + k];
    if (qvel) for (int k = 0; k < m->nv; k++) d->qvel[k] = (real)qvel[(size_t)e * m->nv + k];
    if (act) for (int k = 0; k < m->na; k++) d->act[k] = (real)act[(size_t)e * m->nu + k];
    if (time) d->time = (real)time[e];
  }
  return MYO_OK;
}
int myo_reset(myo_batch* b, const uint8_t* mask, uint64_t seed, void* stream) {
  (void)seed; (void)stream;
  if (!b) return fail(MYO_E_ARG, "myo_reset: null batch");
  const Model* m = b->model->m;
  for (int e = 0; e < b->B; e++) {
    if (mask && !mask[e]) continue;
    myoo_reset(m, b->d[e]);                                   /* mj_resetData: qpos0, zero velocity / activation / warm start / time */
    memset(b->ctrl + (size_t)e * m->nu, 0, (size_t)m->nu * 4);
  }
  return MYO_OK;
}
int myo_step(myo_batch* b, const float* action, int actmap, int nsubsteps, void* stream) {
  (void)stream;
  if (!b || nsubsteps < 0) return fail(MYO_E_ARG, "myo_step: bad argument");
  const Model* m = b->model->m;
  if (actmap != MYO_ACTMAP_NONE && actmap != MYO_ACTMAP_MUSCLE_SIGMOID) return unsupported("this action map");
  if (actmap == MYO_ACTMAP_MUSCLE_SIGMOID && m->na != m->nu) return unsupported("the sigmoid action map on a model with stateless actuators");
  for (int e = 0; e < b->B; e++) {
    Data* d = b->d[e];
    float* c = b->ctrl + (size_t)e * m->nu;
    if (action) for (int i = 0; i < m->nu; i++) {
      float a = action[(size_t)e * m->nu + i];
      c[i] = actmap == MYO_ACTMAP_MUSCLE_SIGMOID ? 1.0f / (1.0f + expf(-5.0f * (a - 0.5f))) : a;      /* base_v0.py:87-91 */
    }
    for (int i = 0; i < m->nu; i++) d->ctrl[i] = (real)c[i];
    int itmax = 0;
    for (int s = 0; s < nsubsteps; s++) {
      int r = myoo_step1(m, d);
      if (r == 1) b->flags[e] |= MYO_FLAG_BAD_STATE;
      if (r == 2) b->flags[e] |= MYO_FLAG_BAD_QACC;
      if (r == 4) b->flags[e] |= MYO_FLAG_CONTACT_OVERFLOW;
      if (d->solver_iter > itmax) itmax = d->solver_iter;
      if (r) break;
    }
    for (int k = 0; k < m->nv; k++) b->qacc[(size_t)e * m->nv + k] = (float)d->qacc[k];
    for (int i = 0; i < m->nu; i++) { b->tenlen[(size_t)e * m->nu + i] = (float)d->actuator_length[i]; b->actforce[(size_t)e * m->nu + i] = (float)d->actuator_force[i]; }
    b->diag[(size_t)e * 8 + 0] = d->nefc; b->diag[(size_t)e * 8 + 1] = d->ncon; b->diag[(size_t)e * 8 + 2] = itmax;
  }
  return MYO_OK;
}
int myo_status(myo_batch* b, int32_t* host_flags) {
  if (!b || !host_flags) return fail(MYO_E_ARG, "myo_status: null argument");
  memcpy(host_flags, b->flags, (size_t)b->B * 4);
  memset(b->flags, 0, (size_t)b->B * 4);
  return MYO_OK;
}
int myo_sync(void* stream) { (void)stream; return MYO_OK; }
int myo_set_env_offset(myo_batch* b, int env_offset) { (void)b; (void)env_offset; return MYO_OK; }   /* (only keys the task RNG, which lives on the GPU side) */
int myo_set_balance(myo_batch* b, int on) { (void)b; (void)on; return MYO_OK; }                         /* speed-only hints: nothing to do */
int myo_set_lanes(int lanes) { (void)lanes; return MYO_OK; }

/* ---- the rest of the header: exported, not implemented on the CPU twin */
int myo_batch_configure(myo_batch* b, const myo_task_config* c) { (void)b; (void)c; return unsupported("myo_batch_configure (tasks)"); }
int myo_batch_configure_walk(myo_batch* b, const myo_walk_config* c) { (void)b; (void)c; return unsupported("myo_batch_configure_walk"); }
int myo_batch_configure_track(myo_batch* b, const myo_track_config* c) { (void)b; (void)c; return unsupported("myo_batch_configure_track"); }
int myo_batch_set_condition(myo_batch* b, int f, int a, int c) { (void)b; (void)f; (void)a; (void)c; return unsupported("myo_batch_set_condition"); }
int myo_batch_set_fatigue_reset(myo_batch* b, int mode, const float* v) { (void)b; (void)mode; (void)v; return unsupported("myo_batch_set_fatigue_reset"); }
int myo_batch_set_geom_override(myo_batch* b, int g, const float* lo, const float* hi) { (void)b; (void)g; (void)lo; (void)hi; return unsupported("myo_batch_set_geom_override"); }
int myo_obs(myo_batch* b, void* s) { (void)b; (void)s; return unsupported("myo_obs"); }
int myo_obs_only(myo_batch* b, void* s) { (void)b; (void)s; return unsupported("myo_obs_only"); }
int myo_obs_reset_only(myo_batch* b, void* s) { (void)b; (void)s; return unsupported("myo_obs_reset_only"); }
int myo_autoreset(myo_batch* b, int n, uint64_t seed, void* s) { (void)b; (void)n; (void)seed; (void)s; return unsupported("myo_autoreset"); }
int myo_random_action(myo_batch* b, float* a, uint64_t seed, uint64_t step, int off, void* s) { (void)b; (void)a; (void)seed; (void)step; (void)off; (void)s; return unsupported("myo_random_action"); }
int myo_read_stamps(myo_batch* b, long long* h, int n) { (void)b; (void)h; (void)n; return unsupported("myo_read_stamps"); }
int myo_bench_rollout(myo_batch* b, int steps, int nsub, uint64_t seed, int mode, int mx, void* s, float* ms) { (void)b; (void)steps; (void)nsub; (void)seed; (void)mode; (void)mx; (void)s; (void)ms; return unsupported("myo_bench_rollout"); }
int myo_bench_last_kernel_ms(myo_batch* b, float* ms) { (void)b; (void)ms; return unsupported("myo_bench_last_kernel_ms"); }
int myo_probe_valu(int device, int w, int it, double* r, int* n) { (void)device; (void)w; (void)it; (void)r; (void)n; return unsupported("myo_probe_valu"); }
const char* myo_bench_last_kernel_name(const myo_batch* b) { (void)b; return "oracle (CPU)"; }
int myo_policy_load(int device, int obs_dim, int act_dim, int nlayers, const int* layer_out, const float* obs_mean, const float* obs_std, const float* const* kernels,
                    const float* const* biases, myo_policy** out) {
  (void)device; (void)obs_dim; (void)act_dim; (void)nlayers; (void)layer_out; (void)obs_mean; (void)obs_std; (void)kernels; (void)biases; (void)out;
  return unsupported("myo_policy_load");
}
void myo_policy_free(myo_policy* p) { (void)p; }
int myo_policy_act(myo_policy* p, const float* obs, int B, float* act, int det, uint64_t seed, uint64_t step, int off, void* s) {
  (void)p; (void)obs; (void)B; (void)act; (void)det; (void)seed; (void)step; (void)off; (void)s;
  return unsupported("myo_policy_act");
}
