// myo_hip.hip -- MI355X (gfx950) batched musculoskeletal stepper: kernels + C ABI (include/myo_hip.h).
//
// One translation unit, split over csrc/ for readability:
//   myo_common.h        limits, device-side structs (model tables, per-env batch arrays, task records), vector math
//   myo_physics.h       tendon wrapping, muscle model, action map (+ muscle conditions), impedance, contact frames, MPR
//   myo_task_track.h    MyoDM TrackEnv as a fused task of the TRK step kernel: reference lookup, reward / done, masked reset
//   myo_kernel_lanes.h  step_kernel<G>: the first kernel, G = 16 / 32 / 64 lanes per env, all state in LDS (cross-check / fallback)
//   myo_kernel_wave.h   step_kernel_w: one env per 64-lane wavefront (default), substep scheduler, size-specialised instantiations
//   myo_kernels_aux.h   RNG, placement hint, random actions, policy inference, reset, observation kernels
//   myo_hip.hip         host side: model upload, batches, launches, the extern "C" entry points
//
// Execution model: one environment = one wavefront = one workgroup; the whole working set of an env (link
// frames, sparse tendon Jacobian rows, spatial inertias, mass matrix, contact rows, Newton vectors) lives in that wave's LDS slice
// and registers for all substeps of an env step; HBM is read once (state + action) and written once (state, observation).
// Model constants are read through the scalar / L1 caches from the DevModel tables produced by myosuite_mjx_amd/lowering.py.
//
// Physics restated per substep (reference: third-party MuJoCo reached at myosuite/physics/mj_sim_scene.py:55; algorithms per
// MuJoCo documentation [3P]): kinematics -> spatial tendons w/ wrapping -> muscle FLV forces -> CRB mass matrix + RNE bias ->
// collision -> limit / equality / contact rows -> Newton solver -> semi-implicit Euler with implicit joint damping.
#include "myo_common.h"
#include "myo_physics.h"
#include "myo_task_track.h"
#include "myo_kernel_lanes.h"
#include "myo_kernel_wave.h"
#include "myo_kernels_aux.h"

// ================================================================================================
// host side
// ================================================================================================
static int g_lanes = 64;  // lanes per env (16 / 32 / 64); MYO_LANES env var or myo_set_lanes(); 64 = wave-per-env kernel
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MYO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct BlobRec { char name[32]; uint32_t dtype, ndim, shape[4]; uint64_t nbytes, offset; };

struct myo_model {
  int device = 0;
  DevModel dm{};
  DevModelW dw{};
  DevModel* d_dm = nullptr;     // device copies of the model structs for the wave kernel
  DevModelW* d_dw = nullptr;
  int env_lds_bytes_w = 0;
  bool wave_ok = false, generic_ok = false;
  int n_cu = 0;                 // compute units of the model's device (scheduler sizing)
  int kin_floats = 0;           // LDS scratch the two-phase kinematics needs (lowering.py hip_kin_size)
  bool rk4 = false;             // <option integrator="RK4">: the RK4 instantiations of the wave kernel (generic sizes, no scheduler)
  bool trk = false;             // TrackEnv model class: step_kernel_w<36,20,32,2,2,false,0,false,true>
  bool hand_sizes = false, leg_sizes = false, terrain_sizes = false;   // table sizes equal Sizes<1> / Sizes<2>: the size-specialised instantiations may be used
  int wave_cfg = 0;             // 0: step_kernel_w<24,8,32,1,4> (hand / finger), 1: step_kernel_w<36,20,48,2,2> (legs)
  int nq = 0;
  int has_tl = 0;
  bool has_affine = false;   // some actuator is a stateless affine one (motor / position / velocity): wave kernel only
  myo_dims dims{};
  std::vector<void*> dev_allocs;
  std::vector<float> qpos0, jnt_lo, jnt_hi;
  std::vector<int> cg_geom, cg_type_h;   // collision geom -> compiled geom id, and its type (host copies)
  std::vector<int> body_link;                       // body -> link, pose of the body inside the link frame (walk task)
  std::vector<float> body_lpos, body_lquat, mass;   // mass = [total, static bodies' mass-weighted COM xyz]
  float* d_qpos0 = nullptr;
  int env_lds_bytes = 0;
};

struct myo_batch {
  const myo_model* model = nullptr;
  DevBatch db{};
  TaskDev task{};
  int ntarget_alloc = 0, obs_alloc = 0, env_offset = 0;
  std::vector<void*> dev_allocs;
  float *d_tlo = nullptr, *d_thi = nullptr, *d_init = nullptr, *d_jlo = nullptr, *d_jhi = nullptr, *d_action = nullptr, *d_rnd = nullptr;
  float* d_initv = nullptr;
  float *d_init2 = nullptr, *d_initv2 = nullptr, *d_fatvec = nullptr;   // walk reset "random": second keyframe; fatigue reset vector
  const char* last_kernel = "step_kernel";   // name of the step-kernel instantiation of the last myo_step / bench launch
  DevWalk* d_walk = nullptr;
  DevTrack* d_track = nullptr;    // MYO_TASK_TRACK: device copy of the task record (tables hang off it)
  float* d_metrics = nullptr;     // [B][4]
  int track_frames = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint64_t bench_step = 0;
  long long* d_stamps = nullptr;
  int* d_order = nullptr;
  int* d_sched = nullptr;       // substep scheduler: 8 queues x (4 control words + ring)
  int sched_stride = 0;
  int balance = 1;
  std::vector<hipEvent_t> kev;   // per-launch event pairs around the step kernel (bench only)
  int kev_pending = 0;           // pairs recorded by asynchronous bench calls and not collected yet
  float last_kernel_ms = 0.f;
};

static const BlobRec* blob_find(const uint8_t* blob, const char* name) {
  uint32_t n;
  memcpy(&n, blob + 8, 4);
  for (uint32_t i = 0; i < n; i++) {
    const BlobRec* r = (const BlobRec*)(blob + 16 + (size_t)i * sizeof(BlobRec));
    if (!strncmp(r->name, name, 32)) return r;
  }
  return nullptr;
}

template <typename T> static int upload(myo_model* m, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t nb = (v.size() + 4) * sizeof(T);
  HIPCHK(hipMalloc(&p, nb));
  HIPCHK(hipMemset(p, 0, nb));
  if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  m->dev_allocs.push_back(p);
  *out = (const T*)p;
  return 0;
}
// (out is a template so that the struct fields' device-side address-space types do not matter to this host code)
template <class P> static int load_f(myo_model* m, const uint8_t* blob, const char* name, P* out, std::vector<float>* keep = nullptr) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 0) return fail(MYO_E_BLOB, std::string("model blob lacks f64 array ") + name);
  size_t n = r->nbytes / 8;
  std::vector<float> v(n);
  const double* src = (const double*)(blob + r->offset);
  for (size_t i = 0; i < n; i++) v[i] = (float)src[i];
  if (keep) *keep = v;
  const float* tmp = nullptr;
  int rc = upload<float>(m, v, &tmp);
  *out = (P)tmp;
  return rc;
}
template <class P> static int load_i(myo_model* m, const uint8_t* blob, const char* name, P* out, std::vector<int>* keep = nullptr) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 1) return fail(MYO_E_BLOB, std::string("model blob lacks i32 array ") + name);
  size_t n = r->nbytes / 4;
  std::vector<int> v(n);
  memcpy(v.data(), blob + r->offset, n * 4);
  if (keep) *keep = v;
  const int* tmp = nullptr;
  int rc = upload<int>(m, v, &tmp);
  *out = (P)tmp;
  return rc;
}

static void build_layout_w(const DevModel& d, DevModelW& w, int nvt, int kc, int nc, int nj = 3) {
  w.lay = layout_w(w.nq, d.nv, d.nu, d.nl, d.ngt, d.maxnnz, d.ncg, w.has_tl != 0, nvt, kc, nc, nj);   // (myo_kernel_wave.h: shared with the compile-time layouts)
}

static void build_layout(DevModel& d) {
  Lay& Y = d.lay;
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  int nv = d.nv, nu = d.nu, nl = d.nl, ntri = nv * (nv + 1) / 2;
  Y.qpos = take(nv); Y.qvel = take(nv); Y.act = take(nu); Y.ctrl = take(nu); Y.warm = take(nv);
  Y.lpos = take(3 * nl); Y.lmat = take(9 * nl); Y.lquat = take(4 * nl); Y.axis = take(3 * nv); Y.anchor = take(3 * nv);
  Y.tJ = take(d.ngt * d.maxnnz); Y.tlen = take(d.ngt); Y.tforce = take(nu); Y.actdot = take(nu);
  Y.qfa = take(nv); Y.smooth = take(nv); Y.qas = take(nv); Y.qacc = take(nv); Y.Ma = take(nv); Y.grad = take(nv);
  Y.search = take(nv); Y.Mv = take(nv); Y.qfc = take(nv);
  Y.Mp = take(ntri); Y.Hp = take(ntri);
  Y.lsign = take(nv); Y.laref = take(nv); Y.lD = take(nv); Y.ljar = take(nv); Y.ljv = take(nv);
  // region A (spatial dynamics) is dead once Mp / smooth exist; region B (collision + contact rows) aliases it
  int regA = o;
  Y.cdof = take(6 * nv); Y.cinert = take(10 * nl); Y.crb = take(10 * nl); Y.cvel = take(6 * nl); Y.cacc = take(6 * nl); Y.cfrc = take(6 * nl);
  int endA = o;
  o = regA;
  Y.gpos = take(3 * d.ncg); Y.gmat = take(9 * d.ncg); Y.cand = take(NCAND);
  Y.cdist = take(NCON); Y.cpos = take(3 * NCON); Y.cnrm = take(3 * NCON); Y.cpair = take(NCON); Y.cJ = take(NCON * 3 * KCMAX);
  Y.caref = take(4 * NCON); Y.cD = take(NCON); Y.cjar = take(4 * NCON); Y.cjv = take(4 * NCON); Y.cimp = take(NCON);
  if (o < endA) o = endA;
  // pad so that the 64/G env slices start on different LDS banks
  o = (o + 31) / 32 * 32 + 8;
  Y.total = o;
}

extern "C" {

const char* myo_last_error(void) { return g_err.c_str(); }
int myo_version(void) { return 1; }

int myo_model_load(const void* blobv, size_t nbytes, int device, myo_model** out) {
  if (!blobv || !out || nbytes < 16) return fail(MYO_E_ARG, "myo_model_load: bad arguments");
  const uint8_t* blob = (const uint8_t*)blobv;
  uint32_t ver;
  memcpy(&ver, blob + 4, 4);
  if (memcmp(blob, "MYOB", 4) || ver != 3) return fail(MYO_E_BLOB, "myo_model_load: not a MYOB v3 blob");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MYO_E_HIP, "myo_model_load: no such HIP device (a GPU is required; there is no CPU fallback)");
  HIPCHK(hipSetDevice(device));
  myo_model* m = new myo_model();
  m->device = device;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->n_cu = prop.multiProcessorCount; }
  DevModel& d = m->dm;
  const BlobRec* hs = blob_find(blob, "hip_sizes");
  const BlobRec* sz = blob_find(blob, "sizes");
  const BlobRec* op = blob_find(blob, "opt");
  if (!hs || !sz || !op) { delete m; return fail(MYO_E_BLOB, "myo_model_load: blob lacks hip_* tables (run lowering)"); }
  const int* H = (const int*)(blob + hs->offset);
  const int* S = (const int*)(blob + sz->offset);
  const double* O = (const double*)(blob + op->offset);
  d.nl = H[0]; d.nlevel = H[1]; d.nv = H[2]; d.nu = H[3]; d.ngt = H[4]; d.nseg = H[5]; d.maxnnz = H[7]; d.nwg = H[8];
  d.ncg = H[9]; d.npair = H[10]; d.maxkc = H[11]; d.ns = H[12]; d.nM = S[11];
  m->nq = S[0];
  if (d.ngt != d.nu) { delete m; return fail(MYO_E_UNSUPPORTED, "limited-only tendons are not supported by the HIP path yet"); }
  d.timestep = (float)O[0]; d.grav[0] = (float)O[1]; d.grav[1] = (float)O[2]; d.grav[2] = (float)O[3];
  d.tolerance = (float)O[4]; d.iterations = (int)O[5]; d.ls_iterations = (int)O[6]; d.ls_tolerance = (float)O[7];
  d.meaninertia = (float)O[9];
  d.newton_scale = 1.0f / (d.meaninertia * (float)(d.nv > 1 ? d.nv : 1));
  int rc = 0;
  std::vector<float> c0, jl;
  const float* tmp;
#define LF(field, name) if ((rc = load_f(m, blob, name, &d.field))) { myo_model_free(m); return rc; }
#define LI(field, name) if ((rc = load_i(m, blob, name, &d.field))) { myo_model_free(m); return rc; }
  LI(level_adr, "hip_level_adr") LI(link_parent, "hip_link_parent") LI(link_dofadr, "hip_link_dofadr") LI(link_dofnum, "hip_link_dofnum")
  LI(child_adr, "hip_child_adr") LI(child, "hip_child") LI(dof_link, "hip_dof_link") LI(dof_type, "hip_dof_type")
  LI(dof_parent, "dof_parentid") LI(site_link, "hip_site_link") LI(wg_link, "hip_wg_link") LI(gt_seg_adr, "hip_gt_seg_adr")
  LI(gt_seg_num, "hip_gt_seg_num") LI(gt_dofs, "hip_gt_dofs") LI(seg, "hip_seg") LI(dl, "hip_dl") LI(col_adr, "hip_col_adr")
  LI(col, "hip_col") LI(cg_link, "hip_cg_link") LI(cg_type, "hip_cg_type") LI(pair_i, "hip_pair_i") LI(pair_dl, "hip_pair_dl")
  LF(link_pos, "hip_link_pos") LF(link_quat, "hip_link_quat") LF(link_mass, "hip_link_mass") LF(link_com, "hip_link_com")
  LF(link_inertia, "hip_link_inertia") LF(dof_pos, "hip_dof_pos") LF(dof_axis, "hip_dof_axis") LF(dof_damping, "dof_damping")
  LF(dof_armature, "dof_armature") LF(site_lpos, "hip_site_lpos") LF(wg_lpos, "hip_wg_lpos") LF(wg_lmat, "hip_wg_lmat")
  LF(wg_radius, "hip_wg_radius") LF(seg_div, "hip_seg_div") LF(gt_len0, "hip_gt_len0") LF(act, "hip_act") LF(cg_lpos, "hip_cg_lpos") LF(cg_lmat, "hip_cg_lmat")
  LF(cg_size, "hip_cg_size") LF(cg_rbound, "hip_cg_rbound") LF(pair_f, "hip_pair_f")
#undef LF
#undef LI
  if ((rc = load_f(m, blob, "qpos0", &d.qpos0, &m->qpos0))) { myo_model_free(m); return rc; }
  if ((rc = load_f(m, blob, "hip_jl", &d.jl, &jl))) { myo_model_free(m); return rc; }
  if ((rc = load_f(m, blob, "hip_c0", &tmp, &c0))) { myo_model_free(m); return rc; }
  d.c0[0] = c0[0]; d.c0[1] = c0[1]; d.c0[2] = c0[2];
  {
    std::vector<float> org;
    if ((rc = load_f(m, blob, "hip_origin", &tmp, &org))) { myo_model_free(m); return rc; }
    d.origin[0] = org[0]; d.origin[1] = org[1]; d.origin[2] = org[2];
  }
  m->jnt_lo.resize(d.nv); m->jnt_hi.resize(d.nv);
  for (int i = 0; i < d.nv; i++) { m->jnt_lo[i] = jl[12 * i + 1]; m->jnt_hi[i] = jl[12 * i + 2]; }
  build_layout(d);
  m->env_lds_bytes = d.lay.total * 4;
  {  // wave-per-env kernel tables (one env per wavefront, nv <= 24)
    DevModelW& w = m->dw;
    std::vector<int> nws;
    const int* tmpi;
    if ((rc = load_i(m, blob, "hip_seg_order", &w.seg_order)) || (rc = load_i(m, blob, "hip_seg_tendon", &w.seg_tendon)) ||
        (rc = load_i(m, blob, "hip_gt_dl", &w.gt_dl)) || (rc = load_f(m, blob, "hip_link_mat0", &w.link_mat0)) ||
        (rc = load_i(m, blob, "hip_nwrapseg", &tmpi, &nws))) { myo_model_free(m); return rc; }
    w.nwrapseg = nws[0];
    w.ndl = H[6];
    {
      std::vector<float> tlv;
      if ((rc = load_f(m, blob, "hip_tl", &w.tl, &tlv))) { myo_model_free(m); return rc; }
      w.has_tl = 0;
      for (int t = 0; t < d.ngt; t++) if (tlv[12 * t] != 0) w.has_tl = 1;
      m->has_tl = w.has_tl;
    }
    bool plane_pairs = false, condim1 = false;
    {
      std::vector<int> fl, pi;
      if ((rc = load_i(m, blob, "hip_flags", &tmpi, &fl)) || (rc = load_i(m, blob, "hip_link_free", &w.link_free)) ||
          (rc = load_i(m, blob, "hip_dof_qposadr", &w.dof_qposadr)) || (rc = load_i(m, blob, "hip_link_chain_adr", &w.link_chain_adr)) ||
          (rc = load_i(m, blob, "hip_link_chain", &w.link_chain)) || (rc = load_i(m, blob, "hip_eq_i", &w.eq_i)) ||
          (rc = load_i(m, blob, "hip_kin_base", &w.kin_base)) || (rc = load_i(m, blob, "hip_kin_adr", &w.kin_adr)) || (rc = load_i(m, blob, "hip_kin_vec", &w.kin_vec)) ||
          (rc = load_f(m, blob, "hip_eq_f", &w.eq_f)) || (rc = load_i(m, blob, "hip_pair_i", &tmpi, &pi))) { myo_model_free(m); return rc; }
      {
        std::vector<int> ks;
        if ((rc = load_i(m, blob, "hip_kin_size", &tmpi, &ks))) { myo_model_free(m); return rc; }
        m->kin_floats = ks[0];   // checked against the Hessian scratch (nvt x (nvt + 1)) once the kernel class is known
        w.kin_dnmax = ks.size() > 1 ? ks[1] : 6;
      }
      w.has_free = fl[0]; w.nq = fl[1]; w.neq = fl[2];
      if (fl.size() < 6) { myo_model_free(m); return fail(MYO_E_BLOB, "hip_flags: blob predates the actuator-kind tables; recompile the model"); }
      w.has_j0 = fl[3]; d.na_obs = fl[4]; m->has_affine = fl[5] != 0;
      { const int* t2; if ((rc = load_i(m, blob, "hip_cg_geom", &t2, &m->cg_geom)) || (rc = load_i(m, blob, "hip_cg_type", &t2, &m->cg_type_h))) { myo_model_free(m); return rc; } }
      {
        std::vector<int> hi; std::vector<float> hf; const float* tf2;
        if ((rc = load_i(m, blob, "hip_hf_i", &tmpi, &hi)) || (rc = load_f(m, blob, "hip_hf_f", &tf2, &hf))) { myo_model_free(m); return rc; }
        w.hf.on = hi[0]; w.hf.nrow = hi[1]; w.hf.ncol = hi[2]; w.hf.cg = hi[3];
        for (int k = 0; k < 4; k++) w.hf.size[k] = hf[k];
        for (int k = 0; k < 3; k++) w.hf.pos[k] = hf[4 + k];
        if (w.hf.on && (w.hf.nrow > 128 || w.hf.ncol > 100 || d.npair > 1023)) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "height field: at most 128 x 100 cells and 1023 pairs"); }
      }
      if ((rc = load_f(m, blob, "hip_gt_j0", &w.gt_j0)) || (rc = load_i(m, blob, "hip_act_obs", &d.act_obs))) { myo_model_free(m); return rc; }
      const float* tf;
      if ((rc = load_i(m, blob, "hip_body_link", &tmpi, &m->body_link)) || (rc = load_f(m, blob, "hip_body_lpos", &tf, &m->body_lpos)) ||
          (rc = load_f(m, blob, "hip_body_lquat", &tf, &m->body_lquat)) || (rc = load_f(m, blob, "hip_mass", &tf, &m->mass))) { myo_model_free(m); return rc; }
      if (w.nq != m->nq) { myo_model_free(m); return fail(MYO_E_BLOB, "hip_flags disagrees with sizes"); }
      for (int p = 0; p < d.npair; p++) { if (pi[6 * p + 4] >= 2) plane_pairs = true; if (pi[6 * p + 5] == 1) condim1 = true; }
      // the 16/32-lane generic kernel covers fixed-base models with hinge / slide joints and capsule / convex pairs only
      m->generic_ok = !w.has_free && w.neq == 0 && !plane_pairs && !condim1 && w.nq == d.nv && d.maxkc <= KCMAX && !w.has_tl && !m->has_affine;
    }
    // TrackEnv model class (lowering: hip_trk = condim-4 pairs | friction loss | box / hull geoms): tables of the TRK instantiation
    w.fl = nullptr; w.mesh_vert = nullptr; w.mesh_rec = nullptr; w.mesh_startrec = nullptr; w.mesh_aabb = nullptr;
    m->trk = false;
    if (blob_find(blob, "hip_trk")) {
      std::vector<int> tk;
      if ((rc = load_i(m, blob, "hip_trk", &tmpi, &tk)) || (rc = load_f(m, blob, "hip_fl", &w.fl)) || (rc = load_f(m, blob, "hip_mesh_vert", &w.mesh_vert)) ||
          (rc = load_f(m, blob, "hip_mesh_rec", &w.mesh_rec)) || (rc = load_f(m, blob, "hip_mesh_startrec", &w.mesh_startrec)) || (rc = load_f(m, blob, "hip_mesh_aabb", &w.mesh_aabb))) { myo_model_free(m); return rc; }
      m->trk = tk[0] || tk[1] || tk[2];
    }
    {  // self-contained per-lane records (DevModelW::seg_rec, dl_pk, ...): denormalised copies of the tables loaded above
      auto BI = [&](const char* n) { std::vector<int> v; const BlobRec* r = blob_find(blob, n); if (r && r->dtype == 1) { v.resize(r->nbytes / 4); memcpy(v.data(), blob + r->offset, r->nbytes); } return v; };
      auto BF = [&](const char* n) { std::vector<float> v; const BlobRec* r = blob_find(blob, n); if (r && r->dtype == 0) { v.resize(r->nbytes / 8); const double* sp = (const double*)(blob + r->offset); for (size_t i = 0; i < v.size(); i++) v[i] = (float)sp[i]; } return v; };
      const std::vector<int> seg = BI("hip_seg"), seg_order = BI("hip_seg_order"), seg_tendon = BI("hip_seg_tendon"), site_link = BI("hip_site_link"), wg_link = BI("hip_wg_link"),
                             dl = BI("hip_dl"), dof_type = BI("hip_dof_type");
      const std::vector<float> seg_div = BF("hip_seg_div"), site_lpos = BF("hip_site_lpos"), wg_lpos = BF("hip_wg_lpos"), wg_lmat = BF("hip_wg_lmat"), wg_radius = BF("hip_wg_radius");
      auto fi = [](int v) { float f; memcpy(&f, &v, 4); return f; };
      std::vector<int> dlp(std::max<size_t>(dl.size() / 3, 1), 0);
      for (size_t i = 0; i < dl.size() / 3; i++) {
        const int dd = dl[3 * i], sg = dl[3 * i + 1], slot = dl[3 * i + 2];
        if (dd < 0 || dd > 127 || slot < 0 || slot > 255 || sg < -32768 || sg > 32767) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "moment-arm list entry does not fit the packed word"); }
        dlp[i] = dd | ((dof_type[dd] == 3 ? 1 : 0) << 7) | (slot << 8) | (int)((unsigned)sg << 16);
      }
      // (each segment's three lists are copied to 16-byte rows of their own: a lane reads a list four entries per load, the first row ahead of its use)
      std::vector<int> dl4;
      std::vector<float> rec((size_t)std::max(d.nseg, 1) * SEGR * 4, 0.f);
      for (int idx = 0; idx < d.nseg; idx++) {
        const int si = seg_order[idx];
        const int* S = &seg[12 * (size_t)si];
        float* R = &rec[(size_t)idx * SEGR * 4];
        for (int k = 0; k < 2; k++) { R[4 * k] = fi(site_link[S[k]]); for (int c = 0; c < 3; c++) R[4 * k + 1 + c] = site_lpos[3 * (size_t)S[k] + c]; }
        R[8] = fi(S[2]); R[9] = fi(S[3] >= 0 ? site_link[S[3]] : -2); R[10] = 1.0f / seg_div[si]; R[11] = fi(seg_tendon[si]);
        for (int k = 0; k < 3; k++) {
          const int a0 = S[4 + 2 * k], n = S[5 + 2 * k], adr4 = (int)(dl4.size() / 4);
          if (adr4 >= (1 << 20) || n < 0 || n >= (1 << 11) || (n > 0 && (a0 < 0 || (size_t)(a0 + n) > dl.size() / 3))) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "tendon moment-arm lists too long for the packed segment record"); }
          for (int i = 0; i < n; i++) dl4.push_back(dlp[a0 + i]);
          while (dl4.size() % 4 || dl4.size() == (size_t)adr4 * 4) dl4.push_back(0);     // whole rows; an empty list still owns one (the lane loads it unconditionally)
          R[12 + k] = fi(adr4 | (n << 20));
        }
        R[15] = fi(S[10]);
        if (S[2] >= 0) {
          const int g = S[2];
          if (S[3] >= 0) for (int c = 0; c < 3; c++) R[16 + c] = site_lpos[3 * (size_t)S[3] + c];
          R[19] = wg_radius[g];
          R[20] = fi(wg_link[g]); for (int c = 0; c < 3; c++) R[21 + c] = wg_lpos[3 * (size_t)g + c];
          for (int c = 0; c < 9; c++) R[24 + c] = wg_lmat[9 * (size_t)g + c];
        }
      }
      const float* tf4 = nullptr; const int* ti4 = nullptr;
      if (dl4.empty()) dl4.resize(4, 0);
      if ((rc = upload<float>(m, rec, &tf4)) || (rc = upload<int>(m, dl4, &ti4))) { myo_model_free(m); return rc; }
      w.seg_rec = (decltype(w.seg_rec))tf4; w.dl_pk = (decltype(w.dl_pk))ti4;
      // collision geoms, pairs and the pairs' dof lists
      const std::vector<int> cg_link = BI("hip_cg_link"), cg_type = BI("hip_cg_type"), pair_i = BI("hip_pair_i"), pair_dl = BI("hip_pair_dl");
      const std::vector<float> cg_lpos = BF("hip_cg_lpos"), cg_lmat = BF("hip_cg_lmat"), cg_size = BF("hip_cg_size"), cg_rb = BF("hip_cg_rbound"), pair_f = BF("hip_pair_f");
      std::vector<float> grec((size_t)std::max(d.ncg, 1) * 16, 0.f), prec((size_t)std::max(d.npair, 1) * 16, 0.f);
      for (int g = 0; g < d.ncg; g++) {
        float* R = &grec[(size_t)g * 16];
        R[0] = fi(cg_link[g]); for (int c = 0; c < 3; c++) R[1 + c] = cg_lpos[3 * (size_t)g + c];
        for (int c = 0; c < 9; c++) R[4 + c] = cg_lmat[9 * (size_t)g + c];
        R[13] = fi(cg_type[g]); R[14] = cg_rb[g];
      }
      for (int q = 0; q < d.npair; q++) {
        const int* P = &pair_i[6 * (size_t)q];
        const float* F = &pair_f[12 * (size_t)q];
        if (P[0] > 255 || P[1] > 255 || P[3] > 255 || P[4] > 15 || P[5] > 15) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "collision pair does not fit the packed pair record"); }
        float* R = &prec[(size_t)q * 16];
        R[0] = fi(P[0] | (P[1] << 8) | (P[4] << 16) | (P[5] << 20) | (P[3] << 24)); R[1] = F[0]; R[2] = F[1]; R[3] = fi(P[2]);
        for (int c = 0; c < 3; c++) { R[4 + c] = cg_size[3 * (size_t)P[0] + c]; R[8 + c] = cg_size[3 * (size_t)P[1] + c]; }
        R[7] = cg_rb[P[0]]; R[11] = cg_rb[P[1]];
        R[12] = fi(cg_type[P[0]] | (cg_type[P[1]] << 8));
      }
      // (a contact carries pair | dofs << 11 | dof-list start << 16 in one word: step kernel, narrow phase -> row stage)
      if (d.npair > 2048 || pair_dl.size() / 2 >= (1u << 16) || d.maxkc > 31) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "more than 2048 collision pairs, 65535 contact dof-list entries or 31 dofs per contact"); }
      std::vector<int> pdl(std::max<size_t>(pair_dl.size() / 2, 1), 0);
      for (size_t i = 0; i < pair_dl.size() / 2; i++) {
        const int dd = pair_dl[2 * i], sg = pair_dl[2 * i + 1];
        if (dd < 0 || dd > 127 || sg < -(1 << 22) || sg > (1 << 22)) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "contact dof-list entry does not fit the packed word"); }
        pdl[i] = dd | ((dof_type[dd] == 3 ? 1 : 0) << 7) | (int)((unsigned)sg << 8);
      }
      const float *tg4 = nullptr, *tp4 = nullptr; const int* tq4 = nullptr;
      if ((rc = upload<float>(m, grec, &tg4)) || (rc = upload<float>(m, prec, &tp4)) || (rc = upload<int>(m, pdl, &tq4))) { myo_model_free(m); return rc; }
      w.cg_rec = (decltype(w.cg_rec))tg4; w.pair_rec = (decltype(w.pair_rec))tp4; w.pair_dl_pk = (decltype(w.pair_dl_pk))tq4;
      // tree words: one packed word per lane and round for the sweeps over the kinematic tree (DevModelW::kin_pk, link_desc, link_adof, dof_anc)
      {
        const std::vector<int> lpar = BI("hip_link_parent"), dlink = BI("hip_dof_link"), dpar = BI("dof_parentid"), kadr = BI("hip_kin_adr"), kvec = BI("hip_kin_vec"),
                               chadr = BI("hip_link_chain_adr"), chain = BI("hip_link_chain");
        if (d.nl > 64 || d.nv > 64) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "more than 64 links or dofs"); }
        std::vector<int> kpk;
        for (size_t L = 0; L + 1 < kadr.size(); L++) {
          for (int e0 = kadr[L]; e0 < kadr[L + 1]; e0 += 64) {
            for (int i = 0; i < 64; i++) {
              const int e = e0 + i;
              if (e >= kadr[L + 1]) { kpk.push_back(-1); continue; }
              const int w0 = kvec[2 * (size_t)e], src = kvec[2 * (size_t)e + 1], l = w0 & 255, kind = (w0 >> 8) & 3, ix = w0 >> 16, par1 = lpar[l] + 1;
              if (src < 0 || src >= 2048 || l >= 64 || ix < 0 || ix >= 64 || par1 < 0 || par1 > 64) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "kinematics entry does not fit the packed word"); }
              kpk.push_back((int)((unsigned)src | ((unsigned)l << 11) | ((unsigned)kind << 17) | ((unsigned)ix << 19) | ((unsigned)par1 << 25)));
            }
          }
        }
        w.kin_nround = (int)(kpk.size() / 64);
        kpk.resize(kpk.size() + 64, -1);                       // padding round: the loop prefetches one round ahead
        std::vector<unsigned long long> desc(d.nl, 0), adof(d.nl, 0), anc(d.nv, 0);
        for (int l = d.nl - 1; l >= 0; l--) { desc[l] |= 1ull << l; if (lpar[l] >= 0) { if (lpar[l] >= l) { myo_model_free(m); return fail(MYO_E_BLOB, "links are not in tree order"); } desc[lpar[l]] |= desc[l]; } }
        for (int q = 0; q < d.nv; q++) { if (dpar[q] >= q) { myo_model_free(m); return fail(MYO_E_BLOB, "dofs are not in tree order"); } anc[q] = (1ull << q) | (dpar[q] >= 0 ? anc[dpar[q]] : 0ull); }
        unsigned long long frot = 0, fj3 = 0;
        for (int l = 0; l < d.nl; l++) {
          int prev = -1;
          for (int c = chadr[l]; c < chadr[l + 1]; c++) {
            const int e = chain[c], q = e & 255, j = (e >> 8) & 7, fr = e >> 12;
            if (q <= prev || q >= d.nv) { myo_model_free(m); return fail(MYO_E_BLOB, "link dof chain is not root-first"); }
            prev = q;
            adof[l] |= 1ull << q;
            if (fr && j >= 3) frot |= 1ull << q;
            if (fr && j == 3) fj3 |= 1ull << q;
          }
        }
        w.free_rot[0] = (unsigned)frot; w.free_rot[1] = (unsigned)(frot >> 32); w.free_j3[0] = (unsigned)fj3; w.free_j3[1] = (unsigned)(fj3 >> 32);
        auto split = [](const std::vector<unsigned long long>& v) { std::vector<int> o(std::max<size_t>(2 * v.size(), 2), 0); for (size_t i = 0; i < v.size(); i++) { o[2 * i] = (int)(unsigned)v[i]; o[2 * i + 1] = (int)(unsigned)(v[i] >> 32); } return o; };
        const int *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
        if ((rc = upload<int>(m, kpk, &t0)) || (rc = upload<int>(m, split(desc), &t1)) || (rc = upload<int>(m, split(adof), &t2)) || (rc = upload<int>(m, split(anc), &t3))) { myo_model_free(m); return rc; }
        w.kin_pk = (decltype(w.kin_pk))t0; w.link_desc = (decltype(w.link_desc))t1; w.link_adof = (decltype(w.link_adof))t2; w.dof_anc = (decltype(w.dof_anc))t3;
      }
    }
    const bool common = d.nl <= 64 && d.ncg <= (m->trk ? 128 : 64) && w.nq <= 64 && w.neq <= 64 && d.maxnnz <= 20;   // (geom ids are bytes in the pair record; the TRK kernel loops over geoms)
    const bool needs_full = w.has_free || w.neq > 0 || plane_pairs || condim1 || m->trk;
    if (m->trk) {
      if (!(common && d.nv <= 36 && d.nu <= 128 && d.ngt <= 128 && d.maxkc <= 20 && !w.hf.on && !w.has_tl)) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "condim-4 / friction-loss / box / mesh model exceeds the limits of the TRK step kernel"); }
      m->wave_ok = true; m->wave_cfg = 2; m->generic_ok = false;
      build_layout_w(d, w, 36, 20, 32, 4);
    }
    else if (common && !needs_full && d.nv <= 24 && d.nu <= 64 && d.ngt <= 64 && d.maxkc <= 8) {
      m->wave_ok = true; m->wave_cfg = 0;
      build_layout_w(d, w, 24, 8, 32);
    }
    else if (common && d.nv <= 36 && d.nu <= 128 && d.ngt <= 128 && d.maxkc <= 20) { m->wave_ok = true; m->wave_cfg = 1; build_layout_w(d, w, 36, 20, 32); }
    else { m->wave_ok = false; build_layout_w(d, w, 24, 8, 32); }
    m->hand_sizes = m->wave_ok && m->wave_cfg == 0 && sizes_match<1>(w.nq, d.nv, d.nu, d.nl, d.nlevel, d.maxnnz, d.ngt, d.nseg, d.ncg, d.npair);
    if (m->trk) m->generic_ok = false;
    if (blob_find(blob, "integrator")) { std::vector<int> ig; if ((rc = load_i(m, blob, "integrator", &tmpi, &ig))) { myo_model_free(m); return rc; } m->rk4 = !ig.empty() && ig[0] == 1; }
    if (m->rk4) {
      if (!m->wave_ok || m->trk || w.hf.on) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "RK4: wave kernel models without height field / TrackEnv features only"); }
      m->generic_ok = false;
    }
    m->leg_sizes = m->wave_ok && m->wave_cfg == 1 && sizes_match<2>(w.nq, d.nv, d.nu, d.nl, d.nlevel, d.maxnnz, d.ngt, d.nseg, d.ncg, d.npair);
    m->terrain_sizes = m->wave_ok && m->wave_cfg == 1 && w.hf.on && sizes_match<3>(w.nq, d.nv, d.nu, d.nl, d.nlevel, d.maxnnz, d.ngt, d.nseg, d.ncg, d.npair);
    {  // the specialised instantiations also build in the dof tree (tree-sparse factorisation): it must be the model's
      std::vector<int> dpar;
      if ((rc = load_i(m, blob, "dof_parentid", &tmpi, &dpar))) { myo_model_free(m); return rc; }
      auto same_tree = [&](const int* ref, int n) { if ((int)dpar.size() != n) return false; for (int i = 0; i < n; i++) if (dpar[i] != ref[i]) return false; return true; };
      if (m->hand_sizes && !same_tree(SpecTree<1>::parent, SpecTree<1>::nv)) m->hand_sizes = false;
      if ((m->leg_sizes || m->terrain_sizes) && !same_tree(SpecTree<2>::parent, SpecTree<2>::nv)) m->leg_sizes = m->terrain_sizes = false;
    }
    if (m->rk4) m->hand_sizes = m->leg_sizes = m->terrain_sizes = false;
    if (w.has_tl) m->hand_sizes = m->leg_sizes = m->terrain_sizes = false;   // the specialised instantiations compile the tendon-limit rows out
    // ... and their LDS layout in: it must be the one this model was given
    if (m->hand_sizes && !layout_match<1, 24, 8, 32, 3>(w.lay)) m->hand_sizes = false;
    if (m->leg_sizes && !layout_match<2, 36, 20, 32, 3>(w.lay)) m->leg_sizes = false;
    if (m->terrain_sizes && !layout_match<3, 36, 20, 32, 3>(w.lay)) m->terrain_sizes = false;
    if (const char* e = getenv("MYO_NO_SPEC")) if (atoi(e) == 1) m->hand_sizes = m->leg_sizes = m->terrain_sizes = false;   // tests: force the run-time-sized instantiations
    if (!m->wave_ok && !m->generic_ok) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "model exceeds the limits of both step kernels (nv <= 36, nu <= 128, pair dofs <= 20)"); }
    { const int nvt = m->wave_cfg == 0 ? 24 : 36;
      if (m->wave_ok && m->kin_floats > nvt * (nvt + 1)) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "kinematics scratch exceeds the Hessian scratch it borrows"); } }
    m->env_lds_bytes_w = w.lay.total * 4;
    if (m->wave_ok && m->env_lds_bytes_w > 64 * 1024) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "wave kernel working set exceeds 64 KB of LDS"); }
    void* p1 = nullptr; void* p2 = nullptr;
    if (hipMalloc(&p1, sizeof(DevModel)) != hipSuccess || hipMalloc(&p2, sizeof(DevModelW)) != hipSuccess) { myo_model_free(m); return fail(MYO_E_NOMEM, "hipMalloc model structs"); }
    m->dev_allocs.push_back(p1); m->dev_allocs.push_back(p2);
    m->d_dm = (DevModel*)p1; m->d_dw = (DevModelW*)p2;
    if (hipMemcpy(p1, &d, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(p2, &w, sizeof(DevModelW), hipMemcpyHostToDevice) != hipSuccess) { myo_model_free(m); return fail(MYO_E_HIP, "upload model structs"); }
  }
  m->dims = myo_dims{S[0], S[1], S[2], d.na_obs, S[4], S[8], S[7], d.nl, 0, m->wave_ok ? m->env_lds_bytes_w : m->env_lds_bytes, 64,
                     m->wave_ok ? (m->wave_cfg >= 1 ? 32 : NCONW) : NCON, d.timestep};
  if (4 * m->env_lds_bytes > 160 * 1024) {
    if (!m->wave_ok) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "model working set exceeds 160 KB of LDS per workgroup"); }
    m->generic_ok = false;
  }
  *out = m;
  return MYO_OK;
}

void myo_model_free(myo_model* m) {
  if (!m) return;
  for (void* p : m->dev_allocs) (void)hipFree(p);
  delete m;
}

int myo_model_dims(const myo_model* m, myo_dims* out) {
  if (!m || !out) return fail(MYO_E_ARG, "myo_model_dims: null");
  *out = m->dims;
  return MYO_OK;
}

int myo_model_set_switch(myo_model* m, int dc, int dl, int de) {
  if (!m) return fail(MYO_E_ARG, "null model");
  m->dm.disable_contact = dc; m->dm.disable_limit = dl; m->dm.disable_ellipsoid = de;
  if (m->d_dm) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(m->d_dm, &m->dm, sizeof(DevModel), hipMemcpyHostToDevice)); }
  return MYO_OK;
}

static int balloc(myo_batch* b, void** p, size_t nbytes) {
  HIPCHK(hipMalloc(p, nbytes));
  HIPCHK(hipMemset(*p, 0, nbytes));
  b->dev_allocs.push_back(*p);
  return 0;
}

int myo_batch_create(const myo_model* m, int B, myo_batch** out) {
  if (!m || !out || B <= 0) return fail(MYO_E_ARG, "myo_batch_create: bad arguments");
  HIPCHK(hipSetDevice(m->device));
  myo_batch* b = new myo_batch();
  b->model = m;
  DevBatch& d = b->db;
  d.B = B;
  int nv = m->dm.nv, nu = m->dm.nu, nq = m->nq, rc;
  b->ntarget_alloc = nv > 24 ? nv : 24;
  b->obs_alloc = 3 * nv + 4 * nu + 64;
#define BA(ptr, n) if ((rc = balloc(b, (void**)&ptr, (size_t)(n) * 4))) { myo_batch_free(b); return rc; }
  BA(d.qpos, (size_t)B * nq) BA(d.qvel, (size_t)B * nv) BA(d.act, (size_t)B * nu) BA(d.ctrl, (size_t)B * nu) BA(d.warm, (size_t)B * nv)
  BA(d.time, B) BA(d.target, (size_t)B * b->ntarget_alloc) BA(d.obs, (size_t)B * b->obs_alloc) BA(d.reward, B) BA(d.done, B)
  BA(d.solved, B) BA(d.qacc, (size_t)B * nv) BA(d.tenlen, (size_t)B * nu) BA(d.actforce, (size_t)B * nu) BA(d.sitexpos, (size_t)B * 24)
  BA(d.flags, B) BA(d.diag, (size_t)B * 8) BA(d.elapsed, B) BA(d.episode, B) BA(d.mprw, (size_t)B * 64)
  BA(b->d_tlo, b->ntarget_alloc) BA(b->d_thi, b->ntarget_alloc) BA(b->d_init, nq) BA(b->d_jlo, nv) BA(b->d_jhi, nv) BA(b->d_rnd, 4 * (size_t)nq)
  BA(b->d_action, (size_t)B * nu)
  BA(d.fatigue, (size_t)B * 3 * nu)
  d.hfield = nullptr; d.gsize = nullptr; d.gsize_cg = -1;
  d.ovf = nullptr; d.ovf_cand = nullptr; d.ovf_row = 0; d.ovf_rows = 0; d.linkx = nullptr; d.track = nullptr; d.env_offset = 0;
  if (m->trk) { BA(d.linkx, (size_t)B * 12 * m->dm.nl) }
  if (m->wave_ok) {   // contact-table overflow rows of the wave kernel (instantiations <24,8,...> and <36,20,...>)
    const int kc = m->wave_cfg == 0 ? 8 : 20, nj = m->trk ? 4 : 3;
    d.ovf_row = 8 + nj * kc + (kc + 3) / 4 + (m->trk ? TRK_STATE : 0);
    d.ovf_rows = m->trk ? NCX2 : NCX;
    BA(d.ovf, (size_t)B * d.ovf_rows * d.ovf_row) BA(d.ovf_cand, (size_t)B * NCANDX)
  }
  if (m->dw.hf.on) { BA(d.hfield, (size_t)B * m->dw.hf.nrow * m->dw.hf.ncol) }   // zero-filled: flat terrain at the geom's height
  BA(b->d_initv, nv) BA(b->d_init2, nq) BA(b->d_initv2, nv) BA(b->d_fatvec, nu)
  { void* pw = nullptr; if ((rc = balloc(b, &pw, sizeof(DevWalk)))) { myo_batch_free(b); return rc; } b->d_walk = (DevWalk*)pw; }
  BA(b->d_stamps, (size_t)B * 12 * 2 * 2)      // 2 x 12 long long per workgroup (diagnostic build)
  BA(b->d_order, B)
  b->sched_stride = (B + 7) / 8 + 1;                 // per queue with 8 queues; launch_step widens it when the device shows fewer XCDs
  BA(b->d_sched, 32 + B + 64)
#undef BA
  if (const char* e = getenv("MYO_LANES")) { int g = atoi(e); if (g == 16 || g == 32 || g == 64) g_lanes = g; }
  HIPCHK(hipMemcpy(b->d_jlo, m->jnt_lo.data(), nv * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_jhi, m->jnt_hi.data(), nv * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_init, m->qpos0.data(), nq * 4, hipMemcpyHostToDevice));
  // default: every env at qpos0
  std::vector<float> q((size_t)B * nq);
  for (int e = 0; e < B; e++) memcpy(&q[(size_t)e * nq], m->qpos0.data(), nq * 4);
  HIPCHK(hipMemcpy(d.qpos, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  {
    std::vector<float> f((size_t)B * 3 * nu, 0.f);
    for (int e = 0; e < B; e++) for (int i = 0; i < nu; i++) f[(size_t)e * 3 * nu + nu + i] = 1.f;      // MR = 1
    HIPCHK(hipMemcpy(d.fatigue, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    d.fat_dt = m->dm.timestep; d.reaf_epl = d.reaf_eip = -1;
  }
  b->task.task = MYO_TASK_NONE; b->task.frame_skip = 1; b->task.obs_dim = 0; b->task.ntarget = 0;
  b->task.jnt_lo = b->d_jlo; b->task.jnt_hi = b->d_jhi; b->task.init_qpos = b->d_init; b->task.target_lo = b->d_tlo; b->task.target_hi = b->d_thi;
  b->task.init_qvel = nullptr;
  b->task.init_qpos_alt = nullptr; b->task.init_qvel_alt = nullptr; b->task.reset_noise_std = 0.f; b->task.fatigue_mode = 0; b->task.fatigue_vec = nullptr;
  HIPCHK(hipEventCreate(&b->ev0));
  HIPCHK(hipEventCreate(&b->ev1));
  *out = b;
  return MYO_OK;
}

void myo_batch_free(myo_batch* b) {
  if (!b) return;
  for (hipEvent_t e : b->kev) (void)hipEventDestroy(e);
  for (void* p : b->dev_allocs) (void)hipFree(p);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  delete b;
}

int myo_batch_size(const myo_batch* b) { return b ? b->db.B : 0; }

int myo_batch_configure(myo_batch* b, const myo_task_config* c) {
  if (!b || !c) return fail(MYO_E_ARG, "myo_batch_configure: null");
  const DevModel& dm = b->model->dm;
  TaskDev& T = b->task;
  int nv = dm.nv, nu = dm.nu;
  if (c->ntarget > b->ntarget_alloc || c->ntip > 8) return fail(MYO_E_ARG, "myo_batch_configure: ntarget/ntip too large");
  if (c->task == MYO_TASK_WALK) return fail(MYO_E_ARG, "use myo_batch_configure_walk for the walk task");
  T.init_qvel = nullptr; T.rnd = nullptr;
  T.terrain = 0; T.hf_n = 0;
  for (int k = 0; k < 3; k++) T.tip_lpos[k] = c->tip_lpos[k];
  if (c->reset_noise_lo || c->reset_noise_hi || c->reset_clip_lo || c->reset_clip_hi) {
    if (!(c->reset_noise_lo && c->reset_noise_hi && c->reset_clip_lo && c->reset_clip_hi)) return fail(MYO_E_ARG, "reset noise: all four arrays or none");
    const float* src[4] = {c->reset_noise_lo, c->reset_noise_hi, c->reset_clip_lo, c->reset_clip_hi};
    for (int k = 0; k < 4; k++) HIPCHK(hipMemcpy(b->d_rnd + (size_t)k * b->model->nq, src[k], (size_t)b->model->nq * 4, hipMemcpyHostToDevice));
    T.rnd = b->d_rnd;
  }
  if (c->init_qvel) { HIPCHK(hipMemcpy(b->d_initv, c->init_qvel, (size_t)nv * 4, hipMemcpyHostToDevice)); T.init_qvel = b->d_initv; }
  T.task = c->task; T.frame_skip = c->frame_skip; T.reset_random = c->reset_random; T.target_generate = c->target_generate;
  T.ntarget = c->ntarget; T.ntip = c->ntip;
  for (int i = 0; i < 8; i++) T.tip_site[i] = c->tip_site[i];
  T.pose_thd = c->pose_thd; T.far_th = c->far_th; T.near_th = c->near_th;
  T.w_pose = c->w_pose; T.w_bonus = c->w_bonus; T.w_act_reg = c->w_act_reg; T.w_penalty = c->w_penalty; T.w_reach = c->w_reach;
  if (c->task == MYO_TASK_POSE) { if (c->ntarget != nv) return fail(MYO_E_ARG, "pose task: ntarget must equal nq"); T.obs_dim = 3 * nv + b->model->dm.na_obs; }
  else if (c->task == MYO_TASK_REACH) { if (c->ntarget != 3 * c->ntip) return fail(MYO_E_ARG, "reach task: ntarget must be 3*ntip"); T.obs_dim = 2 * nv + 6 * c->ntip + b->model->dm.na_obs; }
  else if (c->task == MYO_TASK_STAND) {
    if (c->ntarget != 3) return fail(MYO_E_ARG, "stand task: ntarget must be 3 (target position)");
    if (!(b->model->wave_ok && b->model->dw.has_free && b->model->nq == nv + 1)) return fail(MYO_E_UNSUPPORTED, "stand task needs a model with a free root joint");
    T.obs_dim = b->model->nq + nv + 6 + b->model->dm.na_obs;
  }
  else if (c->task == MYO_TASK_HOLD) {
    if (c->ntarget != 3) return fail(MYO_E_ARG, "hold task: ntarget must be 3 (goal position)");
    if (!(b->model->wave_ok && b->model->dw.has_free && b->model->nq == nv + 1 && nv > 6)) return fail(MYO_E_UNSUPPORTED, "hold task needs a model whose last joint is one free object");
    T.obs_dim = (b->model->nq - 7) + (nv - 6) + 6 + b->model->dm.na_obs;
  }
  else T.obs_dim = 0;
  T.nq = b->model->nq;
  if (T.obs_dim > b->obs_alloc) return fail(MYO_E_ARG, "obs_dim too large");
  if (c->ntarget > 0) {
    if (!c->target_lo) return fail(MYO_E_ARG, "target_lo required");
    HIPCHK(hipMemcpy(b->d_tlo, c->target_lo, c->ntarget * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_thi, c->target_hi ? c->target_hi : c->target_lo, c->ntarget * 4, hipMemcpyHostToDevice));
  }
  if (c->reset_random && b->model->nq != nv) return fail(MYO_E_ARG, "reset_random needs a model without free / ball joints");
  HIPCHK(hipMemcpy(b->d_init, c->init_qpos ? c->init_qpos : b->model->qpos0.data(), b->model->nq * 4, hipMemcpyHostToDevice));
  return MYO_OK;
}

int myo_batch_set_geom_override(myo_batch* b, int geom_id, const float* lo, const float* hi) {
  if (!b) return fail(MYO_E_ARG, "myo_batch_set_geom_override: null");
  const myo_model* m = b->model;
  if (!lo || !hi) { b->task.gsize_type = 0; b->db.gsize_cg = -1; return MYO_OK; }
  if (!(m->wave_ok && m->wave_cfg == 1) || m->leg_sizes || m->dw.hf.on) return fail(MYO_E_UNSUPPORTED, "geom override: generic large-kernel models only");
  int cg = -1;
  for (size_t i = 0; i < m->cg_geom.size(); i++) if (m->cg_geom[i] == geom_id) cg = (int)i;
  if (cg < 0) return fail(MYO_E_ARG, "geom override: not a collision geom of this model");
  const int ty = m->cg_type_h[cg];
  if (ty != GEOM_SPHERE && ty != GEOM_CAPSULE && ty != GEOM_ELLIPSOID && ty != GEOM_CYLINDER) return fail(MYO_E_UNSUPPORTED, "geom override: sphere / capsule / ellipsoid / cylinder geoms only");
  if (!b->db.gsize) {
    int rc = balloc(b, (void**)&b->db.gsize, (size_t)b->db.B * 4 * 4);
    if (rc) return rc;
  }
  b->db.gsize_cg = cg;
  b->task.gsize_type = ty;
  for (int k = 0; k < 3; k++) { b->task.gsize_lo[k] = lo[k]; b->task.gsize_hi[k] = hi[k]; }
  return MYO_OK;
}

int myo_batch_configure_walk(myo_batch* b, const myo_walk_config* c) {
  if (!b || !c) return fail(MYO_E_ARG, "myo_batch_configure_walk: null");
  const myo_model* m = b->model;
  const DevModel& dm = m->dm;
  const int nq = m->nq, nv = dm.nv, nu = dm.nu, nb = (int)m->body_link.size();
  if (!(m->wave_ok && m->wave_cfg == 1 && m->dw.has_free)) return fail(MYO_E_UNSUPPORTED, "walk task needs a free-floating model on the large wave kernel");
  const int bodies[4] = {c->body_talus_l, c->body_talus_r, c->body_pelvis, c->body_torso};
  for (int k = 0; k < 4; k++) if (bodies[k] < 1 || bodies[k] >= nb || m->body_link[bodies[k]] < 0) return fail(MYO_E_ARG, "walk task: bad body id");
  if (m->body_link[c->body_torso] != 0) return fail(MYO_E_UNSUPPORTED, "walk task: torso must be welded to the free root body");
  const int qa[6] = {c->qadr_hip_flexion_l, c->qadr_hip_flexion_r, c->qadr_joint_angle[0], c->qadr_joint_angle[1], c->qadr_joint_angle[2], c->qadr_joint_angle[3]};
  for (int k = 0; k < 6; k++) if (qa[k] < 0 || qa[k] >= nq) return fail(MYO_E_ARG, "walk task: bad qpos address");
  if (c->frame_skip <= 0 || c->hip_period <= 0 || !c->init_qpos) return fail(MYO_E_ARG, "walk task: frame_skip, hip_period, init_qpos required");
  DevWalk w{};
  w.obs_dim = (nq - 2) + nv + 16 + 4 * nu;
  if (w.obs_dim > b->obs_alloc) return fail(MYO_E_ARG, "obs_dim too large");
  w.hip_period = c->hip_period; w.dt = (float)c->frame_skip * dm.timestep;
  w.min_height = c->min_height; w.max_rot = c->max_rot; w.target_x_vel = c->target_x_vel; w.target_y_vel = c->target_y_vel;
  for (int k = 0; k < 4; k++) { w.target_rot[k] = c->target_rot[k]; w.lquat_tor[k] = m->body_lquat[4 * c->body_torso + k]; w.qadr_ja[k] = c->qadr_joint_angle[k]; }
  w.link_tl = m->body_link[c->body_talus_l]; w.link_tr = m->body_link[c->body_talus_r];
  w.link_pel = m->body_link[c->body_pelvis]; w.link_tor = m->body_link[c->body_torso];
  for (int k = 0; k < 3; k++) {
    w.lpos_tl[k] = m->body_lpos[3 * c->body_talus_l + k]; w.lpos_tr[k] = m->body_lpos[3 * c->body_talus_r + k];
    w.lpos_pel[k] = m->body_lpos[3 * c->body_pelvis + k]; w.static_mcom[k] = m->mass[1 + k];
  }
  w.qadr_hfl = c->qadr_hip_flexion_l; w.qadr_hfr = c->qadr_hip_flexion_r;
  w.w_vel = c->w_vel_reward; w.w_done = c->w_done; w.w_cyc = c->w_cyclic_hip; w.w_rot = c->w_ref_rot; w.w_ja = c->w_joint_angle_rew;
  w.mass_total = m->mass[0];
  w.knee_height = c->knee_height;
  if (c->terrain != MYO_TERRAIN_NONE && !(m->dw.hf.on && m->dw.hf.nrow == 100 && m->dw.hf.ncol == 100))
    return fail(MYO_E_UNSUPPORTED, "terrain walk needs a model with a colliding 100 x 100 height field");
  if (c->terrain < 0 || c->terrain > MYO_TERRAIN_STAIRS) return fail(MYO_E_ARG, "walk task: bad terrain kind");
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_walk, &w, sizeof w, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_init, c->init_qpos, (size_t)nq * 4, hipMemcpyHostToDevice));
  if (c->init_qvel) HIPCHK(hipMemcpy(b->d_initv, c->init_qvel, (size_t)nv * 4, hipMemcpyHostToDevice));
  TaskDev& T = b->task;
  T.task = MYO_TASK_WALK; T.frame_skip = c->frame_skip; T.reset_random = 0; T.target_generate = 0; T.ntarget = 0; T.ntip = 0;
  T.obs_dim = w.obs_dim;
  T.init_qvel = c->init_qvel ? b->d_initv : nullptr;
  T.init_qpos_alt = nullptr; T.init_qvel_alt = nullptr; T.reset_noise_std = 0.f;
  if (c->init_qpos_alt) {
    HIPCHK(hipMemcpy(b->d_init2, c->init_qpos_alt, (size_t)nq * 4, hipMemcpyHostToDevice));
    T.init_qpos_alt = b->d_init2; T.reset_noise_std = c->reset_noise_std;
    if (c->init_qvel_alt) { HIPCHK(hipMemcpy(b->d_initv2, c->init_qvel_alt, (size_t)nv * 4, hipMemcpyHostToDevice)); T.init_qvel_alt = b->d_initv2; }
  }
  T.terrain = c->terrain; T.hf_n = c->terrain ? m->dw.hf.nrow * m->dw.hf.ncol : 0; T.terrain_lo = c->terrain_scalar_lo; T.terrain_hi = c->terrain_scalar_hi;
  return MYO_OK;
}

int myo_batch_configure_track(myo_batch* b, const myo_track_config* c) {
  if (!b || !c) return fail(MYO_E_ARG, "myo_batch_configure_track: null");
  const myo_model* m = b->model;
  if (!(m->wave_ok && m->trk)) return fail(MYO_E_UNSUPPORTED, "track task: models of the TrackEnv class only (condim-4 / friction-loss / hull geoms)");
  const int nq = m->nq, nv = m->dm.nv, nu = m->dm.nu;
  if (c->n_frames <= 0 || c->ref_type < 0 || c->ref_type > 2 || c->horizon < 1 || c->robot_dim < 0 || c->object_dim < 0 || c->robot_dim > 64 ||
      c->robot_dim + c->object_dim > 64 || c->robot_dim > nq || !c->ref_time || !c->init_qpos || !c->ctrl_lo || !c->ctrl_hi)
    return fail(MYO_E_ARG, "myo_batch_configure_track: bad reference / pose arguments");
  if ((c->robot_dim > 0 && (!c->ref_robot || c->robot_horizon < 1)) || (c->object_dim > 0 && (!c->ref_object || c->object_horizon < 1)))
    return fail(MYO_E_ARG, "myo_batch_configure_track: reference rows missing");
  if (c->ref_type == 1 && ((c->robot_dim > 0 && c->robot_horizon < 2) || (c->object_dim > 0 && c->object_horizon < 2))) return fail(MYO_E_ARG, "RANDOM reference needs two rows");
  if (c->object_link < 0 || c->object_link >= m->dm.nl || c->wrist_link < 0 || c->wrist_link >= m->dm.nl) return fail(MYO_E_ARG, "myo_batch_configure_track: link out of range");
  if (nq + nv > b->obs_alloc) return fail(MYO_E_ARG, "obs_dim too large");
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipDeviceSynchronize());
  const int B = b->db.B;
  DevTrack K{};
  K.ref_type = c->ref_type; K.horizon = c->horizon; K.robot_horizon = c->robot_horizon; K.object_horizon = c->object_horizon;
  K.robot_dim = c->robot_dim; K.object_dim = c->object_dim; K.has_vel = c->ref_robot_vel != nullptr;
  K.extrapolate = c->motion_extrapolation; K.linear = c->interpolation_linear; K.autoreset = c->autoreset;
  K.term_obj = c->terminate_obj_fail; K.term_pose = c->terminate_pose_fail; K.start_time = c->motion_start_time; K.max_steps = c->max_episode_steps;
  int rc;
  auto up = [&](const void* src, size_t nbytes, const void** dst) -> int {
    void* p = nullptr;
    if ((rc = balloc(b, &p, nbytes ? nbytes : 8))) return rc;
    if (nbytes) HIPCHK(hipMemcpy(p, src, nbytes, hipMemcpyHostToDevice));
    *dst = p;
    return 0;
  };
  if ((rc = up(c->ref_time, (size_t)c->horizon * 8, (const void**)&K.T))) return rc;
  if ((rc = up(c->ref_robot, (size_t)c->robot_horizon * c->robot_dim * 8, (const void**)&K.robot))) return rc;
  if (K.has_vel) { if ((rc = up(c->ref_robot_vel, (size_t)c->robot_horizon * c->robot_dim * 8, (const void**)&K.robot_vel))) return rc; } else K.robot_vel = nullptr;
  if ((rc = up(c->ref_object, (size_t)c->object_horizon * c->object_dim * 8, (const void**)&K.object))) return rc;
  if ((rc = up(c->init_qpos, (size_t)nq * 4, (const void**)&K.init_qpos)) || (rc = up(c->ctrl_lo, (size_t)nu * 4, (const void**)&K.lo)) ||
      (rc = up(c->ctrl_hi, (size_t)nu * 4, (const void**)&K.hi))) return rc;
  K.obj_link = c->object_link; K.wrist_link = c->wrist_link;
  for (int k = 0; k < 3; k++) { K.obj_p[k] = c->object_ipos[k]; K.wrist_p[k] = c->wrist_ipos[k]; }
  for (int k = 0; k < 9; k++) K.obj_R[k] = c->object_imat[k];
  K.lift_z = c->lift_z; K.obj_err_scale = c->obj_err_scale; K.base_err_scale = c->base_err_scale; K.lift_bonus_mag = c->lift_bonus_mag;
  K.qpos_w = c->qpos_reward_weight; K.qpos_err_scale = c->qpos_err_scale; K.qvel_w = c->qvel_reward_weight; K.qvel_err_scale = c->qvel_err_scale;
  K.obj_fail2 = c->obj_fail_thresh * c->obj_fail_thresh; K.base_fail2 = c->base_fail_thresh * c->base_fail_thresh; K.qpos_fail = c->qpos_fail_thresh;
  K.w_pose = c->w_pose; K.w_object = c->w_object; K.w_bonus = c->w_bonus; K.w_penalty = c->w_penalty;
  K.ref_pitch = 2 * c->robot_dim + c->object_dim;
  K.seed = c->seed;
  { void* p = nullptr; if ((rc = balloc(b, &p, (size_t)B * K.ref_pitch * 4))) return rc; K.ref = (float*)p; }
  if (!b->d_metrics) { void* p = nullptr; if ((rc = balloc(b, &p, (size_t)B * 4 * 4))) return rc; b->d_metrics = (float*)p; }
  K.metrics = b->d_metrics;
  if (!b->d_track) { void* p = nullptr; if ((rc = balloc(b, &p, sizeof(DevTrack)))) return rc; b->d_track = (DevTrack*)p; }
  HIPCHK(hipMemcpy(b->d_track, &K, sizeof(DevTrack), hipMemcpyHostToDevice));
  b->db.track = b->d_track;
  b->track_frames = c->n_frames;
  // the generic reset / observation plumbing: TrackEnv.reset puts every env at init_qpos with zero velocity, activation, control and time
  TaskDev& T = b->task;
  T = TaskDev{};
  T.task = MYO_TASK_TRACK; T.frame_skip = c->n_frames; T.nq = nq; T.obs_dim = nq + nv;
  HIPCHK(hipMemcpy(b->d_init, c->init_qpos, (size_t)nq * 4, hipMemcpyHostToDevice));
  T.init_qpos = b->d_init; T.jnt_lo = b->d_jlo; T.jnt_hi = b->d_jhi; T.target_lo = b->d_tlo; T.target_hi = b->d_thi;
  return MYO_OK;
}

static int field_info(myo_batch* b, int f, void** p, size_t* pitch, size_t* width) {
  const DevModel& dm = b->model->dm;
  DevBatch& d = b->db;
  size_t nv = dm.nv, nu = dm.nu;
  switch (f) {
    case MYO_F_QPOS: *p = d.qpos; *pitch = *width = (size_t)b->model->nq; break;
    case MYO_F_QVEL: *p = d.qvel; *pitch = *width = nv; break;
    case MYO_F_ACT: *p = d.act; *pitch = *width = nu; break;
    case MYO_F_CTRL: *p = d.ctrl; *pitch = *width = nu; break;
    case MYO_F_WARMSTART: *p = d.warm; *pitch = *width = nv; break;
    case MYO_F_TIME: *p = d.time; *pitch = *width = 1; break;
    case MYO_F_TARGET: *p = d.target; *pitch = *width = b->task.ntarget > 0 ? b->task.ntarget : 1; break;
    case MYO_F_OBS: *p = d.obs; *pitch = *width = b->task.obs_dim > 0 ? b->task.obs_dim : 1; break;
    case MYO_F_REWARD: *p = d.reward; *pitch = *width = 1; break;
    case MYO_F_DONE: *p = d.done; *pitch = *width = 1; break;
    case MYO_F_SOLVED: *p = d.solved; *pitch = *width = 1; break;
    case MYO_F_FLAGS: *p = d.flags; *pitch = *width = 1; break;
    case MYO_F_DIAG: *p = d.diag; *pitch = *width = 8; break;
    case MYO_F_QACC: *p = d.qacc; *pitch = *width = nv; break;
    case MYO_F_TENLEN: *p = d.tenlen; *pitch = *width = nu; break;
    case MYO_F_ACTFORCE: *p = d.actforce; *pitch = *width = nu; break;
    case MYO_F_ELAPSED: *p = d.elapsed; *pitch = *width = 1; break;
    case MYO_F_LINKX:
      if (!d.linkx) return fail(MYO_E_ARG, "MYO_F_LINKX: this model's kernel does not export link frames");
      *p = d.linkx; *pitch = *width = (size_t)12 * b->model->dm.nl; break;
    case MYO_F_ACTION: *p = b->d_action; *pitch = *width = nu; break;
    case MYO_F_FATIGUE: *p = d.fatigue; *pitch = *width = 3 * nu; break;
    case MYO_F_GEOMSIZE:
      if (!d.gsize) return fail(MYO_E_ARG, "MYO_F_GEOMSIZE: no geom override set (myo_batch_set_geom_override)");
      *p = d.gsize; *pitch = *width = 4; break;
    case MYO_F_HFIELD:
      if (!d.hfield) return fail(MYO_E_ARG, "MYO_F_HFIELD: the model has no colliding height field");
      *p = d.hfield; *pitch = *width = b->model->dw.hf.nrow * b->model->dw.hf.ncol; break;
    case MYO_F_METRICS:
      if (!b->d_metrics) return fail(MYO_E_ARG, "MYO_F_METRICS: the track task is not configured (myo_batch_configure_track)");
      *p = b->d_metrics; *pitch = *width = 4; break;
    case MYO_F_SITEXPOS: *p = d.sitexpos; *pitch = *width = b->task.ntip > 0 ? 3 * b->task.ntip : 1; break;
    default: return fail(MYO_E_ARG, "unknown field");
  }
  return MYO_OK;
}

int myo_batch_field(myo_batch* b, int field, void** dev_ptr, size_t* pitch, size_t* width) {
  if (!b || !dev_ptr || !pitch || !width) return fail(MYO_E_ARG, "myo_batch_field: null");
  return field_info(b, field, dev_ptr, pitch, width);
}

int myo_batch_read(myo_batch* b, int field, void* host, size_t nbytes) {
  void* p; size_t pitch, width;
  if (!b || !host) return fail(MYO_E_ARG, "myo_batch_read: null");
  int rc = field_info(b, field, &p, &pitch, &width);
  if (rc) return rc;
  if (nbytes != (size_t)b->db.B * width * 4) return fail(MYO_E_ARG, "myo_batch_read: size mismatch");
  HIPCHK(hipSetDevice(b->model->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, p, nbytes, hipMemcpyDeviceToHost));
  return MYO_OK;
}

int myo_batch_write(myo_batch* b, int field, const void* host, size_t nbytes) {
  void* p; size_t pitch, width;
  if (!b || !host) return fail(MYO_E_ARG, "myo_batch_write: null");
  int rc = field_info(b, field, &p, &pitch, &width);
  if (rc) return rc;
  if (nbytes != (size_t)b->db.B * width * 4) return fail(MYO_E_ARG, "myo_batch_write: size mismatch");
  HIPCHK(hipSetDevice(b->model->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(p, host, nbytes, hipMemcpyHostToDevice));
  return MYO_OK;
}

int myo_reset(myo_batch* b, const uint8_t* mask_dev, uint64_t seed, void* stream) {
  if (!b) return fail(MYO_E_ARG, "myo_reset: null");
  const DevModel& dm = b->model->dm;
  HIPCHK(hipSetDevice(b->model->device));
  int B = b->db.B;
  hipLaunchKernelGGL(reset_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, b->db, b->task, b->model->nq, dm.nv, dm.nu, dm.qpos0, mask_dev, seed,
                     b->env_offset, 0);
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

int myo_autoreset(myo_batch* b, int max_episode_steps, uint64_t seed, void* stream) {
  if (!b || max_episode_steps <= 0) return fail(MYO_E_ARG, "myo_autoreset: bad arguments");
  const DevModel& dm = b->model->dm;
  HIPCHK(hipSetDevice(b->model->device));
  int B = b->db.B;
  hipLaunchKernelGGL(reset_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, b->db, b->task, b->model->nq, dm.nv, dm.nu, dm.qpos0,
                     (const uint8_t*)nullptr, seed, b->env_offset, max_episode_steps);
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

int myo_set_env_offset(myo_batch* b, int env_offset) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  b->env_offset = env_offset;
  b->db.env_offset = env_offset;
  return MYO_OK;
}

int myo_set_state(myo_batch* b, const float* qpos, const float* qvel, const float* act, const float* time, void* stream) {
  if (!b) return fail(MYO_E_ARG, "myo_set_state: null");
  const DevModel& dm = b->model->dm;
  size_t B = b->db.B;
  hipStream_t s = (hipStream_t)stream;
  if (qpos) HIPCHK(hipMemcpyAsync(b->db.qpos, qpos, B * b->model->nq * 4, hipMemcpyDeviceToDevice, s));
  if (qvel) HIPCHK(hipMemcpyAsync(b->db.qvel, qvel, B * dm.nv * 4, hipMemcpyDeviceToDevice, s));
  if (act) HIPCHK(hipMemcpyAsync(b->db.act, act, B * dm.nu * 4, hipMemcpyDeviceToDevice, s));
  if (time) HIPCHK(hipMemcpyAsync(b->db.time, time, B * 4, hipMemcpyDeviceToDevice, s));
  return MYO_OK;
}

static int launch_step(myo_batch* b, const float* action, int actmap, int nsub, hipStream_t s, int kflags = 0) {
  const myo_model* m = b->model;
  const DevWalk* wk = b->task.task == MYO_TASK_WALK ? b->d_walk : nullptr;
  if (kflags && !(wk && g_lanes == 64)) return fail(MYO_E_ARG, "observation pass: walk task on the wave kernel only");
  // models the wave kernel cannot take fall back to 16 lanes; models only the wave kernel can take always use it
  const int G = (g_lanes == 64 && !m->wave_ok) ? 16 : ((g_lanes != 64 && !m->generic_ok) ? 64 : g_lanes), EPW = 64 / G;
  int grid = (b->db.B + EPW - 1) / EPW;
  size_t lds = (size_t)EPW * m->env_lds_bytes;
  {  // kernel attributes are per device: one flag per device ordinal for the lanes kernels and the reach observation kernel too
    static std::mutex attr_g_mu;
    static bool attr_g_dev[64] = {};
    std::lock_guard<std::mutex> attr_g_lock(attr_g_mu);
    bool& attr_set = attr_g_dev[m->device & 63];
    if (!attr_set) {
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)reach_obs_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
  }
  long long* st = b->d_stamps;
#if MYO_POISON
  {  // MYO_POISON_MODE (diagnostic build only): bit 0 scratch memory, bit 1 vector registers, bit 2 scalar registers; default all
    static const int pmode = [] { const char* e = getenv("MYO_POISON_MODE"); return e ? atoi(e) : 7; }();
    if (pmode & 1) hipLaunchKernelGGL(scratch_poison_kernel, dim3(16384), dim3(64), 0, s, (float*)b->d_stamps, 0);
    if (pmode & 2) hipLaunchKernelGGL(vgpr_poison_kernel, dim3(8192), dim3(64), 0, s, 0x7fc0dead);
    if (pmode & 4) hipLaunchKernelGGL(sgpr_poison_kernel, dim3(32768), dim3(64), 0, s);
  }
#endif
  if (!(G == 64 && m->wave_ok) && !m->generic_ok)
    return fail(MYO_E_UNSUPPORTED, "this model (tendon limits / free joint / equalities / plane contacts) needs the wave-per-env kernel (lanes = 64)");
  if (G == 64 && m->wave_ok) {
    // kernel attributes are per device: one flag per device ordinal (a process that opens models on two GPUs)
    static std::mutex attr_mu;
    static bool attr_w_dev[64] = {};
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    bool& attr_w = attr_w_dev[m->device & 63];
    if (!attr_w) {
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 3, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 4, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 3, false, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_w = true;
    }
    const int* order = nullptr;
    int Bn = b->db.B;
    // substep scheduler: MYO_SCHED=1 forces it, 0 disables it; default (auto) uses it where it was measured to pay: when the launch
    // holds at least twice as many envs as the chip holds waves of this kernel (MyoLeg: 8 waves per CU; +6 % at 4096 envs).  With as
    // many waves as envs every wave just re-takes its own env and only the overhead is left (MyoHand at 4096 envs: -15 %)
    static const int sched_mode = [] { const char* e = getenv("MYO_SCHED"); return e ? atoi(e) : -1; }();   // process-wide configuration (environment)
    const int n_cu = m->n_cu > 0 ? m->n_cu : 256;                                                        // per model = per device
    const int resident = n_cu * (m->wave_cfg == 0 ? 16 : (m->wave_cfg == 1 ? 8 : std::max(1, (160 * 1024) / std::max(1, m->env_lds_bytes_w))));
    const bool sched_ok = !kflags && Bn >= 64 && Bn <= SCHED_ENV_MASK && nsub + (wk ? 1 : 0) <= 15 && nsub > 0;
    const bool sched = sched_ok && m->wave_cfg == 1 && !m->rk4 && !(m->dw.hf.on && !m->terrain_sizes) && (sched_mode == 1 || (sched_mode == -1 && m->wave_cfg == 1 && Bn >= 2 * resident));
    if (b->balance && Bn >= 1024 && Bn % 4 == 0 && !kflags && !sched) {
      static const int prio_mode = [] { const char* e = getenv("MYO_PRIO"); return e ? atoi(e) : 2; }();
      hipLaunchKernelGGL(balance_kernel, dim3(1), dim3(1024), 0, s, (const int*)b->db.diag, Bn, b->d_order, Bn / 4, prio_mode);
      order = b->d_order;
    }
    // one queue per XCD the device (or its partition) shows: MI355X has 32 CUs per XCD, 8 XCDs in SPX mode (a queue keyed on
    // XCC_ID & 7 would never be drained on a DPX / QPX / CPX partition)
    const int nqueue = std::max(1, std::min(8, n_cu / 32));
    SchedDev S{b->d_sched, b->d_sched + 32, nqueue == 8 ? b->sched_stride : (Bn + nqueue - 1) / nqueue + 1, nsub + (wk ? 1 : 0), nqueue};
    if (!kflags)   // instantiation chosen below, as rocprofv3 prints it (bench.py reports it next to the kernel time)
      b->last_kernel = m->rk4 ? (m->wave_cfg == 0 ? "step_kernel_w<24,8,32,1,3,false,0,false,false,true>" : "step_kernel_w<36,20,32,2,2,false,0,false,false,true>") :
                       m->wave_cfg == 2 ? "step_kernel_w<36,20,32,2,2,false,0,false,true>" :
                       sched ? (m->dw.hf.on ? "step_kernel_w<36,20,32,2,2,true,3,true>" : (m->leg_sizes ? "step_kernel_w<36,20,32,2,2,true,2,false>" : "step_kernel_w<36,20,32,2,2,true,0,false>"))
                             : (m->wave_cfg == 0 ? (m->hand_sizes ? "step_kernel_w<24,8,32,1,4,false,1,false>" : "step_kernel_w<24,8,32,1,3,false,0,false>")
                                                 : (m->dw.hf.on ? (m->terrain_sizes ? "step_kernel_w<36,20,32,2,2,false,3,true>" : "step_kernel_w<36,20,32,2,2,false,0,true>") : (m->leg_sizes ? "step_kernel_w<36,20,32,2,2,false,2,false>" : "step_kernel_w<36,20,32,2,2,false,0,false>")));
    if (sched) {
      hipLaunchKernelGGL(sched_init_kernel, dim3(1), dim3(1024), 0, s, (const int*)b->db.diag, Bn, S);
      int grid = Bn < resident ? Bn : resident;    // persistent waves: no more workgroups than the chip holds at once
      static const int grid_override = [] { const char* e = getenv("MYO_SCHED_GRID"); return e ? atoi(e) : 0; }();
      if (grid_override > 0 && grid_override < grid) grid = grid_override;
      if (m->dw.hf.on)
        hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, true, 3, true>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, wk, 0, S);
      else if (m->leg_sizes)
        hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, true, 2>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, wk, 0, S);
      else
        hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, true, 0>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, wk, 0, S);
    } else if (m->rk4 && m->wave_cfg == 0)
      hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 3, false, 0, false, false, true>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, (const DevWalk*)nullptr, 0, S);
    else if (m->rk4)
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 0, false, false, true>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    else if (m->wave_cfg == 2)   // TrackEnv model class
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 0, false, true>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    else if (m->wave_cfg == 0 && m->hand_sizes)
      hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 4, false, 1>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, (const DevWalk*)nullptr, 0, S);
    else if (m->wave_cfg == 0)
      hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 3, false, 0>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, (const DevWalk*)nullptr, 0, S);
    else if (m->dw.hf.on && m->terrain_sizes)     // terrain models: the instantiations with the height-field narrow phase
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 3, true>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    else if (m->dw.hf.on)
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 0, true>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    else if (m->leg_sizes)
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 2>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    else
      hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, false, 0>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, wk, kflags, S);
    HIPCHK(hipGetLastError());
    return MYO_OK;
  }
  if (G == 16) hipLaunchKernelGGL(step_kernel<16>, dim3(grid), dim3(64), lds, s, m->dm, b->db, action, actmap, nsub, st);
  else if (G == 32) hipLaunchKernelGGL(step_kernel<32>, dim3(grid), dim3(64), lds, s, m->dm, b->db, action, actmap, nsub, st);
  else hipLaunchKernelGGL(step_kernel<64>, dim3(grid), dim3(64), lds, s, m->dm, b->db, action, actmap, nsub, st);
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

int myo_step(myo_batch* b, const float* action_dev, int actmap, int nsubsteps, void* stream) {
  if (!b || nsubsteps < 0) return fail(MYO_E_ARG, "myo_step: bad arguments");
  HIPCHK(hipSetDevice(b->model->device));
  return launch_step(b, action_dev, actmap, nsubsteps, (hipStream_t)stream);
}

static int launch_obs(myo_batch* b, hipStream_t s, int obs_only = 0, int reset_only = 0) {
  const myo_model* m = b->model;
  int B = b->db.B;
  if (b->task.task == MYO_TASK_WALK) {
    // the walk observation lives in the step kernel: run it with zero substeps as an observation-only pass
    return launch_step(b, nullptr, MYO_ACTMAP_NONE, 0, s, KF_AUX | (obs_only ? KF_OBS_ONLY : 0) | (reset_only ? KF_RESET_ONLY : 0));
  } else if (b->task.task == MYO_TASK_POSE || b->task.task == MYO_TASK_HOLD || b->task.task == MYO_TASK_STAND || b->task.task == MYO_TASK_TRACK) {
    hipLaunchKernelGGL(obs_kernel, dim3(B), dim3(64), 0, s, m->dm, b->db, b->task, obs_only, reset_only);
  } else if (b->task.task == MYO_TASK_REACH) {
    const int EPW = 4;
    hipLaunchKernelGGL(reach_obs_kernel<16>, dim3((B + EPW - 1) / EPW), dim3(64), (size_t)EPW * m->env_lds_bytes, s, m->dm, b->db, b->task, obs_only);
  } else {
    return fail(MYO_E_ARG, "myo_obs: no task configured");
  }
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

int myo_obs(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "myo_obs: null");
  HIPCHK(hipSetDevice(b->model->device));
  return launch_obs(b, (hipStream_t)stream);
}

int myo_obs_only(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "myo_obs_only: null");
  HIPCHK(hipSetDevice(b->model->device));
  return launch_obs(b, (hipStream_t)stream, 1);
}

int myo_batch_set_condition(myo_batch* b, int frame_skip, int epl_actuator, int eip_actuator) {
  if (!b || frame_skip <= 0) return fail(MYO_E_ARG, "myo_batch_set_condition: bad arguments");
  const int nu = b->model->dm.nu;
  if (epl_actuator >= nu || eip_actuator >= nu) return fail(MYO_E_ARG, "myo_batch_set_condition: actuator id out of range");
  b->db.fat_dt = (float)frame_skip * b->model->dm.timestep;
  b->db.reaf_epl = epl_actuator; b->db.reaf_eip = eip_actuator;
  return MYO_OK;
}

int myo_batch_set_fatigue_reset(myo_batch* b, int mode, const float* fatigue_vec) {
  if (!b || mode < 0 || mode > 2 || (mode == 2 && !fatigue_vec)) return fail(MYO_E_ARG, "myo_batch_set_fatigue_reset: mode 0 / 1 / 2 (+ vector)");
  HIPCHK(hipSetDevice(b->model->device));
  if (mode == 2) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(b->d_fatvec, fatigue_vec, (size_t)b->model->dm.nu * 4, hipMemcpyHostToDevice)); }
  b->task.fatigue_mode = mode; b->task.fatigue_vec = mode == 2 ? b->d_fatvec : nullptr;
  return MYO_OK;
}

int myo_obs_reset_only(myo_batch* b, void* stream) {
  if (!b) return fail(MYO_E_ARG, "myo_obs_reset_only: null");
  HIPCHK(hipSetDevice(b->model->device));
  return launch_obs(b, (hipStream_t)stream, 1, 1);
}

int myo_status(myo_batch* b, int32_t* host_flags) {
  if (!b || !host_flags) return fail(MYO_E_ARG, "myo_status: null");
  HIPCHK(hipSetDevice(b->model->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host_flags, b->db.flags, (size_t)b->db.B * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(b->db.flags, 0, (size_t)b->db.B * 4));
  return MYO_OK;
}

int myo_random_action(myo_batch* b, float* action_dev, uint64_t seed, uint64_t step, int env_offset, void* stream) {
  if (!b || !action_dev) return fail(MYO_E_ARG, "myo_random_action: null");
  size_t n = (size_t)b->db.B * b->model->dm.nu;
  hipLaunchKernelGGL(random_action_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, action_dev, b->db.B,
                     b->model->dm.nu, seed, step, env_offset);
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

int myo_set_balance(myo_batch* b, int on) {
  if (!b) return fail(MYO_E_ARG, "null batch");
  b->balance = on;
  return MYO_OK;
}

int myo_set_lanes(int lanes) {
  if (lanes != 16 && lanes != 32 && lanes != 64) return fail(MYO_E_ARG, "lanes per env must be 16, 32 or 64");
  g_lanes = lanes;
  return MYO_OK;
}

/* diagnostic build only (MYO_STAMPS=1): per-workgroup clock64 totals of the 10 stages of the last myo_step */
int myo_read_stamps(myo_batch* b, long long* host, int nwg) {
  if (!b || !host) return fail(MYO_E_ARG, "myo_read_stamps: null");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, b->d_stamps, (size_t)nwg * 12 * sizeof(long long), hipMemcpyDeviceToHost));
  return MYO_STAMPS ? MYO_OK : 1;
}

int myo_sync(void* stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return MYO_OK;
}

int myo_bench_rollout(myo_batch* b, int steps, int nsubsteps, uint64_t seed, int mode, int max_episode_steps, void* stream, float* ms_out) {
  if (!b || steps <= 0) return fail(MYO_E_ARG, "myo_bench_rollout: bad arguments");
  HIPCHK(hipSetDevice(b->model->device));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (!(mode & MYO_BENCH_FRESH_ACTIONS)) { rc = myo_random_action(b, b->d_action, seed, b->bench_step, b->env_offset, stream); if (rc) return rc; }
  // ms_out == NULL: asynchronous -- the launches are only enqueued (a multi-GPU caller interleaves its collective on the same stream)
  // and their kernel event pairs pile up until myo_bench_last_kernel_ms collects them
  const int base = ms_out ? 0 : b->kev_pending;
  while ((int)b->kev.size() < 2 * (base + steps)) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); b->kev.push_back(e); }
  if (ms_out) { b->kev_pending = 0; HIPCHK(hipEventRecord(b->ev0, s)); }
  for (int i = 0; i < steps; i++) {
    if (mode & MYO_BENCH_FRESH_ACTIONS) { rc = myo_random_action(b, b->d_action, seed, b->bench_step++, b->env_offset, stream); if (rc) return rc; }
    HIPCHK(hipEventRecord(b->kev[2 * (base + i)], s));       // brackets the dominant kernel (+ its tiny placement kernel) on its own stream
    rc = launch_step(b, b->d_action, b->task.task == MYO_TASK_TRACK ? MYO_ACTMAP_CTRLRANGE : MYO_ACTMAP_MUSCLE_SIGMOID, nsubsteps, s);
    if (rc) return rc;
    HIPCHK(hipEventRecord(b->kev[2 * (base + i) + 1], s));
    const int tk = b->task.task;
    if (tk == MYO_TASK_TRACK) continue;   // observation, reward, done and the masked reset are the step kernel's own epilogue
    if ((mode & MYO_BENCH_OBS) && (mode & MYO_BENCH_AUTORESET) && max_episode_steps > 0 && (tk == MYO_TASK_POSE || tk == MYO_TASK_HOLD || tk == MYO_TASK_STAND)) {
      // state-only observations: observation + auto-reset + first observation of the new episodes in ONE launch (post_kernel)
      hipLaunchKernelGGL(post_kernel, dim3(b->db.B), dim3(64), 0, s, b->model->dm, b->db, b->task, b->model->nq, b->model->dm.qpos0, seed, b->env_offset, max_episode_steps);
      HIPCHK(hipGetLastError());
      continue;
    }
    if ((mode & MYO_BENCH_OBS) && b->task.task != MYO_TASK_NONE && b->task.task != MYO_TASK_WALK) { rc = launch_obs(b, s); if (rc) return rc; }   // walk: fused into the step launch
    if ((mode & MYO_BENCH_AUTORESET) && max_episode_steps > 0) {
      rc = myo_autoreset(b, max_episode_steps, seed, stream); if (rc) return rc;
      if ((mode & MYO_BENCH_OBS) && b->task.task != MYO_TASK_NONE) { rc = launch_obs(b, s, 1, 1); if (rc) return rc; }
    }
  }
  if (!ms_out) { b->kev_pending = base + steps; return MYO_OK; }
  HIPCHK(hipEventRecord(b->ev1, s));
  HIPCHK(hipEventSynchronize(b->ev1));
  HIPCHK(hipEventElapsedTime(ms_out, b->ev0, b->ev1));
  float tot = 0.f;
  for (int i = 0; i < steps; i++) { float t; HIPCHK(hipEventElapsedTime(&t, b->kev[2 * i], b->kev[2 * i + 1])); tot += t; }
  b->last_kernel_ms = tot;
  return MYO_OK;
}

/* total HIP-event milliseconds spent in the step kernel launches of the last synchronous myo_bench_rollout call, or -- after
 * asynchronous calls (ms_out == NULL) -- of all launches enqueued since the last collection (waits for them) */
const char* myo_bench_last_kernel_name(const myo_batch* b) { return b ? b->last_kernel : ""; }

/* measured VALU issue peak of the device at `waves_per_simd` resident waves per SIMD (1, 2, 4 or 8): wave64 v_fma_f32 instructions per second,
 * chip-wide.  One workgroup of 4 * waves_per_simd waves per CU (the LDS request keeps a second workgroup off the CU). */
int myo_probe_valu(int device, int waves_per_simd, int iters, double* wave_insts_per_s, int* n_cu_out) {
  if (!wave_insts_per_s || !(waves_per_simd >= 1 && waves_per_simd <= 4 || waves_per_simd == 8) || iters < 1) return fail(MYO_E_ARG, "myo_probe_valu: waves_per_simd must be 1..4 or 8");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  const int ncu = prop.multiProcessorCount;
  const size_t lds = 60 * 1024;   // with 160 KB per CU at most two such workgroups fit: grid = ncu keeps it to one in practice, 2 * ncu gives 8 waves / SIMD
  HIPCHK(hipFuncSetAttribute((const void*)valu_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  float* out = nullptr;
  const int wg_per_cu = waves_per_simd == 8 ? 2 : 1;      // 8 waves per SIMD = two 1024-thread workgroups per CU (2 x 60 KB of LDS fit)
  const int threads = 256 * (waves_per_simd / wg_per_cu);
  HIPCHK(hipMalloc(&out, (size_t)2 * ncu * threads * 4));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(valu_probe_kernel, dim3(ncu * wg_per_cu), dim3(threads), lds, 0, out, iters / 8 + 1, 1.0f);   // warm-up (clocks, code cache)
  HIPCHK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(valu_probe_kernel, dim3(ncu * wg_per_cu), dim3(threads), lds, 0, out, iters, 1.0f);
  HIPCHK(hipEventRecord(e1, 0));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipFree(out); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *wave_insts_per_s = (double)ncu * 4.0 * waves_per_simd * (double)iters * 256.0 / ((double)ms * 1e-3);
  if (n_cu_out) *n_cu_out = ncu;
  return MYO_OK;
}

int myo_bench_last_kernel_ms(myo_batch* b, float* ms_out) {
  if (!b || !ms_out) return fail(MYO_E_ARG, "null");
  if (b->kev_pending > 0) {
    HIPCHK(hipSetDevice(b->model->device));
    HIPCHK(hipEventSynchronize(b->kev[2 * b->kev_pending - 1]));
    float tot = 0.f;
    for (int i = 0; i < b->kev_pending; i++) { float t; HIPCHK(hipEventElapsedTime(&t, b->kev[2 * i], b->kev[2 * i + 1])); tot += t; }
    b->last_kernel_ms = tot;
    b->kev_pending = 0;
  }
  *ms_out = b->last_kernel_ms;
  return MYO_OK;
}

struct myo_policy {
  int device = 0;
  PolicyDev pd{};
  std::vector<void*> dev_allocs;
};

int myo_policy_load(int device, int obs_dim, int act_dim, int nlayers, const int* layer_out, const float* obs_mean, const float* obs_std,
                    const float* const* kernels, const float* const* biases, myo_policy** out) {
  if (!out || !layer_out || !obs_mean || !obs_std || !kernels || !biases || obs_dim <= 0 || act_dim <= 0 || nlayers <= 0 || nlayers > 8)
    return fail(MYO_E_ARG, "myo_policy_load: bad arguments");
  if (layer_out[nlayers - 1] != 2 * act_dim) return fail(MYO_E_ARG, "myo_policy_load: last layer must have 2*act_dim outputs (loc, scale)");
  for (int l = 0; l < nlayers; l++) if (layer_out[l] <= 0 || layer_out[l] > 512) return fail(MYO_E_UNSUPPORTED, "myo_policy_load: layer width must be in 1..512");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MYO_E_HIP, "myo_policy_load: no such HIP device");
  HIPCHK(hipSetDevice(device));
  myo_policy* p = new myo_policy();
  p->device = device;
  PolicyDev& P = p->pd;
  P.obs_dim = obs_dim; P.act_dim = act_dim; P.nlayers = nlayers;
  auto up = [&](const float* src, size_t n, const float** dst) -> int {
    void* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return fail(MYO_E_NOMEM, "hipMalloc policy");
    p->dev_allocs.push_back(d);
    if (hipMemcpy(d, src, n * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(MYO_E_HIP, "hipMemcpy policy");
    *dst = (const float*)d;
    return 0;
  };
  int rc = 0, nin = obs_dim;
  if ((rc = up(obs_mean, obs_dim, &P.mean)) || (rc = up(obs_std, obs_dim, &P.std))) { myo_policy_free(p); return rc; }
  for (int l = 0; l < nlayers; l++) {
    P.width[l] = layer_out[l];
    if ((rc = up(kernels[l], (size_t)nin * layer_out[l], &P.W[l])) || (rc = up(biases[l], layer_out[l], &P.b[l]))) { myo_policy_free(p); return rc; }
    nin = layer_out[l];
  }
  *out = p;
  return MYO_OK;
}

void myo_policy_free(myo_policy* p) {
  if (!p) return;
  for (void* d : p->dev_allocs) (void)hipFree(d);
  delete p;
}

int myo_policy_act(myo_policy* p, const float* obs_dev, int B, float* action_dev, int deterministic, uint64_t seed, uint64_t step,
                   int env_offset, void* stream) {
  if (!p || !obs_dev || !action_dev || B <= 0) return fail(MYO_E_ARG, "myo_policy_act: bad arguments");
  HIPCHK(hipSetDevice(p->device));
  int stride = p->pd.obs_dim > POL_MAXW ? p->pd.obs_dim : POL_MAXW;
  for (int l = 0; l < p->pd.nlayers; l++) if (p->pd.width[l] > stride) stride = p->pd.width[l];
  size_t lds = (size_t)2 * POL_ENVS * stride * 4;
  if (lds > 64 * 1024) return fail(MYO_E_UNSUPPORTED, "myo_policy_act: observation too wide for the LDS tile");
  hipLaunchKernelGGL(policy_kernel, dim3((B + POL_ENVS - 1) / POL_ENVS), dim3(POL_ENVS * POL_MAXW), lds, (hipStream_t)stream, p->pd, obs_dev, B,
                     action_dev, deterministic, seed, step, env_offset);
  HIPCHK(hipGetLastError());
  return MYO_OK;
}

}  // extern "C"
