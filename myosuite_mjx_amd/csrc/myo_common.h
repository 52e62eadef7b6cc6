// myo_common.h -- shared definitions: limits, device-side model / batch / task structs, small vector math.
// Part of the single translation unit myo_hip.hip (included there, in this order); not a stand-alone header.
#ifndef MYO_COMMON_H
#define MYO_COMMON_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/myo_hip.h"

#define MINVALF 1e-15f
#define MAXVALF 1e10f
#define MINIMPF 0.0001f
#define MAXIMPF 0.9999f
#define NCON 32   // contact slots per env
#define NCAND 128 // broad-phase survivors per env
#ifndef LS_FLOOR
#define LS_FLOOR 1e-6f   // float32 floor of the line-search slope tolerance, relative to the initial slope
#endif
#ifndef LS_NOISE
#define LS_NOISE 1e-6f   // line-search slope below this fraction of its own cancelling terms = float32 round-off
#endif
#ifndef NEWTON_NOISE
#define NEWTON_NOISE 1e-6f   // Newton stops when the cost improvement is below this fraction of the cost itself (float32 round-off of the cost)
#endif
#ifndef GRAD_NOISE
#define GRAD_NOISE 2e-7f   // Newton stops when the gradient norm is below this fraction of the norm of its cancelling terms
#endif
#define KCMAX 8   // max dofs in a contact pair's jacobian (checked against the model at load)
#define GEOM_SPHERE 2
#define GEOM_CAPSULE 3
#define GEOM_ELLIPSOID 4
#define GEOM_CYLINDER 5

// ------------------------------------------------------------------------------------------------
// device-side model: sizes + device pointers + LDS layout; passed by value as a kernel argument
struct Lay {
  int qpos, qvel, act, ctrl, warm;                       // persistent state
  int lpos, lmat, lquat, axis, anchor;                   // kinematics
  int tJ, tlen, tforce, actdot;                          // tendons / muscles
  int cdof, cinert, crb, cvel, cacc, cfrc;               // spatial dynamics
  int qfa, smooth, qas, qacc, Ma, grad, search, Mv, qfc; // nv-vectors
  int Mp, Hp;                                            // packed lower-triangular matrices
  int gpos, gmat, cand;                                  // collision geoms (world), broad-phase list
  int cdist, cpos, cnrm, cpair, cJ, caref, cD, cjar, cjv, cimp; // contacts and their rows
  int lsign, laref, lD, ljar, ljv;                       // joint-limit rows
  int total;                                             // floats per env (padded)
};

// Model tables are read through pointers stored in these structs.  A pointer loaded from memory has no known address space, and the compiler
// then emits FLAT loads for every table read (address-space check per access, and the loads count on lgkmcnt as well as vmcnt, so every LDS
// wait also waits for them).  In device code the fields are therefore typed as GLOBAL pointers (address space 1): plain global_load with an
// SGPR base.  Host code sees ordinary pointers; the layout is identical.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const int __attribute__((address_space(1)))* gpi;
typedef const float __attribute__((address_space(1)))* gpf;
#else
typedef const int* gpi;
typedef const float* gpf;
#endif

struct DevModel {
  int nl, nlevel, nv, nu, ngt, nseg, maxnnz, nwg, ncg, npair, maxkc, ns, nM;
  int iterations, ls_iterations;
  int disable_contact, disable_limit, disable_ellipsoid;
  float timestep, grav[3], tolerance, ls_tolerance, meaninertia, c0[3], origin[3];
  float newton_scale;   // 1 / (meaninertia * max(nv, 1)): scale of the solver's stopping tests
  gpi level_adr, link_parent, link_dofadr, link_dofnum, child_adr, child, dof_link, dof_type, dof_parent;
  gpi site_link, wg_link, gt_seg_adr, gt_seg_num, gt_dofs, seg, dl, col_adr, col;
  gpi cg_link, cg_type, pair_i, pair_dl;
  gpf link_pos, link_quat, link_mass, link_com, link_inertia, dof_pos, dof_axis, qpos0, dof_damping, dof_armature;
  gpi act_obs;   // [nu] slot of the actuator's activation in the observation's act block (sim.data.act order), -1: stateless actuator
  int na_obs;           // number of stateful (muscle) actuators = MuJoCo's na
  gpf site_lpos, wg_lpos, wg_lmat, wg_radius, seg_div, gt_len0, act, cg_lpos, cg_lmat, cg_size, cg_rbound, pair_f, jl;
  Lay lay;
};

// counter-based RNG (splitmix64 of (seed, stream, counter)) -> U[0,1)
__device__ __host__ inline float u01(uint64_t seed, uint64_t a, uint64_t b) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xBF58476D1CE4E5B9ull * (b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// MyoDM TrackEnv as a task of the TRK step kernel (mjx/myodm_v0.py:185-304; myo_batch_configure_track): action scaling onto the
// actuator control ranges and the reference lookup in the kernel's prologue, observation / reward / done / metrics / masked reset in its
// epilogue.  The reference tables are float64 like the reference's own arrays (the lookup compares rounded times for equality).
struct DevTrack {
  int ref_type;                         // 0 FIXED (one row), 1 RANDOM (two rows: low / high), 2 TRACK (a motion)
  int horizon, robot_horizon, object_horizon, robot_dim, object_dim, has_vel;
  int extrapolate, linear;              // motion_extrapolation; interpolation "linear" instead of the reference's arithmetic
  int autoreset, term_obj, term_pose;
  int max_steps;                        // gym TimeLimit of the registered ids (0: none): truncation flag in the `solved` row, reset with autoreset
  double start_time;
  const double *T, *robot, *robot_vel, *object;
  const float *init_qpos, *lo, *hi;     // [nq] reset pose; [nu] actuator_ctrlrange
  int obj_link, wrist_link;             // links carrying the object body and the wrist (lunate) body
  float obj_p[3], obj_R[9], wrist_p[3]; // xipos / ximat of those bodies inside their link frames
  float lift_z, obj_err_scale, base_err_scale, lift_bonus_mag, qpos_w, qpos_err_scale, qvel_w, qvel_err_scale;
  float obj_fail2, base_fail2, qpos_fail;   // squared thresholds (:236-241 squares the already-unsquared errors once more)
  float w_pose, w_object, w_bonus, w_penalty;
  float* ref;                           // [B][ref_pitch] robot | robot_vel | object rows of the current env step (prologue -> epilogue)
  int ref_pitch;
  float* metrics;                       // [B][4] pose, object, bonus, penalty
  uint64_t seed;                        // RANDOM references: counter RNG keyed by (seed, global env id, env step)
};

// per-batch device pointers (env-major, pitch = row length)
struct DevBatch {
  int B;
  float *qpos, *qvel, *act, *ctrl, *warm, *time, *target, *obs, *reward, *done, *solved, *qacc, *tenlen, *actforce, *sitexpos;
  int *flags, *diag, *elapsed, *episode;
  int* mprw;   // [B][64] MPR warm-start table carried between the substep tasks of the scheduler (word 63: entry count)
  float* fatigue;          // [B][3][nu]: MA, MR, MF of the 3CC-r fatigue model (muscle condition "fatigue")
  float fat_dt;            // its time step = frame_skip * timestep
  int reaf_epl, reaf_eip;  // actuator ids of the EIP -> EPL tendon transfer (muscle condition "reafferentation")
  float* hfield;           // [B][nrow * ncol] height-field elevation per env (terrain models; NULL otherwise)
  float* gsize;            // [B][4] per-env size (3) + bounding radius of ONE collision geom (ObjHoldRandomEnvV0 re-draws the object's size); NULL: none
  int gsize_cg;            // its collision-geom index
  // overflow of the LDS contact table (wave kernel): contacts NC .. NC + NCX - 1 of an env keep their point / normal / jacobian rows in
  // HBM (L2-resident in practice; touched by ~0.4 % of myoHandPoseRandom reset poses), candidates beyond NCAND their pair ids
  float* ovf;              // [B][NCX][ovf_row] floats: dist, pos[3], normal[3], pair id, cJ[3 * KC], dof ids (byte-packed)
  int* ovf_cand;           // [B][NCANDX]
  int ovf_row;             // floats per overflow row (0: no overflow storage, the LDS table is the capacity)
  int ovf_rows;            // rows per env: NCX, or NCX2 for the TRK models whose contacts 64 .. 127 form a second bank (one lane carries two contacts)
  float* linkx;            // [B][12 * nl] link frames of the last substep's position stage (TRK models; NULL otherwise)
  const DevTrack* track;   // MYO_TASK_TRACK configured (TRK models): prologue / epilogue of the step kernel; NULL otherwise
  int env_offset;          // global id of env 0 (RNG streams are keyed by global env id)
};
#define NCX 48      // overflow contact rows ALLOCATED per env; a kernel with NC LDS slots uses 64 - NC of them: 64 contacts in all
#define NCX2 96     // TRK models: 128 contacts, 32 in LDS + 96 rows; rows of the second bank also hold that contact's solver state
#define TRK_STATE 36  // floats of solver state at the end of a TRK overflow row: aref[6] | D mu mu_t D_t kc | jar[6] | jv[6] | 4 force + 7 Hessian coefficients | active-set word | pad, one per lane
#define NCANDX 256  // overflow candidates per env (MyoHand has 289 pairs: NCAND + NCANDX covers every pair)

struct TaskDev {
  int task, frame_skip, reset_random, target_generate, ntarget, ntip, obs_dim, nq;
  int tip_site[8];
  float pose_thd, far_th, near_th, w_pose, w_bonus, w_act_reg, w_penalty, w_reach;
  const float *target_lo, *target_hi, *init_qpos, *jnt_lo, *jnt_hi;
  const float* init_qvel;   // walk task: reset velocity (NULL = zero)
  const float* rnd;         // [4][nq] reset noise lo | hi and clip lo | hi per qpos entry (NULL: none)
  float tip_lpos[3];        // stand task: tip site in the root link's frame
  int gsize_type;           // per-env geom size override: geom type (0: off), size ~ U(gsize_lo, gsize_hi) per axis at every reset
  float gsize_lo[3], gsize_hi[3];
  const float *init_qpos_alt, *init_qvel_alt;   // walk reset_type "random": second keyframe (NULL: off) ...
  float reset_noise_std;                        // ... and the std of the normal noise on qpos (root height / quaternion excepted)
  int fatigue_mode;                             // fatigue compartments at reset: 0 rested, 1 random, 2 vector
  const float* fatigue_vec;
  int terrain, hf_n;        // terrain walk: myo_terrain kind and cells of the elevation grid re-drawn at reset (0: none)
  float terrain_lo, terrain_hi;
};

// walk task (walk_v0.py:WalkEnvV0): its observation needs a forward pass at the post-step state, which the wave kernel
// runs itself as one extra kinematics / tendon / velocity pass after the last substep (no second kernel, no state re-read)
struct DevWalk {
  int obs_dim, hip_period;
  float dt, min_height, max_rot, target_x_vel, target_y_vel;
  float target_rot[4];
  int link_tl, link_tr, link_pel, link_tor;          // links holding talus_l, talus_r, pelvis, torso
  float lpos_tl[3], lpos_tr[3], lpos_pel[3], lquat_tor[4];
  int qadr_hfl, qadr_hfr, qadr_ja[4];
  float w_vel, w_done, w_cyc, w_rot, w_ja;
  float mass_total, static_mcom[3];
  float knee_height;   // > 0 (TerrainEnvV0): also done when COM height - mean feet height falls below it
};
enum { KF_AUX = 1, KF_OBS_ONLY = 2, KF_RESET_ONLY = 4 };   // step_kernel_w flags: observation pass without stepping / without reward / only for just-reset envs

// ------------------------------------------------------------------------------------------------
// small device math
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ float norm3(const float* a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ float normalize3(float* a) {
  float n = norm3(a);
  if (n < MINVALF) { a[0] = 1; a[1] = 0; a[2] = 0; } else { float i = 1.0f / n; a[0] *= i; a[1] *= i; a[2] *= i; }
  return n;
}
__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ void quat2mat(float* R, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void mulquat(float* r, const float* a, const float* b) {
  float t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
__device__ __forceinline__ void matvec(float* r, const float* R, const float* v) {
  float a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
        c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
__device__ __forceinline__ void matTvec(float* r, const float* R, const float* v) {
  float a = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], b = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
        c = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
__device__ __forceinline__ void matmul3(float* C, const float* A, const float* B) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
__device__ __forceinline__ int tri(int i, int j) { return (i * (i + 1)) / 2 + j; }  // j <= i

template <int G> __device__ __forceinline__ float grp_sum(float x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, G);
  return x;
}
template <int G> __device__ __forceinline__ int grp_sumi(int x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, G);
  return x;
}
template <int G> __device__ __forceinline__ int grp_maxi(int x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x = max(x, __shfl_xor(x, m, G));
  return x;
}
#define SYNC() __syncthreads()
// position of this lane among the set lanes of its G-lane group, and the group's set count
template <int G> __device__ __forceinline__ void grp_rank(bool pred, int grp, int sub, int* rank, int* count) {
  unsigned long long bal = __ballot(pred);
  unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
  unsigned long long g = (bal >> (grp * (G & 63))) & gmask;
  *rank = __popcll(g & ((1ull << sub) - 1ull));
  *count = __popcll(g);
}
#ifndef MYO_STAMPS
#define MYO_STAMPS 0
#endif
#if MYO_STAMPS
#define STAMP(k) do { long long t1_ = clock64(); st_acc[k] += t1_ - st_t0; st_t0 = t1_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#define GFOR(i, n) for (int i = sub; i < (n); i += G)

// 10-element spatial inertia times a motion vector (ang, lin)
__device__ __forceinline__ void mul_inert_vec(float* r, const float* i, const float* v) {
  float a0 = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  float a1 = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  float a2 = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  float a3 = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  float a4 = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  float a5 = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
  r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3; r[4] = a4; r[5] = a5;
}
__device__ __forceinline__ void cross_motion(float* r, const float* vel, const float* v) {
  float a[3], b[3];
  cross3(r, vel, v);
  cross3(a, vel, v + 3);
  cross3(b, vel + 3, v);
  r[3] = a[0] + b[0]; r[4] = a[1] + b[1]; r[5] = a[2] + b[2];
}
__device__ __forceinline__ void cross_force(float* r, const float* vel, const float* f) {
  float a[3], b[3];
  cross3(a, vel, f);
  cross3(b, vel + 3, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  cross3(r + 3, vel, f + 3);
}

#endif  // MYO_COMMON_H
