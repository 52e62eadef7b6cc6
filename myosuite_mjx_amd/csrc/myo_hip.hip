// myo_hip.hip -- MI355X (gfx950) batched musculoskeletal stepper: kernels + C ABI (include/myo_hip.h).
//
// Execution model: one environment is stepped by a group of G=16 adjacent
// lanes (one DPP row); a 64-lane wavefront therefore carries 4 environments and a workgroup is
// exactly one wavefront, so every cross-lane hand-off is wave-synchronous.  The whole working set
// of an environment (link frames, sparse tendon Jacobian rows, spatial inertias, mass matrix,
// contact rows, Newton vectors) lives in that group's slice of LDS for all `nsubsteps` substeps;
// HBM is read once (state + action) and written once (state) per env step.  Model constants are
// shared by all lanes and are read through the scalar / L1 caches from `DevModel` tables produced
// by myosuite_mjx_amd/lowering.py.
//
// Physics restated per substep (reference: third-party MuJoCo reached at
// myosuite/physics/mj_sim_scene.py:55; algorithms per MuJoCo documentation [3P]):
//   kinematics -> spatial tendons w/ wrapping -> muscle FLV forces -> CRB mass matrix + RNE bias ->
//   collision (capsule/ellipsoid) -> joint-limit + pyramidal contact rows -> Newton solver ->
//   semi-implicit Euler with implicit joint damping.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

struct DevModel {
  int nl, nlevel, nv, nu, ngt, nseg, maxnnz, nwg, ncg, npair, maxkc, ns, nM;
  int iterations, ls_iterations;
  int disable_contact, disable_limit, disable_ellipsoid;
  float timestep, grav[3], tolerance, ls_tolerance, meaninertia, c0[3], origin[3];
  const int *level_adr, *link_parent, *link_dofadr, *link_dofnum, *child_adr, *child, *dof_link, *dof_type, *dof_parent;
  const int *site_link, *wg_link, *gt_seg_adr, *gt_seg_num, *gt_dofs, *seg, *dl, *col_adr, *col;
  const int *cg_link, *cg_type, *pair_i, *pair_dl;
  const float *link_pos, *link_quat, *link_mass, *link_com, *link_inertia, *dof_pos, *dof_axis, *qpos0, *dof_damping,
      *dof_armature;
  const float *site_lpos, *wg_lpos, *wg_lmat, *wg_radius, *seg_div, *gt_len0, *act, *cg_lpos, *cg_lmat, *cg_size, *cg_rbound,
      *pair_f, *jl;
  Lay lay;
};

// per-batch device pointers (env-major, pitch = row length)
struct DevBatch {
  int B;
  float *qpos, *qvel, *act, *ctrl, *warm, *time, *target, *obs, *reward, *done, *solved, *qacc, *tenlen, *actforce, *sitexpos;
  int *flags, *diag, *elapsed, *episode;
  float* fatigue;          // [B][3][nu]: MA, MR, MF of the 3CC-r fatigue model (muscle condition "fatigue")
  float fat_dt;            // its time step = frame_skip * timestep
  int reaf_epl, reaf_eip;  // actuator ids of the EIP -> EPL tendon transfer (muscle condition "reafferentation")
};

struct TaskDev {
  int task, frame_skip, reset_random, target_generate, ntarget, ntip, obs_dim;
  int tip_site[8];
  float pose_thd, far_th, near_th, w_pose, w_bonus, w_act_reg, w_penalty, w_reach;
  const float *target_lo, *target_hi, *init_qpos, *jnt_lo, *jnt_hi;
  const float* init_qvel;   // walk task: reset velocity (NULL = zero)
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

// ------------------------------------------------------------------------------------------------
// tendon wrapping (2-D circle wrap, inside wrap, sphere / cylinder lifting) -- float twin of the oracle
__device__ __forceinline__ bool is_intersect(const float* p1, const float* p2, const float* p3, const float* p4) {
  float det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1]);
  if (fabsf(det) < MINVALF) return false;
  float a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det;
  float b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det;
  return a >= 0 && a <= 1 && b >= 0 && b <= 1;
}

__device__ float wrap_circle(float* pnt, const float* d, const float* sd, bool has_side, float rad) {
  float sq0 = d[0] * d[0] + d[1] * d[1], sq1 = d[2] * d[2] + d[3] * d[3], sqr = rad * rad;
  if (sq0 < sqr || sq1 < sqr || rad < MINVALF) return -1;
  float dif[2] = {d[2] - d[0], d[3] - d[1]};
  float dd = dif[0] * dif[0] + dif[1] * dif[1];
  if (dd < MINVALF) return -1;
  float a = clipf(-(dif[0] * d[0] + dif[1] * d[1]) / dd, 0.f, 1.f);
  float tmp[2] = {a * dif[0] + d[0], a * dif[1] + d[1]};
  if (tmp[0] * tmp[0] + tmp[1] * tmp[1] > sqr && (!has_side || sd[0] * tmp[0] + sd[1] * tmp[1] >= 0)) return -1;
  float s0 = sqrtf(sq0 - sqr), s1 = sqrtf(sq1 - sqr);
  float sol[2][4], good[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float sgn = i == 0 ? 1.f : -1.f;
    sol[i][0] = (d[0] * sqr + sgn * rad * d[1] * s0) / sq0;
    sol[i][1] = (d[1] * sqr - sgn * rad * d[0] * s0) / sq0;
    sol[i][2] = (d[2] * sqr - sgn * rad * d[3] * s1) / sq1;
    sol[i][3] = (d[3] * sqr + sgn * rad * d[2] * s1) / sq1;
    if (has_side) {
      float t0 = sol[i][0] + sol[i][2], t1 = sol[i][1] + sol[i][3];
      float n = sqrtf(t0 * t0 + t1 * t1);
      if (n > MINVALF) { t0 /= n; t1 /= n; }
      good[i] = t0 * sd[0] + t1 * sd[1];
    } else {
      float t0 = sol[i][0] - sol[i][2], t1 = sol[i][1] - sol[i][3];
      good[i] = -(t0 * t0 + t1 * t1);
    }
    // a grazing solution (tangent points closer than 1e-3 rad) makes the segment-intersection test
    // meaningless in float; skip it there (changes the length by O(r*1e-9), see DESIGN.md "float safeguards")
    float gz0 = sol[i][0] - sol[i][2], gz1 = sol[i][1] - sol[i][3];
    bool grazing = gz0 * gz0 + gz1 * gz1 < 1e-6f * sqr;
    if (!grazing && is_intersect(d, sol[i], d + 2, sol[i] + 2)) good[i] = -10000.f;
  }
  int i = good[0] > good[1] ? 0 : 1;
#pragma unroll
  for (int k = 0; k < 4; k++) pnt[k] = i == 0 ? sol[0][k] : sol[1][k];
  bool grazing = (pnt[0] - pnt[2]) * (pnt[0] - pnt[2]) + (pnt[1] - pnt[3]) * (pnt[1] - pnt[3]) < 1e-6f * sqr;
  if (!grazing && is_intersect(d, pnt, d + 2, pnt + 2)) return -1;
  return rad * acosf(clipf((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqr, -1.f, 1.f));
}

__device__ float wrap_inside(float* pnt, const float* d, float rad) {
  const float zinit = 1.f - 1e-7f, tolerance = 1e-6f;
  float len0 = sqrtf(d[0] * d[0] + d[1] * d[1]), len1 = sqrtf(d[2] * d[2] + d[3] * d[3]);
  float dif[2] = {d[2] - d[0], d[3] - d[1]};
  float dd = dif[0] * dif[0] + dif[1] * dif[1];
  if (len0 <= rad || len1 <= rad || rad < MINVALF || len0 < MINVALF || len1 < MINVALF) return -1;
  if (dd > MINVALF) {
    float a = -(dif[0] * d[0] + dif[1] * d[1]) / dd;
    if (a > 0 && a < 1) {
      float t0 = a * dif[0] + d[0], t1 = a * dif[1] + d[1];
      if (sqrtf(t0 * t0 + t1 * t1) <= rad) return -1;
    }
  }
  pnt[0] = 0.5f * (d[0] + d[2]); pnt[1] = 0.5f * (d[1] + d[3]);
  float n = sqrtf(pnt[0] * pnt[0] + pnt[1] * pnt[1]);
  if (n > MINVALF) { pnt[0] *= rad / n; pnt[1] *= rad / n; }
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  float A = rad / len0, B = rad / len1;
  float cosG = (len0 * len0 + len1 * len1 - dd) / (2 * len0 * len1);
  if (cosG < -1 + MINVALF) return -1;
  if (cosG > 1 - MINVALF) return 0;
  float Gang = acosf(cosG);
  // Newton on theta = asin(z): same root as MuJoCo's iteration in z, but well conditioned in float near z -> 1
  (void)zinit;
  float th = 1.57079632679f - 4.4721360e-4f;
  float sn = sinf(th), f = asinf(A * sn) + asinf(B * sn) - 2 * th + Gang;
  if (f > 0) return 0;
  for (int iter = 0; iter < 20 && fabsf(f) > tolerance; iter++) {
    float cs = cosf(th);
    float df = A * cs / fmaxf(MINVALF, sqrtf(1 - A * A * sn * sn)) + B * cs / fmaxf(MINVALF, sqrtf(1 - B * B * sn * sn)) - 2;
    th = clipf(th - f / df, 1e-6f, 1.57079632679f);
    sn = sinf(th);
    f = asinf(A * sn) + asinf(B * sn) - 2 * th + Gang;
  }
  float vec[2], ang;
  if (d[0] * d[3] - d[1] * d[2] > 0) { vec[0] = d[0] / len0; vec[1] = d[1] / len0; ang = th - asinf(A * sn); }
  else { vec[0] = d[2] / len1; vec[1] = d[3] / len1; ang = th - asinf(B * sn); }
  float sa, ca;
  sincosf(ang, &sa, &ca);
  pnt[0] = rad * (ca * vec[0] - sa * vec[1]);
  pnt[1] = rad * (sa * vec[0] + ca * vec[1]);
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  return 0;
}

// returns wrap length (<0: no wrap); wpnt = two world points
__device__ float wrap_geom(float* wpnt, const float* x0, const float* x1, const float* xpos, const float* xmat, float radius,
                           bool cylinder, const float* side, bool has_side) {
  float p[6], s[3] = {0, 0, 0}, tmp[3], axis[6], d[4], sd[2] = {0, 0}, pnt[4], res[6];
  tmp[0] = x0[0] - xpos[0]; tmp[1] = x0[1] - xpos[1]; tmp[2] = x0[2] - xpos[2];
  matTvec(p, xmat, tmp);
  tmp[0] = x1[0] - xpos[0]; tmp[1] = x1[1] - xpos[1]; tmp[2] = x1[2] - xpos[2];
  matTvec(p + 3, xmat, tmp);
  if (norm3(p) < MINVALF || norm3(p + 3) < MINVALF) return -1;
  if (has_side) {
    tmp[0] = side[0] - xpos[0]; tmp[1] = side[1] - xpos[1]; tmp[2] = side[2] - xpos[2];
    matTvec(s, xmat, tmp);
  }
  if (!cylinder) {
    axis[0] = p[0]; axis[1] = p[1]; axis[2] = p[2];
    normalize3(axis);
    float nrmv[3];
    cross3(nrmv, p, p + 3);
    float nrm = norm3(nrmv);
    if (nrm < MINVALF) {
      int i = 0;
      if (fabsf(axis[1]) > fabsf(axis[0]) && fabsf(axis[1]) > fabsf(axis[2])) i = 1;
      if (fabsf(axis[2]) > fabsf(axis[0]) && fabsf(axis[2]) > fabsf(axis[1])) i = 2;
      float t[3] = {i == 0 ? 0.f : 1.f, i == 1 ? 0.f : 1.f, i == 2 ? 0.f : 1.f};
      cross3(nrmv, axis, t);
      nrm = norm3(nrmv);
    }
    float inv = 1.0f / nrm;
    nrmv[0] *= inv; nrmv[1] *= inv; nrmv[2] *= inv;
    cross3(axis + 3, nrmv, axis);
    normalize3(axis + 3);
    d[0] = dot3(p, axis); d[1] = dot3(p, axis + 3); d[2] = dot3(p + 3, axis); d[3] = dot3(p + 3, axis + 3);
    if (has_side) { sd[0] = dot3(s, axis); sd[1] = dot3(s, axis + 3); }
  } else {
    d[0] = p[0]; d[1] = p[1]; d[2] = p[3]; d[3] = p[4];
    if (has_side) { sd[0] = s[0]; sd[1] = s[1]; }
  }
  float wlen;
  float sdn = sqrtf(sd[0] * sd[0] + sd[1] * sd[1]);
  if (has_side && sdn < radius) {
    wlen = wrap_inside(pnt, d, radius);
  } else {
    if (has_side && sdn > MINVALF) { sd[0] /= sdn; sd[1] /= sdn; }
    wlen = wrap_circle(pnt, d, sd, has_side, radius);
  }
  if (wlen < 0) return -1;
  if (!cylinder) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      res[k] = axis[k] * pnt[0] + axis[3 + k] * pnt[1];
      res[3 + k] = axis[k] * pnt[2] + axis[3 + k] * pnt[3];
    }
  } else {
    float L0 = sqrtf((p[0] - pnt[0]) * (p[0] - pnt[0]) + (p[1] - pnt[1]) * (p[1] - pnt[1]));
    float L1 = sqrtf((p[3] - pnt[2]) * (p[3] - pnt[2]) + (p[4] - pnt[3]) * (p[4] - pnt[3]));
    float tot = L0 + wlen + L1;
    res[0] = pnt[0]; res[1] = pnt[1]; res[3] = pnt[2]; res[4] = pnt[3];
    res[2] = p[2] + (p[5] - p[2]) * L0 / tot;
    res[5] = p[2] + (p[5] - p[2]) * (L0 + wlen) / tot;
    float h = res[5] - res[2];
    wlen = sqrtf(wlen * wlen + h * h);
  }
  matvec(wpnt, xmat, res);
  matvec(wpnt + 3, xmat, res + 3);
#pragma unroll
  for (int k = 0; k < 3; k++) { wpnt[k] += xpos[k]; wpnt[3 + k] += xpos[k]; }
  return wlen;
}

// ------------------------------------------------------------------------------------------------
// muscle model (MuJoCo mju_muscleGain / Bias / Dynamics) on an actuator record (lowering.ACT_FLTS)
__device__ __forceinline__ float muscle_fl(float L, float lmin, float lmax) {
  if (lmin <= L && L <= lmax) {
    float a = 0.5f * (lmin + 1), b = 0.5f * (1 + lmax), x;
    if (L <= a) { x = (L - lmin) / fmaxf(MINVALF, a - lmin); return 0.5f * x * x; }
    else if (L <= 1) { x = (1 - L) / fmaxf(MINVALF, 1 - a); return 1 - 0.5f * x * x; }
    else if (L <= b) { x = (L - 1) / fmaxf(MINVALF, b - 1); return 1 - 0.5f * x * x; }
    else { x = (lmax - L) / fmaxf(MINVALF, lmax - b); return 0.5f * x * x; }
  }
  return 0;
}
__device__ __forceinline__ void muscle(const float* A, float len, float vel, float act, float ctrl, float* force, float* actdot) {
  float r0 = A[0], r1 = A[1], F0 = A[2], lmin = A[3], lmax = A[4], vmax = A[5], fpmax = A[6], fvmax = A[7], lr0 = A[8], lr1 = A[9];
  float L0 = (lr1 - lr0) / fmaxf(MINVALF, r1 - r0);
  float L = r0 + (len - lr0) / fmaxf(MINVALF, L0);
  float V = vel / fmaxf(MINVALF, L0 * vmax);
  float FL = muscle_fl(L, lmin, lmax), FV, y = fvmax - 1;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1) * (V + 1);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / fmaxf(MINVALF, y);
  else FV = fvmax;
  float gain = -F0 * FL * FV;
  float b = 0.5f * (1 + lmax), bias, x;
  if (L <= 1) bias = 0;
  else if (L <= b) { x = (L - 1) / fmaxf(MINVALF, b - 1); bias = -A[15] * fpmax * 0.5f * x * x; }   // A[15]: peak force of biasprm
  else { x = (L - b) / fmaxf(MINVALF, b - 1); bias = -A[15] * fpmax * (0.5f + x); }
  *force = gain * act + bias;
  float cc = clipf(clipf(ctrl, A[12], A[13]), 0.f, 1.f), ac = clipf(act, 0.f, 1.f);
  float tau_act = A[10] * (0.5f + 1.5f * ac), tau_deact = A[11] / (0.5f + 1.5f * ac);
  float dctrl = cc - act;
  *actdot = dctrl / fmaxf(MINVALF, dctrl > 0 ? tau_act : tau_deact);
}

// normalised action -> muscle excitation (BaseV0.step, envs/myo/base_v0.py:83-109), one actuator of one env:
//   sigmoid re-projection (:87-91); muscle condition "fatigue": the excitation becomes the 3CC-r model's active compartment MA after
//   one update with the target load TL = sigmoid(a) (envs/myo/fatigue.py:61-108; F, R, r of :10-18); "reafferentation": EPL is driven
//   by EIP's command and EIP is silenced (:105-109)
__device__ __forceinline__ float action_map(const DevBatch& Bt, const float* __restrict__ actprm, const float* __restrict__ action, int env,
                                            int i, int nu, int actmap) {
  int src = i;
  if (actmap == MYO_ACTMAP_SIGMOID_REAFFERENTATION) { if (i == Bt.reaf_epl) src = Bt.reaf_eip; else if (i == Bt.reaf_eip) return 0.f; }
  float c = action[(size_t)env * nu + src];
  if (actmap == MYO_ACTMAP_NONE) return c;
  c = 1.0f / (1.0f + expf(-5.0f * (c - 0.5f)));
  if (actmap == MYO_ACTMAP_SIGMOID_FATIGUE) {
    float* S = Bt.fatigue + (size_t)env * 3 * nu;
    float MA = S[i], MR = S[nu + i], MF = S[2 * nu + i];
    const float F = 0.00912f, R = 0.1f * 0.00094f, rr = 10.f * 15.f, dt = Bt.fat_dt, TL = c;
    float LD = (0.5f + 1.5f * MA) / actprm[16 * i + 10], LR = (0.5f + 1.5f * MA) / actprm[16 * i + 11];
    float C, rR;
    if (MA < TL) { C = (MR > TL - MA) ? LD * (TL - MA) : LD * MR; rR = R; }
    else { C = LR * (TL - MA); rR = rr * R; }
    float lo = fmaxf(-MA / dt + F * MA, (MR - 1) / dt + rR * MF), hi = fminf((1 - MA) / dt + F * MA, MR / dt + rR * MF);
    C = fminf(fmaxf(C, lo), hi);                    // np.clip(C, lo, hi)
    S[i] = MA + (C - F * MA) * dt;
    S[nu + i] = MR + (-C + rR * MF) * dt;
    S[2 * nu + i] = MF + (F * MA - rR * MF) * dt;
    c = S[i];
  }
  return c;
}

__device__ __forceinline__ float impedance(const float* solimp, float pos, float margin) {
  float dmin = clipf(solimp[0], MINIMPF, MAXIMPF), dmax = clipf(solimp[1], MINIMPF, MAXIMPF);
  float width = fmaxf(MINVALF, solimp[2]), mid = clipf(solimp[3], MINIMPF, MAXIMPF), power = fmaxf(1.f, solimp[4]);
  if (dmin == dmax || width <= MINVALF) return 0.5f * (dmin + dmax);
  float x = fabsf((pos - margin) / width);
  if (x >= 1) return dmax;
  if (x == 0) return dmin;
  float y;
  if (power == 1) y = x;
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
  else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
__device__ __forceinline__ void kbi(float solref0, float solref1, float dmax_in, float timestep, float* K, float* B) {
  float dmax = clipf(dmax_in, MINIMPF, MAXIMPF);
  if (solref0 > 0) {
    float tc = fmaxf(solref0, 2 * timestep);
    *K = 1.0f / fmaxf(MINVALF, dmax * dmax * tc * tc * solref1 * solref1);
    *B = 2.0f / fmaxf(MINVALF, dmax * tc);
  } else {
    *K = -solref0 / fmaxf(MINVALF, dmax * dmax);
    *B = -solref1 / fmaxf(MINVALF, dmax);
  }
}

__device__ __forceinline__ void make_frame(const float* n, float* t1, float* t2) {  // mju_makeFrame
  t1[0] = 0; t1[1] = 0; t1[2] = 0;
  if (n[1] < 0.5f && n[1] > -0.5f) t1[1] = 1; else t1[2] = 1;
  float t = dot3(n, t1);
  t1[0] -= t * n[0]; t1[1] -= t * n[1]; t1[2] -= t * n[2];
  normalize3(t1);
  cross3(t2, n, t1);
}

// ------------------------------------------------------------------------------------------------
// convex collision for ellipsoid pads: margin-inflated MPR (float twin of the oracle's mpr_penetration)
struct CObj { float pos[3], mat[9], size[3]; int type; float margin; };  // by value: keeps everything in registers
// support point of the un-inflated shape in its own frame, for a direction given in that frame
__device__ __forceinline__ void support_local(int type, const float* size, const float* dl, float* pl) {
  if (type == GEOM_ELLIPSOID) {
    float s[3] = {size[0] * dl[0], size[1] * dl[1], size[2] * dl[2]};
    float n = norm3(s);
    float inv = n > MINVALF ? 1.0f / n : 0.f;
    pl[0] = size[0] * s[0] * inv; pl[1] = size[1] * s[1] * inv; pl[2] = size[2] * s[2] * inv;
  } else if (type == GEOM_CYLINDER) {
    float n = sqrtf(dl[0] * dl[0] + dl[1] * dl[1]);
    float inv = n > MINVALF ? size[0] / n : 0.f;
    pl[0] = dl[0] * inv; pl[1] = dl[1] * inv; pl[2] = dl[2] >= 0 ? size[1] : -size[1];
  } else {  // sphere / capsule
    float n = norm3(dl);
    float inv = n > MINVALF ? size[0] / n : 0.f;
    pl[0] = dl[0] * inv; pl[1] = dl[1] * inv; pl[2] = dl[2] * inv;
    if (type == GEOM_CAPSULE) pl[2] += dl[2] >= 0 ? size[1] : -size[1];
  }
}
struct Sup { float v[3], v1[3]; };  // Minkowski point and its witness on obj1 (the witness on obj2 is v1 - v)
// Minkowski-difference support of the two margin-inflated shapes.  Contract of the wave kernel's caller: obj `a` sits in the
// identity frame at the origin (the pair is expressed in geom 1's frame) and `dir` is a unit vector, so a's support needs no
// rotation and the spherical inflation is just +-margin * dir (no norm, no division).
__device__ void mink_support(const CObj& a, const CObj& b, const float* dir, Sup& s) {
  float nd[3] = {-dir[0], -dir[1], -dir[2]}, dl[3], pl[3], w2[3];
  support_local(a.type, a.size, dir, s.v1);
  matTvec(dl, b.mat, nd);
  support_local(b.type, b.size, dl, pl);
  matvec(w2, b.mat, pl);
  const float m2 = a.margin + b.margin;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    s.v1[k] += a.margin * dir[k];
    s.v[k] = s.v1[k] - (w2[k] + b.pos[k]) + b.margin * dir[k];
  }
  (void)m2;
}
__device__ __forceinline__ void portal_dir(const Sup* p, float* dir) {
  float a[3] = {p[2].v[0] - p[1].v[0], p[2].v[1] - p[1].v[1], p[2].v[2] - p[1].v[2]};
  float b[3] = {p[3].v[0] - p[1].v[0], p[3].v[1] - p[1].v[1], p[3].v[2] - p[1].v[2]};
  cross3(dir, a, b);
  normalize3(dir);
}
__device__ __forceinline__ void expand_portal(Sup* p, const Sup& v4) {
  float va[3];
  cross3(va, v4.v, p[0].v);
  if (dot3(p[1].v, va) > 0) { if (dot3(p[2].v, va) > 0) p[1] = v4; else p[3] = v4; }
  else { if (dot3(p[3].v, va) > 0) p[2] = v4; else p[1] = v4; }
}
__device__ bool mpr_penetration(const CObj& o1, const CObj& o2, float tol, int maxit, float* depth, float* dirout, float* posout, int* nsup = nullptr) {
  Sup p[4];
  float dir[3], va[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { p[0].v1[k] = o1.pos[k]; p[0].v[k] = o1.pos[k] - o2.pos[k]; }
  if (norm3(p[0].v) < MINVALF) p[0].v[0] += 1e-5f;
  dir[0] = -p[0].v[0]; dir[1] = -p[0].v[1]; dir[2] = -p[0].v[2];
  normalize3(dir);
  mink_support(o1, o2, dir, p[1]);
  if (dot3(p[1].v, dir) < 0) return false;
  cross3(dir, p[0].v, p[1].v);
  if (norm3(dir) < 1e-12f) {
    *depth = norm3(p[1].v);
#pragma unroll
    for (int k = 0; k < 3; k++) { dirout[k] = p[1].v[k]; posout[k] = p[1].v1[k] - 0.5f * p[1].v[k]; }
    normalize3(dirout);
    return true;
  }
  normalize3(dir);
  mink_support(o1, o2, dir, p[2]);
  if (dot3(p[2].v, dir) < 0) return false;
#pragma unroll
  for (int k = 0; k < 3; k++) { va[k] = p[1].v[k] - p[0].v[k]; vb[k] = p[2].v[k] - p[0].v[k]; }
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, p[0].v) > 0) { Sup t = p[1]; p[1] = p[2]; p[2] = t; dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    mink_support(o1, o2, dir, p[3]);
    if (dot3(p[3].v, dir) < 0) return false;
    bool cont = false;
    cross3(va, p[1].v, p[3].v);
    if (dot3(va, p[0].v) < -MINVALF) { p[2] = p[3]; cont = true; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v);
      if (dot3(va, p[0].v) < -MINVALF) { p[1] = p[3]; cont = true; }
    }
    if (!cont) break;
#pragma unroll
    for (int k = 0; k < 3; k++) { va[k] = p[1].v[k] - p[0].v[k]; vb[k] = p[2].v[k] - p[0].v[k]; }
    cross3(dir, va, vb);
    normalize3(dir);
  }
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    portal_dir(p, dir);
    if (dot3(dir, p[1].v) >= 0) break;
    Sup v4;
    mink_support(o1, o2, dir, v4);
    float dv4 = dot3(v4.v, dir);
    float dmin = fminf(fminf(dv4 - dot3(p[1].v, dir), dv4 - dot3(p[2].v, dir)), dv4 - dot3(p[3].v, dir));
    if (dv4 < 0 || dmin <= tol) return false;
    expand_portal(p, v4);
  }
  Sup v4;
  for (int it = 0;; it++) {
    portal_dir(p, dir);
    mink_support(o1, o2, dir, v4);
    float dv4 = dot3(v4.v, dir);
    float dmin = fminf(fminf(dv4 - dot3(p[1].v, dir), dv4 - dot3(p[2].v, dir)), dv4 - dot3(p[3].v, dir));
    if (dmin <= tol || it > maxit) { if (nsup) *nsup = it; break; }
    expand_portal(p, v4);
  }
  // output from the final support plane (see the oracle's mpr_penetration for the rationale)
  *depth = dot3(v4.v, dir);
  // contact position: barycentric coordinates of the origin in the tetrahedron (v0, portal) (libccd findPos)
  float bw[4], cr[3];
  cross3(cr, p[1].v, p[2].v); bw[0] = dot3(cr, p[3].v);
  cross3(cr, p[3].v, p[2].v); bw[1] = dot3(cr, p[0].v);
  cross3(cr, p[0].v, p[1].v); bw[2] = dot3(cr, p[3].v);
  cross3(cr, p[2].v, p[1].v); bw[3] = dot3(cr, p[0].v);
  float sum = bw[0] + bw[1] + bw[2] + bw[3];
  if (sum <= 0) {
    bw[0] = 0;
    cross3(cr, p[2].v, p[3].v); bw[1] = dot3(cr, dir);
    cross3(cr, p[3].v, p[1].v); bw[2] = dot3(cr, dir);
    cross3(cr, p[1].v, p[2].v); bw[3] = dot3(cr, dir);
    sum = bw[1] + bw[2] + bw[3];
  }
  float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    dirout[k] = dir[k];
    posout[k] = inv * (bw[0] * (p[0].v1[k] - 0.5f * p[0].v[k]) + bw[1] * (p[1].v1[k] - 0.5f * p[1].v[k]) +
                       bw[2] * (p[2].v1[k] - 0.5f * p[2].v[k]) + bw[3] * (p[3].v1[k] - 0.5f * p[3].v[k]));
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// dense packed-lower-triangular Cholesky / solves on G lanes (matrix in LDS); all lanes must call
template <int G> __device__ void chol_packed(float* H, int n, int sub) {
  for (int j = 0; j < n; j++) {
    SYNC();
    float d = sqrtf(fmaxf(H[tri(j, j)], MINVALF));
    float inv = 1.0f / d;
    SYNC();
    for (int i = j + 1 + sub; i < n; i += G) H[tri(i, j)] *= inv;
    if (sub == 0) H[tri(j, j)] = d;
    SYNC();
    for (int i = j + 1 + sub; i < n; i += G) {
      float lij = H[tri(i, j)];
      for (int k = j + 1; k <= i; k++) H[tri(i, k)] -= lij * H[tri(k, j)];
    }
  }
  SYNC();
}
// x <- (L L^T)^-1 x, x in LDS
template <int G> __device__ void chol_solve(const float* L, float* x, int n, int sub) {
  for (int j = 0; j < n; j++) {
    SYNC();
    float xj = x[j] / L[tri(j, j)];
    SYNC();
    if (sub == 0) x[j] = xj;
    for (int i = j + 1 + sub; i < n; i += G) x[i] -= L[tri(i, j)] * xj;
  }
  for (int j = n - 1; j >= 0; j--) {
    SYNC();
    float xj = x[j] / L[tri(j, j)];
    SYNC();
    if (sub == 0) x[j] = xj;
    for (int i = sub; i < j; i += G) x[i] -= L[tri(j, i)] * xj;
  }
  SYNC();
}
// y = M x for packed symmetric M (rows distributed over lanes); y, x in LDS
template <int G> __device__ void symv_packed(const float* Mp, const float* x, float* y, int n, int sub) {
  GFOR(i, n) {
    float s = 0;
    for (int j = 0; j <= i; j++) s += Mp[tri(i, j)] * x[j];
    for (int j = i + 1; j < n; j++) s += Mp[tri(j, i)] * x[j];
    y[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// position stage pieces
template <int G> __device__ void stage_kinematics(const DevModel& M, float* E, int sub) {
  const Lay& Y = M.lay;
  for (int L = 0; L < M.nlevel; L++) {
    int a = M.level_adr[L], b = M.level_adr[L + 1];
    for (int l = a + sub; l < b; l += G) {
      float pos[3], q[4], R[9];
      int par = M.link_parent[l];
      const float* lp = M.link_pos + 3 * l;
      const float* lq = M.link_quat + 4 * l;
      if (par < 0) {
        pos[0] = lp[0]; pos[1] = lp[1]; pos[2] = lp[2];
        q[0] = lq[0]; q[1] = lq[1]; q[2] = lq[2]; q[3] = lq[3];
      } else {
        float v[3];
        matvec(v, E + Y.lmat + 9 * par, lp);
        pos[0] = E[Y.lpos + 3 * par] + v[0]; pos[1] = E[Y.lpos + 3 * par + 1] + v[1]; pos[2] = E[Y.lpos + 3 * par + 2] + v[2];
        mulquat(q, E + Y.lquat + 4 * par, lq);
      }
      int da = M.link_dofadr[l], dn = M.link_dofnum[l];
      for (int k = 0; k < dn; k++) {
        int d = da + k;
        quat2mat(R, q);
        float ax[3], an[3];
        matvec(ax, R, M.dof_axis + 3 * d);
        matvec(an, R, M.dof_pos + 3 * d);
        an[0] += pos[0]; an[1] += pos[1]; an[2] += pos[2];
        E[Y.axis + 3 * d] = ax[0]; E[Y.axis + 3 * d + 1] = ax[1]; E[Y.axis + 3 * d + 2] = ax[2];
        E[Y.anchor + 3 * d] = an[0]; E[Y.anchor + 3 * d + 1] = an[1]; E[Y.anchor + 3 * d + 2] = an[2];
        float ang = E[Y.qpos + d] - M.qpos0[d];
        if (M.dof_type[d] == 3) {
          float s, c;
          sincosf(0.5f * ang, &s, &c);
          float ql[4] = {c, M.dof_axis[3 * d] * s, M.dof_axis[3 * d + 1] * s, M.dof_axis[3 * d + 2] * s};
          mulquat(q, q, ql);
          quat2mat(R, q);
          float v[3];
          matvec(v, R, M.dof_pos + 3 * d);
          pos[0] = an[0] - v[0]; pos[1] = an[1] - v[1]; pos[2] = an[2] - v[2];
        } else {
          pos[0] += ax[0] * ang; pos[1] += ax[1] * ang; pos[2] += ax[2] * ang;
        }
      }
      float n = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      q[0] *= n; q[1] *= n; q[2] *= n; q[3] *= n;
      quat2mat(R, q);
#pragma unroll
      for (int k = 0; k < 3; k++) E[Y.lpos + 3 * l + k] = pos[k];
#pragma unroll
      for (int k = 0; k < 4; k++) E[Y.lquat + 4 * l + k] = q[k];
#pragma unroll
      for (int k = 0; k < 9; k++) E[Y.lmat + 9 * l + k] = R[k];
    }
    SYNC();
  }
}

__device__ __forceinline__ void site_world(const DevModel& M, const float* E, int s, float* out) {
  int l = M.site_link[s];
  const float* lp = M.site_lpos + 3 * s;
  if (l < 0) { out[0] = lp[0]; out[1] = lp[1]; out[2] = lp[2]; return; }
  float v[3];
  matvec(v, E + M.lay.lmat + 9 * l, lp);
  out[0] = E[M.lay.lpos + 3 * l] + v[0]; out[1] = E[M.lay.lpos + 3 * l + 1] + v[1]; out[2] = E[M.lay.lpos + 3 * l + 2] + v[2];
}

// straight tendon piece pa->pb: add its length and its sparse moment arms (dof list adr,n)
__device__ __forceinline__ float add_straight(const DevModel& M, float* E, float* Jrow, const float* pa, const float* pb, int adr,
                                              int n, float invdiv) {
  float dif[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  float dist = norm3(dif);
  if (n > 0 && dist > MINVALF) {
    float inv = 1.0f / dist;
    dif[0] *= inv; dif[1] *= inv; dif[2] *= inv;
    for (int k = 0; k < n; k++) {
      const int* e = M.dl + 3 * (adr + k);
      int d = e[0];
      const float* ax = E + M.lay.axis + 3 * d;
      float col;
      if (M.dof_type[d] == 3) {
        const float* an = E + M.lay.anchor + 3 * d;
        float r[3] = {pb[0] - an[0], pb[1] - an[1], pb[2] - an[2]}, c[3];
        cross3(c, ax, r);
        col = dot3(dif, c);
      } else {
        col = dot3(dif, ax);
      }
      Jrow[e[2]] += (float)e[1] * col * invdiv;
    }
  }
  return dist * invdiv;
}

template <int G> __device__ void stage_tendon(const DevModel& M, float* E, int sub) {
  const Lay& Y = M.lay;
  for (int gt = sub; gt < M.ngt; gt += G) {
    float* Jrow = E + Y.tJ + gt * M.maxnnz;
    for (int k = 0; k < M.maxnnz; k++) Jrow[k] = 0;
    float L = M.gt_len0[gt];   // constant same-link segments, folded at lowering time
    int s0 = M.gt_seg_adr[gt], sn = M.gt_seg_num[gt];
    for (int si = s0; si < s0 + sn; si++) {
      const int* S = M.seg + 12 * si;
      float invdiv = 1.0f / M.seg_div[si];
      float p0[3], p1[3];
      site_world(M, E, S[0], p0);
      site_world(M, E, S[1], p1);
      float wlen = -1, wp[6];
      if (S[2] >= 0) {
        int g = S[2], gl = M.wg_link[g];
        float gpos[3], gmat[9], side[3] = {0, 0, 0};
        if (gl < 0) {
#pragma unroll
          for (int k = 0; k < 3; k++) gpos[k] = M.wg_lpos[3 * g + k];
#pragma unroll
          for (int k = 0; k < 9; k++) gmat[k] = M.wg_lmat[9 * g + k];
        } else {
          float v[3];
          matvec(v, E + Y.lmat + 9 * gl, M.wg_lpos + 3 * g);
#pragma unroll
          for (int k = 0; k < 3; k++) gpos[k] = E[Y.lpos + 3 * gl + k] + v[k];
          matmul3(gmat, E + Y.lmat + 9 * gl, M.wg_lmat + 9 * g);
        }
        if (S[3] >= 0) site_world(M, E, S[3], side);
        wlen = wrap_geom(wp, p0, p1, gpos, gmat, M.wg_radius[g], S[10] != 0, side, S[3] >= 0);
      }
      if (wlen < 0) {
        L += add_straight(M, E, Jrow, p0, p1, S[4], S[5], invdiv);
      } else {
        L += add_straight(M, E, Jrow, p0, wp, S[6], S[7], invdiv);
        L += wlen * invdiv;
        L += add_straight(M, E, Jrow, wp + 3, p1, S[8], S[9], invdiv);
      }
    }
    E[Y.tlen + gt] = L;
    if (gt < M.nu) {
      const float* A = M.act + 16 * gt;
      float vel = 0;
      for (int k = 0; k < M.maxnnz; k++) {
        int d = M.gt_dofs[gt * M.maxnnz + k];
        if (d >= 0) vel += Jrow[k] * E[Y.qvel + d];
      }
      float f, ad;
      muscle(A, A[14] * L, A[14] * vel, E[Y.act + gt], E[Y.ctrl + gt], &f, &ad);
      E[Y.tforce + gt] = f * A[14];
      E[Y.actdot + gt] = ad;
    }
  }
  SYNC();
  GFOR(d, M.nv) {
    float s = 0;
    for (int k = M.col_adr[d]; k < M.col_adr[d + 1]; k++) {
      int t = M.col[2 * k], slot = M.col[2 * k + 1];
      s += E[Y.tJ + t * M.maxnnz + slot] * E[Y.tforce + t];
    }
    E[Y.qfa + d] = s;
  }
}

// composite inertia (CRB) mass matrix + RNE bias; leaves Mp (packed) and smooth = passive - bias + actuator
template <int G> __device__ void stage_dynamics(const DevModel& M, float* E, int sub) {
  const Lay& Y = M.lay;
  GFOR(l, M.nl) {
    const float* R = E + Y.lmat + 9 * l;
    const float* I = M.link_inertia + 6 * l;
    float Il[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, T[9], Iw[9], com[3];
    matmul3(T, R, Il);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Iw[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
    matvec(com, R, M.link_com + 3 * l);
    float mass = M.link_mass[l];
    float dif[3] = {E[Y.lpos + 3 * l] + com[0] - M.c0[0], E[Y.lpos + 3 * l + 1] + com[1] - M.c0[1], E[Y.lpos + 3 * l + 2] + com[2] - M.c0[2]};
    float ci[10];
    ci[0] = Iw[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    ci[1] = Iw[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    ci[2] = Iw[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    ci[3] = Iw[1] - mass * dif[0] * dif[1];
    ci[4] = Iw[2] - mass * dif[0] * dif[2];
    ci[5] = Iw[5] - mass * dif[1] * dif[2];
    ci[6] = mass * dif[0]; ci[7] = mass * dif[1]; ci[8] = mass * dif[2]; ci[9] = mass;
#pragma unroll
    for (int k = 0; k < 10; k++) { E[Y.cinert + 10 * l + k] = ci[k]; E[Y.crb + 10 * l + k] = ci[k]; }
  }
  GFOR(d, M.nv) {
    const float* ax = E + Y.axis + 3 * d;
    float c[6];
    if (M.dof_type[d] == 3) {
      float off[3] = {M.c0[0] - E[Y.anchor + 3 * d], M.c0[1] - E[Y.anchor + 3 * d + 1], M.c0[2] - E[Y.anchor + 3 * d + 2]};
      c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
      cross3(c + 3, ax, off);
    } else {
      c[0] = c[1] = c[2] = 0; c[3] = ax[0]; c[4] = ax[1]; c[5] = ax[2];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) E[Y.cdof + 6 * d + k] = c[k];
  }
  GFOR(i, (M.nv * (M.nv + 1)) / 2) E[Y.Mp + i] = 0;
  SYNC();
  // RNE forward pass, one tree level per phase
  for (int L = 0; L < M.nlevel; L++) {
    int a = M.level_adr[L], b = M.level_adr[L + 1];
    for (int l = a + sub; l < b; l += G) {
      int par = M.link_parent[l];
      float cvel[6], cacc[6];
      if (par < 0) {
        cvel[0] = cvel[1] = cvel[2] = cvel[3] = cvel[4] = cvel[5] = 0;
        cacc[0] = cacc[1] = cacc[2] = 0; cacc[3] = -M.grav[0]; cacc[4] = -M.grav[1]; cacc[5] = -M.grav[2];
      } else {
#pragma unroll
        for (int k = 0; k < 6; k++) { cvel[k] = E[Y.cvel + 6 * par + k]; cacc[k] = E[Y.cacc + 6 * par + k]; }
      }
      int da = M.link_dofadr[l], dn = M.link_dofnum[l];
      for (int j = 0; j < dn; j++) {
        int d = da + j;
        float cd[6], cdd[6], qv = E[Y.qvel + d];
#pragma unroll
        for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
        cross_motion(cdd, cvel, cd);
#pragma unroll
        for (int k = 0; k < 6; k++) { cacc[k] += cdd[k] * qv; cvel[k] += cd[k] * qv; }
      }
      float ci[10], f[6], t[6], t1[6];
#pragma unroll
      for (int k = 0; k < 10; k++) ci[k] = E[Y.cinert + 10 * l + k];
      mul_inert_vec(f, ci, cacc);
      mul_inert_vec(t, ci, cvel);
      cross_force(t1, cvel, t);
#pragma unroll
      for (int k = 0; k < 6; k++) { E[Y.cvel + 6 * l + k] = cvel[k]; E[Y.cacc + 6 * l + k] = cacc[k]; E[Y.cfrc + 6 * l + k] = f[k] + t1[k]; }
    }
    SYNC();
  }
  // backward accumulation of forces and composite inertias
  for (int L = M.nlevel - 2; L >= 0; L--) {
    int a = M.level_adr[L], b = M.level_adr[L + 1];
    for (int l = a + sub; l < b; l += G) {
      for (int ci = M.child_adr[l]; ci < M.child_adr[l + 1]; ci++) {
        int c = M.child[ci];
#pragma unroll
        for (int k = 0; k < 6; k++) E[Y.cfrc + 6 * l + k] += E[Y.cfrc + 6 * c + k];
#pragma unroll
        for (int k = 0; k < 10; k++) E[Y.crb + 10 * l + k] += E[Y.crb + 10 * c + k];
      }
    }
    SYNC();
  }
  GFOR(d, M.nv) {
    int l = M.dof_link[d];
    float cd[6], buf[6], crb[10];
#pragma unroll
    for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
#pragma unroll
    for (int k = 0; k < 10; k++) crb[k] = E[Y.crb + 10 * l + k];
    float bias = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) bias += cd[k] * E[Y.cfrc + 6 * l + k];
    mul_inert_vec(buf, crb, cd);
    int a = d;
    while (a >= 0) {
      float s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) s += E[Y.cdof + 6 * a + k] * buf[k];
      if (a == d) s += M.dof_armature[d];
      E[Y.Mp + tri(d, a)] = s;
      a = M.dof_parent[a];
    }
    E[Y.smooth + d] = -M.dof_damping[d] * E[Y.qvel + d] - bias + E[Y.qfa + d];
  }
  SYNC();
}

// collision: world geom frames, broad phase (bounding spheres), narrow phase -> contact list. returns ncon (group-uniform)
template <int G> __device__ int stage_collision(const DevModel& M, float* E, int sub, int grp, int* flags) {
  const Lay& Y = M.lay;
  if (M.disable_contact) return 0;
  GFOR(g, M.ncg) {
    int l = M.cg_link[g];
    if (l < 0) {
#pragma unroll
      for (int k = 0; k < 3; k++) E[Y.gpos + 3 * g + k] = M.cg_lpos[3 * g + k];
#pragma unroll
      for (int k = 0; k < 9; k++) E[Y.gmat + 9 * g + k] = M.cg_lmat[9 * g + k];
    } else {
      float v[3], R[9];
      matvec(v, E + Y.lmat + 9 * l, M.cg_lpos + 3 * g);
#pragma unroll
      for (int k = 0; k < 3; k++) E[Y.gpos + 3 * g + k] = E[Y.lpos + 3 * l + k] + v[k];
      matmul3(R, E + Y.lmat + 9 * l, M.cg_lmat + 9 * g);
#pragma unroll
      for (int k = 0; k < 9; k++) E[Y.gmat + 9 * g + k] = R[k];
    }
  }
  SYNC();
  int ncand = 0;
  int* cand = (int*)(E + Y.cand);
  for (int base = 0; base < M.npair; base += G) {
    int p = base + sub;
    bool hit = false;
    if (p < M.npair) {
      const int* P = M.pair_i + 6 * p;
      if (!(M.disable_ellipsoid && !P[4])) {
        const float* x1 = E + Y.gpos + 3 * P[0];
        const float* x2 = E + Y.gpos + 3 * P[1];
        float dif[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
        float bound = M.cg_rbound[P[0]] + M.cg_rbound[P[1]] + M.pair_f[12 * p];
        hit = dot3(dif, dif) <= bound * bound;
      }
    }
    int rk, cnt;
    grp_rank<G>(hit, grp, sub, &rk, &cnt);
    int pos = ncand + rk;
    if (hit && pos < NCAND) cand[pos] = p;
    ncand += cnt;
  }
  if (ncand > NCAND) { *flags |= MYO_FLAG_CAND_OVERFLOW; ncand = NCAND; }
  SYNC();
  int ncon = 0;
  int maxc = ncand;
#pragma unroll
  for (int m = 32; m >= G; m >>= 1) maxc = max(maxc, __shfl_xor(maxc, m, 64));  // wave-uniform trip count
  for (int base = 0; base < maxc; base += G) {
    int ci = base + sub;
    bool hit = false;
    float dist = 0, cpos[3] = {0, 0, 0}, nrm[3] = {1, 0, 0};
    int p = -1;
    if (ci < ncand) {
      p = cand[ci];
      const int* P = M.pair_i + 6 * p;
      int g1 = P[0], g2 = P[1];
      float margin = M.pair_f[12 * p];
      const float *x1 = E + Y.gpos + 3 * g1, *x2 = E + Y.gpos + 3 * g2, *R1 = E + Y.gmat + 9 * g1, *R2 = E + Y.gmat + 9 * g2;
      const float *sz1 = M.cg_size + 3 * g1, *sz2 = M.cg_size + 3 * g2;
      if (P[4]) {  // capsule-capsule (mjraw_CapsuleCapsule)
        float a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
        float dif[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]};
        float mb = -dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = 1 - mb * mb, xa, xb;
        if (fabsf(det) >= MINVALF) {
          xa = (u - mb * v) / det;
          xb = (v - mb * u) / det;
          if (xa > sz1[1]) { xa = sz1[1]; xb = v - mb * sz1[1]; }
          else if (xa < -sz1[1]) { xa = -sz1[1]; xb = v + mb * sz1[1]; }
          if (xb > sz2[1]) { xb = sz2[1]; xa = clipf(u - mb * sz2[1], -sz1[1], sz1[1]); }
          else if (xb < -sz2[1]) { xb = -sz2[1]; xa = clipf(u + mb * sz2[1], -sz1[1], sz1[1]); }
        } else {
          xa = clipf(u, -sz1[1], sz1[1]);
          xb = clipf(v - mb * xa, -sz2[1], sz2[1]);
          xa = clipf(u - mb * xb, -sz1[1], sz1[1]);
        }
        float v1[3] = {x1[0] + a1[0] * xa, x1[1] + a1[1] * xa, x1[2] + a1[2] * xa};
        float v2[3] = {x2[0] + a2[0] * xb, x2[1] + a2[1] * xb, x2[2] + a2[2] * xb};
        float dd[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        float cd = norm3(dd);
        if (cd <= margin + sz1[0] + sz2[0]) {
          if (cd < MINVALF) { dd[0] = 1; dd[1] = 0; dd[2] = 0; } else { float inv = 1.0f / cd; dd[0] *= inv; dd[1] *= inv; dd[2] *= inv; }
          dist = cd - sz1[0] - sz2[0];
#pragma unroll
          for (int k = 0; k < 3; k++) { cpos[k] = v1[k] + dd[k] * (sz1[0] + 0.5f * dist); nrm[k] = dd[k]; }
          hit = true;
        }
      } else {
        // MPR in geom1's own frame (identity for obj1, relative pose R1^T R2, R1^T (x2 - x1) for obj2; float resolution ~1e-9 m)
        float rel[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
        CObj o1, o2;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) o2.mat[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
        matTvec(o2.pos, R1, rel);
#pragma unroll
        for (int k = 0; k < 9; k++) o1.mat[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) { o1.pos[k] = 0.f; o1.size[k] = sz1[k]; o2.size[k] = sz2[k]; }
        o1.type = M.cg_type[g1]; o2.type = M.cg_type[g2]; o1.margin = o2.margin = 0.5f * margin;
        float depth, dir[3], pos[3];
        if (mpr_penetration(o1, o2, 1e-8f, 60, &depth, dir, pos)) {
          dist = margin - depth;
          normalize3(dir);
          float dw[3], pw[3];
          matvec(dw, R1, dir);
          matvec(pw, R1, pos);
#pragma unroll
          for (int k = 0; k < 3; k++) { cpos[k] = pw[k] + x1[k]; nrm[k] = dw[k]; }
          hit = true;
        }
      }
      // contacts at or beyond the inclusion margin generate no rows (margin - gap)
      if (hit && !(dist < margin - M.pair_f[12 * p + 1])) hit = false;
    }
    int rk, cnt;
    grp_rank<G>(hit, grp, sub, &rk, &cnt);
    int pos = ncon + rk;
    if (hit && pos < NCON) {
      E[Y.cdist + pos] = dist;
#pragma unroll
      for (int k = 0; k < 3; k++) { E[Y.cpos + 3 * pos + k] = cpos[k]; E[Y.cnrm + 3 * pos + k] = nrm[k]; }
      ((int*)(E + Y.cpair))[pos] = p;
    }
    ncon += cnt;
  }
  if (ncon > NCON) { *flags |= MYO_FLAG_CONTACT_OVERFLOW; ncon = NCON; }
  SYNC();
  return ncon;
}

// constraint rows: joint limits (one row per violated side) and pyramidal contact rows (4 per contact)
template <int G> __device__ void stage_constraints(const DevModel& M, float* E, int sub, int ncon) {
  const Lay& Y = M.lay;
  GFOR(d, M.nv) {
    const float* J = M.jl + 12 * d;
    float sign = 0, aref = 0, D = 0;
    if (J[0] != 0 && !M.disable_limit) {
      float q = E[Y.qpos + d], margin = J[3];
      float dlo = q - J[1], dhi = J[2] - q, dist = 0;
      if (dlo < margin && dlo <= dhi) { sign = 1; dist = dlo; }
      else if (dhi < margin) { sign = -1; dist = dhi; }
      if (sign != 0) {
        float imp = impedance(J + 6, dist, margin), K, B;
        float R = fmaxf(MINVALF, (1 - imp) / imp * J[11]);
        kbi(J[4], J[5], J[7], M.timestep, &K, &B);
        aref = -B * (sign * E[Y.qvel + d]) - K * imp * (dist - margin);
        D = 1.0f / R;
      }
    }
    E[Y.lsign + d] = sign; E[Y.laref + d] = aref; E[Y.lD + d] = D;
  }
  GFOR(c, ncon) {
    int p = ((const int*)(E + Y.cpair))[c];
    const int* P = M.pair_i + 6 * p;
    const float* F = M.pair_f + 12 * p;
    float n[3] = {E[Y.cnrm + 3 * c], E[Y.cnrm + 3 * c + 1], E[Y.cnrm + 3 * c + 2]}, t1[3], t2[3];
    float cp[3] = {E[Y.cpos + 3 * c], E[Y.cpos + 3 * c + 1], E[Y.cpos + 3 * c + 2]};
    make_frame(n, t1, t2);
    float vn = 0, vt1 = 0, vt2 = 0;
    float* cJ = E + Y.cJ + c * 3 * KCMAX;
    for (int k = 0; k < P[3]; k++) {
      int d = M.pair_dl[2 * (P[2] + k)];
      float sg = (float)M.pair_dl[2 * (P[2] + k) + 1];
      const float* ax = E + Y.axis + 3 * d;
      float col[3];
      if (M.dof_type[d] == 3) {
        float r[3] = {cp[0] - E[Y.anchor + 3 * d], cp[1] - E[Y.anchor + 3 * d + 1], cp[2] - E[Y.anchor + 3 * d + 2]};
        cross3(col, ax, r);
      } else { col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2]; }
      float jn = sg * dot3(n, col), j1 = sg * dot3(t1, col), j2 = sg * dot3(t2, col), qv = E[Y.qvel + d];
      cJ[k] = jn; cJ[KCMAX + k] = j1; cJ[2 * KCMAX + k] = j2;
      vn += jn * qv; vt1 += j1 * qv; vt2 += j2 * qv;
    }
    float dist = E[Y.cdist + c], incl = F[0] - F[1], mu = F[2];
    float imp = impedance(F + 6, dist, incl), K, B;
    kbi(F[4], F[5], F[7], M.timestep, &K, &B);
    float R0 = fmaxf(MINVALF, (1 - imp) / imp * F[3] * (1 + mu * mu));
    float Rpy = fmaxf(MINVALF, 2 * mu * mu * R0);
    E[Y.cD + c] = 1.0f / Rpy;
    float pos = -K * imp * (dist - incl);
    E[Y.caref + 4 * c + 0] = -B * (vn + mu * vt1) + pos;
    E[Y.caref + 4 * c + 1] = -B * (vn - mu * vt1) + pos;
    E[Y.caref + 4 * c + 2] = -B * (vn + mu * vt2) + pos;
    E[Y.caref + 4 * c + 3] = -B * (vn - mu * vt2) + pos;
    E[Y.cimp + c] = mu;
  }
  SYNC();
}

// rows' J*x - aref for x in LDS; writes ljar/cjar (or ljv/cjv when dst_is_jv, without subtracting aref)
template <int G> __device__ void rows_apply(const DevModel& M, float* E, int sub, int ncon, const float* x, bool jv) {
  const Lay& Y = M.lay;
  GFOR(d, M.nv) {
    float s = E[Y.lsign + d];
    if (jv) E[Y.ljv + d] = s * x[d]; else E[Y.ljar + d] = s * x[d] - E[Y.laref + d];
  }
  GFOR(c, ncon) {
    int p = ((const int*)(E + Y.cpair))[c];
    const int* P = M.pair_i + 6 * p;
    const float* cJ = E + Y.cJ + c * 3 * KCMAX;
    float an = 0, a1 = 0, a2 = 0, mu = E[Y.cimp + c];
    for (int k = 0; k < P[3]; k++) {
      float xv = x[M.pair_dl[2 * (P[2] + k)]];
      an += cJ[k] * xv; a1 += cJ[KCMAX + k] * xv; a2 += cJ[2 * KCMAX + k] * xv;
    }
    float r[4] = {an + mu * a1, an - mu * a1, an + mu * a2, an - mu * a2};
    if (jv) {
#pragma unroll
      for (int k = 0; k < 4; k++) E[Y.cjv + 4 * c + k] = r[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) E[Y.cjar + 4 * c + k] = r[k] - E[Y.caref + 4 * c + k];
    }
  }
}

// constraint cost of the rows for jar currently in LDS (group-reduced)
template <int G> __device__ float rows_cost(const DevModel& M, const float* E, int sub, int ncon) {
  const Lay& Y = M.lay;
  float c = 0;
  GFOR(d, M.nv) { float j = E[Y.ljar + d]; if (E[Y.lsign + d] != 0 && j < 0) c += 0.5f * E[Y.lD + d] * j * j; }
  GFOR(k, 4 * ncon) { float j = E[Y.cjar + k]; if (j < 0) c += 0.5f * E[Y.cD + (k >> 2)] * j * j; }
  return grp_sum<G>(c);
}

// qfc = J^T f for the current jar (f = -D*jar on active rows); also adds J^T D J (active) into Hp when Hp != null
template <int G> __device__ void rows_force_hessian(const DevModel& M, float* E, int sub, int ncon, int maxncon, bool hess) {
  const Lay& Y = M.lay;
  GFOR(d, M.nv) {
    float s = E[Y.lsign + d], j = E[Y.ljar + d], D = E[Y.lD + d];
    bool act = s != 0 && j < 0;
    E[Y.qfc + d] = act ? -s * D * j : 0.f;
    if (hess && act) E[Y.Hp + tri(d, d)] += D;
  }
  SYNC();
  for (int c = 0; c < maxncon; c++) {
    if (c < ncon) {
      int p = ((const int*)(E + Y.cpair))[c];
      const int* P = M.pair_i + 6 * p;
      int kc = P[3];
      const float* cJ = E + Y.cJ + c * 3 * KCMAX;
      float D = E[Y.cD + c], mu = E[Y.cimp + c];
      float j0 = E[Y.cjar + 4 * c], j1 = E[Y.cjar + 4 * c + 1], j2 = E[Y.cjar + 4 * c + 2], j3 = E[Y.cjar + 4 * c + 3];
      float w0 = j0 < 0 ? D : 0.f, w1 = j1 < 0 ? D : 0.f, w2 = j2 < 0 ? D : 0.f, w3 = j3 < 0 ? D : 0.f;
      float f0 = -w0 * j0, f1 = -w1 * j1, f2 = -w2 * j2, f3 = -w3 * j3;
      float Fn = f0 + f1 + f2 + f3, Ft1 = mu * (f0 - f1), Ft2 = mu * (f2 - f3);
      if (sub < kc) {
        int d = M.pair_dl[2 * (P[2] + sub)];
        E[Y.qfc + d] += Fn * cJ[sub] + Ft1 * cJ[KCMAX + sub] + Ft2 * cJ[2 * KCMAX + sub];
      }
      if (hess) {
        float W = w0 + w1 + w2 + w3, A1 = mu * (w0 - w1), A2 = mu * (w2 - w3), B1 = mu * mu * (w0 + w1), B2 = mu * mu * (w2 + w3);
        for (int t = sub; t < kc * kc; t += G) {
          int a = t / kc, b = t - a * kc;
          int da = M.pair_dl[2 * (P[2] + a)], db = M.pair_dl[2 * (P[2] + b)];
          if (da >= db) {
            float na = cJ[a], nb = cJ[b], ta = cJ[KCMAX + a], tb = cJ[KCMAX + b], ua = cJ[2 * KCMAX + a], ub = cJ[2 * KCMAX + b];
            E[Y.Hp + tri(da, db)] += W * na * nb + A1 * (na * tb + ta * nb) + A2 * (na * ub + ua * nb) + B1 * ta * tb + B2 * ua * ub;
          }
        }
      }
    }
    SYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// the fused env-step kernel
template <int G>
__global__ void __launch_bounds__(64) step_kernel(DevModel M, DevBatch Bt, const float* __restrict__ action, int actmap, int nsub,
                                                  long long* stamps) {
  extern __shared__ __align__(16) float smem[];
#if MYO_STAMPS
  long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long st_t0 = clock64();
#endif
  const Lay& Y = M.lay;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int EPW = 64 / G;
  int env = blockIdx.x * EPW + grp;
  const bool valid = env < Bt.B;
  if (!valid) env = Bt.B - 1;  // duplicate work, never stored
  float* E = smem + grp * Y.total;
  const int nv = M.nv, nu = M.nu;
  // ---- load state, map action to control
  GFOR(i, nv) {
    E[Y.qpos + i] = Bt.qpos[(size_t)env * nv + i];
    E[Y.qvel + i] = Bt.qvel[(size_t)env * nv + i];
    E[Y.warm + i] = Bt.warm[(size_t)env * nv + i];
  }
  GFOR(i, nu) {
    E[Y.act + i] = Bt.act[(size_t)env * nu + i];
    float c;
    if (action) c = action_map(Bt, M.act, action, env, i, nu, actmap);
    else c = Bt.ctrl[(size_t)env * nu + i];
    E[Y.ctrl + i] = c;
  }
  float time = Bt.time[env];
  int flags = 0, d_nefc = 0, d_ncon = 0, d_iter = 0;
  bool alive = true;
  SYNC();
  const float h = M.timestep;
  const float scale = 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1));
  for (int step = 0; step < nsub; step++) {
    // mj_checkPos / mj_checkVel
    {
      int bad = 0;
      GFOR(i, nv) { float a = E[Y.qpos + i], b = E[Y.qvel + i]; if (!(a == a) || fabsf(a) > MAXVALF || !(b == b) || fabsf(b) > MAXVALF) bad = 1; }
      bad = grp_maxi<G>(bad);
      if (bad && alive) { flags |= MYO_FLAG_BAD_STATE; alive = false; }
    }
    STAMP(0);
    stage_kinematics<G>(M, E, sub);
    STAMP(1);
    stage_tendon<G>(M, E, sub);
    SYNC();
    STAMP(2);
    stage_dynamics<G>(M, E, sub);
    STAMP(3);
    int ncon = stage_collision<G>(M, E, sub, grp, &flags);
    STAMP(4);
    stage_constraints<G>(M, E, sub, ncon);
    STAMP(5);
    // ---- unconstrained acceleration: qas = M^-1 smooth
    GFOR(i, (nv * (nv + 1)) / 2) E[Y.Hp + i] = E[Y.Mp + i];
    GFOR(i, nv) E[Y.qas + i] = E[Y.smooth + i];
    chol_packed<G>(E + Y.Hp, nv, sub);
    chol_solve<G>(E + Y.Hp, E + Y.qas, nv, sub);
    STAMP(6);
    // ---- constraint solver
    int nlim = 0;
    GFOR(d, nv) nlim += E[Y.lsign + d] != 0 ? 1 : 0;
    nlim = grp_sumi<G>(nlim);
    int nefc = nlim + 4 * ncon;
    int maxncon = ncon;
#pragma unroll
    for (int m = 32; m >= G; m >>= 1) maxncon = max(maxncon, __shfl_xor(maxncon, m, 64));
    int iters = 0;
    if (__any(nefc > 0)) {
      bool solving = nefc > 0;
      // warmstart: compare cost(qacc_warmstart) with cost(qacc_smooth)
      rows_apply<G>(M, E, sub, ncon, E + Y.warm, false);
      symv_packed<G>(E + Y.Mp, E + Y.warm, E + Y.Ma, nv, sub);
      SYNC();
      float cw = 0;
      GFOR(i, nv) cw += 0.5f * (E[Y.Ma + i] - E[Y.smooth + i]) * (E[Y.warm + i] - E[Y.qas + i]);
      cw = grp_sum<G>(cw) + rows_cost<G>(M, E, sub, ncon);
      SYNC();
      rows_apply<G>(M, E, sub, ncon, E + Y.qas, false);
      SYNC();
      float cs = rows_cost<G>(M, E, sub, ncon);
      bool use_smooth = cw > cs || !(cw == cw);
      GFOR(i, nv) E[Y.qacc + i] = use_smooth ? E[Y.qas + i] : E[Y.warm + i];
      SYNC();
      if (!use_smooth) rows_apply<G>(M, E, sub, ncon, E + Y.qacc, false);  // jar currently holds the qas version
      symv_packed<G>(E + Y.Mp, E + Y.qacc, E + Y.Ma, nv, sub);
      SYNC();
      float cost = 0;
      GFOR(i, nv) cost += 0.5f * (E[Y.Ma + i] - E[Y.smooth + i]) * (E[Y.qacc + i] - E[Y.qas + i]);
      cost = grp_sum<G>(cost) + rows_cost<G>(M, E, sub, ncon);
      for (int it = 0; it < M.iterations; it++) {
        if (!__any(solving)) break;
        // gradient, Hessian, Newton direction
        GFOR(i, (nv * (nv + 1)) / 2) E[Y.Hp + i] = E[Y.Mp + i];
        SYNC();
        rows_force_hessian<G>(M, E, sub, ncon, maxncon, true);
        GFOR(i, nv) { float g = E[Y.Ma + i] - E[Y.smooth + i] - E[Y.qfc + i]; E[Y.grad + i] = g; E[Y.search + i] = -g; }
        chol_packed<G>(E + Y.Hp, nv, sub);
        chol_solve<G>(E + Y.Hp, E + Y.search, nv, sub);
        symv_packed<G>(E + Y.Mp, E + Y.search, E + Y.Mv, nv, sub);
        rows_apply<G>(M, E, sub, ncon, E + Y.search, true);
        SYNC();
        // exact line search on the piecewise-quadratic cost along `search`
        float g1 = 0, g2 = 0, sn = 0;
        GFOR(i, nv) { float s = E[Y.search + i]; g1 += s * (E[Y.Ma + i] - E[Y.smooth + i]); g2 += 0.5f * s * E[Y.Mv + i]; sn += s * s; }
        g1 = grp_sum<G>(g1); g2 = grp_sum<G>(g2); sn = sqrtf(grp_sum<G>(sn));
        float alpha = 0, lo = 0, hi = -1, dlo = 0, d2lo = 0, dhi = 0, d2hi = 0, d1init = 0;
        bool ls_on = solving && sn >= MINVALF;
        for (int lsit = -1; lsit < M.ls_iterations; lsit++) {
          if (!__any(ls_on)) break;
          float a = (lsit < 0) ? 0.f : alpha;
          float d1 = 0, d2 = 0;
          GFOR(d, nv) {
            if (E[Y.lsign + d] != 0) {
              float jv = E[Y.ljv + d], x = E[Y.ljar + d] + a * jv, D = E[Y.lD + d];
              if (x < 0) { d1 += D * x * jv; d2 += D * jv * jv; }
            }
          }
          GFOR(k, 4 * ncon) {
            float jv = E[Y.cjv + k], x = E[Y.cjar + k] + a * jv, D = E[Y.cD + (k >> 2)];
            if (x < 0) { d1 += D * x * jv; d2 += D * jv * jv; }
          }
          d1 = grp_sum<G>(d1) + g1 + 2 * a * g2;
          d2 = grp_sum<G>(d2) + 2 * g2;
          if (!ls_on) continue;
          if (lsit < 0) {
            if (d1 >= 0 || d2 <= 0) { ls_on = false; alpha = 0; continue; }
            dlo = d1; d2lo = d2; d1init = fabsf(d1);
            alpha = -d1 / d2;
            continue;
          }
          float gtol = fmaxf(M.tolerance * M.ls_tolerance * sn / scale, LS_FLOOR * d1init);
          if (fabsf(d1) < gtol) { ls_on = false; continue; }
          if (d1 < 0) { lo = alpha; dlo = d1; d2lo = d2; } else { hi = alpha; dhi = d1; d2hi = d2; }
          float cand = alpha - d1 / d2;
          if (hi < 0) {
            if (!(cand > lo)) { ls_on = false; continue; }
            alpha = cand;
          } else {
            if (!(cand > lo && cand < hi)) {
              float c2 = d1 < 0 ? hi - dhi / d2hi : lo - dlo / d2lo;
              cand = (c2 > lo && c2 < hi) ? c2 : 0.5f * (lo + hi);
            }
            if (cand == alpha || hi - lo <= 1e-7f * hi) { ls_on = false; continue; }
            alpha = cand;
          }
        }
        bool moved = solving && alpha > 0;
        if (solving && !moved) solving = false;
        SYNC();
        if (moved) {
          GFOR(i, nv) { E[Y.qacc + i] += alpha * E[Y.search + i]; E[Y.Ma + i] += alpha * E[Y.Mv + i]; E[Y.ljar + i] += alpha * E[Y.ljv + i]; }
          GFOR(k, 4 * ncon) E[Y.cjar + k] += alpha * E[Y.cjv + k];
        }
        SYNC();
        float newcost = 0;
        GFOR(i, nv) newcost += 0.5f * (E[Y.Ma + i] - E[Y.smooth + i]) * (E[Y.qacc + i] - E[Y.qas + i]);
        newcost = grp_sum<G>(newcost) + rows_cost<G>(M, E, sub, ncon);
        if (moved) {
          float improvement = scale * (cost - newcost);
          cost = newcost;
          iters++;
          float gn = 0;
          GFOR(i, nv) gn += E[Y.grad + i] * E[Y.grad + i];
          gn = scale * sqrtf(grp_sum<G>(gn));
          if (improvement < fmaxf(M.tolerance, 1e-6f * scale * fabsf(cost)) || gn < M.tolerance) solving = false;
        }
      }
      // final constraint force for the converged qacc
      SYNC();
      rows_force_hessian<G>(M, E, sub, ncon, maxncon, false);
      if (nefc == 0) { GFOR(i, nv) { E[Y.qacc + i] = E[Y.qas + i]; E[Y.qfc + i] = 0; } }
    } else {
      GFOR(i, nv) { E[Y.qacc + i] = E[Y.qas + i]; E[Y.qfc + i] = 0; }
    }
    SYNC();
    STAMP(7);
    d_nefc = nefc; d_ncon = ncon; d_iter = max(d_iter, iters);
    // mj_checkAcc
    {
      int bad = 0;
      GFOR(i, nv) { float a = E[Y.qacc + i]; if (!(a == a) || fabsf(a) > MAXVALF) bad = 1; }
      bad = grp_maxi<G>(bad);
      if (bad && alive) { flags |= MYO_FLAG_BAD_QACC; alive = false; }
    }
    // ---- Euler with implicit joint damping: (M + h*B) qaccE = smooth + qfc
    GFOR(i, (nv * (nv + 1)) / 2) E[Y.Hp + i] = E[Y.Mp + i];
    GFOR(i, nv) { E[Y.warm + i] = E[Y.qacc + i]; E[Y.search + i] = E[Y.smooth + i] + E[Y.qfc + i]; }
    SYNC();
    GFOR(i, nv) E[Y.Hp + tri(i, i)] += h * M.dof_damping[i];
    chol_packed<G>(E + Y.Hp, nv, sub);
    chol_solve<G>(E + Y.Hp, E + Y.search, nv, sub);
    if (alive) {
      GFOR(i, nu) E[Y.act + i] += h * E[Y.actdot + i];
      GFOR(i, nv) { float v = E[Y.qvel + i] + h * E[Y.search + i]; E[Y.qvel + i] = v; E[Y.qpos + i] += h * v; }
      time += h;
    }
    SYNC();
    STAMP(8);
  }
  // a bad env is reset like mj_resetData (mj_sim_scene.py:56-61)
  if (!alive) {
    GFOR(i, nv) { E[Y.qpos + i] = M.qpos0[i]; E[Y.qvel + i] = 0; E[Y.warm + i] = 0; }
    GFOR(i, nu) { E[Y.act + i] = 0; E[Y.ctrl + i] = 0; }
    time = 0;
  }
  SYNC();
  if (valid) {
    GFOR(i, nv) {
      Bt.qpos[(size_t)env * nv + i] = E[Y.qpos + i];
      Bt.qvel[(size_t)env * nv + i] = E[Y.qvel + i];
      Bt.warm[(size_t)env * nv + i] = E[Y.warm + i];
      Bt.qacc[(size_t)env * nv + i] = E[Y.qacc + i];
    }
    GFOR(i, nu) {
      Bt.act[(size_t)env * nu + i] = E[Y.act + i];
      Bt.ctrl[(size_t)env * nu + i] = E[Y.ctrl + i];
      Bt.tenlen[(size_t)env * nu + i] = E[Y.tlen + i];
      Bt.actforce[(size_t)env * nu + i] = E[Y.tforce + i];
    }
    if (sub == 0) {
      Bt.time[env] = time;
      Bt.elapsed[env] += 1;
      Bt.flags[env] |= flags;
      Bt.diag[(size_t)env * 8 + 0] = d_nefc; Bt.diag[(size_t)env * 8 + 1] = d_ncon; Bt.diag[(size_t)env * 8 + 2] = d_iter;
    }
  }
#if MYO_STAMPS
  STAMP(9);
  if (stamps && lane == 0) for (int k = 0; k < 12; k++) stamps[(size_t)blockIdx.x * 12 + k] = st_acc[k];
#endif
}

// ================================================================================================
// WAVE-PER-ENV KERNEL (lanes_per_env = 64): one wavefront steps one environment.
//  * per-dof quantities (qacc, M rows, gradient, search direction, limit rows ...) live in the registers of lane = dof;
//    contact rows live in lane = contact; dense Cholesky / triangular solves / M*v run on registers with v_readlane
//    broadcasts -- no barriers, no LDS round trips;
//  * J^T f and J^T D J are scattered with LDS float atomics (one wave => deterministic order);
//  * tendons run lane = segment (wrapping segments first, then straight ones) instead of lane = tendon;
//  * the LDS slice is <= 10 KB so 16 envs (= 16 waves, 4 per SIMD) are resident per CU.
// ================================================================================================
#define NCONW 32
struct LayW {
  int qpos, qvel, act, ctrl, lpos, lmat, axis, anchor, xv, qfc, sq, X;
  int tJ, tlen, tforce, seglen, dlval;            // region X, tendon phase
  int cdof, cinert, crb, cvel, cacc, cfrc;        // region X, dynamics phase
  int gpos, gax, cand, cdist, cpos, cnrm, cpair, cJ, cdofs;  // region X, collision + solver phase
  int Mp;                                                     // packed mass matrix, aliases gpos/gax/cand once the contact rows exist
  int tJp;                                                    // persistent sparse tendon rows (only for models with tendon limits)
  int total;
};
struct DevModelW {
  LayW lay;
  const int *seg_order, *seg_tendon, *gt_dl;
  const float* link_mat0;
  int nwrapseg, ndl, has_tl;
  const float* tl;
  int nq, has_free, neq;          // free-floating root (nq = nv + 1), joint-coupling equalities
  const int *link_free, *dof_qposadr, *eq_i;
  const float* eq_f;
};

__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rdlanei(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror  -> every lane of a 16-lane row holds the row sum
  return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
#define WFOR(i, n) for (int i = lane; i < (n); i += 64)
// dof id k (0..KC-1) of contact c from the byte-packed table (CDW = ints per contact, a constexpr of the kernel)
#define CDOF(E_, Y_, c_, k_) ((int)((((const unsigned int*)((E_) + (Y_).cdofs))[CDW * (c_) + ((k_) >> 2)] >> (8 * ((k_) & 3))) & 255u))

// in: r[k] = H[lane][k] (k <= lane). out: r[k] = L[lane][k], returns 1/L[lane][lane].  All indices are compile-time.
template <int NVT> __device__ __forceinline__ float chol_rows(float (&r)[NVT], int lane) {
  float invd = 1.0f;
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    float s = r[j];
#pragma unroll
    for (int k = 0; k < j; k++) s -= r[k] * rdlane(r[k], j);
    float pj = fmaxf(rdlane(s, j), MINVALF);
    float inv = __builtin_amdgcn_rsqf(pj);   // v_rsq_f32 (1 ulp); pj >= 1e-15, no denormal handling needed
    float dj = pj * inv;
    r[j] = (lane == j) ? dj : s * inv;
    if (lane == j) invd = inv;
  }
  return invd;
}
// x <- (L L^T)^-1 b ; L rows in registers, L^T columns read from the LDS copy T[j*(NVT+1) + lane]
template <int NVT> __device__ __forceinline__ float chol_solve_rows(const float (&r)[NVT], float invd, float b, const float* T, int lane) {
  float y = b;
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    float yj = rdlane(y, j) * rdlane(invd, j);
    y = (lane == j) ? yj : (lane > j ? y - r[j] * yj : y);
  }
  const float* Tc = T + (lane < NVT ? lane : 0);
#pragma unroll
  for (int j = NVT - 1; j >= 0; j--) {
    float xj = rdlane(y, j) * rdlane(invd, j);
    float cj = Tc[j * (NVT + 1)];
    y = (lane == j) ? xj : (lane < j ? y - cj * xj : y);
  }
  return y;
}
// y_lane = sum_k M[lane][k] x_k with M packed lower-triangular in LDS (rows beyond nv read as zero)
template <int NVT> __device__ __forceinline__ float symv_lds(const float* Mp, float x, int lane, int nv) {
  float s = 0;
  const int d = lane < nv ? lane : 0;
  const int based = (d * (d + 1)) / 2;
#pragma unroll
  for (int k = 0; k < NVT; k++) {
    int kk = k < nv ? k : 0;
    int adr = (kk <= d) ? based + kk : (kk * (kk + 1)) / 2 + d;
    float mv = (k < nv && lane < nv) ? Mp[adr] : 0.f;
    s += mv * rdlane(x, k);
  }
  return s;
}

__device__ __forceinline__ void site_world_w(const DevModel& M, const LayW& Y, const float* E, int s, float* out) {
  int l = M.site_link[s];
  const float* lp = M.site_lpos + 3 * s;
  float a = lp[0], b = lp[1], c = lp[2];
  if (l < 0) { out[0] = a; out[1] = b; out[2] = c; return; }
  const float* R = E + Y.lmat + 9 * l;
  const float* P = E + Y.lpos + 3 * l;
  out[0] = P[0] + R[0] * a + R[1] * b + R[2] * c;
  out[1] = P[1] + R[3] * a + R[4] * b + R[5] * c;
  out[2] = P[2] + R[6] * a + R[7] * b + R[8] * c;
}
__device__ __forceinline__ void geom_world_pos(const DevModel& M, const LayW& Y, const float* E, int g, float* out) {
  int l = M.cg_link[g];
  const float* lp = M.cg_lpos + 3 * g;
  float a = lp[0], b = lp[1], c = lp[2];
  if (l < 0) { out[0] = a; out[1] = b; out[2] = c; return; }
  const float* R = E + Y.lmat + 9 * l;
  const float* P = E + Y.lpos + 3 * l;
  out[0] = P[0] + R[0] * a + R[1] * b + R[2] * c;
  out[1] = P[1] + R[3] * a + R[4] * b + R[5] * c;
  out[2] = P[2] + R[6] * a + R[7] * b + R[8] * c;
}
__device__ __forceinline__ void geom_world_mat(const DevModel& M, const LayW& Y, const float* E, int g, float* R) {
  int l = M.cg_link[g];
  if (l < 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = M.cg_lmat[9 * g + k];
  } else {
    matmul3(R, E + Y.lmat + 9 * l, M.cg_lmat + 9 * g);
  }
}
// moment-arm entries of one straight tendon piece into dlval[]
__device__ __forceinline__ float straight_w(const DevModel& M, const LayW& Y, float* E, const float* pa, const float* pb, int adr, int n,
                                            float invdiv, bool active) {
  float dif[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  float dist = norm3(dif);
  float inv = dist > MINVALF ? 1.0f / dist : 0.f;
  dif[0] *= inv; dif[1] *= inv; dif[2] *= inv;
  for (int k = 0; k < n; k++) {
    const int* e = M.dl + 3 * (adr + k);
    int d = e[0];
    const float* ax = E + Y.axis + 3 * d;
    float col;
    if (M.dof_type[d] == 3) {
      const float* an = E + Y.anchor + 3 * d;
      float r[3] = {pb[0] - an[0], pb[1] - an[1], pb[2] - an[2]}, c[3];
      cross3(c, ax, r);
      col = dot3(dif, c);
    } else col = dot3(dif, ax);
    E[Y.dlval + adr + k] = active ? (float)e[1] * col * invdiv : 0.f;
  }
  return active ? dist * invdiv : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Substep-granular dynamic scheduling (opt-in, MYO_SCHED=1).  With one workgroup per env, a launch of B = 4096 envs fills every
// wave slot of the chip exactly once and lasts as long as its slowest SIMD (env work varies +-12 %).  Here the waves are
// persistent instead: the unit of work is ONE substep of one env.  Each XCD owns a FIFO ring of its envs (state stays in that
// XCD's L2); a wave takes a ticket, waits until the ticket's slot is published, loads the env's state, runs the substep, stores
// the state and publishes the env's next substep at the tail.  Envs advance in near lock-step, so the imbalance that is left is
// that of a single substep.  No wave ever waits while it holds work, published work is always held by a running wave, and the
// ticket count is fixed (envs x substeps), so every wave terminates; spins are capped anyway and a timeout raises a flag.
template <bool S> __device__ __forceinline__ float ldstate(const float* p) {
  if (S) return __int_as_float(__hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return *p;
}
template <bool S> __device__ __forceinline__ int ldstatei(const int* p) {
  if (S) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
struct SchedDev {
  int* ctl;      // [8][4]: head (next ticket), tail (next publish index), n (envs of this queue), error
  int* ring;     // [8][stride]: gen << 24 | substep << 20 | env
  int stride, nsubtot;
};
#define SCHED_ENV_MASK 0xFFFFF
__global__ void __launch_bounds__(1024) sched_init_kernel(const int* __restrict__ diag, int B, SchedDev S) {
  __shared__ int hist[256], start[256];
  __shared__ int cmax_s;
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  if (t == 0) cmax_s = 1;
  if (t < 8) { int n = (B - t + 7) / 8; S.ctl[4 * t] = 0; S.ctl[4 * t + 1] = n; S.ctl[4 * t + 2] = n; S.ctl[4 * t + 3] = 0; }
  __syncthreads();
  int cm = 1;
  for (int e = t; e < B; e += 1024) cm = max(cm, diag[(size_t)e * 8 + 3]);
  atomicMax(&cmax_s, cm);
  __syncthreads();
  const int cmax = cmax_s;
  for (int e = t; e < B; e += 1024) atomicAdd(&hist[255 - min(255, (int)(255LL * diag[(size_t)e * 8 + 3] / cmax))], 1);   // bucket 0 = heaviest
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 256; k++) { start[k] = acc; acc += hist[k]; } }
  __syncthreads();
  for (int e = t; e < B; e += 1024) {
    int b = 255 - min(255, (int)(255LL * diag[(size_t)e * 8 + 3] / cmax));
    int r = atomicAdd(&start[b], 1);                  // rank by descending predicted cost: heavy envs are served first
    S.ring[(r & 7) * S.stride + (r >> 3)] = e;        // generation 0, substep 0
  }
}

// table sizes of the compiled config models (after lowering): SPEC = 1 (MyoHand, myohand_pose.xml) and SPEC = 2 (MyoLeg, myolegs.xml)
// instantiations of the wave kernel take their loop bounds from here; SPEC = 0 reads them from the model at run time
template <int SPEC> struct Sizes { static constexpr int nq = 0, nv = 0, nu = 0, nl = 0, nlevel = 0, maxnnz = 0, nseg = 0, ncg = 0, npair = 0; };
template <> struct Sizes<1> { static constexpr int nq = 23, nv = 23, nu = 39, nl = 17, nlevel = 5, maxnnz = 7, nseg = 116, ncg = 27, npair = 289; };
template <> struct Sizes<2> { static constexpr int nq = 35, nv = 34, nu = 80, nl = 13, nlevel = 6, maxnnz = 11, nseg = 100, ncg = 32, npair = 45; };
template <int SPEC> static bool sizes_match(int nq, int nv, int nu, int nl, int nlevel, int maxnnz, int ngt, int nseg, int ncg, int npair) {
  typedef Sizes<SPEC> Z;
  return nq == Z::nq && nv == Z::nv && nu == Z::nu && nl == Z::nl && nlevel == Z::nlevel && maxnnz == Z::maxnnz && ngt == Z::nu && nseg == Z::nseg &&
         ncg == Z::ncg && npair == Z::npair;
}

template <int NVT, int KC, int NC, int NTR, int WPE, bool SCHED, int SPEC>
__global__ void __launch_bounds__(64, WPE) step_kernel_w(const DevModel* __restrict__ Mp, const DevModelW* __restrict__ Wp, DevBatch Bt,
                                                        const float* __restrict__ action, int actmap, int nsub, long long* stamps,
                                                        const int* __restrict__ order, const DevWalk* __restrict__ wk, int kflags, SchedDev S) {
  extern __shared__ __align__(16) float E[];
  // the model structs stay in (scalar-cached) global memory: fields are s_load-ed where they are used instead of
  // pinning ~150 SGPRs for the whole kernel
  const DevModel& M = *Mp;
  const DevModelW& W = *Wp;
  const LayW& Y = W.lay;
  const int lane_id = threadIdx.x;
  // workgroup -> env map: a speed-only placement hint (envs sorted by last step's cost, see balance_kernel); results of an
  // env never depend on which workgroup steps it
  const int oe = (!SCHED && order) ? order[blockIdx.x] : blockIdx.x;
  int env = oe & 0x0FFFFFFF;
  // the four waves of a SIMD come from different cost quartiles (balance_kernel); the predicted-heavy ones get a higher issue
  // priority so that the launch's critical path -- its heaviest waves -- is not slowed down by lighter neighbours that have slack
  if (!SCHED) {
    switch (oe >> 28) {
      case 3: __builtin_amdgcn_s_setprio(3); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: break;
    }
  }
  // SPEC != 0: the model has exactly the table sizes of Sizes<SPEC> (checked by myo_model_load): loop bounds become compile-time
  // constants (+3 % measured on MyoHand); SPEC = 0 reads them from the model
  typedef Sizes<SPEC> Z;
  const int nv = SPEC ? Z::nv : M.nv, nu = SPEC ? Z::nu : M.nu, nq = SPEC ? Z::nq : W.nq;
  const int nl_ = SPEC ? Z::nl : M.nl, nlevel_ = SPEC ? Z::nlevel : M.nlevel, maxnnz_ = SPEC ? Z::maxnnz : M.maxnnz, ngt_ = SPEC ? Z::nu : M.ngt,
            nseg_ = SPEC ? Z::nseg : M.nseg, ncg_ = SPEC ? Z::ncg : M.ncg, npair_ = SPEC ? Z::npair : M.npair;
  constexpr int CDW = (KC + 3) / 4;   // ints per contact holding its KC byte-packed dof ids
  // the small instantiation (hand / finger class) is compiled without the free-joint, equality, plane-contact and condim-1 code;
  // myo_model_load routes any model that needs one of those to the large instantiation
  constexpr bool FULL = NVT > 24;
  const bool has_free = FULL && W.has_free;
  const int neq = FULL ? W.neq : 0;
  const bool walk = FULL && wk != nullptr;   // fused observation / reward pass of the walk task after the last substep
  if (!SCHED && FULL && (kflags & KF_RESET_ONLY) && Bt.elapsed[env] != 0) return;   // wave-uniform: refresh only the envs an auto-reset just touched
#if MYO_STAMPS
  long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long st_t0 = clock64();
#endif
  const int nsubtot = nsub + (walk ? 1 : 0);
  const float h = M.timestep;
  const float scale = 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1));
  const float damping = lane_id < nv ? M.dof_damping[lane_id] : 0.f;
  // scheduler state of this wave: the queue of the XCD it runs on
  const int sq_q = SCHED ? (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7) : 0;
  int* const sq_ctl = SCHED ? S.ctl + 4 * sq_q : nullptr;
  int* const sq_ring = SCHED ? S.ring + (size_t)sq_q * S.stride : nullptr;
  const int sq_n = SCHED ? sq_ctl[2] : 0;
  int last_cost = 0;
  for (;;) {   // task loop: one (env, substep) per pass when SCHED, a single pass over all substeps of this workgroup's env otherwise
  int s0 = 0, s1 = nsubtot;
  if (SCHED) {
    int t = 0;
    if (lane_id == 0) t = atomicAdd(&sq_ctl[0], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= sq_n * nsubtot) break;                       // every ticket of this queue is taken: this wave is done
    const int gen = t / sq_n, slot = t - gen * sq_n;
    int v = 0, spins = 0;
    for (;;) {                                            // wait until the slot of this ticket has been published
      v = __hip_atomic_load(&sq_ring[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 24) == gen || ++spins > (1 << 21)) break;
      __builtin_amdgcn_s_sleep(8);
    }
    v = __builtin_amdgcn_readfirstlane(v);
    if ((v >> 24) != gen) { if (lane_id == 0) { atomicOr(&Bt.flags[0], MYO_FLAG_SCHED_TIMEOUT); sq_ctl[3] = 1; } break; }
    env = v & SCHED_ENV_MASK;
    s0 = (v >> 20) & 15; s1 = s0 + 1;
  }
  // ---- state: LDS copies of what other lanes gather; per-dof / per-actuator scalars stay in registers.  Under the scheduler the
  // rows were written by another CU of this XCD: agent-scope loads read them from L2 instead of a possibly stale L1 line
  float warm = 0.f, qacc = 0.f, actdot[NTR];
#pragma unroll
  for (int r = 0; r < NTR; r++) actdot[r] = 0.f;
  if (lane_id < nq) E[Y.qpos + lane_id] = ldstate<SCHED>(Bt.qpos + (size_t)env * nq + lane_id);
  if (lane_id < nv) {
    E[Y.qvel + lane_id] = ldstate<SCHED>(Bt.qvel + (size_t)env * nv + lane_id);
    warm = ldstate<SCHED>(Bt.warm + (size_t)env * nv + lane_id);
  }
  for (int i = lane_id; i < nu; i += 64) {
    E[Y.act + i] = ldstate<SCHED>(Bt.act + (size_t)env * nu + i);
    float c;
    if (action && s0 == 0) c = action_map(Bt, M.act, action, env, i, nu, actmap);   // the action map runs once per env step
    else c = ldstate<SCHED>(Bt.ctrl + (size_t)env * nu + i);
    E[Y.ctrl + i] = c;
  }
  float time = ldstate<SCHED>(Bt.time + env);
  int flags = 0, d_nefc = 0, d_ncon = 0, d_iter = 0, d_cost = 0;
  int f_cand = 0, f_mpr = 0, f_ncon = 0, f_iter = 0, f_itcon = 0, f_ls = 0, f_fact = 0;   // work features of this env step (placement cost model)
  if (SCHED && s0 > 0) {   // accumulators of the earlier substeps of this env step
    const int* D = Bt.diag + (size_t)env * 8;
    int a2 = ldstatei<SCHED>(D + 2), a4 = ldstatei<SCHED>(D + 4), a5 = ldstatei<SCHED>(D + 5), a6 = ldstatei<SCHED>(D + 6), a7 = ldstatei<SCHED>(D + 7);
    d_nefc = ldstatei<SCHED>(D); d_ncon = ldstatei<SCHED>(D + 1);   // the observation pass has no rows of its own: keep the last substep's
    d_iter = a2; f_cand = a4 & 0xFFFF; f_ncon = a4 >> 16; f_mpr = a5; f_itcon = a6 & 0xFFFF; f_iter = a6 >> 16; f_ls = a7 & 0xFFFF; f_fact = a7 >> 16;
  }
  bool alive = true;
  SYNC();
  for (int step = s0; step < s1; step++) {
    const bool op = walk && step == nsub;   // observation pass: position / velocity stages at the post-step state, then out
    // compiler-only barrier: keeps the (substep-invariant) model-table loads inside the loop body instead of hoisting
    // ~60 values per lane out of it and spilling them to scratch
    asm volatile("" ::: "memory");
    int lane;   // opaque per-iteration copy of the lane id: address arithmetic derived from it cannot be hoisted (and spilled)
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_id));
    {  // mj_checkPos / mj_checkVel
      bool bad = false;
      if (lane < nq) { float a = E[Y.qpos + lane]; bad = !(a == a) || fabsf(a) > MAXVALF; }
      if (lane < nv) { float b = E[Y.qvel + lane]; bad = bad || !(b == b) || fabsf(b) > MAXVALF; }
      if (__any(bad) && alive && !op) { flags |= MYO_FLAG_BAD_STATE; alive = false; }
    }
    STAMP(0);
    // ---------------------------------------------------------------- kinematics (lane = link, level by level)
    for (int L = 0; L < nlevel_; L++) {
      int l = M.level_adr[L] + lane;
      if (l < M.level_adr[L + 1]) {
        float pos[3], R[9];
        int par = M.link_parent[l];
        const float* lp = M.link_pos + 3 * l;
        if (par < 0) {
          pos[0] = lp[0]; pos[1] = lp[1]; pos[2] = lp[2];
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = W.link_mat0[9 * l + k];
        } else {
          float v[3];
          matvec(v, E + Y.lmat + 9 * par, lp);
          pos[0] = E[Y.lpos + 3 * par] + v[0]; pos[1] = E[Y.lpos + 3 * par + 1] + v[1]; pos[2] = E[Y.lpos + 3 * par + 2] + v[2];
          matmul3(R, E + Y.lmat + 9 * par, W.link_mat0 + 9 * l);
        }
        int da = M.link_dofadr[l], dn = M.link_dofnum[l];
        if (has_free && W.link_free[l]) {
          // free joint: pose straight from qpos (position + unit quaternion); its 3 translational dofs act like slides along
          // the world axes and its 3 rotational dofs like hinges about the body axes through the body origin
          int qa = W.dof_qposadr[da];
          pos[0] = E[Y.qpos + qa]; pos[1] = E[Y.qpos + qa + 1]; pos[2] = E[Y.qpos + qa + 2];
          float q[4] = {E[Y.qpos + qa + 3], E[Y.qpos + qa + 4], E[Y.qpos + qa + 5], E[Y.qpos + qa + 6]};
          float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
          q[0] *= qn; q[1] *= qn; q[2] *= qn; q[3] *= qn;
          quat2mat(R, q);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            E[Y.axis + 3 * (da + k)] = k == 0 ? 1.f : 0.f; E[Y.axis + 3 * (da + k) + 1] = k == 1 ? 1.f : 0.f; E[Y.axis + 3 * (da + k) + 2] = k == 2 ? 1.f : 0.f;
            E[Y.axis + 3 * (da + 3 + k)] = R[k]; E[Y.axis + 3 * (da + 3 + k) + 1] = R[3 + k]; E[Y.axis + 3 * (da + 3 + k) + 2] = R[6 + k];
#pragma unroll
            for (int c = 0; c < 3; c++) { E[Y.anchor + 3 * (da + k) + c] = pos[c]; E[Y.anchor + 3 * (da + 3 + k) + c] = pos[c]; }
          }
          dn = 0;
        }
        for (int k = 0; k < dn; k++) {
          int d = da + k;
          const float* al = M.dof_axis + 3 * d;
          float ax[3], an[3];
          matvec(ax, R, al);
          matvec(an, R, M.dof_pos + 3 * d);
          an[0] += pos[0]; an[1] += pos[1]; an[2] += pos[2];
          E[Y.axis + 3 * d] = ax[0]; E[Y.axis + 3 * d + 1] = ax[1]; E[Y.axis + 3 * d + 2] = ax[2];
          E[Y.anchor + 3 * d] = an[0]; E[Y.anchor + 3 * d + 1] = an[1]; E[Y.anchor + 3 * d + 2] = an[2];
          float ang = E[Y.qpos + W.dof_qposadr[d]] - M.qpos0[W.dof_qposadr[d]];
          if (M.dof_type[d] == 3) {
            float sn, cs;
            sincosf(ang, &sn, &cs);
            float oc = 1 - cs, x = al[0], y = al[1], z = al[2];
            float Rj[9] = {cs + oc * x * x, oc * x * y - sn * z, oc * x * z + sn * y, oc * x * y + sn * z, cs + oc * y * y, oc * y * z - sn * x,
                           oc * x * z - sn * y, oc * y * z + sn * x, cs + oc * z * z};
            matmul3(R, R, Rj);
            float v[3];
            matvec(v, R, M.dof_pos + 3 * d);
            pos[0] = an[0] - v[0]; pos[1] = an[1] - v[1]; pos[2] = an[2] - v[2];
          } else {
            pos[0] += ax[0] * ang; pos[1] += ax[1] * ang; pos[2] += ax[2] * ang;
          }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) E[Y.lpos + 3 * l + k] = pos[k];
#pragma unroll
        for (int k = 0; k < 9; k++) E[Y.lmat + 9 * l + k] = R[k];
      }
      SYNC();
    }
    // reference point of the spatial (6-D) quantities: fixed for fixed-base models, the root link's origin for free-floating ones
    const float c0[3] = {has_free ? E[Y.lpos] : M.c0[0], has_free ? E[Y.lpos + 1] : M.c0[1], has_free ? E[Y.lpos + 2] : M.c0[2]};
    STAMP(1);
    // ---------------------------------------------------------------- tendons: lane = segment
    float tlen_r[NTR], tvel_r[NTR];
    for (int base = 0; base < nseg_; base += 64) {
      int idx = base + lane;
      if (idx < nseg_) {
        int si = W.seg_order[idx];
        const int* S = M.seg + 12 * si;
        float invdiv = 1.0f / M.seg_div[si];
        float p0[3], p1[3];
        site_world_w(M, Y, E, S[0], p0);
        site_world_w(M, Y, E, S[1], p1);
        float wlen = -1, wp[6];
        if (S[2] >= 0) {
          int g = S[2], gl = M.wg_link[g];
          float gpos[3], gmat[9], side[3] = {0, 0, 0};
          if (gl < 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) gpos[k] = M.wg_lpos[3 * g + k];
#pragma unroll
            for (int k = 0; k < 9; k++) gmat[k] = M.wg_lmat[9 * g + k];
          } else {
            float v[3];
            matvec(v, E + Y.lmat + 9 * gl, M.wg_lpos + 3 * g);
#pragma unroll
            for (int k = 0; k < 3; k++) gpos[k] = E[Y.lpos + 3 * gl + k] + v[k];
            matmul3(gmat, E + Y.lmat + 9 * gl, M.wg_lmat + 9 * g);
          }
          if (S[3] >= 0) site_world_w(M, Y, E, S[3], side);
          wlen = wrap_geom(wp, p0, p1, gpos, gmat, M.wg_radius[g], S[10] != 0, side, S[3] >= 0);
        }
        bool wr = wlen >= 0;
        float L = straight_w(M, Y, E, p0, p1, S[4], S[5], invdiv, !wr);
        if (S[2] >= 0) {
          L += straight_w(M, Y, E, p0, wp, S[6], S[7], invdiv, wr);
          L += straight_w(M, Y, E, wp + 3, p1, S[8], S[9], invdiv, wr);
          if (wr) L += wlen * invdiv;
        }
        E[Y.seglen + si] = L;
      }
    }
    SYNC();
#pragma unroll
    for (int rr = 0; rr < NTR; rr++) {  // lane = tendon (NTR rounds of 64): gather its segments, then the muscle
      int gt = lane + 64 * rr;
      tlen_r[rr] = 0.f; tvel_r[rr] = 0.f;
      if (gt >= ngt_) continue;
      float* Jrow = E + Y.tJ + gt * maxnnz_;
      for (int k = 0; k < maxnnz_; k++) Jrow[k] = 0;
      float L = M.gt_len0[gt];   // constant same-link segments, folded at lowering time
      for (int si = M.gt_seg_adr[gt]; si < M.gt_seg_adr[gt] + M.gt_seg_num[gt]; si++) L += E[Y.seglen + si];
      int e0 = W.gt_dl[2 * gt], en = W.gt_dl[2 * gt + 1];
      for (int e = e0; e < e0 + en; e++) Jrow[M.dl[3 * e + 2]] += E[Y.dlval + e];
      E[Y.tlen + gt] = L;
      tlen_r[rr] = L;
      float vel = 0;
      for (int k = 0; k < maxnnz_; k++) {
        int d = M.gt_dofs[gt * maxnnz_ + k];
        if (d >= 0) vel += Jrow[k] * E[Y.qvel + d];
        if (W.has_tl) E[Y.tJp + gt * maxnnz_ + k] = Jrow[k];
      }
      tvel_r[rr] = vel;
      if (gt < nu) {
        const float* A = M.act + 16 * gt;
        float f, ad;
        muscle(A, A[14] * L, A[14] * vel, E[Y.act + gt], E[Y.ctrl + gt], &f, &ad);
        actdot[rr] = ad;
        E[Y.tforce + gt] = f * A[14];
      }
    }
    SYNC();
    float qfa = 0.f;
    if (lane < nv) {
      for (int k = M.col_adr[lane]; k < M.col_adr[lane + 1]; k++) {
        int t = M.col[2 * k], slot = M.col[2 * k + 1];
        qfa += E[Y.tJ + t * maxnnz_ + slot] * E[Y.tforce + t];
      }
    }
    if (step == nsub - 1) {   // diagnostics of the last substep
      for (int i = lane; i < nu; i += 64) { Bt.tenlen[(size_t)env * nu + i] = E[Y.tlen + i]; Bt.actforce[(size_t)env * nu + i] = E[Y.tforce + i]; }
    }
    if (FULL && op) {   // walk observation, muscle block (walk_v0.py:283-285,354-361): length, clipped velocity, clipped force / 1000, then act
      float* o = Bt.obs + (size_t)env * wk->obs_dim + (nq - 2 + nv + 16);
#pragma unroll
      for (int rr = 0; rr < NTR; rr++) {
        int gt = lane + 64 * rr;
        if (gt < nu) {
          float g = M.act[16 * gt + 14];
          o[gt] = g * tlen_r[rr];
          o[nu + gt] = clipf(g * tvel_r[rr], -100.f, 100.f);
          o[2 * nu + gt] = clipf(E[Y.tforce + gt] / (g != 0.f ? g : 1.f) * 1e-3f, -100.f, 100.f);
          o[3 * nu + gt] = E[Y.act + gt];
        }
      }
    }
    SYNC();  // region X changes owner: tendon scratch -> spatial dynamics
    STAMP(2);
    // ---------------------------------------------------------------- CRB + RNE (lane = link / dof)
    if (lane < nl_) {
      int l = lane;
      const float* R = E + Y.lmat + 9 * l;
      const float* I = M.link_inertia + 6 * l;
      float Il[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, T[9], Iw[9], com[3];
      matmul3(T, R, Il);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Iw[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
      matvec(com, R, M.link_com + 3 * l);
      float mass = M.link_mass[l];
      float dif[3] = {E[Y.lpos + 3 * l] + com[0] - c0[0], E[Y.lpos + 3 * l + 1] + com[1] - c0[1], E[Y.lpos + 3 * l + 2] + com[2] - c0[2]};
      float ci[10];
      ci[0] = Iw[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
      ci[1] = Iw[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
      ci[2] = Iw[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
      ci[3] = Iw[1] - mass * dif[0] * dif[1];
      ci[4] = Iw[2] - mass * dif[0] * dif[2];
      ci[5] = Iw[5] - mass * dif[1] * dif[2];
      ci[6] = mass * dif[0]; ci[7] = mass * dif[1]; ci[8] = mass * dif[2]; ci[9] = mass;
#pragma unroll
      for (int k = 0; k < 10; k++) { E[Y.cinert + 10 * l + k] = ci[k]; E[Y.crb + 10 * l + k] = ci[k]; }
    }
    if (lane < nv) {
      int d = lane;
      const float* ax = E + Y.axis + 3 * d;
      float c[6];
      if (M.dof_type[d] == 3) {
        float off[3] = {c0[0] - E[Y.anchor + 3 * d], c0[1] - E[Y.anchor + 3 * d + 1], c0[2] - E[Y.anchor + 3 * d + 2]};
        c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
        cross3(c + 3, ax, off);
      } else { c[0] = c[1] = c[2] = 0; c[3] = ax[0]; c[4] = ax[1]; c[5] = ax[2]; }
#pragma unroll
      for (int k = 0; k < 6; k++) E[Y.cdof + 6 * d + k] = c[k];
    }
    WFOR(i, NVT * (NVT + 1)) E[Y.sq + i] = 0;
    SYNC();
    for (int L = 0; L < nlevel_; L++) {
      int l = M.level_adr[L] + lane;
      if (l < M.level_adr[L + 1]) {
        int par = M.link_parent[l];
        float cvel[6], cacc[6];
        if (par < 0) {
          cvel[0] = cvel[1] = cvel[2] = cvel[3] = cvel[4] = cvel[5] = 0;
          cacc[0] = cacc[1] = cacc[2] = 0; cacc[3] = -M.grav[0]; cacc[4] = -M.grav[1]; cacc[5] = -M.grav[2];
        } else {
#pragma unroll
          for (int k = 0; k < 6; k++) { cvel[k] = E[Y.cvel + 6 * par + k]; cacc[k] = E[Y.cacc + 6 * par + k]; }
        }
        int da = M.link_dofadr[l], dn = M.link_dofnum[l];
        const bool isfree = has_free && W.link_free[l];
        float cvel_rot[6];
        for (int j = 0; j < dn; j++) {
          int d = da + j;
          float cd[6], cdd[6], qv = E[Y.qvel + d];
#pragma unroll
          for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
          if (isfree && j == 3) {
#pragma unroll
            for (int k = 0; k < 6; k++) cvel_rot[k] = cvel[k];   // velocity after the translations, before any of the 3 rotations
          }
          cross_motion(cdd, (isfree && j >= 3) ? cvel_rot : cvel, cd);
#pragma unroll
          for (int k = 0; k < 6; k++) { cacc[k] += cdd[k] * qv; cvel[k] += cd[k] * qv; }
        }
        float ci[10], f[6], t[6], t1[6];
#pragma unroll
        for (int k = 0; k < 10; k++) ci[k] = E[Y.cinert + 10 * l + k];
        mul_inert_vec(f, ci, cacc);
        mul_inert_vec(t, ci, cvel);
        cross_force(t1, cvel, t);
#pragma unroll
        for (int k = 0; k < 6; k++) { E[Y.cvel + 6 * l + k] = cvel[k]; E[Y.cacc + 6 * l + k] = cacc[k]; E[Y.cfrc + 6 * l + k] = f[k] + t1[k]; }
      }
      SYNC();
    }
    if (FULL && op) {
      // ---- walk observation / reward (walk_v0.py:268-316, 363-470) from link frames and link velocities of this pass
      float* o = Bt.obs + (size_t)env * wk->obs_dim;
      if (lane < nq - 2) o[lane] = E[Y.qpos + 2 + lane];                    // qpos_without_xy
      if (lane < nv) o[nq - 2 + lane] = E[Y.qvel + lane] * wk->dt;          // qvel * dt
      float mc[3] = {0.f, 0.f, 0.f}, ml = 0.f;
      if (lane < nl_) {
        float cw[3];
        matvec(cw, E + Y.lmat + 9 * lane, M.link_com + 3 * lane);
        ml = M.link_mass[lane];
#pragma unroll
        for (int k = 0; k < 3; k++) mc[k] = ml * (E[Y.lpos + 3 * lane + k] + cw[k]);
      }
      const float mmov = wave_sum(ml);
      const float sx = wave_sum(mc[0]), sy = wave_sum(mc[1]), sz = wave_sum(mc[2]);
      // MuJoCo's cvel is the velocity of the body-fixed point that coincides with the root's subtree COM (COM of the moving bodies)
      const float cm[3] = {sx / mmov, sy / mmov, sz / mmov};
      float mv[2] = {0.f, 0.f};
      if (lane < nl_) {
        const float* cv = E + Y.cvel + 6 * lane;
        float r[3] = {cm[0] - c0[0], cm[1] - c0[1], cm[2] - c0[2]}, wr[3];
        cross3(wr, cv, r);
        mv[0] = ml * (cv[3] + wr[0]); mv[1] = ml * (cv[4] + wr[1]);
      }
      const float cvx = -wave_sum(mv[0]) / wk->mass_total, cvy = -wave_sum(mv[1]) / wk->mass_total;   // walk_v0.py:438-444 (note the minus)
      const float height = (sz + wk->static_mcom[2]) / wk->mass_total;                                  // walk_v0.py:446-450,465-470
      if (lane == 0) {
        const int sb = nq - 2 + nv;
        o[sb] = cvx; o[sb + 1] = cvy;
        float q[4] = {E[Y.qpos + 3], E[Y.qpos + 4], E[Y.qpos + 5], E[Y.qpos + 6]};
        float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        float u[4] = {q[0] * qn, q[1] * qn, q[2] * qn, q[3] * qn};
        const float* t = wk->lquat_tor;
        float tq[4] = {u[0] * t[0] - u[1] * t[1] - u[2] * t[2] - u[3] * t[3], u[0] * t[1] + u[1] * t[0] + u[2] * t[3] - u[3] * t[2],
                       u[0] * t[2] - u[1] * t[3] + u[2] * t[0] + u[3] * t[1], u[0] * t[3] + u[1] * t[2] - u[2] * t[1] + u[3] * t[0]};
        float tn = 1.0f / sqrtf(tq[0] * tq[0] + tq[1] * tq[1] + tq[2] * tq[2] + tq[3] * tq[3]);
        o[sb + 2] = tq[0] * tn; o[sb + 3] = tq[1] * tn; o[sb + 4] = tq[2] * tn; o[sb + 5] = tq[3] * tn;   // torso xquat
        float pl[3], pr[3], pp[3], v[3];
        matvec(v, E + Y.lmat + 9 * wk->link_tl, wk->lpos_tl);
#pragma unroll
        for (int k = 0; k < 3; k++) pl[k] = E[Y.lpos + 3 * wk->link_tl + k] + v[k];
        matvec(v, E + Y.lmat + 9 * wk->link_tr, wk->lpos_tr);
#pragma unroll
        for (int k = 0; k < 3; k++) pr[k] = E[Y.lpos + 3 * wk->link_tr + k] + v[k];
        matvec(v, E + Y.lmat + 9 * wk->link_pel, wk->lpos_pel);
#pragma unroll
        for (int k = 0; k < 3; k++) pp[k] = E[Y.lpos + 3 * wk->link_pel + k] + v[k];
        o[sb + 6] = pl[2]; o[sb + 7] = pr[2];                                    // feet heights (talus_l, talus_r)
        o[sb + 8] = height;
#pragma unroll
        for (int k = 0; k < 3; k++) { o[sb + 9 + k] = pl[k] - pp[k]; o[sb + 12 + k] = pr[k] - pp[k]; }   // feet relative to the pelvis
        const float phase = fmodf((float)Bt.elapsed[env] / (float)wk->hip_period, 1.0f);
        o[sb + 15] = phase;
        if (!(kflags & KF_OBS_ONLY)) {
          float dvy = wk->target_y_vel - cvy, dvx = wk->target_x_vel - cvx;
          float vel_reward = expf(-dvy * dvy) + expf(-dvx * dvx);
          float d0 = 0.8f * cosf(phase * 6.283185307179586f + 3.141592653589793f) - E[Y.qpos + wk->qadr_hfl];
          float d1 = 0.8f * cosf(phase * 6.283185307179586f) - E[Y.qpos + wk->qadr_hfr];
          float cyclic = sqrtf(d0 * d0 + d1 * d1);
          float dq[4] = {q[0] - wk->target_rot[0], q[1] - wk->target_rot[1], q[2] - wk->target_rot[2], q[3] - wk->target_rot[3]};
          float ref_rot = expf(-5.0f * sqrtf(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]));
          float mag = 0.25f * (fabsf(E[Y.qpos + wk->qadr_ja[0]]) + fabsf(E[Y.qpos + wk->qadr_ja[1]]) + fabsf(E[Y.qpos + wk->qadr_ja[2]]) +
                               fabsf(E[Y.qpos + wk->qadr_ja[3]]));
          float ja = expf(-5.0f * mag);
          float r00 = 1.0f - 2.0f * (q[2] * q[2] + q[3] * q[3]) / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
          float done = (height < wk->min_height || fabsf(r00) > wk->max_rot) ? 1.f : 0.f;
          Bt.reward[env] = wk->w_vel * vel_reward + wk->w_done * done + wk->w_cyc * cyclic + wk->w_rot * ref_rot + wk->w_ja * ja;
          Bt.done[env] = done;
          Bt.solved[env] = vel_reward >= 1.0f ? 1.f : 0.f;
        }
      }
      break;
    }
    for (int L = nlevel_ - 2; L >= 0; L--) {
      int l = M.level_adr[L] + lane;
      if (l < M.level_adr[L + 1]) {
        for (int ci = M.child_adr[l]; ci < M.child_adr[l + 1]; ci++) {
          int c = M.child[ci];
#pragma unroll
          for (int k = 0; k < 6; k++) E[Y.cfrc + 6 * l + k] += E[Y.cfrc + 6 * c + k];
#pragma unroll
          for (int k = 0; k < 10; k++) E[Y.crb + 10 * l + k] += E[Y.crb + 10 * c + k];
        }
      }
      SYNC();
    }
    float smooth = 0.f;
    if (lane < nv) {
      int d = lane, l = M.dof_link[d];
      float cd[6], buf[6], crb[10];
#pragma unroll
      for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
#pragma unroll
      for (int k = 0; k < 10; k++) crb[k] = E[Y.crb + 10 * l + k];
      float bias = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) bias += cd[k] * E[Y.cfrc + 6 * l + k];
      mul_inert_vec(buf, crb, cd);
      int a = d;
      while (a >= 0) {
        float sdot = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sdot += E[Y.cdof + 6 * a + k] * buf[k];
        if (a == d) sdot += M.dof_armature[d];
        E[Y.sq + d * (NVT + 1) + a] = sdot;   // full symmetric copy: (d,a) and (a,d)
        E[Y.sq + a * (NVT + 1) + d] = sdot;
        a = M.dof_parent[a];
      }
      smooth = -damping * E[Y.qvel + d] - bias + qfa;
    }
    SYNC();  // region X changes owner: dynamics scratch -> collision / contact rows
    STAMP(3);
    // ---------------------------------------------------------------- collision (geom frames computed on the fly)
    int ncon = 0;
    if (!M.disable_contact) {
      int ncand = 0;
      int* cand = (int*)(E + Y.cand);
      if (lane < ncg_) {   // world centre and long axis (3rd column) of every collision geom
        float x[3], R[9];
        geom_world_pos(M, Y, E, lane, x);
        geom_world_mat(M, Y, E, lane, R);
        E[Y.gpos + 3 * lane] = x[0]; E[Y.gpos + 3 * lane + 1] = x[1]; E[Y.gpos + 3 * lane + 2] = x[2];
        E[Y.gax + 3 * lane] = R[2]; E[Y.gax + 3 * lane + 1] = R[5]; E[Y.gax + 3 * lane + 2] = R[8];
      }
      SYNC();
      for (int base = 0; base < npair_; base += 64) {
        int p = base + lane;
        bool hit = false;
        if (p < npair_) {
          const int* P = M.pair_i + 6 * p;
          if (!(M.disable_ellipsoid && P[4] == 0)) {
            int g1 = P[0], g2 = P[1];
            const float *x1 = E + Y.gpos + 3 * g1, *x2 = E + Y.gpos + 3 * g2;
            float dif[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
            float bound = M.cg_rbound[g1] + M.cg_rbound[g2] + M.pair_f[12 * p];
            if (FULL && P[4] >= 2) hit = dot3(dif, E + Y.gax + 3 * g1) <= M.cg_rbound[g2] + M.pair_f[12 * p];   // plane: signed distance of the bounding sphere
            else hit = dot3(dif, dif) <= bound * bound;
            if (hit && !P[4]) {
              // conservative refinement before the expensive MPR: replace a capsule's bounding sphere by the distance
              // from the other geom's centre to the capsule's SEGMENT (a bound on the true distance, never excludes a contact)
              float b1 = M.cg_rbound[g1], b2 = M.cg_rbound[g2];
              float c1[3] = {x1[0], x1[1], x1[2]}, c2[3] = {x2[0], x2[1], x2[2]};
              if (M.cg_type[g1] == GEOM_CAPSULE) {
                const float* a = E + Y.gax + 3 * g1;
                float hh = M.cg_size[3 * g1 + 1], t = clipf(dot3(dif, a), -hh, hh);
                c1[0] += t * a[0]; c1[1] += t * a[1]; c1[2] += t * a[2];
                b1 = M.cg_size[3 * g1];
              }
              if (M.cg_type[g2] == GEOM_CAPSULE) {
                const float* a = E + Y.gax + 3 * g2;
                float nd[3] = {c1[0] - x2[0], c1[1] - x2[1], c1[2] - x2[2]};
                float hh = M.cg_size[3 * g2 + 1], t = clipf(dot3(nd, a), -hh, hh);
                c2[0] += t * a[0]; c2[1] += t * a[1]; c2[2] += t * a[2];
                b2 = M.cg_size[3 * g2];
              }
              float d2[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
              float bb = b1 + b2 + M.pair_f[12 * p];
              hit = dot3(d2, d2) <= bb * bb;
              if (hit) {
                // separating-axis test along the centre line: the two (margin-inflated) convex shapes cannot touch if their
                // support widths along that axis do not reach across the centre distance.  MPR would report "no contact" for
                // exactly these pairs, after a dozen support evaluations; this costs one support width per shape
                float dn = norm3(dif);
                if (dn > MINVALF) {
                  float inv = 1.0f / dn, ax[3] = {dif[0] * inv, dif[1] * inv, dif[2] * inv}, wsum = M.pair_f[12 * p];
#pragma unroll
                  for (int side = 0; side < 2; side++) {
                    const int g = side ? g2 : g1;
                    const float* sz = M.cg_size + 3 * g;
                    const int ty = M.cg_type[g];
                    if (ty == GEOM_CAPSULE) wsum += sz[0] + sz[1] * fabsf(dot3(E + Y.gax + 3 * g, ax));
                    else if (ty == GEOM_SPHERE) wsum += sz[0];
                    else {
                      float R[9], dl[3];
                      geom_world_mat(M, Y, E, g, R);
                      matTvec(dl, R, ax);
                      if (ty == GEOM_ELLIPSOID) { float sv[3] = {sz[0] * dl[0], sz[1] * dl[1], sz[2] * dl[2]}; wsum += norm3(sv); }
                      else wsum += sz[0] * sqrtf(dl[0] * dl[0] + dl[1] * dl[1]) + sz[1] * fabsf(dl[2]);   // cylinder
                    }
                  }
                  hit = dn <= wsum * 1.0001f + 1e-6f;   // conservative: never excludes a touching pair
                }
              }
            }
          }
        }
        unsigned long long bal = __ballot(hit);
        int pos = ncand + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < NCAND) cand[pos] = p;
        ncand += __popcll(bal);
      }
      if (ncand > NCAND) { flags |= MYO_FLAG_CAND_OVERFLOW; ncand = NCAND; }
      f_cand += ncand;
      SYNC();
      STAMP(6);
      for (int base = 0; base < ncand; base += 64) {
        int ci = base + lane;
        int nsup = -8;                    // support evaluations of this lane's MPR refinement (-8: not an MPR pair)
        bool hit = false, hit2 = false;   // a plane-capsule pair can give two contacts (one per end sphere)
        float dist = 0, dist2 = 0, cpos[3] = {0, 0, 0}, cpos2[3] = {0, 0, 0}, nrm[3] = {1, 0, 0};
        int p = -1;
        if (ci < ncand) {
          p = cand[ci];
          const int* P = M.pair_i + 6 * p;
          int g1 = P[0], g2 = P[1];
          float margin = M.pair_f[12 * p];
          const float *x1 = E + Y.gpos + 3 * g1, *x2 = E + Y.gpos + 3 * g2;
          const float *sz1 = M.cg_size + 3 * g1, *sz2 = M.cg_size + 3 * g2;
          if (P[4] == 1) {
            const float *a1 = E + Y.gax + 3 * g1, *a2 = E + Y.gax + 3 * g2;
            float dif[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]};
            float mb = -dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = 1 - mb * mb, xa, xb;
            if (fabsf(det) >= MINVALF) {
              xa = (u - mb * v) / det;
              xb = (v - mb * u) / det;
              if (xa > sz1[1]) { xa = sz1[1]; xb = v - mb * sz1[1]; }
              else if (xa < -sz1[1]) { xa = -sz1[1]; xb = v + mb * sz1[1]; }
              if (xb > sz2[1]) { xb = sz2[1]; xa = clipf(u - mb * sz2[1], -sz1[1], sz1[1]); }
              else if (xb < -sz2[1]) { xb = -sz2[1]; xa = clipf(u + mb * sz2[1], -sz1[1], sz1[1]); }
            } else {
              xa = clipf(u, -sz1[1], sz1[1]);
              xb = clipf(v - mb * xa, -sz2[1], sz2[1]);
              xa = clipf(u - mb * xb, -sz1[1], sz1[1]);
            }
            float v1[3] = {x1[0] + a1[0] * xa, x1[1] + a1[1] * xa, x1[2] + a1[2] * xa};
            float v2[3] = {x2[0] + a2[0] * xb, x2[1] + a2[1] * xb, x2[2] + a2[2] * xb};
            float dd[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
            float cd = norm3(dd);
            if (cd <= margin + sz1[0] + sz2[0]) {
              if (cd < MINVALF) { dd[0] = 1; dd[1] = 0; dd[2] = 0; } else { float inv = 1.0f / cd; dd[0] *= inv; dd[1] *= inv; dd[2] *= inv; }
              dist = cd - sz1[0] - sz2[0];
#pragma unroll
              for (int k = 0; k < 3; k++) { cpos[k] = v1[k] + dd[k] * (sz1[0] + 0.5f * dist); nrm[k] = dd[k]; }
              hit = true;
            }
          } else if (FULL && P[4] == 2) {   // plane - capsule (mjc_PlaneCapsule): the two end spheres against the plane
            const float *n = E + Y.gax + 3 * g1, *ax = E + Y.gax + 3 * g2;
            float r = sz2[0], hh = sz2[1];
#pragma unroll
            for (int k = 0; k < 3; k++) nrm[k] = n[k];
            float eA[3] = {x2[0] - hh * ax[0] - x1[0], x2[1] - hh * ax[1] - x1[1], x2[2] - hh * ax[2] - x1[2]};
            float eB[3] = {x2[0] + hh * ax[0] - x1[0], x2[1] + hh * ax[1] - x1[1], x2[2] + hh * ax[2] - x1[2]};
            float dA = dot3(eA, n) - r, dB = dot3(eB, n) - r;
            if (dA <= margin) {
              hit = true; dist = dA;
#pragma unroll
              for (int k = 0; k < 3; k++) cpos[k] = eA[k] + x1[k] - n[k] * (r + 0.5f * dA);
            }
            if (dB <= margin) {
              hit2 = true; dist2 = dB;
#pragma unroll
              for (int k = 0; k < 3; k++) cpos2[k] = eB[k] + x1[k] - n[k] * (r + 0.5f * dB);
            }
          } else if (FULL && P[4] == 3) {   // plane - ellipsoid (mjc_PlaneConvex): deepest support point along -normal
            const float* n = E + Y.gax + 3 * g1;
            float R2[9], nl[3], sp[3], pw[3];
            geom_world_mat(M, Y, E, g2, R2);
            matTvec(nl, R2, n);
            float sv[3] = {sz2[0] * nl[0], sz2[1] * nl[1], sz2[2] * nl[2]};
            float nn = norm3(sv), inv = nn > MINVALF ? -1.0f / nn : 0.f;
            sp[0] = sz2[0] * sv[0] * inv; sp[1] = sz2[1] * sv[1] * inv; sp[2] = sz2[2] * sv[2] * inv;
            matvec(pw, R2, sp);
            float rel[3] = {x2[0] - x1[0] + pw[0], x2[1] - x1[1] + pw[1], x2[2] - x1[2] + pw[2]};
            float d = dot3(rel, n);
#pragma unroll
            for (int k = 0; k < 3; k++) nrm[k] = n[k];
            if (d <= margin) {
              hit = true; dist = d;
#pragma unroll
              for (int k = 0; k < 3; k++) cpos[k] = x2[k] + pw[k] - n[k] * 0.5f * d;
            }
          } else {
            const float zero3[3] = {0.f, 0.f, 0.f};
            nsup = 0;
            // MPR in geom1's own frame: obj1 needs no rotation / translation at all (identity frame), obj2 carries the
            // relative pose R1^T R2, R1^T (x2 - x1); normal and position are rotated back afterwards
            float R1[9];
            geom_world_mat(M, Y, E, g1, R1);
            CObj o1, o2;
            {
              float R2[9], rel[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
              geom_world_mat(M, Y, E, g2, R2);
#pragma unroll
              for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) o2.mat[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
              matTvec(o2.pos, R1, rel);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) o1.mat[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) { o1.pos[k] = 0.f; o1.size[k] = sz1[k]; o2.size[k] = sz2[k]; }
            o1.type = M.cg_type[g1]; o2.type = M.cg_type[g2]; o1.margin = o2.margin = 0.5f * margin;
            float depth, dir[3], pos[3];
            if (mpr_penetration(o1, o2, 1e-8f, 60, &depth, dir, pos, &nsup)) {
              dist = margin - depth;
              normalize3(dir);
              float dw[3], pw[3];
              matvec(dw, R1, dir);
              matvec(pw, R1, pos);
#pragma unroll
              for (int k = 0; k < 3; k++) { cpos[k] = pw[k] + x1[k]; nrm[k] = dw[k]; }
              hit = true;
            }
          }
          if (hit && !(dist < margin - M.pair_f[12 * p + 1])) hit = false;
          if (hit2 && !(dist2 < margin - M.pair_f[12 * p + 1])) hit2 = false;
        }
        {  // slowest lane of this round: MPR lanes cost ~8 + refinement steps, analytic pairs ~1
          int w = nsup + 8;
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x4E, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x141, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x140, 0xf, 0xf, true));
          f_mpr += max(max(rdlanei(w, 0), rdlanei(w, 16)), max(rdlanei(w, 32), rdlanei(w, 48)));
        }
        unsigned long long bal = __ballot(hit);
        int pos = ncon + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < NC) {
          E[Y.cdist + pos] = dist;
#pragma unroll
          for (int k = 0; k < 3; k++) { E[Y.cpos + 3 * pos + k] = cpos[k]; E[Y.cnrm + 3 * pos + k] = nrm[k]; }
          ((int*)(E + Y.cpair))[pos] = p;
        }
        ncon += __popcll(bal);
        bal = FULL ? __ballot(hit2) : 0ull;
        if (bal) {
          pos = ncon + __popcll(bal & ((1ull << lane) - 1ull));
          if (hit2 && pos < NC) {
            E[Y.cdist + pos] = dist2;
#pragma unroll
            for (int k = 0; k < 3; k++) { E[Y.cpos + 3 * pos + k] = cpos2[k]; E[Y.cnrm + 3 * pos + k] = nrm[k]; }
            ((int*)(E + Y.cpair))[pos] = p;
          }
          ncon += __popcll(bal);
        }
      }
      if (ncon > NC) { flags |= MYO_FLAG_CONTACT_OVERFLOW; ncon = NC; }
      SYNC();
    }
    STAMP(4);
    // ---------------------------------------------------------------- constraint rows (registers: lane = dof / lane = contact)
    float lsign = 0.f, laref = 0.f, lD = 0.f;
    if (lane < nv && !M.disable_limit) {
      const float* J = M.jl + 12 * lane;
      if (J[0] != 0) {
        float q = E[Y.qpos + W.dof_qposadr[lane]], margin = J[3];
        float dlo = q - J[1], dhi = J[2] - q, dist = 0;
        if (dlo < margin && dlo <= dhi) { lsign = 1; dist = dlo; }
        else if (dhi < margin) { lsign = -1; dist = dhi; }
        if (lsign != 0) {
          float imp = impedance(J + 6, dist, margin), K, B;
          float R = fmaxf(MINVALF, (1 - imp) / imp * J[11]);
          kbi(J[4], J[5], J[7], M.timestep, &K, &B);
          laref = -B * (lsign * E[Y.qvel + lane]) - K * imp * (dist - margin);
          lD = 1.0f / R;
        }
      }
    }
    float caref[4] = {0, 0, 0, 0}, cD = 0.f, cmu = 0.f;
    int ckc = 0;
    if (lane < ncon) {
      int c = lane;
      int p = ((const int*)(E + Y.cpair))[c];
      const int* P = M.pair_i + 6 * p;
      const float* F = M.pair_f + 12 * p;
      float n[3] = {E[Y.cnrm + 3 * c], E[Y.cnrm + 3 * c + 1], E[Y.cnrm + 3 * c + 2]}, t1[3], t2[3];
      float cp[3] = {E[Y.cpos + 3 * c], E[Y.cpos + 3 * c + 1], E[Y.cpos + 3 * c + 2]};
      make_frame(n, t1, t2);
      if (FULL && P[4] == 2) {   // plane - capsule: first tangent along the capsule axis (MuJoCo's frame for this pair type)
        const float* ax = E + Y.gax + 3 * P[1];
        float t = dot3(ax, n), y[3] = {ax[0] - t * n[0], ax[1] - t * n[1], ax[2] - t * n[2]};
        float yn = norm3(y);
        if (yn >= 0.5f) {
          float inv = 1.0f / yn;
          t1[0] = y[0] * inv; t1[1] = y[1] * inv; t1[2] = y[2] * inv;
          cross3(t2, n, t1);
        }
      }
      float vn = 0, vt1 = 0, vt2 = 0;
      float* cJ = E + Y.cJ + c * 3 * KC;
      unsigned int dpk[CDW];
#pragma unroll
      for (int k = 0; k < CDW; k++) dpk[k] = 0;
      ckc = P[3];
#pragma unroll
      for (int k = 0; k < KC; k++) {
        float jn = 0, j1 = 0, j2 = 0;
        int d = 0;
        if (k < ckc) {
          d = M.pair_dl[2 * (P[2] + k)];
          float sg = (float)M.pair_dl[2 * (P[2] + k) + 1];
          const float* ax = E + Y.axis + 3 * d;
          float col[3];
          if (M.dof_type[d] == 3) {
            float r[3] = {cp[0] - E[Y.anchor + 3 * d], cp[1] - E[Y.anchor + 3 * d + 1], cp[2] - E[Y.anchor + 3 * d + 2]};
            cross3(col, ax, r);
          } else { col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2]; }
          jn = sg * dot3(n, col); j1 = sg * dot3(t1, col); j2 = sg * dot3(t2, col);
          float qv = E[Y.qvel + d];
          vn += jn * qv; vt1 += j1 * qv; vt2 += j2 * qv;
        }
        cJ[k] = jn; cJ[KC + k] = j1; cJ[2 * KC + k] = j2;
        dpk[k >> 2] |= (unsigned int)d << (8 * (k & 3));   // padded entries: zero jacobian, dof 0
      }
#pragma unroll
      for (int k = 0; k < CDW; k++) ((unsigned int*)(E + Y.cdofs))[CDW * c + k] = dpk[k];
      float dist = E[Y.cdist + c], incl = F[0] - F[1];
      cmu = F[2];
      float imp = impedance(F + 6, dist, incl), K, B;
      kbi(F[4], F[5], F[7], M.timestep, &K, &B);
      if (FULL && P[5] == 1) {
        // condim 1 (explicit <pair>): one frictionless row = four identical "pyramid" rows with mu = 0 and D/4 each
        cmu = 0.f;
        cD = 0.25f / fmaxf(MINVALF, (1 - imp) / imp * F[3]);
      } else {
        float R0 = fmaxf(MINVALF, (1 - imp) / imp * F[3] * (1 + cmu * cmu));
        cD = 1.0f / fmaxf(MINVALF, 2 * cmu * cmu * R0);
      }
      float pos = -K * imp * (dist - incl);
      caref[0] = -B * (vn + cmu * vt1) + pos; caref[1] = -B * (vn - cmu * vt1) + pos;
      caref[2] = -B * (vn + cmu * vt2) + pos; caref[3] = -B * (vn - cmu * vt2) + pos;
    }
    // efc row count as MuJoCo reports it: 4 pyramid rows per condim-3 contact, 1 per frictionless (condim-1) contact
    int nefc = __popcll(__ballot(lsign != 0.f)) + 4 * ncon - 3 * __popcll(__ballot(lane < ncon && cmu == 0.f));
    const int ncon_real = ncon;
    if (W.has_tl) {
      // an active tendon limit becomes a frictionless pseudo-contact: jacobian = +-(sparse tendon row), mu = 0 and D/4 on each
      // of the four identical "pyramid" rows, which together act exactly like the single MuJoCo limit row
      int nt = ncon;
#pragma unroll
      for (int rr = 0; rr < NTR; rr++) {
      const int gt = lane + 64 * rr;
      bool tact = false;
      float t_aref = 0.f, t_D = 0.f, t_sign = 0.f;
      if (gt < ngt_ && !M.disable_limit) {
        const float* T = W.tl + 12 * gt;
        if (T[0] != 0) {
          float margin = T[3], dlo = tlen_r[rr] - T[1], dhi = T[2] - tlen_r[rr], dist = 0;
          if (dlo < margin && dlo <= dhi) { t_sign = 1; dist = dlo; }
          else if (dhi < margin) { t_sign = -1; dist = dhi; }
          if (t_sign != 0) {
            float imp = impedance(T + 6, dist, margin), K, B;
            float R = fmaxf(MINVALF, (1 - imp) / imp * T[11]);
            kbi(T[4], T[5], T[7], M.timestep, &K, &B);
            t_aref = -B * (t_sign * tvel_r[rr]) - K * imp * (dist - margin);
            t_D = 1.0f / R;
            tact = true;
          }
        }
      }
      unsigned long long bal = __ballot(tact);
      int slot = nt + __popcll(bal & ((1ull << lane) - 1ull));
      if (tact && slot < NC) {
        float* cJ = E + Y.cJ + slot * 3 * KC;
        unsigned int dpk[CDW];
#pragma unroll
        for (int k = 0; k < CDW; k++) dpk[k] = 0;
        int kc = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
          int d = k < maxnnz_ ? M.gt_dofs[gt * maxnnz_ + k] : -1;
          float jv = d >= 0 ? t_sign * E[Y.tJp + gt * maxnnz_ + k] : 0.f;
          if (d >= 0) kc = k + 1; else d = 0;
          cJ[k] = jv; cJ[KC + k] = 0.f; cJ[2 * KC + k] = 0.f;
          dpk[k >> 2] |= (unsigned int)d << (8 * (k & 3));
        }
#pragma unroll
        for (int k = 0; k < CDW; k++) ((unsigned int*)(E + Y.cdofs))[CDW * slot + k] = dpk[k];
        E[Y.cdist + slot] = t_aref; E[Y.cpos + 3 * slot] = t_D; E[Y.cpos + 3 * slot + 1] = (float)kc;
      }
      nt += __popcll(bal);
      }
      if (nt > NC) { flags |= MYO_FLAG_CONTACT_OVERFLOW; nt = NC; }
      SYNC();
      if (lane >= ncon && lane < nt) {
        float a = E[Y.cdist + lane];
        caref[0] = caref[1] = caref[2] = caref[3] = a;
        cD = 0.25f * E[Y.cpos + 3 * lane]; cmu = 0.f; ckc = (int)E[Y.cpos + 3 * lane + 1];
      }
      nefc += nt - ncon;
      ncon = nt;
    }
    // joint-coupling equalities q1 - q1_0 = poly(q2 - q2_0) (mj_instantiateEquality, mjEQ_JOINT): lane = equality, two
    // jacobian entries (+1 at dof 1, -poly' at dof 2), always active (quadratic cost on both sides)
    float eJ2 = 0.f, eD = 0.f, earef = 0.f, ejar = 0.f, ejv = 0.f;
    int ed1 = 0, ed2 = 0;
    const bool eact = lane < neq;
    if (eact) {
      const int* Q = W.eq_i + 4 * lane;
      const float* F = W.eq_f + 16 * lane;
      ed1 = Q[0]; ed2 = Q[1];
      float x = E[Y.qpos + Q[3]] - F[6];
      float pos = E[Y.qpos + Q[2]] - F[5] - (F[0] + x * (F[1] + x * (F[2] + x * (F[3] + x * F[4]))));
      eJ2 = -(F[1] + x * (2 * F[2] + x * (3 * F[3] + x * 4 * F[4])));
      float vel = E[Y.qvel + ed1] + eJ2 * E[Y.qvel + ed2];
      float imp = impedance(F + 9, pos, 0.f), K, B;
      kbi(F[7], F[8], F[10], M.timestep, &K, &B);
      earef = -B * vel - K * imp * pos;
      eD = 1.0f / fmaxf(MINVALF, (1 - imp) / imp * F[14]);
    }
    nefc += neq;
    SYNC();
    // the mass matrix moves from the square buffer (about to be reused for the Hessian) to a packed copy that
    // aliases the now dead broad-phase scratch
    if (lane < nv) {
      const int based = (lane * (lane + 1)) / 2;
#pragma unroll
      for (int k = 0; k < NVT; k++) if (k <= lane) E[Y.Mp + based + k] = E[Y.sq + lane * (NVT + 1) + k];
    }
    const float* Mp = E + Y.Mp;
    SYNC();
    STAMP(5);
    // ---------------------------------------------------------------- solver: Newton iterations, then the Euler solve, sharing ONE
    // instance of the unrolled register Cholesky.  phase 0 = Newton, 1 = unconstrained (nefc == 0), 2 = Euler (implicit damping)
    float Ma = 0.f, grad = 0.f, qfc = 0.f, ljar = 0.f, ljv = 0.f, cost = 0.f, qaccE = 0.f;
    float cjar[4] = {0, 0, 0, 0}, cjv[4] = {0, 0, 0, 0};
    int phase = nefc > 0 ? 0 : 1, iters = 0;
    if (phase == 0) {  // start from the warm start (MuJoCo also tries qacc_smooth; the minimiser is the same)
      qacc = warm;
      Ma = symv_lds<NVT>(Mp, qacc, lane, nv);
      ljar = lsign * qacc - laref;
      if (lane < nv) E[Y.xv + lane] = qacc;
      SYNC();
      if (lane < ncon) {
        const float* cJ = E + Y.cJ + lane * 3 * KC;
        float an = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) { float xv = E[Y.xv + CDOF(E, Y, lane, k)]; an += cJ[k] * xv; a1 += cJ[KC + k] * xv; a2 += cJ[2 * KC + k] * xv; }
        cjar[0] = an + cmu * a1 - caref[0]; cjar[1] = an - cmu * a1 - caref[1]; cjar[2] = an + cmu * a2 - caref[2]; cjar[3] = an - cmu * a2 - caref[3];
      }
      if (eact) ejar = E[Y.xv + ed1] + eJ2 * E[Y.xv + ed2] - earef;
    }
    bool first = true;
    int sig_prev = -1;
    const int lane_s = lane;
    while (true) {
      int lane;   // opaque copy again: keeps the 24 per-lane symv addresses from being hoisted out of the loop and spilled
      asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_s));
      float r[NVT], rhs, invd;
      bool refactor = true;
      if (phase == 0) {
        // forces of the active rows, J^T f (LDS atomics), cost, gradient; convergence test; then, only if the iteration goes on and
        // the active set differs from the one whose Hessian was factorised last, the Hessian blocks (LDS atomics)
        bool lact = lsign != 0.f && ljar < 0;
        float w0 = cjar[0] < 0 ? cD : 0.f, w1 = cjar[1] < 0 ? cD : 0.f, w2 = cjar[2] < 0 ? cD : 0.f, w3 = cjar[3] < 0 ? cD : 0.f;
        float f0 = -w0 * cjar[0], f1 = -w1 * cjar[1], f2 = -w2 * cjar[2], f3 = -w3 * cjar[3];
        if (lane < nv) E[Y.qfc + lane] = lact ? -lsign * lD * ljar : 0.f;
        SYNC();
        if (lane < ncon) {
          const float* cJ = E + Y.cJ + lane * 3 * KC;
          float Fn = f0 + f1 + f2 + f3, Ft1 = cmu * (f0 - f1), Ft2 = cmu * (f2 - f3);
          for (int k = 0; k < ckc; k++) atomicAdd(&E[Y.qfc + CDOF(E, Y, lane, k)], Fn * cJ[k] + Ft1 * cJ[KC + k] + Ft2 * cJ[2 * KC + k]);
        }
        if (eact) { float f = -eD * ejar; atomicAdd(&E[Y.qfc + ed1], f); atomicAdd(&E[Y.qfc + ed2], eJ2 * f); }
        SYNC();
        qfc = lane < nv ? E[Y.qfc + lane] : 0.f;
        float cst = lact ? 0.5f * lD * ljar * ljar : 0.f;
        cst += 0.5f * eD * ejar * ejar;
        cst += 0.5f * (w0 * cjar[0] * cjar[0] + w1 * cjar[1] * cjar[1] + w2 * cjar[2] * cjar[2] + w3 * cjar[3] * cjar[3]);
        cst += 0.5f * qacc * Ma - qacc * smooth;          // Gauss term up to a constant
        float newcost = wave_sum(cst);
        grad = Ma - smooth - qfc;
        if (!first) {
          float improvement = scale * (cost - newcost);
          float gn = scale * sqrtf(wave_sum(grad * grad));
          iters++;
          f_itcon += ncon;
          if (improvement < fmaxf(M.tolerance, 1e-6f * scale * fabsf(newcost)) || gn < M.tolerance || iters >= M.iterations) phase = 2;
        }
        cost = newcost;
        if (phase == 0) {
          // H = M + J^T D J depends on the state only through the set of active rows: same set as last time -> same factor
          const int sig = (lact ? 1 : 0) | (w0 != 0.f ? 2 : 0) | (w1 != 0.f ? 4 : 0) | (w2 != 0.f ? 8 : 0) | (w3 != 0.f ? 16 : 0);
          refactor = first || __any(sig != sig_prev);
          sig_prev = sig;
          if (refactor) {
            f_fact++;
            WFOR(i, NVT * (NVT + 1)) E[Y.sq + i] = 0;
            SYNC();
            if (lane < nv && lact) E[Y.sq + lane * (NVT + 1) + lane] = lD;
            float Wn = w0 + w1 + w2 + w3, A1 = cmu * (w0 - w1), A2 = cmu * (w2 - w3), B1 = cmu * cmu * (w0 + w1), B2 = cmu * cmu * (w2 + w3);
            SYNC();
            for (int c = 0; c < ncon; c++) {   // one contact per step, lanes = entries of its kc x kc block
              int kc = rdlanei(ckc, c);
              float sW = rdlane(Wn, c), sA1 = rdlane(A1, c), sA2 = rdlane(A2, c), sB1 = rdlane(B1, c), sB2 = rdlane(B2, c);
              if (sW == 0.f) continue;
              for (int t = lane; t < kc * kc; t += 64) {
                int a = t / kc, b = t - a * kc;
                const float* cJ = E + Y.cJ + c * 3 * KC;
                int da = CDOF(E, Y, c, a), db = CDOF(E, Y, c, b);
                if (da >= db) {
                  float na = cJ[a], nb = cJ[b], ta = cJ[KC + a], tb = cJ[KC + b], ua = cJ[2 * KC + a], ub = cJ[2 * KC + b];
                  atomicAdd(&E[Y.sq + da * (NVT + 1) + db], sW * na * nb + sA1 * (na * tb + ta * nb) + sA2 * (na * ub + ua * nb) + sB1 * ta * tb + sB2 * ua * ub);
                }
              }
            }
            if (eact) {
              atomicAdd(&E[Y.sq + ed1 * (NVT + 1) + ed1], eD);
              atomicAdd(&E[Y.sq + ed2 * (NVT + 1) + ed2], eD * eJ2 * eJ2);
              atomicAdd(&E[Y.sq + max(ed1, ed2) * (NVT + 1) + min(ed1, ed2)], eD * eJ2);
            }
            SYNC();
          }
        }
        first = false;
      }
      rhs = phase == 0 ? -grad : (phase == 1 ? smooth : smooth + qfc);
      if (refactor) {
        const int dd = lane < nv ? lane : 0;
        const int based = (dd * (dd + 1)) / 2;
        const float diag_add = phase == 2 ? h * damping : 0.f;
#pragma unroll
        for (int k = 0; k < NVT; k++) {
          float mv = (lane < nv && k <= lane) ? Mp[based + (k <= dd ? k : 0)] : 0.f;                 // lower row of M
          float hv = (phase == 0 && lane < nv) ? E[Y.sq + lane * (NVT + 1) + k] : 0.f;              // J^T D J (active rows)
          r[k] = (lane < nv) ? mv + hv + (k == lane ? diag_add : 0.f) : (k == lane ? 1.f : 0.f);
        }
        SYNC();
        invd = chol_rows<NVT>(r, lane);
        if (lane < NVT) {
#pragma unroll
          for (int k = 0; k < NVT; k++) E[Y.sq + lane * (NVT + 1) + k] = r[k];
        }
        SYNC();
      } else {
        // the factor of the previous iteration is still in LDS (row-major L): reload this lane's row
        const int ll = lane < NVT ? lane : 0;
#pragma unroll
        for (int k = 0; k < NVT; k++) r[k] = E[Y.sq + ll * (NVT + 1) + k];
        invd = 1.0f / E[Y.sq + ll * (NVT + 1) + ll];
      }
      float x = chol_solve_rows<NVT>(r, invd, rhs, E + Y.sq, lane);
      if (phase == 1) { qacc = x; qfc = 0.f; phase = 2; continue; }
      if (phase == 2) { qaccE = x; break; }
      // ---- Newton: exact line search along x
      float search = lane < nv ? x : 0.f;
      float Mv = symv_lds<NVT>(Mp, search, lane, nv);
      ljv = lsign * search;
      if (lane < nv) E[Y.xv + lane] = search;
      SYNC();
      if (lane < ncon) {
        const float* cJ = E + Y.cJ + lane * 3 * KC;
        float an = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) { float xv = E[Y.xv + CDOF(E, Y, lane, k)]; an += cJ[k] * xv; a1 += cJ[KC + k] * xv; a2 += cJ[2 * KC + k] * xv; }
        cjv[0] = an + cmu * a1; cjv[1] = an - cmu * a1; cjv[2] = an + cmu * a2; cjv[3] = an - cmu * a2;
      }
      if (eact) ejv = E[Y.xv + ed1] + eJ2 * E[Y.xv + ed2];
      float g1 = wave_sum(search * (Ma - smooth)), g2 = wave_sum(0.5f * search * Mv), sn = sqrtf(wave_sum(search * search));
      float alpha = 0, lo = 0, hi = -1, dlo = 0, d2lo = 0, dhi = 0, d2hi = 0, d1init = 0;
      bool ls_on = sn >= MINVALF;
      for (int lsit = -1; lsit < M.ls_iterations && ls_on; lsit++) {
        float a = (lsit < 0) ? 0.f : alpha;
        float p1 = 0, p2 = 0;
        if (lsign != 0.f) { float xx = ljar + a * ljv; if (xx < 0) { p1 += lD * xx * ljv; p2 += lD * ljv * ljv; } }
        p1 += eD * (ejar + a * ejv) * ejv; p2 += eD * ejv * ejv;
#pragma unroll
        for (int k = 0; k < 4; k++) { float xx = cjar[k] + a * cjv[k]; if (xx < 0) { p1 += cD * xx * cjv[k]; p2 += cD * cjv[k] * cjv[k]; } }
        float d1 = wave_sum(p1) + g1 + 2 * a * g2;
        float d2 = wave_sum(p2) + 2 * g2;
        if (lsit < 0) {
          if (d1 >= 0 || d2 <= 0) { ls_on = false; alpha = 0; break; }
          dlo = d1; d2lo = d2; d1init = fabsf(d1);
          alpha = -d1 / d2;
          continue;
        }
        f_ls++;
        float gtol = fmaxf(M.tolerance * M.ls_tolerance * sn / scale, LS_FLOOR * d1init);
        if (fabsf(d1) < gtol) break;
        if (d1 < 0) { lo = alpha; dlo = d1; d2lo = d2; } else { hi = alpha; dhi = d1; d2hi = d2; }
        float cand = alpha - d1 / d2;
        if (hi < 0) {
          if (!(cand > lo)) break;
          alpha = cand;
        } else {
          if (!(cand > lo && cand < hi)) {
            float c2 = d1 < 0 ? hi - dhi / d2hi : lo - dlo / d2lo;
            cand = (c2 > lo && c2 < hi) ? c2 : 0.5f * (lo + hi);
          }
          if (cand == alpha || hi - lo <= 1e-7f * hi) break;
          alpha = cand;
        }
      }
      if (!(alpha > 0)) { phase = 2; continue; }   // no descent left: keep qacc / qfc of this iterate
      qacc += alpha * search; Ma += alpha * Mv; ljar += alpha * ljv; ejar += alpha * ejv;
#pragma unroll
      for (int k = 0; k < 4; k++) cjar[k] += alpha * cjv[k];
    }
    STAMP(7);
    d_nefc = nefc; d_ncon = ncon_real; d_iter = max(d_iter, iters);
    f_ncon += ncon; f_iter += iters;
    {  // mj_checkAcc
      bool bad = lane < nv && (!(qacc == qacc) || fabsf(qacc) > MAXVALF);
      if (__any(bad) && alive) { flags |= MYO_FLAG_BAD_QACC; alive = false; }
    }
    warm = qacc;
    if (alive) {
#pragma unroll
      for (int rr = 0; rr < NTR; rr++) if (lane + 64 * rr < nu) E[Y.act + lane + 64 * rr] += h * actdot[rr];
      if (lane < nv) {
        float v = E[Y.qvel + lane] + h * qaccE;
        E[Y.qvel + lane] = v;
        if (!(has_free && lane >= 3 && lane < 6)) E[Y.qpos + W.dof_qposadr[lane]] += h * v;
      }
      if (has_free) {   // root free joint: quaternion integrated with the body-frame angular velocity (mju_quatIntegrate)
        SYNC();
        if (lane == 3) {
          float w[3] = {E[Y.qvel + 3], E[Y.qvel + 4], E[Y.qvel + 5]};
          float wn = norm3(w), ang = h * wn;
          float q[4] = {E[Y.qpos + 3], E[Y.qpos + 4], E[Y.qpos + 5], E[Y.qpos + 6]};
          if (wn >= MINVALF) {
            float sn, cs;
            sincosf(0.5f * ang, &sn, &cs);
            float inv = sn / wn, r[4] = {cs, w[0] * inv, w[1] * inv, w[2] * inv}, o[4];
            o[0] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
            o[1] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
            o[2] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
            o[3] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
            float on = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            E[Y.qpos + 3] = o[0] * on; E[Y.qpos + 4] = o[1] * on; E[Y.qpos + 5] = o[2] * on; E[Y.qpos + 6] = o[3] * on;
          }
        }
      }
      time += h;
    }
    SYNC();
    STAMP(8);
  }
  if (!SCHED && FULL && (kflags & KF_AUX)) return;   // observation-only launch: the state arrays are not touched
  if (!alive) {  // a bad env is reset like mj_resetData (mj_sim_scene.py:56-61)
    if (lane_id < nq) E[Y.qpos + lane_id] = M.qpos0[lane_id];
    if (lane_id < nv) { E[Y.qvel + lane_id] = 0; warm = 0; }
    for (int i = lane_id; i < nu; i += 64) { E[Y.act + i] = 0; E[Y.ctrl + i] = 0; }
    time = 0;
  }
  if (lane_id < nq) Bt.qpos[(size_t)env * nq + lane_id] = E[Y.qpos + lane_id];
  if (lane_id < nv) {
    Bt.qvel[(size_t)env * nv + lane_id] = E[Y.qvel + lane_id];
    Bt.warm[(size_t)env * nv + lane_id] = warm;
    Bt.qacc[(size_t)env * nv + lane_id] = qacc;
  }
  for (int i = lane_id; i < nu; i += 64) {
    Bt.act[(size_t)env * nu + i] = E[Y.act + i];
    Bt.ctrl[(size_t)env * nu + i] = E[Y.ctrl + i];
  }
  if (lane_id == 0) {
    Bt.time[env] = time;
    if (s1 == nsubtot) Bt.elapsed[env] += 1;
    if (SCHED) { if (flags) atomicOr(&Bt.flags[env], flags); } else Bt.flags[env] |= flags;
    Bt.diag[(size_t)env * 8 + 0] = d_nefc; Bt.diag[(size_t)env * 8 + 1] = d_ncon; Bt.diag[(size_t)env * 8 + 2] = d_iter;
    // predicted work of this env's NEXT step for the placement hint, in units of 1024 single-wave cycles: linear model of this
    // step's work features and the last substep's contact / row counts, fitted on one-wave-per-SIMD runs where a wave's duration is
    // its own work (tools/gpu_cost_fit2.py; correlation with the next step's measured duration 0.92 hand / 0.90 legs)
    d_cost = FULL ? 1909 + ((7436 * f_cand - 28892 * f_ncon + 1362 * f_mpr + 25014 * f_iter + 2046 * f_itcon - 1413 * f_ls + 13292 * d_nefc + 343707 * d_ncon) >> 10)
                  : 2236 + ((-141 * f_cand - 3413 * f_ncon + 2188 * f_mpr + 8971 * f_iter - 107 * f_itcon - 122 * f_ls - 518 * d_nefc + 74475 * d_ncon) >> 10);
    d_cost = max(d_cost, 1);
    Bt.diag[(size_t)env * 8 + 3] = d_cost;
    Bt.diag[(size_t)env * 8 + 4] = f_cand | (f_ncon << 16); Bt.diag[(size_t)env * 8 + 5] = f_mpr;
    Bt.diag[(size_t)env * 8 + 6] = f_itcon | (f_iter << 16); Bt.diag[(size_t)env * 8 + 7] = f_ls | (f_fact << 16);
  }
  last_cost = d_cost;
  STAMP(9);
  if (!SCHED) break;
  // publish this env's next substep: the state rows written above must have reached L2 before the ring entry becomes visible
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (s1 < nsubtot && lane_id == 0) {
    const int tt = atomicAdd(&sq_ctl[1], 1);
    const int g2 = tt / sq_n, sl = tt - g2 * sq_n;
    __hip_atomic_store(&sq_ring[sl], (g2 << 24) | (s1 << 20) | env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SYNC();   // the next task reuses this wave's LDS slice
  }  // task loop
#if MYO_STAMPS
  st_acc[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave/simd/cu/sh/se ids (placement census)
  st_acc[11] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFF) | ((long long)(oe >> 28) << 8) | ((long long)last_cost << 16);   // HW_REG_XCC_ID, issue priority, cost estimate
  if (stamps && lane_id == 0) for (int k = 0; k < 12; k++) stamps[(size_t)blockIdx.x * 12 + k] = st_acc[k];
#endif
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG (splitmix64 of (seed, stream, counter)) -> U[0,1)
__device__ __host__ inline float u01(uint64_t seed, uint64_t a, uint64_t b) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xBF58476D1CE4E5B9ull * (b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// Placement hint for the wave-per-env kernel.  All B envs are co-resident (4 waves per SIMD), so a launch ends when the
// slowest SIMD ends; envs differ ~2x in work (contacts, Newton iterations) and that work is strongly correlated from one
// env step to the next.  Sort envs by last step's cost (counting sort, one workgroup) and deal them out so that the waves
// that land on one SIMD come from different cost quartiles (snake order over `nslot` = B/4 slots).  Dispatch order is not
// a contract: this only ever changes speed.
__global__ void __launch_bounds__(1024) balance_kernel(const int* __restrict__ diag, int B, int* __restrict__ order, int nslot, int prio_mode) {
  __shared__ int hist[256], start[256];
  __shared__ int cmax_s;
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  if (t == 0) cmax_s = 1;
  __syncthreads();
  int cm = 1;
  for (int e = t; e < B; e += 1024) cm = max(cm, diag[(size_t)e * 8 + 3]);
  atomicMax(&cmax_s, cm);
  __syncthreads();
  const int cmax = cmax_s;
  for (int e = t; e < B; e += 1024) atomicAdd(&hist[255 - min(255, (int)(255LL * diag[(size_t)e * 8 + 3] / cmax))], 1);   // bucket 0 = heaviest
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 256; k++) { start[k] = acc; acc += hist[k]; } }
  __syncthreads();
  for (int e = t; e < B; e += 1024) {
    int b = 255 - min(255, (int)(255LL * diag[(size_t)e * 8 + 3] / cmax));
    int r = atomicAdd(&start[b], 1);                       // rank by descending cost (ties in arbitrary order)
    int q = r / nslot, i = r - q * nslot;
    int wg = q * nslot + ((q & 1) ? nslot - 1 - i : i);    // snake: slot i gets ranks i, 2*nslot-1-i, 2*nslot+i, ...
    int pr = prio_mode == 2 ? (q < 3 ? 3 - q : 0) : (prio_mode == 1 ? (q == 0 ? 1 : 0) : (prio_mode == 3 ? (q < 2 ? 1 : 0) : 0));
    order[wg < B ? wg : r] = e | (pr << 28);   // env id + issue priority of its cost quartile
  }
}

__global__ void random_action_kernel(float* action, int B, int nu, uint64_t seed, uint64_t step, int env_offset) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * nu) return;
  size_t e = i / nu, k = i % nu;
  action[i] = 2.0f * u01(seed, (uint64_t)(e + env_offset) * 1024 + k, step) - 1.0f;
}

// ------------------------------------------------------------------------------------------------
// policy inference (brax PPO network family): 8 envs per 256-thread workgroup, thread = (env, hidden unit); activations ping-pong
// through LDS, weights are read coalesced across units and shared by the 8 envs through the cache.  ~13 kMAC per env for
// the hand observation: negligible next to the physics step, so plain FMAs (no MFMA)
#define POL_ENVS 8
#define POL_MAXW 64
struct PolicyDev {
  int obs_dim, act_dim, nlayers;
  int width[8];               // output width of each layer
  const float* W[8];
  const float* b[8];
  const float *mean, *std;
};
__global__ void __launch_bounds__(POL_ENVS * POL_MAXW) policy_kernel(PolicyDev P, const float* __restrict__ obs, int B, float* __restrict__ action,
                                                                     int deterministic, uint64_t seed, uint64_t step, int env_offset) {
  extern __shared__ float sh[];                       // [POL_ENVS][max(obs_dim, POL_MAXW)] x 2
  const int j = threadIdx.x % POL_MAXW, le = threadIdx.x / POL_MAXW;
  const int e = blockIdx.x * POL_ENVS + le;
  int stride = max(P.obs_dim, POL_MAXW);
  for (int l = 0; l < P.nlayers; l++) stride = max(stride, P.width[l]);
  float* xin = sh + le * stride;
  float* xout = sh + (POL_ENVS + le) * stride;
  if (e < B)
    for (int i = j; i < P.obs_dim; i += POL_MAXW) xin[i] = (obs[(size_t)e * P.obs_dim + i] - P.mean[i]) / P.std[i];
  __syncthreads();
  int nin = P.obs_dim;
  for (int l = 0; l < P.nlayers; l++) {
    const int nout = P.width[l];
    if (e < B) {
      const float* Wl = P.W[l];
      for (int jj = j; jj < nout; jj += POL_MAXW) {
        float acc = P.b[l][jj];
        for (int i = 0; i < nin; i++) acc += xin[i] * Wl[(size_t)i * nout + jj];
        xout[jj] = (l + 1 < P.nlayers) ? acc / (1.0f + expf(-acc)) : acc;     // swish on hidden layers, linear head
      }
    }
    __syncthreads();
    float* t = xin; xin = xout; xout = t;
    nin = nout;
  }
  if (e < B) {
    for (int jj = j; jj < P.act_dim; jj += POL_MAXW) {
      float loc = xin[jj], a = loc;
      if (!deterministic) {
        float raw = xin[P.act_dim + jj];
        float scale = (raw > 20.f ? raw : log1pf(expf(raw))) + 0.001f;
        uint64_t ge = (uint64_t)(e + env_offset);
        float u1 = fmaxf(u01(seed ^ 0x5851F42D4C957F2Dull, ge * 1024 + jj, step), 1e-7f), u2 = u01(seed ^ 0x14057B7EF767814Full, ge * 1024 + jj, step);
        a = loc + scale * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);   // Box-Muller
      }
      action[(size_t)e * P.act_dim + jj] = tanhf(a);
    }
  }
}

// auto_max > 0: gym TimeLimit / done auto-reset (reset iff done or elapsed >= auto_max); else mask-driven reset.
// One 64-lane workgroup per env: envs that are not reset leave after one test, the others write their rows coalesced
// (one thread per env needed ~500 serialised scattered stores per reset: 27 ms for a full reset of 4096 leg envs)
__global__ void __launch_bounds__(64) reset_kernel(DevBatch Bt, TaskDev T, int nq, int nv, int nu, const float* qpos0, const uint8_t* mask, uint64_t seed,
                                                  int env_offset, int auto_max) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= Bt.B) return;
  if (auto_max > 0) { if (!(Bt.done[e] > 0.f || Bt.elapsed[e] >= auto_max)) return; }
  else if (mask && !mask[e]) return;
  seed += 0x632BE59BD9B4E019ull * (uint64_t)Bt.episode[e];  // a fresh RNG stream per (env, episode)
  __syncthreads();                                            // every lane has read done / elapsed / episode before lane 0 updates them
  if (lane == 0) { Bt.episode[e] += 1; Bt.elapsed[e] = 0; Bt.done[e] = 0.f; Bt.time[e] = 0; }
  uint64_t ge = (uint64_t)(e + env_offset);
  for (int i = lane; i < nq; i += 64) {
    float q = T.init_qpos ? T.init_qpos[i] : qpos0[i];
    if (T.reset_random) q = T.jnt_lo[i] + (T.jnt_hi[i] - T.jnt_lo[i]) * u01(seed, ge * 4096 + i, 1);   // nq == nv checked at configure
    Bt.qpos[(size_t)e * nq + i] = q;
  }
  for (int i = lane; i < nv; i += 64) {
    Bt.qvel[(size_t)e * nv + i] = T.init_qvel ? T.init_qvel[i] : 0.f;
    Bt.warm[(size_t)e * nv + i] = 0;
  }
  for (int i = lane; i < nu; i += 64) {
    Bt.act[(size_t)e * nu + i] = 0; Bt.ctrl[(size_t)e * nu + i] = 0;
    // fatigue compartments: all motor units resting (CumulativeFatigue.reset defaults, fatigue.py:130-134)
    Bt.fatigue[(size_t)e * 3 * nu + i] = 0.f; Bt.fatigue[(size_t)e * 3 * nu + nu + i] = 1.f; Bt.fatigue[(size_t)e * 3 * nu + 2 * nu + i] = 0.f;
  }
  for (int i = lane; i < T.ntarget; i += 64) {
    float lo = T.target_lo[i], hi = T.target_hi[i];
    Bt.target[(size_t)e * T.ntarget + i] = T.target_generate ? lo + (hi - lo) * u01(seed, ge * 4096 + 2048 + i, 2) : lo;
  }
}

// observation + reward (pose_v0.py:98-138, obs_vec_dict.py:86-98); one 64-lane workgroup per env, rows written coalesced
__global__ void __launch_bounds__(64) obs_kernel(DevModel M, DevBatch Bt, TaskDev T, int obs_only, int reset_only) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= Bt.B) return;
  if (reset_only && Bt.elapsed[e] != 0) return;    // refresh only the rows of envs an auto-reset just touched
  const int nv = M.nv, nu = M.nu;
  float dt = (float)T.frame_skip * M.timestep;
  float* o = Bt.obs + (size_t)e * T.obs_dim;
  const float* q = Bt.qpos + (size_t)e * nv;
  const float* v = Bt.qvel + (size_t)e * nv;
  const float* a = Bt.act + (size_t)e * nu;
  if (T.task == MYO_TASK_POSE) {
    float err2 = 0, act2 = 0;
    for (int i = lane; i < nv; i += 64) {
      float qi = q[i], pe = Bt.target[(size_t)e * T.ntarget + i] - qi;
      o[i] = qi; o[nv + i] = v[i] * dt; o[2 * nv + i] = pe;
      err2 += pe * pe;
    }
    for (int i = lane; i < nu; i += 64) { float ai = a[i]; o[3 * nv + i] = ai; act2 += ai * ai; }
    if (obs_only) return;
    float dist = sqrtf(wave_sum(err2));
    float actn = sqrtf(wave_sum(act2)) / (float)(nu > 0 ? nu : 1);
    if (lane == 0) {
      float bonus = (dist < T.pose_thd ? 1.f : 0.f) + (dist < 1.5f * T.pose_thd ? 1.f : 0.f);
      float pen = dist > T.far_th ? -1.f : 0.f;
      Bt.reward[e] = T.w_pose * (-dist) + T.w_bonus * bonus + T.w_act_reg * (-actn) + T.w_penalty * pen;
      Bt.solved[e] = dist < T.pose_thd ? 1.f : 0.f;
      Bt.done[e] = dist > T.far_th ? 1.f : 0.f;
    }
  }
}

// reach task needs tip positions: per-env group kernel reusing the kinematics stage
template <int G>
__global__ void __launch_bounds__(64) reach_obs_kernel(DevModel M, DevBatch Bt, TaskDev T, int obs_only) {
  extern __shared__ __align__(16) float smem[];
  const Lay& Y = M.lay;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  int env = blockIdx.x * (64 / G) + grp;
  const bool valid = env < Bt.B;
  if (!valid) env = Bt.B - 1;
  float* E = smem + grp * Y.total;
  const int nv = M.nv, nu = M.nu;
  GFOR(i, nv) E[Y.qpos + i] = Bt.qpos[(size_t)env * nv + i];
  SYNC();
  stage_kinematics<G>(M, E, sub);
  float dt = (float)T.frame_skip * M.timestep;
  float* o = Bt.obs + (size_t)env * T.obs_dim;
  float err2 = 0;
  GFOR(i, T.ntip) {
    float p[3];
    site_world(M, E, T.tip_site[i], p);
    for (int k = 0; k < 3; k++) {
      p[k] += M.origin[k];  // kernels work relative to the lowered origin; observations are world coordinates
      float tg = Bt.target[(size_t)env * T.ntarget + 3 * i + k];
      float re = tg - p[k];
      err2 += re * re;
      if (valid) {
        o[2 * nv + 3 * i + k] = p[k];
        o[2 * nv + 3 * T.ntip + 3 * i + k] = re;
        Bt.sitexpos[(size_t)env * 3 * T.ntip + 3 * i + k] = p[k];
      }
    }
  }
  err2 = grp_sum<G>(err2);
  float actn = 0;
  GFOR(i, nu) { float a = Bt.act[(size_t)env * nu + i]; actn += a * a; if (valid) o[2 * nv + 6 * T.ntip + i] = a; }
  actn = sqrtf(grp_sum<G>(actn)) / (float)(nu > 0 ? nu : 1);
  if (valid) {
    GFOR(i, nv) { o[i] = E[Y.qpos + i]; o[nv + i] = Bt.qvel[(size_t)env * nv + i] * dt; }
    if (sub == 0 && !obs_only) {
      float dist = sqrtf(err2);
      float near_th = T.near_th, far_th = Bt.time[env] > 2 * dt ? T.far_th : 1e30f;
      float bonus = (dist < 2 * near_th ? 1.f : 0.f) + (dist < near_th ? 1.f : 0.f);
      float pen = dist > far_th ? -1.f : 0.f;
      Bt.reward[env] = T.w_reach * (-dist) + T.w_bonus * bonus + T.w_act_reg * (-actn) + T.w_penalty * pen;
      Bt.solved[env] = dist < near_th ? 1.f : 0.f;
      Bt.done[env] = dist > far_th ? 1.f : 0.f;
    }
  }
}

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
  bool hand_sizes = false, leg_sizes = false;   // table sizes equal Sizes<1> / Sizes<2>: the size-specialised instantiations may be used
  int wave_cfg = 0;             // 0: step_kernel_w<24,8,32,1,4> (hand / finger), 1: step_kernel_w<36,20,48,2,2> (legs)
  int nq = 0;
  int has_tl = 0;
  myo_dims dims{};
  std::vector<void*> dev_allocs;
  std::vector<float> qpos0, jnt_lo, jnt_hi;
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
  float *d_tlo = nullptr, *d_thi = nullptr, *d_init = nullptr, *d_jlo = nullptr, *d_jhi = nullptr, *d_action = nullptr;
  float* d_initv = nullptr;
  DevWalk* d_walk = nullptr;
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
static int load_f(myo_model* m, const uint8_t* blob, const char* name, const float** out, std::vector<float>* keep = nullptr) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 0) return fail(MYO_E_BLOB, std::string("model blob lacks f64 array ") + name);
  size_t n = r->nbytes / 8;
  std::vector<float> v(n);
  const double* src = (const double*)(blob + r->offset);
  for (size_t i = 0; i < n; i++) v[i] = (float)src[i];
  if (keep) *keep = v;
  return upload<float>(m, v, out);
}
static int load_i(myo_model* m, const uint8_t* blob, const char* name, const int** out, std::vector<int>* keep = nullptr) {
  const BlobRec* r = blob_find(blob, name);
  if (!r || r->dtype != 1) return fail(MYO_E_BLOB, std::string("model blob lacks i32 array ") + name);
  size_t n = r->nbytes / 4;
  std::vector<int> v(n);
  memcpy(v.data(), blob + r->offset, n * 4);
  if (keep) *keep = v;
  return upload<int>(m, v, out);
}

static void build_layout_w(const DevModel& d, DevModelW& w, int nvt, int kc, int nc) {
  LayW& Y = w.lay;
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  int nv = d.nv, nu = d.nu, nl = d.nl;
  Y.qpos = take(w.nq); Y.qvel = take(nv); Y.act = take(nu); Y.ctrl = take(nu);
  Y.lpos = take(3 * nl); Y.lmat = take(9 * nl); Y.axis = take(3 * nv); Y.anchor = take(3 * nv);
  Y.xv = take(nvt); Y.qfc = take(nvt); Y.sq = take(nvt * (nvt + 1));
  Y.tJp = w.has_tl ? take(d.ngt * d.maxnnz) : 0;
  Y.X = o;
  Y.tJ = take(d.ngt * d.maxnnz); Y.tlen = take(d.ngt); Y.tforce = take(nu); Y.seglen = take(d.nseg); Y.dlval = take(w.ndl);
  int endT = o;
  o = Y.X;
  Y.cdof = take(6 * nv); Y.cinert = take(10 * nl); Y.crb = take(10 * nl); Y.cvel = take(6 * nl); Y.cacc = take(6 * nl); Y.cfrc = take(6 * nl);
  int endD = o;
  o = Y.X;
  Y.Mp = o;
  Y.gpos = take(3 * d.ncg); Y.gax = take(3 * d.ncg);
  Y.cand = take(NCAND);
  if (o - Y.Mp < (nvt * (nvt + 1)) / 2) o = Y.Mp + (nvt * (nvt + 1)) / 2; Y.cdist = take(nc); Y.cpos = take(3 * nc); Y.cnrm = take(3 * nc); Y.cpair = take(nc);
  Y.cJ = take(nc * 3 * kc); Y.cdofs = take(nc * ((kc + 3) / 4));   // kc dof ids per contact, one byte each
  if (o < endT) o = endT;
  if (o < endD) o = endD;
  Y.total = o;
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
          (rc = load_i(m, blob, "hip_dof_qposadr", &w.dof_qposadr)) || (rc = load_i(m, blob, "hip_eq_i", &w.eq_i)) ||
          (rc = load_f(m, blob, "hip_eq_f", &w.eq_f)) || (rc = load_i(m, blob, "hip_pair_i", &tmpi, &pi))) { myo_model_free(m); return rc; }
      w.has_free = fl[0]; w.nq = fl[1]; w.neq = fl[2];
      const float* tf;
      if ((rc = load_i(m, blob, "hip_body_link", &tmpi, &m->body_link)) || (rc = load_f(m, blob, "hip_body_lpos", &tf, &m->body_lpos)) ||
          (rc = load_f(m, blob, "hip_body_lquat", &tf, &m->body_lquat)) || (rc = load_f(m, blob, "hip_mass", &tf, &m->mass))) { myo_model_free(m); return rc; }
      if (w.nq != m->nq) { myo_model_free(m); return fail(MYO_E_BLOB, "hip_flags disagrees with sizes"); }
      for (int p = 0; p < d.npair; p++) { if (pi[6 * p + 4] >= 2) plane_pairs = true; if (pi[6 * p + 5] == 1) condim1 = true; }
      // the 16/32-lane generic kernel covers fixed-base models with hinge / slide joints and capsule / convex pairs only
      m->generic_ok = !w.has_free && w.neq == 0 && !plane_pairs && !condim1 && w.nq == d.nv && d.maxkc <= KCMAX && !w.has_tl;
    }
    const bool common = d.nl <= 64 && d.ncg <= 64 && w.nq <= 64 && w.neq <= 64 && d.maxnnz <= 20;
    const bool needs_full = w.has_free || w.neq > 0 || plane_pairs || condim1;
    if (common && !needs_full && d.nv <= 24 && d.nu <= 64 && d.ngt <= 64 && d.maxkc <= 8) { m->wave_ok = true; m->wave_cfg = 0; build_layout_w(d, w, 24, 8, 32); }
    else if (common && d.nv <= 36 && d.nu <= 128 && d.ngt <= 128 && d.maxkc <= 20) { m->wave_ok = true; m->wave_cfg = 1; build_layout_w(d, w, 36, 20, 32); }
    else { m->wave_ok = false; build_layout_w(d, w, 24, 8, 32); }
    m->hand_sizes = m->wave_ok && m->wave_cfg == 0 && sizes_match<1>(w.nq, d.nv, d.nu, d.nl, d.nlevel, d.maxnnz, d.ngt, d.nseg, d.ncg, d.npair);
    m->leg_sizes = m->wave_ok && m->wave_cfg == 1 && sizes_match<2>(w.nq, d.nv, d.nu, d.nl, d.nlevel, d.maxnnz, d.ngt, d.nseg, d.ncg, d.npair);
    if (const char* e = getenv("MYO_NO_SPEC")) if (atoi(e) == 1) m->hand_sizes = m->leg_sizes = false;   // tests: force the run-time-sized instantiations
    if (!m->wave_ok && !m->generic_ok) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "model exceeds the limits of both step kernels (nv <= 36, nu <= 128, pair dofs <= 20)"); }
    m->env_lds_bytes_w = w.lay.total * 4;
    if (m->wave_ok && m->env_lds_bytes_w > 64 * 1024) { myo_model_free(m); return fail(MYO_E_UNSUPPORTED, "wave kernel working set exceeds 64 KB of LDS"); }
    void* p1 = nullptr; void* p2 = nullptr;
    if (hipMalloc(&p1, sizeof(DevModel)) != hipSuccess || hipMalloc(&p2, sizeof(DevModelW)) != hipSuccess) { myo_model_free(m); return fail(MYO_E_NOMEM, "hipMalloc model structs"); }
    m->dev_allocs.push_back(p1); m->dev_allocs.push_back(p2);
    m->d_dm = (DevModel*)p1; m->d_dw = (DevModelW*)p2;
    if (hipMemcpy(p1, &d, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(p2, &w, sizeof(DevModelW), hipMemcpyHostToDevice) != hipSuccess) { myo_model_free(m); return fail(MYO_E_HIP, "upload model structs"); }
  }
  m->dims = myo_dims{S[0], S[1], S[2], S[3], S[4], S[8], S[7], d.nl, 0, m->wave_ok ? m->env_lds_bytes_w : m->env_lds_bytes, 64,
                     m->wave_ok ? (m->wave_cfg == 1 ? 32 : NCONW) : NCON, d.timestep};
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
  BA(d.flags, B) BA(d.diag, (size_t)B * 8) BA(d.elapsed, B) BA(d.episode, B)
  BA(b->d_tlo, b->ntarget_alloc) BA(b->d_thi, b->ntarget_alloc) BA(b->d_init, nq) BA(b->d_jlo, nv) BA(b->d_jhi, nv)
  BA(b->d_action, (size_t)B * nu)
  BA(d.fatigue, (size_t)B * 3 * nu)
  BA(b->d_initv, nv)
  { void* pw = nullptr; if ((rc = balloc(b, &pw, sizeof(DevWalk)))) { myo_batch_free(b); return rc; } b->d_walk = (DevWalk*)pw; }
  BA(b->d_stamps, (size_t)B * 12 * 2)
  BA(b->d_order, B)
  b->sched_stride = (B + 7) / 8 + 1;
  BA(b->d_sched, 32 + 8 * b->sched_stride)
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
  T.init_qvel = nullptr;
  T.task = c->task; T.frame_skip = c->frame_skip; T.reset_random = c->reset_random; T.target_generate = c->target_generate;
  T.ntarget = c->ntarget; T.ntip = c->ntip;
  for (int i = 0; i < 8; i++) T.tip_site[i] = c->tip_site[i];
  T.pose_thd = c->pose_thd; T.far_th = c->far_th; T.near_th = c->near_th;
  T.w_pose = c->w_pose; T.w_bonus = c->w_bonus; T.w_act_reg = c->w_act_reg; T.w_penalty = c->w_penalty; T.w_reach = c->w_reach;
  if (c->task == MYO_TASK_POSE) { if (c->ntarget != nv) return fail(MYO_E_ARG, "pose task: ntarget must equal nq"); T.obs_dim = 3 * nv + nu; }
  else if (c->task == MYO_TASK_REACH) { if (c->ntarget != 3 * c->ntip) return fail(MYO_E_ARG, "reach task: ntarget must be 3*ntip"); T.obs_dim = 2 * nv + 6 * c->ntip + nu; }
  else T.obs_dim = 0;
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
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(b->d_walk, &w, sizeof w, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->d_init, c->init_qpos, (size_t)nq * 4, hipMemcpyHostToDevice));
  if (c->init_qvel) HIPCHK(hipMemcpy(b->d_initv, c->init_qvel, (size_t)nv * 4, hipMemcpyHostToDevice));
  TaskDev& T = b->task;
  T.task = MYO_TASK_WALK; T.frame_skip = c->frame_skip; T.reset_random = 0; T.target_generate = 0; T.ntarget = 0; T.ntip = 0;
  T.obs_dim = w.obs_dim;
  T.init_qvel = c->init_qvel ? b->d_initv : nullptr;
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
    case MYO_F_ACTION: *p = b->d_action; *pitch = *width = nu; break;
    case MYO_F_FATIGUE: *p = d.fatigue; *pitch = *width = 3 * nu; break;
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
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)step_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)step_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)step_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)reach_obs_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  long long* st = b->d_stamps;
  if (!(G == 64 && m->wave_ok) && !m->generic_ok)
    return fail(MYO_E_UNSUPPORTED, "this model (tendon limits / free joint / equalities / plane contacts) needs the wave-per-env kernel (lanes = 64)");
  if (G == 64 && m->wave_ok) {
    static bool attr_w = false;
    if (!attr_w) {
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 4, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 4, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<24, 8, 32, 1, 4, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)step_kernel_w<36, 20, 32, 2, 2, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_w = true;
    }
    const int* order = nullptr;
    int Bn = b->db.B;
    // substep scheduler: MYO_SCHED=1 forces it, 0 disables it; default (auto) uses it where it was measured to pay: when the launch
    // holds at least twice as many envs as the chip holds waves of this kernel (MyoLeg: 8 waves per CU; +6 % at 4096 envs).  With as
    // many waves as envs every wave just re-takes its own env and only the overhead is left (MyoHand at 4096 envs: -15 %)
    static int sched_mode = -2, n_cu = 0;
    if (sched_mode == -2) {
      const char* e = getenv("MYO_SCHED"); sched_mode = e ? atoi(e) : -1;
      hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) n_cu = prop.multiProcessorCount;
      if (n_cu <= 0) n_cu = 256;
    }
    const int resident = n_cu * (m->wave_cfg == 1 ? 8 : 16);
    const bool sched_ok = !kflags && Bn >= 64 && Bn <= SCHED_ENV_MASK && nsub + (wk ? 1 : 0) <= 15 && nsub > 0;
    const bool sched = sched_ok && (sched_mode == 1 || (sched_mode == -1 && m->wave_cfg == 1 && Bn >= 2 * resident));
    if (b->balance && Bn >= 1024 && Bn % 4 == 0 && !kflags && !sched) {
      static int prio_mode = -1;
      if (prio_mode < 0) { const char* e = getenv("MYO_PRIO"); prio_mode = e ? atoi(e) : 2; }
      hipLaunchKernelGGL(balance_kernel, dim3(1), dim3(1024), 0, s, (const int*)b->db.diag, Bn, b->d_order, Bn / 4, prio_mode);
      order = b->d_order;
    }
    SchedDev S{b->d_sched, b->d_sched + 32, b->sched_stride, nsub + (wk ? 1 : 0)};
    if (sched) {
      hipLaunchKernelGGL(sched_init_kernel, dim3(1), dim3(1024), 0, s, (const int*)b->db.diag, Bn, S);
      int grid = Bn < resident ? Bn : resident;    // persistent waves: no more workgroups than the chip holds at once
      static int grid_override = -1;
      if (grid_override < 0) { const char* e = getenv("MYO_SCHED_GRID"); grid_override = e ? atoi(e) : 0; }
      if (grid_override > 0 && grid_override < grid) grid = grid_override;
      if (m->wave_cfg == 0)
        hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 4, true, 0>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, (const DevWalk*)nullptr, 0, S);
      else if (m->leg_sizes)
        hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, true, 2>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, wk, 0, S);
      else
        hipLaunchKernelGGL((step_kernel_w<36, 20, 32, 2, 2, true, 0>), dim3(grid), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                           (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, (const int*)nullptr, wk, 0, S);
    } else if (m->wave_cfg == 0 && m->hand_sizes)
      hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 4, false, 1>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, (const DevWalk*)nullptr, 0, S);
    else if (m->wave_cfg == 0)
      hipLaunchKernelGGL((step_kernel_w<24, 8, 32, 1, 4, false, 0>), dim3(Bn), dim3(64), (size_t)m->env_lds_bytes_w, s, (const DevModel*)m->d_dm,
                         (const DevModelW*)m->d_dw, b->db, action, actmap, nsub, st, order, (const DevWalk*)nullptr, 0, S);
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
  } else if (b->task.task == MYO_TASK_POSE) {
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
    rc = launch_step(b, b->d_action, MYO_ACTMAP_MUSCLE_SIGMOID, nsubsteps, s);
    if (rc) return rc;
    HIPCHK(hipEventRecord(b->kev[2 * (base + i) + 1], s));
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
