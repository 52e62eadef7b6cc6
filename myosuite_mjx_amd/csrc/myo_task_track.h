// myo_task_track.h -- MyoDM TrackEnv as a task of the TRK step kernel (MYO_TASK_TRACK): device side of mjx/myodm_v0.py:185-304.
// Part of the single translation unit myo_hip.hip (included before myo_kernel_wave.h); not a stand-alone header.
//
//   prologue (once per env step, before the first substep):   ctrl = (action + 1) (hi - lo) / 2 + lo                     (:272-275)
//                                                              reference row at the PRE-step time + motion_start_time      (:278-279)
//   epilogue (after the last substep):                         obs = [qpos, qvel]                                          (:297-304)
//                                                              reward / done / metrics (compute_reward)                    (:185-267)
//                                                              masked reset of the envs that are done (autoreset)
// The reference lookup restates mjx/reference_motion.py == logger/reference_motion.py, quirks included (times rounded to 4 decimals and compared
// for equality; between frames  blend = t - T[i] / dt  and  robot = (1 - blend) ** robot[i] + blend robot[i + 1]; `linear` switches to the evident
// intent), in float64 like the reference's arrays; the result is cast to float32 where the reward consumes it.
#ifndef MYO_TASK_TRACK_H
#define MYO_TASK_TRACK_H

// lane i < robot_dim: robot[i] (and robot_vel[i]); lane robot_dim + j: object[j].  Written to this env's row of K.ref (robot | robot_vel | object).
__device__ __forceinline__ void track_lookup(const DevTrack& K, int env, int genv, float time, int elapsed, int lane) {
  const int nr = K.robot_dim, no = K.object_dim;
  float* out = K.ref + (size_t)env * K.ref_pitch;
  const bool isr = lane < nr, iso = lane >= nr && lane < nr + no;
  const int j = isr ? lane : lane - nr;
  if (K.ref_type == 0) {                                               // FIXED: the one row
    if (isr) { out[j] = (float)K.robot[j]; if (K.has_vel) out[nr + j] = (float)K.robot_vel[j]; }
    if (iso) out[2 * nr + j] = (float)K.object[j];
    return;
  }
  if (K.ref_type == 1) {                                               // RANDOM: uniform between the two rows, a fresh draw per env step
    const uint64_t key = (uint64_t)genv * 4096 + (uint64_t)lane;
    if (isr) {
      const double u = (double)u01(K.seed, key, (uint64_t)elapsed * 4 + 0), a = K.robot[j], b = K.robot[nr + j];
      out[j] = (float)(a + (b - a) * u);
      if (K.has_vel) { const double u2 = (double)u01(K.seed, key, (uint64_t)elapsed * 4 + 1), c = K.robot_vel[j], d = K.robot_vel[nr + j]; out[nr + j] = (float)(c + (d - c) * u2); }
    }
    if (iso) { const double u = (double)u01(K.seed, key, (uint64_t)elapsed * 4 + 2), a = K.object[j], b = K.object[no + j]; out[2 * nr + j] = (float)(a + (b - a) * u); }
    return;
  }
  // TRACK
  const int H = K.horizon;
  const double t = rint(((double)time + K.start_time) * 1e4) / 1e4;
  int lo = 0, hi = H;                                                  // number of frame times <= t (searchsorted side = "right")
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (K.T[mid] <= t) lo = mid + 1; else hi = mid; }
  int idx = min(max(lo - 1, 0), H - 1);
  const bool held = t >= K.T[H - 1];
  if (held) idx = H - 1;
  const bool exact = held || K.T[idx] == t;
  const int nxt = min(idx + 1, H - 1);
  const double dt = exact ? 1.0 : K.T[nxt] - K.T[idx];
  const double blend = K.linear ? (t - K.T[idx]) / dt : t - K.T[idx] / dt;
  if (isr) {
    const double* tabs[2] = {K.robot, K.robot_vel};
    for (int w = 0; w < (K.has_vel ? 2 : 1); w++) {
      const double* A = tabs[w];
      double v;
      if (K.robot_horizon <= 1) v = A[j];
      else if (exact) v = A[(size_t)idx * nr + j];
      else if (K.linear) v = (1.0 - blend) * A[(size_t)idx * nr + j] + blend * A[(size_t)nxt * nr + j];
      else v = pow(1.0 - blend, A[(size_t)idx * nr + j]) + blend * A[(size_t)nxt * nr + j];
      out[w * nr + j] = (float)v;
    }
  }
  if (iso) {
    double v;
    if (K.object_horizon <= 1) v = K.object[j];
    else if (exact) v = K.object[(size_t)idx * no + j];
    else v = (1.0 - blend) * K.object[(size_t)idx * no + j] + blend * K.object[(size_t)nxt * no + j];
    out[2 * nr + j] = (float)v;
  }
}

// rotation matrix -> unit quaternion with w >= 0 (mjx/quat_math.py:108-169: four branches by the largest diagonal term)
__device__ __forceinline__ void track_mat2quat(const float* M, float* q) {
  const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
  float w, x, y, z;
  if (m22 >= 0.f) {
    if (!(m00 < -m11)) { const float s = 2.f * sqrtf(fmaxf(1.f + m00 + m11 + m22, 1e-30f)); w = 0.25f * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s; }
    else { const float s = 2.f * sqrtf(fmaxf(1.f - m00 - m11 + m22, 1e-30f)); w = (m10 - m01) / s; x = (m20 + m02) / s; y = (m12 + m21) / s; z = 0.25f * s; }
  } else {
    if (m00 > m11) { const float s = 2.f * sqrtf(fmaxf(1.f + m00 - m11 - m22, 1e-30f)); w = (m21 - m12) / s; x = 0.25f * s; y = (m01 + m10) / s; z = (m20 + m02) / s; }
    else { const float s = 2.f * sqrtf(fmaxf(1.f - m00 + m11 - m22, 1e-30f)); w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25f * s; z = (m12 + m21) / s; }
  }
  const float sg = w < 0.f ? -1.f : 1.f;
  q[0] = sg * w; q[1] = sg * x; q[2] = sg * y; q[3] = sg * z;
}

// compute_reward (:185-267) on the stepped state: qpos / qvel are the post-step rows (LDS), the body frames those of the last substep's
// position stage (what an MJX pipeline_state holds after mjx.step).  lpos / lmat: link frames in LDS (origin-shifted), org = model origin.
// Every lane computes the (uniform) scalars; lane 0 writes.  Returns done.
template <class SumF>
__device__ __forceinline__ float track_reward(const DevTrack& K, const DevBatch& Bt, int env, int lane, const float* qpos, const float* qvel,
                                              const float* lpos, const float* lmat, const float* org, SumF wsum) {
  const int nr = K.robot_dim;
  const float* R = K.ref + (size_t)env * K.ref_pitch;
  float qe = 0.f, ve = 0.f;
  if (lane < nr) {
    const float d = qpos[lane] - R[lane];
    qe = d * d;
    if (K.has_vel) { const float dv = qvel[lane] - R[nr + lane]; ve = dv * dv; }
  }
  const float q2 = wsum(qe), v2 = wsum(ve);
  const float* tc = R + 2 * nr;            // target object pose: com (3) | quaternion (4)
  // xipos / ximat of the object body and xipos of the wrist body from their link frames
  float com[3], Rm[9], wr[3];
  {
    const float *Lp = lpos + 3 * K.obj_link, *Lr = lmat + 9 * K.obj_link;
    float v[3];
    matvec(v, Lr, K.obj_p);
    com[0] = Lp[0] + org[0] + v[0]; com[1] = Lp[1] + org[1] + v[1]; com[2] = Lp[2] + org[2] + v[2];
    matmul3(Rm, Lr, K.obj_R);
    const float *Wp = lpos + 3 * K.wrist_link, *Wr = lmat + 9 * K.wrist_link;
    matvec(v, Wr, K.wrist_p);
    wr[0] = Wp[0] + org[0] + v[0]; wr[1] = Wp[1] + org[1] + v[1]; wr[2] = Wp[2] + org[2] + v[2];
  }
  float a[4];
  track_mat2quat(Rm, a);
  const float e0 = tc[0] - com[0], e1 = tc[1] - com[1], e2 = tc[2] - com[2];
  const float obj_com_err = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  // rotation_distance(curr, targ, euler = False) = |quatDiff2Vel(targ, curr, 1)[0]| (:180-183): diff = curr * conj(targ)
  const float b[4] = {tc[3], -tc[4], -tc[5], -tc[6]};
  const float d0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], d1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
              d2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], d3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  const float obj_rot_err = fabsf(2.f * atan2f(sqrtf(d1 * d1 + d2 * d2 + d3 * d3), d0)) / 3.14159265358979323846f;
  const float obj_reward = expf(-K.obj_err_scale * obj_com_err) * expf(-K.obj_err_scale * obj_rot_err);
  const float lift_bonus = (tc[2] >= K.lift_z && com[2] >= K.lift_z) ? 1.f : 0.f;
  const float qpos_reward = expf(-K.qpos_err_scale * q2);
  const float qvel_reward = K.has_vel ? expf(-K.qvel_err_scale * v2) : 1.f;      // exp(-0.1 * norm2([0])) = 1 (:208-218)
  const float pose_reward = K.qpos_w * qpos_reward, vel_reward = K.qvel_w * qvel_reward;
  const float b0 = com[0] - wr[0], b1 = com[1] - wr[1], b2 = com[2] - wr[2];
  const float base_error = sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
  const float base_reward = expf(-K.base_err_scale * base_error);
  bool term = false;
  if (K.term_obj) term = term || (obj_com_err * obj_com_err >= K.obj_fail2) || (base_error * base_error >= K.base_fail2);
  if (K.term_pose) term = term || (q2 >= K.qpos_fail);
  const float done = term ? 1.f : 0.f;
  const float m_pose = pose_reward + vel_reward, m_obj = obj_reward + base_reward, m_bonus = K.lift_bonus_mag * lift_bonus;
  float rew = K.w_pose * m_pose;
  rew = rew + K.w_object * m_obj;
  rew = rew + K.w_bonus * m_bonus;
  rew = rew + K.w_penalty * done;
  if (lane == 0) {
    Bt.reward[env] = rew; Bt.done[env] = done;
    float* mt = K.metrics + 4 * (size_t)env;
    mt[0] = m_pose; mt[1] = m_obj; mt[2] = m_bonus; mt[3] = done;
  }
  return done;
}

#endif  // MYO_TASK_TRACK_H
