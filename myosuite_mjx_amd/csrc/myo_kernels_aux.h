// myo_kernels_aux.h -- small kernels: RNG, placement hint, random actions, policy inference, reset, observations.
// Part of the single translation unit myo_hip.hip (included there, in this order); not a stand-alone header.
#ifndef MYO_KERNELS_AUX_H
#define MYO_KERNELS_AUX_H

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
  // costs of this thread's envs: read once (the first four stay in registers, a launch of 4096 envs has exactly four per thread)
  int c4[4] = {0, 0, 0, 0};
  int cm = 1;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int e = t + 1024 * k; if (e < B) { c4[k] = diag[(size_t)e * 8 + 3]; cm = max(cm, c4[k]); } }
  for (int e = t + 4096; e < B; e += 1024) cm = max(cm, diag[(size_t)e * 8 + 3]);
  for (int off = 32; off > 0; off >>= 1) cm = max(cm, __shfl_xor(cm, off));
  if ((t & 63) == 0) atomicMax(&cmax_s, cm);
  __syncthreads();
  const float sc = 255.0f / (float)cmax_s;
#define BUCKET(c) (255 - min(255, (int)(sc * (float)(c))))     /* bucket 0 = heaviest */
#pragma unroll
  for (int k = 0; k < 4; k++) if (t + 1024 * k < B) atomicAdd(&hist[BUCKET(c4[k])], 1);
  for (int e = t + 4096; e < B; e += 1024) atomicAdd(&hist[BUCKET(diag[(size_t)e * 8 + 3])], 1);
  __syncthreads();
  if (t < 64) {   // exclusive prefix sum of the 256 bins by one wave: 4 bins per lane, then a wave scan of the lane totals
    const int h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    int tot = h0 + h1 + h2 + h3, inc = tot;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (t >= off) inc += v; }
    const int base = inc - tot;
    start[4 * t] = base; start[4 * t + 1] = base + h0; start[4 * t + 2] = base + h0 + h1; start[4 * t + 3] = base + h0 + h1 + h2;
  }
  __syncthreads();
  auto place = [&](int e, int c) {
    const int r = atomicAdd(&start[BUCKET(c)], 1);          // rank by descending cost (ties in arbitrary order)
    const int q = r / nslot, i = r - q * nslot;
    const int wg = q * nslot + ((q & 1) ? nslot - 1 - i : i);   // snake: slot i gets ranks i, 2*nslot-1-i, 2*nslot+i, ...
    const int pr = prio_mode == 2 ? (q < 3 ? 3 - q : 0) : (prio_mode == 1 ? (q == 0 ? 1 : 0) : (prio_mode == 3 ? (q < 2 ? 1 : 0) : 0));
    order[wg < B ? wg : r] = e | (pr << 28);                 // env id + issue priority of its cost quartile
  };
#pragma unroll
  for (int k = 0; k < 4; k++) if (t + 1024 * k < B) place(t + 1024 * k, c4[k]);
  for (int e = t + 4096; e < B; e += 1024) place(e, diag[(size_t)e * 8 + 3]);
#undef BUCKET
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
__device__ __forceinline__ bool reset_body(const DevBatch& Bt, const TaskDev& T, int nq, int nv, int nu, const float* qpos0, const uint8_t* mask, uint64_t seed,
                                           int env_offset, int auto_max, const int e, const int lane) {
  if (auto_max > 0) { if (!(Bt.done[e] > 0.f || Bt.elapsed[e] >= auto_max)) return false; }
  else if (mask && !mask[e]) return false;
  seed += 0x632BE59BD9B4E019ull * (uint64_t)Bt.episode[e];  // a fresh RNG stream per (env, episode)
  __syncthreads();                                            // every lane has read done / elapsed / episode before lane 0 updates them
  if (lane == 0) { Bt.episode[e] += 1; Bt.elapsed[e] = 0; Bt.done[e] = 0.f; Bt.time[e] = 0; }
  uint64_t ge = (uint64_t)(e + env_offset);
  // walk reset_type "random" (walk_v0.py:316-332): one coin per episode picks the keyframe
  const bool alt = T.init_qpos_alt && u01(seed ^ 0xA24BAED4963EE407ull, ge, 6) >= 0.5f;
  for (int i = lane; i < nq; i += 64) {
    float q = alt ? T.init_qpos_alt[i] : (T.init_qpos ? T.init_qpos[i] : qpos0[i]);
    if (T.init_qpos_alt && T.reset_noise_std > 0.f && !(i >= 2 && i <= 6)) {   // every coordinate but the root height and quaternion
      const float u1 = fmaxf(u01(seed ^ 0x9FB21C651E98DF25ull, ge * 4096 + i, 7), 1e-7f), u2 = u01(seed ^ 0x2545F4914F6CDD1Dull, ge * 4096 + i, 7);
      q += T.reset_noise_std * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    if (T.reset_random) q = T.jnt_lo[i] + (T.jnt_hi[i] - T.jnt_lo[i]) * u01(seed, ge * 4096 + i, 1);   // nq == nv checked at configure
    else if (T.rnd) {   // init + U(noise), clipped (walk_v0.py:152-167)
      const float nl = T.rnd[i], nh = T.rnd[nq + i];
      q = fminf(fmaxf(q + nl + (nh - nl) * u01(seed, ge * 4096 + i, 1), T.rnd[2 * nq + i]), T.rnd[3 * nq + i]);
    }
    Bt.qpos[(size_t)e * nq + i] = q;
  }
  for (int i = lane; i < nv; i += 64) {
    Bt.qvel[(size_t)e * nv + i] = (alt && T.init_qvel_alt) ? T.init_qvel_alt[i] : (T.init_qvel ? T.init_qvel[i] : 0.f);
    Bt.warm[(size_t)e * nv + i] = 0;
  }
  for (int i = lane; i < nu; i += 64) {
    Bt.act[(size_t)e * nu + i] = 0; Bt.ctrl[(size_t)e * nu + i] = 0;
    // fatigue compartments: all motor units resting (CumulativeFatigue.reset defaults, fatigue.py:130-134)
    float MA = 0.f, MR = 1.f, MF = 0.f;
    if (T.fatigue_mode == 1) {   // fatigue_reset_random (fatigue.py:115-122)
      const float nf = u01(seed ^ 0x94D049BB133111EBull, ge * 4096 + i, 8), ap = u01(seed ^ 0xBF58476D1CE4E5B9ull, ge * 4096 + i, 8);
      MA = nf * ap; MR = nf * (1.f - ap); MF = 1.f - nf;
    } else if (T.fatigue_mode == 2 && T.fatigue_vec) { MF = T.fatigue_vec[i]; MR = 1.f - MF; }   // fatigue_reset_vec (:124-130)
    Bt.fatigue[(size_t)e * 3 * nu + i] = MA; Bt.fatigue[(size_t)e * 3 * nu + nu + i] = MR; Bt.fatigue[(size_t)e * 3 * nu + 2 * nu + i] = MF;
  }
  for (int i = lane; i < T.ntarget; i += 64) {
    float lo = T.target_lo[i], hi = T.target_hi[i];
    Bt.target[(size_t)e * T.ntarget + i] = T.target_generate ? lo + (hi - lo) * u01(seed, ge * 4096 + 2048 + i, 2) : lo;
  }
  if (T.gsize_type && Bt.gsize && lane == 0) {
    // ObjHoldRandomEnvV0.reset (obj_hold_v0.py:133-139): a fresh size for the object's geom per episode (its mass and inertia stay)
    float sz[3];
    for (int k = 0; k < 3; k++) sz[k] = T.gsize_lo[k] + (T.gsize_hi[k] - T.gsize_lo[k]) * u01(seed ^ 0xD1B54A32D192ED03ull, ge * 8 + k, 5);
    float rb = T.gsize_type == GEOM_ELLIPSOID ? fmaxf(sz[0], fmaxf(sz[1], sz[2])) : (T.gsize_type == GEOM_CAPSULE ? sz[0] + sz[1] : (T.gsize_type == GEOM_CYLINDER ? sqrtf(sz[0] * sz[0] + sz[1] * sz[1]) : sz[0]));
    float* G = Bt.gsize + 4 * (size_t)e;
    G[0] = sz[0]; G[1] = sz[1]; G[2] = sz[2]; G[3] = rb;
  }
  if (T.terrain && Bt.hfield) {
    // TerrainEnvV0.reset (walk_v0.py:563-622): a fresh 100 x 100 elevation grid per episode (in units of the height field's z scale).
    // Distribution parity only for the random draws, as for every reset.
    float* H = Bt.hfield + (size_t)e * T.hf_n;
    const int n = T.hf_n;
    if (T.terrain == MYO_TERRAIN_ROUGH) {
      // rough ~ U(-0.5, 0.5)^n, min-max normalised over the grid, * 0.08 - 0.02 (:563-567)
      float mn = 1e30f, mx = -1e30f;
      for (int i = lane; i < n; i += 64) { float u = u01(seed ^ 0x9E3779B97F4A7C15ull, ge * 16384 + i, 3); mn = fminf(mn, u); mx = fmaxf(mx, u); }
      for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
      const float inv = 1.0f / fmaxf(mx - mn, 1e-12f);
      for (int i = lane; i < n; i += 64) H[i] = (u01(seed ^ 0x9E3779B97F4A7C15ull, ge * 16384 + i, 3) - mn) * inv * 0.08f - 0.02f;
    } else {
      const float sc = T.terrain_lo + (T.terrain_hi - T.terrain_lo) * u01(seed ^ 0x9E3779B97F4A7C15ull, ge * 16384, 4);
      for (int i = lane; i < n; i += 64) {
        const int k = n - 1 - i;             // np.flip(grid, [0, 1]) of a row-major grid = the flat array reversed
        float v;
        if (T.terrain == MYO_TERRAIN_HILLY) {
          // 3000 flat cells at the top level, then -2 + 0.5 (sin(linspace(0, 3 pi, 7000) + pi/2) - 1), min-max normalised: 0.5 + 0.5 cos(t) (:569-592)
          v = k < 3000 ? 1.0f : 0.5f + 0.5f * cosf((float)(k - 3000) * (3.0f * 3.14159265358979f / 6999.0f));
        } else {
          // 52 flat rows, then 12 stairs of 4 rows each, 0.1 high, normalised by 2 + 0.1 * 12 (:594-620)
          const int row = k / 100;
          v = row < 52 ? 0.0f : 0.1f * (float)((row - 52) / 4) / 3.2f;
        }
        H[i] = v * sc;
      }
    }
  }
  return true;
}
__global__ void __launch_bounds__(64) reset_kernel(DevBatch Bt, TaskDev T, int nq, int nv, int nu, const float* qpos0, const uint8_t* mask, uint64_t seed,
                                                  int env_offset, int auto_max) {
  if ((int)blockIdx.x >= Bt.B) return;
  reset_body(Bt, T, nq, nv, nu, qpos0, mask, seed, env_offset, auto_max, blockIdx.x, threadIdx.x);
}

// observation + reward (pose_v0.py:98-138, obs_vec_dict.py:86-98); one 64-lane workgroup per env, rows written coalesced
__device__ __forceinline__ void obs_body(const DevModel& M, const DevBatch& Bt, const TaskDev& T, int obs_only, const int e, const int lane) {
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
    for (int i = lane; i < nu; i += 64) { float ai = a[i]; const int sl = M.act_obs[i]; if (sl >= 0) { o[3 * nv + sl] = ai; act2 += ai * ai; } }
    if (obs_only) return;
    float dist = sqrtf(wave_sum(err2));
    float actn = sqrtf(wave_sum(act2)) / (float)(M.na_obs > 0 ? M.na_obs : 1);
    if (lane == 0) {
      float bonus = (dist < T.pose_thd ? 1.f : 0.f) + (dist < 1.5f * T.pose_thd ? 1.f : 0.f);
      float pen = dist > T.far_th ? -1.f : 0.f;
      Bt.reward[e] = T.w_pose * (-dist) + T.w_bonus * bonus + T.w_act_reg * (-actn) + T.w_penalty * pen;
      Bt.solved[e] = dist < T.pose_thd ? 1.f : 0.f;
      Bt.done[e] = dist > T.far_th ? 1.f : 0.f;
    }
  } else if (T.task == MYO_TASK_STAND) {
    // walk_v0.py:68-127 (ReachEnvV0 on the legs).  The tip site rides on the free root link: world position = root position + R(root quat) tip_lpos
    const int nq = T.nq;
    const float* qq = Bt.qpos + (size_t)e * nq;
    float vel2 = 0.f, act2 = 0.f, err2 = 0.f;
    for (int i = lane; i < nq; i += 64) o[i] = qq[i];
    for (int i = lane; i < nv; i += 64) { const float vd = v[i] * dt; o[nq + i] = vd; vel2 += vd * vd; }
    if (lane < 3) {
      float R[9];
      const float qn = 1.0f / sqrtf(qq[3] * qq[3] + qq[4] * qq[4] + qq[5] * qq[5] + qq[6] * qq[6]);
      const float uq[4] = {qq[3] * qn, qq[4] * qn, qq[5] * qn, qq[6] * qn};
      quat2mat(R, uq);
      const float p = qq[lane] + R[3 * lane] * T.tip_lpos[0] + R[3 * lane + 1] * T.tip_lpos[1] + R[3 * lane + 2] * T.tip_lpos[2] + M.origin[lane];
      const float er = Bt.target[(size_t)e * T.ntarget + lane] - p;
      o[nq + nv + lane] = p; o[nq + nv + 3 + lane] = er;
      err2 = er * er;
    }
    for (int i = lane; i < nu; i += 64) { float ai = a[i]; const int sl = M.act_obs[i]; if (sl >= 0) { o[nq + nv + 6 + sl] = ai; act2 += ai * ai; } }
    if (obs_only) return;
    const float dist = sqrtf(wave_sum(err2)), veld = sqrtf(wave_sum(vel2));
    const float actn = sqrtf(wave_sum(act2)) / (float)(M.na_obs > 0 ? M.na_obs : 1);
    if (lane == 0) {
      const float far_th = Bt.time[e] > 2.f * dt ? T.far_th : 1e30f;
      const float bonus = (dist < 2.f * T.near_th ? 1.f : 0.f) + (dist < T.near_th ? 1.f : 0.f);
      const float pen = dist > far_th ? 1.f : 0.f;
      Bt.reward[e] = T.w_reach * (10.0f - dist - 10.0f * veld) + T.w_bonus * bonus + T.w_act_reg * (-100.0f * actn) + T.w_penalty * (-pen);
      Bt.solved[e] = dist < T.near_th ? 1.f : 0.f;
      Bt.done[e] = pen;
    }
  } else if (T.task == MYO_TASK_TRACK) {
    // TrackEnv.get_obs (mjx/myodm_v0.py:297-304): [qpos, qvel]; reward / done belong to the step kernel's epilogue (they need the reference row
    // of the step and the body frames of its last substep) and are not recomputed here
    const int nq = T.nq;
    const float* qq = Bt.qpos + (size_t)e * nq;
    for (int i = lane; i < nq; i += 64) o[i] = qq[i];
    for (int i = lane; i < nv; i += 64) o[nq + i] = v[i];
  } else if (T.task == MYO_TASK_HOLD) {
    // obj_hold_v0.py:66-118.  The object's site sits at the origin of its free body, whose world position is the free joint's qpos
    // (free-floating models are not origin-shifted); the goal site is world-fixed = the target row
    const int nq = T.nq, nqh = nq - 7, nvh = nv - 6;
    const float* qq = Bt.qpos + (size_t)e * nq;
    for (int i = lane; i < nqh; i += 64) o[i] = qq[i];
    for (int i = lane; i < nvh; i += 64) o[nqh + i] = v[i] * dt;
    float err2 = 0.f, act2 = 0.f;
    if (lane < 3) {
      float p = qq[nqh + lane] + M.origin[lane], er = Bt.target[(size_t)e * T.ntarget + lane] - p;
      o[nqh + nvh + lane] = p; o[nqh + nvh + 3 + lane] = er;
      err2 = er * er;
    }
    for (int i = lane; i < nu; i += 64) { float ai = a[i]; const int sl = M.act_obs[i]; if (sl >= 0) { o[nqh + nvh + 6 + sl] = ai; act2 += ai * ai; } }
    if (obs_only) return;
    float dist = sqrtf(wave_sum(err2));
    float actn = sqrtf(wave_sum(act2)) / (float)(M.na_obs > 0 ? M.na_obs : 1);
    if (lane == 0) {
      float bonus = (dist < 2.f * T.near_th ? 1.f : 0.f) + (dist < T.near_th ? 1.f : 0.f);
      float drop = dist > T.far_th ? 1.f : 0.f;
      Bt.reward[e] = T.w_reach * (-dist) + T.w_bonus * bonus + T.w_act_reg * (-actn) + T.w_penalty * (-drop);
      Bt.solved[e] = dist < T.near_th ? 1.f : 0.f;
      Bt.done[e] = drop;
    }
  }
}
__global__ void __launch_bounds__(64) obs_kernel(DevModel M, DevBatch Bt, TaskDev T, int obs_only, int reset_only) {
  const int e = blockIdx.x;
  if (e >= Bt.B) return;
  if (reset_only && Bt.elapsed[e] != 0) return;    // refresh only the rows of envs an auto-reset just touched
  obs_body(M, Bt, T, obs_only, e, threadIdx.x);
}
// observation / reward / done of the stepped state, gym TimeLimit + done auto-reset, and the first observation of the new episode for the
// envs that were reset: the three launches of the per-step epilogue (obs_kernel, reset_kernel, obs_kernel(reset_only)) in one -- on
// the critical path between two step kernels every launch costs a few microseconds of dispatch gap
__global__ void __launch_bounds__(64) post_kernel(DevModel M, DevBatch Bt, TaskDev T, int nq, const float* qpos0, uint64_t seed, int env_offset, int auto_max) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= Bt.B) return;
  obs_body(M, Bt, T, 0, e, lane);
  __syncthreads();                       // reward / done of this env written (lane 0) before every lane tests them
  if (reset_body(Bt, T, nq, M.nv, M.nu, qpos0, nullptr, seed, env_offset, auto_max, e, lane)) {
    __syncthreads();                     // the new state rows are complete before they are read back
    obs_body(M, Bt, T, 1, e, lane);
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
  GFOR(i, nu) { float a = Bt.act[(size_t)env * nu + i]; const int sl = M.act_obs[i]; if (sl >= 0) { actn += a * a; if (valid) o[2 * nv + 6 * T.ntip + sl] = a; } }
  actn = sqrtf(grp_sum<G>(actn)) / (float)(M.na_obs > 0 ? M.na_obs : 1);
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

// ------------------------------------------------------------------------------------------------
// VALU issue-rate probe (bench.py roofline): every wave runs `iters` x 256 independent v_fma_f32 (8 accumulator chains, so that a wave's
// own dependent-issue latency is not the limit), with W waves resident on every SIMD (workgroup = 4 W waves, one workgroup per CU forced
// by its LDS request).  wave-instructions / elapsed time = what the chip's VALUs can issue at that occupancy, at the clock the chip
// actually holds under this load.
#if MYO_POISON
// diagnostic build: fills the private (scratch) memory of every wave slot with NaNs before each step launch, so that a register spill
// which is stored under a partial exec mask and reloaded under a wider one (or any other read of a scratch word the kernel did not write)
// turns into wrong numbers instead of depending on what earlier kernels left there
__global__ void __launch_bounds__(64) scratch_poison_kernel(float* sink, int n) {
  volatile float buf[512];
  for (int i = 0; i < 512; i++) buf[i] = __int_as_float(0x7fc00000 | i);
  if (n < 0) sink[threadIdx.x] = buf[(-n) & 511];
}
// ... and the register files: a wave starts with whatever the previous wave left in its VGPR / AGPR / SGPR allocation, so a read of a
// register the kernel never wrote (a value defined under a lane predicate and read by other lanes through a shuffle / v_readlane, an
// `implicit-def`) is order-dependent in exactly the way a stale scratch word is.  One wave per SIMD owning all 512 vector registers writes
// NaN patterns into every one of them; a second kernel at 8 waves per SIMD does the same for the scalar registers.
#define MYO_R1(p, n) p #n
#define MYO_R10(M, p, t) M(p, t##0) M(p, t##1) M(p, t##2) M(p, t##3) M(p, t##4) M(p, t##5) M(p, t##6) M(p, t##7) M(p, t##8) M(p, t##9)
#define MYO_R100(M, p, h) MYO_R10(M, p, h##0) MYO_R10(M, p, h##1) MYO_R10(M, p, h##2) MYO_R10(M, p, h##3) MYO_R10(M, p, h##4) \
                          MYO_R10(M, p, h##5) MYO_R10(M, p, h##6) MYO_R10(M, p, h##7) MYO_R10(M, p, h##8) MYO_R10(M, p, h##9)
#define MYO_R256(M, p) MYO_R10(M, p, ) MYO_R10(M, p, 1) MYO_R10(M, p, 2) MYO_R10(M, p, 3) MYO_R10(M, p, 4) MYO_R10(M, p, 5) MYO_R10(M, p, 6) \
                       MYO_R10(M, p, 7) MYO_R10(M, p, 8) MYO_R10(M, p, 9) MYO_R100(M, p, 1) MYO_R10(M, p, 20) MYO_R10(M, p, 21) MYO_R10(M, p, 22) \
                       MYO_R10(M, p, 23) MYO_R10(M, p, 24) M(p, 250) M(p, 251) M(p, 252) M(p, 253) M(p, 254) M(p, 255)
#define MYO_R102(M, p) MYO_R10(M, p, ) MYO_R10(M, p, 1) MYO_R10(M, p, 2) MYO_R10(M, p, 3) MYO_R10(M, p, 4) MYO_R10(M, p, 5) MYO_R10(M, p, 6) \
                       MYO_R10(M, p, 7) MYO_R10(M, p, 8) MYO_R10(M, p, 9) M(p, 100) M(p, 101)
#define MYO_WV(p, n) "v_mov_b32 v" #n ", %0\n\t"
#define MYO_WA(p, n) "v_accvgpr_write_b32 a" #n ", %0\n\t"
#define MYO_WS(p, n) "s_mov_b32 s" #n ", 0x7fc0beef\n\t"
#define MYO_CL(p, n) , p #n
__global__ void __launch_bounds__(64, 1) vgpr_poison_kernel(int bits) {
  asm volatile(MYO_R256(MYO_WV, "") MYO_R256(MYO_WA, "") "s_nop 0" : : "s"(bits) : "memory" MYO_R256(MYO_CL, "v") MYO_R256(MYO_CL, "a"));
}
__global__ void __launch_bounds__(64, 8) sgpr_poison_kernel() {
  asm volatile(MYO_R102(MYO_WS, "") "s_nop 0" : : : "memory" MYO_R102(MYO_CL, "s"));
}
#endif
__global__ void __launch_bounds__(1024) valu_probe_kernel(float* out, int iters, float seed) {
  extern __shared__ float lds_dummy[];
  float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f, a5 = seed + 5.f, a6 = seed + 6.f, a7 = seed + 7.f;
  const float m = 1.0000001f, c = 1e-9f * (float)threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 32; u++) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
  }
  if (out) out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + (threadIdx.x == 4096 ? lds_dummy[0] : 0.f);
}

#endif  // MYO_KERNELS_AUX_H
