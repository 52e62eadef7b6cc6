// myo_physics.h -- model pieces shared by both step kernels: tendon wrapping, muscle model, action map, impedance, frames, MPR.
// Part of the single translation unit myo_hip.hip (included there, in this order); not a stand-alone header.
#ifndef MYO_PHYSICS_H
#define MYO_PHYSICS_H

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
  bool crossing[2];     // the candidate's two straight pieces intersect: it is rejected, also when it is the better of the two
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float sgn = i == 0 ? 1.f : -1.f;
    sol[i][0] = (d[0] * sqr + sgn * rad * d[1] * s0) / sq0;
    sol[i][1] = (d[1] * sqr - sgn * rad * d[0] * s0) / sq0;
    sol[i][2] = (d[2] * sqr - sgn * rad * d[3] * s1) / sq1;
    sol[i][3] = (d[3] * sqr + sgn * rad * d[2] * s1) / sq1;
    if (has_side) {
      // MuJoCo scores a candidate by the direction of the sum of its two tangent points against the side site.  When the tangent points are
      // nearly antipodal (a half-turn wrap) that sum is the difference of two almost opposite vectors: in float32 its direction is round-off,
      // the score any number in [-1, 1], and the half-turn candidate can beat the right one -- a 1 cm jump of the tendon length (found in round 3
      // on tendon UI_UB4: |sum| = 1.6e-6 r).  The sum is always perpendicular to the chord between the tangent points (both have length r), so
      // whichever of the two is longer gives the direction; the sum then only contributes its sign, which is what is undetermined at a half turn.
      float t0 = sol[i][0] + sol[i][2], t1 = sol[i][1] + sol[i][3];
      const float c0 = sol[i][0] - sol[i][2], c1 = sol[i][1] - sol[i][3];
      const float n2 = t0 * t0 + t1 * t1, c2 = c0 * c0 + c1 * c1;
      if (n2 >= c2) {
        const float n = sqrtf(n2);
        if (n > MINVALF) { t0 /= n; t1 /= n; }
        good[i] = t0 * sd[0] + t1 * sd[1];
      } else {
        const float inv = __builtin_amdgcn_rsqf(c2), sg = (t1 * c0 - t0 * c1) < 0.f ? -1.f : 1.f;   // unit perpendicular of the chord (-c1, c0) / |c|
        good[i] = sg * (c0 * sd[1] - c1 * sd[0]) * inv;
      }
    } else {
      float t0 = sol[i][0] - sol[i][2], t1 = sol[i][1] - sol[i][3];
      good[i] = -(t0 * t0 + t1 * t1);
    }
    // a grazing solution (tangent points closer than 1e-3 rad) makes the segment-intersection test
    // meaningless in float; skip it there (changes the length by O(r*1e-9), see DESIGN.md "float safeguards")
    float gz0 = sol[i][0] - sol[i][2], gz1 = sol[i][1] - sol[i][3];
    bool grazing = gz0 * gz0 + gz1 * gz1 < 1e-6f * sqr;
    crossing[i] = !grazing && is_intersect(d, sol[i], d + 2, sol[i] + 2);
    if (crossing[i]) good[i] = -10000.f;
  }
  int i = good[0] > good[1] ? 0 : 1;
#pragma unroll
  for (int k = 0; k < 4; k++) pnt[k] = i == 0 ? sol[0][k] : sol[1][k];
  if (i == 0 ? crossing[0] : crossing[1]) return -1;      // the same test MuJoCo repeats on the chosen candidate: reuse its result
  return rad * acosf(clipf((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqr, -1.f, 1.f));
}

// Short polynomial forms of the inverse / direct trigonometry of the inside-wrap Newton solve, which 15 of MyoHand's 63 wrapping segments run
// every substep (the library calls carry range reduction and special cases this solve never needs: asinf ~30, sincosf ~45 instructions).
// asin on [0, 1]: pi/2 - sqrt(1 - x) P7(x) (Abramowitz & Stegun 4.4.46, |error| <= 2e-8 before float rounding); sin / cos on [0, pi/2]:
// the single-precision minimax polynomials on [0, pi/4] with the complement swap above pi/4.  The solve stops at |f| <= 1e-6.
// -DMYO_EXACT_TRIG=1 (A/B build, tools/gpu_trig_ab.py): the library functions instead, to attribute parity drift to these forms
#ifndef MYO_EXACT_TRIG
#define MYO_EXACT_TRIG 0
#endif
__device__ __forceinline__ float asin01f(float x) {
  if (MYO_EXACT_TRIG) return asinf(fminf(x, 1.f));
  float p = -0.0012624911f;
  p = p * x + 0.0066700901f; p = p * x - 0.0170881256f; p = p * x + 0.0308918810f; p = p * x - 0.0501743046f;
  p = p * x + 0.0889789874f; p = p * x - 0.2145988016f; p = p * x + 1.5707963050f;
  return 1.57079632679f - sqrtf(fmaxf(0.f, 1.f - x)) * p;
}
__device__ __forceinline__ float acos11f(float x) {       // acos on [-1, 1]
  const float a = asin01f(fabsf(x));
  return x >= 0.f ? 1.57079632679f - a : 1.57079632679f + a;
}
__device__ __forceinline__ void sincos_q1f(float th, float* sn, float* cs) {   // th in [0, pi/2]
  if (MYO_EXACT_TRIG) { sincosf(th, sn, cs); return; }
  const bool hi = th > 0.785398163f;
  const float r = hi ? 1.57079632679f - th : th, z = r * r;
  const float s = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const float c = 1.f + z * (-0.5f + z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f)));
  *sn = hi ? c : s; *cs = hi ? s : c;
}
// sin / cos of a joint angle (kinematics): Cody-Waite reduction by pi/2 in two fused steps (|x| of a few turns: joint coordinates), the same
// [-pi/4, pi/4] polynomials as above, quadrant by swap and sign -- about 25 instructions against the library's ~60 with its large-argument path
__device__ __forceinline__ void sincos_jf(float x, float* sn, float* cs) {
#ifdef MYO_KIN_LIBM
  sincosf(x, sn, cs); return;
#endif
  if (MYO_EXACT_TRIG) { sincosf(x, sn, cs); return; }
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.37113882867379116e-8f, r);
  const float z = r * r;
  const float s = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const float c = 1.f + z * (-0.5f + z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f)));
  const int q = (int)k;
  const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
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
  float Gang = acos11f(cosG);
  // Newton on theta = asin(z): same root as MuJoCo's iteration in z, but well conditioned in float near z -> 1
  (void)zinit;
  float th = 1.57079632679f - 4.4721360e-4f;
  float sn = 0.9999999f, cs = 4.4721359e-4f;          // sin / cos of the start angle
  float f = asin01f(A * sn) + asin01f(B * sn) - 2 * th + Gang;
  if (f > 0) return 0;
  for (int iter = 0; iter < 20 && fabsf(f) > tolerance; iter++) {
    float df = A * cs / fmaxf(MINVALF, sqrtf(1 - A * A * sn * sn)) + B * cs / fmaxf(MINVALF, sqrtf(1 - B * B * sn * sn)) - 2;
    th = clipf(th - f / df, 1e-6f, 1.57079632679f);
    sincos_q1f(th, &sn, &cs);
    f = asin01f(A * sn) + asin01f(B * sn) - 2 * th + Gang;
  }
  float vec[2], ang;
  if (d[0] * d[3] - d[1] * d[2] > 0) { vec[0] = d[0] / len0; vec[1] = d[1] / len0; ang = th - asin01f(A * sn); }
  else { vec[0] = d[2] / len1; vec[1] = d[3] / len1; ang = th - asin01f(B * sn); }
  float sa, ca;
  sincos_q1f(clipf(ang, 0.f, 1.57079632679f), &sa, &ca);
  pnt[0] = rad * (ca * vec[0] - sa * vec[1]);
  pnt[1] = rad * (sa * vec[0] + ca * vec[1]);
  pnt[2] = pnt[0]; pnt[3] = pnt[1];
  return 0;
}

// returns wrap length (<0: no wrap); wpnt = two world points.  Two copies of one body: the wave kernels inline wrap_geom_inl (an out-of-line
// call passes its array arguments through scratch memory: 3-5 % on every wave kernel; round 1 kept the hand kernel on the out-of-line copy for
// 0.6 %, but once that kernel grew the compiler stopped inlining it on its own and the call cost 2.7 %); the 16 / 32-lane cross-check kernel
// keeps the out-of-line wrap_geom
#define MYO_WRAP_GEOM_BODY  \
  float p[6], s[3] = {0, 0, 0}, tmp[3], axis[6], d[4], sd[2] = {0, 0}, pnt[4], res[6];  \
  tmp[0] = x0[0] - xpos[0]; tmp[1] = x0[1] - xpos[1]; tmp[2] = x0[2] - xpos[2];  \
  matTvec(p, xmat, tmp);  \
  tmp[0] = x1[0] - xpos[0]; tmp[1] = x1[1] - xpos[1]; tmp[2] = x1[2] - xpos[2];  \
  matTvec(p + 3, xmat, tmp);  \
  if (norm3(p) < MINVALF || norm3(p + 3) < MINVALF) return -1;  \
  if (has_side) {  \
    tmp[0] = side[0] - xpos[0]; tmp[1] = side[1] - xpos[1]; tmp[2] = side[2] - xpos[2];  \
    matTvec(s, xmat, tmp);  \
  }  \
  if (!cylinder) {  \
    axis[0] = p[0]; axis[1] = p[1]; axis[2] = p[2];  \
    normalize3(axis);  \
    float nrmv[3];  \
    cross3(nrmv, p, p + 3);  \
    float nrm = norm3(nrmv);  \
    if (nrm < MINVALF) {  \
      int i = 0;  \
      if (fabsf(axis[1]) > fabsf(axis[0]) && fabsf(axis[1]) > fabsf(axis[2])) i = 1;  \
      if (fabsf(axis[2]) > fabsf(axis[0]) && fabsf(axis[2]) > fabsf(axis[1])) i = 2;  \
      float t[3] = {i == 0 ? 0.f : 1.f, i == 1 ? 0.f : 1.f, i == 2 ? 0.f : 1.f};  \
      cross3(nrmv, axis, t);  \
      nrm = norm3(nrmv);  \
    }  \
    float inv = 1.0f / nrm;  \
    nrmv[0] *= inv; nrmv[1] *= inv; nrmv[2] *= inv;  \
    cross3(axis + 3, nrmv, axis);  /* unit already: cross product of two orthonormal vectors */  \
    d[0] = dot3(p, axis); d[1] = dot3(p, axis + 3); d[2] = dot3(p + 3, axis); d[3] = dot3(p + 3, axis + 3);  \
    if (has_side) { sd[0] = dot3(s, axis); sd[1] = dot3(s, axis + 3); }  \
  } else {  \
    d[0] = p[0]; d[1] = p[1]; d[2] = p[3]; d[3] = p[4];  \
    if (has_side) { sd[0] = s[0]; sd[1] = s[1]; }  \
  }  \
  float wlen;  \
  float sdn = sqrtf(sd[0] * sd[0] + sd[1] * sd[1]);  \
  if (has_side && sdn < radius) {  \
    wlen = wrap_inside(pnt, d, radius);  \
  } else {  \
    if (has_side && sdn > MINVALF) { sd[0] /= sdn; sd[1] /= sdn; }  \
    wlen = wrap_circle(pnt, d, sd, has_side, radius);  \
  }  \
  if (wlen < 0) return -1;  \
  if (!cylinder) {  \
_Pragma("unroll")  \
    for (int k = 0; k < 3; k++) {  \
      res[k] = axis[k] * pnt[0] + axis[3 + k] * pnt[1];  \
      res[3 + k] = axis[k] * pnt[2] + axis[3 + k] * pnt[3];  \
    }  \
  } else {  \
    float L0 = sqrtf((p[0] - pnt[0]) * (p[0] - pnt[0]) + (p[1] - pnt[1]) * (p[1] - pnt[1]));  \
    float L1 = sqrtf((p[3] - pnt[2]) * (p[3] - pnt[2]) + (p[4] - pnt[3]) * (p[4] - pnt[3]));  \
    float tot = L0 + wlen + L1;  \
    res[0] = pnt[0]; res[1] = pnt[1]; res[3] = pnt[2]; res[4] = pnt[3];  \
    res[2] = p[2] + (p[5] - p[2]) * L0 / tot;  \
    res[5] = p[2] + (p[5] - p[2]) * (L0 + wlen) / tot;  \
    float h = res[5] - res[2];  \
    wlen = sqrtf(wlen * wlen + h * h);  \
  }  \
  matvec(wpnt, xmat, res);  \
  matvec(wpnt + 3, xmat, res + 3);  \
_Pragma("unroll")  \
  for (int k = 0; k < 3; k++) { wpnt[k] += xpos[k]; wpnt[3 + k] += xpos[k]; }  \
  return wlen;

__device__ float wrap_geom(float* wpnt, const float* x0, const float* x1, const float* xpos, const float* xmat, float radius,
                           bool cylinder, const float* side, bool has_side) {
  MYO_WRAP_GEOM_BODY
}
__device__ __forceinline__ float wrap_geom_inl(float* wpnt, const float* x0, const float* x1, const float* xpos, const float* xmat, float radius,
                                               bool cylinder, const float* side, bool has_side) {
  MYO_WRAP_GEOM_BODY
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
  if (actmap == MYO_ACTMAP_CTRLRANGE) {   // TrackEnv.step (mjx/myodm_v0.py:272-275): [-1, 1] onto actuator_ctrlrange, every actuator alike
    const DevTrack* K = Bt.track;
    const float t = (c + 1.0f) * (K->hi[i] - K->lo[i]) * 0.5f;   // (two statements: no fused multiply-add, the reference rounds the product first)
    return t + K->lo[i];
  }
  // stateless (non-muscle) actuators: untouched when the model has muscles (base_v0.py:87-93 only re-projects the muscle entries),
  // re-projected from [-1, 1] onto their ctrlrange when it has none (Robot.process_actuator, robot/robot.py:773-782); the lowering
  // stores the applicable scale / offset in the record
  if (actprm[16 * i + 10] < 0.f) return actprm[16 * i + 6] + c * actprm[16 * i + 5];
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
  else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);   // MuJoCo's default power: no powf (~100 instructions each)
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
// Every convex shape of the config models is "scaled sphere (+) axial segment": support(d) = S^2 d / |S d| + h sign(d_z) e_z with
//   ellipsoid S = semi-axes, h = 0 | sphere S = (r,r,r), h = 0 | capsule S = (r,r,r), h = half length | cylinder S = (r,r,0), h = half length.
// One branch-free formula instead of a per-type switch: in a wave that mixes pad / capsule pairs every lane used to walk through all
// the type branches of both shapes at each of the ~23 support evaluations of an MPR call.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float4 __attribute__((address_space(1)))* gpf4;
#else
typedef const float4* gpf4;
#endif
// by value: registers.  Polytope kernels only: verts (small hulls are scanned), rec / srec (vertex graph as self-contained records, lowering.py
// hip_mesh_rec / hip_mesh_startrec) and the warm start of the climb: hv = the support vertex of this object's previous query [x, y, z, word]
// (word < 0: none yet).  Successive MPR directions are close, so the next climb usually ends after one look at the neighbours.
struct CObj { float pos[3], mat[9], S[3], h, margin; int vadr, sadr; mutable float hv[4]; };
// the mesh tables themselves are wave-uniform and travel beside the objects (scalar registers): a CObj only keeps where its own vertices
// (vadr, in vertices) and its start-direction table (sadr, in records) begin -- two ints instead of three 64-bit pointers per object
struct MeshTab { gpf vert = nullptr; gpf4 rec = nullptr, srec = nullptr; };
__device__ __forceinline__ void cobj_shape(CObj& o, int type, const float* size) {
  if (type == GEOM_ELLIPSOID) { o.S[0] = size[0]; o.S[1] = size[1]; o.S[2] = size[2]; o.h = 0.f; }
  else if (type == GEOM_CYLINDER) { o.S[0] = size[0]; o.S[1] = size[0]; o.S[2] = 0.f; o.h = size[1]; }
  else { o.S[0] = o.S[1] = o.S[2] = size[0]; o.h = type == GEOM_CAPSULE ? size[1] : 0.f; }
}
// support point of the un-inflated shape in its own frame, for a direction given in that frame
__device__ __forceinline__ void support_local(const float* S, float h, const float* dl, float* pl) {
  float s[3] = {S[0] * dl[0], S[1] * dl[1], S[2] * dl[2]};
  // v_rsq_f32 on the squared norm.  (Written as 1.0f / norm3(s), the optimiser turned the three products with the reciprocal into three divisions of
  // its own making, which no longer carried the fast-division marking of the source: three full IEEE expansions, ~30 instructions, per support call --
  // sixty per MPR step.  The intrinsic cannot be folded.)
  const float n2 = dot3(s, s);
  const float inv = n2 > MINVALF * MINVALF ? __builtin_amdgcn_rsqf(n2) : 0.f;
  pl[0] = S[0] * s[0] * inv; pl[1] = S[1] * s[1] * inv; pl[2] = S[2] * s[2] * inv + (dl[2] >= 0 ? h : -h);
}
// polytope shapes (TrackEnv kernels, MPR mode 2): h = -2: box with half sizes S; h = -3: convex hull, S[0] vertices at `verts` (support = best vertex)
// (hull: vertex list + vertex graph, lowering.py hip_mesh_*: nadr[v] .. nadr[v + 1] index the mesh-local neighbour numbers of vertex v in nbr,
// start = six axis-extreme vertices)
__device__ __forceinline__ void cobj_shape_poly(CObj& o, int type, const float* size) {
  o.vadr = 0; o.sadr = 0; o.hv[0] = o.hv[1] = o.hv[2] = 0.f; o.hv[3] = -1.f;
  if (type == 6) { o.S[0] = size[0]; o.S[1] = size[1]; o.S[2] = size[2]; o.h = -2.f; }
  else if (type == 7) {
    const int adr = (int)size[0];
    o.S[0] = size[1]; o.S[1] = o.S[2] = 0.f; o.h = -3.f;
    o.vadr = adr; o.sadr = 96 * (int)size[2];
  } else cobj_shape(o, type, size);
}
template <int MODE> __device__ __forceinline__ void support_shape(const CObj& o, const float* dl, float* pl, const MeshTab& T = MeshTab{}) {
  if (MODE == 2 && o.h == -2.f) { pl[0] = dl[0] >= 0.f ? o.S[0] : -o.S[0]; pl[1] = dl[1] >= 0.f ? o.S[1] : -o.S[1]; pl[2] = dl[2] >= 0.f ? o.S[2] : -o.S[2]; return; }
  if (MODE == 2 && o.h == -3.f) {
    const int n = (int)o.S[0];
    float bd = -1e30f, bx = 0.f, by = 0.f, bz = 0.f;
#ifndef MYO_HULL_SCAN_MAX
#define MYO_HULL_SCAN_MAX 0      // (a scan of a 10 .. 24-vertex hull is a serial chain of dependent-latency loads: 364 k -> 263 k cycles per substep in the
                                  //  narrow phase of contact-rich TrackEnv states when every hull climbs its vertex graph instead)
#endif
    if (n <= MYO_HULL_SCAN_MAX) {          // small hull: scan
      for (int i = 0; i < n; i++) {
        const float x = T.vert[3 * (o.vadr + i)], y = T.vert[3 * (o.vadr + i) + 1], z = T.vert[3 * (o.vadr + i) + 2], t = x * dl[0] + y * dl[1] + z * dl[2];
        if (t > bd) { bd = t; bx = x; by = y; bz = z; }
      }
    } else {                // climb the hull's vertex graph: a vertex no neighbour beats is the support vertex (convexity)
      float bw = o.hv[3];
      if (bw >= 0.f) { bx = o.hv[0]; by = o.hv[1]; bz = o.hv[2]; bd = bx * dl[0] + by * dl[1] + bz * dl[2]; }
      else {   // first query of this object: start from the direction table (6 cube faces x 4 x 4 cells, lowering.py _cube_dirs)
        const float a0 = fabsf(dl[0]), a1 = fabsf(dl[1]), a2 = fabsf(dl[2]);
        const int axis = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
        const float dm = axis == 0 ? dl[0] : (axis == 1 ? dl[1] : dl[2]), du = axis == 0 ? dl[1] : (axis == 1 ? dl[2] : dl[0]), dv = axis == 0 ? dl[2] : (axis == 1 ? dl[0] : dl[1]);
        const float inv = 2.0f / fmaxf(fabsf(dm), MINVALF);
        const int iu = min(3, max(0, (int)(du * inv + 2.0f))), iv = min(3, max(0, (int)(dv * inv + 2.0f)));
        const float4 r = T.srec[o.sadr + ((2 * axis + (dm < 0.f ? 1 : 0)) * 4 + iu) * 4 + iv];
        bd = r.x * dl[0] + r.y * dl[1] + r.z * dl[2]; bx = r.x; by = r.y; bz = r.z; bw = r.w;
      }
      for (int it = 0; it < 128; it++) {
        const int word = (int)bw, e0 = word >> 8, deg = word & 255;
        bool moved = false;
#ifndef MYO_HULL_BURST
#define MYO_HULL_BURST 8
#endif
        for (int e = 0; e < deg; e += MYO_HULL_BURST) {       // lists are padded to a multiple of eight records: eight independent 16-byte loads in flight at a time
          float4 r[MYO_HULL_BURST];
#pragma unroll
          for (int k = 0; k < MYO_HULL_BURST; k++) r[k] = T.rec[e0 + e + k];
#pragma unroll
          for (int k = 0; k < MYO_HULL_BURST; k++) {
            const float t = r[k].x * dl[0] + r[k].y * dl[1] + r[k].z * dl[2];
            if (t > bd) { bd = t; bx = r[k].x; by = r[k].y; bz = r[k].z; bw = r[k].w; moved = true; }
          }
        }
        if (!moved) break;
      }
      o.hv[0] = bx; o.hv[1] = by; o.hv[2] = bz; o.hv[3] = bw;
    }
    pl[0] = bx; pl[1] = by; pl[2] = bz;
    return;
  }
  support_local(o.S, o.h, dl, pl);
}
struct Sup { float v[3], v1[3]; };  // Minkowski point and its witness on obj1 (the witness on obj2 is v1 - v)
// Minkowski-difference support of the two margin-inflated shapes.  Contract of the wave kernel's caller: obj `a` sits in the
// identity frame at the origin (the pair is expressed in geom 1's frame) and `dir` is a unit vector, so a's support needs no
// rotation and the spherical inflation is just +-margin * dir (no norm, no division).
// HF (height-field kernels only): obj `a` may instead be a triangular prism of the height field (mjc_ConvexHField's prism_support), marked by
// a.h < 0 and stored in a's otherwise unused frame slots: a.mat = x[3] | y[3] | z_top[3] of the three columns, a.S[0] = z of the base;
// only the bottom or the top triangle can be extremal, by the sign of dir_z.
// HF is an int mode: 0 smooth primitives, 1 (true) height-field prisms in `a`, 2 polytopes (box / convex hull) in `a` or `b`
template <int HF = 0>
__device__ void mink_support(const CObj& a, const CObj& b, const float* dir, Sup& s, const MeshTab& T = MeshTab{}) {
  float nd[3] = {-dir[0], -dir[1], -dir[2]}, dl[3], pl[3], w2[3];
  if (HF == 1 && a.h < 0.f) {
    const bool top = dir[2] >= 0.f;
    float d0 = dir[0] * a.mat[0] + dir[1] * a.mat[3], d1 = dir[0] * a.mat[1] + dir[1] * a.mat[4], d2 = dir[0] * a.mat[2] + dir[1] * a.mat[5];
    if (top) { d0 += dir[2] * a.mat[6]; d1 += dir[2] * a.mat[7]; d2 += dir[2] * a.mat[8]; }
    int best = d1 > d0 ? 1 : 0;
    if (d2 > fmaxf(d0, d1)) best = 2;
    // blend by 0/1 weights rather than selecting between struct fields: a select of two loads becomes a load from a selected address,
    // and one dynamic address is enough to push the whole CObj into scratch memory
    const float w0 = best == 0 ? 1.f : 0.f, w1 = best == 1 ? 1.f : 0.f, w2 = best == 2 ? 1.f : 0.f;
    s.v1[0] = w0 * a.mat[0] + w1 * a.mat[1] + w2 * a.mat[2];
    s.v1[1] = w0 * a.mat[3] + w1 * a.mat[4] + w2 * a.mat[5];
    const float zt = w0 * a.mat[6] + w1 * a.mat[7] + w2 * a.mat[8];
    s.v1[2] = top ? zt : a.S[0];
  } else
  support_shape<HF>(a, dir, s.v1, T);
  matTvec(dl, b.mat, nd);
  support_shape<HF>(b, dl, pl, T);
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
// nwarm (optional): the contact normal this pair had in the previous substep of the same env step, in geom 1's frame.  Three support
// points a small angle around it form a portal that is already ~R eps^2 / 2 flat; if the origin ray passes through it, portal
// discovery and the first ~10 halving steps are skipped.  Otherwise (pair moved too much, new contact) the cold path runs.
#ifndef MPR_WARM_EPS
#define MPR_WARM_EPS 0.05f
#endif
template <int HF>
__device__ __forceinline__ bool mpr_penetration_t(const CObj& o1, const CObj& o2, float tol, int maxit, float* depth, float* dirout, float* posout, int* nsup = nullptr,
                                  const float* nwarm = nullptr, const MeshTab& T = MeshTab{}) {
  Sup p[4];
  float dir[3], va[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { p[0].v1[k] = o1.pos[k]; p[0].v[k] = o1.pos[k] - o2.pos[k]; }
  if (norm3(p[0].v) < MINVALF) p[0].v[0] += 1e-5f;
  bool warm_ok = false;
  if (nwarm) {
    float t1[3], t2[3];
    make_frame(nwarm, t1, t2);
    bool beyond = true;
#pragma unroll
    for (int k = 1; k <= 3; k++) {
      const float c = k == 1 ? 1.f : -0.5f, sn = k == 1 ? 0.f : (k == 2 ? 0.8660254f : -0.8660254f);
#pragma unroll
      for (int i = 0; i < 3; i++) dir[i] = nwarm[i] + MPR_WARM_EPS * (c * t1[i] + sn * t2[i]);
      normalize3(dir);
      mink_support<HF>(o1, o2, dir, p[k], T);
      beyond = beyond && dot3(p[k].v, dir) >= 0;
    }
    float s12, s23, s31;
    cross3(va, p[1].v, p[2].v); s12 = dot3(va, p[0].v);
    cross3(va, p[2].v, p[3].v); s23 = dot3(va, p[0].v);
    cross3(va, p[3].v, p[1].v); s31 = dot3(va, p[0].v);
    if (beyond && s12 >= 0 && s23 >= 0 && s31 >= 0) { Sup t = p[2]; p[2] = p[3]; p[3] = t; warm_ok = true; }   // cold-path winding: all three <= 0
    else warm_ok = beyond && s12 <= 0 && s23 <= 0 && s31 <= 0;
  }
  if (!warm_ok) {
  dir[0] = -p[0].v[0]; dir[1] = -p[0].v[1]; dir[2] = -p[0].v[2];
  normalize3(dir);
  mink_support<HF>(o1, o2, dir, p[1], T);
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
  mink_support<HF>(o1, o2, dir, p[2], T);
  if (dot3(p[2].v, dir) < 0) return false;
#pragma unroll
  for (int k = 0; k < 3; k++) { va[k] = p[1].v[k] - p[0].v[k]; vb[k] = p[2].v[k] - p[0].v[k]; }
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, p[0].v) > 0) { Sup t = p[1]; p[1] = p[2]; p[2] = t; dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    mink_support<HF>(o1, o2, dir, p[3], T);
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
  }  // cold portal discovery
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    portal_dir(p, dir);
    if (dot3(dir, p[1].v) >= 0) break;
    Sup v4;
    mink_support<HF>(o1, o2, dir, v4, T);
    float dv4 = dot3(v4.v, dir);
    float dmin = fminf(fminf(dv4 - dot3(p[1].v, dir), dv4 - dot3(p[2].v, dir)), dv4 - dot3(p[3].v, dir));
    if (dv4 < 0 || dmin <= tol) return false;
    expand_portal(p, v4);
  }
  Sup v4;
  for (int it = 0;; it++) {
    portal_dir(p, dir);
    mink_support<HF>(o1, o2, dir, v4, T);
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
// the plain (non height-field) instantiation behind an ordinary function, as before the height-field variant existed: the caller's register
// allocation around this call is sensitive to how it is inlined (67 vs 93 spilled VGPRs in the MyoHand kernel)
__device__ bool mpr_penetration(const CObj& o1, const CObj& o2, float tol, int maxit, float* depth, float* dirout, float* posout, int* nsup = nullptr,
                                const float* nwarm = nullptr) {
  return mpr_penetration_t<0>(o1, o2, tol, maxit, depth, dirout, posout, nsup, nwarm);
}

// Same algorithm with the portal's obj-1 witness points and the interior point kept in LDS (wl: 12 floats of per-lane scratch) instead of registers: they are only
// needed once, for the contact position at the very end, and 9 VGPRs fewer across the refinement loop is what the MyoHand kernel's
// register allocation needs (its only spills sit around this call).  Obj 1 sits at the origin of its own frame (o1.pos = 0).
__device__ __forceinline__ void wl_set(float* wl, int k, const float* w) { wl[3 * (k - 1)] = w[0]; wl[3 * (k - 1) + 1] = w[1]; wl[3 * (k - 1) + 2] = w[2]; }
__device__ __forceinline__ void wl_copy(float* wl, int dst, int src) { wl[3 * (dst - 1)] = wl[3 * (src - 1)]; wl[3 * (dst - 1) + 1] = wl[3 * (src - 1) + 1]; wl[3 * (dst - 1) + 2] = wl[3 * (src - 1) + 2]; }
__device__ __forceinline__ void wl_swap(float* wl, int a, int b) {
#pragma unroll
  for (int k = 0; k < 3; k++) { const float t = wl[3 * (a - 1) + k]; wl[3 * (a - 1) + k] = wl[3 * (b - 1) + k]; wl[3 * (b - 1) + k] = t; }
}
// (portal vertices as four separate 3-vectors and component-wise swaps: an array of structs copied with struct assignments stays an
// addressable stack object -- memcpy between allocas -- and then lives in scratch memory instead of registers)
#define V3SWAP(a, b) { float t0_ = a[0], t1_ = a[1], t2_ = a[2]; a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; b[0] = t0_; b[1] = t1_; b[2] = t2_; }
#define V3COPY(a, b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
template <int HF = 0>
__device__ __forceinline__ bool mpr_penetration_wl(const CObj& o1, const CObj& o2, float tol, int maxit, float* depth, float* dirout, float* posout, int* nsup,
                                                   const float* nwarm, float* wl, const MeshTab& T = MeshTab{}) {
  float p1[3], p2[3], p3[3];   // (p0, the interior point, is read in a handful of places only: it lives in wl[9..11])
  Sup s;
  float dir[3], va[3], vb[3];
  {
    float q[3] = {-o2.pos[0], -o2.pos[1], -o2.pos[2]};
    if (norm3(q) < MINVALF) q[0] += 1e-5f;
    wl[9] = q[0]; wl[10] = q[1]; wl[11] = q[2];
  }
#define P0LOAD() const float p0[3] = {wl[9], wl[10], wl[11]}
  bool warm_ok = false;
  if (nwarm) {
    float t1[3], t2[3];
    make_frame(nwarm, t1, t2);
    bool beyond = true;
#pragma unroll
    for (int i = 0; i < 3; i++) dir[i] = nwarm[i] + MPR_WARM_EPS * t1[i];
    normalize3(dir);
    mink_support<HF>(o1, o2, dir, s, T);
    V3COPY(p1, s.v); wl_set(wl, 1, s.v1);
    beyond = beyond && dot3(s.v, dir) >= 0;
#pragma unroll
    for (int i = 0; i < 3; i++) dir[i] = nwarm[i] + MPR_WARM_EPS * (-0.5f * t1[i] + 0.8660254f * t2[i]);
    normalize3(dir);
    mink_support<HF>(o1, o2, dir, s, T);
    V3COPY(p2, s.v); wl_set(wl, 2, s.v1);
    beyond = beyond && dot3(s.v, dir) >= 0;
#pragma unroll
    for (int i = 0; i < 3; i++) dir[i] = nwarm[i] + MPR_WARM_EPS * (-0.5f * t1[i] - 0.8660254f * t2[i]);
    normalize3(dir);
    mink_support<HF>(o1, o2, dir, s, T);
    V3COPY(p3, s.v); wl_set(wl, 3, s.v1);
    beyond = beyond && dot3(s.v, dir) >= 0;
    float s12, s23, s31;
    P0LOAD();
    cross3(va, p1, p2); s12 = dot3(va, p0);
    cross3(va, p2, p3); s23 = dot3(va, p0);
    cross3(va, p3, p1); s31 = dot3(va, p0);
    if (beyond && s12 >= 0 && s23 >= 0 && s31 >= 0) { V3SWAP(p2, p3); wl_swap(wl, 2, 3); warm_ok = true; }
    else warm_ok = beyond && s12 <= 0 && s23 <= 0 && s31 <= 0;
  }
  if (!warm_ok) {
    P0LOAD();
    dir[0] = -p0[0]; dir[1] = -p0[1]; dir[2] = -p0[2];
    normalize3(dir);
    mink_support<HF>(o1, o2, dir, s, T);
    V3COPY(p1, s.v); wl_set(wl, 1, s.v1);
    if (dot3(p1, dir) < 0) return false;
    cross3(dir, p0, p1);
    if (norm3(dir) < 1e-12f) {
      *depth = norm3(p1);
#pragma unroll
      for (int k = 0; k < 3; k++) { dirout[k] = p1[k]; posout[k] = s.v1[k] - 0.5f * p1[k]; }
      normalize3(dirout);
      return true;
    }
    normalize3(dir);
    mink_support<HF>(o1, o2, dir, s, T);
    V3COPY(p2, s.v); wl_set(wl, 2, s.v1);
    if (dot3(p2, dir) < 0) return false;
#pragma unroll
    for (int k = 0; k < 3; k++) { va[k] = p1[k] - p0[k]; vb[k] = p2[k] - p0[k]; }
    cross3(dir, va, vb);
    normalize3(dir);
    if (dot3(dir, p0) > 0) { V3SWAP(p1, p2); wl_swap(wl, 1, 2); dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2]; }
    for (int it = 0;; it++) {
      if (it > maxit) return false;
      mink_support<HF>(o1, o2, dir, s, T);
      V3COPY(p3, s.v); wl_set(wl, 3, s.v1);
      if (dot3(p3, dir) < 0) return false;
      bool cont = false;
      cross3(va, p1, p3);
      if (dot3(va, p0) < -MINVALF) { V3COPY(p2, p3); wl_copy(wl, 2, 3); cont = true; }
      if (!cont) {
        cross3(va, p3, p2);
        if (dot3(va, p0) < -MINVALF) { V3COPY(p1, p3); wl_copy(wl, 1, 3); cont = true; }
      }
      if (!cont) break;
#pragma unroll
      for (int k = 0; k < 3; k++) { va[k] = p1[k] - p0[k]; vb[k] = p2[k] - p0[k]; }
      cross3(dir, va, vb);
      normalize3(dir);
    }
  }
#define PORTAL_DIR() { float a_[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, b_[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]}; cross3(dir, a_, b_); \
    const float n2_ = dot3(dir, dir); if (n2_ < MINVALF * MINVALF) { dir[0] = 1.f; dir[1] = 0.f; dir[2] = 0.f; } else { const float i_ = __builtin_amdgcn_rsqf(n2_); dir[0] *= i_; dir[1] *= i_; dir[2] *= i_; } }
#define EXPAND_PORTAL() { \
    P0LOAD(); float va_[3]; cross3(va_, s.v, p0); \
    const int idx_ = dot3(p1, va_) > 0 ? (dot3(p2, va_) > 0 ? 1 : 3) : (dot3(p3, va_) > 0 ? 2 : 1); \
    _Pragma("unroll") for (int k = 0; k < 3; k++) { p1[k] = idx_ == 1 ? s.v[k] : p1[k]; p2[k] = idx_ == 2 ? s.v[k] : p2[k]; p3[k] = idx_ == 3 ? s.v[k] : p3[k]; } \
    wl_set(wl, idx_, s.v1); }
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    PORTAL_DIR();
    if (dot3(dir, p1) >= 0) break;
    mink_support<HF>(o1, o2, dir, s, T);
    float dv4 = dot3(s.v, dir);
    float dmin = dv4 - dot3(p1, dir);   // dir is the portal's normal: p1, p2, p3 have the same component along it (the oracle takes the min of the three: equal up to round-off)
    if (dv4 < 0 || dmin <= tol) return false;
    EXPAND_PORTAL();
  }
  for (int it = 0;; it++) {
    PORTAL_DIR();
    mink_support<HF>(o1, o2, dir, s, T);
    float dv4 = dot3(s.v, dir);
    float dmin = dv4 - dot3(p1, dir);   // dir is the portal's normal: p1, p2, p3 have the same component along it (the oracle takes the min of the three: equal up to round-off)
    if (dmin <= tol || it > maxit) { if (nsup) *nsup = it; break; }
    EXPAND_PORTAL();
  }
#undef PORTAL_DIR
#undef EXPAND_PORTAL
  *depth = dot3(s.v, dir);
  P0LOAD();
  float bw[4], cr[3];
  cross3(cr, p1, p2); bw[0] = dot3(cr, p3);
  cross3(cr, p3, p2); bw[1] = dot3(cr, p0);
  cross3(cr, p0, p1); bw[2] = dot3(cr, p3);
  cross3(cr, p2, p1); bw[3] = dot3(cr, p0);
  float sum = bw[0] + bw[1] + bw[2] + bw[3];
  if (sum <= 0) {
    bw[0] = 0;
    cross3(cr, p2, p3); bw[1] = dot3(cr, dir);
    cross3(cr, p3, p1); bw[2] = dot3(cr, dir);
    cross3(cr, p1, p2); bw[3] = dot3(cr, dir);
    sum = bw[1] + bw[2] + bw[3];
  }
  float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    dirout[k] = dir[k];
    posout[k] = inv * (bw[0] * (0.f - 0.5f * p0[k]) + bw[1] * (wl[k] - 0.5f * p1[k]) +
                       bw[2] * (wl[3 + k] - 0.5f * p2[k]) + bw[3] * (wl[6 + k] - 0.5f * p3[k]));
  }
  return true;
}
#undef P0LOAD
#undef V3SWAP
#undef V3COPY

#endif  // MYO_PHYSICS_H
