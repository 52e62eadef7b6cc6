// myo_kernel_lanes.h -- first kernel: G = 16 / 32 / 64 lanes per env, everything in LDS (cross-check and fallback; myo_set_lanes).
// Part of the single translation unit myo_hip.hip (included there, in this order); not a stand-alone header.
#ifndef MYO_KERNEL_LANES_H
#define MYO_KERNEL_LANES_H

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
        for (int k = 0; k < 3; k++) o1.pos[k] = 0.f;
        cobj_shape(o1, M.cg_type[g1], sz1); cobj_shape(o2, M.cg_type[g2], sz2);
        o1.margin = o2.margin = 0.5f * margin;
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

#endif  // MYO_KERNEL_LANES_H
