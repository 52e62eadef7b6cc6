// myo_kernel_wave.h -- wave-per-env kernel step_kernel_w (default path), its LDS layout, substep scheduler, size specialisations.
// Part of the single translation unit myo_hip.hip (included there, in this order); not a stand-alone header.
#ifndef MYO_KERNEL_WAVE_H
#define MYO_KERNEL_WAVE_H

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
#define MPRW 14    // entries of the MPR warm-start table (pair id + normal): 224 B, the hand slice stays within 10 240 B = 16 waves per CU
static_assert(4 * MPRW <= 63, "the scheduler carries the table in a 64-word row (word 63 = entry count)");
struct LayW {
  int qpos, qvel, act, ctrl, lpos, lmat, axis, anchor, xv, qfc, sq, mprw, X;
  int tJ, tlen, tforce;                           // region X, tendon phase
  int cdof, cinert, crb, cvel, cacc, cfrc;        // region X, dynamics phase
  int gpos, gax, cand, cdist, cpos, cnrm, cpair, cJ, cdofs;  // region X, collision + solver phase
  int Mp;                                                     // packed mass matrix, aliases gpos/gax/cand once the contact rows exist
  int tJp;                                                    // persistent sparse tendon rows (only for models with tendon limits)
  int total;
};
// The one definition of the wave kernel's LDS layout: myo_model_load runs it on the model's sizes, the size-specialised instantiations
// evaluate it at compile time on Sizes<SPEC> (every LDS address is then an immediate: round 2 read ~250 layout words per substep
// through scalar loads, each behind an s_waitcnt that also drains the LDS queue).
__host__ __device__ constexpr LayW layout_w(int nq, int nv, int nu, int nl, int ngt, int maxnnz, int ncg, bool has_tl, int nvt, int kc, int nc, int nj) {
  LayW Y{};
  int o = 0;
  Y.qpos = o; o += nq; Y.qvel = o; o += nv; Y.act = o; o += nu; Y.ctrl = o; o += nu;
  Y.lpos = o; o += 3 * nl; Y.lmat = o; o += 9 * nl; Y.axis = o; o += 3 * nv; Y.anchor = o; o += 3 * nv;
  Y.xv = o; o += nvt; Y.qfc = o; o += nvt; Y.sq = o; o += nvt * (nvt + 1); Y.mprw = o; o += 4 * MPRW;
  Y.tJp = 0;
  if (has_tl) { Y.tJp = o; o += ngt * maxnnz + 2 * ngt; }   // + tendon lengths and velocities, read again by the tendon-limit rows
  Y.X = o;
  Y.tJ = o; o += ngt * maxnnz; Y.tlen = o; o += ngt; Y.tforce = o; o += nu;
  const int endT = o;
  o = Y.X;
  Y.cdof = o; o += 6 * nv; Y.cinert = o; o += 10 * nl; Y.crb = o; o += 10 * nl; Y.cvel = o; o += 6 * nl; Y.cacc = o; o += 6 * nl; Y.cfrc = o; o += 6 * nl;
  const int endD = o;
  o = Y.X;
  Y.Mp = o;
  Y.gpos = o; o += 3 * ncg; Y.gax = o; o += 3 * ncg;
  Y.cand = o; o += NCAND;
  if (o - Y.Mp < (nvt * (nvt + 1)) / 2) o = Y.Mp + (nvt * (nvt + 1)) / 2;
  Y.cdist = o; o += nc; Y.cpos = o; o += 3 * nc; Y.cnrm = o; o += 3 * nc; Y.cpair = o; o += nc;
  Y.cJ = o; o += nc * nj * kc; Y.cdofs = o; o += nc * ((kc + 3) / 4);   // nj jacobian rows of kc entries per contact; kc dof ids per contact, one byte each
  if (o < endT) o = endT;
  if (o < endD) o = endD;
  Y.total = o;
  return Y;
}
// colliding height field (terrain models): world-fixed, axis-aligned; elevation data lives per env in DevBatch.hfield
struct HfDev { int on, nrow, ncol, cg; float size[4], pos[3]; };
struct DevModelW {
  LayW lay;
  gpi seg_order, seg_tendon, gt_dl;
  gpf link_mat0;
  int nwrapseg, ndl, has_tl;
  gpf tl;
  int nq, has_free, neq;          // free-floating root (nq = nv + 1), joint-coupling equalities
  int has_j0;                     // some actuator drives a joint directly: constant moment arms gt_j0 [ngt][maxnnz]
  gpf gt_j0;
  HfDev hf;                       // (kept last: the field offsets of the tables above feed the hot loops' scalar loads)
  gpi link_free, dof_qposadr, eq_i, link_chain_adr, link_chain;
  int kin_dnmax;                    // longest joint chain of a non-free link (uniform trip count of the phase-1 loop)
  gpi kin_base, kin_adr, kin_vec;   // two-phase kinematics (lowering.py hip_kin_*): scratch base per link, per-level lists of (link, vector) entries
  gpf eq_f;
  gpf mesh_rec, mesh_startrec, mesh_aabb;   // hull vertex graphs as float4 records (lowering.py hip_mesh_rec / hip_mesh_startrec); [nmesh][6] vertex bounding boxes
  gpf fl, mesh_vert;    // TRK models: friction-loss rows [nv][4] = loss, D, B, -; hull vertices of the mesh geoms
  // Self-contained per-lane records (built by myo_model_load from the tables above, 16-byte rows): what a lane needs for its item arrives
  // in a few independent 16-byte loads instead of a chain of index -> table -> table reads of single words (round 2: 3.1 k vector and 4.2 k
  // scalar loads per wave and env step, nearly every one with its latency exposed).
  gpf4 seg_rec;         // [nseg, in seg_order][SEGR]: site 0 (link, lpos) | site 1 | wrap geom, side link, 1 / divisor, tendon | three dof-list words,
                        //   wrap type | side lpos, radius | wrap geom link, lpos | its rotation (9)
  gpi dl_pk;            // moment-arm lists, one word per entry: dof | hinge << 7 | row slot << 8 | sign << 16; every list of a segment starts a 16-byte row of its own
  gpf4 cg_rec;          // [ncg][4]: link, lpos | rotation (9) | type, bounding radius
  gpf4 pair_rec;        // [npair][4]: g1 | g2 << 8 | narrow-phase type << 16 | condim << 20 | dofs << 24, margin, gap, dof-list start |
                        //   size 1, bounding radius 1 | size 2, bounding radius 2 | type 1 | type 2 << 8
  gpi pair_dl_pk;       // contact dof lists in one word per entry: dof | hinge << 7 | sign << 8
  // Tree words (built by myo_model_load): the sweeps over the kinematic tree take one packed word per lane and round, loaded ahead of the stage,
  // instead of walking level_adr -> child_adr -> child, link_chain_adr -> link_chain or dof_parent chains of dependent loads
  gpi kin_pk;           // phase 2 of the kinematics: [round][64] scratch offset | link << 11 | kind << 17 | ix << 19 | (parent + 1) << 25; all ones = idle lane;
  int kin_nround;       //   the rounds of a level are contiguous, one padding round closes the table (the loop prefetches a round ahead)
  gpi link_desc;        // [nl][2] links in the subtree of link l, itself included (64-bit mask, low word first)
  gpi link_adof;        // [nl][2] dofs on the path root -> link l, its own included
  gpi dof_anc;          // [nv][2] dof d and its ancestors
  unsigned int free_rot[2], free_j3[2];   // dofs that are rotations of a free joint / the first rotation of one
};
#define SEGR 9
#ifndef MPR_TOL
#define MPR_TOL 1e-8f      // portal refinement stops when the support plane gains less than this (metres)
#endif

__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rdlanei(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// wave-uniform float kept in a scalar register (a VGPR copy of it would be one more value live across every stage)
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
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
// this lane's index in the wave, recomputed where it is needed (two instructions): a copy of threadIdx.x kept for the whole kernel is a
// register that is live across every stage, and was the first thing the allocator spilled
__device__ __forceinline__ int wave_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// dof id k (0..KC-1) of contact c from the byte-packed table (CDW = ints per contact, a constexpr of the kernel)
#define CDOF(E_, Y_, c_, k_) ((int)((((const unsigned int*)((E_) + (Y_).cdofs))[CDW * (c_) + ((k_) >> 2)] >> (8 * ((k_) & 3))) & 255u))
// same from a pointer to the contact's own packed words (LDS row or HBM overflow row)
#define CDOFP(W_, k_) ((int)(((W_)[(k_) >> 2] >> (8 * ((k_) & 3))) & 255u))

// Register factorisation H = L D L^T (unit lower L, D = pivots), lane = row.  in: r[k] = H[lane][k] for k <= lane and ZERO above the diagonal.
// out: r[k] = L[lane][k] for k < lane and zero from the diagonal on; returns 1 / D[lane].  All indices are compile-time.
// Right-looking (outer-product) order: once column j is final, every later column k takes its update r[k] -= L[.][j] (D_j L[k][j]) at once, so the
// NVT - 1 - j updates of a step are independent of each other and the dependent chain of a factorisation is the NVT pivot steps.
// Why L D L^T rather than Cholesky: no square root (v_rcp of the pivot), and both triangular solves run on the SAME unit-diagonal factor with no
// division or scaling per step -- two instructions per forward step, three per backward step (ldl_solve_rows), against seven before.  The zeros from
// the diagonal on are what lets the solves skip every lane test: a lane above the diagonal multiplies by an exact zero.
template <int NVT> __device__ __forceinline__ float chol_rows(float (&r)[NVT], int lane) {
  float invd = 1.0f;
#pragma unroll
  for (int j = 0; j < NVT; j++) {
    const float pj = fmaxf(rdlane(r[j], j), MINVALF);
    const float ip = __builtin_amdgcn_rcpf(pj);   // v_rcp_f32 (1 ulp); pj >= 1e-15, no denormal handling needed
    const float col = lane > j ? r[j] : 0.f;      // D_j L[lane][j]; zero on and above the diagonal, so those lanes take no update below
    const float lt = col * ip;
    r[j] = lt;
    if (lane == j) invd = ip;
#pragma unroll
    for (int k = j + 1; k < NVT; k++) r[k] -= lt * rdlane(col, k);
  }
  return invd;
}
// Dof trees of the size-specialised instantiations (parent dof of each dof; myo_model_load checks the model's dof_parentid against them
// before it selects a specialised instantiation).  The mass matrix M and M + h D couple a dof only with its ancestors and descendants:
// factorised LEAVES FIRST (lane i <-> dof nv - 1 - i) the Cholesky factor keeps exactly that pattern, no fill-in (Featherstone; MuJoCo's
// L^T D L does the same) -- MyoHand 93 of 253 entries below the diagonal, MyoLeg 317 of 561.
template <int SPEC> struct SpecTree { static constexpr int nv = 0; static constexpr int parent[1] = {-1}; };
template <> struct SpecTree<1> { static constexpr int nv = 23;
  static constexpr int parent[23] = {-1, 0, 1, 2, 3, 4, 5, 2, 7, 8, 9, 2, 11, 12, 13, 2, 15, 16, 17, 2, 19, 20, 21}; };
template <> struct SpecTree<2> { static constexpr int nv = 34;
  static constexpr int parent[34] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 8, 17, 18, 5, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 22, 31, 32}; };
template <> struct SpecTree<3> : SpecTree<2> {};
// is dof a an ancestor of dof d?
template <int SPEC> __host__ __device__ constexpr bool tree_anc(int a, int d) {
  int p = SpecTree<SPEC>::parent[d];
  while (p >= 0) { if (p == a) return true; p = SpecTree<SPEC>::parent[p]; }
  return false;
}
// chol_rows on the leaves-first permuted matrix of a tree-structured model: the updates whose factor entry L[K][J] is structurally zero
// (dof of K is not an ancestor of the dof of J) are not emitted (`if constexpr` over index sequences: a run-time predicate inside
// `#pragma unroll` loops blocked the unrolling and put the rows into scratch memory).  Right-looking like chol_rows.
template <int NVT, int SPEC, int J, int K> __device__ __forceinline__ void tree_update(float (&r)[NVT], float lt, float col) {
  constexpr int nv = SpecTree<SPEC>::nv;
  if constexpr (K > J && J < nv && K < nv) {
    if constexpr (tree_anc<SPEC>(nv - 1 - K, nv - 1 - J)) r[K] -= lt * rdlane(col, K);
  }
}
template <int NVT, int SPEC, int J, int... Ks> __device__ __forceinline__ void tree_col(float (&r)[NVT], float lt, float col, std::integer_sequence<int, Ks...>) {
  (tree_update<NVT, SPEC, J, Ks>(r, lt, col), ...);
}
template <int NVT, int SPEC, int J> __device__ __forceinline__ void tree_step(float (&r)[NVT], float& invd, int lane) {
  const float pj = fmaxf(rdlane(r[J], J), MINVALF);
  const float ip = __builtin_amdgcn_rcpf(pj);
  const float col = lane > J ? r[J] : 0.f;
  const float lt = col * ip;
  r[J] = lt;
  if (lane == J) invd = ip;
  tree_col<NVT, SPEC, J>(r, lt, col, std::make_integer_sequence<int, NVT>{});
}
template <int NVT, int SPEC, int... Js> __device__ __forceinline__ void tree_all(float (&r)[NVT], float& invd, int lane, std::integer_sequence<int, Js...>) {
  (tree_step<NVT, SPEC, Js>(r, invd, lane), ...);
}
template <int NVT, int SPEC> __device__ __forceinline__ float chol_rows_tree(float (&r)[NVT], int lane) {
  float invd = 1.0f;
  tree_all<NVT, SPEC>(r, invd, lane, std::make_integer_sequence<int, NVT>{});
  return invd;
}
// x <- (L D L^T)^-1 b ; rows of the unit lower L in registers (zero from the diagonal on), invd = 1 / D[lane], the columns of L^T read from the
// LDS copy T[j * (NVT + 1) + lane] (row j of L: zero for lane >= j)
template <int NVT> __device__ __forceinline__ float chol_solve_rows(const float (&r)[NVT], float invd, float b, const float* T, int lane) {
  float y = b;
#pragma unroll
  for (int j = 0; j < NVT; j++) y = fmaf(-r[j], rdlane(y, j), y);
  y *= invd;
  const float* Tc = T + (lane < NVT ? lane : 0);
#pragma unroll
  for (int j = NVT - 1; j >= 0; j--) y = fmaf(-Tc[j * (NVT + 1)], rdlane(y, j), y);
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

template <class LY> __device__ __forceinline__ void site_world_w(const DevModel& M, const LY& Y, const float* E, int s, float* out) {
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
// world position of a point given in a link's frame (link < 0: world-fixed)
template <class LY> __device__ __forceinline__ void frame_point(const LY& Y, const float* E, int l, const float* lp, float* out) {
  if (l < 0) { out[0] = lp[0]; out[1] = lp[1]; out[2] = lp[2]; return; }
  const float* R = E + Y.lmat + 9 * l;
  const float* P = E + Y.lpos + 3 * l;
  out[0] = P[0] + R[0] * lp[0] + R[1] * lp[1] + R[2] * lp[2];
  out[1] = P[1] + R[3] * lp[0] + R[4] * lp[1] + R[5] * lp[2];
  out[2] = P[2] + R[6] * lp[0] + R[7] * lp[1] + R[8] * lp[2];
}
// world centre / rotation of collision geom g from its record (DevModelW::cg_rec: link, lpos | rotation | type, bounding radius)
template <class LY> __device__ __forceinline__ void geom_world_pos(const DevModelW& W, const LY& Y, const float* E, int g, float* out) {
  const float4 r0 = W.cg_rec[4 * g];
  const float lp[3] = {r0.y, r0.z, r0.w};
  frame_point(Y, E, __float_as_int(r0.x), lp, out);
}
template <class LY> __device__ __forceinline__ void geom_world_mat(const DevModelW& W, const LY& Y, const float* E, int g, float* R) {
  const gpf4 G = W.cg_rec + 4 * g;
  const float4 r0 = G[0], r1 = G[1], r2 = G[2], r3 = G[3];
  const int l = __float_as_int(r0.x);
  const float lm[9] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
  if (l < 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = lm[k];
  } else {
    matmul3(R, E + Y.lmat + 9 * l, lm);
  }
}
// moment-arm entries of one straight tendon piece
// Jt = this tendon's sparse jacobian row in LDS (zeroed before the segment rounds): the entries are accumulated with LDS float atomics
// by the segment lanes themselves (one wave: deterministic order) instead of being gathered entry by entry by the tendon's lane
#if defined(__HIP_DEVICE_COMPILE__)
typedef const int4 __attribute__((address_space(1)))* gpi4;
#else
typedef const int4* gpi4;
#endif
template <class LY> __device__ __forceinline__ float straight_w(const DevModelW& W, const LY& Y, float* E, float* Jt, const float* pa, const float* pb, int adr4, int n,
                                                              float invdiv, bool active, const int4& first) {
  float dif[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  float dist = norm3(dif);
  float inv = dist > MINVALF ? __builtin_amdgcn_rcpf(dist) : 0.f;   // (the intrinsic: `1.0f / dist` times three became three full divisions)
  dif[0] *= inv; dif[1] *= inv; dif[2] *= inv;
  // a lane that does not keep this piece (wrapping segment vs direct piece, or the reverse) runs zero iterations: the wave's trip count is
  // the longest dof list among the lanes that DO keep it, and zero when none does.  Four entries per 16-byte load; the first row was
  // loaded by the caller ahead of the wrap geometry (`first`), so the usual list (<= 4 entries) costs no exposed load at all.
  const int nn = active ? n : 0;
  for (int k0 = 0; k0 < nn; k0 += 4) {
    int4 q = first;
    if (k0) q = ((gpi4)W.dl_pk)[adr4 + (k0 >> 2)];
    const int qe[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u >= nn) continue;
      const int e = qe[u];            // dof | hinge << 7 | row slot << 8 | sign << 16: one word instead of three plus dof_type[dof]
      const int d = e & 127;
      const float* ax = E + Y.axis + 3 * d;
      float col;
      if (e & 128) {
        const float* an = E + Y.anchor + 3 * d;
        float r[3] = {pb[0] - an[0], pb[1] - an[1], pb[2] - an[2]}, c[3];
        cross3(c, ax, r);
        col = dot3(dif, c);
      } else col = dot3(dif, ax);
      atomicAdd(&Jt[(e >> 8) & 255], (float)(e >> 16) * col * invdiv);
    }
  }
  return active ? dist * invdiv : 0.f;
}

// ------------------------------------------------------------------------------------------------
// ---- height field vs convex primitive (mjc_ConvexHField [3P], restated in oracle/myo_oracle.c convex_hfield) -------------------------
// sub-grid of cells under the geom's AABB (rel = geom centre - height field position, ext = AABB half extents); false: cannot touch
__device__ __forceinline__ bool hf_range(const HfDev& H, const float* rel, const float* ext, float rb, float margin, int& r0, int& r1, int& c0, int& c1, float& zmin) {
  if (H.size[0] < rel[0] - rb - margin || -H.size[0] > rel[0] + rb + margin || H.size[1] < rel[1] - rb - margin || -H.size[1] > rel[1] + rb + margin) return false;
  if (H.size[2] < rel[2] - rb - margin || -H.size[3] > rel[2] + rb + margin) return false;
  const float lo[3] = {rel[0] - ext[0], rel[1] - ext[1], rel[2] - ext[2]}, hi[3] = {rel[0] + ext[0], rel[1] + ext[1], rel[2] + ext[2]};
  if (lo[0] - margin > H.size[0] || hi[0] + margin < -H.size[0] || lo[1] - margin > H.size[1] || hi[1] + margin < -H.size[1] ||
      lo[2] - margin > H.size[2] || hi[2] + margin < -H.size[3]) return false;
  c0 = max(0, (int)floorf((lo[0] + H.size[0]) / (2.f * H.size[0]) * (float)(H.ncol - 1)));
  c1 = min(H.ncol - 1, (int)ceilf((hi[0] + H.size[0]) / (2.f * H.size[0]) * (float)(H.ncol - 1)));
  r0 = max(0, (int)floorf((lo[1] + H.size[1]) / (2.f * H.size[1]) * (float)(H.nrow - 1)));
  r1 = min(H.nrow - 1, (int)ceilf((hi[1] + H.size[1]) / (2.f * H.size[1]) * (float)(H.nrow - 1)));
  zmin = lo[2];
  return r1 > r0 && c1 > c0;
}
// the three strip vertices ending at zig-zag index j of cell row r: vertex jj sits at column jj / 2, row r + 1 (jj even) or r (jj odd)
__device__ __forceinline__ void hf_prism(const HfDev& H, const float* data, int r, int j, float* x, float* y, float* z) {
  const float dx = 2.f * H.size[0] / (float)(H.ncol - 1), dy = 2.f * H.size[1] / (float)(H.nrow - 1);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int jj = j - 2 + k, c = jj >> 1, rr = r + ((jj & 1) ? 0 : 1);
    x[k] = dx * (float)c - H.size[0]; y[k] = dy * (float)rr - H.size[1]; z[k] = data[rr * H.ncol + c] * H.size[2];
  }
}
// walks the prisms of the sub-grid in mjc_ConvexHField's order; counts those whose top is not wholly below the geom and, when out != NULL,
// writes their candidate words (pair | row << 10 | zig-zag index << 17) from position `at`
__device__ __forceinline__ int hf_walk(const HfDev& H, const float* data, int r0, int r1, int c0, int c1, float zcut, int p, int* out, int at, int cap) {
  int n = 0;
  for (int r = r0; r < r1; r++)
    for (int j = 2 * c0 + 2; j <= 2 * c1 + 1; j++) {
      float x[3], y[3], z[3];
      hf_prism(H, data, r, j, x, y, z);
      if (z[0] < zcut && z[1] < zcut && z[2] < zcut) continue;
      if (out && at + n < cap) out[at + n] = p | (r << 10) | (j << 17);
      n++;
    }
  return n;
}

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
  int* ring;     // [nqueue][stride]: gen << 24 | substep << 20 | env
  int stride, nsubtot;
  int nqueue;    // queues in use = XCDs of the device (or of its partition: 8 in SPX mode, 4 / 2 / 1 in DPX / QPX / CPX, from the CU count):
                 // a wave serves queue XCC_ID % nqueue, so no queue is left without waves when fewer than 8 XCDs are visible
};
#define SCHED_ENV_MASK 0xFFFFF
__global__ void __launch_bounds__(1024) sched_init_kernel(const int* __restrict__ diag, int B, SchedDev S) {
  __shared__ int hist[256], start[256];
  __shared__ int cmax_s;
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  if (t == 0) cmax_s = 1;
  const int nqu = S.nqueue;
  if (t < 8) { int n = t < nqu ? (B - t + nqu - 1) / nqu : 0; S.ctl[4 * t] = 0; S.ctl[4 * t + 1] = n; S.ctl[4 * t + 2] = n; S.ctl[4 * t + 3] = 0; }
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
    S.ring[(r % nqu) * S.stride + r / nqu] = e;       // generation 0, substep 0
  }
}

// table sizes of the compiled config models (after lowering): SPEC = 1 (MyoHand, myohand_pose.xml) and SPEC = 2 (MyoLeg, myolegs.xml)
// instantiations of the wave kernel take their loop bounds from here; SPEC = 0 reads them from the model at run time
template <int SPEC> struct Sizes { static constexpr int nq = 0, nv = 0, nu = 0, nl = 0, nlevel = 0, maxnnz = 0, nseg = 0, ncg = 0, npair = 0; };
template <> struct Sizes<1> { static constexpr int nq = 23, nv = 23, nu = 39, nl = 17, nlevel = 5, maxnnz = 7, nseg = 116, ncg = 27, npair = 289; };
template <> struct Sizes<2> { static constexpr int nq = 35, nv = 34, nu = 80, nl = 13, nlevel = 6, maxnnz = 11, nseg = 100, ncg = 32, npair = 45; };
template <> struct Sizes<3> { static constexpr int nq = 35, nv = 34, nu = 80, nl = 13, nlevel = 6, maxnnz = 11, nseg = 100, ncg = 33, npair = 76; };   // MyoLeg + colliding height field
template <int SPEC> static bool sizes_match(int nq, int nv, int nu, int nl, int nlevel, int maxnnz, int ngt, int nseg, int ncg, int npair) {
  typedef Sizes<SPEC> Z;
  return nq == Z::nq && nv == Z::nv && nu == Z::nu && nl == Z::nl && nlevel == Z::nlevel && maxnnz == Z::maxnnz && ngt == Z::nu && nseg == Z::nseg &&
         ncg == Z::ncg && npair == Z::npair;
}

// compile-time layout of a size-specialised instantiation (SPEC models have no tendon limits: myo_model_load checks it)
template <int SPEC, int NVT, int KC, int NC, int NJ> struct LayC {
  typedef Sizes<SPEC> Z;
  static constexpr LayW L = layout_w(Z::nq, Z::nv, Z::nu, Z::nl, Z::nu, Z::maxnnz, Z::ncg, false, NVT, KC, NC, NJ);
  static constexpr int qpos = L.qpos, qvel = L.qvel, act = L.act, ctrl = L.ctrl, lpos = L.lpos, lmat = L.lmat, axis = L.axis, anchor = L.anchor, xv = L.xv,
                       qfc = L.qfc, sq = L.sq, mprw = L.mprw, X = L.X, tJ = L.tJ, tlen = L.tlen, tforce = L.tforce, cdof = L.cdof, cinert = L.cinert, crb = L.crb,
                       cvel = L.cvel, cacc = L.cacc, cfrc = L.cfrc, gpos = L.gpos, gax = L.gax, cand = L.cand, cdist = L.cdist, cpos = L.cpos, cnrm = L.cnrm,
                       cpair = L.cpair, cJ = L.cJ, cdofs = L.cdofs, Mp = L.Mp, tJp = L.tJp, total = L.total;
};
template <int SPEC, int NVT, int KC, int NC, int NJ> static bool layout_match(const LayW& a) {
  const LayW b = layout_w(Sizes<SPEC>::nq, Sizes<SPEC>::nv, Sizes<SPEC>::nu, Sizes<SPEC>::nl, Sizes<SPEC>::nu, Sizes<SPEC>::maxnnz, Sizes<SPEC>::ncg, false, NVT, KC, NC, NJ);
  return memcmp(&a, &b, sizeof(LayW)) == 0;
}

// TRK (MyoDM TrackEnv model class): condim-4 contacts (6 pyramid rows, a 4th jacobian row for the spin about the normal), joint friction-loss
// rows, box / convex-hull shapes in the narrow phase.  All of it sits behind `if constexpr (TRK)`: the other instantiations compile as before.
// RK4: mj_RungeKutta(4) instead of mj_Euler -- every substep runs the whole forward pass four times (state X0 + h a F[i-1], a = 1/2, 1/2, 1)
// and ends on X0 + h (F0 + 2 F1 + 2 F2 + F3) / 6; the saved state and the weighted derivative sums live in the registers of lane = dof /
// lane = actuator.  No implicit joint damping (an Euler-only feature of MuJoCo).
template <int NVT, int KC, int NC, int NTR, int WPE, bool SCHED, int SPEC, bool HF = false, bool TRK = false, bool RK4 = false>
__global__ void __launch_bounds__(64, WPE) step_kernel_w(const DevModel* __restrict__ Mp, const DevModelW* __restrict__ Wp, DevBatch Bt,
                                                        const float* __restrict__ action, int actmap, int nsub, long long* stamps,
                                                        const int* __restrict__ order, const DevWalk* __restrict__ wk, int kflags, SchedDev S) {
  extern __shared__ __align__(16) float E[];
  // the model structs stay in (scalar-cached) global memory: fields are s_load-ed where they are used instead of
  // pinning ~150 SGPRs for the whole kernel
  const DevModel& M = *Mp;
  const DevModelW& W = *Wp;
  // LDS layout: compile-time constants in the size-specialised instantiations (every LDS address an immediate), read from the model otherwise
  const LayC<SPEC, NVT, KC, NC, (TRK ? 4 : 3)> Yc{};
  const auto& Y = [&]() -> const auto& { if constexpr (SPEC != 0) return Yc; else return W.lay; }();
#define lane_id wave_lane()
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
  constexpr int NJ = TRK ? 4 : 3;     // jacobian rows per contact: normal, two tangents (, spin about the normal)
  constexpr int NR = TRK ? 6 : 4;     // pyramid rows per contact
  // the small instantiation (hand / finger class) is compiled without the free-joint, equality, plane-contact and condim-1 code;
  // myo_model_load routes any model that needs one of those to the large instantiation
  constexpr bool FULL = NVT > 24;
  const bool has_free = FULL && W.has_free;
  const int neq = FULL ? W.neq : 0;
  // tendon limits: never in the size-specialised and TRK instantiations (myo_model_load checks it), so their tendon lengths / velocities
  // do not stay live in registers from the tendon stage to the row stage
  const bool has_tl = (SPEC != 0 || TRK) ? false : (W.has_tl != 0);
  const bool walk = FULL && wk != nullptr;   // fused observation / reward pass of the walk task after the last substep
  if (!SCHED && FULL && (kflags & KF_RESET_ONLY) && Bt.elapsed[env] != 0) return;   // wave-uniform: refresh only the envs an auto-reset just touched
#if MYO_STAMPS
  long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long st_t0 = clock64();
  long long st_sub[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_s0 = 0;   // finer split (second half of the stamps buffer)
#define SUB0() do { st_s0 = clock64(); } while (0)
#define SUB(k) do { long long t1_ = clock64(); st_sub[k] += t1_ - st_s0; st_s0 = t1_; } while (0)
#else
#define SUB0() do { } while (0)
#define SUB(k) do { } while (0)
#endif
  const int nsubtot = nsub + (walk ? 1 : 0);
  const float h = M.timestep;
  // scheduler state of this wave: the queue of the XCD it runs on
  const int sq_q = SCHED ? (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7) % S.nqueue : 0;
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
  float actdot[NTR];   // (RK4 only)
  // the solver's warm start (= the previous substep's qacc) stays in the batch row Bt.warm between substeps: one coalesced store after the
  // solve, one load before the next (issued ahead of the row stage); a register for it would be live across every stage of the substep and
  // the 16-envs-per-CU LDS slice has no 24 floats to spare
  float* const warm_row = Bt.warm + (size_t)env * nv;
#pragma unroll
  for (int r = 0; r < NTR; r++) actdot[r] = 0.f;
#ifndef MYO_POISON_BITS
#define MYO_POISON_BITS 0x7fc00000
#endif
#if MYO_POISON   // diagnostic build: LDS words start as NaN, so a read of a word this launch never wrote shows up in the results
  // MYO_POISON = 1: everything; 2: state + frames (qpos .. anchor); 3: xv, qfc; 4: sq; 5: mprw / tJp; 6: region X
  { const int lo_ = MYO_POISON == 1 ? 0 : MYO_POISON == 2 ? 0 : MYO_POISON == 3 ? Y.xv : MYO_POISON == 4 ? Y.sq : MYO_POISON == 5 ? Y.mprw : Y.X;
    const int hi_ = MYO_POISON == 1 ? Y.total : MYO_POISON == 2 ? Y.xv : MYO_POISON == 3 ? Y.sq : MYO_POISON == 4 ? Y.mprw : MYO_POISON == 5 ? Y.X : Y.total;
    for (int i = lo_ + lane_id; i < hi_; i += 64) E[i] = __int_as_float(MYO_POISON_BITS); }
  SYNC();
#endif
  if (lane_id < nq) E[Y.qpos + lane_id] = ldstate<SCHED>(Bt.qpos + (size_t)env * nq + lane_id);
  if (lane_id < nv) {
    E[Y.qvel + lane_id] = ldstate<SCHED>(Bt.qvel + (size_t)env * nv + lane_id);
  }
  for (int i = lane_id; i < nu; i += 64) {
    E[Y.act + i] = ldstate<SCHED>(Bt.act + (size_t)env * nu + i);
    float c;
    if (action && s0 == 0) c = action_map(Bt, M.act, action, env, i, nu, actmap);   // the action map runs once per env step
    else c = ldstate<SCHED>(Bt.ctrl + (size_t)env * nu + i);
    E[Y.ctrl + i] = c;
  }
  float time = uniformf(ldstate<SCHED>(Bt.time + env));
  // MYO_TASK_TRACK (TRK models): this launch is a whole TrackEnv.step -- reference row of the pre-step time now, reward / done / reset at the end
  const bool track_on = TRK && Bt.track != nullptr && action != nullptr && actmap == MYO_ACTMAP_CTRLRANGE && nsub > 0;
  if constexpr (TRK) { if (track_on) track_lookup(*Bt.track, env, env + Bt.env_offset, time, Bt.elapsed[env], lane_id); }
  int flags = 0, d_nefc = 0, d_ncon = 0, d_iter = 0, d_cost = 0;
  int f_cand = 0, f_mpr = 0, f_ncon = 0, f_iter = 0, f_itcon = 0, f_ls = 0, f_fact = 0;   // work features of this env step (placement cost model)
  if (SCHED && s0 > 0) {   // accumulators of the earlier substeps of this env step
    const int* D = Bt.diag + (size_t)env * 8;
    int a2 = ldstatei<SCHED>(D + 2), a4 = ldstatei<SCHED>(D + 4), a5 = ldstatei<SCHED>(D + 5), a6 = ldstatei<SCHED>(D + 6), a7 = ldstatei<SCHED>(D + 7);
    d_nefc = ldstatei<SCHED>(D); d_ncon = ldstatei<SCHED>(D + 1);   // the observation pass has no rows of its own: keep the last substep's
    d_iter = a2; f_cand = a4 & 0xFFFF; f_ncon = a4 >> 16; f_mpr = a5; f_itcon = a6 & 0xFFFF; f_iter = a6 >> 16; f_ls = a7 & 0xFFFF; f_fact = a7 >> 16;
  }
  // per-env size of one collision geom (DevBatch.gsize), generic FULL instantiations only: the size-specialised and hand kernels keep
  // reading the model tables unconditionally
  constexpr bool OVR = FULL && SPEC == 0 && !HF;
  // pair record -> locals.  (OVR: one geom's size / bounding radius may be a per-env value, DevBatch.gsize)
  struct PairL { int g1, g2, dl, kc, pt, cd, t1, t2; float margin, gap, rb1, rb2, s1[3], s2[3]; };
  struct PairRaw { float4 q0, q1, q2, q3; };
  auto pair_raw = [&](int p) -> PairRaw { const gpf4 Q = W.pair_rec + 4 * (size_t)p; return PairRaw{Q[0], Q[1], Q[2], Q[3]}; };
  auto pair_decode = [&](const PairRaw& r) -> PairL {
    const float4 q0 = r.q0, q1 = r.q1, q2 = r.q2, q3 = r.q3;
    const int w = __float_as_int(q0.x), tt = __float_as_int(q3.x);
    PairL L;
    L.g1 = w & 255; L.g2 = (w >> 8) & 255; L.pt = (w >> 16) & 15; L.cd = (w >> 20) & 15; L.kc = (w >> 24) & 255; L.dl = __float_as_int(q0.w);
    L.margin = q0.y; L.gap = q0.z; L.t1 = tt & 255; L.t2 = (tt >> 8) & 255;
    L.s1[0] = q1.x; L.s1[1] = q1.y; L.s1[2] = q1.z; L.rb1 = q1.w; L.s2[0] = q2.x; L.s2[1] = q2.y; L.s2[2] = q2.z; L.rb2 = q2.w;
    if (OVR && Bt.gsize) {
      const float* G = Bt.gsize + 4 * (size_t)env;
      if (L.g1 == Bt.gsize_cg) { L.s1[0] = G[0]; L.s1[1] = G[1]; L.s1[2] = G[2]; L.rb1 = G[3]; }
      if (L.g2 == Bt.gsize_cg) { L.s2[0] = G[0]; L.s2[1] = G[1]; L.s2[2] = G[2]; L.rb2 = G[3]; }
    }
    return L;
  };
  auto pair_load = [&](int p) -> PairL { return pair_decode(pair_raw(p)); };
  // contacts NC .. NC + NCX - 1 live in this env's HBM overflow rows [dist, pos3, normal3, pair, cJ[3 KC], dof words]; lane = contact still holds
  // for all 64.  The first NC contacts (all of them for > 99.5 % of the states) never leave LDS.
  // TRK: a second bank of 64 (contacts 64 .. 127: lane = contact - 64), whose per-contact solver state lives in the contact's row as well
  constexpr int NCXK = TRK ? (128 - NC) : ((64 - NC) < NCX ? (64 - NC) : NCX);   // overflow rows this instantiation uses
  float* const ovf_env = Bt.ovf ? Bt.ovf + (size_t)env * Bt.ovf_rows * Bt.ovf_row : nullptr;
  const int ovf_row = Bt.ovf_row;
  const int nct = ovf_env ? NC + NCXK : NC;
  // narrow-phase round width: the MPR's per-lane LDS scratch (9 floats) lives in the contact-jacobian area, which holds 64 lanes' worth only
  // when NC * NJ * KC >= 576; the low-LDS instantiations (NC = 16) run the narrow phase in rounds of 32 candidates (typical count: 10-20)
  constexpr int RND = (NC * NJ * KC >= 768) ? 64 : 32;
  static_assert(RND * 12 <= NC * NJ * KC, "MPR scratch must fit the contact-jacobian area");
  int* const ovf_cand = (!HF && Bt.ovf_cand) ? Bt.ovf_cand + (size_t)env * NCANDX : nullptr;
  bool alive = true;
  int n_mprw = 0;   // MPR warm-start table (pair id + last contact normal in geom 1's frame): entries of the previous substep
  if (SCHED && s0 > 0) {   // ... which another wave ran: the table travels through the batch like the state rows, so that a scheduled
    // launch computes exactly what the one-wave-per-env launch does
    const int* Wt = Bt.mprw + (size_t)env * 64;
    n_mprw = __builtin_amdgcn_readfirstlane(ldstatei<SCHED>(Wt + 63));
    if (lane_id < 4 * n_mprw) ((int*)(E + Y.mprw))[lane_id] = ldstatei<SCHED>(Wt + lane_id);
  }
  SYNC();
  static_assert(!(RK4 && SCHED), "the substep scheduler hands out Euler substeps");
  for (int step = s0; step < s1; step++) {
    const bool op = walk && step == nsub;   // observation pass: position / velocity stages at the post-step state, then out
    // RK4 state of this substep (dead code otherwise): X0 and the weighted sums of the stage derivatives
    float rk_v0 = 0.f, rk_q0 = 0.f, rk_sv = 0.f, rk_sa = 0.f, rk_t0 = 0.f, rk_quat[4] = {1.f, 0.f, 0.f, 0.f}, rk_a0[NTR], rk_sd[NTR];
#pragma unroll
    for (int rr = 0; rr < NTR; rr++) { rk_a0[rr] = 0.f; rk_sd[rr] = 0.f; }
    int rk_stage = 0;
  rk_next_stage:
    // compiler-only barrier: keeps the (substep-invariant) model-table loads inside the loop body instead of hoisting
    // ~60 values per lane out of it and spilling them to scratch
    asm volatile("" ::: "memory");
    int lane;   // opaque per-iteration copy of the lane id: address arithmetic derived from it cannot be hoisted (and spilled)
    lane = wave_lane();
    {  // mj_checkPos / mj_checkVel
      bool bad = false;
      if (lane < nq) { float a = E[Y.qpos + lane]; bad = !(a == a) || fabsf(a) > MAXVALF; }
      if (lane < nv) { float b = E[Y.qvel + lane]; bad = bad || !(b == b) || fabsf(b) > MAXVALF; }
      if (__any(bad) && alive && !op) { flags |= MYO_FLAG_BAD_STATE; alive = false; }
    }
    STAMP(0);
    // ---------------------------------------------------------------- kinematics (lane = link, level by level)
    // Phase 1, lane = link: the link's own joint chain in its PARENT's frame -- rotation columns, origin and, per dof, axis and anchor --
    // written as 4 + 2 * dofnum vectors to the (at this point dead) Hessian scratch; no link waits for another one here, so the sines /
    // cosines and the joint rotations of all links are evaluated side by side instead of level after level.  Free-joint links (roots)
    // take their world pose straight from qpos.  All 64 lanes run the arithmetic on a clamped link index and only the stores are
    // predicated (a variant with the trigonometry inside `if (lane < ...)` miscompiled in the generic instantiation).
    unsigned int kw = (unsigned int)W.kin_pk[lane];   // phase 2, round 0 (its latency hides behind the joint trigonometry of phase 1)
    {
      const int l = lane < nl_ ? lane : 0;
      const bool mine = lane < nl_;
      const float* lp = M.link_pos + 3 * l;
      float A[9], c[3] = {lp[0], lp[1], lp[2]};
#pragma unroll
      for (int k = 0; k < 9; k++) A[k] = W.link_mat0[9 * l + k];
      const int da = M.link_dofadr[l];
      int dn = M.link_dofnum[l];
      const bool isfree = has_free && W.link_free[l];
      if (isfree) {
        // free joint: pose straight from qpos (position + unit quaternion); its 3 translational dofs act like slides along
        // the world axes and its 3 rotational dofs like hinges about the body axes through the body origin
        const int qa = W.dof_qposadr[da];
        float pos[3] = {E[Y.qpos + qa], E[Y.qpos + qa + 1], E[Y.qpos + qa + 2]}, R[9];
        float q[4] = {E[Y.qpos + qa + 3], E[Y.qpos + qa + 4], E[Y.qpos + qa + 5], E[Y.qpos + qa + 6]};
        float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        q[0] *= qn; q[1] *= qn; q[2] *= qn; q[3] *= qn;
        quat2mat(R, q);
        if (mine) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            E[Y.axis + 3 * (da + k)] = k == 0 ? 1.f : 0.f; E[Y.axis + 3 * (da + k) + 1] = k == 1 ? 1.f : 0.f; E[Y.axis + 3 * (da + k) + 2] = k == 2 ? 1.f : 0.f;
            E[Y.axis + 3 * (da + 3 + k)] = R[k]; E[Y.axis + 3 * (da + 3 + k) + 1] = R[3 + k]; E[Y.axis + 3 * (da + 3 + k) + 2] = R[6 + k];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) { E[Y.anchor + 3 * (da + k) + cc] = pos[cc]; E[Y.anchor + 3 * (da + 3 + k) + cc] = pos[cc]; }
          }
#pragma unroll
          for (int k = 0; k < 3; k++) E[Y.lpos + 3 * l + k] = pos[k];
#pragma unroll
          for (int k = 0; k < 9; k++) E[Y.lmat + 9 * l + k] = R[k];
        }
        dn = 0;
      }
      float* const ks = E + Y.sq + W.kin_base[l];
      // uniform trip count (the model's longest chain) with the body predicated per lane: a loop whose trip count differs between the lanes
      // would be a long divergent region around the trigonometry, with register spills inside it
      const int dnmax = W.kin_dnmax;
      for (int k = 0; k < dnmax; k++) {
        const bool act = k < dn;
        const int d = act ? da + k : da;
        const float* al = M.dof_axis + 3 * d;
        const float* dp = M.dof_pos + 3 * d;
        float ax[3], an[3];
        matvec(ax, A, al);
        matvec(an, A, dp);
        an[0] += c[0]; an[1] += c[1]; an[2] += c[2];
        if (mine && act) {
          ks[12 + 6 * k] = ax[0]; ks[13 + 6 * k] = ax[1]; ks[14 + 6 * k] = ax[2];
          ks[15 + 6 * k] = an[0]; ks[16 + 6 * k] = an[1]; ks[17 + 6 * k] = an[2];
        }
        const int qa = W.dof_qposadr[d];
        const float ang = E[Y.qpos + qa] - M.qpos0[qa];
        const bool hinge = M.dof_type[d] == 3;
        float sn, cs;
        sincos_jf(ang, &sn, &cs);
        const float oc = 1 - cs, x = al[0], y = al[1], z = al[2];
        const float Rj[9] = {cs + oc * x * x, oc * x * y - sn * z, oc * x * z + sn * y, oc * x * y + sn * z, cs + oc * y * y, oc * y * z - sn * x,
                             oc * x * z - sn * y, oc * y * z + sn * x, cs + oc * z * z};
        float An[9], v[3];
        matmul3(An, A, Rj);
        matvec(v, An, dp);
        const bool rot = act && hinge, lin = act && !hinge;
#pragma unroll
        for (int i = 0; i < 9; i++) A[i] = rot ? An[i] : A[i];
#pragma unroll
        for (int i = 0; i < 3; i++) c[i] = rot ? an[i] - v[i] : (lin ? c[i] + ax[i] * ang : c[i]);
      }
      if (mine && !isfree) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { ks[3 * cc] = A[cc]; ks[3 * cc + 1] = A[3 + cc]; ks[3 * cc + 2] = A[6 + cc]; }   // column cc of the local rotation
        ks[9] = c[0]; ks[10] = c[1]; ks[11] = c[2];
      }
    }
    SYNC();
    // Phase 2, level by level, lane = (link of the level, vector): world = parent rotation x local vector (+ parent origin for points).
    // One packed word per lane and round (DevModelW::kin_pk), the next round's word in flight while this one is worked on.
    for (int r = 0; r < W.kin_nround; r++) {
      const unsigned int w0 = kw;
      kw = (unsigned int)W.kin_pk[(r + 1) * 64 + lane];
      if (w0 != 0xFFFFFFFFu) {
        const int src = w0 & 2047, l = (w0 >> 11) & 63, kind = (w0 >> 17) & 3, ix = (w0 >> 19) & 63, par = (int)(w0 >> 25) - 1;
        const float v[3] = {E[Y.sq + src], E[Y.sq + src + 1], E[Y.sq + src + 2]};
        float w[3] = {v[0], v[1], v[2]};
        if (par >= 0) {
          matvec(w, E + Y.lmat + 9 * par, v);
          if (kind & 1) { w[0] += E[Y.lpos + 3 * par]; w[1] += E[Y.lpos + 3 * par + 1]; w[2] += E[Y.lpos + 3 * par + 2]; }
        }
        if (kind == 0) { E[Y.lmat + 9 * l + ix] = w[0]; E[Y.lmat + 9 * l + 3 + ix] = w[1]; E[Y.lmat + 9 * l + 6 + ix] = w[2]; }
        else {
          float* const dst = kind == 1 ? E + Y.lpos + 3 * l : (kind == 2 ? E + Y.axis + 3 * ix : E + Y.anchor + 3 * ix);
          dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2];
        }
      }
      SYNC();
    }
    if constexpr (TRK) {   // link frames of the last substep's position stage (MYO_F_LINKX), world coordinates
      if (Bt.linkx && step == nsub - 1) {
        float* o = Bt.linkx + (size_t)env * 12 * nl_;
        for (int i = lane; i < 12 * nl_; i += 64) { const int l = i / 12, k = i - 12 * l; o[i] = k < 3 ? E[Y.lpos + 3 * l + k] + M.origin[k] : E[Y.lmat + 9 * l + (k - 3)]; }
      }
    }
    // reference point of the spatial (6-D) quantities: fixed for fixed-base models, the root link's origin for free-floating ones
    const float c0[3] = {has_free ? E[Y.lpos] : M.c0[0], has_free ? E[Y.lpos + 1] : M.c0[1], has_free ? E[Y.lpos + 2] : M.c0[2]};
    STAMP(1);
    SUB0();
    // ---------------------------------------------------------------- tendons: lane = segment
    float tlen_r[NTR], tvel_r[NTR];
    if (W.has_j0) {   // joint transmission: constant moment arm, length = arm * joint coordinate (mj_transmission, mjTRN_JOINT)
      WFOR(i, ngt_ * maxnnz_) E[Y.tJ + i] = W.gt_j0[i];
      WFOR(i, ngt_) {
        float L = M.gt_len0[i];
        for (int k = 0; k < maxnnz_; k++) { const float a = W.gt_j0[i * maxnnz_ + k]; if (a != 0.f) L += a * E[Y.qpos + W.dof_qposadr[M.gt_dofs[i * maxnnz_ + k]]]; }
        E[Y.tlen + i] = L;
      }
    } else {
    WFOR(i, ngt_ * maxnnz_) E[Y.tJ + i] = 0.f;
    WFOR(i, ngt_) E[Y.tlen + i] = M.gt_len0[i];   // constant same-link segments, folded at lowering time
    }
    SYNC();
    for (int base = 0; base < nseg_; base += 64) {
      int idx = base + lane;
      if (idx < nseg_) {
        // the segment's record: four independent 16-byte loads (five more for a wrapping segment) carry everything the old chain
        // seg_order -> seg -> site_link / site_lpos / wg_* read word by word
        const gpf4 SR = W.seg_rec + (size_t)idx * SEGR;
        const float4 r0 = SR[0], r1 = SR[1], r2 = SR[2], r3 = SR[3];
        // (the wrapping segments come first in the order: in their rounds every lane asks for the whole record at once instead of waiting for
        // `g` to arrive before the second half is requested)
        float4 r4 = r0, r5 = r0, r6 = r0, r7 = r0, r8 = r0;
        if (base < W.nwrapseg) { r4 = SR[4]; r5 = SR[5]; r6 = SR[6]; r7 = SR[7]; r8 = SR[8]; }
        const int g = __float_as_int(r2.x), side_l = __float_as_int(r2.y), gts = __float_as_int(r2.w);
        const float invdiv = r2.z;
        float p0[3], p1[3];
        { const float lp[3] = {r0.y, r0.z, r0.w}; frame_point(Y, E, __float_as_int(r0.x), lp, p0); }
        { const float lp[3] = {r1.y, r1.z, r1.w}; frame_point(Y, E, __float_as_int(r1.x), lp, p1); }
        // first rows of the segment's moment-arm lists, in flight while the wrap geometry is worked out
        const int wa = __float_as_int(r3.x), wb = __float_as_int(r3.y), wc = __float_as_int(r3.z);
        const gpi4 DL = (gpi4)W.dl_pk;
        const int4 ea = DL[wa & 0xFFFFF];
        int4 eb = ea, ec = ea;
        if (g >= 0) { eb = DL[wb & 0xFFFFF]; ec = DL[wc & 0xFFFFF]; }
        float wlen = -1, wp[6];
        if (g >= 0) {
          const int gl = __float_as_int(r5.x);
          const float glp[3] = {r5.y, r5.z, r5.w}, glm[9] = {r6.x, r6.y, r6.z, r6.w, r7.x, r7.y, r7.z, r7.w, r8.x};
          float gpos[3], gmat[9], side[3] = {0, 0, 0};
          if (gl < 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) gpos[k] = glp[k];
#pragma unroll
            for (int k = 0; k < 9; k++) gmat[k] = glm[k];
          } else {
            float v[3];
            matvec(v, E + Y.lmat + 9 * gl, glp);
#pragma unroll
            for (int k = 0; k < 3; k++) gpos[k] = E[Y.lpos + 3 * gl + k] + v[k];
            matmul3(gmat, E + Y.lmat + 9 * gl, glm);
          }
          if (side_l != -2) { const float lp[3] = {r4.x, r4.y, r4.z}; frame_point(Y, E, side_l, lp, side); }
          wlen = wrap_geom_inl(wp, p0, p1, gpos, gmat, r4.w, __float_as_int(r3.w) != 0, side, side_l != -2);   // always inline: an out-of-line copy passes its arrays through scratch memory
        }
        SUB(7);
        bool wr = wlen >= 0;
        float* Jt = E + Y.tJ + gts * maxnnz_;
        float L = straight_w(W, Y, E, Jt, p0, p1, wa & 0xFFFFF, wa >> 20, invdiv, !wr, ea);
        if (g >= 0) {
          L += straight_w(W, Y, E, Jt, p0, wp, wb & 0xFFFFF, wb >> 20, invdiv, wr, eb);
          L += straight_w(W, Y, E, Jt, wp + 3, p1, wc & 0xFFFFF, wc >> 20, invdiv, wr, ec);
          if (wr) L += wlen * invdiv;
        }
        atomicAdd(&E[Y.tlen + gts], L);
      }
    }
    if (lane < nv) E[Y.qfc + lane] = 0.f;   // actuator forces are scattered to their dofs below (the solver's force scratch is free here)
    SYNC();
    SUB(8);
#pragma unroll
    for (int rr = 0; rr < NTR; rr++) {  // lane = tendon (NTR rounds of 64): gather its segments, then the muscle
      int gt = lane + 64 * rr;
      tlen_r[rr] = 0.f; tvel_r[rr] = 0.f;
      if (gt >= ngt_) continue;
      const float* Jrow = E + Y.tJ + gt * maxnnz_;
      const float L = E[Y.tlen + gt];
      tlen_r[rr] = L;
      float vel = 0;
      for (int k = 0; k < maxnnz_; k++) {
        int d = M.gt_dofs[gt * maxnnz_ + k];
        if (d >= 0) vel += Jrow[k] * E[Y.qvel + d];
        if (has_tl) E[Y.tJp + gt * maxnnz_ + k] = Jrow[k];
      }
      tvel_r[rr] = vel;
      if (has_tl) { E[Y.tJp + ngt_ * maxnnz_ + gt] = L; E[Y.tJp + ngt_ * maxnnz_ + ngt_ + gt] = vel; }
      if (gt < nu) {
        const float* A = M.act + 16 * gt;
        float f, ad;
        if (A[10] < 0.f) { f = A[0] * clipf(E[Y.ctrl + gt], A[12], A[13]) + A[1] + A[14] * (A[2] * L + A[3] * vel); ad = 0.f; }   // stateless affine actuator
        else muscle(A, A[14] * L, A[14] * vel, E[Y.act + gt], E[Y.ctrl + gt], &f, &ad);
        if constexpr (RK4) actdot[rr] = ad;
        else if (!op) E[Y.act + gt] += h * ad;   // Euler: the activation is advanced right here (nothing reads it again in this substep; a bad-state env is
                                                 // reset as a whole afterwards), so no derivative stays live in a register across collision and solver
        const float ft = f * A[14];
        E[Y.tforce + gt] = ft;
        // J^T f of the actuators, lane = tendon: its <= maxnnz moment arms go to their dofs with LDS atomics (one lane per dof walking the
        // dof's whole column -- two dozen tendons for the wrist dofs -- was the longer chain)
        for (int k = 0; k < maxnnz_; k++) {
          const int d = M.gt_dofs[gt * maxnnz_ + k];
          if (d >= 0) atomicAdd(&E[Y.qfc + d], Jrow[k] * ft);
        }
      }
    }
    SYNC();
    const float qfa = lane < nv ? E[Y.qfc + lane] : 0.f;
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
    SUB(9);
    STAMP(2);
    // ---------------------------------------------------------------- CRB + RNE (lane = link / dof)
    // tree words of this lane's link / dof and of the subtree-sum tasks, loaded here so that the sweeps below find them in registers
    typedef unsigned long long ull;
    constexpr bool M64 = NVT > 32;      // link / dof masks of the small instantiations fit the low word
    auto ld_mask = [&](gpi tab, int i) -> ull { const unsigned int lo = (unsigned int)tab[2 * i], hi = M64 ? (unsigned int)tab[2 * i + 1] : 0u; return (ull)lo | ((ull)hi << 32); };
    constexpr int BKG = SPEC ? (16 * Z::nl + 63) / 64 : 4;   // rounds of subtree-sum tasks per group (size-specialised: all of them)
    const ull adof_m = lane < nl_ ? ld_mask(W.link_adof, lane) : 0ull, anc_m = lane < nv ? ld_mask(W.dof_anc, lane) : 0ull;
    ull desc_m[BKG];
#pragma unroll
    for (int u = 0; u < BKG; u++) { const int l = (u * 64 + lane) >> 4; desc_m[u] = l < nl_ ? ld_mask(W.link_desc, l) : 0ull; }
    const int my_link = M.dof_link[lane < nv ? lane : 0];
    const float my_arm = M.dof_armature[lane < nv ? lane : 0], my_damp = M.dof_damping[lane < nv ? lane : 0];
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
    SUB(10);
    // velocity / acceleration sweep (mj_comVel + the forward half of mj_rne), ONE pass: lane = link walks the dofs of its whole
    // ancestor chain root-first (lowering table).  The chains are <= 7 dofs long, so redoing a parent's sums in every descendant
    // lane costs less than a level-by-level sweep with one barrier and a handful of active lanes per level.
    if (lane < nl_) {
      const int l = lane;
      float cvel[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cacc[6] = {0.f, 0.f, 0.f, -M.grav[0], -M.grav[1], -M.grav[2]}, cvel_rot[6];
      const ull frot = has_free ? ((ull)W.free_rot[0] | ((ull)W.free_rot[1] << 32)) : 0ull, fj3 = has_free ? ((ull)W.free_j3[0] | ((ull)W.free_j3[1] << 32)) : 0ull;
      for (ull am = adof_m; am; am &= am - 1ull) {   // dofs of the chain root-first = ascending (DevModelW::link_adof): no table read inside the loop
        const int d = __builtin_ctzll(am);
        const bool rotf = has_free && ((frot >> d) & 1ull), j3 = has_free && ((fj3 >> d) & 1ull);
        float cd[6], cdd[6], qv = E[Y.qvel + d];
#pragma unroll
        for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
        if (j3) {
#pragma unroll
          for (int k = 0; k < 6; k++) cvel_rot[k] = cvel[k];   // velocity after the translations, before any of the 3 rotations
        }
        cross_motion(cdd, rotf ? cvel_rot : cvel, cd);
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
      for (int k = 0; k < 6; k++) { E[Y.cvel + 6 * l + k] = cvel[k]; E[Y.cfrc + 6 * l + k] = f[k] + t1[k]; }
    }
    SYNC();
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
          if (wk->knee_height > 0.f && height - 0.5f * (pl[2] + pr[2]) < wk->knee_height) done = 1.f;   // TerrainEnvV0._get_knee_condition (walk_v0.py:660-671)
          Bt.reward[env] = wk->w_vel * vel_reward + wk->w_done * done + wk->w_cyc * cyclic + wk->w_rot * ref_rot + wk->w_ja * ja;
          Bt.done[env] = done;
          Bt.solved[env] = vel_reward >= 1.0f ? 1.f : 0.f;
        }
      }
      break;
    }
    // subtree sums of the link forces (6) and composite inertias (10): lane = (link, component) adds up the link's whole subtree (DevModelW::link_desc)
    // from the values the links wrote themselves, so no task waits for another one -- no level-by-level sweep with its chain of
    // level_adr -> child_adr -> child reads.  In place: a group of rounds reads, then writes; a later group (higher links) only reads links above its
    // own, which no earlier group has written.
    {
      const int nbk = (16 * nl_ + 63) >> 6;
      for (int r0 = 0; r0 < nbk; r0 += BKG) {
        float acc[BKG];
#pragma unroll
        for (int u = 0; u < BKG; u++) {
          const int idx = (r0 + u) * 64 + lane, l = idx >> 4, k = idx & 15;
          const int base = k < 6 ? Y.cfrc + k : Y.crb + (k - 6), str = k < 6 ? 6 : 10;
          ull dm = r0 == 0 ? desc_m[u] : (l < nl_ ? ld_mask(W.link_desc, l) : 0ull);
          float a = 0.f;
          for (; dm; dm &= dm - 1ull) a += E[base + str * __builtin_ctzll(dm)];
          acc[u] = a;
        }
        SYNC();
#pragma unroll
        for (int u = 0; u < BKG; u++) {
          const int idx = (r0 + u) * 64 + lane, l = idx >> 4, k = idx & 15;
          if (l < nl_) E[(k < 6 ? Y.cfrc + k : Y.crb + (k - 6)) + (k < 6 ? 6 : 10) * l] = acc[u];
        }
        SYNC();
      }
    }
    float smooth = 0.f;
    if (lane < nv) {
      int d = lane, l = my_link;
      float cd[6], buf[6], crb[10];
#pragma unroll
      for (int k = 0; k < 6; k++) cd[k] = E[Y.cdof + 6 * d + k];
#pragma unroll
      for (int k = 0; k < 10; k++) crb[k] = E[Y.crb + 10 * l + k];
      float bias = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) bias += cd[k] * E[Y.cfrc + 6 * l + k];
      mul_inert_vec(buf, crb, cd);
      for (ull am = anc_m; am;) {   // the dof and its ancestors (DevModelW::dof_anc), highest first
        const int a = 63 - __builtin_clzll(am);
        am ^= 1ull << a;
        float sdot = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sdot += E[Y.cdof + 6 * a + k] * buf[k];
        if (a == d) sdot += my_arm;
        E[Y.sq + d * (NVT + 1) + a] = sdot;   // full symmetric copy: (d,a) and (a,d)
        E[Y.sq + a * (NVT + 1) + d] = sdot;
      }
      smooth = -my_damp * E[Y.qvel + d] - bias + qfa;
    }
    SYNC();  // region X changes owner: dynamics scratch -> collision / contact rows
    SUB(11);
    STAMP(3);
    // ---------------------------------------------------------------- collision (geom frames computed on the fly)
    int ncon = 0;
    if (!M.disable_contact) {
      int ncand = 0;
      int* cand = (int*)(E + Y.cand);
      PairRaw pnext = pair_raw(min(lane, npair_ > 0 ? npair_ - 1 : 0));   // broad phase, round 0: requested here, behind the geom frames
      for (int g = lane; g < ncg_; g += 64) {   // world centre and long axis (3rd column) of every collision geom (more than 64: MyoDM teapot, wineglass)
        float x[3], R[9];
        geom_world_pos(W, Y, E, g, x);
        geom_world_mat(W, Y, E, g, R);
        E[Y.gpos + 3 * g] = x[0]; E[Y.gpos + 3 * g + 1] = x[1]; E[Y.gpos + 3 * g + 2] = x[2];
        E[Y.gax + 3 * g] = R[2]; E[Y.gax + 3 * g + 1] = R[5]; E[Y.gax + 3 * g + 2] = R[8];
      }
      SYNC();
      for (int base = 0; base < npair_; base += 64) {
        int p = base + lane;
        bool hit = false;
        int nh = 0, hr0 = 0, hr1 = 0, hc0 = 0, hc1 = 0;   // height-field pair: cell range under the geom, prisms that can touch it
        float hzcut = 0.f;
        const PairRaw praw = pnext;                                   // this round's record was requested a round ago
        pnext = pair_raw(min(p + 64, npair_ > 0 ? npair_ - 1 : 0));   // next round's, in flight while this one is tested
        if (p < npair_) {
          const PairL Q = pair_decode(praw);   // one record: four independent 16-byte loads (was pair_i -> cg_rbound / cg_type / cg_size -> pair_f, word by word)
          const int P[6] = {Q.g1, Q.g2, Q.dl, Q.kc, Q.pt, Q.cd};
          if (HF && P[4] == 4) {
            const int g2 = P[1], ty = Q.t2;
            const float *x2 = E + Y.gpos + 3 * g2, *ax = E + Y.gax + 3 * g2, *sz = Q.s2;
            const float rel[3] = {x2[0] - W.hf.pos[0], x2[1] - W.hf.pos[1], x2[2] - W.hf.pos[2]}, margin = Q.margin;
            float ext[3];
            if (ty == GEOM_ELLIPSOID) {
              float R[9];
              geom_world_mat(W, Y, E, g2, R);
#pragma unroll
              for (int k = 0; k < 3; k++) { const float a = R[3 * k] * sz[0], b = R[3 * k + 1] * sz[1], c = R[3 * k + 2] * sz[2]; ext[k] = sqrtf(a * a + b * b + c * c); }
            } else {
#pragma unroll
              for (int k = 0; k < 3; k++)
                ext[k] = ty == GEOM_SPHERE ? sz[0] : (ty == GEOM_CAPSULE ? sz[0] + sz[1] * fabsf(ax[k]) : sz[1] * fabsf(ax[k]) + sz[0] * sqrtf(fmaxf(0.f, 1.f - ax[k] * ax[k])));
            }
            float zmin;
            if (hf_range(W.hf, rel, ext, Q.rb2, margin, hr0, hr1, hc0, hc1, zmin)) {
              hzcut = zmin - margin;
              nh = hf_walk(W.hf, Bt.hfield + (size_t)env * W.hf.nrow * W.hf.ncol, hr0, hr1, hc0, hc1, hzcut, p, nullptr, 0, 0);
            }
          } else if (!(M.disable_ellipsoid && P[4] == 0)) {
            int g1 = P[0], g2 = P[1];
            const float *x1 = E + Y.gpos + 3 * g1, *x2 = E + Y.gpos + 3 * g2;
            float dif[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
            float bound = Q.rb1 + Q.rb2 + Q.margin;
            if (FULL && P[4] >= 2) hit = dot3(dif, E + Y.gax + 3 * g1) <= Q.rb2 + Q.margin;   // plane: signed distance of the bounding sphere
            else hit = dot3(dif, dif) <= bound * bound;
            if constexpr (TRK) {
              // a box (table top: bounding sphere 0.7 m) is tested as a box, a hull as the bounding box of its vertices in the mesh frame
              // (lowering.py hip_mesh_aabb: centre | half sizes): distance from the other geom's centre to that box against the other
              // geom's bounding sphere + margin.  The airplane's outer hull has a 0.10 m bounding sphere and thin wings.
              const int t1 = Q.t1, t2 = Q.t2;
              if (hit && P[4] == 5) {   // plane - hull: the lowest corner of the hull's vertex bounding box along the plane normal
                float R2[9], nl[3];
                const float* n = E + Y.gax + 3 * g1;
                geom_world_mat(W, Y, E, g2, R2);
                matTvec(nl, R2, n);
                gpf bx = W.mesh_aabb + 6 * (int)Q.s2[2];
                const float low = dot3(dif, n) + nl[0] * bx[0] + nl[1] * bx[1] + nl[2] * bx[2] - (fabsf(nl[0]) * bx[3] + fabsf(nl[1]) * bx[4] + fabsf(nl[2]) * bx[5]);
                hit = low <= Q.margin;
              }
              if (hit && (t1 >= 6 || t2 >= 6) && P[4] == 0) {
#pragma unroll
                for (int side = 0; side < 2; side++) {
                  const int gb = side ? g2 : g1, go = side ? g1 : g2, tb = side ? t2 : t1;
                  if (tb < 6 || !hit) continue;
                  float Rb[9], cl[3], dd[3] = {E[Y.gpos + 3 * go] - E[Y.gpos + 3 * gb], E[Y.gpos + 3 * go + 1] - E[Y.gpos + 3 * gb + 1], E[Y.gpos + 3 * go + 2] - E[Y.gpos + 3 * gb + 2]};
                  geom_world_mat(W, Y, E, gb, Rb);
                  matTvec(cl, Rb, dd);
                  const float* sb = side ? Q.s2 : Q.s1;
                  float hx = sb[0], hy = sb[1], hz = sb[2];
                  if (tb == 7) { gpf bx = W.mesh_aabb + 6 * (int)sb[2]; cl[0] -= bx[0]; cl[1] -= bx[1]; cl[2] -= bx[2]; hx = bx[3]; hy = bx[4]; hz = bx[5]; }
                  const float ex = fmaxf(fabsf(cl[0]) - hx, 0.f), ey = fmaxf(fabsf(cl[1]) - hy, 0.f), ez = fmaxf(fabsf(cl[2]) - hz, 0.f);
                  const float lim = (side ? Q.rb1 : Q.rb2) + Q.margin;
                  hit = ex * ex + ey * ey + ez * ez <= lim * lim;
                }
              }
            }
            if (hit && !P[4]) {
              // conservative refinement before the expensive MPR: replace a capsule's bounding sphere by the distance
              // from the other geom's centre to the capsule's SEGMENT (a bound on the true distance, never excludes a contact)
              float b1 = Q.rb1, b2 = Q.rb2;
              float c1[3] = {x1[0], x1[1], x1[2]}, c2[3] = {x2[0], x2[1], x2[2]};
              if (Q.t1 == GEOM_CAPSULE) {
                const float* a = E + Y.gax + 3 * g1;
                float hh = Q.s1[1], t = clipf(dot3(dif, a), -hh, hh);
                c1[0] += t * a[0]; c1[1] += t * a[1]; c1[2] += t * a[2];
                b1 = Q.s1[0];
              }
              if (Q.t2 == GEOM_CAPSULE) {
                const float* a = E + Y.gax + 3 * g2;
                float nd[3] = {c1[0] - x2[0], c1[1] - x2[1], c1[2] - x2[2]};
                float hh = Q.s2[1], t = clipf(dot3(nd, a), -hh, hh);
                c2[0] += t * a[0]; c2[1] += t * a[1]; c2[2] += t * a[2];
                b2 = Q.s2[0];
              }
              float d2[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
              float bb = b1 + b2 + Q.margin;
              hit = dot3(d2, d2) <= bb * bb;
              if (hit) {
                // separating-axis test along the centre line: the two (margin-inflated) convex shapes cannot touch if their
                // support widths along that axis do not reach across the centre distance.  MPR would report "no contact" for
                // exactly these pairs, after a dozen support evaluations; this costs one support width per shape
                float dn = norm3(dif);
                if (dn > MINVALF) {
                  float inv = 1.0f / dn, ax[3] = {dif[0] * inv, dif[1] * inv, dif[2] * inv}, wsum = Q.margin;
#pragma unroll
                  for (int side = 0; side < 2; side++) {
                    const int g = side ? g2 : g1;
                    const float* sz = side ? Q.s2 : Q.s1;
                    const int ty = side ? Q.t2 : Q.t1;
                    if (TRK && ty >= 6) wsum += 1e9f;   // box / hull: no cheap support width here, the pair goes to MPR
                    else if (ty == GEOM_CAPSULE) wsum += sz[0] + sz[1] * fabsf(dot3(E + Y.gax + 3 * g, ax));
                    else if (ty == GEOM_SPHERE) wsum += sz[0];
                    else {
                      float R[9], dl[3];
                      geom_world_mat(W, Y, E, g, R);
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
        if (hit) { if (pos < NCAND) cand[pos] = p; else if (ovf_cand && pos < NCAND + NCANDX) ovf_cand[pos - NCAND] = p; }
        ncand += __popcll(bal);
        if (HF) {   // height-field pairs expand into one candidate per prism, appended in pair order
          unsigned long long hb = __ballot(nh > 0);
          int myat = 0;
          while (hb) {
            const int L = __ffsll((long long)hb) - 1;
            hb &= hb - 1ull;
            if (lane == L) myat = ncand;
            ncand += rdlanei(nh, L);
          }
          if (nh > 0) hf_walk(W.hf, Bt.hfield + (size_t)env * W.hf.nrow * W.hf.ncol, hr0, hr1, hc0, hc1, hzcut, p, cand, myat, NCAND);
        }
      }
      { const int candcap = ovf_cand ? NCAND + NCANDX : NCAND; if (ncand > candcap) { flags |= MYO_FLAG_CAND_OVERFLOW; ncand = candcap; } }
      f_cand += ncand;
      int n_mprw_new = 0;
      SYNC();
      STAMP(6);
      MeshTab MT;   // hull tables of the TRK models (wave-uniform)
      if constexpr (TRK) { MT.vert = W.mesh_vert; MT.rec = (gpf4)W.mesh_rec; MT.srec = (gpf4)W.mesh_startrec; }
      for (int base = 0; base < ncand; base += RND) {
        int ci = (RND == 64 || lane < RND) ? base + lane : ncand;
        int nsup = -8;                    // support evaluations of this lane's MPR refinement (-8: not an MPR pair)
        bool mpr_hit = false;             // this lane's MPR call found a contact: its normal seeds the next substep's call
        float mpr_n[3] = {0.f, 0.f, 0.f};
        bool hit = false, hit2 = false;   // a plane-capsule pair can give two contacts (one per end sphere)
        float dist = 0, dist2 = 0, cpos[3] = {0, 0, 0}, cpos2[3] = {0, 0, 0}, nrm[3] = {1, 0, 0};
        int p = -1, cword = 0;            // cword: what the row stage needs of the pair without another table read (pair | dofs << 11 | dof-list start << 16)
        if (ci < ncand) {
          const int cw = (ci < NCAND) ? cand[ci] : ovf_cand[ci - NCAND];
          p = HF ? (cw & 1023) : cw;
          const PairL Q = pair_load(p);
          cword = p | (Q.kc << 11) | (Q.dl << 16);
          const int P[6] = {Q.g1, Q.g2, Q.dl, Q.kc, Q.pt, Q.cd};
          int g1 = P[0], g2 = P[1];
          float margin = Q.margin;
          const float *x1 = E + Y.gpos + 3 * g1, *x2 = E + Y.gpos + 3 * g2;
          const float *sz1 = Q.s1, *sz2 = Q.s2;
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
          } else if (FULL && TRK && P[4] == 5) {   // plane - convex hull: deepest vertex along -normal (one contact)
            const float* n = E + Y.gax + 3 * g1;
            float R2[9], nl[3], pw[3];
            geom_world_mat(W, Y, E, g2, R2);
            matTvec(nl, R2, n);
            CObj oh;
            cobj_shape_poly(oh, 7, sz2);
            const float dn[3] = {-nl[0], -nl[1], -nl[2]};
            float sp[3];
            support_shape<2>(oh, dn, sp, MT);        // vertex-graph climb (scan for small hulls) instead of a pass over all vertices
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
          } else if (FULL && P[4] == 3) {   // plane - ellipsoid (mjc_PlaneConvex): deepest support point along -normal
            const float* n = E + Y.gax + 3 * g1;
            float R2[9], nl[3], sp[3], pw[3];
            geom_world_mat(W, Y, E, g2, R2);
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
            if constexpr (HF) {   // height-field kernels: generic convex pairs and prisms share one MPR call site
            const float zero3[3] = {0.f, 0.f, 0.f};
            nsup = 0;
            // MPR in geom1's own frame: obj1 needs no rotation / translation at all (identity frame), obj2 carries the
            // relative pose R1^T R2, R1^T (x2 - x1); normal and position are rotated back afterwards
            float R1[9], cen[3] = {0.f, 0.f, 0.f};   // (cen: prism centroid; dead code unless HF -- nothing extra stays live across the MPR call)
            CObj o1, o2;
            const bool prism = HF && P[4] == 4;
            if (prism) {
              // obj1 = one triangular prism of the height field, about its centroid, in the (axis-aligned) height-field frame
              float hx[3], hy[3], hz[3];
              hf_prism(W.hf, Bt.hfield + (size_t)env * W.hf.nrow * W.hf.ncol, (cw >> 10) & 127, cw >> 17, hx, hy, hz);
              cen[0] = (hx[0] + hx[1] + hx[2]) * (1.f / 3.f); cen[1] = (hy[0] + hy[1] + hy[2]) * (1.f / 3.f); cen[2] = 0.5f * ((hz[0] + hz[1] + hz[2]) * (1.f / 3.f) - W.hf.size[3]);
#pragma unroll
              for (int k = 0; k < 3; k++) { o1.mat[k] = hx[k] - cen[0]; o1.mat[3 + k] = hy[k] - cen[1]; o1.mat[6 + k] = hz[k] - cen[2]; o1.pos[k] = 0.f; }
              o1.S[0] = -W.hf.size[3] - cen[2]; o1.S[1] = o1.S[2] = 0.f; o1.h = -1.f;
#pragma unroll
              for (int k = 0; k < 9; k++) R1[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
              geom_world_mat(W, Y, E, g2, o2.mat);
#pragma unroll
              for (int k = 0; k < 3; k++) o2.pos[k] = x2[k] - x1[k] - cen[k];
              cobj_shape(o2, Q.t2, sz2);
            } else {
            geom_world_mat(W, Y, E, g1, R1);
            {
              float R2[9], rel[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
              geom_world_mat(W, Y, E, g2, R2);
#pragma unroll
              for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) o2.mat[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
              matTvec(o2.pos, R1, rel);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) o1.mat[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) o1.pos[k] = 0.f;
            cobj_shape(o1, Q.t1, sz1); cobj_shape(o2, Q.t2, sz2);
            }
            o1.margin = o2.margin = 0.5f * margin;
            float depth, dir[3], pos[3], nw[3] = {0.f, 0.f, 0.f};
            bool have_nw = false;
            for (int i = 0; i < (prism ? 0 : n_mprw); i++) {   // (prisms are not warm-started: the table is keyed by pair)
              if (((const int*)(E + Y.mprw))[4 * i] == p) { nw[0] = E[Y.mprw + 4 * i + 1]; nw[1] = E[Y.mprw + 4 * i + 2]; nw[2] = E[Y.mprw + 4 * i + 3]; have_nw = true; }
            }
            bool pen;
            if constexpr (HF) pen = mpr_penetration_wl<true>(o1, o2, MPR_TOL, 60, &depth, dir, pos, &nsup, have_nw ? nw : nullptr, E + Y.cJ + 12 * lane);
            else pen = mpr_penetration(o1, o2, MPR_TOL, 60, &depth, dir, pos, &nsup, have_nw ? nw : nullptr);
            if (pen) {
              dist = margin - depth;
              normalize3(dir);
              mpr_hit = !prism; mpr_n[0] = dir[0]; mpr_n[1] = dir[1]; mpr_n[2] = dir[2];
              float dw[3], pw[3];
              matvec(dw, R1, dir);
              matvec(pw, R1, pos);
#pragma unroll
              for (int k = 0; k < 3; k++) { cpos[k] = pw[k] + x1[k] + (prism ? cen[k] : 0.f); nrm[k] = dw[k]; }
              hit = true;
            }
                      } else {   // all other kernels: the code exactly as it was before the height-field variant existed (register allocation of the hand kernel is sensitive to it)
            const float zero3[3] = {0.f, 0.f, 0.f};
            nsup = 0;
            // MPR in geom1's own frame: obj1 needs no rotation / translation at all (identity frame), obj2 carries the
            // relative pose R1^T R2, R1^T (x2 - x1); normal and position are rotated back afterwards
            float R1[9];
            geom_world_mat(W, Y, E, g1, R1);
            CObj o1, o2;
            {
              float R2[9], rel[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
              geom_world_mat(W, Y, E, g2, R2);
#pragma unroll
              for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) o2.mat[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j];
              matTvec(o2.pos, R1, rel);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) o1.mat[k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 3; k++) o1.pos[k] = 0.f;
            if constexpr (TRK) {
              cobj_shape_poly(o1, Q.t1, sz1);
              cobj_shape_poly(o2, Q.t2, sz2);
            }
            else { cobj_shape(o1, Q.t1, sz1); cobj_shape(o2, Q.t2, sz2); }
            o1.margin = o2.margin = 0.5f * margin;
            float depth, dir[3], pos[3], nw[3] = {0.f, 0.f, 0.f};
            bool have_nw = false;
            for (int i = 0; i < n_mprw; i++) {
              if (((const int*)(E + Y.mprw))[4 * i] == p) { nw[0] = E[Y.mprw + 4 * i + 1]; nw[1] = E[Y.mprw + 4 * i + 2]; nw[2] = E[Y.mprw + 4 * i + 3]; have_nw = true; }
            }
            // portal witnesses in per-lane LDS scratch: the contact-jacobian area of region X, not written before the rows stage
            // (TRK: a 1e-6 tolerance -- MuJoCo's ccd default -- was measured: narrow phase -15 %, but the one-substep qpos error p50 grows 2.7e-6 -> 1.5e-5)
            if (mpr_penetration_wl<TRK ? 2 : 0>(o1, o2, MPR_TOL, 60, &depth, dir, pos, &nsup, have_nw ? nw : nullptr, E + Y.cJ + 12 * lane, MT)) {
              dist = margin - depth;
              normalize3(dir);
              mpr_hit = true; mpr_n[0] = dir[0]; mpr_n[1] = dir[1]; mpr_n[2] = dir[2];
              float dw[3], pw[3], R1b[9];
              geom_world_mat(W, Y, E, g1, R1b);   // recomputed (9 LDS reads + a 3x3 product) instead of kept live across the portal refinement
              matvec(dw, R1b, dir);
              matvec(pw, R1b, pos);
#pragma unroll
              for (int k = 0; k < 3; k++) { cpos[k] = pw[k] + x1[k]; nrm[k] = dw[k]; }
              hit = true;
            }
                      }
          }
          if (hit && !(dist < margin - Q.gap)) hit = false;
          if (hit2 && !(dist2 < margin - Q.gap)) hit2 = false;
        }
        {  // slowest lane of this round: MPR lanes cost ~8 + refinement steps, analytic pairs ~1
          int w = nsup + 8;
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x4E, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x141, 0xf, 0xf, true));
          w = max(w, __builtin_amdgcn_update_dpp(0, w, 0x140, 0xf, 0xf, true));
          f_mpr += max(max(rdlanei(w, 0), rdlanei(w, 16)), max(rdlanei(w, 32), rdlanei(w, 48)));
        }
        {  // rebuild the warm-start table from this round's MPR contacts (all lookups of the round are done)
          SYNC();
          unsigned long long wb = __ballot(mpr_hit);
          int wpos = n_mprw_new + __popcll(wb & ((1ull << lane) - 1ull));
          if (mpr_hit && wpos < MPRW) {
            ((int*)(E + Y.mprw))[4 * wpos] = p;
            E[Y.mprw + 4 * wpos + 1] = mpr_n[0]; E[Y.mprw + 4 * wpos + 2] = mpr_n[1]; E[Y.mprw + 4 * wpos + 3] = mpr_n[2];
          }
          n_mprw_new = min(n_mprw_new + (int)__popcll(wb), MPRW);
        }
        unsigned long long bal = __ballot(hit);
        int pos = ncon + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < NC) {
          E[Y.cdist + pos] = dist;
#pragma unroll
          for (int k = 0; k < 3; k++) { E[Y.cpos + 3 * pos + k] = cpos[k]; E[Y.cnrm + 3 * pos + k] = nrm[k]; }
          ((int*)(E + Y.cpair))[pos] = cword;
        } else if (hit && pos < nct) {
          float* g = ovf_env + (pos - NC) * ovf_row;
          g[0] = dist;
#pragma unroll
          for (int k = 0; k < 3; k++) { g[1 + k] = cpos[k]; g[4 + k] = nrm[k]; }
          ((int*)g)[7] = cword;
        }
        ncon += __popcll(bal);
        bal = FULL ? __ballot(hit2) : 0ull;
        if (bal) {
          pos = ncon + __popcll(bal & ((1ull << lane) - 1ull));
          if (hit2 && pos < NC) {
            E[Y.cdist + pos] = dist2;
#pragma unroll
            for (int k = 0; k < 3; k++) { E[Y.cpos + 3 * pos + k] = cpos2[k]; E[Y.cnrm + 3 * pos + k] = nrm[k]; }
            ((int*)(E + Y.cpair))[pos] = cword;
          } else if (hit2 && pos < nct) {
            float* g = ovf_env + (pos - NC) * ovf_row;
            g[0] = dist2;
#pragma unroll
            for (int k = 0; k < 3; k++) { g[1 + k] = cpos2[k]; g[4 + k] = nrm[k]; }
            ((int*)g)[7] = cword;
          }
          ncon += __popcll(bal);
        }
      }
      if (ncon > nct) { flags |= MYO_FLAG_CONTACT_OVERFLOW; ncon = nct; }
      n_mprw = n_mprw_new;
      SYNC();
    } else n_mprw = 0;
    STAMP(4);
    // ---------------------------------------------------------------- constraint rows (registers: lane = dof / lane = contact)
    const float warm_r = lane < nv ? ldstate<SCHED>(warm_row + lane) : 0.f;   // (its latency hides behind the row stage)
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
          lD = __builtin_amdgcn_rcpf(R);
        }
      }
    }
    float caref[NR], cD = 0.f, cmu = 0.f;
#pragma unroll
    for (int k = 0; k < NR; k++) caref[k] = 0.f;
    float cmut = 0.f, cD2 = 0.f;   // TRK: torsional coefficient and the weight of the two torsional rows (0 for a condim-3 contact)
    int ckc = 0;
    // TRK, more than 64 contacts: contact c in [64, 128) belongs to lane c - 64 ("second bank").  Its per-contact solver state -- what bank 0
    // keeps in the registers above -- lives in a block at the end of the contact's overflow row, is loaded where a stage needs it and stored
    // back; loops over (contact, dof slot) read a bank-1 contact's coefficients from that block instead of shuffling them out of registers.
    // Everything of it sits behind `bank1` (wave-uniform, ncon > 64): the common case pays a scalar test per stage.
    constexpr int ROWS = 8 + NJ * KC + CDW;      // offset of the state block in an overflow row
    enum { S_AREF = 0, S_D = NR, S_MU = NR + 1, S_MUT = NR + 2, S_D2 = NR + 3, S_KC = NR + 4, S_JAR = NR + 5, S_JV = 2 * NR + 5, S_FC = 3 * NR + 5, S_HC = 3 * NR + 9, S_SIG = 3 * NR + 16 };
    static_assert(!TRK || 3 * NR + 17 <= TRK_STATE, "state block of a second-bank contact");
    const bool bank1 = TRK && ncon > 64;
    auto st1 = [&](int c) -> float* { return ovf_env + (size_t)(c - NC) * ovf_row + ROWS; };
    const bool b1lane = bank1 && lane + 64 < ncon;   // this lane also owns a bank-1 contact
    int nefc_b1 = 0;
    // (generic lambda: instantiated once with LDS pointers and once with the HBM overflow row, so that each copy keeps its own address space)
    auto build_row = [&](const float* pdist, const float* ppos, const float* pnrm, const int* ppair, float* cJ, unsigned int* cdw) {
      const int cw_ = ppair[0], p = cw_ & 2047;
      const float4 pq0_ = W.pair_rec[4 * (size_t)(p)]; const int pw_ = __float_as_int(pq0_.x); const int P[6] = {pw_ & 255, (pw_ >> 8) & 255, (int)((unsigned int)cw_ >> 16), (cw_ >> 11) & 31, (pw_ >> 16) & 15, (pw_ >> 20) & 15};
      const float* F = M.pair_f + 12 * p;
      float n[3] = {pnrm[0], pnrm[1], pnrm[2]}, t1[3], t2[3];
      float cp[3] = {ppos[0], ppos[1], ppos[2]};
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
      float vn = 0, vt1 = 0, vt2 = 0, vs = 0;
      unsigned int dpk[CDW];
#pragma unroll
      for (int k = 0; k < CDW; k++) dpk[k] = 0;
      ckc = P[3];
#pragma unroll
      for (int k = 0; k < KC; k++) {
        float jn = 0, j1 = 0, j2 = 0, js = 0;
        int d = 0;
        if (k < ckc) {
          const int de_ = W.pair_dl_pk[P[2] + k];   // dof | hinge << 7 | sign << 8
          d = de_ & 127;
          float sg = (float)(de_ >> 8);
          const float* ax = E + Y.axis + 3 * d;
          float col[3];
          if (de_ & 128) {
            float r[3] = {cp[0] - E[Y.anchor + 3 * d], cp[1] - E[Y.anchor + 3 * d + 1], cp[2] - E[Y.anchor + 3 * d + 2]};
            cross3(col, ax, r);
          } else { col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2]; }
          jn = sg * dot3(n, col); j1 = sg * dot3(t1, col); j2 = sg * dot3(t2, col);
          float qv = E[Y.qvel + d];
          vn += jn * qv; vt1 += j1 * qv; vt2 += j2 * qv;
          if constexpr (TRK) { js = (de_ & 128) ? sg * dot3(n, ax) : 0.f; vs += js * qv; }   // relative angular velocity about the normal
        }
        cJ[k] = jn; cJ[KC + k] = j1; cJ[2 * KC + k] = j2;
        if constexpr (TRK) cJ[3 * KC + k] = js;
        dpk[k >> 2] |= (unsigned int)d << (8 * (k & 3));   // padded entries: zero jacobian, dof 0
      }
#pragma unroll
      for (int k = 0; k < CDW; k++) cdw[k] = dpk[k];
      float dist = pdist[0], incl = F[0] - F[1];
      cmu = F[2];
      float imp = impedance(F + 6, dist, incl), K, B;
      kbi(F[4], F[5], F[7], M.timestep, &K, &B);
      if (FULL && P[5] == 1) {
        // condim 1 (explicit <pair>): one frictionless row = four identical "pyramid" rows with mu = 0 and D/4 each
        cmu = 0.f;
        cD = 0.25f / fmaxf(MINVALF, (1 - imp) / imp * F[3]);
      } else {
        float R0 = fmaxf(MINVALF, (1 - imp) / imp * F[3] * (1 + cmu * cmu));
        cD = __builtin_amdgcn_rcpf(fmaxf(MINVALF, 2 * cmu * cmu * R0));
      }
      float pos = -K * imp * (dist - incl);
      caref[0] = -B * (vn + cmu * vt1) + pos; caref[1] = -B * (vn - cmu * vt1) + pos;
      caref[2] = -B * (vn + cmu * vt2) + pos; caref[3] = -B * (vn - cmu * vt2) + pos;
      if constexpr (TRK) {
        if (P[5] == 4) { cmut = F[11]; cD2 = cD; caref[4] = -B * (vn + cmut * vs) + pos; caref[5] = -B * (vn - cmut * vs) + pos; }
      }
    };
    if constexpr (KC == 8 && !FULL && !TRK) {
      // hand / finger kernels: jacobians with lane = (contact of the pass, dof slot), eight contacts per pass -- each lane one column of the
      // three rows (normal, two tangents) instead of one lane walking its contact's eight dofs -- then the row constants with lane = contact
      float vn_c = 0.f, vt1_c = 0.f, vt2_c = 0.f;
      const int rg = lane >> 3, rk = lane & 7;
      for (int c0 = 0; c0 < ncon; c0 += 8) {
        const int c = c0 + rg;
        float pv0 = 0.f, pv1 = 0.f, pv2 = 0.f;
        auto row_cols = [&](const float* ppos, const float* pnrm, const int* ppair, float* cJ, unsigned int* cdw) {
          const int cw_ = ppair[0];
          const int P[4] = {0, 0, (int)((unsigned int)cw_ >> 16), (cw_ >> 11) & 31};      // dof-list start and length travel with the contact: no pair-table read here
          const float n[3] = {pnrm[0], pnrm[1], pnrm[2]}, cp[3] = {ppos[0], ppos[1], ppos[2]};
          float t1[3], t2[3];
          make_frame(n, t1, t2);
          float jn = 0.f, j1 = 0.f, j2 = 0.f;
          int d = 0;
          if (rk < P[3]) {
            const int de_ = W.pair_dl_pk[P[2] + rk];   // dof | hinge << 7 | sign << 8
            d = de_ & 127;
            const float sg = (float)(de_ >> 8);
            const float* ax = E + Y.axis + 3 * d;
            float col[3];
            if (de_ & 128) {
              const float r[3] = {cp[0] - E[Y.anchor + 3 * d], cp[1] - E[Y.anchor + 3 * d + 1], cp[2] - E[Y.anchor + 3 * d + 2]};
              cross3(col, ax, r);
            } else { col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2]; }
            jn = sg * dot3(n, col); j1 = sg * dot3(t1, col); j2 = sg * dot3(t2, col);
            const float qv = E[Y.qvel + d];
            pv0 = jn * qv; pv1 = j1 * qv; pv2 = j2 * qv;
          }
          cJ[rk] = jn; cJ[KC + rk] = j1; cJ[2 * KC + rk] = j2;
          ((unsigned char*)cdw)[rk] = (unsigned char)d;   // padded entries: zero jacobian, dof 0
        };
        if (c < ncon) {
          if (c < NC) row_cols(E + Y.cpos + 3 * c, E + Y.cnrm + 3 * c, (const int*)(E + Y.cpair) + c, E + Y.cJ + c * NJ * KC, (unsigned int*)(E + Y.cdofs) + CDW * c);
          else { float* g = ovf_env + (c - NC) * ovf_row; row_cols(g + 1, g + 4, (const int*)g + 7, g + 8, (unsigned int*)(g + 8 + NJ * KC)); }
        }
#pragma unroll
        for (int m_ = 1; m_ < 8; m_ <<= 1) { pv0 += __shfl_xor(pv0, m_); pv1 += __shfl_xor(pv1, m_); pv2 += __shfl_xor(pv2, m_); }
        const int src = (8 * (lane - c0)) & 63;      // lane = contact c0 + j takes the sums of group j
        const float a0 = __shfl(pv0, src), a1 = __shfl(pv1, src), a2 = __shfl(pv2, src);
        if (lane >= c0 && lane < c0 + 8) { vn_c = a0; vt1_c = a1; vt2_c = a2; }
      }
      SYNC();
      if (lane < ncon) {
        const float* g = lane < NC ? nullptr : ovf_env + (lane - NC) * ovf_row;
        const int cw_ = lane < NC ? ((const int*)(E + Y.cpair))[lane] : ((const int*)g)[7], p = cw_ & 2047;
        const float dist = lane < NC ? E[Y.cdist + lane] : g[0];
        const float* F = M.pair_f + 12 * p;
        ckc = (cw_ >> 11) & 31;
        const float incl = F[0] - F[1];
        cmu = F[2];
        float imp = impedance(F + 6, dist, incl), K, B;
        kbi(F[4], F[5], F[7], M.timestep, &K, &B);
        const float R0 = fmaxf(MINVALF, (1 - imp) / imp * F[3] * (1 + cmu * cmu));
        cD = __builtin_amdgcn_rcpf(fmaxf(MINVALF, 2 * cmu * cmu * R0));
        const float pos = -K * imp * (dist - incl);
        caref[0] = -B * (vn_c + cmu * vt1_c) + pos; caref[1] = -B * (vn_c - cmu * vt1_c) + pos;
        caref[2] = -B * (vn_c + cmu * vt2_c) + pos; caref[3] = -B * (vn_c - cmu * vt2_c) + pos;
      }
    } else if constexpr (KC == 20) {
      // 36-dof kernels: the same idea with three contacts per pass (lane = (contact of the pass, one of its 20 dof slots)); the three or four
      // velocity sums of a contact meet in LDS atomics on the (still unused) search-vector scratch
      const int rg = lane / KC, rk = lane - rg * KC;
      float vn_c = 0.f, vt1_c = 0.f, vt2_c = 0.f, vs_c = 0.f;
      float vn_1 = 0.f, vt1_1 = 0.f, vt2_1 = 0.f, vs_1 = 0.f;   // (TRK: the lane's bank-1 contact)
      for (int c0 = 0; c0 < ncon; c0 += 3) {
        if (lane < 12) E[Y.xv + lane] = 0.f;
        SYNC();
        const int c = c0 + rg;
        auto row_cols = [&](const float* ppos, const float* pnrm, const int* ppair, float* cJ, unsigned int* cdw) {
          const int cw_ = ppair[0];
          const float4 pq0_ = W.pair_rec[4 * (size_t)(cw_ & 2047)]; const int pw_ = __float_as_int(pq0_.x); const int P[6] = {pw_ & 255, (pw_ >> 8) & 255, (int)((unsigned int)cw_ >> 16), (cw_ >> 11) & 31, (pw_ >> 16) & 15, (pw_ >> 20) & 15};
          const float n[3] = {pnrm[0], pnrm[1], pnrm[2]}, cp[3] = {ppos[0], ppos[1], ppos[2]};
          float t1[3], t2[3];
          make_frame(n, t1, t2);
          if (P[4] == 2) {   // plane - capsule: first tangent along the capsule axis (MuJoCo's frame for this pair type)
            const float* ax = E + Y.gax + 3 * P[1];
            float t = dot3(ax, n), y[3] = {ax[0] - t * n[0], ax[1] - t * n[1], ax[2] - t * n[2]};
            float yn = norm3(y);
            if (yn >= 0.5f) {
              float inv = 1.0f / yn;
              t1[0] = y[0] * inv; t1[1] = y[1] * inv; t1[2] = y[2] * inv;
              cross3(t2, n, t1);
            }
          }
          float jn = 0.f, j1 = 0.f, j2 = 0.f, js = 0.f;
          int d = 0;
          if (rk < P[3]) {
            const int de_ = W.pair_dl_pk[P[2] + rk];   // dof | hinge << 7 | sign << 8
            d = de_ & 127;
            const float sg = (float)(de_ >> 8);
            const float* ax = E + Y.axis + 3 * d;
            float col[3];
            const bool hinge = (de_ & 128) != 0;
            if (hinge) {
              const float r[3] = {cp[0] - E[Y.anchor + 3 * d], cp[1] - E[Y.anchor + 3 * d + 1], cp[2] - E[Y.anchor + 3 * d + 2]};
              cross3(col, ax, r);
            } else { col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2]; }
            jn = sg * dot3(n, col); j1 = sg * dot3(t1, col); j2 = sg * dot3(t2, col);
            const float qv = E[Y.qvel + d];
            atomicAdd(&E[Y.xv + 4 * rg], jn * qv); atomicAdd(&E[Y.xv + 4 * rg + 1], j1 * qv); atomicAdd(&E[Y.xv + 4 * rg + 2], j2 * qv);
            if constexpr (TRK) { js = hinge ? sg * dot3(n, ax) : 0.f; atomicAdd(&E[Y.xv + 4 * rg + 3], js * qv); }   // relative angular velocity about the normal
          }
          cJ[rk] = jn; cJ[KC + rk] = j1; cJ[2 * KC + rk] = j2;
          if constexpr (TRK) cJ[3 * KC + rk] = js;
          ((unsigned char*)cdw)[rk] = (unsigned char)d;   // padded entries: zero jacobian, dof 0
        };
        if (rg < 3 && c < ncon) {
          if (c < NC) row_cols(E + Y.cpos + 3 * c, E + Y.cnrm + 3 * c, (const int*)(E + Y.cpair) + c, E + Y.cJ + c * NJ * KC, (unsigned int*)(E + Y.cdofs) + CDW * c);
          else { float* g = ovf_env + (c - NC) * ovf_row; row_cols(g + 1, g + 4, (const int*)g + 7, g + 8, (unsigned int*)(g + 8 + NJ * KC)); }
        }
        SYNC();
        if (lane >= c0 && lane < c0 + 3) { const int g = lane - c0; vn_c = E[Y.xv + 4 * g]; vt1_c = E[Y.xv + 4 * g + 1]; vt2_c = E[Y.xv + 4 * g + 2]; vs_c = E[Y.xv + 4 * g + 3]; }
        if constexpr (TRK) {
          if (b1lane && lane + 64 >= c0 && lane + 64 < c0 + 3) { const int g = lane + 64 - c0; vn_1 = E[Y.xv + 4 * g]; vt1_1 = E[Y.xv + 4 * g + 1]; vt2_1 = E[Y.xv + 4 * g + 2]; vs_1 = E[Y.xv + 4 * g + 3]; }
        }
        SYNC();
      }
      // row constants of contact c from its velocity sums (lane = contact & 63)
      auto row_consts = [&](const int cc, const float vn_c, const float vt1_c, const float vt2_c, const float vs_c, float (&caref)[NR], float& cD, float& cmu, float& cmut,
                            float& cD2, int& ckc) {
        const int lane = cc;     // (the body below is the one-bank code: it indexes the contact tables with `lane`)
        const float* g = lane < NC ? nullptr : ovf_env + (lane - NC) * ovf_row;
        const int cw_ = lane < NC ? ((const int*)(E + Y.cpair))[lane] : ((const int*)g)[7], p = cw_ & 2047;
        const float dist = lane < NC ? E[Y.cdist + lane] : g[0];
        const float4 pq0_ = W.pair_rec[4 * (size_t)(p)]; const int pw_ = __float_as_int(pq0_.x); const int P[6] = {pw_ & 255, (pw_ >> 8) & 255, __float_as_int(pq0_.w), (pw_ >> 24) & 255, (pw_ >> 16) & 15, (pw_ >> 20) & 15};
        const float* F = M.pair_f + 12 * p;
        ckc = P[3];
        const float incl = F[0] - F[1];
        cmu = F[2];
        float imp = impedance(F + 6, dist, incl), K, B;
        kbi(F[4], F[5], F[7], M.timestep, &K, &B);
        if (P[5] == 1) {
          // condim 1 (explicit <pair>): one frictionless row = four identical "pyramid" rows with mu = 0 and D/4 each
          cmu = 0.f;
          cD = 0.25f / fmaxf(MINVALF, (1 - imp) / imp * F[3]);
        } else {
          float R0 = fmaxf(MINVALF, (1 - imp) / imp * F[3] * (1 + cmu * cmu));
          cD = __builtin_amdgcn_rcpf(fmaxf(MINVALF, 2 * cmu * cmu * R0));
        }
        const float pos = -K * imp * (dist - incl);
        caref[0] = -B * (vn_c + cmu * vt1_c) + pos; caref[1] = -B * (vn_c - cmu * vt1_c) + pos;
        caref[2] = -B * (vn_c + cmu * vt2_c) + pos; caref[3] = -B * (vn_c - cmu * vt2_c) + pos;
        if constexpr (TRK) {
          if (P[5] == 4) { cmut = F[11]; cD2 = cD; caref[4] = -B * (vn_c + cmut * vs_c) + pos; caref[5] = -B * (vn_c - cmut * vs_c) + pos; }
        }
      };
      if (lane < ncon) row_consts(lane, vn_c, vt1_c, vt2_c, vs_c, caref, cD, cmu, cmut, cD2, ckc);
      if constexpr (TRK) {
        float a1[NR], D1 = 0.f, mu1 = 0.f, mut1 = 0.f, D21 = 0.f;
        int kc1 = 0;
#pragma unroll
        for (int k = 0; k < NR; k++) a1[k] = 0.f;
        if (b1lane) {
          row_consts(lane + 64, vn_1, vt1_1, vt2_1, vs_1, a1, D1, mu1, mut1, D21, kc1);
          float* S = st1(lane + 64);
#pragma unroll
          for (int k = 0; k < NR; k++) S[S_AREF + k] = a1[k];
          S[S_D] = D1; S[S_MU] = mu1; S[S_MUT] = mut1; S[S_D2] = D21; ((int*)S)[S_KC] = kc1; ((int*)S)[S_SIG] = -1;
        }
        if (bank1) nefc_b1 = 2 * __popcll(__ballot(b1lane && D21 != 0.f)) - 3 * __popcll(__ballot(b1lane && mu1 == 0.f));
      }
    } else
    if (lane < ncon) {
      if (lane < NC) build_row(E + Y.cdist + lane, E + Y.cpos + 3 * lane, E + Y.cnrm + 3 * lane, (const int*)(E + Y.cpair) + lane, E + Y.cJ + lane * NJ * KC,
                               (unsigned int*)(E + Y.cdofs) + CDW * lane);
      else { float* g = ovf_env + (lane - NC) * ovf_row; build_row(g, g + 1, g + 4, (const int*)g + 7, g + 8, (unsigned int*)(g + 8 + NJ * KC)); }
    }
    // efc row count as MuJoCo reports it: 4 pyramid rows per condim-3 contact, 1 per frictionless (condim-1) contact
    int nefc = __popcll(__ballot(lsign != 0.f)) + 4 * ncon - 3 * __popcll(__ballot(lane < ncon && cmu == 0.f));
    // TRK: friction-loss rows (mj_instantiateFriction): lane = dof, J = e_dof, aref = -B qvel; force saturates at +-f outside |jar| < f / D
    float flf = 0.f, flD = 0.f, flaref = 0.f, fljar = 0.f, fljv = 0.f, flrf = 0.f;
    if constexpr (TRK) {
      nefc += 2 * __popcll(__ballot(lane < ncon && cD2 != 0.f));
      nefc += nefc_b1;
      if (lane < nv) {
        const float* FL = W.fl + 4 * lane;
        flf = FL[0];
        if (flf > 0.f) { flD = FL[1]; flaref = -FL[2] * E[Y.qvel + lane]; flrf = flf / flD; }
      }
      nefc += __popcll(__ballot(flf > 0.f));
    }
    const int ncon_real = ncon;
    if (has_tl) {
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
          const float tl_len = E[Y.tJp + ngt_ * maxnnz_ + gt], tl_vel = E[Y.tJp + ngt_ * maxnnz_ + ngt_ + gt];   // (LDS copies: not live in registers across collision)
          float margin = T[3], dlo = tl_len - T[1], dhi = T[2] - tl_len, dist = 0;
          if (dlo < margin && dlo <= dhi) { t_sign = 1; dist = dlo; }
          else if (dhi < margin) { t_sign = -1; dist = dhi; }
          if (t_sign != 0) {
            float imp = impedance(T + 6, dist, margin), K, B;
            float R = fmaxf(MINVALF, (1 - imp) / imp * T[11]);
            kbi(T[4], T[5], T[7], M.timestep, &K, &B);
            t_aref = -B * (t_sign * tl_vel) - K * imp * (dist - margin);
            t_D = 1.0f / R;
            tact = true;
          }
        }
      }
      unsigned long long bal = __ballot(tact);
      int slot = nt + __popcll(bal & ((1ull << lane) - 1ull));
      const int ntcap = min(nct, 64);     // tendon-limit rows share the 64 lanes with the contacts; slots beyond the LDS table live in the overflow rows like contacts
      if (tact && slot < ntcap) {
        float* const row = slot < NC ? nullptr : ovf_env + (slot - NC) * ovf_row;
        float* cJ = slot < NC ? E + Y.cJ + slot * NJ * KC : row + 8;
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
          if constexpr (TRK) cJ[3 * KC + k] = 0.f;
          dpk[k >> 2] |= (unsigned int)d << (8 * (k & 3));
        }
        unsigned int* const dw = slot < NC ? (unsigned int*)(E + Y.cdofs) + CDW * slot : (unsigned int*)(row + 8 + NJ * KC);
#pragma unroll
        for (int k = 0; k < CDW; k++) dw[k] = dpk[k];
        if (slot < NC) { E[Y.cdist + slot] = t_aref; E[Y.cpos + 3 * slot] = t_D; E[Y.cpos + 3 * slot + 1] = (float)kc; }
        else { row[0] = t_aref; row[1] = t_D; row[2] = (float)kc; }
      }
      nt += __popcll(bal);
      }
      // only a tendon-limit row that found no slot is a loss (a has_tl model with more than NC contacts used to be cut back to NC
      // and flagged even when no tendon limit was active)
      if (nt > min(nct, 64)) { flags |= MYO_FLAG_CONTACT_OVERFLOW; nt = min(nct, 64); }
      SYNC();
      if (lane >= ncon && lane < nt) {
        const float* const row = lane < NC ? nullptr : ovf_env + (lane - NC) * ovf_row;
        const float a = lane < NC ? E[Y.cdist + lane] : row[0];
        caref[0] = caref[1] = caref[2] = caref[3] = a;   // (TRK: cD2 stays 0 for these rows)
        cD = 0.25f * (lane < NC ? E[Y.cpos + 3 * lane] : row[1]); cmu = 0.f; ckc = (int)(lane < NC ? E[Y.cpos + 3 * lane + 1] : row[2]);
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
      eD = __builtin_amdgcn_rcpf(fmaxf(MINVALF, (1 - imp) / imp * F[14]));
    }
    nefc += neq;
    SYNC();
    // the mass matrix moves from the square buffer (about to be reused for the Hessian) to a packed copy that
    // aliases the now dead broad-phase scratch
    // ... and this lane's full (symmetric) row stays in registers for the solver: M * x is then NVT broadcast-multiply-adds with no LDS read or
    // address arithmetic (twice and a half per substep), and the Newton rows H = M + J^T D J start from it
    // (36-dof instantiations only: they run two waves per SIMD on a 256-register budget.  In the 24-dof kernels, at four waves per SIMD, the 24 extra
    // live registers spill -- 23 in the headline kernel -- so those keep reading the packed copy.)
#ifdef MYO_NO_MROW
    constexpr bool MROW = false;
#else
    constexpr bool MROW = NVT > 24 && !RK4;      // (the Runge-Kutta twins keep four stage derivatives per lane: no room for the row)
#endif
    float mrow[MROW ? NVT : 1];
    if constexpr (MROW) {
      const int ml = lane < nv ? lane : 0;
#pragma unroll
      for (int k = 0; k < NVT; k++) mrow[k] = lane < nv ? E[Y.sq + ml * (NVT + 1) + k] : 0.f;
    }
    if (lane < nv) {
      const int based = (lane * (lane + 1)) / 2;
#pragma unroll
      for (int k = 0; k < NVT; k++) if (k <= lane) E[Y.Mp + based + k] = E[Y.sq + lane * (NVT + 1) + k];
    }
    const float* Mp = E + Y.Mp;
    auto symv_reg = [&](float x, int lane) -> float {      // (lane by value: inside the Newton loop it is that loop's opaque copy, see there)
      if constexpr (MROW) {
        float s0 = 0.f, s1 = 0.f;      // two partial sums: half the dependent chain
#pragma unroll
        for (int k = 0; k < NVT; k += 2) { s0 = fmaf(mrow[k], rdlane(x, k), s0); if (k + 1 < NVT) s1 = fmaf(mrow[k + 1], rdlane(x, k + 1), s1); }
        return s0 + s1;
      } else return symv_lds<NVT>(Mp, x, lane, nv);
    };
    SYNC();
    STAMP(5);
    SUB0();
    // ---------------------------------------------------------------- solver: Newton iterations, then the Euler solve, sharing ONE
    // instance of the unrolled register Cholesky.  phase 0 = Newton, 1 = unconstrained (nefc == 0), 2 = Euler (implicit damping)
    float Ma = 0.f, grad = 0.f, qfc = 0.f, ljar = 0.f, ljv = 0.f, cost = 0.f, qaccE = 0.f, qacc = 0.f;
    float cjar[NR], cjv[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) { cjar[k] = 0.f; cjv[k] = 0.f; }
    int phase = nefc > 0 ? 0 : 1, iters = 0;
    if (phase == 0) {  // start from the warm start (MuJoCo also tries qacc_smooth; the minimiser is the same)
      qacc = warm_r;
      Ma = symv_reg(qacc, lane);
      ljar = lsign * qacc - laref;
      if constexpr (TRK) fljar = qacc - flaref;
      if (lane < nv) E[Y.xv + lane] = qacc;
      SYNC();
      auto row_jar = [&](const float* cJ, const unsigned int* cdw) {
        float an = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
          float xv = E[Y.xv + CDOFP(cdw, k)]; an += cJ[k] * xv; a1 += cJ[KC + k] * xv; a2 += cJ[2 * KC + k] * xv;
          if constexpr (TRK) a3 += cJ[3 * KC + k] * xv;
        }
        cjar[0] = an + cmu * a1 - caref[0]; cjar[1] = an - cmu * a1 - caref[1]; cjar[2] = an + cmu * a2 - caref[2]; cjar[3] = an - cmu * a2 - caref[3];
        if constexpr (TRK) { cjar[4] = an + cmut * a3 - caref[4]; cjar[5] = an - cmut * a3 - caref[5]; }
      };
      if (lane < ncon) {
        if (lane < NC) row_jar(E + Y.cJ + lane * NJ * KC, (const unsigned int*)(E + Y.cdofs) + CDW * lane);
        else { const float* g = ovf_env + (lane - NC) * ovf_row; row_jar(g + 8, (const unsigned int*)(g + 8 + NJ * KC)); }
      }
      if constexpr (TRK) {
        if (b1lane) {   // J * warm - aref of the lane's bank-1 contact, into its state block
          const float* g = ovf_env + (lane + 64 - NC) * ovf_row;
          const unsigned int* cdw = (const unsigned int*)(g + 8 + NJ * KC);
          float* S = st1(lane + 64);
          float an = 0, a1 = 0, a2 = 0, a3 = 0;
          for (int k = 0; k < KC; k++) { const float xv = E[Y.xv + CDOFP(cdw, k)]; an += g[8 + k] * xv; a1 += g[8 + KC + k] * xv; a2 += g[8 + 2 * KC + k] * xv; a3 += g[8 + 3 * KC + k] * xv; }
          const float mu = S[S_MU], mut = S[S_MUT];
          S[S_JAR] = an + mu * a1 - S[S_AREF]; S[S_JAR + 1] = an - mu * a1 - S[S_AREF + 1]; S[S_JAR + 2] = an + mu * a2 - S[S_AREF + 2]; S[S_JAR + 3] = an - mu * a2 - S[S_AREF + 3];
          S[S_JAR + 4] = an + mut * a3 - S[S_AREF + 4]; S[S_JAR + 5] = an - mut * a3 - S[S_AREF + 5];
        }
      }
      if (eact) ejar = E[Y.xv + ed1] + eJ2 * E[Y.xv + ed2] - earef;
    }
    SUB(6);
    bool first = true;
    int sig_prev = -1;
    while (true) {
      const int lane = wave_lane();   // opaque copy again: keeps the 24 per-lane symv addresses from being hoisted out of the loop and spilled
      float r[NVT], rhs, invd;
      bool refactor = true;
      if (phase == 0) {
        // forces of the active rows, J^T f (LDS atomics), cost, gradient; convergence test; then, only if the iteration goes on and
        // the active set differs from the one whose Hessian was factorised last, the Hessian blocks (LDS atomics)
        bool lact = lsign != 0.f && ljar < 0;
        float w0 = cjar[0] < 0 ? cD : 0.f, w1 = cjar[1] < 0 ? cD : 0.f, w2 = cjar[2] < 0 ? cD : 0.f, w3 = cjar[3] < 0 ? cD : 0.f;
        float f0 = -w0 * cjar[0], f1 = -w1 * cjar[1], f2 = -w2 * cjar[2], f3 = -w3 * cjar[3];
        float w4 = 0.f, w5 = 0.f, f4 = 0.f, f5 = 0.f, flforce = 0.f, flcost = 0.f;
        bool flquad = false;
        if constexpr (TRK) {
          w4 = cjar[4] < 0 ? cD2 : 0.f; w5 = cjar[5] < 0 ? cD2 : 0.f;
          f4 = -w4 * cjar[4]; f5 = -w5 * cjar[5];
          if (flf > 0.f) {
            if (fljar <= -flrf) { flforce = flf; flcost = flf * (-0.5f * flrf - fljar); }
            else if (fljar >= flrf) { flforce = -flf; flcost = flf * (-0.5f * flrf + fljar); }
            else { flforce = -flD * fljar; flcost = 0.5f * flD * fljar * fljar; flquad = true; }
          }
        }
        float cst_b1 = 0.f;      // TRK second bank: cost of the lane's bank-1 contact; its force / Hessian coefficients go to its state block
        bool sig_b1_changed = false;
        if constexpr (TRK) {
          if (b1lane) {
            float* S = st1(lane + 64);
            const float D = S[S_D], D2 = S[S_D2], mu = S[S_MU], mut = S[S_MUT];
            float w[NR], f[NR];
            int sg = 0;
#pragma unroll
            for (int k = 0; k < NR; k++) {
              const float jr = S[S_JAR + k];
              w[k] = jr < 0 ? (k < 4 ? D : D2) : 0.f;
              f[k] = -w[k] * jr;
              cst_b1 += 0.5f * w[k] * jr * jr;
              sg |= (w[k] != 0.f ? 2 : 0) << k;
            }
            S[S_FC] = f[0] + f[1] + f[2] + f[3] + f[4] + f[5]; S[S_FC + 1] = mu * (f[0] - f[1]); S[S_FC + 2] = mu * (f[2] - f[3]); S[S_FC + 3] = mut * (f[4] - f[5]);
            S[S_HC] = w[0] + w[1] + w[2] + w[3] + w[4] + w[5]; S[S_HC + 1] = mu * (w[0] - w[1]); S[S_HC + 2] = mu * (w[2] - w[3]);
            S[S_HC + 3] = mu * mu * (w[0] + w[1]); S[S_HC + 4] = mu * mu * (w[2] + w[3]); S[S_HC + 5] = mut * (w[4] - w[5]); S[S_HC + 6] = mut * mut * (w[4] + w[5]);
            sig_b1_changed = ((const int*)S)[S_SIG] != sg;
            ((int*)S)[S_SIG] = sg;
          }
        }
        if (lane < nv) E[Y.qfc + lane] = (lact ? -lsign * lD * ljar : 0.f) + flforce;
        SYNC();
        if constexpr (KC == 8 || TRK) {
          // J^T f with lane = (contact of the pass, dof slot), 64 / KC contacts per pass: the contact's three (four) force components come from
          // its own lane by shuffle, every lane adds one entry (one lane per contact walking its dofs was the longer chain)
          const float Fn = f0 + f1 + f2 + f3 + f4 + f5, Ft1 = cmu * (f0 - f1), Ft2 = cmu * (f2 - f3), Ft3 = cmut * (f4 - f5);
          constexpr int FG = 64 / KC;
          const int fg = lane / KC, fk = lane - fg * KC;
          for (int c0 = 0; c0 < ncon; c0 += FG) {
            const int c = c0 + fg;
            const bool on = fg < FG && c < ncon;
            const int cs = on ? c : 0;
            float sFn = __shfl(Fn, cs & 63), sF1 = __shfl(Ft1, cs & 63), sF2 = __shfl(Ft2, cs & 63), sF3 = TRK ? __shfl(Ft3, cs & 63) : 0.f;
            int kc = __shfl(ckc, cs & 63);
            if constexpr (TRK) {
              if (on && c >= 64) { const float* S = st1(c); sFn = S[S_FC]; sF1 = S[S_FC + 1]; sF2 = S[S_FC + 2]; sF3 = S[S_FC + 3]; kc = ((const int*)S)[S_KC]; }
            }
            if (on && fk < kc) {
              if (c < NC) {
                const float* cJ = E + Y.cJ + c * NJ * KC;
                atomicAdd(&E[Y.qfc + CDOF(E, Y, c, fk)], sFn * cJ[fk] + sF1 * cJ[KC + fk] + sF2 * cJ[2 * KC + fk] + (TRK ? sF3 * cJ[(NJ - 1) * KC + fk] : 0.f));
              } else {
                const float* cJ = ovf_env + (c - NC) * ovf_row + 8;
                const unsigned int* cdw = (const unsigned int*)(cJ + NJ * KC);
                atomicAdd(&E[Y.qfc + CDOFP(cdw, fk)], sFn * cJ[fk] + sF1 * cJ[KC + fk] + sF2 * cJ[2 * KC + fk] + (TRK ? sF3 * cJ[(NJ - 1) * KC + fk] : 0.f));
              }
            }
          }
        } else if (lane < ncon) {   // MyoLeg (<= 10 contacts of 20 dofs): one lane per contact measured 1 % faster than three contacts per pass
          float Fn = f0 + f1 + f2 + f3 + f4 + f5, Ft1 = cmu * (f0 - f1), Ft2 = cmu * (f2 - f3);
          if (lane < NC) {
            const float* cJ = E + Y.cJ + lane * NJ * KC;
            for (int k = 0; k < ckc; k++) atomicAdd(&E[Y.qfc + CDOF(E, Y, lane, k)], Fn * cJ[k] + Ft1 * cJ[KC + k] + Ft2 * cJ[2 * KC + k]);
          } else {
            const float* cJ = ovf_env + (lane - NC) * ovf_row + 8;
            const unsigned int* cdw = (const unsigned int*)(cJ + NJ * KC);
            for (int k = 0; k < ckc; k++) atomicAdd(&E[Y.qfc + CDOFP(cdw, k)], Fn * cJ[k] + Ft1 * cJ[KC + k] + Ft2 * cJ[2 * KC + k]);
          }
        }
        if (eact) { float f = -eD * ejar; atomicAdd(&E[Y.qfc + ed1], f); atomicAdd(&E[Y.qfc + ed2], eJ2 * f); }
        SYNC();
        qfc = lane < nv ? E[Y.qfc + lane] : 0.f;
        float cst = lact ? 0.5f * lD * ljar * ljar : 0.f;
        cst += 0.5f * eD * ejar * ejar;
        cst += 0.5f * (w0 * cjar[0] * cjar[0] + w1 * cjar[1] * cjar[1] + w2 * cjar[2] * cjar[2] + w3 * cjar[3] * cjar[3]);
        if constexpr (TRK) cst += 0.5f * (w4 * cjar[4] * cjar[4] + w5 * cjar[5] * cjar[5]) + flcost + cst_b1;
        cst += 0.5f * qacc * Ma - qacc * smooth;          // Gauss term up to a constant
        float newcost = wave_sum(cst);
        grad = Ma - smooth - qfc;
        if (!first) {
          const float scale = M.newton_scale;   // 1 / (meaninertia * nv), computed at load: a scalar load here instead of a VGPR live across every stage
          float improvement = scale * (cost - newcost);
          float gn = scale * sqrtf(wave_sum(grad * grad));
          // float32 round-off of the gradient's own terms: below it the iteration only chases noise (float32 oracle build: 2.7 -> 1.9
          // Newton iterations per substep with this test, the float64 build needs 1.9; solution unchanged)
          const float gterm = fabsf(Ma) + fabsf(smooth) + fabsf(qfc);
          float gnoise = GRAD_NOISE * scale * sqrtf(wave_sum(gterm * gterm));
          iters++;
          f_itcon += ncon;
          if (improvement < fmaxf(M.tolerance, NEWTON_NOISE * scale * fabsf(newcost)) || gn < fmaxf(M.tolerance, gnoise) || iters >= M.iterations) phase = 2;
        }
        cost = newcost;
        SUB(0);
        if (phase == 0) {
          // H = M + J^T D J depends on the state only through the set of active rows: same set as last time -> same factor
          const int sig = (lact ? 1 : 0) | (w0 != 0.f ? 2 : 0) | (w1 != 0.f ? 4 : 0) | (w2 != 0.f ? 8 : 0) | (w3 != 0.f ? 16 : 0) |
                          (TRK ? ((w4 != 0.f ? 32 : 0) | (w5 != 0.f ? 64 : 0) | (flquad ? 128 : 0)) : 0);
          refactor = first || __any(sig != sig_prev || sig_b1_changed);
          sig_prev = sig;
          if (refactor) {
            f_fact++;
            // the Hessian buffer starts as M (lower rows; identity rows for the padding lanes) plus the limit / friction-loss diagonal, and the contact
            // blocks are added on top: the factorisation then reads finished rows instead of combining two LDS reads and three selects per entry
            if (lane < NVT) {
              const int dd_ = lane < nv ? lane : 0;
              const int based_ = (dd_ * (dd_ + 1)) / 2;
              const float dg_ = (lane < nv) ? ((lact ? lD : 0.f) + (flquad ? flD : 0.f)) : 1.f;
#pragma unroll
              for (int k = 0; k < NVT; k++) {
                float mv_;
                if constexpr (MROW) mv_ = (k <= lane) ? mrow[k] : 0.f;
                else mv_ = (lane < nv && k <= lane) ? Mp[based_ + (k <= dd_ ? k : 0)] : 0.f;
                E[Y.sq + lane * (NVT + 1) + k] = mv_ + (k == lane ? dg_ : 0.f);
              }
            }
            float Wn = w0 + w1 + w2 + w3 + w4 + w5, A1 = cmu * (w0 - w1), A2 = cmu * (w2 - w3), B1 = cmu * cmu * (w0 + w1), B2 = cmu * cmu * (w2 + w3);
            const float A3 = cmut * (w4 - w5), B3 = cmut * cmut * (w4 + w5);
            SYNC();
            // 64 / KC contacts per pass, lane = (contact of the pass, row a of its kc x kc block): the lane folds the contact's weights into
            // its row (hv = n_b p_n + t1_b p_t + t2_b p_u [+ s_b p_s]) and walks the columns b; the entries of different contacts meet in
            // the LDS atomics.  (One contact per pass with lane = block entry spent 7 passes per contact on a 20-dof block, half of the
            // lanes above the diagonal: 25 % of the TrackEnv kernel.)
            {
              constexpr int HG = 64 / KC;
              const int hg = lane / KC, ha = lane - hg * KC;
              for (int c0 = 0; c0 < ncon; c0 += HG) {
                const int c = c0 + hg;
                const bool on = hg < HG && c < ncon;
                const int cs = on ? c : 0;
                float sW = __shfl(Wn, cs & 63), sA1 = __shfl(A1, cs & 63), sA2 = __shfl(A2, cs & 63), sB1 = __shfl(B1, cs & 63), sB2 = __shfl(B2, cs & 63);
                float sA3 = TRK ? __shfl(A3, cs & 63) : 0.f, sB3 = TRK ? __shfl(B3, cs & 63) : 0.f;
                int kc = __shfl(ckc, cs & 63);
                if constexpr (TRK) {
                  if (on && c >= 64) {
                    const float* S = st1(c);
                    sW = S[S_HC]; sA1 = S[S_HC + 1]; sA2 = S[S_HC + 2]; sB1 = S[S_HC + 3]; sB2 = S[S_HC + 4]; sA3 = S[S_HC + 5]; sB3 = S[S_HC + 6]; kc = ((const int*)S)[S_KC];
                  }
                }
                if (on && sW != 0.f && ha < kc) {
                  auto hrow = [&](const float* cJ, const unsigned int* cdw) {
                    const int da = CDOFP(cdw, ha);
                    const float na = cJ[ha], ta = cJ[KC + ha], ua = cJ[2 * KC + ha];
                    float pn = sW * na + sA1 * ta + sA2 * ua, ps = 0.f;
                    const float pt = sA1 * na + sB1 * ta, pu = sA2 * na + sB2 * ua;
                    if constexpr (TRK) { const float sa = cJ[3 * KC + ha]; pn += sA3 * sa; ps = sA3 * na + sB3 * sa; }
                    for (int hb0 = 0; hb0 < kc; hb0 += 4) {   // four columns at a time: all reads first, then the atomics back to back
                      float hv[4];
                      int db[4];
#pragma unroll
                      for (int u = 0; u < 4; u++) {
                        const int hb = min(hb0 + u, KC - 1);      // (measured: without the clamp the reads merge into wide LDS loads and the TRK assembly gets 20 % slower)
                        db[u] = (hb0 + u < kc) ? CDOFP(cdw, hb) : 0x7fffffff;
                        hv[u] = pn * cJ[hb] + pt * cJ[KC + hb] + pu * cJ[2 * KC + hb];
                        if constexpr (TRK) hv[u] += ps * cJ[3 * KC + hb];
                      }
#pragma unroll
                      for (int u = 0; u < 4; u++) if (da >= db[u]) atomicAdd(&E[Y.sq + da * (NVT + 1) + db[u]], hv[u]);
                    }
                  };
                  if (c < NC) hrow(E + Y.cJ + c * NJ * KC, (const unsigned int*)(E + Y.cdofs) + CDW * c);
                  else { const float* g = ovf_env + (c - NC) * ovf_row; hrow(g + 8, (const unsigned int*)(g + 8 + NJ * KC)); }
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
        SUB(1);
      }
      rhs = phase == 0 ? -grad : (phase == 1 ? smooth : smooth + qfc);
      float x;
#ifndef MYO_NO_TREE
#define MYO_NO_TREE 0
#endif
      if (SPEC != 0 && !RK4 && phase != 0 && !MYO_NO_TREE) {
        // unconstrained and Euler solves of the size-specialised instantiations: M (+ h D) factorised leaves first, tree-sparse
        constexpr int NVS = SpecTree<SPEC>::nv > 0 ? SpecTree<SPEC>::nv : 1;
        const bool act = lane < NVS;
        const int q = act ? NVS - 1 - lane : 0;                   // this lane's dof in leaves-first order
        const float dadd = phase == 2 ? h * M.dof_damping[q] : 0.f;
        const float rhs_p = act ? __shfl(rhs, q) : 0.f;
#pragma unroll
        for (int k = 0; k < NVT; k++) {
          const int qk = k < NVS ? NVS - 1 - k : 0;               // compile-time after unrolling; qk >= q where k <= lane
          const float mv = (act && k <= lane) ? Mp[(qk * (qk + 1)) / 2 + q] : 0.f;
          r[k] = act ? mv + (k == lane ? dadd : 0.f) : (k == lane ? 1.f : 0.f);
        }
        SYNC();
        invd = chol_rows_tree<NVT, SPEC>(r, lane);
        if (lane < NVT) {
#pragma unroll
          for (int k = 0; k < NVT; k++) E[Y.sq + lane * (NVT + 1) + k] = r[k];
          E[Y.sq + lane * (NVT + 1) + NVT] = invd;
        }
        SYNC();
        SUB(2);
        const float xp = chol_solve_rows<NVT>(r, invd, rhs_p, E + Y.sq, lane);
        x = __shfl(xp, act ? NVS - 1 - lane : lane);
      } else {
      if (refactor) {
        if (phase == 0) {          // Newton: the buffer already holds M + the diagonal terms + J^T D J (see the assembly above)
          const int ll_ = lane < NVT ? lane : 0;
#pragma unroll
          for (int k = 0; k < NVT; k++) r[k] = lane < NVT ? E[Y.sq + ll_ * (NVT + 1) + k] : 0.f;
        } else {                   // unconstrained / Euler solves of the instantiations without a tree-sparse path: M (+ h D)
          const int dd = lane < nv ? lane : 0;
          const int based = (dd * (dd + 1)) / 2;
          const float diag_add = (phase == 2 && !RK4) ? h * M.dof_damping[dd] : 0.f;
#pragma unroll
          for (int k = 0; k < NVT; k++) {
            float mv;
            if constexpr (MROW) mv = (k <= lane) ? mrow[k] : 0.f;                                      // lower row of M (zero for lanes >= nv)
            else mv = (lane < nv && k <= lane) ? Mp[based + (k <= dd ? k : 0)] : 0.f;
            r[k] = (lane < nv) ? mv + (k == lane ? diag_add : 0.f) : (k == lane ? 1.f : 0.f);
          }
        }
        SYNC();
        invd = chol_rows<NVT>(r, lane);
        if (lane < NVT) {
#pragma unroll
          for (int k = 0; k < NVT; k++) E[Y.sq + lane * (NVT + 1) + k] = r[k];
          E[Y.sq + lane * (NVT + 1) + NVT] = invd;      // 1 / D in the padding column, for the iterations that reuse the factor
        }
        SYNC();
      } else {
        // the factor of the previous iteration is still in LDS (row-major L): reload this lane's row
        const int ll = lane < NVT ? lane : 0;
#pragma unroll
        for (int k = 0; k < NVT; k++) r[k] = E[Y.sq + ll * (NVT + 1) + k];
        invd = E[Y.sq + ll * (NVT + 1) + NVT];
      }
      SUB(2);
      x = chol_solve_rows<NVT>(r, invd, rhs, E + Y.sq, lane);
      }
      SUB(3);
      if (phase == 1) { qacc = x; qfc = 0.f; phase = 2; continue; }
      if (phase == 2) { qaccE = x; break; }
      // ---- Newton: exact line search along x
      float search = lane < nv ? x : 0.f;
      float Mv = symv_reg(search, lane);
      ljv = lsign * search;
      if constexpr (TRK) fljv = search;
      if (lane < nv) E[Y.xv + lane] = search;
      SYNC();
      auto row_jv = [&](const float* cJ, const unsigned int* cdw) {
        float an = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
          float xv = E[Y.xv + CDOFP(cdw, k)]; an += cJ[k] * xv; a1 += cJ[KC + k] * xv; a2 += cJ[2 * KC + k] * xv;
          if constexpr (TRK) a3 += cJ[3 * KC + k] * xv;
        }
        cjv[0] = an + cmu * a1; cjv[1] = an - cmu * a1; cjv[2] = an + cmu * a2; cjv[3] = an - cmu * a2;
        if constexpr (TRK) { cjv[4] = an + cmut * a3; cjv[5] = an - cmut * a3; }
      };
      if (lane < ncon) {
        if (lane < NC) row_jv(E + Y.cJ + lane * NJ * KC, (const unsigned int*)(E + Y.cdofs) + CDW * lane);
        else { const float* g = ovf_env + (lane - NC) * ovf_row; row_jv(g + 8, (const unsigned int*)(g + 8 + NJ * KC)); }
      }
      if constexpr (TRK) {
        if (b1lane) {   // J * search of the lane's bank-1 contact, into its state block
          const float* g = ovf_env + (lane + 64 - NC) * ovf_row;
          const unsigned int* cdw = (const unsigned int*)(g + 8 + NJ * KC);
          float* S = st1(lane + 64);
          float an = 0, a1 = 0, a2 = 0, a3 = 0;
          for (int k = 0; k < KC; k++) { const float xv = E[Y.xv + CDOFP(cdw, k)]; an += g[8 + k] * xv; a1 += g[8 + KC + k] * xv; a2 += g[8 + 2 * KC + k] * xv; a3 += g[8 + 3 * KC + k] * xv; }
          const float mu = S[S_MU], mut = S[S_MUT];
          S[S_JV] = an + mu * a1; S[S_JV + 1] = an - mu * a1; S[S_JV + 2] = an + mu * a2; S[S_JV + 3] = an - mu * a2; S[S_JV + 4] = an + mut * a3; S[S_JV + 5] = an - mut * a3;
        }
      }
      if (eact) ejv = E[Y.xv + ed1] + eJ2 * E[Y.xv + ed2];
      float g1 = wave_sum(search * (Ma - smooth)), g2 = wave_sum(0.5f * search * Mv), sn = sqrtf(wave_sum(search * search));
      SUB(4);
      float alpha = 0, lo = 0, hi = -1, dlo = 0, d2lo = 0, dhi = 0, d2hi = 0, d1init = 0;
      bool ls_on = sn >= MINVALF;
      for (int lsit = -1; lsit < M.ls_iterations && ls_on; lsit++) {
        float a = (lsit < 0) ? 0.f : alpha;
        float p1 = 0, p2 = 0;
        if (lsign != 0.f) { float xx = ljar + a * ljv; if (xx < 0) { p1 += lD * xx * ljv; p2 += lD * ljv * ljv; } }
        p1 += eD * (ejar + a * ejv) * ejv; p2 += eD * ejv * ejv;
#pragma unroll
        for (int k = 0; k < 4; k++) { float xx = cjar[k] + a * cjv[k]; if (xx < 0) { p1 += cD * xx * cjv[k]; p2 += cD * cjv[k] * cjv[k]; } }
        if constexpr (TRK) {
#pragma unroll
          for (int k = 4; k < 6; k++) { float xx = cjar[k] + a * cjv[k]; if (xx < 0) { p1 += cD2 * xx * cjv[k]; p2 += cD2 * cjv[k] * cjv[k]; } }
          if (flf > 0.f) {
            const float xx = fljar + a * fljv;
            if (xx <= -flrf) p1 -= flf * fljv;
            else if (xx >= flrf) p1 += flf * fljv;
            else { p1 += flD * xx * fljv; p2 += flD * fljv * fljv; }
          }
          if (b1lane) {   // (state block re-read per evaluation: a handful of L2 hits on a path that exists for > 64 contacts only)
            const float* S = st1(lane + 64);
            const float D = S[S_D], D2 = S[S_D2];
#pragma unroll
            for (int k = 0; k < NR; k++) { const float jv = S[S_JV + k], xx = S[S_JAR + k] + a * jv, Dk = k < 4 ? D : D2; if (xx < 0) { p1 += Dk * xx * jv; p2 += Dk * jv * jv; } }
          }
        }
        const float sp1 = wave_sum(p1);
        float d1 = sp1 + g1 + 2 * a * g2;
        float d2 = wave_sum(p2) + 2 * g2;
        if (lsit < 0) {
          if (d1 >= 0 || d2 <= 0) { ls_on = false; alpha = 0; break; }
          dlo = d1; d2lo = d2; d1init = fabsf(d1);
          alpha = -d1 / d2;
          continue;
        }
        f_ls++;
        // stop when the slope is below MuJoCo's tolerance -- or below the float32 round-off of the terms that cancel in it: without
        // the second test the search chases noise (measured on the float32 oracle build: 4.6 -> 1.45 evaluations per search, the float64
        // build needs 1.6; solution unchanged)
        float gtol = fmaxf(fmaxf(M.tolerance * M.ls_tolerance * sn / M.newton_scale, LS_FLOOR * d1init), LS_NOISE * (fabsf(g1) + fabsf(2 * a * g2) + fabsf(sp1)));
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
      SUB(5);
      if (!(alpha > 0)) { phase = 2; continue; }   // no descent left: keep qacc / qfc of this iterate
      qacc += alpha * search; Ma += alpha * Mv; ljar += alpha * ljv; ejar += alpha * ejv;
      if constexpr (TRK) fljar += alpha * fljv;
#pragma unroll
      for (int k = 0; k < NR; k++) cjar[k] += alpha * cjv[k];
      if constexpr (TRK) {
        if (b1lane) {
          float* S = st1(lane + 64);
#pragma unroll
          for (int k = 0; k < NR; k++) S[S_JAR + k] += alpha * S[S_JV + k];
        }
      }
    }
    STAMP(7);
    d_nefc = nefc; d_ncon = ncon_real; d_iter = max(d_iter, iters);
    f_ncon += ncon; f_iter += iters;
    {  // mj_checkAcc
      bool bad = lane < nv && (!(qacc == qacc) || fabsf(qacc) > MAXVALF);
      if (__any(bad) && alive) { flags |= MYO_FLAG_BAD_QACC; alive = false; }
    }
    if (lane < nv) warm_row[lane] = qacc;
    if constexpr (RK4) {
      if (alive) {
        const float Bw = (rk_stage == 0 || rk_stage == 3) ? (1.f / 6.f) : (1.f / 3.f), a = rk_stage < 2 ? 0.5f : 1.f;
        const bool last = rk_stage == 3;
        bool frot = false;
        if (rk_stage == 0) rk_t0 = time;
#pragma unroll
        for (int rr = 0; rr < NTR; rr++) {
          const int i = lane + 64 * rr;
          if (i < nu) {
            if (rk_stage == 0) { rk_a0[rr] = E[Y.act + i]; rk_sd[rr] = 0.f; }
            rk_sd[rr] += Bw * actdot[rr];
            E[Y.act + i] = rk_a0[rr] + h * (last ? rk_sd[rr] : a * actdot[rr]);
          }
        }
        float vint = 0.f;   // the velocity this lane's coordinate is advanced with, from X0, over h
        if (lane < nv) {
          const float vcur = E[Y.qvel + lane];
          const int fl = M.dof_link[lane];
          frot = has_free && W.link_free[fl] && lane - M.link_dofadr[fl] >= 3;
          if (rk_stage == 0) { rk_v0 = vcur; rk_sv = 0.f; rk_sa = 0.f; if (!frot) rk_q0 = E[Y.qpos + W.dof_qposadr[lane]]; }
          rk_sv += Bw * vcur; rk_sa += Bw * qaccE;
          vint = last ? rk_sv : a * vcur;
          E[Y.qvel + lane] = rk_v0 + h * (last ? rk_sa : a * qaccE);
          if (!frot) E[Y.qpos + W.dof_qposadr[lane]] = rk_q0 + h * vint;
          E[Y.xv + lane] = vint;
        }
        if (has_free) {
          SYNC();
          const int fl = lane < nv ? M.dof_link[lane] : 0;
          if (frot && lane - M.link_dofadr[fl] == 3) {
            const int qa = W.dof_qposadr[M.link_dofadr[fl]] + 3;
            if (rk_stage == 0) { rk_quat[0] = E[Y.qpos + qa]; rk_quat[1] = E[Y.qpos + qa + 1]; rk_quat[2] = E[Y.qpos + qa + 2]; rk_quat[3] = E[Y.qpos + qa + 3]; }
            float w[3] = {E[Y.xv + lane], E[Y.xv + lane + 1], E[Y.xv + lane + 2]};
            float wn = norm3(w), ang = h * wn;
            float o[4] = {rk_quat[0], rk_quat[1], rk_quat[2], rk_quat[3]};
            if (wn >= MINVALF) {
              float sn, cs;
              sincosf(0.5f * ang, &sn, &cs);
              const float inv = sn / wn, r[4] = {cs, w[0] * inv, w[1] * inv, w[2] * inv}, *q = rk_quat;
              o[0] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
              o[1] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
              o[2] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
              o[3] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
            }
            const float on = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            E[Y.qpos + qa] = o[0] * on; E[Y.qpos + qa + 1] = o[1] * on; E[Y.qpos + qa + 2] = o[2] * on; E[Y.qpos + qa + 3] = o[3] * on;
          }
        }
        time = uniformf(rk_t0 + (last ? h : a * h));
      }
      SYNC();
      if (alive && ++rk_stage < 4) goto rk_next_stage;
    } else
    if (alive) {
      bool frot = false;
      if (lane < nv) {
        float v = E[Y.qvel + lane] + h * qaccE;
        E[Y.qvel + lane] = v;
        // rotational dofs of a free joint (dofs 3..5 of its link) integrate through the quaternion below
        const int fl = M.dof_link[lane];
        frot = has_free && W.link_free[fl] && lane - M.link_dofadr[fl] >= 3;
        if (!frot) E[Y.qpos + W.dof_qposadr[lane]] += h * v;
      }
      if (has_free) {   // free joints: quaternion integrated with the body-frame angular velocity (mju_quatIntegrate); one lane per joint
        SYNC();
        const int fl = lane < nv ? M.dof_link[lane] : 0;
        if (frot && lane - M.link_dofadr[fl] == 3) {
          const int qa = W.dof_qposadr[M.link_dofadr[fl]] + 3;
          float w[3] = {E[Y.qvel + lane], E[Y.qvel + lane + 1], E[Y.qvel + lane + 2]};
          float wn = norm3(w), ang = h * wn;
          float q[4] = {E[Y.qpos + qa], E[Y.qpos + qa + 1], E[Y.qpos + qa + 2], E[Y.qpos + qa + 3]};
          if (wn >= MINVALF) {
            float sn, cs;
            sincosf(0.5f * ang, &sn, &cs);
            float inv = sn / wn, r[4] = {cs, w[0] * inv, w[1] * inv, w[2] * inv}, o[4];
            o[0] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
            o[1] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
            o[2] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
            o[3] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
            float on = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            E[Y.qpos + qa] = o[0] * on; E[Y.qpos + qa + 1] = o[1] * on; E[Y.qpos + qa + 2] = o[2] * on; E[Y.qpos + qa + 3] = o[3] * on;
          }
        }
      }
      time = uniformf(time + h);
    }
    SYNC();
    STAMP(8);
  }
  if (!SCHED && FULL && (kflags & KF_AUX)) return;   // observation-only launch: the state arrays are not touched
  if (!alive) {  // a bad env is reset like mj_resetData (mj_sim_scene.py:56-61)
    if (lane_id < nq) E[Y.qpos + lane_id] = M.qpos0[lane_id];
    if (lane_id < nv) { E[Y.qvel + lane_id] = 0; warm_row[lane_id] = 0.f; }
    for (int i = lane_id; i < nu; i += 64) { E[Y.act + i] = 0; E[Y.ctrl + i] = 0; }
    time = 0;
  }
  bool track_reset = false;
  if constexpr (TRK) {
    if (track_on) {   // epilogue of TrackEnv.step (mjx/myodm_v0.py:185-267, 297-304): reward / done / metrics on the stepped state, masked reset, observation
      SYNC();
      const DevTrack& K = *Bt.track;
      const float done = uniformf(track_reward(K, Bt, env, lane_id, E + Y.qpos, E + Y.qvel, E + Y.lpos, E + Y.lmat, M.origin, [](float v) { return wave_sum(v); }));
      const bool trunc = K.max_steps > 0 && !(done > 0.f) && Bt.elapsed[env] + 1 >= K.max_steps;   // gym TimeLimit of the registered ids
      if (lane_id == 0) Bt.solved[env] = trunc ? 1.f : 0.f;
      if (K.autoreset && (done > 0.f || trunc)) {   // (wave-uniform) TrackEnv.reset of this env: init_qpos, zero velocity / activation / control / warm start / time
        SYNC();
        if (lane_id < nq) E[Y.qpos + lane_id] = K.init_qpos[lane_id];
        if (lane_id < nv) { E[Y.qvel + lane_id] = 0.f; warm_row[lane_id] = 0.f; }
        for (int i = lane_id; i < nu; i += 64) { E[Y.act + i] = 0.f; E[Y.ctrl + i] = 0.f; }
        time = 0.f;
        track_reset = true;
      }
      SYNC();
      float* o = Bt.obs + (size_t)env * (nq + nv);
      if (lane_id < nq) o[lane_id] = E[Y.qpos + lane_id];
      if (lane_id < nv) o[nq + lane_id] = E[Y.qvel + lane_id];
    }
  }
  if (lane_id < nq) Bt.qpos[(size_t)env * nq + lane_id] = E[Y.qpos + lane_id];
  if (lane_id < nv) {
    Bt.qvel[(size_t)env * nv + lane_id] = E[Y.qvel + lane_id];
    Bt.qacc[(size_t)env * nv + lane_id] = warm_row[lane_id];   // (= the last substep's qacc; zero for an env that was just reset, like mj_resetData)
  }
  for (int i = lane_id; i < nu; i += 64) {
    Bt.act[(size_t)env * nu + i] = E[Y.act + i];
    Bt.ctrl[(size_t)env * nu + i] = E[Y.ctrl + i];
  }
  if (SCHED && s1 < nsubtot) {
    int* Wt = Bt.mprw + (size_t)env * 64;
    if (lane_id < 4 * n_mprw) Wt[lane_id] = ((const int*)(E + Y.mprw))[lane_id];
    if (lane_id == 63) Wt[63] = n_mprw;
  }
  if (lane_id == 0) {
    Bt.time[env] = time;
    if (s1 == nsubtot) Bt.elapsed[env] = track_reset ? 0 : Bt.elapsed[env] + 1;
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
  if (stamps && lane_id == 0) for (int k = 0; k < 12; k++) { stamps[(size_t)blockIdx.x * 12 + k] = st_acc[k]; stamps[((size_t)gridDim.x + blockIdx.x) * 12 + k] = st_sub[k]; }
#endif
}

#undef lane_id
#endif  // MYO_KERNEL_WAVE_H
