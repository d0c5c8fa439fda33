// On-device Broyden root-find of g(x) = f(x) - x, plus the small dense pieces around a solve (gfx950).
//
// Reference: broyden() in dirichlet/psignn/utilities/solver.py:116-207 ("good" Broyden with the
// inverse Jacobian kept as  B = -I + U V^T,  matvec/rmatvec at :96-114, ls=False so the step is 1).
//
// Storage: U, V are (thr, M) row-contiguous, M = N*d (the reference keeps U as (1,N,d,thr) with the
// rank index innermost, i.e. strided sweeps, and zero-fills 2*thr*M floats up front).
// Per iteration, with k stored pairs, dx = previous update, dg = g_new - g_old:
//   dots  pass : a_j = dx.U_j, c_j = V_j.dg, b_j = V_j.g_new            (U and V read once)
//   axpy  pass : vT = -dx + sum a_j V_j ;  D1 = dx + dg - sum c_j U_j ;  D2 = g_new - sum b_j U_j
//                                                                        (U and V read once more)
//   final      : u = D1 / (vT.dg), NaN->0 ;  update = D2 - u (vT.g_new)
// = 4 k M floats of compulsory traffic (the reference sweeps 6 k M).  All convergence bookkeeping
// (norms, traces, lowest iterate, the three stop tests of solver.py:176-183) lives in a device-side
// status block; the host only polls a done flag.  Reductions use fixed-shape partial sums ->
// bitwise reproducible run to run.
#include "vec_helpers.h"
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

struct Status {
  int32_t n_iter;       // iterations done
  int32_t done;
  int32_t prot_break;
  int32_t stop_reason;
  int32_t lowest_step;
  int32_t cur;          // buffer index of x_est
  int32_t low;          // buffer index of the lowest-residual iterate
  int32_t nxt;          // buffer index the next iterate is written to
  double lowest_rel, lowest_abs_at;  // lowest rel, and lowest abs (tracked independently, solver.py:170-175)
  double rel0;
  double abs0;          // first entry of the abs trace (protective break in stop_mode = "abs")
  int32_t lowest_step_abs;
  int32_t stop_abs;     // 0: stop_mode = "rel" (every reference call site), 1: "abs"
  double s, beta;       // vT.dg, vT.g
};

// ---- dot-product partials: one value per BLOCK and stored pair, written as coalesced rows -------------------------------------
// Round 2 had the lead lane of every wave store its wave sum of every pair straight to memory: one 4-byte store instruction per
// wave and pair, ~1 M scattered partial-line writes per sweep at 1M nodes.  Measured in isolation (scripts/ubench_sweep2.hip,
// profiles/r3_ubench_sweep2.txt): a sweep that streams at 0.78 of the HBM peak without them runs at 0.65 (4 floats per lane) /
// 0.71 (16 floats per lane) with them -- the wave reductions themselves cost nothing.  Now the lead lanes stash their wave sums
// in LDS, and after every 64 pairs (and after the last one) the block's first 64 threads add the four waves' values in a fixed
// order and store ONE contiguous row segment: part[plane][block * ldp + j].  Same bench: 0.76.  The reduce kernels read a
// column of that (blocks x pairs) matrix per pair -- 64-byte sectors from L2, a quarter as many elements as before.
#define PARTA_LD 32            // row pitch of the folded sweep's partials: at most U2R_KB_MAX + 1 = 25 kept pairs
// (block_sum_col, vec_helpers.h: RB = 1 024 threads per reduce block was tried for the column reads:
// k_reduce_cb 9.7 vs 8.3 us, k_reduce_a_check unchanged -- a block's time is the 64-byte sectors it pulls through ONE CU, not
// its load rounds; what helps is more blocks per column, RA below.)
#ifndef SWEEP_V_AHEAD
#define SWEEP_V_AHEAD 0   // 1: sweep 2 (16 floats per lane) requests the next pair's row before it works on the current one (A/B: slower)
#endif
#define RB 256
#define RA 8     // row chunks per column of the a-reduction: coef_a arrives as RA partial sums that sweep 2 adds up itself
struct psignn_broyden {
  const psignn_plan* plan = nullptr;
  int64_t M = 0;
  int seq_len = D;
  int thr = 0;
  int keep_trace = 0;
  int vec = 16;             // elements per thread of the vector kernels
  int nblk = 0, npart = 0;
  int vec_ax = 0, nblk_ax = 0;  // axpy / final pass: own vector width on mid-size vectors (see broyden_alloc)
  int jgroups = 1;          // block rows of the U/V sweeps (short vectors are also split over the stored pairs)
  float* jpart = nullptr;   // (jgroups, 3, M) partial axpy sums when jgroups > 1
  float *U = nullptr, *V = nullptr;
  float* xbuf = nullptr;    // (thr+2, M) with trace, else (3, M)
  // g = f(x) - x of two consecutive iterates: g of iterate i lives in gbuf[i & 1].  Round 2 kept g and dg = g_new - g_old as two
  // vectors that f had to read (g_old) and write (g_new, dg); every consumer of dg also reads g_new, so it now forms dg = g_new - g_old
  // itself from the two g buffers (the same fp32 subtraction): f neither reads g_old nor writes dg -- 80 MB less per 1M-node step
  float* gbuf[2] = {nullptr, nullptr};
  float *upd = nullptr, *fx = nullptr, *fwork = nullptr;
  float* nrm_part = nullptr;  // norm partials of the f / residual kernel: 2 * nn floats
  float* part2 = nullptr;     // three-sweep update: block partials of vT.dg, vT.g (2 * nblk floats)
  int uvu = 0;                // the update runs as three single-array sweeps U, V, U (broyden_alloc)
  int vec_u = 0, nblk_u = 0, npart_u = 0;   // their vector width / blocks / per-wave partials per stored pair
  int u2d_kmax = 0, nblk4 = 0, a_ready = 0; // k_sweep_u2d: up to this many stored pairs are all kept; its blocks; a of the next iteration comes (partly) from it
  int u2d_keep = 0, a_from = 0;             // beyond u2d_kmax: the most recent u2d_keep pairs are kept; pairs before a_from still need sweep 1
  int u2d_reg = 1;                          // kept values in registers (k_sweep_u2r<KB>, the default) or in per-thread LDS slots (k_sweep_u2d; PSIGNN_U2D_FORM=lds)
  float* parta = nullptr;                   // its per-block partials of a: (nblk4, PARTA_LD) -- entry q of a row = pair j_keep0 + q
  int nn_cap = 0;
  float *h0p = nullptr, *prbp = nullptr, *nrmp = nullptr;  // plan-order copies of h_initial, prb_data, normals
  float* part = nullptr;    // dot partials: 3 planes (a, c, b) of (blocks, ldp) -- one value per BLOCK and stored pair, pairs contiguous
                            // (pair_rows below); its head is reused for the axpy pass's s, beta partials
  int ldp = 0;              // row pitch of a plane: thr rounded up to 64
  int64_t pstride = 0;      // plane stride (floats): max blocks of any sweep form x ldp
  float* coef = nullptr;    // (3 + RA, thr): a, c, b; then the RA row-chunk sums of a (three-sweep forms: what sweep 2 reads)
  Status* st = nullptr;     // device
  double *rel_trace = nullptr, *abs_trace = nullptr;  // device, thr entries
  Status* h_st = nullptr;   // pinned host mirror
  size_t bytes = 0;
  int ext_iter = 0;
  int64_t ld = 0;           // row pitch (floats) of U and V
  int stop_abs = 0;         // stop_mode of the next solve
  int64_t size_hint = 0;    // elements of ALL vectors swept together (batched shard): picks the vector width / j-split
  int plan_order = 1;       // 0 while iterates are kept in the caller's numbering (adjoint solve on the gather kernels)
  // PSIGNN_GRAPH=1 (experiment): the launches of each poll_every-iteration chunk captured into a HIP
  // graph, cached per chunk and re-used by later solves with the same arguments
  std::vector<hipGraphExec_t> graphs;
  uint64_t graph_key = 0;
};

__global__ void k_init_status(Status* st, double* rel_trace, double* abs_trace, int thr, int stop_abs = 0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->abs0 = 0.0; st->lowest_step_abs = 0; st->stop_abs = stop_abs;
    st->n_iter = 0; st->done = 0; st->prot_break = 0; st->stop_reason = 0; st->lowest_step = 0;
    st->cur = 0; st->low = 0; st->nxt = 1;
    st->lowest_rel = 1e8; st->lowest_abs_at = 1e8; st->rel0 = 0.0; st->s = 0.0; st->beta = 0.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < thr; i += gridDim.x * blockDim.x) {
    rel_trace[i] = 0.0;
    abs_trace[i] = 0.0;
  }
}

// gx = fx0 - x0 ; upd = gx ; xbuf[0] = x0
template <int VEC>
__global__ __launch_bounds__(TB) void k_begin(int64_t M, const float* __restrict__ x0, const float* __restrict__ fx0,
                                              float* __restrict__ xb, float* __restrict__ gx, float* __restrict__ upd) {
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  float a[VEC], b[VEC];
  ldv<VEC>(x0, e0, M, a);
  ldv<VEC>(fx0, e0, M, b);
  stv<VEC>(xb, e0, M, a);
#pragma unroll
  for (int i = 0; i < VEC; ++i) b[i] -= a[i];
  stv<VEC>(gx, e0, M, b);
  stv<VEC>(upd, e0, M, b);
}

// x_next = x_cur + upd   (line_search with on=False: s = 1, solver.py:85-94)
template <int VEC>
__global__ __launch_bounds__(TB) void k_xnext(int64_t M, const Status* __restrict__ st, float* __restrict__ xb,
                                              const float* __restrict__ upd, float* __restrict__ copy_out) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  const float* xc = xb + (int64_t)st->cur * M;
  float* xn = xb + (int64_t)st->nxt * M;
  float a[VEC], b[VEC];
  ldv<VEC>(xc, e0, M, a);
  ldv<VEC>(upd, e0, M, b);
#pragma unroll
  for (int i = 0; i < VEC; ++i) a[i] += b[i];
  stv<VEC>(xn, e0, M, a);
  if (copy_out) stv<VEC>(copy_out, e0, M, a);
}

// line search (solver.py:61-94, ls=True): trial point x_cur + s * upd -> out; commit: the step becomes upd <- s * upd
template <int VEC>
__global__ __launch_bounds__(TB) void k_xtrial(int64_t M, const Status* __restrict__ st, const float* __restrict__ xb,
                                               float* __restrict__ upd, float s, float* __restrict__ out, int commit) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  const float* xc = xb + (int64_t)st->cur * M;
  float a[VEC], b[VEC];
  ldv<VEC>(xc, e0, M, a);
  ldv<VEC>(upd, e0, M, b);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    b[i] *= s;
    a[i] += b[i];
  }
  if (out) stv<VEC>(out, e0, M, a);
  if (commit) stv<VEC>(upd, e0, M, b);
}

// g_new = fx - x_next ; per-wave partials of |g_new|^2 and |fx|^2   (dg = g_new - g_old is formed by its consumers)
template <int VEC>
__global__ __launch_bounds__(TB) void k_resid(int64_t M, const Status* __restrict__ st, const float* __restrict__ xb,
                                              const float* __restrict__ fx, float* __restrict__ gnew,
                                              float* __restrict__ part, int npart) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  float sg = 0.f, sf = 0.f;
  if (e0 < M) {
    const float* xn = xb + (int64_t)st->nxt * M;
    float x[VEC], f[VEC];
    ldv<VEC>(xn, e0, M, x);
    ldv<VEC>(fx, e0, M, f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float gn = f[i] - x[i];
      sg = fmaf(gn, gn, sg);
      sf = fmaf(f[i], f[i], sf);
      x[i] = gn;
    }
    stv<VEC>(gnew, e0, M, x);
  }
  block_pair_store(sg, sf, part, npart);
}

// One block: finish the norms, append to the traces, track the lowest iterate, run the stop tests
// (solver.py:160-183) and rotate the iterate buffers.
// The bookkeeping block of an iteration.  Its time is latency, not work: round 2's form read and wrote the status fields one by
// one (a dozen dependent global round trips on thread 0) and ran the two norm reductions one after the other -- 7.7 us for a
// kernel whose empty launch costs 4.5.  Now the whole status block is read in one go BEFORE the reductions, both norms are summed in
// one pass (same thread-strided shape and tree per column as block_sum_col), everything is decided in registers and the block is
// written back once.
__device__ void check_block(Status* st, const float* __restrict__ part, int npart, double* __restrict__ rel_trace,
                            double* __restrict__ abs_trace, double eps, int thr, int seq_len, int keep_trace, double* sh) {
  __shared__ double sh2[RB];
  Status s = *st;   // (every thread: uniform address, one request; thread 0 is the only writer)
  if (s.done) return;
  double sg, sf;
  block_sum_col2(part, part + npart, npart, sh, sh2, sg, sf);
  if (threadIdx.x != 0) return;
  // torch.norm(...) is an fp32 value read back with .item(); the division is done in Python doubles
  double abs_diff = (double)(float)sqrt(sg);
  double rel_diff = abs_diff / ((double)(float)sqrt(sf) + 1e-9);
  const int n = s.n_iter + 1;
  s.n_iter = n;
  rel_trace[n - 1] = rel_diff;
  abs_trace[n - 1] = abs_diff;
  if (n == 1) {
    s.rel0 = rel_diff;
    s.abs0 = abs_diff;
  }
  // both modes keep their own lowest value / step; the lowest ITERATE follows stop_mode (solver.py:167-172)
  const bool stop_abs = s.stop_abs != 0;
  const bool low_rel = rel_diff < s.lowest_rel, low_abs = abs_diff < s.lowest_abs_at;
  if (low_rel) {
    s.lowest_rel = rel_diff;
    s.lowest_step = n;
  }
  if (low_abs) {
    s.lowest_abs_at = abs_diff;
    s.lowest_step_abs = n;
  }
  const bool new_low = stop_abs ? low_abs : low_rel;
  // buffer rotation: the iterate just evaluated lives in nxt
  const int cur = s.nxt;
  const int low = new_low ? cur : s.low;
  int nxt;
  if (keep_trace) {
    nxt = cur + 1;
  } else {
    nxt = 0;
    while (nxt == cur || nxt == low) ++nxt;
  }
  s.cur = cur;
  s.low = low;
  s.nxt = nxt;
  // stop tests on the objective of stop_mode (solver.py:174-181); protect_thres = 1e6 (abs) / 1e3 (rel) * seq_len
  const double obj = stop_abs ? abs_diff : rel_diff;
  const double* tr = stop_abs ? abs_trace : rel_trace;
  int reason = -1;
  if (obj < eps) {
    reason = 1;
  } else if (obj < 3 * eps && n > 30) {
    double mx = obj, mn = obj;   // (entry n - 1 of the trace is this iteration's value)
    for (int i = n - 30; i < n - 1; ++i) {
      double r = tr[i];
      mx = r > mx ? r : mx;
      mn = r < mn ? r : mn;
    }
    if (mx / mn < 1.3) reason = 2;
  }
  if (reason < 0 && obj > (stop_abs ? s.abs0 * 1e6 : s.rel0 * 1e3) * seq_len) {
    reason = 3;
    s.prot_break = 1;
  }
  if (reason < 0 && n >= thr) reason = 0;
  if (reason >= 0) {
    s.done = 1;
    s.stop_reason = reason;
  }
  *st = s;
}

#define DOTS_PART4 1   // (historic switch of the partials layout; the three-sweep forms below assume the current one)
// dots pass: per-block partials of a_j = dx.U_j, c_j = V_j.dg, b_j = V_j.g   for j < k
template <int VEC>
__device__ __forceinline__ void dots_body(int64_t M, int k, const Status* __restrict__ st,
                                             const float* __restrict__ U, const float* __restrict__ V,
                                             const float* __restrict__ dxv, const float* __restrict__ dgv,
                                             const float* __restrict__ gv, float* __restrict__ part, int64_t pstride, int ldp,
                                             int jstride, int64_t ld) {
  __shared__ PairStash<3> sh;
  if (__builtin_amdgcn_readfirstlane(st->done)) return;
  // blockIdx.y owns the stored pairs [j0, j1): short vectors (small meshes) give few blocks along x, so the
  // sweep is also split over j to cover the 256 CUs (each j still belongs to exactly one block row)
  const int j0 = blockIdx.y * jstride, j1 = min(k, j0 + jstride);
  if (j0 >= j1) return;
  int64_t e0 = elem0<VEC>();
  float dx[VEC], dg[VEC], g[VEC];
  bool act = e0 < M;
  if (act) {
    ldv<VEC>(dxv, e0, M, dx);
    ldv<VEC>(dgv, e0, M, dg);
    ldv<VEC>(gv, e0, M, g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dg[i] = g[i] - dg[i];   // dgv holds g of the previous iterate
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) dx[i] = dg[i] = g[i] = 0.f;
  }
  const int wv = threadIdx.x >> 6;
  const bool lead = (threadIdx.x & 63) == 0;
  float* rowb = part + (int64_t)blockIdx.x * ldp;
  for (int j = j0; j < j1; ++j) {
    float u[VEC], v[VEC];
    float sa = 0.f, sc = 0.f, sb = 0.f;
    if (act) {
      ldv_stream<VEC>(U + (int64_t)j * ld, e0, M, u);
      ldv_stream<VEC>(V + (int64_t)j * ld, e0, M, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sa = fmaf(dx[i], u[i], sa);
        sc = fmaf(v[i], dg[i], sc);
        sb = fmaf(v[i], g[i], sb);
      }
    }
    sa = wave_sum(sa);
    sc = wave_sum(sc);
    sb = wave_sum(sb);
    const int q = (j - j0) & 63;
    if (lead) {
      sh.v[0][wv][q] = sa;
      sh.v[1][wv][q] = sc;
      sh.v[2][wv][q] = sb;
    }
    if (q == 63 || j == j1 - 1) {
      float* const rows[3] = {rowb + (j - q), rowb + pstride + (j - q), rowb + 2 * pstride + (j - q)};
      stash_flush<3>(sh, q + 1, rows);
    }
  }
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_dots(int64_t M, int k, const Status* __restrict__ st,
                                             const float* __restrict__ U, const float* __restrict__ V,
                                             const float* __restrict__ dxv, const float* __restrict__ dgv,
                                             const float* __restrict__ gv, float* __restrict__ part, int64_t pstride, int ldp,
                                             int jstride, int64_t ld) {
  dots_body<VEC>(M, k, st, U, V, dxv, dgv, gv, part, pstride, ldp, jstride, ld);
}

// One launch after the dots pass, grid = (max(k, 1), 4):
//   blocks (j, c < 3), j < k: coef[c][j] = sum of the dots partials;
//   block (0, 3): the iteration's bookkeeping (check_block) from the norm partials of the f / residual kernel.
// The check used to be its own launch before the dots pass; running it here saves a launch per iteration.  The dots
// pass of the final iteration then runs once more than needed (its results are ignored: every later kernel sees done).
__device__ __forceinline__ void reduce_check_body(Status* st, const float* __restrict__ part, int nrows, int64_t pstride, int ldp, int thr, int k,
                                                     float* __restrict__ coef, const float* __restrict__ nrm_part, int nn,
                                                     double* __restrict__ rel_trace, double* __restrict__ abs_trace,
                                                     double eps, int seq_len, int keep_trace, double* sh) {
  if (blockIdx.y == 3) {
    if (blockIdx.x == 0) check_block(st, nrm_part, nn, rel_trace, abs_trace, eps, thr, seq_len, keep_trace, sh);
    return;
  }
  if (st->done) return;
  int j = blockIdx.x, c = blockIdx.y;
  if (j >= k) return;
  const float* col = part + (int64_t)c * pstride + j;   // column j of plane c: one value per block of the dots pass
  double s;
  if (c == 0) {   // a: summed as the three-sweep forms sum it -- RA row chunks, each rounded to float, added in order (k_reduce_a_check
    s = 0.0;      // + sweep 2) -- so that the two forms of the update stay bit-identical
    for (int r = 0; r < RA; ++r) {
      const int lo = (int)((int64_t)nrows * r / RA), hi = (int)((int64_t)nrows * (r + 1) / RA);
      s += (double)(float)block_sum_col(col + (int64_t)lo * ldp, hi - lo, ldp, sh);
    }
  } else {
    s = block_sum_col(col, nrows, ldp, sh);
  }
  if (threadIdx.x == 0) coef[c * thr + j] = (float)s;
}
__global__ __launch_bounds__(RB) void k_reduce_check(Status* st, const float* __restrict__ part, int nrows, int64_t pstride, int ldp, int thr, int k,
                                                     float* __restrict__ coef, const float* __restrict__ nrm_part, int nn,
                                                     double* __restrict__ rel_trace, double* __restrict__ abs_trace,
                                                     double eps, int seq_len, int keep_trace) {
  __shared__ double sh[RB];
  reduce_check_body(st, part, nrows, pstride, ldp, thr, k, coef, nrm_part, nn, rel_trace, abs_trace, eps, seq_len, keep_trace, sh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Three-sweep form of the update (long vectors; launch_update_uvu).  The two-pass form reads U and V twice per iteration:
// dots (a = U^T dx, c = V^T dg, b = V^T g), then axpy (vT = -dx + V a, D1 = dx + dg - U c, D2 = g - U b).  But V a needs only
// a, and U c / U b need only c and b: sweep 1 reads U for a; sweep 2 reads V ONCE for c, b AND V a; sweep 3 reads U for
// U c, U b -- three single-array sweeps instead of four, the same arithmetic on every element and the same partial-sum shapes
// (results bit-identical to the two-pass form; tests/test_gpu_parity.py::test_three_sweep_update_is_bitwise_identical).
template <int VEC>
__device__ __forceinline__ void sweep_u1_body(int64_t M, int k, const Status* __restrict__ st, const float* __restrict__ U,
                                              const float* __restrict__ dxv, float* __restrict__ part, int ldp, int64_t ld) {
  __shared__ PairStash<1> sh;
  if (__builtin_amdgcn_readfirstlane(st->done)) return;
  int64_t e0 = elem0<VEC>();
  float dx[VEC];
  const bool act = e0 < M;
  if (act) {
    ldv<VEC>(dxv, e0, M, dx);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) dx[i] = 0.f;
  }
  const int wv = threadIdx.x >> 6;
  const bool lead = (threadIdx.x & 63) == 0;
  float* rowb = part + (int64_t)blockIdx.x * ldp;      // plane 0 (a)
  for (int j = 0; j < k; ++j) {
    float u[VEC];
    float sa = 0.f;
    if (act) {
      ldv_stream<VEC>(U + (int64_t)j * ld, e0, M, u);
#pragma unroll
      for (int i = 0; i < VEC; ++i) sa = fmaf(dx[i], u[i], sa);
    }
    sa = wave_sum(sa);
    const int q = j & 63;
    if (lead) sh.v[0][wv][q] = sa;
    if (q == 63 || j == k - 1) {
      float* const rows[1] = {rowb + (j - q)};
      stash_flush<1>(sh, q + 1, rows);
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(TB) void k_sweep_u1(int64_t M, int k, const Status* __restrict__ st, const float* __restrict__ U,
                                                 const float* __restrict__ dxv, float* __restrict__ part, int ldp, int64_t ld) {
  sweep_u1_body<VEC>(M, k, st, U, dxv, part, ldp, ld);
}

// grid (max(k, 1), RA + 1): blocks (j, r < RA): row chunk r of column j of the a partials; block (0, RA): the iteration's bookkeeping (as k_reduce_check's)
__device__ __forceinline__ void reduce_a_check_body(Status* st, const float* __restrict__ part, int nrows, int ldp, int thr, int k,
                                                    float* __restrict__ coef, const float* __restrict__ nrm_part, int nn,
                                                    double* __restrict__ rel_trace, double* __restrict__ abs_trace,
                                                    double eps, int seq_len, int keep_trace, double* sh,
                                                    const float* __restrict__ parta = nullptr, int nrows4 = 0, int a_from = 1 << 30) {
  if (blockIdx.y == RA) {
    if (blockIdx.x == 0) check_block(st, nrm_part, nn, rel_trace, abs_trace, eps, thr, seq_len, keep_trace, sh);
    return;
  }
  const int done = st->done;   // (requested first, looked at last)
  const int j = blockIdx.x, r = blockIdx.y;
  if (j >= k) return;
  // pairs from a_from on: per-block partials of the folded sweep 3 of the last iteration (entry j - a_from of its rows); the older
  // pairs: sweep 1's (column j of plane 0).  A column of the folded sweep has one row per 1 024 vector elements (9 771 at 1M
  // nodes), every element a 64-byte sector of its own: RA blocks share it, sweep 2 adds their RA sums (coef + 3 thr + r thr + j).
  const float* col = j >= a_from ? parta + (j - a_from) : part + j;
  const int n = j >= a_from ? nrows4 : nrows;
  const int64_t stride = j >= a_from ? PARTA_LD : ldp;
  const int lo = (int)((int64_t)n * r / RA), hi = (int)((int64_t)n * (r + 1) / RA);
  const double s = block_sum_col(col + (int64_t)lo * stride, hi - lo, stride, sh);
  if (threadIdx.x == 0 && !done) coef[(3 + r) * thr + j] = (float)s;
}
__global__ __launch_bounds__(RB) void k_reduce_a_check(Status* st, const float* __restrict__ part, int nrows, int ldp, int thr, int k,
                                                       float* __restrict__ coef, const float* __restrict__ nrm_part, int nn,
                                                       double* __restrict__ rel_trace, double* __restrict__ abs_trace,
                                                       double eps, int seq_len, int keep_trace, const float* __restrict__ parta,
                                                       int nrows4, int a_from) {
  __shared__ double sh[RB];
  reduce_a_check_body(st, part, nrows, ldp, thr, k, coef, nrm_part, nn, rel_trace, abs_trace, eps, seq_len, keep_trace, sh, parta, nrows4, a_from);
}

// sweep 2: reads V once: partials of c_j = V_j.dg, b_j = V_j.g AND vT = -dx + sum_j a_j V_j; then vT's part of axpy_finish
// (vT.dg with the raw vT, NaN -> 0, vT.g; V[k] = vT; block partials into part2)
template <int VEC>
__device__ __forceinline__ void sweep_v_body(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ V,
                                             const float* __restrict__ dxv, const float* __restrict__ dgv,
                                             const float* __restrict__ gv, const float* __restrict__ coef,
                                             float* __restrict__ part, int64_t pstride, int ldp, float* __restrict__ part2, int nblk, int64_t ld,
                                             int thr) {
  __shared__ PairStash<2> sh;
  if (__builtin_amdgcn_readfirstlane(st->done)) return;
  int64_t e0 = elem0<VEC>();
  float dg[VEC], g[VEC], av[VEC];
  const bool act = e0 < M;
  if (act) {
    ldv<VEC>(dxv, e0, M, av);
    ldv<VEC>(dgv, e0, M, dg);
    ldv<VEC>(gv, e0, M, g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dg[i] = g[i] - dg[i];   // dgv holds g of the previous iterate
#pragma unroll
    for (int i = 0; i < VEC; ++i) av[i] = -av[i];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) dg[i] = g[i] = av[i] = 0.f;
  }
  const int wv = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const bool lead = lane == 0;
  float* rowb = part + (int64_t)blockIdx.x * ldp;      // planes 1 (c) and 2 (b)
  float cw = 0.f;                                       // coef_a of the current chunk of 64 pairs: lane q holds a_{jc + q}
  // One stored pair: the dot products and vT's term of V_j, whose values `v` the caller has requested already.
  auto step = [&](const int j, const float* v) {
    const int q = j & 63;
    // one vector load of 64 coefficients per chunk, handed out by v_readlane: in the batched kernels the table hangs off a
    // descriptor in memory, where a scalar per-pair read is a vector load with its full latency inside the loop
    if (q == 0) {
      double acc = 0.0;
      if (j + lane < k) {
#pragma unroll
        for (int r = 0; r < RA; ++r) acc += (double)coef[(3 + r) * thr + j + lane];   // the RA row-chunk sums of k_reduce_a_check, fixed order
      }
      cw = (float)acc;
    }
    float sc = 0.f, sb = 0.f;
    const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cw), q));
    if (act) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sc = fmaf(v[i], dg[i], sc);
        sb = fmaf(v[i], g[i], sb);
        av[i] = fmaf(ca, v[i], av[i]);
      }
    }
    sc = wave_sum(sc);
    sb = wave_sum(sb);
    if (lead) {
      sh.v[0][wv][q] = sc;
      sh.v[1][wv][q] = sb;
    }
    if (q == 63 || j == k - 1) {
      float* const rows[2] = {rowb + pstride + (j - q), rowb + 2 * pstride + (j - q)};
      stash_flush<2>(sh, q + 1, rows);
    }
  };
  if (SWEEP_V_AHEAD && VEC == 16) {
    // (experiment, off) Two pairs' rows in flight per wave: the sweep holds 96 VGPRs = five waves per SIMD with ONE 4 KB row per wave
    // in flight; requesting the row of pair j + 1 before pair j is worked on (two register buffers, loop unrolled by two) doubles the
    // bytes in flight per wave but costs a wave per SIMD (128 VGPRs).  Measured, interleaved on one box (profiles/r3_ab_sweepv.txt):
    // K = 20 94.4 -> 99.5 us (0.69 -> 0.65), K = 100 361 -> 384 us -- the fifth wave is worth more than the second row.
    float va[VEC], vb[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) va[i] = vb[i] = 0.f;
    if (act && k > 0) ldv_stream<VEC>(V, e0, M, va);
    int j = 0;
    for (; j + 1 < k; j += 2) {
      if (act) ldv_stream<VEC>(V + (int64_t)(j + 1) * ld, e0, M, vb);
      step(j, va);
      if (act && j + 2 < k) ldv_stream<VEC>(V + (int64_t)(j + 2) * ld, e0, M, va);
      step(j + 1, vb);
    }
    if (j < k) step(j, va);
  } else {
    for (int j = 0; j < k; ++j) {
      float v[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = 0.f;
      if (act) ldv_stream<VEC>(V + (int64_t)j * ld, e0, M, v);
      step(j, v);
    }
  }
  float p1 = 0.f, p2 = 0.f;
  if (act) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      p1 = fmaf(av[i], dg[i], p1);       // with the raw vT, as the reference divides before scrubbing
      av[i] = (av[i] != av[i]) ? 0.f : av[i];
      p2 = fmaf(av[i], g[i], p2);
    }
    stv<VEC>(V + (int64_t)k * ld, e0, M, av);
  }
  block_pair_store(p1, p2, part2, nblk);
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_sweep_v(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ V,
                                                const float* __restrict__ dxv, const float* __restrict__ dgv,
                                                const float* __restrict__ gv, const float* __restrict__ coef,
                                                float* __restrict__ part, int64_t pstride, int ldp, float* __restrict__ part2, int nblk,
                                                int64_t ld, int thr) {
  sweep_v_body<VEC>(M, k, st, V, dxv, dgv, gv, coef, part, pstride, ldp, part2, nblk, ld, thr);
}

// grid (max(k, 1), 3): blocks (j, 0 | 1): coef_c[j], coef_b[j]; block (0, 2): s = vT.dg, beta = vT.g from sweep 2's block partials
// (fixed order, fp64, rounded to fp32 like the reference's .item() values -- what every block of k_final used to repeat)
__device__ __forceinline__ void reduce_cb_body(Status* __restrict__ st, const float* __restrict__ part, int nrows, int64_t pstride, int ldp, int thr, int k,
                                               float* __restrict__ coef, const float* __restrict__ part2, int nblk, double* sh) {
  // (the done flag is requested first and looked at last: its round trip overlaps the partial loads of the reduction)
  const int done = st->done;
  if (blockIdx.y == 2) {
    if (blockIdx.x != 0) return;
    __shared__ double sh2[RB];
    double sv, beta;
    block_sum_col2(part2, part2 + nblk, nblk, sh, sh2, sv, beta);
    if (threadIdx.x == 0 && !done) {
      st->s = (double)(float)sv;
      st->beta = (double)(float)beta;
    }
    return;
  }
  const int j = blockIdx.x, c = 1 + blockIdx.y;
  if (j >= k) return;
  const double s = block_sum_col(part + (int64_t)c * pstride + j, nrows, ldp, sh);
  if (threadIdx.x == 0 && !done) coef[c * thr + j] = (float)s;
}
__global__ __launch_bounds__(RB) void k_reduce_cb(Status* __restrict__ st, const float* __restrict__ part, int nrows, int64_t pstride, int ldp, int thr, int k,
                                                  float* __restrict__ coef, const float* __restrict__ part2, int nblk) {
  __shared__ double sh[RB];
  reduce_cb_body(st, part, nrows, pstride, ldp, thr, k, coef, part2, nblk, sh);
}

// sweep 3: reads U once: D1 = dx + dg - sum_j c_j U_j, D2 = g - sum_j b_j U_j, and -- s and beta being known by now -- the
// final step of the update in the same registers: u = D1 / s (NaN -> 0) -> U[k], update = D2 - u * beta (k_final's arithmetic)
template <int VEC>
__device__ __forceinline__ void sweep_u2_body(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                              float* __restrict__ upd, const float* __restrict__ dgv,
                                              const float* __restrict__ gv, const float* __restrict__ coef, int thr, int64_t ld) {
  if (__builtin_amdgcn_readfirstlane(st->done)) return;
  int64_t e0 = elem0<VEC>();
  // (whole waves past the end leave; inside the last wave every lane stays for the coefficient hand-out below -- v_readlane reads
  // the lane's register whether the lane is active or not, but an inactive lane would never have loaded its coefficient)
  if (((int64_t)blockIdx.x * TB + (threadIdx.x & ~63)) * VEC >= M) return;
  const bool act = e0 < M;
  float a1[VEC], a2[VEC], dg[VEC];
  if (act) {
    ldv<VEC>(upd, e0, M, a1);
    ldv<VEC>(dgv, e0, M, dg);
    ldv<VEC>(gv, e0, M, a2);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dg[i] = a2[i] - dg[i];   // dgv holds g of the previous iterate
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) a1[i] = a2[i] = dg[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) a1[i] = a1[i] + dg[i];
  const int lane = threadIdx.x & 63;
  float cwc = 0.f, cwb = 0.f;   // c_j, b_j of the current chunk of 64 pairs, lane q <-> pair jc + q (one vector load each; see sweep_v_body)
  for (int j = 0; j < k; ++j) {
    float u[VEC];
    const int q = j & 63;
    if (q == 0) {
      cwc = (j + lane < k) ? coef[thr + j + lane] : 0.f;
      cwb = (j + lane < k) ? coef[2 * thr + j + lane] : 0.f;
    }
    const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cwc), q));
    const float cb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cwb), q));
    if (act) ldv_stream<VEC>(U + (int64_t)j * ld, e0, M, u);
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) u[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a1[i] = fmaf(-cc, u[i], a1[i]);
      a2[i] = fmaf(-cb, u[i], a2[i]);
    }
  }
  if (!act) return;
  const float sv = (float)st->s, beta = (float)st->beta;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float q = a1[i] / sv;
    q = (q != q) ? 0.f : q;
    a1[i] = q;
    a2[i] = fmaf(-q, beta, a2[i]);
  }
  stv<VEC>(U + (int64_t)k * ld, e0, M, a1);
  stv<VEC>(upd, e0, M, a2);
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_sweep_u2(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                                 float* __restrict__ upd, const float* __restrict__ dgv,
                                                 const float* __restrict__ gv, const float* __restrict__ coef, int thr, int64_t ld) {
  sweep_u2_body<VEC>(M, k, st, U, upd, dgv, gv, coef, thr, ld);
}

// Sweep 3 with the NEXT iteration's first sweep folded in (k <= U2D_KMAX stored pairs).  a_j(next) = U_j . update_new needs
// every U_j value twice: once to build update_new, once for the dot product with it.  For few stored pairs a thread can keep
// the U_j values it streams -- 4 floats per pair -- in a private LDS slot (256 threads x 16 k bytes per block: registers by
// another name, no sharing, no barrier) and take the dot products from there once its piece of update_new is finished:
// the following iteration then needs no sweep over U for a (V and U read once each while k is small -- all of a K = 20
// solve).  Direct dot products, exact; other partial-sum shapes than k_sweep_u1 (4 floats per lane here), so the last bits of
// a differ from the three-sweep form's.  Writes one row of per-block partials: part[block * PARTA_LD + (j - j_keep0)].
#ifndef U2D_UNROLL
#define U2D_UNROLL 8   // stored pairs whose loads are in flight together (one wave per SIMD has to keep the memory pipe busy alone)
#endif
__device__ __forceinline__ void sweep_u2d_body(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                               float* __restrict__ upd, const float* __restrict__ dgv,
                                               const float* __restrict__ gv, const float* __restrict__ coef, int thr,
                                               float* __restrict__ part, int64_t ld, int j_keep0) {
  // j_keep0: first stored pair that is kept (0 while k <= U2D_KMAX: all of them; later only the most recent ones -- the next
  // iteration's sweep 1 then covers the pairs before j_keep0 only)
  extern __shared__ __attribute__((aligned(16))) float4 keep[];   // keep[(j - j_keep0) * TB + tid] = this thread's 4 values of U_j
  if (st->done) return;
  const int64_t e0 = elem0<4>();
  const bool act = e0 + 4 <= M;          // (M = 10 N: a multiple of... not of 4 in general -> the ragged last quad goes the slow way)
  const bool tail = !act && e0 < M;
  float a1[4], a2[4];
  if (act || tail) {
    float dg[4];
    ldv<4>(upd, e0, M, a1);
    ldv<4>(dgv, e0, M, dg);
    ldv<4>(gv, e0, M, a2);
#pragma unroll
    for (int i = 0; i < 4; ++i) dg[i] = a2[i] - dg[i];   // dgv holds g of the previous iterate
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = a1[i] + dg[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = a2[i] = 0.f;
  }
  const int tid = threadIdx.x;
  int j = 0;
  if (act) {
    for (; j + U2D_UNROLL <= k; j += U2D_UNROLL) {
      float4 u[U2D_UNROLL];
#pragma unroll
      for (int q = 0; q < U2D_UNROLL; ++q) u[q] = *reinterpret_cast<const float4*>(U + (int64_t)(j + q) * ld + e0);
#pragma unroll
      for (int q = 0; q < U2D_UNROLL; ++q) {
        const float cc = coef[thr + j + q], cb = coef[2 * thr + j + q];
        if (j + q >= j_keep0) keep[(j + q - j_keep0) * TB + tid] = u[q];
        a1[0] = fmaf(-cc, u[q].x, a1[0]); a1[1] = fmaf(-cc, u[q].y, a1[1]); a1[2] = fmaf(-cc, u[q].z, a1[2]); a1[3] = fmaf(-cc, u[q].w, a1[3]);
        a2[0] = fmaf(-cb, u[q].x, a2[0]); a2[1] = fmaf(-cb, u[q].y, a2[1]); a2[2] = fmaf(-cb, u[q].z, a2[2]); a2[3] = fmaf(-cb, u[q].w, a2[3]);
      }
    }
  }
  for (; j < k; ++j) {
    float u[4] = {0.f, 0.f, 0.f, 0.f};
    const float cc = coef[thr + j], cb = coef[2 * thr + j];
    if (act || tail) ldv<4>(U + (int64_t)j * ld, e0, M, u);
    if (j >= j_keep0) keep[(j - j_keep0) * TB + tid] = make_float4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a1[i] = fmaf(-cc, u[i], a1[i]);
      a2[i] = fmaf(-cb, u[i], a2[i]);
    }
  }
  if (act || tail) {
    const float sv = (float)st->s, beta = (float)st->beta;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float q = a1[i] / sv;
      q = (q != q) ? 0.f : q;
      a1[i] = q;
      a2[i] = fmaf(-q, beta, a2[i]);
    }
    stv<4>(U + (int64_t)k * ld, e0, M, a1);
    stv<4>(upd, e0, M, a2);
    if (tail) {   // lanes past the end contribute nothing to the dot products
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (e0 + i >= M) a1[i] = a2[i] = 0.f;
    }
  }
  // a_j(next) = U_j . update_new for j <= k (U_k = a1, update_new = a2): wave sums stashed, one row of per-block values stored
  __shared__ float ua[TB / 64][PARTA_LD];
  const bool lead = (tid & 63) == 0;
  for (int jj = j_keep0; jj <= k; ++jj) {
    float4 u = jj < k ? keep[(jj - j_keep0) * TB + tid] : make_float4(a1[0], a1[1], a1[2], a1[3]);
    float sa = fmaf(u.x, a2[0], fmaf(u.y, a2[1], fmaf(u.z, a2[2], u.w * a2[3])));
    sa = wave_sum(sa);
    if (lead && jj - j_keep0 < PARTA_LD) ua[tid >> 6][jj - j_keep0] = sa;
  }
  __syncthreads();
  if (tid <= k - j_keep0 && tid < PARTA_LD) part[(int64_t)blockIdx.x * PARTA_LD + tid] = (ua[0][tid] + ua[1][tid]) + (ua[2][tid] + ua[3][tid]);
}
__global__ __launch_bounds__(TB) void k_sweep_u2d(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                                  float* __restrict__ upd, const float* __restrict__ dgv,
                                                  const float* __restrict__ gv, const float* __restrict__ coef, int thr,
                                                  float* __restrict__ part, int64_t ld, int j_keep0) {
  sweep_u2d_body(M, k, st, U, upd, dgv, gv, coef, thr, part, ld, j_keep0);
}

// The same folded sweep with the kept values in REGISTERS (round 3; the default).  The LDS form above pins ONE 256-thread block per
// CU once 16 k bytes per thread exceed 80 KB (k > 19): one wave per SIMD then has to cover the HBM latency alone, and the
// dominant kernel of the headline solve was its slowest sweep (0.58 - 0.61 of peak against 0.63 - 0.70 for k_sweep_v).  A fully
// unrolled array of KB float4 (KB = 8 / 16 / 24: 32 / 64 / 96 VGPRs) is addressed by compile-time indices only, so it stays in
// the register file: 8 / 5 / 4 waves per SIMD, no LDS, all KB loads of a thread in flight at once.  Pairs older than the kept
// window (j < j_keep0) are streamed as in the LDS form.  Same arithmetic in the same order on every element and the same
// partial-sum shapes as the LDS form => bit-identical iterates (tests/test_gpu_solver_forms.py).
// straight-line tail of the fast path for exactly NK kept pairs: NK loads issued back to back, NK fma groups, the update's last
// step, NK + 1 dot products.  Addresses are a wave-uniform row base (SGPR pair) plus ONE 32-bit lane offset shared by every load
// (global_load ... v_off, s[base]): a 64-bit VGPR address per load would cost 2 VGPRs and a 64-bit add each.
// scalar row base + 32-bit byte offset of the lane.  The empty asm pins the base in an SGPR pair and hides its provenance from the
// address reassociation passes (which otherwise chain 64-bit VGPR adds from one row to the next: 2 VGPRs + one VALU add per load);
// the explicit global address space keeps the access a global_load (an integer laundered through asm would become a flat access).
typedef float f4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_so(const float* __restrict__ base, uint32_t boff) {
  uint64_t rb = reinterpret_cast<uint64_t>(base);
  asm("" : "+s"(rb));
  const f4g t = *reinterpret_cast<const __attribute__((address_space(1))) f4g*>(rb + boff);
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_so(float* __restrict__ base, uint32_t boff, float4 v) {
  uint64_t rb = reinterpret_cast<uint64_t>(base);
  asm("" : "+s"(rb));
  f4g t = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<__attribute__((address_space(1))) f4g*>(rb + boff) = t;
}
__device__ __forceinline__ float rfl(float v) {   // wave-uniform value -> SGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <int NK, bool FIRST>
__device__ __forceinline__ void u2r_kept(float (&a1)[4], float (&a2)[4], int k, const Status* __restrict__ st, float* __restrict__ U,
                                         float* __restrict__ upd, const float* __restrict__ dgv, const float* __restrict__ gv,
                                         const float* __restrict__ coef, int thr, float* __restrict__ ua /* this wave's stash row (LDS) */,
                                         int64_t ld, int j_keep0, uint32_t off, bool lead) {
  // (instruction selection works per basic block: the 32-bit offset has to be (re)defined in the block of the accesses for
  // `sgpr base + zext(vgpr32)` to be matched as the saddr addressing mode)
  asm volatile("" : "+v"(off));
  float4 kp[NK > 0 ? NK : 1];
  const float* Ub = U + (int64_t)j_keep0 * ld;
#pragma unroll
  for (int q = 0; q < NK; ++q) kp[q] = ld4_so(Ub + (int64_t)q * ld, off);
  if (FIRST) {   // all pairs are kept (j_keep0 = 0): the three state vectors are requested in the same burst as the NK rows of U
    const float4 x = ld4_so(upd, off), go = ld4_so(dgv, off), g = ld4_so(gv, off);
    a1[0] = x.x + (g.x - go.x); a1[1] = x.y + (g.y - go.y); a1[2] = x.z + (g.z - go.z); a1[3] = x.w + (g.w - go.w);
    a2[0] = g.x; a2[1] = g.y; a2[2] = g.z; a2[3] = g.w;
  }
  // The 2 NK coefficients: lane q of the wave loads the pair of stored pair q (ONE vector load each for c and b), v_readlane hands
  // them to the fma's as scalars.  In the single-mesh kernel they could be scalar loads; in the batched kernel the coefficient
  // table hangs off a descriptor in memory, the compiler cannot prove it read-only and loads every coefficient into a VGPR of
  // its own -- 2 NK registers on top of the 4 NK kept ones.
  const int lane = threadIdx.x & 63;
  float ccv = 0.f, cbv = 0.f;
  if (lane < NK) {
    ccv = coef[thr + j_keep0 + lane];
    cbv = coef[2 * thr + j_keep0 + lane];
  }
  const float sv = rfl((float)st->s), beta = rfl((float)st->beta);
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ccv), q));
    const float cb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbv), q));
    a1[0] = fmaf(-cc, kp[q].x, a1[0]); a1[1] = fmaf(-cc, kp[q].y, a1[1]); a1[2] = fmaf(-cc, kp[q].z, a1[2]); a1[3] = fmaf(-cc, kp[q].w, a1[3]);
    a2[0] = fmaf(-cb, kp[q].x, a2[0]); a2[1] = fmaf(-cb, kp[q].y, a2[1]); a2[2] = fmaf(-cb, kp[q].z, a2[2]); a2[3] = fmaf(-cb, kp[q].w, a2[3]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float q = a1[i] / sv;
    q = (q != q) ? 0.f : q;
    a1[i] = q;
    a2[i] = fmaf(-q, beta, a2[i]);
  }
  st4_so(U + (int64_t)k * ld, off, make_float4(a1[0], a1[1], a1[2], a1[3]));
  st4_so(upd, off, make_float4(a2[0], a2[1], a2[2], a2[3]));
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    float sa = fmaf(kp[q].x, a2[0], fmaf(kp[q].y, a2[1], fmaf(kp[q].z, a2[2], kp[q].w * a2[3])));
    sa = wave_sum(sa);
    if (lead) ua[q] = sa;
  }
  float sa = fmaf(a1[0], a2[0], fmaf(a1[1], a2[1], fmaf(a1[2], a2[2], a1[3] * a2[3])));
  sa = wave_sum(sa);
  if (lead) ua[NK] = sa;
}

template <int KB>
__device__ __forceinline__ void sweep_u2r_body(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                               float* __restrict__ upd, const float* __restrict__ dgv,
                                               const float* __restrict__ gv, const float* __restrict__ coef, int thr,
                                               float* __restrict__ part, int64_t ld, int j_keep0) {
  __shared__ float ua[TB / 64][PARTA_LD];   // wave sums of a_j(next), j = j_keep0 + q: combined per block at the end
  if (__builtin_amdgcn_readfirstlane(st->done)) return;   // (a vector load in the batched kernel: keep the branch scalar)
  const int tid = threadIdx.x;
  const int64_t e0 = elem0<4>();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool lead = (tid & 63) == 0;
  // wave-uniform: the whole span of this wave (256 floats) lies inside the vector -> no per-lane bounds anywhere on the hot path
  const int64_t wave_e0 = ((int64_t)blockIdx.x * TB + (int64_t)__builtin_amdgcn_readfirstlane(tid & ~63)) * 4;
  if (wave_e0 + 256 <= M) {
    // byte offset of the lane, formed in 32-bit arithmetic (the host runs this form for M < 2^30 floats only): elem0<4>() * 4
    uint32_t off = ((blockIdx.x * (uint32_t)TB + ((uint32_t)tid & ~63u)) * 4u + ((uint32_t)tid & 63u) * 4u) * 4u;
    asm volatile("" : "+v"(off));   // opaque 32-bit VGPR: the accesses below are `global_load/store v_off, s[base]` (saddr form)
    float a1[4], a2[4];
#define U2R_CASE(n, first) case (n): u2r_kept<(n) < 0 ? 0 : (n), first>(a1, a2, k, st, U, upd, dgv, gv, coef, thr, ua[wv], ld, j_keep0, off, lead); break;
    if (j_keep0 == 0) {   // every stored pair is kept: one burst of k + 3 loads per thread, no wait in front of it
      switch (k) {
        U2R_CASE(KB - 8, true) U2R_CASE(KB - 7, true) U2R_CASE(KB - 6, true) U2R_CASE(KB - 5, true) U2R_CASE(KB - 4, true)
        U2R_CASE(KB - 3, true) U2R_CASE(KB - 2, true) U2R_CASE(KB - 1, true) U2R_CASE(KB, true)
        default: break;   // (the host only launches KB - 8 < count <= KB; KB = 8 also takes count = 0)
      }
    } else {
      {
        const float4 x = ld4_so(upd, off), go = ld4_so(dgv, off), g = ld4_so(gv, off);
        a1[0] = x.x + (g.x - go.x); a1[1] = x.y + (g.y - go.y); a1[2] = x.z + (g.z - go.z); a1[3] = x.w + (g.w - go.w);
        a2[0] = g.x; a2[1] = g.y; a2[2] = g.z; a2[3] = g.w;
      }
      // pairs before the kept window: streamed, eight pairs' loads in flight
      int j = 0;
      const int lane8 = tid & 7, half = (tid >> 3) & 1;
      for (; j + U2D_UNROLL <= j_keep0; j += U2D_UNROLL) {
        float4 u[U2D_UNROLL];
        const float* Uj = U + (int64_t)j * ld;
        asm volatile("" : "+v"(off));
        // the chunk's 16 coefficients in ONE vector load (lanes 0..7: c_j .. c_j+7, lanes 8..15: b_j ..), handed out by v_readlane: in
        // the batched kernel a per-pair scalar read is a vector load whose latency sits inside the loop (see u2r_kept)
        const float cv = coef[(1 + half) * thr + j + lane8];
#pragma unroll
        for (int q = 0; q < U2D_UNROLL; ++q) u[q] = ld4_so(Uj + (int64_t)q * ld, off);
#pragma unroll
        for (int q = 0; q < U2D_UNROLL; ++q) {
          const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), q));
          const float cb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), 8 + q));
          a1[0] = fmaf(-cc, u[q].x, a1[0]); a1[1] = fmaf(-cc, u[q].y, a1[1]); a1[2] = fmaf(-cc, u[q].z, a1[2]); a1[3] = fmaf(-cc, u[q].w, a1[3]);
          a2[0] = fmaf(-cb, u[q].x, a2[0]); a2[1] = fmaf(-cb, u[q].y, a2[1]); a2[2] = fmaf(-cb, u[q].z, a2[2]); a2[3] = fmaf(-cb, u[q].w, a2[3]);
        }
      }
      for (; j < j_keep0; ++j) {
        const float cc = rfl(coef[thr + j]), cb = rfl(coef[2 * thr + j]);
        asm volatile("" : "+v"(off));
        const float4 u = ld4_so(U + (int64_t)j * ld, off);
        a1[0] = fmaf(-cc, u.x, a1[0]); a1[1] = fmaf(-cc, u.y, a1[1]); a1[2] = fmaf(-cc, u.z, a1[2]); a1[3] = fmaf(-cc, u.w, a1[3]);
        a2[0] = fmaf(-cb, u.x, a2[0]); a2[1] = fmaf(-cb, u.y, a2[1]); a2[2] = fmaf(-cb, u.z, a2[2]); a2[3] = fmaf(-cb, u.w, a2[3]);
      }
      // the kept pairs: one straight-line instance per count (the count is a kernel argument: scalar jump)
      switch (k - j_keep0) {
        U2R_CASE(KB - 8, false) U2R_CASE(KB - 7, false) U2R_CASE(KB - 6, false) U2R_CASE(KB - 5, false) U2R_CASE(KB - 4, false)
        U2R_CASE(KB - 3, false) U2R_CASE(KB - 2, false) U2R_CASE(KB - 1, false) U2R_CASE(KB, false)
        default: break;
      }
    }
#undef U2R_CASE
  } else {
    // ---- the one ragged wave at the end of the vector (and lanes past it): per-lane bounds, kept pairs simply read twice --
    // same operations in the same order on every element, so the same bits as the fast path would give
    const bool in = e0 < M;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    if (in) {
      float dg[4];
      ldv<4>(upd, e0, M, a1);
      ldv<4>(dgv, e0, M, dg);
      ldv<4>(gv, e0, M, a2);
#pragma unroll
      for (int i = 0; i < 4; ++i) dg[i] = a2[i] - dg[i];   // dgv holds g of the previous iterate
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = a1[i] + dg[i];
    }
    for (int j = 0; j < k; ++j) {
      float u[4] = {0.f, 0.f, 0.f, 0.f};
      const float cc = coef[thr + j], cb = coef[2 * thr + j];
      if (in) ldv<4>(U + (int64_t)j * ld, e0, M, u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1[i] = fmaf(-cc, u[i], a1[i]);
        a2[i] = fmaf(-cb, u[i], a2[i]);
      }
    }
    if (in) {
      const float sv = (float)st->s, beta = (float)st->beta;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float q = a1[i] / sv;
        q = (q != q) ? 0.f : q;
        a1[i] = q;
        a2[i] = fmaf(-q, beta, a2[i]);
      }
      stv<4>(U + (int64_t)k * ld, e0, M, a1);
      stv<4>(upd, e0, M, a2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (e0 + i >= M) a1[i] = a2[i] = 0.f;   // lanes past the end contribute nothing to the dot products
    }
    for (int jj = j_keep0; jj <= k; ++jj) {
      float u[4] = {a1[0], a1[1], a1[2], a1[3]};
      if (jj < k) {
        u[0] = u[1] = u[2] = u[3] = 0.f;
        if (in) ldv<4>(U + (int64_t)jj * ld, e0, M, u);
      }
      float sa = fmaf(u[0], a2[0], fmaf(u[1], a2[1], fmaf(u[2], a2[2], u[3] * a2[3])));
      sa = wave_sum(sa);
      if (lead) ua[wv][jj - j_keep0] = sa;
    }
  }
  // one row of per-block values: entry q = a_{j_keep0 + q}(next), the four waves added in a fixed order
  __syncthreads();
  if (tid <= k - j_keep0) part[(int64_t)blockIdx.x * PARTA_LD + tid] = (ua[0][tid] + ua[1][tid]) + (ua[2][tid] + ua[3][tid]);
}
// waves per SIMD the register budget of each instantiation allows (512 / (4 KB + working set), MI355X_MICROARCH register table)
template <int KB> struct U2RWaves { static constexpr int value = KB <= 8 ? 8 : KB <= 16 ? 5 : 4; };
template <int KB>
__global__ __launch_bounds__(TB, U2RWaves<KB>::value) void k_sweep_u2r(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                                                      float* __restrict__ upd, const float* __restrict__ dgv,
                                                                      const float* __restrict__ gv, const float* __restrict__ coef, int thr,
                                                                      float* __restrict__ part, int64_t ld, int j_keep0) {
  sweep_u2r_body<KB>(M, k, st, U, upd, dgv, gv, coef, thr, part, ld, j_keep0);
}

// axpy pass.  Writes vT (NaN->0) to V[k], D1 to U[k] (unscaled), D2 to upd, partials of vT.dg, vT.g.
// gridDim.y == 1: the whole sum over j in one block column.  gridDim.y > 1 (short vectors): block row y sums its
// j-range into jpart[y][3][M] and k_axpy_combine finishes -- fixed grouping, so still reproducible.
template <int VEC>
__device__ __forceinline__ void axpy_finish(int64_t M, int k, int64_t e0, float* av, float* a1, float* a2, const float* dg,
                                            const float* g, float* __restrict__ U, float* __restrict__ V,
                                            float* __restrict__ upd, float& p1, float& p2, int64_t ld) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    p1 = fmaf(av[i], dg[i], p1);       // with the raw vT, as the reference divides before scrubbing
    av[i] = (av[i] != av[i]) ? 0.f : av[i];
    p2 = fmaf(av[i], g[i], p2);
  }
  stv<VEC>(V + (int64_t)k * ld, e0, M, av);
  stv<VEC>(U + (int64_t)k * ld, e0, M, a1);
  stv<VEC>(upd, e0, M, a2);
}

template <int VEC>
__device__ __forceinline__ void axpy_body(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                             float* __restrict__ V, float* __restrict__ upd /* in: dx, out: D2 */,
                                             const float* __restrict__ dgv, const float* __restrict__ gv,
                                             const float* __restrict__ coef, int thr, float* __restrict__ part, int npart,
                                             int jstride, float* __restrict__ jpart, int64_t ld, const bool split) {
  if (st->done) return;
  const int j0 = blockIdx.y * jstride, j1 = min(k, j0 + jstride);
  int64_t e0 = elem0<VEC>();
  float p1 = 0.f, p2 = 0.f;
  if (e0 < M) {
    float av[VEC], a1[VEC], a2[VEC], dg[VEC], g[VEC];
    if (!split) {
      ldv<VEC>(upd, e0, M, av);
      ldv<VEC>(dgv, e0, M, dg);
      ldv<VEC>(gv, e0, M, g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) dg[i] = g[i] - dg[i];   // dgv holds g of the previous iterate
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        a1[i] = av[i] + dg[i];
        a2[i] = g[i];
        av[i] = -av[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) av[i] = a1[i] = a2[i] = 0.f;
    }
    for (int j = j0; j < j1; ++j) {
      float u[VEC], v[VEC];
      float ca = coef[j], cc = coef[thr + j], cb = coef[2 * thr + j];
      ldv_stream<VEC>(U + (int64_t)j * ld, e0, M, u);
      ldv_stream<VEC>(V + (int64_t)j * ld, e0, M, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        av[i] = fmaf(ca, v[i], av[i]);
        a1[i] = fmaf(-cc, u[i], a1[i]);
        a2[i] = fmaf(-cb, u[i], a2[i]);
      }
    }
    if (split) {
      float* base = jpart + (int64_t)blockIdx.y * 3 * M;
      stv<VEC>(base, e0, M, av);
      stv<VEC>(base + M, e0, M, a1);
      stv<VEC>(base + 2 * M, e0, M, a2);
      return;
    }
    axpy_finish<VEC>(M, k, e0, av, a1, a2, dg, g, U, V, upd, p1, p2, ld);
  }
  if (split) return;
  block_pair_store(p1, p2, part, npart);
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_axpy(int64_t M, int k, const Status* __restrict__ st, float* __restrict__ U,
                                             float* __restrict__ V, float* __restrict__ upd /* in: dx, out: D2 */,
                                             const float* __restrict__ dgv, const float* __restrict__ gv,
                                             const float* __restrict__ coef, int thr, float* __restrict__ part, int npart,
                                             int jstride, float* __restrict__ jpart, int64_t ld) {
  axpy_body<VEC>(M, k, st, U, V, upd, dgv, gv, coef, thr, part, npart, jstride, jpart, ld, gridDim.y > 1);
}

// second half of a split axpy pass: init terms + the G partial sums, in group order
template <int VEC>
__device__ __forceinline__ void axpy_combine_body(int64_t M, int k, int G, const Status* __restrict__ st,
                                                     const float* __restrict__ jpart, float* __restrict__ U,
                                                     float* __restrict__ V, float* __restrict__ upd,
                                                     const float* __restrict__ dgv, const float* __restrict__ gv,
                                                     float* __restrict__ part, int npart, int64_t ld) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  float p1 = 0.f, p2 = 0.f;
  if (e0 < M) {
    float av[VEC], a1[VEC], a2[VEC], dg[VEC], g[VEC], t[VEC];
    ldv<VEC>(upd, e0, M, av);
    ldv<VEC>(dgv, e0, M, dg);
    ldv<VEC>(gv, e0, M, g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) dg[i] = g[i] - dg[i];   // dgv holds g of the previous iterate
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a1[i] = av[i] + dg[i];
      a2[i] = g[i];
      av[i] = -av[i];
    }
    for (int y = 0; y < G; ++y) {
      const float* base = jpart + (int64_t)y * 3 * M;
      ldv<VEC>(base, e0, M, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) av[i] += t[i];
      ldv<VEC>(base + M, e0, M, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) a1[i] += t[i];
      ldv<VEC>(base + 2 * M, e0, M, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) a2[i] += t[i];
    }
    axpy_finish<VEC>(M, k, e0, av, a1, a2, dg, g, U, V, upd, p1, p2, ld);
  }
  block_pair_store(p1, p2, part, npart);
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_axpy_combine(int64_t M, int k, int G, const Status* __restrict__ st,
                                                     const float* __restrict__ jpart, float* __restrict__ U,
                                                     float* __restrict__ V, float* __restrict__ upd,
                                                     const float* __restrict__ dgv, const float* __restrict__ gv,
                                                     float* __restrict__ part, int npart, int64_t ld) {
  axpy_combine_body<VEC>(M, k, G, st, jpart, U, V, upd, dgv, gv, part, npart, ld);
}

// u = D1 / s (NaN -> 0) -> U[k] ;  update = D2 - u * beta
template <int VEC>
__device__ __forceinline__ void final_body(int64_t M, int k, Status* __restrict__ st, float* __restrict__ U,
                                              float* __restrict__ upd, int64_t ld, const float* __restrict__ part,
                                              int npart, double* sh) {
  // every block first finishes s = vT.dg and beta = vT.g from the axpy pass's per-block partials (fixed order, fp64,
  // rounded to fp32 like the reference's .item() values) -- formerly a single-block launch of its own
  if (st->done) return;
  const float s = (float)block_sum_partials(part, npart, sh);
  const float beta = (float)block_sum_partials(part + npart, npart, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->s = (double)s;
    st->beta = (double)beta;
  }
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  float u[VEC], d2[VEC];
  float* Uk = U + (int64_t)k * ld;
  ldv<VEC>(Uk, e0, M, u);
  ldv<VEC>(upd, e0, M, d2);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float q = u[i] / s;
    q = (q != q) ? 0.f : q;
    u[i] = q;
    d2[i] = fmaf(-q, beta, d2[i]);
  }
  stv<VEC>(Uk, e0, M, u);
  stv<VEC>(upd, e0, M, d2);
}
template <int VEC>
__global__ __launch_bounds__(TB) void k_final(int64_t M, int k, Status* __restrict__ st, float* __restrict__ U,
                                              float* __restrict__ upd, int64_t ld, const float* __restrict__ part,
                                              int npart) {
  __shared__ double sh[TB];
  final_body<VEC>(M, k, st, U, upd, ld, part, npart, sh);
}

template <int VEC>
__global__ __launch_bounds__(TB) void k_copy_sel(int64_t M, const float* __restrict__ xb, const int32_t* __restrict__ sel,
                                                 int fixed, float* __restrict__ dst) {
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  int idx = sel ? *sel : fixed;
  float a[VEC];
  ldv<VEC>(xb + (int64_t)idx * M, e0, M, a);
  stv<VEC>(dst, e0, M, a);
}

// ------------------------------------------------------------------------------------------ host
// f kernels with a device-selected input buffer (fgnn.hip)
int psignn_f_eval_p(const psignn_plan_t* p, const float* W, int nl, const float* hbase, const int32_t* d_sel,
                    int64_t stride, const float* h0, const float* prb, const float* nrm, float* out, float* work,
                    hipStream_t st);

int psignn_f_tile_fused(const psignn_plan* p, const float* W, int nl, float* xbuf, int64_t M, const int32_t* st_words,
                        int off_done, int off_cur, int off_nxt, const float* upd, float* gnew,
                        const float* h0, const float* prb, const float* nrm, float* part, hipStream_t st);

#define U2R_KB_MAX 24
__global__ __launch_bounds__(TB) void kb_sweep_u2d(const BatchDesc* __restrict__ descs, int k, int j_keep0, int par);
// LDS form of the folded sweep: up to 38 pairs x 16 B x 256 threads of dynamic LDS; asked for once per device
static bool u2d_lds_attr(int dev) {
  static int state[64] = {0};   // 0 unknown, 1 granted, -1 refused
  if (dev < 0 || dev >= 64) return false;
  if (state[dev] == 0) {
    // (38 pairs x 16 B x 256 threads = 152 KB of dynamic LDS at most, next to the kernel's 512 B of static LDS: asking for the whole
    // 160 KB as dynamic is refused -- and the solver then silently ran with the 64 KB limits; tests/test_gpu_solver_forms.py caught it)
    const int want = 38 * TB * 16;
    const bool ok = hipFuncSetAttribute((const void*)k_sweep_u2d, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                    hipFuncSetAttribute((const void*)kb_sweep_u2d, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    state[dev] = ok ? 1 : -1;
  }
  return state[dev] > 0;
}

static int broyden_alloc(psignn_broyden* s) {
  size_t M = (size_t)s->M, thr = (size_t)s->thr;
  // 16 floats per lane from 768 K elements up, 4 below (PSIGNN_VEC16_MIN overrides).  Re-tuned after the coalesced lane
  // mapping, Broyden iterations/s with 4 vs 16 (dirichlet meshes, K = 100): 27 k nodes 12 790 / 10 156, 50 k 8 283 /
  // 7 826, 100 k 4 578 / 4 972, 200 k 2 608 / 2 806, 400 k 1 416 / 1 551 (the old switch point was 4 M elements)
  const int64_t vec16_min = [] {
    const char* e = getenv("PSIGNN_VEC16_MIN");
    return e ? (int64_t)atoll(e) : (int64_t)3 << 18;
  }();
  // A solver that will run inside a batched solve (size_hint = elements of the whole shard) sizes its sweeps for the
  // aggregate: the shard's blocks fill the chip together, so the long-vector width applies and the sweeps need no split
  // over the stored pairs.  Its single-mesh solves use the same shapes, hence the same bits as its batched ones.
  const int64_t eff = std::max<int64_t>(s->M, s->size_hint);
  s->vec = eff >= vec16_min ? 16 : 4;
  s->nblk = (int)cdiv(s->M, (int64_t)s->vec * TB);
  const int64_t eff_blk = cdiv(eff, (int64_t)s->vec * TB);
  s->jgroups = eff_blk >= 768 ? 1 : (int)std::min<int64_t>(8, cdiv(768, eff_blk));
  // Shard of short vectors (8 x 50 k nodes: 123 blocks per mesh, 983 together): the chip is covered, but every block then walks
  // all stored pairs alone; the dots pass split 4 ways over the pairs and the axpy pass unsplit at 4 floats per lane
  // (3 930 blocks) measured 305 / 310 us against 325 / 313 us per launch (K = 100; profiles/r2_batch_sweep_ab.txt)
  const bool short_shard = s->size_hint > s->M && s->vec == 16 && s->nblk < 512;
  if (short_shard) s->jgroups = 4;
  if (const char* e = getenv("PSIGNN_JGROUPS")) s->jgroups = std::max(1, std::min(8, atoi(e)));   // A/B knob
  s->npart = s->nblk * (TB / 64);
  // Mid-size vectors (16 floats per lane, but too few blocks to fill the chip): the dots pass is split over the stored
  // pairs (free: every pair's partial sums are independent), the axpy pass would need a combine launch after such a split
  // -- it runs unsplit with 4 floats per lane instead (100 k nodes: 91 us vs 79 + 20 us).  The thread -> element mapping
  // is per kernel; only the pair partials that k_final reads must follow the axpy pass's block count.
  s->vec_ax = s->vec;
  s->nblk_ax = s->nblk;
  const char* e_ax = getenv("PSIGNN_VEC_AX4");   // A/B knob: 1 forces the 4-float unsplit axpy pass, 0 forbids it
  if (e_ax ? (atoi(e_ax) != 0 && s->vec == 16)
           : (s->vec == 16 && s->jgroups > 1 && (short_shard || cdiv(s->M, (int64_t)4 * TB) >= 512))) {
    s->vec_ax = 4;
    s->nblk_ax = (int)cdiv(s->M, (int64_t)4 * TB);
  }
  // three-sweep update (launch_update): where an UNSPLIT sweep covers the chip -- long vectors at 16 floats per lane, mid-size
  // vectors and shards of short vectors at the 4-float width of their axpy pass.  PSIGNN_UVU=0|1 overrides (A/B, tests).
  s->uvu = 0;
  if (DOTS_PART4) {
    if (s->vec == 16 && s->jgroups == 1 && s->vec_ax == 16) {
      s->uvu = 1; s->vec_u = 16; s->nblk_u = s->nblk;
    } else if (s->vec_ax == 4 && s->vec == 16) {
      s->uvu = 1; s->vec_u = 4; s->nblk_u = s->nblk_ax;
    }
    if (const char* e = getenv("PSIGNN_UVU")) if (atoi(e) == 0) s->uvu = 0;
  }
  s->npart_u = s->nblk_u * (TB / 64);
  s->nblk4 = (int)cdiv(s->M, (int64_t)4 * TB);
  const bool fold_ok = s->uvu != 0;
  // folded sweep 3: kept values in registers (default; at most U2R_KB_MAX pairs) or in LDS (PSIGNN_U2D_FORM=lds: the round-2 form, A/B and tests)
  // Registers where the launch has several rounds of blocks (>= 2 048 blocks of 1 024 floats: 8 per CU); on shorter vectors every
  // block is resident at once, all waves then load together and reduce together, and the LDS form's two rounds overlap better
  // (100k nodes, K = 100: 56.5 vs 64.8 us per launch; profiles/r3_ab_u2d.txt).  M >= 2^30 floats: the register form's 32-bit lane
  // offsets do not reach.
  s->u2d_reg = cdiv(eff, (int64_t)4 * TB) >= 2048 && s->M < ((int64_t)1 << 30);
  if (const char* e = getenv("PSIGNN_U2D_FORM")) s->u2d_reg = ((e[0] == 'l' || e[0] == 'L') ? 0 : 1) && s->M < ((int64_t)1 << 30);
  const int kmax_cap = s->u2d_reg ? U2R_KB_MAX : 38;
  s->u2d_kmax = fold_ok ? 24 : 0;   // LDS form: 96 KB per block at most; 20 ... 32 measure alike at K = 50, K = 20 needs >= 19
  if (const char* e = getenv("PSIGNN_U2D_KMAX")) s->u2d_kmax = fold_ok ? std::max(0, std::min(kmax_cap, atoi(e))) : 0;
  s->u2d_keep = s->u2d_kmax > 0 ? 16 : 0;
  if (const char* e = getenv("PSIGNN_U2D_KEEP")) s->u2d_keep = s->u2d_kmax > 0 ? std::max(0, std::min(s->u2d_kmax, atoi(e))) : 0;
  if (s->u2d_kmax > 0 && !s->u2d_reg) {
    // more than 64 KB of dynamic LDS has to be asked for, per device (the attribute belongs to the device's code object)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!u2d_lds_attr(dev)) {   // not granted: stay within the default 64 KB (16 B x 256 threads x 15 pairs + the kernel's static 512 B)
      s->u2d_kmax = std::min(s->u2d_kmax, 15);
      s->u2d_keep = std::min(s->u2d_keep, 15);
    }
  }
  s->nn_cap = std::max<int>(std::max(s->nblk, s->nblk_ax), s->plan ? (int)s->plan->n_tiles : 0);
  s->ldp = (s->thr + 63) / 64 * 64;
  s->pstride = (int64_t)std::max(std::max(s->nblk, s->nblk_u), std::max(s->nblk_ax, 1)) * s->ldp;
  size_t nx = s->keep_trace ? thr + 2 : 3;
  s->ld = (s->M + 63) / 64 * 64;  // row pitch of U and V: every stored vector starts on a 256-byte boundary
  size_t ld = (size_t)s->ld;
  struct { void** p; size_t n; } allocs[] = {
      {(void**)&s->U, thr * ld * 4},  {(void**)&s->V, thr * ld * 4},   {(void**)&s->xbuf, nx * M * 4},
      {(void**)&s->gbuf[0], M * 4},   {(void**)&s->gbuf[1], M * 4},   {(void**)&s->upd, M * 4},
      {(void**)&s->fx, M * 4},        {(void**)&s->part, 3 * (size_t)s->pstride * 4 + 16}, {(void**)&s->parta, (size_t)s->nblk4 * PARTA_LD * 4 + 16},
      {(void**)&s->coef, (3 + RA) * thr * 4 + 16}, {(void**)&s->st, sizeof(Status)},
      {(void**)&s->nrm_part, 2 * (size_t)s->nn_cap * 4 + 16}, {(void**)&s->part2, 2 * (size_t)std::max(s->nblk, s->nblk_u) * 4 + 16},
      {(void**)&s->rel_trace, thr * 8 + 8}, {(void**)&s->abs_trace, thr * 8 + 8}};
  for (auto& a : allocs) {
    if (hipMalloc(a.p, a.n ? a.n : 16) != hipSuccess) {
      psignn_set_error("broyden: hipMalloc of %zu bytes failed (U+V need 2*thr*N*d*4 bytes)", a.n);
      return PSIGNN_ENOMEM;
    }
    s->bytes += a.n;
  }
  if (s->jgroups > 1 && s->vec_ax == s->vec) {  // scratch of a split axpy pass
    size_t jb = (size_t)s->jgroups * 3 * M * 4;
    if (hipMalloc((void**)&s->jpart, jb) != hipSuccess) {
      psignn_set_error("broyden: hipMalloc of the split-sweep scratch failed");
      return PSIGNN_ENOMEM;
    }
    s->bytes += jb;
  }
  if (s->plan) {
    size_t wf = (size_t)psignn_f_workspace_floats(s->plan) * 4;
    if (hipMalloc((void**)&s->fwork, wf) != hipSuccess) {
      psignn_set_error("broyden: hipMalloc of f workspace failed");
      return PSIGNN_ENOMEM;
    }
    s->bytes += wf;
    size_t N = (size_t)s->plan->N;
    if (hipMalloc((void**)&s->h0p, M * 4) != hipSuccess || hipMalloc((void**)&s->prbp, N * 3 * 4) != hipSuccess ||
        hipMalloc((void**)&s->nrmp, N * 2 * 4) != hipSuccess) {
      psignn_set_error("broyden: hipMalloc of plan-order inputs failed");
      return PSIGNN_ENOMEM;
    }
    s->bytes += M * 4 + N * 20;
  }
  if (hipHostMalloc((void**)&s->h_st, sizeof(Status)) != hipSuccess) {
    psignn_set_error("broyden: hipHostMalloc failed");
    return PSIGNN_ENOMEM;
  }
  return 0;
}

extern "C" void psignn_broyden_destroy(psignn_broyden_t* s) {
  if (!s) return;
  void* ptrs[] = {s->U, s->V, s->xbuf, s->gbuf[0], s->gbuf[1], s->upd, s->fx, s->fwork, s->part, s->coef, s->st, s->nrm_part, s->part2, s->parta,
                  s->rel_trace, s->abs_trace, s->h0p, s->prbp, s->nrmp, s->jpart};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (s->h_st) (void)hipHostFree(s->h_st);
  for (hipGraphExec_t g : s->graphs)
    if (g) (void)hipGraphExecDestroy(g);
  delete s;
}

extern "C" int psignn_broyden_create_n(psignn_broyden_t** out, int64_t n_elems, int seq_len, int threshold, int keep_trace) {
  ARG_CHECK(out, "out is NULL");
  *out = nullptr;
  ARG_CHECK(n_elems > 0 && threshold > 0 && seq_len > 0, "bad sizes");
  ARG_CHECK(cdiv(n_elems, 16 * TB) * (TB / 64) < (int64_t)INT32_MAX, "vector too long");
  psignn_broyden* s = new psignn_broyden();
  s->M = n_elems;
  s->seq_len = seq_len;
  s->thr = threshold;
  s->keep_trace = keep_trace;
  int rc = broyden_alloc(s);
  if (rc) {
    psignn_broyden_destroy(s);
    return rc;
  }
  *out = s;
  return PSIGNN_OK;
}

extern "C" int psignn_broyden_create_for_batch(psignn_broyden_t** out, const psignn_plan_t* plan, int threshold, int keep_trace,
                                               int64_t shard_elems);
extern "C" int psignn_broyden_create(psignn_broyden_t** out, const psignn_plan_t* plan, int threshold, int keep_trace) {
  return psignn_broyden_create_for_batch(out, plan, threshold, keep_trace, 0);
}
extern "C" int psignn_broyden_create_for_batch(psignn_broyden_t** out, const psignn_plan_t* plan, int threshold, int keep_trace,
                                               int64_t shard_elems) {
  ARG_CHECK(out, "out is NULL");
  *out = nullptr;
  ARG_CHECK(plan && threshold > 0 && shard_elems >= 0, "bad arguments");
  psignn_broyden* s = new psignn_broyden();
  s->size_hint = shard_elems;
  s->plan = plan;
  s->M = plan->N * D;
  s->seq_len = D;
  s->thr = threshold;
  s->keep_trace = keep_trace;
  int rc = broyden_alloc(s);
  if (rc) {
    psignn_broyden_destroy(s);
    return rc;
  }
  *out = s;
  return PSIGNN_OK;
}

extern "C" size_t psignn_broyden_bytes(const psignn_broyden_t* s) { return s ? s->bytes : 0; }

// stop_mode of the following solves: 0 = "rel" (default; every call site of the reference), 1 = "abs" (solver.py:116,140,174)
extern "C" int psignn_broyden_set_stop_mode(psignn_broyden_t* s, int abs_mode) {
  ARG_CHECK(s, "NULL solver");
  s->stop_abs = abs_mode ? 1 : 0;
  return PSIGNN_OK;
}

static inline int sel_off_cur() { return offsetof(Status, cur) / 4; }
static inline int sel_off_low() { return offsetof(Status, low) / 4; }
static inline int sel_off_nxt() { return offsetof(Status, nxt) / 4; }

// everything of one iteration after fx = f(x_next) is available; k = pairs stored so far
// fused_npart > 0: the f kernel already produced g, dg and the norm partials (fused_npart entries each)
static void launch_update(psignn_broyden* s, int k, double eps, hipStream_t st, int fused_npart = 0) {
  unsigned g = (unsigned)s->nblk;
  float* const gnew = s->gbuf[(k + 1) & 1];        // g of the iterate just evaluated
  const float* const gold = s->gbuf[k & 1];        // g of the iterate before it
  if (!fused_npart) {
    PROF_BYTES(3 * (int64_t)s->M * 4);
    VLAUNCH("k_resid", st, s->vec, k_resid, (g, TB, 0, st), s->M, s->st, s->xbuf, s->fx, gnew, s->nrm_part, s->nblk);
  }
  const int np = fused_npart ? fused_npart : s->nblk;  // one partial pair per block / per tile
  if (s->uvu) {
    const int kd = k >= s->thr ? 0 : k;
    const unsigned gu = (unsigned)s->nblk_u;
    if (k == 0) s->a_ready = 0;
    // a_j of this iteration: pairs j >= a_from were delivered by the folded sweep 3 of the last iteration, the others need sweep 1
    const int a_from = s->a_ready ? s->a_from : kd;
    if (std::min(a_from, kd) > 0) {
      PROF_BYTES((std::min(a_from, kd) + 1) * (int64_t)s->M * 4);   // its columns of U + dx
      VLAUNCH("k_sweep_u1", st, s->vec_u, k_sweep_u1, (gu, TB, 0, st), s->M, std::min(a_from, kd), s->st, s->U, s->upd, s->part, s->ldp, s->ld);
    }
    LAUNCH("k_reduce_check", st, (k_reduce_a_check<<<dim3(std::max(kd, 1), RA + 1), RB, 0, st>>>(
        s->st, s->part, s->nblk_u, s->ldp, s->thr, kd, s->coef, s->nrm_part, np, s->rel_trace, s->abs_trace, eps, s->seq_len, s->keep_trace,
        s->parta, s->nblk4, a_from)));
    s->a_ready = 0;
    if (k >= s->thr) return;
    // (iteration thr's stop test has just fired: the two sweeps below return at once and state no bytes)
    const bool last = k + 1 >= s->thr;
    PROF_BYTES(last ? 0 : (k + 4) * (int64_t)s->M * 4);   // k columns of V + dx, dg, g; writes V[k]
    VLAUNCH("k_sweep_v", st, s->vec_u, k_sweep_v, (gu, TB, 0, st), s->M, k, s->st, s->V, s->upd, gold, gnew, s->coef, s->part, s->pstride, s->ldp, s->part2, s->nblk_u, s->ld, s->thr);
    LAUNCH("k_reduce_cb", st, (k_reduce_cb<<<dim3(std::max(k, 1), 3), RB, 0, st>>>(s->st, s->part, s->nblk_u, s->pstride, s->ldp, s->thr, k, s->coef, s->part2, s->nblk_u)));
    const int keep0 = k <= s->u2d_kmax ? 0 : k - s->u2d_keep;      // few stored pairs: all kept; later the most recent ones
    if (s->u2d_kmax > 0 && (k <= s->u2d_kmax || s->u2d_keep > 0) && k + 1 < s->thr) {
      PROF_BYTES((k + 5) * (int64_t)s->M * 4);   // k columns of U + update, dg, g; writes U[k], update (whichever form runs)
      const int nk = k - keep0;
      if (s->u2d_reg) {
#define U2R_ARGS s->M, k, s->st, s->U, s->upd, gold, gnew, s->coef, s->thr, s->parta, s->ld, keep0
        if (nk <= 8) LAUNCH("k_sweep_u2d", st, (k_sweep_u2r<8><<<(unsigned)s->nblk4, TB, 0, st>>>(U2R_ARGS)));
        else if (nk <= 16) LAUNCH("k_sweep_u2d", st, (k_sweep_u2r<16><<<(unsigned)s->nblk4, TB, 0, st>>>(U2R_ARGS)));
        else LAUNCH("k_sweep_u2d", st, (k_sweep_u2r<24><<<(unsigned)s->nblk4, TB, 0, st>>>(U2R_ARGS)));
#undef U2R_ARGS
      } else {
        const size_t lds = (size_t)std::max(nk, 1) * TB * 16;
        LAUNCH("k_sweep_u2d", st, (k_sweep_u2d<<<(unsigned)s->nblk4, TB, lds, st>>>(s->M, k, s->st, s->U, s->upd, gold, gnew, s->coef, s->thr, s->parta,
                                                                                    s->ld, keep0)));
      }
      s->a_ready = 1;
      s->a_from = keep0;
    } else {
      PROF_BYTES(last ? 0 : (k + 5) * (int64_t)s->M * 4);
      VLAUNCH("k_sweep_u2", st, s->vec_u, k_sweep_u2, (gu, TB, 0, st), s->M, k, s->st, s->U, s->upd, gold, gnew, s->coef, s->thr, s->ld);
    }
    return;
  }
  // split of the sweeps over the stored pairs: only when there are enough pairs to share out
  const int G = (s->jgroups > 1 && k >= 4 * s->jgroups) ? s->jgroups : 1;
  const int js = (int)cdiv(std::max(k, 1), G);
  const int kd = k >= s->thr ? 0 : k;  // the threshold stop is about to fire: no slot left for another pair
  if (kd > 0) {
    PROF_BYTES((2 * kd + 3) * (int64_t)s->M * 4);
    VLAUNCH("k_dots", st, s->vec, k_dots, (dim3(g, G), TB, 0, st), s->M, kd, s->st, s->U, s->V, s->upd, gold, gnew, s->part, s->pstride, s->ldp, js, s->ld);
  }
  LAUNCH("k_reduce_check", st, (k_reduce_check<<<dim3(std::max(kd, 1), 4), RB, 0, st>>>(
      s->st, s->part, s->nblk, s->pstride, s->ldp, s->thr, kd, s->coef, s->nrm_part, np, s->rel_trace, s->abs_trace, eps, s->seq_len, s->keep_trace)));
  if (k >= s->thr) return;
  if (s->vec_ax != s->vec) {  // unsplit, own width
    const unsigned ga = (unsigned)s->nblk_ax;
    PROF_BYTES(k + 1 >= s->thr ? 0 : (2 * k + 6) * (int64_t)s->M * 4);
    VLAUNCH("k_axpy", st, s->vec_ax, k_axpy, (dim3(ga, 1), TB, 0, st), s->M, k, s->st, s->U, s->V, s->upd, gold, gnew, s->coef, s->thr, s->part, s->nblk_ax, std::max(k, 1), s->jpart, s->ld);
    PROF_BYTES(k + 1 >= s->thr ? 0 : 4 * (int64_t)s->M * 4);
    VLAUNCH("k_final", st, s->vec_ax, k_final, (ga, TB, 0, st), s->M, k, s->st, s->U, s->upd, s->ld, s->part, s->nblk_ax);
    return;
  }
  PROF_BYTES(k + 1 >= s->thr ? 0 : (2 * k + (G > 1 ? 3 * G : 6)) * (int64_t)s->M * 4);   // split: every block row writes its three partial vectors
  VLAUNCH("k_axpy", st, s->vec, k_axpy, (dim3(g, G), TB, 0, st), s->M, k, s->st, s->U, s->V, s->upd, gold, gnew, s->coef, s->thr, s->part, s->nblk, js, s->jpart, s->ld);
  if (G > 1) {
    PROF_BYTES(k + 1 >= s->thr ? 0 : (3 * G + 6) * (int64_t)s->M * 4);
    VLAUNCH("k_axpy_combine", st, s->vec, k_axpy_combine, (g, TB, 0, st), s->M, k, G, s->st, s->jpart, s->U, s->V, s->upd, gold, gnew, s->part, s->nblk, s->ld);
  }
  PROF_BYTES(k + 1 >= s->thr ? 0 : 4 * (int64_t)s->M * 4);
  VLAUNCH("k_final", st, s->vec, k_final, (g, TB, 0, st), s->M, k, s->st, s->U, s->upd, s->ld, s->part, s->nblk);
}

static int read_status(psignn_broyden* s, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(s->h_st, s->st, sizeof(Status), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

static int finish(psignn_broyden* s, float* d_result, psignn_solve_info_t* info, double* h_rel, double* h_abs,
                  hipStream_t st) {
  unsigned g = (unsigned)s->nblk;
  int rc = 0;
  if (d_result) {
    const int32_t* sel_low = reinterpret_cast<const int32_t*>(s->st) + sel_off_low();
    if (s->plan && s->plan_order) {  // iterates live in plan order: select into fx, then back to the caller's numbering
      VPLAIN(s->vec, k_copy_sel, (g, TB, 0, st), s->M, s->xbuf, sel_low, 0, s->fx);
      if ((rc = psignn_plan_permute(s->plan, s->fx, D, d_result, 0, st))) return rc;
    } else {
      VPLAIN(s->vec, k_copy_sel, (g, TB, 0, st), s->M, s->xbuf, sel_low, 0, d_result);
    }
  }
  if ((rc = read_status(s, st))) return rc;
  const Status& h = *s->h_st;
  if (info) {
    info->nstep = h.stop_abs ? h.lowest_step_abs : h.lowest_step;
    info->n_iter = h.n_iter;
    info->prot_break = h.prot_break;
    info->stop_reason = h.stop_reason;
    info->lowest = h.lowest_rel;          // callers pick lowest / lowest_abs by their stop_mode
    info->lowest_abs = h.lowest_abs_at;
  }
  int n = h.n_iter;
  if (h_rel && n > 0) HIP_TRY(hipMemcpy(h_rel, s->rel_trace, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (h_abs && n > 0) HIP_TRY(hipMemcpy(h_abs, s->abs_trace, (size_t)n * 8, hipMemcpyDeviceToHost));
  // pad to `threshold` entries with the lowest values, as the reference pads its lists (solver.py:195-197)
  for (int i = n; i < s->thr; ++i) {
    if (h_rel) h_rel[i] = h.lowest_rel;
    if (h_abs) h_abs[i] = h.lowest_abs_at;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

extern "C" int psignn_broyden_solve(psignn_broyden_t* s, const float* W, int nl, const float* h0, const float* prb,
                                    const float* nrm, double eps, int poll_every, float* d_result,
                                    psignn_solve_info_t* info, double* h_rel, double* h_abs, void* stream) {
  ARG_CHECK(s && s->plan, "solver was not created from a mesh plan");
  ARG_CHECK(W && h0 && prb, "NULL argument");
  ARG_CHECK(!s->plan->mixed || nrm, "mixed plan needs unit normals");
  hipStream_t st = (hipStream_t)stream;
  s->plan_order = 1;
  if (poll_every <= 0) poll_every = 8;
  unsigned g = (unsigned)s->nblk;
  const int32_t* sel_nxt = reinterpret_cast<const int32_t*>(s->st) + sel_off_nxt();
  k_init_status<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->thr, s->stop_abs);
  // node tensors into plan order once per solve (identity copy on an untiled plan)
  int rc;
  const psignn_plan* p = s->plan;
  if ((rc = psignn_plan_permute(p, h0, D, s->h0p, 1, st))) return rc;
  if ((rc = psignn_plan_permute(p, prb, p->mixed ? 3 : 2, s->prbp, 1, st))) return rc;
  if (p->mixed && (rc = psignn_plan_permute(p, nrm, 2, s->nrmp, 1, st))) return rc;
  const float* nrmp = p->mixed ? s->nrmp : nullptr;
  // gx0 = f(x0) - x0, update = gx0 (solver.py:131-136)
  if ((rc = psignn_f_eval_p(p, W, nl, s->h0p, nullptr, 0, s->h0p, s->prbp, nrmp, s->fx, s->fwork, st))) return rc;
  VPLAIN(s->vec, k_begin, (g, TB, 0, st), s->M, s->h0p, s->fx, s->xbuf, s->gbuf[0], s->upd);
  const bool fused = p->tiled && (nl == 1 || p->mixed);
  const int32_t* st_words = reinterpret_cast<const int32_t*>(s->st);
  KNOB_INT(use_graph, [] { const char* e = getenv("PSIGNN_GRAPH"); return (int)(e && atoi(e) != 0); }());
  if (use_graph && fused && !g_prof_on) {
    // one graph per chunk of poll_every iterations (the iteration index is a kernel argument); every kernel returns at once
    // when the device-side done flag is set, so a chunk that overshoots the stop is harmless
    uint64_t key = (uint64_t)(uintptr_t)W * 1000003u ^ (uint64_t)(uintptr_t)nrmp * 7919u ^ (uint64_t)nl * 31u ^ (uint64_t)poll_every;
    double e = eps;
    key ^= *reinterpret_cast<uint64_t*>(&e);
    if (key != s->graph_key) {
      for (hipGraphExec_t g : s->graphs)
        if (g) (void)hipGraphExecDestroy(g);
      s->graphs.clear();
      s->graph_key = key;
    }
    const int nchunk = (int)cdiv(s->thr, poll_every);
    s->graphs.resize(nchunk, nullptr);
    for (int c = 0; c < nchunk; ++c) {
      if (!s->graphs[c]) {
        hipGraph_t graph;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int it = c * poll_every; it < std::min(s->thr, (c + 1) * poll_every); ++it) {
          rc = psignn_f_tile_fused(p, W, nl, s->xbuf, s->M, st_words, offsetof(Status, done) / 4, sel_off_cur(), sel_off_nxt(),
                                   s->upd, s->gbuf[(it + 1) & 1], s->h0p, s->prbp, nrmp, s->nrm_part, st);
          if (rc >= 0) launch_update(s, it, eps, st, rc);
        }
        HIP_TRY(hipStreamEndCapture(st, &graph));
        if (rc < 0) return rc;
        HIP_TRY(hipGraphInstantiate(&s->graphs[c], graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphDestroy(graph));
      }
      HIP_TRY(hipGraphLaunch(s->graphs[c], st));
      rc = read_status(s, st);
      if (rc) return rc;
      if (s->h_st->done) break;
    }
    return finish(s, d_result, info, h_rel, h_abs, st);
  }
  for (int it = 0; it < s->thr; ++it) {
    if (fused) {
      // one kernel: x_next = x_cur + update, f(x_next), g_new, dg, x_next and the norm partials
      rc = psignn_f_tile_fused(p, W, nl, s->xbuf, s->M, st_words, offsetof(Status, done) / 4, sel_off_cur(),
                               sel_off_nxt(), s->upd, s->gbuf[(it + 1) & 1], s->h0p, s->prbp, nrmp, s->nrm_part, st);
      if (rc < 0) return rc;
      launch_update(s, it, eps, st, rc);
    } else {
      PROF_BYTES(3 * (int64_t)s->M * 4);
      VLAUNCH("k_xnext", st, s->vec, k_xnext, (g, TB, 0, st), s->M, s->st, s->xbuf, s->upd, nullptr);
      rc = psignn_f_eval_p(p, W, nl, s->xbuf, sel_nxt, s->M, s->h0p, s->prbp, nrmp, s->fx, s->fwork, st);
      if (rc) return rc;
      launch_update(s, it, eps, st);
    }
    if ((it + 1) % poll_every == 0 || it + 1 == s->thr) {
      rc = read_status(s, st);
      if (rc) return rc;
      if (s->h_st->done) break;
    }
  }
  return finish(s, d_result, info, h_rel, h_abs, st);
}


// ------------------------------------------------------------------------------------------
// Batched solve: the meshes of one shard (BASELINE configs[3]: 8 independent 50k-node meshes per GPU) iterate in lockstep,
// every per-iteration pass is ONE launch over all of them (blockIdx.z = mesh).  Eight concurrent single-mesh solves on
// eight streams reach 3.7 TB/s aggregate -- barely more than one solve alone: each of their kernels spans the chip and pays
// its own ramp and tail.  A block runs the single-mesh kernels' bodies on its mesh's descriptor: same block -> element
// mapping, same partial-sum shapes, own status block and stop test per mesh  =>  bit-identical to the mesh's own solve.
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ int batch_groups(const BatchDesc& d, int k) {
  return (d.jgroups > 1 && k >= 4 * d.jgroups) ? d.jgroups : 1;
}
static __device__ __forceinline__ int idiv_up(int a, int b) { return (a + b - 1) / b; }
// g of iterate i lives in g[i & 1] (see psignn_broyden::gbuf); par = iteration & 1
static __device__ __forceinline__ const float* bd_gold(const BatchDesc& d, int par) { return par ? d.g1 : d.g0; }
static __device__ __forceinline__ float* bd_gnew(const BatchDesc& d, int par) { return par ? d.g0 : d.g1; }

template <int VEC>
__global__ __launch_bounds__(TB) void kb_dots(const BatchDesc* __restrict__ descs, int k, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  const int G = batch_groups(d, k);
  if ((int)blockIdx.x >= d.nblk || (int)blockIdx.y >= G) return;
  dots_body<VEC>(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.V, d.upd, bd_gold(d, par), bd_gnew(d, par), d.part, d.pstride, (d.thr + 63) / 64 * 64,
                 idiv_up(k > 1 ? k : 1, G), d.ld);
}
__global__ __launch_bounds__(RB) void kb_reduce_check(const BatchDesc* __restrict__ descs, int k, double eps) {
  __shared__ double sh[RB];
  const BatchDesc& d = descs[blockIdx.z];
  reduce_check_body(reinterpret_cast<Status*>(d.st), d.part, d.nblk, d.pstride, (d.thr + 63) / 64 * 64, d.thr, k, d.coef, d.nrm_part, d.n_tiles, d.rel_trace,
                    d.abs_trace, eps, d.seq_len, d.keep_trace, sh);
}
template <int VEC>
__global__ __launch_bounds__(TB) void kb_axpy(const BatchDesc* __restrict__ descs, int k, int own_width, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  // own_width: the axpy / final passes run unsplit with their own vector width (broyden_alloc: vec_ax != vec)
  const int G = own_width ? 1 : batch_groups(d, k);
  const int nb = own_width ? d.nblk_ax : d.nblk;
  if ((int)blockIdx.x >= nb || (int)blockIdx.y >= G) return;
  axpy_body<VEC>(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.V, d.upd, bd_gold(d, par), bd_gnew(d, par), d.coef, d.thr, d.part, nb,
                 idiv_up(k > 1 ? k : 1, G), d.jpart, d.ld, G > 1);
}
template <int VEC>
__global__ __launch_bounds__(TB) void kb_axpy_combine(const BatchDesc* __restrict__ descs, int k, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  const int G = batch_groups(d, k);
  if ((int)blockIdx.x >= d.nblk || G <= 1) return;
  axpy_combine_body<VEC>(d.M, k, G, reinterpret_cast<const Status*>(d.st), d.jpart, d.U, d.V, d.upd, bd_gold(d, par), bd_gnew(d, par), d.part, d.nblk, d.ld);
}
template <int VEC>
__global__ __launch_bounds__(TB) void kb_final(const BatchDesc* __restrict__ descs, int k, int own_width) {
  __shared__ double sh[TB];
  const BatchDesc& d = descs[blockIdx.z];
  const int nb = own_width ? d.nblk_ax : d.nblk;
  if ((int)blockIdx.x >= nb) return;
  final_body<VEC>(d.M, k, reinterpret_cast<Status*>(d.st), d.U, d.upd, d.ld, d.part, nb, sh);
}
// three-sweep update, batched (grid (blocks, 1 | 2, meshes))
template <int VEC>
__global__ __launch_bounds__(TB) void kb_sweep_u1(const BatchDesc* __restrict__ descs, int k) {
  const BatchDesc& d = descs[blockIdx.z];
  if ((int)blockIdx.x >= d.nblk_u) return;
  sweep_u1_body<VEC>(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.upd, d.part, (d.thr + 63) / 64 * 64, d.ld);
}
__global__ __launch_bounds__(RB) void kb_reduce_a_check(const BatchDesc* __restrict__ descs, int k, double eps, int a_from) {
  __shared__ double sh[RB];
  const BatchDesc& d = descs[blockIdx.z];
  reduce_a_check_body(reinterpret_cast<Status*>(d.st), d.part, d.nblk_u, (d.thr + 63) / 64 * 64, d.thr, k, d.coef, d.nrm_part, d.n_tiles, d.rel_trace,
                      d.abs_trace, eps, d.seq_len, d.keep_trace, sh, d.parta, d.nblk4, a_from);
}
__global__ __launch_bounds__(TB) void kb_sweep_u2d(const BatchDesc* __restrict__ descs, int k, int j_keep0, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  if ((int)blockIdx.x >= d.nblk4) return;
  sweep_u2d_body(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.upd, bd_gold(d, par), bd_gnew(d, par), d.coef, d.thr, d.parta, d.ld, j_keep0);
}
template <int KB>
__global__ __launch_bounds__(TB, U2RWaves<KB>::value) void kb_sweep_u2r(const BatchDesc* __restrict__ descs, int k, int j_keep0, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  if ((int)blockIdx.x >= d.nblk4) return;
  sweep_u2r_body<KB>(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.upd, bd_gold(d, par), bd_gnew(d, par), d.coef, d.thr, d.parta, d.ld, j_keep0);
}
template <int VEC>
__global__ __launch_bounds__(TB) void kb_sweep_v(const BatchDesc* __restrict__ descs, int k, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  if ((int)blockIdx.x >= d.nblk_u) return;
  sweep_v_body<VEC>(d.M, k, reinterpret_cast<const Status*>(d.st), d.V, d.upd, bd_gold(d, par), bd_gnew(d, par), d.coef, d.part, d.pstride, (d.thr + 63) / 64 * 64, d.part2,
                    d.nblk_u, d.ld, d.thr);
}
__global__ __launch_bounds__(RB) void kb_reduce_cb(const BatchDesc* __restrict__ descs, int k) {
  __shared__ double sh[RB];
  const BatchDesc& d = descs[blockIdx.z];
  reduce_cb_body(reinterpret_cast<Status*>(d.st), d.part, d.nblk_u, d.pstride, (d.thr + 63) / 64 * 64, d.thr, k, d.coef, d.part2, d.nblk_u, sh);
}
template <int VEC>
__global__ __launch_bounds__(TB) void kb_sweep_u2(const BatchDesc* __restrict__ descs, int k, int par) {
  const BatchDesc& d = descs[blockIdx.z];
  if ((int)blockIdx.x >= d.nblk_u) return;
  sweep_u2_body<VEC>(d.M, k, reinterpret_cast<const Status*>(d.st), d.U, d.upd, bd_gold(d, par), bd_gnew(d, par), d.coef, d.thr, d.ld);
}
// *all_done = 1 when every mesh's stop test has fired
__global__ void kb_all_done(const BatchDesc* __restrict__ descs, int n, int off_done, int32_t* __restrict__ all_done) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int a = 1;
    for (int m = 0; m < n; ++m) a &= descs[m].st[off_done] != 0;
    *all_done = a;
  }
}

int psignn_f_tile_fused_batch(const BatchDesc* d_descs, int n_mesh, int n_slots, int max_rows, const float* W, int mixed,
                              int off_done, int off_cur, int off_nxt, int par, hipStream_t st);

// 1 when psignn_broyden_solve_batch takes these solvers together: tiled plans of ONE boundary-condition family and one size
// class (the vector width / split layout / threshold / fold limits that broyden_alloc derives from the shard size) -- a host-side
// question, asked before the solve so that a shard the batched solver cannot take is not an error path.
extern "C" int psignn_broyden_batchable(int n, psignn_broyden_t* const* sv) {
  if (n <= 0 || !sv || !sv[0] || !sv[0]->plan) return 0;
  const psignn_broyden* s0 = sv[0];
  for (int m = 0; m < n; ++m) {
    const psignn_broyden* s = sv[m];
    if (!s || !s->plan || !s->plan->tiled || s->plan->mixed != s0->plan->mixed) return 0;
    if (!(s->vec == s0->vec && s->vec_ax == s0->vec_ax && s->uvu == s0->uvu && s->vec_u == s0->vec_u && s->thr == s0->thr &&
          s->u2d_kmax == s0->u2d_kmax && s->u2d_keep == s0->u2d_keep && s->u2d_reg == s0->u2d_reg))
      return 0;
  }
  return 1;
}

extern "C" int psignn_broyden_solve_batch(int n, psignn_broyden_t** sv, const float* W, int nl, const float* const* h0,
                                          const float* const* prb, const float* const* nrm, double eps, int poll_every,
                                          float* const* d_results, psignn_solve_info_t* infos, double* const* h_rel,
                                          double* const* h_abs, void* stream) {
  ARG_CHECK(n > 0 && sv && W && h0 && prb, "bad arguments");
  ARG_CHECK(nl == 1, "the batched solver runs single-layer blocks");
  hipStream_t st = (hipStream_t)stream;
  if (poll_every <= 0) poll_every = 8;
  // one vector width / split layout / threshold for the whole shard (meshes of one shard are of one size class)
  const psignn_broyden* s0 = sv[0];
  int max_g = 0, max_ga = 0, max_gu = 0, max_g4 = 0, max_G = 1, max_rows = 0, n_slots = 0;
  for (int m = 0; m < n; ++m) {
    const psignn_broyden* s = sv[m];
    ARG_CHECK(s && s->plan && s->plan->tiled, "batched solve: tiled plans only");
    ARG_CHECK(s->plan->mixed == s0->plan->mixed, "batched solve: one boundary-condition family per shard");
    ARG_CHECK(s->vec == s0->vec && s->vec_ax == s0->vec_ax && s->uvu == s0->uvu && s->vec_u == s0->vec_u && s->thr == s0->thr &&
                  s->u2d_kmax == s0->u2d_kmax && s->u2d_keep == s0->u2d_keep && s->u2d_reg == s0->u2d_reg,
              "batched solve: meshes of different size classes (vector width / threshold differ)");
    ARG_CHECK(h0[m] && prb[m], "NULL argument");
    ARG_CHECK(!s->plan->mixed || (nrm && nrm[m]), "mixed plans need unit normals");
    max_g = std::max(max_g, s->nblk);
    max_ga = std::max(max_ga, s->nblk_ax);
    max_gu = std::max(max_gu, s->nblk_u);
    max_G = std::max(max_G, s->jgroups);
    max_rows = std::max(max_rows, s->plan->max_rows);
    n_slots += (int)s->plan->n_tiles;
  }
  const bool own_width = s0->vec_ax != s0->vec;
  int64_t Mtot4 = 0, bf_tot = 0;   // bytes of one state vector / of one fused f evaluation, summed over the shard (profiling records)
  for (int m = 0; m < n; ++m) {
    Mtot4 += sv[m]->M * 4;
    bf_tot += (sv[m]->plan->mixed ? 102 : 89) * sv[m]->plan->N + 20 * sv[m]->plan->Ep + 8 * sv[m]->M;
  }
  // ---- per mesh: status, plan-order inputs, g0 = f(x0) - x0 (exactly the single-mesh prologue)
  std::vector<BatchDesc> hd(n);
  int rc, base = 0;
  for (int m = 0; m < n; ++m) {
    psignn_broyden* s = sv[m];
    const psignn_plan* p = s->plan;
    s->plan_order = 1;
    k_init_status<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->thr, s->stop_abs);
    if ((rc = psignn_plan_permute(p, h0[m], D, s->h0p, 1, st))) return rc;
    if ((rc = psignn_plan_permute(p, prb[m], p->mixed ? 3 : 2, s->prbp, 1, st))) return rc;
    if (p->mixed && (rc = psignn_plan_permute(p, nrm[m], 2, s->nrmp, 1, st))) return rc;
    const float* nrmp = p->mixed ? s->nrmp : nullptr;
    if ((rc = psignn_f_eval_p(p, W, nl, s->h0p, nullptr, 0, s->h0p, s->prbp, nrmp, s->fx, s->fwork, st))) return rc;
    VPLAIN(s->vec, k_begin, ((unsigned)s->nblk, TB, 0, st), s->M, s->h0p, s->fx, s->xbuf, s->gbuf[0], s->upd);
    BatchDesc& d = hd[m];
    d.M = s->M; d.ld = s->ld; d.nblk = s->nblk; d.npart = s->npart; d.nblk_ax = s->nblk_ax; d.jgroups = s->jgroups;
    d.thr = s->thr; d.seq_len = s->seq_len; d.keep_trace = s->keep_trace; d.n_tiles = (int)p->n_tiles; d.tile_base = base;
    d.st = reinterpret_cast<int32_t*>(s->st);
    d.U = s->U; d.V = s->V; d.xbuf = s->xbuf; d.g0 = s->gbuf[0]; d.g1 = s->gbuf[1]; d.upd = s->upd; d.part = s->part; d.coef = s->coef;
    d.nrm_part = s->nrm_part; d.jpart = s->jpart; d.rel_trace = s->rel_trace; d.abs_trace = s->abs_trace;
    d.ctx = p->d_ctx; d.h0p = s->h0p; d.prbp = s->prbp;
    d.part2 = s->part2; d.nblk_u = s->nblk_u; d.npart_u = s->npart_u;
    d.parta = s->parta; d.nblk4 = s->nblk4; d.pad_ = 0; d.nrmp = nrmp; d.pstride = s->pstride;
    max_g4 = std::max(max_g4, s->nblk4);
    base += (int)p->n_tiles;
  }
  BatchDesc* d_descs = nullptr;
  int32_t *d_done = nullptr, *h_done = nullptr;
  auto cleanup = [&]() {
    if (d_descs) (void)hipFree(d_descs);
    if (d_done) (void)hipFree(d_done);
    if (h_done) (void)hipHostFree(h_done);
  };
#define BT(expr)                                                                          \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      psignn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      cleanup();                                                                          \
      return PSIGNN_EHIP;                                                                 \
    }                                                                                     \
  } while (0)
  BT(hipMalloc((void**)&d_descs, sizeof(BatchDesc) * n));
  BT(hipMalloc((void**)&d_done, 4));
  BT(hipHostMalloc((void**)&h_done, 4));
  BT(hipMemcpyAsync(d_descs, hd.data(), sizeof(BatchDesc) * n, hipMemcpyHostToDevice, st));
  const int off_done = offsetof(Status, done) / 4;
  const int thr = s0->thr;
  bool a_ready = false;
  int a_from_next = 0;
  for (int it = 0; it < thr; ++it) {
    PROF_BYTES(bf_tot);
    const int par = it & 1;
    rc = psignn_f_tile_fused_batch(d_descs, n, n_slots, max_rows, W, s0->plan->mixed, off_done, sel_off_cur(), sel_off_nxt(), par, st);
    if (rc) { cleanup(); return rc; }
    const int k = it;
    const int kd = k >= thr ? 0 : k;
    if (s0->uvu) {
      const dim3 gu((unsigned)max_gu, 1, (unsigned)n);
      // (all meshes of the shard carry the same number of stored pairs: one a_from / keep window for the launch)
      const int a_from = a_ready ? a_from_next : kd;
      if (std::min(a_from, kd) > 0) {
        PROF_BYTES((std::min(a_from, kd) + 1) * Mtot4);
        VLAUNCH("k_sweep_u1", st, s0->vec_u, kb_sweep_u1, (gu, TB, 0, st), d_descs, std::min(a_from, kd));
      }
      LAUNCH("k_reduce_check", st, (kb_reduce_a_check<<<dim3((unsigned)std::max(kd, 1), RA + 1, (unsigned)n), RB, 0, st>>>(d_descs, kd, eps, a_from)));
      a_ready = false;
      const bool last = k + 1 >= thr;   // the stop test of iteration thr has fired: the sweeps below return at once
      PROF_BYTES(last ? 0 : (k + 4) * Mtot4);
      VLAUNCH("k_sweep_v", st, s0->vec_u, kb_sweep_v, (gu, TB, 0, st), d_descs, k, par);
      LAUNCH("k_reduce_cb", st, (kb_reduce_cb<<<dim3((unsigned)std::max(k, 1), 3, (unsigned)n), RB, 0, st>>>(d_descs, k)));
      const int keep0 = k <= s0->u2d_kmax ? 0 : k - s0->u2d_keep;
      if (s0->u2d_kmax > 0 && (k <= s0->u2d_kmax || s0->u2d_keep > 0) && k + 1 < thr) {
        PROF_BYTES((k + 5) * Mtot4);
        const int nk = k - keep0;
        const dim3 g4((unsigned)max_g4, 1, (unsigned)n);
        if (s0->u2d_reg) {
          if (nk <= 8) LAUNCH("k_sweep_u2d", st, (kb_sweep_u2r<8><<<g4, TB, 0, st>>>(d_descs, k, keep0, par)));
          else if (nk <= 16) LAUNCH("k_sweep_u2d", st, (kb_sweep_u2r<16><<<g4, TB, 0, st>>>(d_descs, k, keep0, par)));
          else LAUNCH("k_sweep_u2d", st, (kb_sweep_u2r<24><<<g4, TB, 0, st>>>(d_descs, k, keep0, par)));
        } else {
          const size_t lds = (size_t)std::max(nk, 1) * TB * 16;
          LAUNCH("k_sweep_u2d", st, (kb_sweep_u2d<<<g4, TB, lds, st>>>(d_descs, k, keep0, par)));
        }
        a_ready = true;
        a_from_next = keep0;
      } else {
        PROF_BYTES(last ? 0 : (k + 5) * Mtot4);
        VLAUNCH("k_sweep_u2", st, s0->vec_u, kb_sweep_u2, (gu, TB, 0, st), d_descs, k, par);
      }
    } else {
    if (kd > 0) {
      PROF_BYTES((2 * kd + 3) * Mtot4);
      VLAUNCH("k_dots", st, s0->vec, kb_dots, (dim3((unsigned)max_g, (unsigned)max_G, (unsigned)n), TB, 0, st), d_descs, kd, par);
    }
    LAUNCH("k_reduce_check", st, (kb_reduce_check<<<dim3((unsigned)std::max(kd, 1), 4, (unsigned)n), RB, 0, st>>>(d_descs, kd, eps)));
    if (own_width) {
      PROF_BYTES(k + 1 >= thr ? 0 : (2 * k + 6) * Mtot4);
      VLAUNCH("k_axpy", st, s0->vec_ax, kb_axpy, (dim3((unsigned)max_ga, 1, (unsigned)n), TB, 0, st), d_descs, k, 1, par);
      PROF_BYTES(k + 1 >= thr ? 0 : 4 * Mtot4);
      VLAUNCH("k_final", st, s0->vec_ax, kb_final, (dim3((unsigned)max_ga, 1, (unsigned)n), TB, 0, st), d_descs, k, 1);
    } else {
      PROF_BYTES(k + 1 >= thr ? 0 : (2 * k + 6) * Mtot4);
      VLAUNCH("k_axpy", st, s0->vec, kb_axpy, (dim3((unsigned)max_g, (unsigned)max_G, (unsigned)n), TB, 0, st), d_descs, k, 0, par);
      if (max_G > 1 && k >= 4 * 2)   // some mesh may split from k = 4 * jgroups on (jgroups >= 2)
        VLAUNCH("k_axpy_combine", st, s0->vec, kb_axpy_combine, (dim3((unsigned)max_g, 1, (unsigned)n), TB, 0, st), d_descs, k, par);
      PROF_BYTES(k + 1 >= thr ? 0 : 4 * Mtot4);
      VLAUNCH("k_final", st, s0->vec, kb_final, (dim3((unsigned)max_g, 1, (unsigned)n), TB, 0, st), d_descs, k, 0);
    }
    }
    if ((it + 1) % poll_every == 0 || it + 1 == thr) {
      kb_all_done<<<1, 64, 0, st>>>(d_descs, n, off_done, d_done);
      BT(hipMemcpyAsync(h_done, d_done, 4, hipMemcpyDeviceToHost, st));
      BT(hipStreamSynchronize(st));
      if (*h_done) break;
    }
  }
  rc = PSIGNN_OK;
  for (int m = 0; m < n && rc == PSIGNN_OK; ++m)
    rc = finish(sv[m], d_results ? d_results[m] : nullptr, infos ? &infos[m] : nullptr, h_rel ? h_rel[m] : nullptr,
                h_abs ? h_abs[m] : nullptr, st);
  cleanup();
#undef BT
  return rc;
}


// ---- adjoint fixed point  y = J_f(h*)^T y + grad  (the reference's backward hook, dirichlet/psignn/model.py:210-223)
template <int VEC>
__global__ __launch_bounds__(TB) void k_addv(int64_t M, const Status* __restrict__ st, float* __restrict__ a,
                                             const float* __restrict__ b) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  float x[VEC], y[VEC];
  ldv<VEC>(a, e0, M, x);
  ldv<VEC>(b, e0, M, y);
#pragma unroll
  for (int i = 0; i < VEC; ++i) x[i] += y[i];
  stv<VEC>(a, e0, M, x);
}

extern "C" int psignn_f_vjp(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                            const float* nrm, const float* w, float* out, float* work, void* stream);
extern "C" int psignn_f_vjp_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                              const float* nrm, const float* w, float* out, float* work, void* stream);

extern "C" int psignn_broyden_solve_adjoint(psignn_broyden_t* s, const float* W, int nl, const float* h_star,
                                            const float* prb, const float* nrm, const float* grad, double eps,
                                            int poll_every, float* d_result, psignn_solve_info_t* info, double* h_rel,
                                            double* h_abs, void* stream) {
  ARG_CHECK(s && s->plan, "solver was not created from a mesh plan");
  ARG_CHECK(W && h_star && prb && grad, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const psignn_plan* p = s->plan;
  // tiled plans the tiled VJP covers (single-layer dirichlet, mixed): the whole solve in plan order; otherwise the caller's numbering
  const bool tiled = p->tiled && (p->mixed || nl == 1);
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  s->plan_order = tiled ? 1 : 0;
  if (poll_every <= 0) poll_every = 8;
  unsigned g = (unsigned)s->nblk;
  int rc;
  if (tiled) {  // plan-order copies: h* -> fwork tail, prb -> prbp, grad -> dg (free until the first update)
    float* hs_p = s->fwork + p->N * 4 * D;        // fwork = [B (40N) | h*_p (10N) | grad_p (10N) | ...]
    float* gr_p = hs_p + p->N * D;
    if ((rc = psignn_plan_permute(p, h_star, D, hs_p, 1, st))) return rc;
    if ((rc = psignn_plan_permute(p, grad, D, gr_p, 1, st))) return rc;
    if ((rc = psignn_plan_permute(p, prb, p->mixed ? 3 : 2, s->prbp, 1, st))) return rc;
    if (p->mixed) {
      if ((rc = psignn_plan_permute(p, nrm, 2, s->nrmp, 1, st))) return rc;
      nrm = s->nrmp;
    }
    h_star = hs_p;
    grad = gr_p;
    prb = s->prbp;
  }
  k_init_status<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->thr, s->stop_abs);
  // y0 = 0, map(y0) = grad  ->  g0 = grad, update = grad  (solver.py:131-136 with f(0) = grad)
  HIP_TRY(hipMemsetAsync(s->h0p, 0, (size_t)s->M * 4, st));
  VPLAIN(s->vec, k_begin, (g, TB, 0, st), s->M, s->h0p, grad, s->xbuf, s->gbuf[0], s->upd);
  for (int it = 0; it < s->thr; ++it) {
    // x_next = x + update, kept also in h0p (fixed address for the VJP kernels)
    VLAUNCH("k_xnext", st, s->vec, k_xnext, (g, TB, 0, st), s->M, s->st, s->xbuf, s->upd, s->h0p);
    rc = tiled ? psignn_f_vjp_p(p, W, nl, h_star, prb, nrm, s->h0p, s->fx, s->fwork, st)
               : psignn_f_vjp(p, W, nl, h_star, prb, nrm, s->h0p, s->fx, s->fwork, st);
    if (rc) return rc;
    VLAUNCH("k_addv", st, s->vec, k_addv, (g, TB, 0, st), s->M, s->st, s->fx, grad);
    launch_update(s, it, eps, st);
    if ((it + 1) % poll_every == 0 || it + 1 == s->thr) {
      rc = read_status(s, st);
      if (rc) return rc;
      if (s->h_st->done) break;
    }
  }
  return finish(s, d_result, info, h_rel, h_abs, st);
}

extern "C" int psignn_broyden_get_iterate(const psignn_broyden_t* s, int i, float* d_dst, void* stream) {
  ARG_CHECK(s && d_dst, "NULL argument");
  ARG_CHECK(s->keep_trace, "solver was created without keep_trace");
  ARG_CHECK(i >= 0 && i <= s->thr, "iterate index out of range");
  if (s->plan && s->plan_order) {
    VPLAIN(s->vec, k_copy_sel, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->xbuf, nullptr, i, s->fx);
    return psignn_plan_permute(s->plan, s->fx, D, d_dst, 0, stream);
  }
  VPLAIN(s->vec, k_copy_sel, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->xbuf, nullptr, i, d_dst);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Stored rank-one pair j of the last solve: which = 0 -> U_j, 1 -> V_j (the reference's Us[..., j] / VTs[:, j], solver.py:134-135,190-191);
// which = 2 -> the current `update` vector (solver.py:136,192; j ignored).  Caller's numbering.  Read-out for diagnostics and for the parity tests, which check the Broyden recurrences of every
// update form on the device's own state (tests/test_gpu_solver_forms.py).
extern "C" int psignn_broyden_get_pair(const psignn_broyden_t* s, int j, int which, float* d_dst, void* stream) {
  ARG_CHECK(s && d_dst, "NULL argument");
  ARG_CHECK(which >= 0 && which <= 3, "which: 0 = U_j, 1 = V_j, 2 = the current update vector, 3 = partials of the folded sweep");
  if (which == 3) {   // diagnostics: the (nblk4, PARTA_LD) per-block partials of a, as they are (d_dst: at least that many floats)
    HIP_TRY(hipMemcpyAsync(d_dst, s->parta, (size_t)std::min<int64_t>(s->M, (int64_t)s->nblk4 * PARTA_LD) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PSIGNN_OK;
  }
  ARG_CHECK(which == 2 || (j >= 0 && j < s->thr), "pair index out of range");
  const float* row = which == 2 ? s->upd : (which ? s->V : s->U) + (int64_t)j * s->ld;
  if (s->plan && s->plan_order) {
    HIP_TRY(hipMemcpyAsync(s->fx, row, (size_t)s->M * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return psignn_plan_permute(s->plan, s->fx, D, d_dst, 0, stream);
  }
  HIP_TRY(hipMemcpyAsync(d_dst, row, (size_t)s->M * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PSIGNN_OK;
}

// ---- externally driven variant (user-supplied f) ------------------------------------------
extern "C" int psignn_broyden_ext_begin(psignn_broyden_t* s, const float* d_x0, const float* d_fx0, void* stream) {
  ARG_CHECK(s && d_x0 && d_fx0, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  k_init_status<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->thr, s->stop_abs);
  VPLAIN(s->vec, k_begin, ((unsigned)s->nblk, TB, 0, st), s->M, d_x0, d_fx0, s->xbuf, s->gbuf[0], s->upd);
  s->ext_iter = 0;
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
extern "C" int psignn_broyden_ext_next_x(psignn_broyden_t* s, float* d_x_new, void* stream) {
  ARG_CHECK(s && d_x_new, "NULL argument");
  VPLAIN(s->vec, k_xnext, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->st, s->xbuf, s->upd, d_x_new);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
extern "C" int psignn_broyden_ext_trial_x(psignn_broyden_t* s, double step, float* d_x_trial, void* stream) {
  ARG_CHECK(s && d_x_trial, "NULL argument");
  VPLAIN(s->vec, k_xtrial, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->st, s->xbuf, s->upd, (float)step, d_x_trial, 0);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
extern "C" int psignn_broyden_ext_scale_step(psignn_broyden_t* s, double step, void* stream) {
  ARG_CHECK(s, "NULL argument");
  VPLAIN(s->vec, k_xtrial, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->st, s->xbuf, s->upd, (float)step, nullptr, 1);
  s->a_ready = 0;   // the update changed: a = U^T dx has to be swept again
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
extern "C" int psignn_broyden_ext_update(psignn_broyden_t* s, const float* d_fx_new, double eps, int* h_done, void* stream) {
  ARG_CHECK(s && d_fx_new, "NULL argument");
  ARG_CHECK(s->ext_iter < s->thr, "more updates than threshold");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(s->fx, d_fx_new, (size_t)s->M * 4, hipMemcpyDeviceToDevice, st));
  launch_update(s, s->ext_iter, eps, st);
  s->ext_iter++;
  if (h_done) {
    int rc = read_status(s, st);
    if (rc) return rc;
    *h_done = s->h_st->done;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
extern "C" int psignn_broyden_ext_finish(psignn_broyden_t* s, float* d_result, psignn_solve_info_t* info,
                                         double* h_rel, double* h_abs, void* stream) {
  ARG_CHECK(s, "NULL argument");
  return finish(s, d_result, info, h_rel, h_abs, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Small MLP (Encoder / Decoder, model.py:370-392) and the residual SpMV (model.py:157-167)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_mlp2(int64_t n, int din, int hid, int dout, const float* __restrict__ x,
                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                             float* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
  if (r >= n) return;
  float xi[16], hv[16];
  for (int i = 0; i < din; ++i) xi[i] = x[r * din + i];
  for (int o = 0; o < hid; ++o) {
    float s = b1[o];
    for (int i = 0; i < din; ++i) s = fmaf(w1[o * din + i], xi[i], s);
    hv[o] = fmaxf(s, 0.f);
  }
  for (int o = 0; o < dout; ++o) {
    float s = b2[o];
    for (int i = 0; i < hid; ++i) s = fmaf(w2[o * hid + i], hv[i], s);
    out[r * dout + o] = s;
  }
}

extern "C" int psignn_mlp2(const float* x, int64_t n, int din, int hid, int dout, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* out, void* stream) {
  ARG_CHECK(x && w1 && b1 && w2 && b2 && out, "NULL argument");
  ARG_CHECK(n >= 0 && din >= 1 && din <= 16 && hid >= 1 && hid <= 16 && dout >= 1 && dout <= 16, "bad sizes");
  if (n == 0) return PSIGNN_OK;
  k_mlp2<<<(unsigned)cdiv(n, TB), TB, 0, (hipStream_t)stream>>>(n, din, hid, dout, x, w1, b1, w2, b2, out);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

__global__ __launch_bounds__(TB) void k_residual(int64_t N, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                 const float* __restrict__ val, const float* __restrict__ u,
                                                 const float* __restrict__ y, float* __restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x;
  if (r >= N) return;
  float s = 0.f;
  for (int32_t i = ptr[r]; i < ptr[r + 1]; ++i) s = fmaf(val[i], u[col[i]], s);
  out[r] = s - y[r];
}

extern "C" int psignn_residual(const psignn_plan_t* p, const float* u, const float* y, float* out, void* stream) {
  ARG_CHECK(p && u && y && out, "NULL argument");
  k_residual<<<(unsigned)cdiv(p->N, TB), TB, 0, (hipStream_t)stream>>>(p->N, p->a_ptr, p->a_col, p->a_val, u, y, out);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
