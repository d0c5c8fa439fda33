// GMRES on the device for the Newton-Krylov solver (gfx950): Arnoldi with classical Gram-Schmidt applied twice, the
// Hessenberg least-squares problem by Givens rotations, the solution update -- vector sweeps on the same coalesced float4
// kernels as the Broyden solver, the small dense part in one block, nothing on the host but the launch sequence.
//
// Reference: none executable -- dirichlet/psignn/utilities/solver.py:6 imports scipy.optimize.newton_krylov and never calls
// it; BASELINE configs[4] names the Newton-Krylov / JVP path.  The operator is applied by the caller
// (the analytic JVP kernel of the GNN block, csrc/fgnn_tile_jvp.hip): per Arnoldi step it writes A v_j's raw product
// J v_j into basis slot j + 1 and calls psignn_gmres_step, which turns it into the next basis vector:
//   dots pass 1 : w = J v_j (the raw product; the shift enters the Hessenberg's diagonal in `finish`: the Krylov spaces of J and
//                 of J - shift I are the same, and orthogonalising J v_j instead of J v_j - v_j keeps the big -v_j component --
//                 pure cancellation against the basis -- out of the Gram-Schmidt passes), h_i = <v_i, w>, i <= j  (basis read once)
//   axpy pass 1 : w -= sum_i h_i v_i                                                (basis read once)
//   dots pass 2 / axpy pass 2 : the same again on the result ("twice is enough"), with the partials of |w|^2 -- run only when
//                 the first pass cancelled: |w'|^2 < 1/2 |w|^2 (the Daniel-Gragg-Kaufman-Stewart criterion, decided on the
//                 device by k_gm_decide; PSIGNN_GMRES_REORTH=always restores the unconditional second pass)
//   finish      : H[:, j] = h1 + h2 - shift e_j, H[j+1, j] = |w|; previous rotations applied, new rotation, residual |g_{j+1}|;
//                 stop flag when |g_{j+1}| <= eta * beta
//   scale       : v_{j+1} = w / |w|
// = 2 (j + 1) + O(1) vector passes per step, 4 (j + 1) when the second pass runs.  Reductions: fixed-shape partial sums in a fixed order (reproducible).
#include "vec_helpers.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>

struct GmresState {
  int32_t k;          // Arnoldi steps completed
  int32_t done;       // the relative linear residual reached eta (or a breakdown: |w| == 0)
  int32_t breakdown;
  int32_t reorth;     // this step's second Gram-Schmidt pass runs (k_gm_decide)
  double beta;        // |b|
  double resid;       // current |g_{k}| (absolute residual of the least-squares problem)
  double hn;          // |w| of the last step
  double n0sq;        // |w|^2 before the first pass of the current step
  int32_t n_reorth;   // steps of this solve that needed the second pass
  int32_t pad;
};

struct psignn_gmres {
  int64_t M = 0, ld = 0;
  int m = 0;
  int vec = 16, nblk = 0, npart = 0;
  float* V = nullptr;        // caller-owned basis: (m + 1, ld)
  float* part = nullptr;     // (nblk, ldp) dot partials: one value per block and basis vector, vectors contiguous
  int ldp = 0;               // m + 2 rounded up to 64
  float* coef = nullptr;     // (m + 2) coefficients of the current pass (float, like the vectors)
  double* H = nullptr;       // (m + 1, m) column-major: R after the rotations (column j has j + 1 entries) | raw h in work
  double *cs = nullptr, *sn = nullptr, *g = nullptr, *hcol = nullptr, *y = nullptr, *res_hist = nullptr;
  GmresState* st = nullptr;
  GmresState* h_st = nullptr;
  size_t bytes = 0;
};

__global__ void k_gm_init(GmresState* st, double* g, double* res_hist, int m) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->k = 0; st->done = 0; st->breakdown = 0; st->reorth = 1; st->pad = 0; st->beta = 0.0; st->resid = 0.0; st->hn = 0.0;
    st->n0sq = 0.0; st->n_reorth = 0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= m + 1; i += gridDim.x * blockDim.x) {
    g[i] = 0.0;
    res_hist[i] = 0.0;
  }
}

// partials of <a, a> (one per block): part[blockIdx.x]
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_norm2(int64_t M, const float* __restrict__ a, float* __restrict__ part, int npart) {
  int64_t e0 = elem0<VEC>();
  float s = 0.f, z = 0.f;
  if (e0 < M) {
    float x[VEC];
    ldv<VEC>(a, e0, M, x);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = fmaf(x[i], x[i], s);
  }
  block_pair_store(s, z, part, npart);
}

// beta = |b| ; g[0] = beta ; resid = beta
__global__ __launch_bounds__(TB) void k_gm_begin(GmresState* st, const float* __restrict__ part, int nblk, double* __restrict__ g,
                                                 double* __restrict__ res_hist) {
  __shared__ double sh[TB];
  const double s = block_sum_partials(part, nblk, sh);
  if (threadIdx.x == 0) {
    const double beta = (double)(float)sqrt(s);
    st->beta = beta;
    st->resid = beta;
    g[0] = beta;
    res_hist[0] = beta;
    if (!(beta > 0.0)) {   // zero right-hand side: nothing to solve
      st->done = 1;
      st->breakdown = 1;
    }
  }
}

// dst = src * (1 / scale), scale = sqrt of a device double (beta or hn)
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_scale(int64_t M, const float* __restrict__ src, float* __restrict__ dst,
                                                 const double* __restrict__ scale, const GmresState* __restrict__ st, int gate) {
  if (gate && st->done) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  const float inv = 1.f / (float)(*scale);
  float x[VEC];
  ldv<VEC>(src, e0, M, x);
#pragma unroll
  for (int i = 0; i < VEC; ++i) x[i] *= inv;
  stv<VEC>(dst, e0, M, x);
}

// dots pass: optional first transform w <- w - shift * v_j (stored back); per-BLOCK partials of <v_i, w>, i <= j, written as
// coalesced rows part[block * ldp + i] (vec_helpers.h PairStash: round 2's one 4-byte store per wave and basis vector cost the
// sweep 8 % of its rate).  pass 0 also leaves the partials of |w|^2 in column j + 1; pass 1 returns at once unless st->reorth
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_dots(int64_t M, int64_t ld, int j, float shift, const GmresState* __restrict__ st,
                                                const float* __restrict__ V, float* __restrict__ w, float* __restrict__ part,
                                                int ldp, int pass) {
  __shared__ PairStash<1> sh;
  if (st->done || (pass && !st->reorth)) return;
  int64_t e0 = elem0<VEC>();
  const bool act = e0 < M;
  float x[VEC];
  if (act) {
    ldv<VEC>(w, e0, M, x);
    if (shift != 0.f) {
      float v[VEC];
      ldv<VEC>(V + (int64_t)j * ld, e0, M, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = fmaf(-shift, v[i], x[i]);
      stv<VEC>(w, e0, M, x);
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] = 0.f;
  }
  const int wv = threadIdx.x >> 6;
  const bool lead = (threadIdx.x & 63) == 0;
  float* rowb = part + (int64_t)blockIdx.x * ldp;
  const int last = pass == 0 ? j + 1 : j;     // columns 0 .. last: the j + 1 dot products (+ |w|^2 in pass 0)
  for (int i = 0; i <= last; ++i) {
    float s = 0.f;
    if (i <= j) {
      if (act) {
        float v[VEC];
        ldv_stream<VEC>(V + (int64_t)i * ld, e0, M, v);
#pragma unroll
        for (int c = 0; c < VEC; ++c) s = fmaf(v[c], x[c], s);
      }
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) s = fmaf(x[c], x[c], s);
    }
    s = wave_sum(s);
    const int q = i & 63;
    if (lead) sh.v[0][wv][q] = s;
    if (q == 63 || i == last) {
      float* const rows[1] = {rowb + (i - q)};
      stash_flush<1>(sh, q + 1, rows);
    }
  }
}

// one block per coefficient: coef[i] = sum over the blocks of column i of the partials (rounded to float like the vectors they scale)
// (pass 0: one more block, i = n_coef, sums the |w|^2 column into st->n0sq)
__global__ __launch_bounds__(TB) void k_gm_reduce(GmresState* __restrict__ st, const float* __restrict__ part, int nrows, int ldp,
                                                  float* __restrict__ coef, double* __restrict__ hcol, int accumulate, int n_coef) {
  __shared__ double sh[TB];
  if (st->done || (accumulate && !st->reorth)) return;
  const int i = blockIdx.x;
  const double s = block_sum_col(part + i, nrows, ldp, sh);
  if (i == n_coef) {
    if (threadIdx.x == 0) st->n0sq = s;
    return;
  }
  if (threadIdx.x == 0) {
    const float c = (float)s;
    coef[i] = c;
    hcol[i] = accumulate ? hcol[i] + (double)c : (double)c;
  }
}

// axpy pass: w -= sum_{i <= j} coef[i] v_i ; partials of |w|^2 (one per block)
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_axpy(int64_t M, int64_t ld, int j, const GmresState* __restrict__ st,
                                                const float* __restrict__ V, float* __restrict__ w,
                                                const float* __restrict__ coef, float* __restrict__ npartial, int nblk, int pass) {
  if (st->done || (pass && !st->reorth)) return;
  int64_t e0 = elem0<VEC>();
  float s = 0.f, z = 0.f;
  if (e0 < M) {
    float x[VEC];
    ldv<VEC>(w, e0, M, x);
    for (int i = 0; i <= j; ++i) {
      const float c = coef[i];
      float v[VEC];
      ldv_stream<VEC>(V + (int64_t)i * ld, e0, M, v);
#pragma unroll
      for (int q = 0; q < VEC; ++q) x[q] = fmaf(-c, v[q], x[q]);
    }
    stv<VEC>(w, e0, M, x);
#pragma unroll
    for (int q = 0; q < VEC; ++q) s = fmaf(x[q], x[q], s);
  }
  block_pair_store(s, z, npartial, nblk);
}

// One block, between the passes: does the first pass's result need the second one?  |w'|^2 < 1/2 |w|^2 (DGKS); `always` != 0: yes.
__global__ __launch_bounds__(TB) void k_gm_decide(GmresState* st, const float* __restrict__ npartial, int nblk, int always) {
  __shared__ double sh[TB];
  if (st->done) return;
  const double n1sq = block_sum_partials(npartial, nblk, sh);
  if (threadIdx.x == 0) {
    const int r = always || !(n1sq >= 0.5 * st->n0sq);   // (NaN -> reorthogonalise)
    st->reorth = r;
    st->n_reorth += r;
  }
}

// One block: finish column j of the Hessenberg matrix and the least-squares update.
__global__ __launch_bounds__(TB) void k_gm_finish(GmresState* st, const float* __restrict__ npartial, int nblk, int j, int m,
                                                  double* __restrict__ H, double* __restrict__ cs, double* __restrict__ sn,
                                                  double* __restrict__ g, const double* __restrict__ hcol,
                                                  double* __restrict__ res_hist, double eta, double shift) {
  __shared__ double sh[TB];
  if (st->done) return;
  const double s2 = block_sum_partials(npartial, nblk, sh);
  if (threadIdx.x != 0) return;
  const double hn = (double)(float)sqrt(s2);
  st->hn = hn;
  double* col = H + (int64_t)j * (m + 1);
  for (int i = 0; i <= j; ++i) col[i] = hcol[i];
  col[j] -= shift;   // Hessenberg of J - shift I from the Arnoldi relation of J
  col[j + 1] = hn;
  for (int i = 0; i < j; ++i) {   // previous rotations
    const double t = cs[i] * col[i] + sn[i] * col[i + 1];
    col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1];
    col[i] = t;
  }
  const double a = col[j], b = col[j + 1];
  const double r = sqrt(a * a + b * b);
  const double c = r > 0.0 ? a / r : 1.0, s = r > 0.0 ? b / r : 0.0;
  cs[j] = c;
  sn[j] = s;
  col[j] = r;
  col[j + 1] = 0.0;
  g[j + 1] = -s * g[j];
  g[j] = c * g[j];
  const double resid = fabs(g[j + 1]);
  st->resid = resid;
  st->k = j + 1;
  res_hist[j + 1] = resid;
  if (resid <= eta * st->beta) st->done = 1;
  if (!(hn > 0.0)) {   // lucky breakdown: the Krylov space is invariant, the least-squares solution is exact
    st->done = 1;
    st->breakdown = 1;
  }
}

// One block: y = R^{-1} g for the first k columns
__global__ void k_gm_backsolve(const GmresState* __restrict__ st, int k_override, int m, const double* __restrict__ H,
                               const double* __restrict__ g, double* __restrict__ y, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int k = k_override > 0 ? min(k_override, st->k) : st->k;
  for (int i = k - 1; i >= 0; --i) {
    double s = g[i];
    for (int q = i + 1; q < k; ++q) s -= H[(int64_t)q * (m + 1) + i] * y[q];
    const double d = H[(int64_t)i * (m + 1) + i];
    y[i] = d != 0.0 ? s / d : 0.0;
  }
  for (int i = 0; i < k; ++i) coef[i] = (float)y[i];
  for (int i = k; i <= m; ++i) coef[i] = 0.f;
}

// dst = base + scale * sum_{i < k} coef[i] v_i   (base may be NULL: dst = the combination)
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_combine(int64_t M, int64_t ld, const GmresState* __restrict__ st, int k_override,
                                                   const float* __restrict__ V, const float* __restrict__ coef,
                                                   const float* __restrict__ base, float scale, float* __restrict__ dst) {
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  const int k = k_override > 0 ? min(k_override, st->k) : st->k;
  float acc[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
  for (int i = 0; i < k; ++i) {
    const float c = coef[i];
    float v[VEC];
    ldv_stream<VEC>(V + (int64_t)i * ld, e0, M, v);
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = fmaf(c, v[q], acc[q]);
  }
  if (base) {
    float b[VEC];
    ldv<VEC>(base, e0, M, b);
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = fmaf(scale, acc[q], b[q]);
  } else {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] *= scale;
  }
  stv<VEC>(dst, e0, M, acc);
}

// g = fx - x ; partials of |g|^2 and |fx|^2 (one pair per block); optionally b = -g
template <int VEC>
__global__ __launch_bounds__(TB) void k_gm_residual(int64_t M, const float* __restrict__ x, const float* __restrict__ fx,
                                                    float* __restrict__ gout, float* __restrict__ neg_out,
                                                    float* __restrict__ part, int nblk) {
  int64_t e0 = elem0<VEC>();
  float sg = 0.f, sf = 0.f;
  if (e0 < M) {
    float a[VEC], b[VEC];
    ldv<VEC>(x, e0, M, a);
    ldv<VEC>(fx, e0, M, b);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sf = fmaf(b[i], b[i], sf);
      b[i] -= a[i];
      sg = fmaf(b[i], b[i], sg);
    }
    if (gout) stv<VEC>(gout, e0, M, b);
    if (neg_out) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) b[i] = -b[i];
      stv<VEC>(neg_out, e0, M, b);
    }
  }
  block_pair_store(sg, sf, part, nblk);
}
__global__ __launch_bounds__(TB) void k_gm_norms_out(const float* __restrict__ part, int nblk, double* __restrict__ out2) {
  __shared__ double sh[TB];
  const double sg = block_sum_partials(part, nblk, sh);
  const double sf = block_sum_partials(part + nblk, nblk, sh);
  if (threadIdx.x == 0) {
    out2[0] = (double)(float)sqrt(sg);
    out2[1] = (double)(float)sqrt(sf);
  }
}

// ------------------------------------------------------------------------------------------ host
extern "C" void psignn_gmres_destroy(psignn_gmres_t* s) {
  if (!s) return;
  void* ptrs[] = {s->part, s->coef, s->H, s->cs, s->sn, s->g, s->hcol, s->y, s->res_hist, s->st};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (s->h_st) (void)hipHostFree(s->h_st);
  delete s;
}

extern "C" int psignn_gmres_create(psignn_gmres_t** out, int64_t n_elems, int64_t ld, int m_max, float* d_basis) {
  ARG_CHECK(out, "out is NULL");
  *out = nullptr;
  ARG_CHECK(n_elems > 0 && m_max > 0 && d_basis, "bad arguments");
  ARG_CHECK(ld >= n_elems && ld % 4 == 0, "row pitch must be >= n_elems and a multiple of 4 floats");
  psignn_gmres* s = new psignn_gmres();
  s->M = n_elems;
  s->ld = ld;
  s->m = m_max;
  s->V = d_basis;
  s->vec = n_elems >= ((int64_t)3 << 18) ? 16 : 4;
  s->nblk = (int)cdiv(n_elems, (int64_t)s->vec * TB);
  s->npart = s->nblk * (TB / 64);
  s->ldp = (m_max + 2 + 63) / 64 * 64;
  const size_t m = (size_t)m_max;
  struct { void** p; size_t n; } allocs[] = {
      {(void**)&s->part, (size_t)s->nblk * s->ldp * 4 + 16}, {(void**)&s->coef, (m + 2) * 4 + 16},
      {(void**)&s->H, (m + 1) * (m + 1) * 8}, {(void**)&s->cs, (m + 2) * 8}, {(void**)&s->sn, (m + 2) * 8},
      {(void**)&s->g, (m + 3) * 8}, {(void**)&s->hcol, (m + 2) * 8}, {(void**)&s->y, (m + 2) * 8},
      {(void**)&s->res_hist, (m + 3) * 8}, {(void**)&s->st, sizeof(GmresState)}};
  for (auto& a : allocs) {
    if (hipMalloc(a.p, a.n) != hipSuccess) {
      psignn_set_error("gmres: hipMalloc of %zu bytes failed", a.n);
      psignn_gmres_destroy(s);
      return PSIGNN_ENOMEM;
    }
    s->bytes += a.n;
  }
  if (hipHostMalloc((void**)&s->h_st, sizeof(GmresState)) != hipSuccess) {
    psignn_set_error("gmres: hipHostMalloc failed");
    psignn_gmres_destroy(s);
    return PSIGNN_ENOMEM;
  }
  *out = s;
  return PSIGNN_OK;
}

// g = fx - x (d_g, may be NULL), b = -g (d_neg_g, may be NULL), h_norms[0] = |g|, h_norms[1] = |fx| (synchronous read)
extern "C" int psignn_residual_norms(psignn_gmres_t* s, const float* d_x, const float* d_fx, float* d_g, float* d_neg_g,
                                     double* h_norms, void* stream) {
  ARG_CHECK(s && d_x && d_fx && h_norms, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  VLAUNCH("k_gm_residual", st, s->vec, k_gm_residual, ((unsigned)s->nblk, TB, 0, st), s->M, d_x, d_fx, d_g, d_neg_g, s->part, s->nblk);
  k_gm_norms_out<<<1, TB, 0, st>>>(s->part, s->nblk, s->y);
  HIP_TRY(hipMemcpyAsync(h_norms, s->y, 16, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return PSIGNN_OK;
}

// Start a solve of A z = b: beta = |b|, v_0 = b / beta.
extern "C" int psignn_gmres_begin(psignn_gmres_t* s, const float* d_b, void* stream) {
  ARG_CHECK(s && d_b, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)s->nblk;
  k_gm_init<<<4, TB, 0, st>>>(s->st, s->g, s->res_hist, s->m);
  VPLAIN(s->vec, k_gm_norm2, (g, TB, 0, st), s->M, d_b, s->part, s->nblk);
  k_gm_begin<<<1, TB, 0, st>>>(s->st, s->part, s->nblk, s->g, s->res_hist);
  VPLAIN(s->vec, k_gm_scale, (g, TB, 0, st), s->M, d_b, s->V, &s->st->beta, s->st, 0);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Arnoldi step j: basis slot j + 1 holds the caller's raw product P = (operator applied to v_j); the Krylov operator is
// A v = P - shift * v (shift = 1 for A = J_f - I).  h_done (may be NULL): synchronous read of the stop flag.
extern "C" int psignn_gmres_step(psignn_gmres_t* s, int j, double shift, double eta, int* h_done, void* stream) {
  ARG_CHECK(s && j >= 0 && j < s->m, "step index outside the basis");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)s->nblk;
  float* w = s->V + (size_t)(j + 1) * s->ld;
  KNOB_INT(always, [] { const char* e = getenv("PSIGNN_GMRES_REORTH"); return e && strcmp(e, "always") == 0 ? 1 : 0; }());
  for (int pass = 0; pass < 2; ++pass) {
    VLAUNCH("k_gm_dots", st, s->vec, k_gm_dots, (g, TB, 0, st), s->M, s->ld, j, 0.f, s->st, s->V, w, s->part, s->ldp, pass);
    LAUNCH("k_gm_reduce", st, (k_gm_reduce<<<(unsigned)(j + 1 + (pass == 0)), TB, 0, st>>>(s->st, s->part, s->nblk, s->ldp, s->coef, s->hcol, pass, j + 1)));
    VLAUNCH("k_gm_axpy", st, s->vec, k_gm_axpy, (g, TB, 0, st), s->M, s->ld, j, s->st, s->V, w, s->coef, s->part, s->nblk, pass);
    if (pass == 0) LAUNCH("k_gm_decide", st, (k_gm_decide<<<1, TB, 0, st>>>(s->st, s->part, s->nblk, always)));
  }
  LAUNCH("k_gm_finish", st, (k_gm_finish<<<1, TB, 0, st>>>(s->st, s->part, s->nblk, j, s->m, s->H, s->cs, s->sn, s->g, s->hcol, s->res_hist, eta, shift)));
  VLAUNCH("k_gm_scale", st, s->vec, k_gm_scale, (g, TB, 0, st), s->M, w, w, &s->st->hn, s->st, 1);
  if (h_done) {
    HIP_TRY(hipMemcpyAsync(s->h_st, s->st, sizeof(GmresState), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *h_done = s->h_st->done;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// d_dst = d_base + scale * V y (y from the first k columns; k <= 0: all completed steps).  h_info: [k, beta, resid] (may be NULL)
extern "C" int psignn_gmres_solution(psignn_gmres_t* s, int k, const float* d_base, double scale, float* d_dst, double* h_info,
                                     void* stream) {
  ARG_CHECK(s && d_dst, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  k_gm_backsolve<<<1, 64, 0, st>>>(s->st, k, s->m, s->H, s->g, s->y, s->coef);
  VLAUNCH("k_gm_combine", st, s->vec, k_gm_combine, ((unsigned)s->nblk, TB, 0, st), s->M, s->ld, s->st, k, s->V, s->coef, d_base, (float)scale, d_dst);
  if (h_info) {
    HIP_TRY(hipMemcpyAsync(s->h_st, s->st, sizeof(GmresState), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h_info[0] = (double)s->h_st->k;
    h_info[1] = s->h_st->beta;
    h_info[2] = s->h_st->resid;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// number of Arnoldi steps of the current solve whose second Gram-Schmidt pass ran (synchronous read)
extern "C" int psignn_gmres_reorth_count(psignn_gmres_t* s, int* h_count, void* stream) {
  ARG_CHECK(s && h_count, "NULL argument");
  HIP_TRY(hipMemcpyAsync(s->h_st, s->st, sizeof(GmresState), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  *h_count = s->h_st->n_reorth;
  return PSIGNN_OK;
}

// residual history of the last solve: h_res[0] = beta, h_res[i] = |residual| after i steps (m + 1 doubles)
extern "C" int psignn_gmres_history(psignn_gmres_t* s, double* h_res, void* stream) {
  ARG_CHECK(s && h_res, "NULL argument");
  HIP_TRY(hipMemcpyAsync(h_res, s->res_hist, (size_t)(s->m + 1) * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return PSIGNN_OK;
}
