// Picard iteration and Anderson acceleration on the device (gfx950): the vector work, norms, stop tests and the small
// bordered solve of `forward_iteration` / `anderson` (reference: dirichlet/psignn/utilities/solver.py:301-341 and :215-293).
//
// The caller owns the map f: per iteration it asks for the next trial point, evaluates f (the HIP GNN block or any device
// callable) and hands f(x) back.  Everything else stays here, with a device-side status block like the Broyden solver's:
// no host read per iteration unless the caller asks for one, kernels return at once after the stop test has fired.
//
// Anderson (m stored pairs, ring slots X[i], F[i] = f(X[i]); the reference's loop for k = 2 .. threshold-1, n = min(k, m)):
//   gram   : partials of G_i . G_j, G_i = F_i - X_i, i <= j < n                      one pass over 2 n vectors
//   solve  : [[0, 1^T], [1, G G^T + lam I]] [nu; alpha] = [1; 0]   (one block, float64, partial pivoting)
//   mix    : X[k % m] = beta * sum_i alpha_i F_i + (1 - beta) * sum_i alpha_i X_i   one pass
//   (caller: F[k % m] = f(X[k % m]))
//   norms  : partials of |F - X|^2, |F|^2 ; check: rel = abs / (1e-5 + |F|), traces, lowest iterate, stop at rel < eps
// Picard: x <- f(x) with abs = |x - f(x)|, rel = abs / |f(x)| after every evaluation, stop when rel <= eps.
// Reductions use fixed-shape partial sums in a fixed order: bitwise reproducible.
#include "vec_helpers.h"
#include <algorithm>

#define FP_MAX_M 8                                   // history length of Anderson (the reference uses m = 2)
#define FP_NPAIR (FP_MAX_M * (FP_MAX_M + 1) / 2)

struct FpStatus {
  int32_t n_iter;        // loop iterations done (Anderson: k - 2 + 1 after step k; Picard: f evaluations after the first)
  int32_t done;
  int32_t stop_reason;   // 0 threshold, 1 tolerance
  int32_t lowest_step, lowest_step_alt;
  int32_t new_low;       // the iterate just evaluated is the lowest so far (stop_mode's objective)
  int32_t k;             // Anderson: loop index k of the NEXT step; Picard: index of the current iterate
  int32_t stop_abs;
  double lowest, lowest_alt;   // lowest objective in stop_mode / in the other mode
  double alpha[FP_MAX_M];
};

struct psignn_fpiter {
  int64_t M = 0, ld = 0;
  int m = 2, thr = 0, keep_trace = 0;
  int vec = 16, nblk = 0, npart = 0;
  float *X = nullptr, *F = nullptr;     // (m, ld) ring slots
  float* low = nullptr;                 // lowest iterate (Anderson)
  float* trace = nullptr;               // keep_trace: (thr + 2, ld) every iterate
  float* part = nullptr;                // (FP_NPAIR, npart) partials
  FpStatus* st = nullptr;
  FpStatus* h_st = nullptr;             // pinned
  double *rel_trace = nullptr, *abs_trace = nullptr;   // thr entries
  int32_t* low_idx = nullptr;           // Anderson: per loop iteration, trace index of the lowest iterate so far
  double lam = 1e-4, beta = 1.0;
  int kind = 0;                         // 1 picard, 2 anderson
  int host_k = 0;                       // Anderson: loop index of the step being driven; Picard: evaluations handed in
  size_t bytes = 0;
};

__global__ void k_fp_init(FpStatus* st, double* rel_trace, double* abs_trace, int32_t* low_idx, int thr, int stop_abs, int k0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->n_iter = 0; st->done = 0; st->stop_reason = 0; st->lowest_step = 0; st->lowest_step_alt = 0; st->new_low = 0;
    st->k = k0; st->stop_abs = stop_abs; st->lowest = 1e8; st->lowest_alt = 1e8;
    for (int i = 0; i < FP_MAX_M; ++i) st->alpha[i] = 0.0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < thr; i += gridDim.x * blockDim.x) {
    rel_trace[i] = 0.0;
    abs_trace[i] = 0.0;
    low_idx[i] = 0;
  }
}

// dst = src (plain vector copy; optionally only when *flag != 0)
template <int VEC>
__global__ __launch_bounds__(TB) void k_fp_copy(int64_t M, const float* __restrict__ src, float* __restrict__ dst,
                                                const int32_t* __restrict__ flag, const int32_t* __restrict__ done) {
  if (done && *done) return;
  if (flag && !*flag) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  float a[VEC];
  ldv<VEC>(src, e0, M, a);
  stv<VEC>(dst, e0, M, a);
}

// partials of |fx - x|^2 and |fx|^2, ONE pair per block (npart = number of blocks); optionally stores fx into `keep`
template <int VEC>
__global__ __launch_bounds__(TB) void k_fp_norms(int64_t M, const FpStatus* __restrict__ st, const float* __restrict__ x,
                                                 const float* __restrict__ fx, float* __restrict__ keep,
                                                 float* __restrict__ part, int npart) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  float sg = 0.f, sf = 0.f;
  if (e0 < M) {
    float a[VEC], b[VEC];
    ldv<VEC>(x, e0, M, a);
    ldv<VEC>(fx, e0, M, b);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float g = b[i] - a[i];
      sg = fmaf(g, g, sg);
      sf = fmaf(b[i], b[i], sf);
    }
    if (keep) stv<VEC>(keep, e0, M, b);
  }
  block_pair_store(sg, sf, part, npart);
}

// Picard bookkeeping (solver.py:313-331): trace entry, stop when rel <= eps or after `thr` further evaluations.
__global__ __launch_bounds__(TB) void k_picard_check(FpStatus* st, const float* __restrict__ part, int npart,
                                                     double* __restrict__ rel_trace, double* __restrict__ abs_trace, double eps,
                                                     int thr) {
  __shared__ double sh[TB];
  if (st->done) return;
  const double sg = block_sum_partials(part, npart, sh);
  const double sf = block_sum_partials(part + npart, npart, sh);
  if (threadIdx.x != 0) return;
  // torch.linalg.norm values are fp32 tensors; their quotient is an fp32 division (solver.py:315-316)
  const float a = (float)sqrt(sg), r = a / (float)sqrt(sf);
  const int i = st->n_iter;            // entry index: 0 for f(z0), then one per loop pass
  rel_trace[i] = (double)r;
  abs_trace[i] = (double)a;
  st->n_iter = i + 1;
  st->lowest = (double)r;              // "lowest" of forward_iteration is the LAST relative residual
  st->k = i + 1;                       // the current iterate is now z_{i+1} = f(z_i)
  if (!((double)r > eps)) {
    st->done = 1;
    st->stop_reason = 1;
  } else if (i >= thr) {               // `ite < threshold` failed: threshold loop passes were made
    st->done = 1;
    st->stop_reason = 0;
  }
}

// Anderson: gram partials.  Pair index p = j (j + 1) / 2 + i for i <= j.
template <int VEC>
__global__ __launch_bounds__(TB) void k_and_gram(int64_t M, int64_t ld, int n, const FpStatus* __restrict__ st,
                                                 const float* __restrict__ X, const float* __restrict__ F,
                                                 float* __restrict__ part, int npart) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  const bool act = e0 < M;
  float acc[FP_NPAIR];
#pragma unroll
  for (int p = 0; p < FP_NPAIR; ++p) acc[p] = 0.f;
  if (act) {
    // n <= FP_MAX_M rows of VEC floats would not fit in registers for n = 8, VEC = 16: walk the span in float4 pieces
#pragma unroll 1
    for (int q = 0; q < VEC / 4; ++q) {
      const int64_t o = e0 + q * 256;
      float g[FP_MAX_M][4];
#pragma unroll
      for (int i = 0; i < FP_MAX_M; ++i) {
        if (i < n) {
          if (o + 4 <= M) {   // rows start on 256-byte boundaries (ld is a multiple of 64): aligned float4
            const float4 fv = *reinterpret_cast<const float4*>(F + (int64_t)i * ld + o);
            const float4 xv = *reinterpret_cast<const float4*>(X + (int64_t)i * ld + o);
            g[i][0] = fv.x - xv.x; g[i][1] = fv.y - xv.y; g[i][2] = fv.z - xv.z; g[i][3] = fv.w - xv.w;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) g[i][c] = (o + c < M) ? F[(int64_t)i * ld + o + c] - X[(int64_t)i * ld + o + c] : 0.f;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) g[i][c] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < FP_MAX_M; ++j)
#pragma unroll
        for (int i = 0; i <= j; ++i) {
          float s = acc[j * (j + 1) / 2 + i];
#pragma unroll
          for (int c = 0; c < 4; ++c) s = fmaf(g[i][c], g[j][c], s);
          acc[j * (j + 1) / 2 + i] = s;
        }
    }
  }
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const int npairs = n * (n + 1) / 2;
#pragma unroll
  for (int p = 0; p < FP_NPAIR; ++p) {
    if (p < npairs) {   // wave-uniform
      const float s = wave_sum(acc[p]);
      if ((threadIdx.x & 63) == 0) part[(int64_t)p * npart + w] = s;
    }
  }
}

// One block: alpha of the bordered system (solver.py:252-256) in float64.
__global__ __launch_bounds__(TB) void k_and_solve(FpStatus* st, const float* __restrict__ part, int npart, int n, double lam) {
  __shared__ double sh[TB];
  __shared__ double gram[FP_NPAIR];
  if (st->done) return;
  const int npairs = n * (n + 1) / 2;
  for (int p = 0; p < npairs; ++p) {
    const double s = block_sum_partials(part + (int64_t)p * npart, npart, sh);
    if (threadIdx.x == 0) gram[p] = (double)(float)s;   // torch.bmm result is fp32
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const int d = n + 1;
  double A[(FP_MAX_M + 1) * (FP_MAX_M + 2)];   // augmented [H | y], row-major, d x (d + 1)
  for (int r = 0; r < d; ++r)
    for (int c = 0; c <= d; ++c) A[r * (d + 1) + c] = 0.0;
  for (int i = 1; i < d; ++i) A[0 * (d + 1) + i] = A[i * (d + 1) + 0] = 1.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      const double v = gram[j * (j + 1) / 2 + i];
      A[(i + 1) * (d + 1) + (j + 1)] = v;
      A[(j + 1) * (d + 1) + (i + 1)] = v;
    }
  for (int i = 0; i < n; ++i) A[(i + 1) * (d + 1) + (i + 1)] += lam;
  A[0 * (d + 1) + d] = 1.0;
  for (int c = 0; c < d; ++c) {   // Gaussian elimination with partial pivoting
    int piv = c;
    for (int r = c + 1; r < d; ++r)
      if (fabs(A[r * (d + 1) + c]) > fabs(A[piv * (d + 1) + c])) piv = r;
    if (piv != c)
      for (int q = 0; q <= d; ++q) {
        const double t = A[c * (d + 1) + q];
        A[c * (d + 1) + q] = A[piv * (d + 1) + q];
        A[piv * (d + 1) + q] = t;
      }
    const double pv = A[c * (d + 1) + c];
    for (int r = c + 1; r < d; ++r) {
      const double f = A[r * (d + 1) + c] / pv;
      for (int q = c; q <= d; ++q) A[r * (d + 1) + q] -= f * A[c * (d + 1) + q];
    }
  }
  double sol[FP_MAX_M + 1];
  for (int r = d - 1; r >= 0; --r) {
    double s = A[r * (d + 1) + d];
    for (int q = r + 1; q < d; ++q) s -= A[r * (d + 1) + q] * sol[q];
    sol[r] = s / A[r * (d + 1) + r];
  }
  for (int i = 0; i < n; ++i) st->alpha[i] = (double)(float)sol[i + 1];
}

// X[slot] = beta * sum alpha_i F_i + (1 - beta) * sum alpha_i X_i ; also written to `out` (the caller's trial point)
template <int VEC>
__global__ __launch_bounds__(TB) void k_and_mix(int64_t M, int64_t ld, int n, int slot, const FpStatus* __restrict__ st,
                                                float* __restrict__ X, const float* __restrict__ F, float beta,
                                                float* __restrict__ out, float* __restrict__ trace_dst) {
  if (st->done) return;
  int64_t e0 = elem0<VEC>();
  if (e0 >= M) return;
  float accf[VEC], accx[VEC], t[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) accf[i] = accx[i] = 0.f;
  for (int i = 0; i < n; ++i) {
    const float a = (float)st->alpha[i];
    ldv<VEC>(F + (int64_t)i * ld, e0, M, t);
#pragma unroll
    for (int c = 0; c < VEC; ++c) accf[c] = fmaf(a, t[c], accf[c]);
    if (beta != 1.f) {
      ldv<VEC>(X + (int64_t)i * ld, e0, M, t);
#pragma unroll
      for (int c = 0; c < VEC; ++c) accx[c] = fmaf(a, t[c], accx[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < VEC; ++c) accf[c] = beta * accf[c] + (1.f - beta) * accx[c];
  stv<VEC>(X + (int64_t)slot * ld, e0, M, accf);
  if (out) stv<VEC>(out, e0, M, accf);
  if (trace_dst) stv<VEC>(trace_dst, e0, M, accf);
}

// Anderson bookkeeping (solver.py:262-283).
__global__ __launch_bounds__(TB) void k_and_check(FpStatus* st, const float* __restrict__ part, int npart,
                                                  double* __restrict__ rel_trace, double* __restrict__ abs_trace,
                                                  int32_t* __restrict__ low_idx, double eps, int thr, int k) {
  __shared__ double sh[TB];
  if (st->done) {
    // The host runs ahead of the stop by up to poll_every - 1 steps.  The stopping step left new_low = 1 (a tolerance stop is
    // always a new lowest objective): without this reset every later psignn_anderson_update would copy its stale X[slot] over
    // the lowest iterate, and the solve would return iterate k* - r instead of k*.
    if (threadIdx.x == 0) st->new_low = 0;
    return;
  }
  const double sg = block_sum_partials(part, npart, sh);
  const double sf = block_sum_partials(part + npart, npart, sh);
  if (threadIdx.x != 0) return;
  // .norm().item() values are fp32 numbers read back; the quotient is formed in Python doubles (solver.py:262-263)
  const double abs_diff = (double)(float)sqrt(sg);
  const double rel_diff = abs_diff / (1e-5 + (double)(float)sqrt(sf));
  const int i = st->n_iter;
  rel_trace[i] = rel_diff;
  abs_trace[i] = abs_diff;
  st->n_iter = i + 1;
  const bool stop_abs = st->stop_abs != 0;
  const double obj = stop_abs ? abs_diff : rel_diff, alt = stop_abs ? rel_diff : abs_diff;
  st->new_low = 0;
  if (obj < st->lowest) {
    st->lowest = obj;
    st->lowest_step = k;
    st->new_low = 1;
  }
  if (alt < st->lowest_alt) {
    st->lowest_alt = alt;
    st->lowest_step_alt = k;
  }
  low_idx[i] = st->lowest_step;
  st->k = k + 1;
  if (obj < eps) {
    st->done = 1;
    st->stop_reason = 1;
  } else if (k + 1 >= thr) {
    st->done = 1;
    st->stop_reason = 0;
  }
}

// ------------------------------------------------------------------------------------------ host
extern "C" void psignn_fpiter_destroy(psignn_fpiter_t* s) {
  if (!s) return;
  void* ptrs[] = {s->X, s->F, s->low, s->trace, s->part, s->st, s->rel_trace, s->abs_trace, s->low_idx};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (s->h_st) (void)hipHostFree(s->h_st);
  delete s;
}

extern "C" int psignn_fpiter_create(psignn_fpiter_t** out, int64_t n_elems, int m, int threshold, int keep_trace) {
  ARG_CHECK(out, "out is NULL");
  *out = nullptr;
  ARG_CHECK(n_elems > 0 && threshold > 0, "bad sizes");
  ARG_CHECK(m >= 1 && m <= FP_MAX_M, "history length m must be 1..8");
  psignn_fpiter* s = new psignn_fpiter();
  s->M = n_elems;
  s->m = m;
  s->thr = threshold;
  s->keep_trace = keep_trace;
  s->vec = n_elems >= ((int64_t)3 << 18) ? 16 : 4;
  s->nblk = (int)cdiv(n_elems, (int64_t)s->vec * TB);
  s->npart = s->nblk * (TB / 64);
  s->ld = (n_elems + 63) / 64 * 64;
  const size_t ld = (size_t)s->ld, thr = (size_t)threshold;
  struct { void** p; size_t n; } allocs[] = {
      {(void**)&s->X, (size_t)m * ld * 4}, {(void**)&s->F, (size_t)m * ld * 4}, {(void**)&s->low, ld * 4},
      {(void**)&s->trace, keep_trace ? (thr + 2) * ld * 4 : 0}, {(void**)&s->part, (size_t)FP_NPAIR * s->npart * 4 + 16},
      {(void**)&s->st, sizeof(FpStatus)}, {(void**)&s->rel_trace, (thr + 2) * 8}, {(void**)&s->abs_trace, (thr + 2) * 8},
      {(void**)&s->low_idx, (thr + 2) * 4}};
  for (auto& a : allocs) {
    if (a.n == 0) continue;
    if (hipMalloc(a.p, a.n) != hipSuccess) {
      psignn_set_error("fpiter: hipMalloc of %zu bytes failed", a.n);
      psignn_fpiter_destroy(s);
      return PSIGNN_ENOMEM;
    }
    s->bytes += a.n;
  }
  if (hipHostMalloc((void**)&s->h_st, sizeof(FpStatus)) != hipSuccess) {
    psignn_set_error("fpiter: hipHostMalloc failed");
    psignn_fpiter_destroy(s);
    return PSIGNN_ENOMEM;
  }
  *out = s;
  return PSIGNN_OK;
}

extern "C" size_t psignn_fpiter_bytes(const psignn_fpiter_t* s) { return s ? s->bytes : 0; }

static int fp_read_status(psignn_fpiter* s, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(s->h_st, s->st, sizeof(FpStatus), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

extern "C" int psignn_fpiter_poll(psignn_fpiter_t* s, int* h_done, void* stream) {
  ARG_CHECK(s && h_done, "NULL argument");
  int rc = fp_read_status(s, (hipStream_t)stream);
  if (rc) return rc;
  *h_done = s->h_st->done;
  return PSIGNN_OK;
}

// ---- Picard ---------------------------------------------------------------------------------------------------------
// Ring use: X[0] = current iterate z_i.  psignn_picard_update(fx = f(z_i)): norms of (z_i, fx), trace entry i, stop test,
// z_{i+1} = fx.
extern "C" int psignn_picard_begin(psignn_fpiter_t* s, const float* d_x0, void* stream) {
  ARG_CHECK(s && d_x0, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  s->kind = 1;
  s->host_k = 0;
  k_fp_init<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->low_idx, s->thr + 2, 0, 0);
  VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, st), s->M, d_x0, s->X, nullptr, nullptr);
  if (s->keep_trace) VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, st), s->M, d_x0, s->trace, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

extern "C" int psignn_picard_update(psignn_fpiter_t* s, const float* d_fx, double eps, int* h_done, void* stream) {
  ARG_CHECK(s && d_fx && s->kind == 1, "picard_update without picard_begin");
  ARG_CHECK(s->host_k <= s->thr, "more evaluations than threshold + 1");
  hipStream_t st = (hipStream_t)stream;
  const int32_t* done = &s->st->done;
  // norms against the current iterate; f(x) is parked in F[0] and becomes the current iterate only if the test did not
  // fire before this evaluation (a caller that runs ahead of the device status may hand in evaluations past the stop)
  VLAUNCH("k_fp_norms", st, s->vec, k_fp_norms, ((unsigned)s->nblk, TB, 0, st), s->M, s->st, s->X, d_fx, s->F, s->part, s->nblk);
  if (s->keep_trace)
    VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, st), s->M, d_fx, s->trace + (size_t)(s->host_k + 1) * s->ld, nullptr, done);
  VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, st), s->M, s->F, s->X, nullptr, done);
  LAUNCH("k_picard_check", st, (k_picard_check<<<1, TB, 0, st>>>(s->st, s->part, s->nblk, s->rel_trace, s->abs_trace, eps, s->thr)));
  s->host_k++;
  if (h_done) {
    int rc = fp_read_status(s, st);
    if (rc) return rc;
    *h_done = s->h_st->done;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Picard: the current iterate z_i -> d_x (the point the caller evaluates f at next)
extern "C" int psignn_picard_current_x(psignn_fpiter_t* s, float* d_x, void* stream) {
  ARG_CHECK(s && d_x && s->kind == 1, "picard_current_x without picard_begin");
  hipStream_t st = (hipStream_t)stream;
  VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, st), s->M, s->X, d_x, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// ---- Anderson -------------------------------------------------------------------------------------------------------
extern "C" int psignn_anderson_begin(psignn_fpiter_t* s, const float* d_x0, const float* d_f0, const float* d_f1, double lam,
                                     double beta, int stop_abs, void* stream) {
  ARG_CHECK(s && d_x0 && d_f0 && d_f1, "NULL argument");
  ARG_CHECK(s->m >= 2, "anderson needs a history of at least 2");
  hipStream_t st = (hipStream_t)stream;
  s->kind = 2;
  s->lam = lam;
  s->beta = beta;
  s->host_k = 2;
  k_fp_init<<<4, TB, 0, st>>>(s->st, s->rel_trace, s->abs_trace, s->low_idx, s->thr + 2, stop_abs, 2);
  const unsigned g = (unsigned)s->nblk;
  // X[0] = x0, F[0] = f(x0), X[1] = F[0], F[1] = f(F[0])   (solver.py:228-231)
  VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, d_x0, s->X, nullptr, nullptr);
  VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, d_f0, s->F, nullptr, nullptr);
  VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, d_f0, s->X + s->ld, nullptr, nullptr);
  VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, d_f1, s->F + s->ld, nullptr, nullptr);
  HIP_TRY(hipMemsetAsync(s->low, 0, (size_t)s->ld * 4, st));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

extern "C" int psignn_anderson_next_x(psignn_fpiter_t* s, float* d_x_new, void* stream) {
  ARG_CHECK(s && s->kind == 2, "anderson_next_x without anderson_begin");
  ARG_CHECK(s->host_k < s->thr, "loop index reached the threshold");
  hipStream_t st = (hipStream_t)stream;
  const int k = s->host_k, n = std::min(k, s->m), slot = k % s->m;
  const unsigned g = (unsigned)s->nblk;
  VLAUNCH("k_and_gram", st, s->vec, k_and_gram, (g, TB, 0, st), s->M, s->ld, n, s->st, s->X, s->F, s->part, s->npart);
  LAUNCH("k_and_solve", st, (k_and_solve<<<1, TB, 0, st>>>(s->st, s->part, s->npart, n, s->lam)));
  float* tr = s->keep_trace ? s->trace + (size_t)k * s->ld : nullptr;   // trace row k = the iterate of loop index k
  VLAUNCH("k_and_mix", st, s->vec, k_and_mix, (g, TB, 0, st), s->M, s->ld, n, slot, s->st, s->X, s->F, (float)s->beta, d_x_new, tr);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

extern "C" int psignn_anderson_update(psignn_fpiter_t* s, const float* d_fx_new, double eps, int* h_done, void* stream) {
  ARG_CHECK(s && d_fx_new && s->kind == 2, "anderson_update without anderson_begin");
  hipStream_t st = (hipStream_t)stream;
  const int k = s->host_k, slot = k % s->m;
  const unsigned g = (unsigned)s->nblk;
  float* Xs = s->X + (size_t)slot * s->ld;
  float* Fs = s->F + (size_t)slot * s->ld;
  VLAUNCH("k_fp_norms", st, s->vec, k_fp_norms, (g, TB, 0, st), s->M, s->st, Xs, d_fx_new, Fs, s->part, s->nblk);
  LAUNCH("k_and_check", st, (k_and_check<<<1, TB, 0, st>>>(s->st, s->part, s->nblk, s->rel_trace, s->abs_trace, s->low_idx, eps,
                                                           s->thr, k)));
  // lowest_xest = X[k % m].clone() when the objective improved (solver.py:270-272); the check of THIS step may have set done,
  // so the copy is gated on new_low only
  VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, Xs, s->low, &s->st->new_low, nullptr);
  s->host_k++;
  if (h_done) {
    int rc = fp_read_status(s, st);
    if (rc) return rc;
    *h_done = s->h_st->done;
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Result and traces.  Picard: result = the last iterate z (solver.py:333), info.nstep = ite, lowest = last rel.
// Anderson: result = lowest iterate, nstep = its loop index k, lowest / lowest_abs per mode.  h_rel / h_abs: thr + 1 doubles.
// h_low_idx (Anderson, may be NULL): per loop iteration the loop index of the lowest iterate so far (xest_trace).
extern "C" int psignn_fpiter_finish(psignn_fpiter_t* s, float* d_result, psignn_solve_info_t* info, double* h_rel,
                                    double* h_abs, int32_t* h_low_idx, void* stream) {
  ARG_CHECK(s && s->kind != 0, "finish without begin");
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)s->nblk;
  if (d_result) VPLAIN(s->vec, k_fp_copy, (g, TB, 0, st), s->M, s->kind == 1 ? s->X : s->low, d_result, nullptr, nullptr);
  int rc = fp_read_status(s, st);
  if (rc) return rc;
  const FpStatus& h = *s->h_st;
  const int n = h.n_iter;
  if (info) {
    info->n_iter = n;
    info->prot_break = 0;
    info->stop_reason = h.stop_reason;
    if (s->kind == 1) {
      info->nstep = n > 0 ? n - 1 : 0;
      info->lowest = h.lowest;
      info->lowest_abs = 0.0;
    } else {
      info->nstep = h.lowest_step;
      info->lowest = h.stop_abs ? h.lowest_alt : h.lowest;       // lowest rel
      info->lowest_abs = h.stop_abs ? h.lowest : h.lowest_alt;   // lowest abs
    }
  }
  if (h_rel && n > 0) HIP_TRY(hipMemcpy(h_rel, s->rel_trace, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (h_abs && n > 0) HIP_TRY(hipMemcpy(h_abs, s->abs_trace, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (h_low_idx && n > 0) HIP_TRY(hipMemcpy(h_low_idx, s->low_idx, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Iterate i of the last run (needs keep_trace).  Picard: z_i (0 = the start).  Anderson: the trial point of loop index i >= 2.
extern "C" int psignn_fpiter_get_iterate(const psignn_fpiter_t* s, int i, float* d_dst, void* stream) {
  ARG_CHECK(s && d_dst, "NULL argument");
  ARG_CHECK(s->keep_trace, "created without keep_trace");
  ARG_CHECK(i >= 0 && i <= s->thr + 1, "iterate index out of range");
  VPLAIN(s->vec, k_fp_copy, ((unsigned)s->nblk, TB, 0, (hipStream_t)stream), s->M, s->trace + (size_t)i * s->ld, d_dst, nullptr, nullptr);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
