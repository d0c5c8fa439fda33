// Parameter gradient of f_theta:  grad_theta = w^T (d f / d theta)  at a fixed point h  (gfx950, dirichlet family,
// single layer, tiled plan, plan order).
//
// Reference: loss.backward() through  new_H = f(H*, H_init, batch)  with the hooked cotangent (the solution of the
// adjoint system) -- dirichlet/psignn/model.py:203-225, training_class.py:150-163.  Autograd accumulates every
// weight's gradient with one small GEMM per Linear over the (E', 23) / (N, 32) activations.  Here:
//   1. the tiled VJP kernels (fgnn_tile_vjp.hip, PG mode) run as usual and, per node, leave a RECORD of 20 groups of
//      16 floats: the left factors (cotangents at each Linear's output) and right factors (each Linear's input).
//      Edge-level sums collapse to node-level ones:  sum_e dz_e (x) x_i = gt[n] (x) x[n];  sum_e dz_e (x) x_j regrouped
//      by the neighbour = acc[u] (x) x[u] (acc = pass B's neighbour-side sums);  sum_e dz_e (x) a_e = dS (.) moments.
//   2. k_pgrad_outer: every weight gradient is  sum_n A_n (x) B_n  for one (A group, B group) pair -> 16 MFMA
//      accumulator tiles (v_mfma_f32_16x16x4_f32, 4 nodes per instruction, K = nodes).  fp32 MFMA accumulates in k
//      order, a wave owns a fixed node range, partials are combined in a fixed order: bitwise reproducible.
//   3. k_pgrad_reduce: sum of the per-block partial tiles (fp64) and scatter into the flat gradient, laid out like
//      the base section of the packed weights (WLayout: shared | phi_to | phi_from | update).
#include "fgnn_common.h"
#include <stdlib.h>
#include <string.h>

#define PGREC 320
#define PG_TILES 16
typedef float f4 __attribute__((ext_vector_type(4)));

int psignn_f_tile_vjp_rec(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                          const float* w, float* out, float* work, float* rec, hipStream_t st);

// Record tables: accumulator tile t = (A group, B group) of the record; A group -1 = the constant row (column sums).
struct TabF {  // f_theta: 20 groups (layout in k_vjp_tile_a), 16 tiles
  static constexpr int NG = 20, NT = 16;
  __host__ __device__ static constexpr int a(int t) {
    constexpr int tab[NT] = {6, 6, 6, 7, 8, 9, 10, 11, 12, 13, -1, -1, -1, -1, -1, -1};
    return tab[t];
  }
  __host__ __device__ static constexpr int b(int t) {
    constexpr int tab[NT] = {0, 1, 2, 0, 0, 3, 4, 5, 0, 0, 14, 15, 16, 17, 18, 19};
    return tab[t];
  }
};
struct TabX {  // f_theta, mixed family: 30 groups (extra groups in fgnn_vjp.hip PgRec), 24 tiles
  static constexpr int NG = 30, NT = 24;
  __host__ __device__ static constexpr int a(int t) {
    constexpr int tab[NT] = {6, 6, 6, 7, 8, 9, 10, 11, 12, 13, -1, -1, -1, -1, -1, -1, 23, 23, 24, 25, 26, 27, -1, -1};
    return tab[t];
  }
  __host__ __device__ static constexpr int b(int t) {
    constexpr int tab[NT] = {0, 1, 2, 0, 0, 3, 4, 5, 0, 0, 14, 15, 16, 17, 18, 19, 0, 20, 0, 21, 22, 0, 28, 29};
    return tab[t];
  }
};
struct TabM {  // two-layer MLP: groups {x|1, hid|1, d hid, d y}, tiles (d hid) x (x|1), (d y) x (hid|1)
  static constexpr int NG = 4, NT = 2;
  __host__ __device__ static constexpr int a(int t) { return t == 0 ? 2 : 3; }
  __host__ __device__ static constexpr int b(int t) { return t == 0 ? 0 : 1; }
};

template <class Tab>
__global__ __launch_bounds__(256) void k_pgrad_outer(int64_t N, int nodes_per_wave, const float* __restrict__ rec,
                                                     float* __restrict__ part) {
  constexpr int NG = Tab::NG, NT = Tab::NT, REC = 16 * NG;
  constexpr bool BLOCKRED = NT <= 16;   // 4 waves x NT KB of LDS; beyond 64 KB every wave writes its own partial
  __shared__ float sh[BLOCKRED ? 4 * NT * 256 : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nb = gw * nodes_per_wave, ne = min(N, nb + nodes_per_wave);
  f4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
  const float one = (lane & 15) == 0 ? 1.f : 0.f;
  for (int64_t n0 = nb; n0 < ne; n0 += 4) {
    const int64_t node = n0 + (lane >> 4);
    const bool ok = node < ne;
    const float* r = rec + node * REC + (lane & 15);
    float v[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = ok ? r[16 * g] : 0.f;
    const float on = ok ? one : 0.f;
    // A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]: both read the same (node, column) pattern
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tab::a(t) < 0 ? on : v[Tab::a(t) < 0 ? 0 : Tab::a(t)], v[Tab::b(t)],
                                                    acc[t], 0, 0, 0);
  }
  // D[i = 4 (lane >> 4) + r][j = lane & 15] -> [tile][i * 16 + j]
  if (!BLOCKRED) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(gw * NT + t) * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[t][r];
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[BLOCKRED ? (wv * NT + t) * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15) : 0] = acc[t][r];
  __syncthreads();
  for (int i = threadIdx.x; i < NT * 256; i += 256)
    part[(int64_t)blockIdx.x * NT * 256 + i] = (sh[i] + sh[NT * 256 + i]) + (sh[2 * NT * 256 + i] + sh[3 * NT * 256 + i]);
}

// flat offset of accumulator entry (tile t, row i, col j) in the base weight layout, or -1
template <int P>
__device__ __forceinline__ int pg_offset(int t, int i, int j) {
  using L = WLayout<P>;
  constexpr int lo = L::layer(0);
  constexpr int U1 = lo + L::L_UPD + L::UPD_W1;
  switch (t) {
    case 0:  // (dq | ds) x (x | 1)
      if (i < D) return j < D ? U1 + i * L::CAT + j : (j == D ? lo + L::L_UPD + L::UPD_B1 + i : -1);
      if (i == D) return j < D ? L::AL_W + j : (j == D ? L::AL_B : -1);
      return -1;
    case 1:  // (dq | ds) x (mp_to | prb)
      if (i < D) return j < D ? U1 + i * L::CAT + D + j : (j < D + P ? U1 + i * L::CAT + 3 * D + (j - D) : -1);
      if (i == D) return j < D ? L::AL_W + D + j : (j < D + P ? L::AL_W + 3 * D + (j - D) : -1);
      return -1;
    case 2:  // (dq | ds) x mp_from
      if (j >= D) return -1;
      if (i < D) return U1 + i * L::CAT + 2 * D + j;
      return i == D ? L::AL_W + 2 * D + j : -1;
    case 3:
    case 4: {  // own-edge cotangent sums x (x | 1): W1 x_i block and b1
      const int ph = lo + (t == 3 ? L::L_TO : L::L_FROM);
      if (i >= D) return -1;
      return j < D ? ph + L::PHI_W1 + i * L::EIN + j : (j == D ? ph + L::PHI_B1 + i : -1);
    }
    case 5:
    case 6: {  // d mp x (S | deg): W2 and b2
      const int ph = lo + (t == 5 ? L::L_TO : L::L_FROM);
      if (i >= D) return -1;
      return j < D ? ph + L::PHI_W2 + i * D + j : (j == D ? ph + L::PHI_B2 + i : -1);
    }
    case 7:  // d upd0 x (hid | 1): U2 and c2
      if (i >= D) return -1;
      return j < D ? lo + L::L_UPD + L::UPD_W2 + i * D + j : (j == D ? lo + L::L_UPD + L::UPD_B2 + i : -1);
    case 8:
    case 9: {  // neighbour-side cotangent sums x x: W1 x_j block
      const int ph = lo + (t == 8 ? L::L_TO : L::L_FROM);
      return (i < D && j < D) ? ph + L::PHI_W1 + i * L::EIN + D + j : -1;
    }
    case 10: return (i == 0 && j < D) ? L::LN_G + j : -1;
    case 11: return (i == 0 && j < D) ? L::LN_B + j : -1;
    case 12:
    case 13:
    case 14:
    case 15: {  // column sums of dS (.) attr moments: index o * 3 + c, to-block then from-block
      if (i != 0) return -1;
      const int idx = (t - 12) * 16 + j;
      if (idx >= 60) return -1;
      const int ph = lo + (idx < 30 ? L::L_TO : L::L_FROM);
      const int k = idx < 30 ? idx : idx - 30;
      return ph + L::PHI_W1 + (k / 3) * L::EIN + 2 * D + (k % 3);
    }
    default: break;
  }
  // ---- mixed family: Phi_neumann / update_neumann (single layer)
  constexpr int pn = L::phi_neu(1), un = L::upd_neu(1);
  switch (t) {
    case 16:  // dq_n x (x | 1)
      if (i >= D) return -1;
      return j < D ? un + L::NEU_W1 + i * L::NEU_CAT + j : (j == D ? un + L::NEU_B1 + i : -1);
    case 17:  // dq_n x (mp_n | prb | normal)
      return (i < D && j < D + P + 2) ? un + L::NEU_W1 + i * L::NEU_CAT + D + j : -1;
    case 18:  // gn x (x | 1)
      if (i >= D) return -1;
      return j < D ? pn + L::PHI_W1 + i * L::EIN + j : (j == D ? pn + L::PHI_B1 + i : -1);
    case 19:  // d mp_n x (S_n | deg_out)
      if (i >= D) return -1;
      return j < D ? pn + L::PHI_W2 + i * D + j : (j == D ? pn + L::PHI_B2 + i : -1);
    case 20:  // dy_n x (hid_n | 1)
      if (i >= D) return -1;
      return j < D ? un + L::NEU_W2 + i * D + j : (j == D ? un + L::NEU_B2 + i : -1);
    case 21: return (i < D && j < D) ? pn + L::PHI_W1 + i * L::EIN + D + j : -1;
    case 22:
    case 23: {
      if (i != 0) return -1;
      const int idx = (t - 22) * 16 + j;
      return idx < 30 ? pn + L::PHI_W1 + (idx / 3) * L::EIN + 2 * D + (idx % 3) : -1;
    }
    default: return -1;
  }
}

struct MapF {  // f_theta, dirichlet
  __device__ int operator()(int t, int i, int j) const { return pg_offset<2>(t, i, j); }
};
struct MapX {  // f_theta, mixed
  __device__ int operator()(int t, int i, int j) const { return pg_offset<3>(t, i, j); }
};
struct MapM {  // flat [W1 (hid, din) | b1 | W2 (dout, hid) | b2]
  int din, hid, dout;
  __device__ int operator()(int t, int i, int j) const {
    if (t == 0) return i >= hid ? -1 : (j < din ? i * din + j : (j == din ? hid * din + i : -1));
    const int o2 = hid * din + hid;
    return i >= dout ? -1 : (j < hid ? o2 + i * hid + j : (j == hid ? o2 + dout * hid + i : -1));
  }
};

// sum of the per-block partial tiles scattered to the flat gradient: 8 lanes per entry, each adds every 8th partial in
// fp64, then three xor-shuffle steps -- a fixed order, so the result is reproducible.  Grid: nt * 8 blocks of 256.
template <class Map>
__global__ __launch_bounds__(256) void k_pgrad_reduce(int nblk, int nt, const float* __restrict__ part,
                                                      float* __restrict__ grad, Map map) {
  const int e = blockIdx.x * 32 + (threadIdx.x >> 3);  // < nt * 256
  const int sub = threadIdx.x & 7;
  const int t = e >> 8, i = (e >> 4) & 15, j = e & 15;
  const int off = map(t, i, j);
  double s = 0.0;
  if (off >= 0)
    for (int b = sub; b < nblk; b += 8) s += (double)part[(int64_t)b * nt * 256 + e];
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  if (sub == 0 && off >= 0) grad[off] = (float)s;
}

static inline int pgrad_blocks(int64_t N, int* nodes_per_wave) {
  // a wave owns >= 64 nodes (multiple of 4); at most 1024 blocks of 4 waves
  int64_t npw = std::max<int64_t>(64, cdiv(cdiv(N, (int64_t)4096), (int64_t)4) * 4);
  *nodes_per_wave = (int)npw;
  return (int)cdiv(N, npw * 4);
}

int psignn_f_gather_vjp_rec(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                            const float* w, float* out, float* work, float* rec, hipStream_t st);

extern "C" int64_t psignn_f_param_vjp_workspace_floats(const psignn_plan_t* p) {
  if (!p) return 0;
  int npw;
  const int nblk = pgrad_blocks(p->N, &npw);
  // VJP scratch (<= N * 90) + plan-order copies (<= N * 36) + records (N * 480 mixed) + partial tiles (per wave for mixed)
  return p->N * (9 * D + 36 + TabX::NG * 16) + (int64_t)nblk * 4 * TabX::NT * 256;
}

extern "C" int64_t psignn_param_grad_size(int mixed, int nl) {
  return mixed ? WLayout<3>::base_total(nl, true) : WLayout<2>::base_total(nl, false);
}

// Tiled plans of both families, everything in PLAN order: the tiled VJP kernels in record mode, then the MFMA reduction of the
// records (dirichlet: 20 groups, 16 tiles; mixed: 30 groups, 24 tiles -- the Neumann factors of fgnn_vjp.hip's PgRec).
static int param_vjp_tiled(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                           const float* w, float* d_grad, float* d_out_h, float* work, hipStream_t st) {
  const int64_t N = p->N;
  float* B = work;
  float* rec = B + N * 4 * D;
  int npw;
  const int nblk = pgrad_blocks(N, &npw);
  int rc = psignn_f_tile_vjp_rec(p, W, nl, h, prb, nrm, w, d_out_h, B, rec, st);
  if (rc) return rc;
  if (p->mixed) {
    float* part = rec + N * TabX::NG * 16;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<3>::base_total(nl, true) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabX><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabX::NT * 8, 256, 0, st>>>(nblk * 4, TabX::NT, part, d_grad, MapX())));
  } else {
    float* part = rec + N * PGREC;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<2>::base_total(nl, false) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabF><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabF::NT * 8, 256, 0, st>>>(nblk, TabF::NT, part, d_grad, MapF())));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// h, prb, w in PLAN order.  d_grad: psignn_param_grad_size floats (fold slots left zero); d_out_h: (N, 10) = w^T df/dh.
extern "C" int psignn_f_param_vjp_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                                    const float* w, float* d_grad, float* d_out_h, float* work, void* stream) {
  ARG_CHECK(p && W && h && prb && w && d_grad && d_out_h && work, "NULL argument");
  ARG_CHECK(p->tiled && !p->mixed && nl == 1,
            "plan-order parameter gradients: tiled single-layer dirichlet plans (mixed plans: psignn_f_param_vjp, which takes the normals)");
  return param_vjp_tiled(p, W, nl, h, prb, nullptr, w, d_grad, d_out_h, work, (hipStream_t)stream);
}

// ---- backward of the VJP (the Jacobian regulariser's gradient; kernels and derivation in gather_backward.hip)
int psignn_jacreg_records(const psignn_plan* p, const float* W, const float* h, const float* prb, const float* nrm,
                          const float* v, const float* gbar, float* out_h, float* work, float* rec, hipStream_t st);

extern "C" int64_t psignn_f_vjp_backward_workspace_floats(const psignn_plan_t* p) {
  if (!p) return 0;
  int npw;
  const int nblk = pgrad_blocks(2 * p->N, &npw);
  // scratch (<= N * 170) + two records per node + partial tiles (per wave for the mixed family)
  return p->N * (17 * D + 2 * TabX::NG * 16) + (int64_t)nblk * 4 * TabX::NT * 256;
}

// Gradient of  phi = gbar . (J_f(h)^T v) = v^T J_f(h) gbar  (gbar constant) w.r.t. the parameters (d_grad, layout of
// psignn_f_param_vjp) and w.r.t. h (d_grad_h): what autograd's double backward leaves after
// autograd.grad(f(h), h, v, create_graph=True) -- jac_loss_estimate, dirichlet/psignn/model.py:416-435.  Caller's numbering.
extern "C" int psignn_f_vjp_backward(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                                     const float* nrm, const float* v, const float* gbar, float* d_grad, float* d_grad_h,
                                     float* work, void* stream) {
  ARG_CHECK(p && W && h && prb && v && gbar && d_grad && d_grad_h && work, "NULL argument");
  ARG_CHECK(nl == 1, "the backward of the VJP is implemented for single-layer blocks");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  float* rec = work + N * 17 * D;
  int npw;
  const int nblk = pgrad_blocks(2 * N, &npw);
  int rc = psignn_jacreg_records(p, W, h, prb, nrm, v, gbar, d_grad_h, work, rec, st);
  if (rc) return rc;
  if (p->mixed) {
    float* part = rec + 2 * N * TabX::NG * 16;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<3>::base_total(nl, true) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabX><<<nblk, 256, 0, st>>>(2 * N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabX::NT * 8, 256, 0, st>>>(nblk * 4, TabX::NT, part, d_grad, MapX())));
  } else {
    float* part = rec + 2 * N * TabF::NG * 16;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<2>::base_total(nl, false) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabF><<<nblk, 256, 0, st>>>(2 * N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabF::NT * 8, 256, 0, st>>>(nblk, TabF::NT, part, d_grad, MapF())));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// ---- DS-GPS: backward of one recurrent update (kernels in gather_backward.hip)
struct TabG {  // groups: see gather_backward.hip
  static constexpr int NG = 20, NT = 19;
  __host__ __device__ static constexpr int a(int t) {
    constexpr int tab[NT] = {6, 6, 6, 7, 8, 9, 10, 11, 12, 13, 11, 11, -1, -1, -1, -1, 14, 14, 14};
    return tab[t];
  }
  __host__ __device__ static constexpr int b(int t) {
    constexpr int tab[NT] = {0, 1, 2, 0, 0, 3, 4, 0, 0, 0, 1, 2, 16, 17, 18, 19, 5, 1, 2};
    return tab[t];
  }
};
struct TabGX {  // mixed family: TabG + the Neumann tiles of TabX (groups 20..29)
  static constexpr int NG = 30, NT = 27;
  __host__ __device__ static constexpr int a(int t) {
    constexpr int tab[NT] = {6, 6, 6, 7, 8, 9, 10, 11, 12, 13, 11, 11, -1, -1, -1, -1, 14, 14, 14, 23, 23, 24, 25, 26, 27, -1, -1};
    return tab[t];
  }
  __host__ __device__ static constexpr int b(int t) {
    constexpr int tab[NT] = {0, 1, 2, 0, 0, 3, 4, 0, 0, 0, 1, 2, 16, 17, 18, 19, 5, 1, 2, 0, 20, 0, 21, 22, 0, 28, 29};
    return tab[t];
  }
};
// gradient layout: the f_theta base layout (Phi slots of layer 0; mixed: phi_neumann | update_neumann too) followed by the
// gates [Wz (10 x (30+P)) | bz | Wr | br | Wc | bc]
template <int P>
struct MapG {
  __device__ int operator()(int t, int i, int j) const {
    constexpr int CAT = 3 * D + P, GSZ = D * CAT + D;
    constexpr int G0 = WLayout<P>::base_total(1, P == 3);
    int gate = -1, blk = 0;
    switch (t) {
      case 0: gate = 0; blk = 0; break;
      case 1: gate = 0; blk = 1; break;
      case 2: gate = 0; blk = 2; break;
      case 7: gate = 1; blk = 0; break;
      case 10: gate = 1; blk = 1; break;
      case 11: gate = 1; blk = 2; break;
      case 16: gate = 2; blk = 0; break;
      case 17: gate = 2; blk = 1; break;
      case 18: gate = 2; blk = 2; break;
      default: return pg_offset<P>(t < 19 ? t : t - 3, i, j);   // Phi tiles as in TabF; 19..26 = TabX's Neumann tiles 16..23
    }
    if (i >= D) return -1;
    const int base = G0 + gate * GSZ;
    if (blk == 0) return j < D ? base + i * CAT + j : (j == D ? base + D * CAT + i : -1);
    if (blk == 1) return j < D ? base + i * CAT + D + j : (j < D + P ? base + i * CAT + 3 * D + (j - D) : -1);
    return j < D ? base + i * CAT + 2 * D + j : -1;
  }
};

int psignn_dsgps_step_records(const psignn_plan* p, const float* Wf, const float* Wg, const float* h, const float* prb,
                              const float* nrm, const float* w, float* out_h, float* work, float* rec, hipStream_t st);

extern "C" int64_t psignn_dsgps_grad_size(int mixed) {
  return mixed ? WLayout<3>::base_total(1, true) + 3 * (D * (3 * D + 3) + D) : WLayout<2>::base_total(1, false) + 3 * (D * (3 * D + 2) + D);
}
extern "C" int64_t psignn_dsgps_step_backward_workspace_floats(const psignn_plan_t* p) {
  if (!p) return 0;
  int npw;
  const int nblk = pgrad_blocks(p->N, &npw);
  return p->N * (17 * D + TabGX::NG * 16) + (int64_t)nblk * 4 * TabGX::NT * 256;
}

// w^T (d h' / d theta) -> d_grad (psignn_dsgps_grad_size floats) and w^T (d h' / d h) -> d_out_h for one DS-GPS update
// h' = step(h) (caller's numbering).  d_phi_weights: the Phi modules (mixed: and update_neumann) in the f_theta weight
// layout (psignn_weights_size(mixed, 1) floats); d_gate_weights: [Wz|bz|Wr|br|Wc|bc].
extern "C" int psignn_dsgps_step_backward(const psignn_plan_t* p, const float* d_phi_weights, const float* d_gate_weights,
                                          const float* h, const float* prb, const float* nrm, const float* w, float* d_grad,
                                          float* d_out_h, float* work, void* stream) {
  ARG_CHECK(p && d_phi_weights && d_gate_weights && h && prb && w && d_grad && d_out_h && work, "NULL argument");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  float* rec = work + N * 17 * D;
  int npw;
  const int nblk = pgrad_blocks(N, &npw);
  int rc = psignn_dsgps_step_records(p, d_phi_weights, d_gate_weights, h, prb, nrm, w, d_out_h, work, rec, st);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)psignn_dsgps_grad_size(p->mixed) * 4, st));
  if (p->mixed) {
    float* part = rec + N * TabGX::NG * 16;
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabGX><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabGX::NT * 8, 256, 0, st>>>(nblk * 4, TabGX::NT, part, d_grad, MapG<3>())));
  } else {
    float* part = rec + N * TabG::NG * 16;
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabG><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabG::NT * 8, 256, 0, st>>>(nblk * 4, TabG::NT, part, d_grad, MapG<2>())));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// ---- DSS: backward of one update (kernels in gather_backward.hip); gradient in the f_theta layout with three node inputs
int psignn_dss_step_records(const psignn_plan* p, const float* Wf, float alpha, const float* h, const float* bp,
                            const float* w, float* out_h, float* work, float* rec, hipStream_t st);

extern "C" int64_t psignn_dss_grad_size(void) { return WLayout<3>::base_total(1, false); }
extern "C" int64_t psignn_dss_step_backward_workspace_floats(const psignn_plan_t* p) {
  if (!p) return 0;
  int npw;
  const int nblk = pgrad_blocks(p->N, &npw);
  return p->N * (13 * D + PGREC) + (int64_t)nblk * TabF::NT * 256;
}

// w^T (d h' / d theta_t) -> d_grad (psignn_dss_grad_size floats: shared | phi_to{W1 (10x23: columns 20, 21 unused, 22 = the
// edge feature), b1, W2, b2} | phi_from | psi{W1 (10x33), b1, W2, b2}) and w^T (d h' / d h) -> d_out_h for one DSS update.
// d_weights_t: update t's modules in the f_theta weight layout (three node inputs, no Neumann blocks).
extern "C" int psignn_dss_step_backward(const psignn_plan_t* p, const float* d_weights_t, float alpha, const float* h,
                                        const float* bprime, const float* w, float* d_grad, float* d_out_h, float* work,
                                        void* stream) {
  ARG_CHECK(p && d_weights_t && h && bprime && w && d_grad && d_out_h && work, "NULL argument");
  ARG_CHECK(!p->mixed, "DSS plans carry no boundary-condition tags");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  float* rec = work + N * 13 * D;
  float* part = rec + N * PGREC;
  int npw;
  const int nblk = pgrad_blocks(N, &npw);
  int rc = psignn_dss_step_records(p, d_weights_t, alpha, h, bprime, w, d_out_h, work, rec, st);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)psignn_dss_grad_size() * 4, st));
  LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabF><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
  LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabF::NT * 8, 256, 0, st>>>(nblk, TabF::NT, part, d_grad, MapX())));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of the two-layer MLP (Encoder / Decoder, model.py:370-392; y = W2 relu(W1 x + b1) + b2) and the
// transposed residual SpMV -- what autograd runs for the autoencoder / residual terms of the training loss
// (dirichlet/psignn/model.py:58-99, 157-167).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mlp2_bwd(int64_t n, int din, int hid, int dout, const float* __restrict__ x,
                                                  const float* __restrict__ gy, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  float* __restrict__ gx, float* __restrict__ rec) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float g[4][16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) g[q][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < din) g[0][i] = x[r * din + i];
    if (i < dout) g[3][i] = gy[r * dout + i];
  }
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    if (o < hid) {
      float s = b1[o];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < din) s = fmaf(w1[o * din + i], g[0][i], s);
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < dout) d = fmaf(w2[k * hid + o], g[3][k], d);
      g[1][o] = fmaxf(s, 0.f);
      g[2][o] = s > 0.f ? d : 0.f;
    }
  }
  if (gx) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < din) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (o < hid) s = fmaf(w1[o * din + i], g[2][o], s);
        gx[r * din + i] = s;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // the constant column of the two right-factor groups
    if (i == din) g[0][i] = 1.f;
    if (i == hid) g[1][i] = 1.f;
  }
  float4* q = reinterpret_cast<float4*>(rec + r * 64);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) q[a * 4 + i] = make_float4(g[a][4 * i], g[a][4 * i + 1], g[a][4 * i + 2], g[a][4 * i + 3]);
}

extern "C" int64_t psignn_mlp2_backward_workspace_floats(int64_t n) {
  int npw;
  const int nblk = pgrad_blocks(n, &npw);
  return n * 64 + (int64_t)nblk * TabM::NT * 256;
}

// d_gflat: [W1 (hid, din) | b1 (hid) | W2 (dout, hid) | b2 (dout)] gradients; d_gx (n, din) may be NULL.
extern "C" int psignn_mlp2_backward(const float* x, const float* gy, int64_t n, int din, int hid, int dout, const float* w1,
                                    const float* b1, const float* w2, float* d_gx, float* d_gflat, float* work,
                                    void* stream) {
  ARG_CHECK(x && gy && w1 && b1 && w2 && d_gflat && work, "NULL argument");
  ARG_CHECK(n >= 0 && din >= 1 && din <= 15 && hid >= 1 && hid <= 15 && dout >= 1 && dout <= 15, "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(d_gflat, 0, (size_t)(hid * din + hid + dout * hid + dout) * 4, st));
  if (n == 0) return PSIGNN_OK;
  float* rec = work;
  float* part = rec + n * 64;
  int npw;
  const int nblk = pgrad_blocks(n, &npw);
  LAUNCH("k_mlp2_bwd", st, (k_mlp2_bwd<<<(unsigned)cdiv(n, (int64_t)256), 256, 0, st>>>(n, din, hid, dout, x, gy, w1, b1, w2, d_gx, rec)));
  LAUNCH("k_pgrad_outer_mlp", st, (k_pgrad_outer<TabM><<<nblk, 256, 0, st>>>(n, npw, rec, part)));
  LAUNCH("k_pgrad_reduce_mlp", st, (k_pgrad_reduce<<<TabM::NT * 8, 256, 0, st>>>(nblk, TabM::NT, part, d_gflat, MapM{din, hid, dout})));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// out = A^T r  (A = COO(edge_index, a_ij) incl. the diagonal): in-edge lists of the plan + the caller's a_ij
__global__ __launch_bounds__(256) void k_residual_t(int64_t N, const int32_t* __restrict__ csc_ptr,
                                                    const int32_t* __restrict__ csc_nbr, const int32_t* __restrict__ csc_eid,
                                                    const float* __restrict__ a_ij, const int32_t* __restrict__ a_ptr,
                                                    const int32_t* __restrict__ a_col, const float* __restrict__ a_val,
                                                    const float* __restrict__ r, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int32_t i = csc_ptr[c]; i < csc_ptr[c + 1]; ++i) s = fmaf(a_ij[csc_eid[i]], r[csc_nbr[i]], s);
  float dg = 0.f;
  for (int32_t i = a_ptr[c]; i < a_ptr[c + 1]; ++i)
    if (a_col[i] == (int32_t)c) dg += a_val[i];
  out[c] = fmaf(dg, r[c], s);
}

extern "C" int psignn_residual_t(const psignn_plan_t* p, const float* d_a_ij, const float* d_r, float* d_out, void* stream) {
  ARG_CHECK(p && d_a_ij && d_r && d_out, "NULL argument");
  k_residual_t<<<(unsigned)cdiv(p->N, (int64_t)256), 256, 0, (hipStream_t)stream>>>(
      p->N, p->csc_ptr, p->csc_nbr, p->csc_eid, d_a_ij, p->a_ptr, p->a_col, p->a_val, d_r, d_out);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Caller-order parameter-VJP for every plan: tiled plans of both families run the tiled kernels (permutation passes around
// them; mixed family since round 3); untiled plans run the global-gather kernels in PG mode.  d_normals: mixed only.
extern "C" int psignn_f_param_vjp(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                                  const float* nrm, const float* w, float* d_grad, float* d_out_h, float* work,
                                  void* stream) {
  ARG_CHECK(p && W && h && prb && w && d_grad && d_out_h && work, "NULL argument");
  ARG_CHECK(nl == 1, "parameter gradients are implemented for single-layer blocks");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  int rc;
  KNOB_INT(mixed_tiled, [] { const char* e = getenv("PSIGNN_MIXED_PGRAD"); return (int)!(e && strcmp(e, "gather") == 0); }());
  if (p->tiled && (!p->mixed || mixed_tiled)) {
    const int P = p->mixed ? 3 : 2;
    float* hp = work;
    float* wp = hp + N * D;
    float* op = wp + N * D;
    float* pp = op + N * D;  // (N, 2 | 3)
    float* np_ = pp + N * P; // (N, 2) unit normals, mixed plans
    float* rest = np_ + (p->mixed ? N * 2 : 0);
    rest += (4 - ((rest - work) & 3)) & 3;   // the VJP's B rows are read as float4
    if ((rc = psignn_plan_permute(p, h, D, hp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, w, D, wp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, prb, P, pp, 1, stream))) return rc;
    if (p->mixed && (rc = psignn_plan_permute(p, nrm, 2, np_, 1, stream))) return rc;
    if ((rc = param_vjp_tiled(p, W, nl, hp, pp, p->mixed ? np_ : nullptr, wp, d_grad, op, rest, st))) return rc;
    return psignn_plan_permute(p, op, D, d_out_h, 0, stream);
  }
  float* scratch = work;                      // Pj + B: N * 90 floats at most
  float* rec = scratch + N * 9 * D;
  int npw;
  const int nblk = pgrad_blocks(N, &npw);
  if ((rc = psignn_f_gather_vjp_rec(p, W, nl, h, prb, nrm, w, d_out_h, scratch, rec, st))) return rc;
  if (p->mixed) {
    float* part = rec + N * TabX::NG * 16;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<3>::base_total(nl, true) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabX><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabX::NT * 8, 256, 0, st>>>(nblk * 4, TabX::NT, part, d_grad, MapX())));
  } else {
    float* part = rec + N * TabF::NG * 16;
    HIP_TRY(hipMemsetAsync(d_grad, 0, (size_t)WLayout<2>::base_total(nl, false) * 4, st));
    LAUNCH("k_pgrad_outer", st, (k_pgrad_outer<TabF><<<nblk, 256, 0, st>>>(N, npw, rec, part)));
    LAUNCH("k_pgrad_reduce", st, (k_pgrad_reduce<<<TabF::NT * 8, 256, 0, st>>>(nblk, TabF::NT, part, d_grad, MapF())));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
