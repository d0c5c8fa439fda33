// DSS, the Deep Statistical Solver baseline of the reference (dirichlet/dss/model.py:97-120), on the tiled mesh plan
// (gfx950).  SURVEY §8f-4: again the gather -> edge MLP -> segment sum of fgnn_tile.hip (tile_helpers.h), here with
//   * a scalar edge feature (normalised a_ij): the plan carries it as edge_attr = (0, 0, a_ij_norm), so the 21-wide first
//     Phi layer [x_i | x_j | a] maps onto the 23-wide blocks with two zero attr rows;
//   * separate weights for each of the k updates (phi_to_list[t], phi_from_list[t], psi_list[t]);
//   * the node update  h <- h + alpha * Psi_t([h | mess_to | mess_from | b'_norm]),  Psi = Linear(33,10)-ReLU-Linear(10,10);
//   * no boundary rows: the Dirichlet condition lives in b' (model.py:107-117), H_0 = 0.
// One launch per step, two state buffers in plan order.
//
// Weight buffer per step (floats, transposed [in k][out o]; engine.pack_dss):
//   0 W1j_to^T 100 | 100 W1j_from^T | 200 W1i_to^T | 300 W1i_from^T | 400 A_to^T 30 (rows 0,1 zero) | 430 A_from^T 30 |
//   460 b1_to | 470 b1_from | 480 W2_to^T 100 | 580 b2_to | 590 W2_from^T 100 | 690 b2_from |
//   700 P1^T (33 x 10) 330 | 1030 c1 | 1040 P2^T 100 | 1140 c2                                   -- 1150 per step
#include "tile_helpers.h"

namespace dss {
constexpr int W1J_TO = 0, W1J_FR = 100, W1I_TO = 200, W1I_FR = 300, A_TO = 400, A_FR = 430, B1_TO = 460, B1_FR = 470;
constexpr int W2_TO = 480, B2_TO = 580, W2_FR = 590, B2_FR = 690, P1 = 700, C1 = 1030, P2 = 1040, C2 = 1140, STEP = 1150;
constexpr int P = 3;
}  // namespace dss

__global__ __launch_bounds__(TILE_THREADS) void k_dss_tile(int n_tiles, int chunk, const int32_t* __restrict__ tile_ptr,
                                                           const int32_t* __restrict__ tile_slice,
                                                           const int32_t* __restrict__ halo, const int32_t* __restrict__ halo_cnt,
                                                           const int32_t* __restrict__ slice_off,
                                                           const uint8_t* __restrict__ slice_deg, const uint4* __restrict__ ell,
                                                           const float* __restrict__ W, float alpha,
                                                           const float* __restrict__ h, const float* __restrict__ bp,
                                                           float* __restrict__ out) {
  constexpr int RS = 20;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  float x[D];
  for (int row = tid; row < n_t + n_h; row += TILE_THREADS) {
    const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
    float xr[D];
    load10(h + node * D, xr);
    if (row == tid) {
#pragma unroll
      for (int o = 0; o < D; ++o) x[o] = xr[o];
    }
    v2f ta[5], tb[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ta[p] = tb[p] = splat(0.f);
    PHASE();
    mv2<D>(W + dss::W1J_TO, xr, ta);
    PHASE();
    mv2<D>(W + dss::W1J_FR, xr, tb);
    float4* q = reinterpret_cast<float4*>(lds + row * RS);
    q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
    q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
    q[2] = make_float4(ta[4].x, ta[4].y, tb[0].x, tb[0].y);
    q[3] = make_float4(tb[1].x, tb[1].y, tb[2].x, tb[2].y);
    q[4] = make_float4(tb[3].x, tb[3].y, tb[4].x, tb[4].y);
  }
  __syncthreads();
  if (tid >= n_t) return;
  const int64_t n = (int64_t)t0 + tid;
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (tid >> 6);
  const uint4* slots = ell + (int64_t)slice_off[slice] * 64 + lane;
  const int nslots = slice_deg[slice];
  v2f Pi[5], Pi2[5], S_to[5], S_fr[5];
  float deg_in, deg_out;
  ld5(W + dss::B1_TO, Pi);
#pragma unroll
  for (int p = 0; p < 5; ++p) S_to[p] = S_fr[p] = splat(0.f);
  PHASE();
  mv2<D>(W + dss::W1I_TO, x, Pi);
  ld5(W + dss::B1_FR, Pi2);
  PHASE();
  mv2<D>(W + dss::W1I_FR, x, Pi2);
  PHASE();
  edge_pass_both_clamp<RS>(slots, nslots, lds, W + dss::A_TO, W + dss::A_FR, Pi, Pi2, S_to, S_fr, deg_in, deg_out);
  // the node state is dead during the slot walk (five waves per SIMD instead of four): read it back, as k_f_tile does
  PHASE();
  load10(h + n * D, x);
  PHASE();
  v2f mt[5], mf[5], b[5];
  ld5(W + dss::B2_TO, b);
#pragma unroll
  for (int p = 0; p < 5; ++p) mt[p] = splat(deg_in) * b[p];
  PHASE();
  mv2<D>(W + dss::W2_TO, reinterpret_cast<const float*>(S_to), mt);
  ld5(W + dss::B2_FR, b);
#pragma unroll
  for (int p = 0; p < 5; ++p) mf[p] = splat(deg_out) * b[p];
  PHASE();
  mv2<D>(W + dss::W2_FR, reinterpret_cast<const float*>(S_fr), mf);
  float pq[dss::P];
#pragma unroll
  for (int k = 0; k < dss::P; ++k) pq[k] = bp[n * dss::P + k];
  v2f q[5], c[5];
  ld5(W + dss::C1, q);
  PHASE();
  mv2<D>(W + dss::P1, x, q);
  PHASE();
  mv2<D>(W + dss::P1 + 10 * D, reinterpret_cast<const float*>(mt), q);
  PHASE();
  mv2<D>(W + dss::P1 + 20 * D, reinterpret_cast<const float*>(mf), q);
  mv2<dss::P>(W + dss::P1 + 30 * D, pq, q);
#pragma unroll
  for (int p = 0; p < 5; ++p) q[p] = __builtin_elementwise_max(q[p], splat(0.f));
  ld5(W + dss::C2, c);
  PHASE();
  mv2<D>(W + dss::P2, reinterpret_cast<const float*>(q), c);
  float y[D];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    y[2 * p] = fmaf(alpha, c[p].x, x[2 * p]);
    y[2 * p + 1] = fmaf(alpha, c[p].y, x[2 * p + 1]);
  }
  store10(out + n * D, y);
}

extern "C" int64_t psignn_dss_weights_size(int k) { return (int64_t)k * dss::STEP; }

// k updates from H_0 = 0 with per-step weights.  d_bprime (N, 3) and d_out (N, 10) in the caller's numbering;
// d_work: N * 23 floats.  The plan must have been created with edge_attr = (0, 0, a_ij_norm).
extern "C" int psignn_dss_forward(const psignn_plan_t* p, const float* W, int k, float alpha, const float* d_bprime,
                                  float* d_out, float* d_work, void* stream) {
  ARG_CHECK(p && W && d_bprime && d_out && d_work, "NULL argument");
  ARG_CHECK(k >= 1, "step count must be positive");
  ARG_CHECK(p->tiled, "DSS kernels need a tiled plan (mesh positions)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  float* a = d_work;
  float* b = a + N * D;
  float* bpp = b + N * D;  // (N, 3)
  int rc;
  if ((rc = psignn_plan_permute(p, d_bprime, dss::P, bpp, 1, stream))) return rc;
  HIP_TRY(hipMemsetAsync(a, 0, (size_t)N * D * 4, st));
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = (unsigned)(chunk * 8);
  const size_t lds = (size_t)p->max_rows * 20 * 4;
  float* cur = a;
  for (int i = 0; i < k; ++i) {
    float* dst = (cur == a) ? b : a;
    LAUNCH("k_dss_tile", st, (k_dss_tile<<<grid, TILE_THREADS, lds, st>>>(
        (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell,
        W + (int64_t)i * dss::STEP, alpha, cur, bpp, dst)));
    cur = dst;
  }
  HIP_TRY(hipGetLastError());
  return psignn_plan_permute(p, cur, D, d_out, 0, stream);
}

// One update (step t's weights) with the state and b'_norm in PLAN order: for callers that keep every iterate
// (DeepStatisticalSolver.forward decodes and scores each of them, model.py:59-95).
extern "C" int psignn_dss_step_p(const psignn_plan_t* p, const float* W, int t, float alpha, const float* d_h,
                                 const float* d_bprime_p, float* d_out, void* stream) {
  ARG_CHECK(p && W && d_h && d_bprime_p && d_out, "NULL argument");
  ARG_CHECK(t >= 0, "negative step index");
  ARG_CHECK(p->tiled, "DSS kernels need a tiled plan (mesh positions)");
  ARG_CHECK(d_out != d_h, "out must not alias the state");
  hipStream_t st = (hipStream_t)stream;
  const int chunk = (int)cdiv(p->n_tiles, 8);
  LAUNCH("k_dss_tile", st, (k_dss_tile<<<(unsigned)(chunk * 8), TILE_THREADS, (size_t)p->max_rows * 20 * 4, st>>>(
      (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell,
      W + (int64_t)t * dss::STEP, alpha, d_h, d_bprime_p, d_out)));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
