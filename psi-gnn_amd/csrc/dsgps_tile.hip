// DS-GPS, the unrolled recurrent baseline of the reference (dirichlet/dsgps/model.py:130-163), on the tiled mesh plan
// (gfx950).  SURVEY §8f-4: the same gather -> edge MLP -> segment sum as PSI-GNN's f (stage 1 / stage 2 of
// fgnn_tile.hip, shared through tile_helpers.h) with a GRU-style node update and k fixed steps instead of a root-find:
//     c      = [h | Phi_to(h) | Phi_from(h) | prb]                      (32)
//     alpha  = sigmoid(Wz c + bz);  reset = sigmoid(Wr c + br)          (10 each; MLPActivation([32, 10], Sigmoid))
//     corr   = tanh(Wc [reset * h | Phi_to(h) | Phi_from(h) | prb] + bc)
//     h'     = h + alpha * corr;   Dirichlet rows <- H_0 rows           (model.py:141-152)
// One launch per step; the k steps run back to back on the stream, ping-ponging two state buffers in plan order.
// Mixed family (mixed/dsgps/model.py:50-95): prb has 3 columns (gates 33 wide) and Neumann rows are REPLACED by
// update_neumann([h | Phi_neumann(h) | prb | normal]) exactly as in the mixed PSI-GNN block (no LayerNorm here); the
// Neumann weights use the transposed WLayout<3>::N_* block layout, appended after the gates.
//
// Weight buffer (floats; matrices transposed [in k][out o] for the packed-fp32 matvecs; built by engine.pack_dsgps):
//   0    W1j_to^T 100 | 100 W1j_from^T | 200 W1i_to^T | 300 W1i_from^T | 400 A_to^T (30, rows 0,1 negated: an in-edge
//   carries the mirrored attr) | 430 A_from^T | 460 b1_to | 470 b1_from            -- same order as WLayout::T_*
//   480  W2_to^T 100 | 580 b2_to | 590 W2_from^T | 690 b2_from
//   700  Wz^T 320 | 1020 bz | 1030 Wr^T 320 | 1350 br | 1360 Wc^T 320 | 1680 bc    -- total 1690
#include "tile_helpers.h"

template <int PP>
struct DsL {
  static constexpr int W1J_TO = 0, W1J_FR = 100, W1I_TO = 200, W1I_FR = 300, A_TO = 400, A_FR = 430, B1_TO = 460, B1_FR = 470;
  static constexpr int W2_TO = 480, B2_TO = 580, W2_FR = 590, B2_FR = 690;
  static constexpr int P = PP, G = (3 * D + PP) * D;
  static constexpr int WZ = 700, BZ = WZ + G, WR = BZ + D, BR = WR + G, WC = BR + D, BC = WC + G, NEU = BC + D;
  static constexpr int TOTAL = NEU + (PP == 3 ? WLayout<3>::TPN_SZ : 0);
};

// The 30 gate activations per node cost more VALU issue slots than the three (32, 10) matvecs when written with the
// libm-accurate expf / tanhf / division (measured: 150 us per 1M-node step).  v_exp_f32 / v_rcp_f32 are accurate to
// ~1 ulp, far inside the parity tolerance of the k-step recurrence (tests/test_gpu_dsgps.py).
__device__ __forceinline__ float fast_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return fmaf(2.f, fast_sigmoid(2.f * v), -1.f); }

template <int P, bool MIXED>
__global__ __launch_bounds__(TILE_THREADS) void k_dsgps_tile(int n_tiles, int chunk, const int32_t* __restrict__ tile_ptr,
                                                             const int32_t* __restrict__ tile_slice,
                                                             const int32_t* __restrict__ halo, const int32_t* __restrict__ halo_cnt,
                                                             const int32_t* __restrict__ slice_off,
                                                             const uint8_t* __restrict__ slice_deg, const uint4* __restrict__ ell,
                                                             const uint8_t* __restrict__ flags, const float* __restrict__ W,
                                                             const float* __restrict__ h, const float* __restrict__ h0,
                                                             const float* __restrict__ prb, const float* __restrict__ nrm,
                                                             float* __restrict__ out) {
  using dsl = DsL<P>;
  using LN = WLayout<3>;              // N_* offsets of the transposed Neumann block
  constexpr int RS = MIXED ? 32 : 20;
  const float* TN = W + dsl::NEU;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);  // XCD-contiguous tile runs, as in k_f_tile
  if (tile >= n_tiles) return;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  // ---- stage 1: neighbour-side projections of tile + halo rows -> LDS
  float x[D];
  bool tile_neu = false;   // Phi_neumann projections are read by Neumann lanes only: boundary tiles
  if (MIXED) tile_neu = __syncthreads_or(tid < n_t ? (flags[t0 + tid] & FLAG_NEUMANN) : 0) != 0;
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
    mv2<D>(W + dsl::W1J_TO, xr, ta);
    PHASE();
    mv2<D>(W + dsl::W1J_FR, xr, tb);
    float4* q = reinterpret_cast<float4*>(lds + row * RS);
    q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
    q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
    q[2] = make_float4(ta[4].x, ta[4].y, tb[0].x, tb[0].y);
    q[3] = make_float4(tb[1].x, tb[1].y, tb[2].x, tb[2].y);
    q[4] = make_float4(tb[3].x, tb[3].y, tb[4].x, tb[4].y);
    if (MIXED && tile_neu) {
#pragma unroll
      for (int p = 0; p < 5; ++p) ta[p] = splat(0.f);
      PHASE();
      mv2<D>(TN + LN::N_W1J, xr, ta);
      q[5] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
      q[6] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
      reinterpret_cast<float2*>(q + 7)[0] = make_float2(ta[4].x, ta[4].y);
    }
  }
  __syncthreads();
  if (tid >= n_t) return;
  // ---- stage 2: one tile node per lane
  const int64_t n = (int64_t)t0 + tid;
  float y[D];
  const uint8_t fl = flags[n];
  if (fl & FLAG_DIRICHLET) {  // H[update+1][index_dirichlet] = H['0'][index_dirichlet]  (model.py:152)
    load10(h0 + n * D, y);
    store10(out + n * D, y);
    return;
  }
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (tid >> 6);
  const uint4* slots = ell + (int64_t)slice_off[slice] * 64 + lane;
  const int nslots = slice_deg[slice];
  v2f Pi[5], Pi2[5], S_to[5], S_fr[5];
  float deg_in, deg_out;
  ld5(W + dsl::B1_TO, Pi);
#pragma unroll
  for (int p = 0; p < 5; ++p) S_to[p] = S_fr[p] = splat(0.f);
  PHASE();
  mv2<D>(W + dsl::W1I_TO, x, Pi);
  ld5(W + dsl::B1_FR, Pi2);
  PHASE();
  mv2<D>(W + dsl::W1I_FR, x, Pi2);
  PHASE();
  edge_pass_both_clamp<RS>(slots, nslots, lds, W + dsl::A_TO, W + dsl::A_FR, Pi, Pi2, S_to, S_fr, deg_in, deg_out);
  // the node state is dead during the slot walk (five waves per SIMD instead of four): read it back, as k_f_tile does
  PHASE();
  load10(h + n * D, x);
  PHASE();
  if (MIXED && (fl & FLAG_NEUMANN)) {
    // H[update+1][index_neumann] = update_neumann([h | Phi_neumann(h) | prb | normal])   (mixed/dsgps/model.py:88-93)
    v2f S_n[5], hid[5], gN[5], y2[5];
    ld5(TN + LN::N_B1, Pi);
#pragma unroll
    for (int p = 0; p < 5; ++p) S_n[p] = splat(0.f);
    PHASE();
    mv2<D>(TN + LN::N_W1I, x, Pi);
    edge_pass<RS, 2 * D, SLOT_OUT>(slots, nslots, lds, TN + LN::N_A, Pi, S_n);
    ld5(TN + LN::N_NB1, hid);
    ld5(TN + LN::N_gN, gN);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid[p] = __builtin_elementwise_fma(splat(deg_out), gN[p], hid[p]);
    PHASE();
    mv2<D>(TN + LN::N_N1H, x, hid);
    PHASE();
    mv2<D>(TN + LN::N_GN, reinterpret_cast<const float*>(S_n), hid);
    float pn[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pn[k] = prb[n * P + k];
    pn[P] = nrm[n * 2];
    pn[P + 1] = nrm[n * 2 + 1];
    mv2<P + 2>(TN + LN::N_N1P, pn, hid);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid[p] = __builtin_elementwise_max(hid[p], splat(0.f));
    ld5(TN + LN::N_NB2, y2);
    PHASE();
    mv2<D>(TN + LN::N_N2, reinterpret_cast<const float*>(hid), y2);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      y[2 * p] = y2[p].x;
      y[2 * p + 1] = y2[p].y;
    }
    store10(out + n * D, y);
    return;
  }
  // second Phi layer: mess = W2 S + deg b2
  v2f mt[5], mf[5], b[5];
  ld5(W + dsl::B2_TO, b);
#pragma unroll
  for (int p = 0; p < 5; ++p) mt[p] = splat(deg_in) * b[p];
  PHASE();
  mv2<D>(W + dsl::W2_TO, reinterpret_cast<const float*>(S_to), mt);
  ld5(W + dsl::B2_FR, b);
#pragma unroll
  for (int p = 0; p < 5; ++p) mf[p] = splat(deg_out) * b[p];
  PHASE();
  mv2<D>(W + dsl::W2_FR, reinterpret_cast<const float*>(S_fr), mf);
  const float* mto = reinterpret_cast<const float*>(mt);
  const float* mfr = reinterpret_cast<const float*>(mf);
  float pq[P];
#pragma unroll
  for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
  // gates: rows of the transposed (32, 10) blocks are [h 0..9 | mess_to 10..19 | mess_from 20..29 | prb 30..31]
  v2f z[5], r[5], c[5];
  ld5(W + dsl::BZ, z);
  PHASE();
  mv2<D>(W + dsl::WZ, x, z);
  PHASE();
  mv2<D>(W + dsl::WZ + 10 * D, mto, z);
  PHASE();
  mv2<D>(W + dsl::WZ + 20 * D, mfr, z);
  PHASE();
  mv2<P>(W + dsl::WZ + 30 * D, pq, z);
  ld5(W + dsl::BR, r);
  PHASE();
  mv2<D>(W + dsl::WR, x, r);
  PHASE();
  mv2<D>(W + dsl::WR + 10 * D, mto, r);
  PHASE();
  mv2<D>(W + dsl::WR + 20 * D, mfr, r);
  PHASE();
  mv2<P>(W + dsl::WR + 30 * D, pq, r);
  float rh[D];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    rh[2 * p] = x[2 * p] * fast_sigmoid(r[p].x);
    rh[2 * p + 1] = x[2 * p + 1] * fast_sigmoid(r[p].y);
  }
  ld5(W + dsl::BC, c);
  PHASE();
  mv2<D>(W + dsl::WC, rh, c);
  PHASE();
  mv2<D>(W + dsl::WC + 10 * D, mto, c);
  PHASE();
  mv2<D>(W + dsl::WC + 20 * D, mfr, c);
  PHASE();
  mv2<P>(W + dsl::WC + 30 * D, pq, c);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    y[2 * p] = fmaf(fast_sigmoid(z[p].x), fast_tanh(c[p].x), x[2 * p]);
    y[2 * p + 1] = fmaf(fast_sigmoid(z[p].y), fast_tanh(c[p].y), x[2 * p + 1]);
  }
  store10(out + n * D, y);
}

extern "C" int64_t psignn_dsgps_weights_size(int mixed) { return mixed ? DsL<3>::TOTAL : DsL<2>::TOTAL; }

static void dsgps_launch(const psignn_plan* p, const float* W, const float* cur, const float* h0p, const float* prbp,
                         const float* nrmp, float* dst, hipStream_t st) {
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = (unsigned)(chunk * 8);
  if (p->mixed) {
    LAUNCH("k_dsgps_tile", st, (k_dsgps_tile<3, true><<<grid, TILE_THREADS, (size_t)p->max_rows * 32 * 4, st>>>(
        (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell,
        p->flags_p, W, cur, h0p, prbp, nrmp, dst)));
  } else {
    LAUNCH("k_dsgps_tile", st, (k_dsgps_tile<2, false><<<grid, TILE_THREADS, (size_t)p->max_rows * 20 * 4, st>>>(
        (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell,
        p->flags_p, W, cur, h0p, prbp, nrmp, dst)));
  }
}

// k updates from d_h0 (the encoder state; also the Dirichlet rows of every iterate).  d_h0, d_prb, d_normals, d_out in
// the caller's numbering (d_normals: mixed plans only); d_work: 4 * N * 10 floats.  k = 0 copies d_h0.
extern "C" int psignn_dsgps_forward(const psignn_plan_t* p, const float* W, int k, const float* d_h0, const float* d_prb,
                                    const float* d_normals, float* d_out, float* d_work, void* stream) {
  ARG_CHECK(p && W && d_h0 && d_prb && d_out && d_work, "NULL argument");
  ARG_CHECK(k >= 0, "negative step count");
  ARG_CHECK(!p->mixed || d_normals, "mixed plan needs unit normals");
  ARG_CHECK(p->tiled, "DS-GPS kernels need a tiled plan (mesh positions)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = p->N;
  const int P = p->mixed ? 3 : 2;
  float* h0p = d_work;
  float* a = h0p + N * D;
  float* b = a + N * D;
  float* prbp = b + N * D;   // (N, P)
  float* nrmp = prbp + N * P;  // (N, 2), mixed
  int rc;
  if ((rc = psignn_plan_permute(p, d_h0, D, h0p, 1, stream))) return rc;
  if ((rc = psignn_plan_permute(p, d_prb, P, prbp, 1, stream))) return rc;
  if (p->mixed && (rc = psignn_plan_permute(p, d_normals, 2, nrmp, 1, stream))) return rc;
  const float* cur = h0p;
  for (int i = 0; i < k; ++i) {
    float* dst = (i & 1) ? b : a;
    dsgps_launch(p, W, cur, h0p, prbp, p->mixed ? nrmp : nullptr, dst, st);
    cur = dst;
  }
  HIP_TRY(hipGetLastError());
  return psignn_plan_permute(p, cur, D, d_out, 0, stream);
}

// One update in PLAN order (state, H_0, prb and normals already permuted): for callers that keep every iterate
// (ModelDSGPS.forward records a loss per step, model.py:64-118).
extern "C" int psignn_dsgps_step_p(const psignn_plan_t* p, const float* W, const float* d_h, const float* d_h0,
                                   const float* d_prb, const float* d_normals, float* d_out, void* stream) {
  ARG_CHECK(p && W && d_h && d_h0 && d_prb && d_out, "NULL argument");
  ARG_CHECK(p->tiled, "DS-GPS kernels need a tiled plan");
  ARG_CHECK(!p->mixed || d_normals, "mixed plan needs unit normals");
  ARG_CHECK(d_out != d_h, "out must not alias the state");
  dsgps_launch(p, W, d_h, d_h0, d_prb, d_normals, d_out, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
