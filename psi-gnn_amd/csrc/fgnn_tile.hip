// f_theta on the tiled plan: one workgroup per mesh tile, neighbour projections staged in LDS (gfx950).
//
// Same arithmetic as fgnn.hip (reference: dirichlet/psignn/model.py:279-300, mixed/psignn/model.py:216-245),
// different data movement:
//   stage 1  every lane takes one row of the tile + halo (<= 256 + HALO_CAP rows): loads h (tile rows are
//            one coalesced 40 B/lane stream, halo rows a short indexed gather), projects it with the
//            neighbour-side weights W1j_{to,from[,neu]} and parks the result in LDS
//   stage 2  every lane owns one tile node: walks its sliced-ELL in- and out-lists (coalesced uint16 LDS row
//            ids + SoA edge_attr), gathers the projected neighbour rows with ds_read_b128, sums relu terms in
//            the plan's canonical order (no atomics), then gate / update MLP / LayerNorm / boundary rows
// All node-level tensors are in PLAN order (plan->perm); the solver keeps its state in that order.
#include "fgnn_common.h"

template <bool MIXED>
struct TileRow {
  static constexpr int RS = MIXED ? 36 : 24;   // floats per LDS row: [to 10|pad 2][from 10|pad 2]([neu 10|pad 2])
};

__device__ __forceinline__ void lds_store10(float* __restrict__ p, const float* __restrict__ t) {
  reinterpret_cast<float4*>(p)[0] = make_float4(t[0], t[1], t[2], t[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(t[4], t[5], t[6], t[7]);
  reinterpret_cast<float2*>(p)[4] = make_float2(t[8], t[9]);
}
__device__ __forceinline__ void lds_load10(const float* __restrict__ p, float* __restrict__ t) {
  float4 a = reinterpret_cast<const float4*>(p)[0];
  float4 b = reinterpret_cast<const float4*>(p)[1];
  float2 c = reinterpret_cast<const float2*>(p)[4];
  t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w;
  t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
  t[8] = c.x; t[9] = c.y;
}

// S[o] = sum over the lane's ELL slots of relu(Pi[o] + Pj[row][o] + W1a[o,:] . attr)
template <int RS>
__device__ __forceinline__ int ell_sum(int64_t row0, int nslots, int lane, const uint16_t* __restrict__ ell_idx,
                                       const float* __restrict__ ell_attr, const float* __restrict__ W1, int ld,
                                       const float* __restrict__ lds, int col, const float* Pi, float* S) {
  int deg = 0;
#pragma unroll
  for (int o = 0; o < D; ++o) S[o] = 0.f;
  for (int r = 0; r < nslots; ++r) {
    const int64_t row = row0 + r;
    const unsigned li = ell_idx[row * 64 + lane];
    const float a0 = ell_attr[(row * 3 + 0) * 64 + lane];
    const float a1 = ell_attr[(row * 3 + 1) * 64 + lane];
    const float a2 = ell_attr[(row * 3 + 2) * 64 + lane];
    if (li != ELL_EMPTY) {
      ++deg;
      float pj[D];
      lds_load10(lds + (int)li * RS + col, pj);
#pragma unroll
      for (int o = 0; o < D; ++o) {
        float z = Pi[o] + pj[o];
        z = fmaf(W1[o * ld + 2 * D + 0], a0, z);
        z = fmaf(W1[o * ld + 2 * D + 1], a1, z);
        z = fmaf(W1[o * ld + 2 * D + 2], a2, z);
        S[o] += fmaxf(z, 0.f);
      }
    }
  }
  return deg;  // the node's real degree (padding slots excluded), for the deg * b2 term
}

template <int P, bool MIXED>
__global__ __launch_bounds__(256) void k_f_tile(int n_tiles, int chunk, const int32_t* __restrict__ tile_ptr,
                                                const int32_t* __restrict__ tile_slice, const int32_t* __restrict__ halo,
                                                const int32_t* __restrict__ halo_cnt, const int32_t* __restrict__ slice_off,
                                                const uint8_t* __restrict__ slice_deg, const uint16_t* __restrict__ ell_idx,
                                                const float* __restrict__ ell_attr, const uint8_t* __restrict__ flags,
                                                const float* __restrict__ W, int lofs, int nofs, int unofs, int apply_ln,
                                                const float* __restrict__ h, const int32_t* __restrict__ hsel,
                                                int64_t hstride, const float* __restrict__ h0,
                                                const float* __restrict__ prb, const float* __restrict__ nrm,
                                                float* __restrict__ out) {
  using L = WLayout<P>;
  constexpr int RS = TileRow<MIXED>::RS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so that
  // neighbouring tiles' halo rows hit the same L2.  Speed only; any mapping is correct.
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  if (hsel) h += (int64_t)(*hsel) * hstride;

  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  const float* Wn = W + nofs;

  // ---- stage 1: neighbour-side projections of tile + halo rows -> LDS
  float x[D];
  for (int row = tid; row < n_t + n_h; row += 256) {
    const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
    float xr[D], t[D];
    load10(h + node * D, xr);
    if (row == tid) {
#pragma unroll
      for (int o = 0; o < D; ++o) x[o] = xr[o];
    }
    float* dst = lds + row * RS;
    matvec10<D, false>(Wto + L::PHI_W1, L::EIN, D, xr, t);
    lds_store10(dst, t);
    matvec10<D, false>(Wfr + L::PHI_W1, L::EIN, D, xr, t);
    lds_store10(dst + 12, t);
    if (MIXED) {
      matvec10<D, false>(Wn + L::PHI_W1, L::EIN, D, xr, t);
      lds_store10(dst + 24, t);
    }
  }
  __syncthreads();
  if (tid >= n_t) return;

  // ---- stage 2: one tile node per lane
  const int64_t n = (int64_t)t0 + tid;
  const uint8_t fl = flags[n];
  if (fl & FLAG_DIRICHLET) {  // Dirichlet rows <- h_initial rows (model.py:298)
    float r[D];
    load10(h0 + n * D, r);
    store10(out + n * D, r);
    return;
  }
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (tid >> 6);
  const int64_t row0 = slice_off[slice];
  const int din = slice_deg[2 * slice], dout = slice_deg[2 * slice + 1];

  float Pi[D], S[D], mp_to[D], mp_fr[D];
  // Phi_to: in-edges, aggregated at the column index
#pragma unroll
  for (int o = 0; o < D; ++o) Pi[o] = Wto[L::PHI_B1 + o];
  matvec10<D, true>(Wto + L::PHI_W1, L::EIN, 0, x, Pi);
  const int deg_in = ell_sum<RS>(row0, din, lane, ell_idx, ell_attr, Wto + L::PHI_W1, L::EIN, lds, 0, Pi, S);
#pragma unroll
  for (int o = 0; o < D; ++o) mp_to[o] = (float)deg_in * Wto[L::PHI_B2 + o];
  matvec10<D, true>(Wto + L::PHI_W2, D, 0, S, mp_to);
  // Phi_from: out-edges, aggregated at the row index
#pragma unroll
  for (int o = 0; o < D; ++o) Pi[o] = Wfr[L::PHI_B1 + o];
  matvec10<D, true>(Wfr + L::PHI_W1, L::EIN, 0, x, Pi);
  const int deg_out = ell_sum<RS>(row0 + din, dout, lane, ell_idx, ell_attr, Wfr + L::PHI_W1, L::EIN, lds, 12, Pi, S);
#pragma unroll
  for (int o = 0; o < D; ++o) mp_fr[o] = (float)deg_out * Wfr[L::PHI_B2 + o];
  matvec10<D, true>(Wfr + L::PHI_W2, D, 0, S, mp_fr);

  float y[D];
  if (MIXED && (fl & FLAG_NEUMANN)) {
    // Phi_neumann (Phi_from type) + update_neumann: the row is REPLACED (mixed/psignn/model.py:236,241)
    const float* Un = W + unofs;
    float mp_n[D], hid[D];
#pragma unroll
    for (int o = 0; o < D; ++o) Pi[o] = Wn[L::PHI_B1 + o];
    matvec10<D, true>(Wn + L::PHI_W1, L::EIN, 0, x, Pi);
    ell_sum<RS>(row0 + din, dout, lane, ell_idx, ell_attr, Wn + L::PHI_W1, L::EIN, lds, 24, Pi, S);
#pragma unroll
    for (int o = 0; o < D; ++o) mp_n[o] = (float)deg_out * Wn[L::PHI_B2 + o];
    matvec10<D, true>(Wn + L::PHI_W2, D, 0, S, mp_n);
#pragma unroll
    for (int o = 0; o < D; ++o) hid[o] = Un[L::NEU_B1 + o];
    matvec10<D, true>(Un + L::NEU_W1, L::NEU_CAT, 0, x, hid);
    matvec10<D, true>(Un + L::NEU_W1, L::NEU_CAT, D, mp_n, hid);
    float pq[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    pq[P] = nrm[n * 2];
    pq[P + 1] = nrm[n * 2 + 1];
    matvec10<P + 2, true>(Un + L::NEU_W1, L::NEU_CAT, 2 * D, pq, hid);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      hid[o] = fmaxf(hid[o], 0.f);
      y[o] = Un[L::NEU_B2 + o];
    }
    matvec10<D, true>(Un + L::NEU_W2, D, 0, hid, y);
  } else {
    // gate + update MLP on cat = [h | mp_to | mp_from | prb]
    const float* Wu = W + lofs + L::L_UPD;
    const float* Wa = W + L::AL_W;
    float pq[P];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    float al = W[L::AL_B];
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[k], x[k], al);
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[D + k], mp_to[k], al);
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[2 * D + k], mp_fr[k], al);
#pragma unroll
    for (int k = 0; k < P; ++k) al = fmaf(Wa[3 * D + k], pq[k], al);
    al = 1.f / (1.f + expf(-al));
    float hid[D], upd[D];
#pragma unroll
    for (int o = 0; o < D; ++o) hid[o] = Wu[L::UPD_B1 + o];
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 0, x, hid);
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, D, mp_to, hid);
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 2 * D, mp_fr, hid);
    matvec10<P, true>(Wu + L::UPD_W1, L::CAT, 3 * D, pq, hid);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      hid[o] = fmaxf(hid[o], 0.f);
      upd[o] = Wu[L::UPD_B2 + o];
    }
    matvec10<D, true>(Wu + L::UPD_W2, D, 0, hid, upd);
#pragma unroll
    for (int o = 0; o < D; ++o) y[o] = fmaf(al, upd[o], x[o]);
  }
  if (apply_ln) {  // LayerNorm(10), eps 1e-5, biased variance, affine (model.py:293)
    float mu = 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) mu += y[o];
    mu *= (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) {
      float c = y[o] - mu;
      var = fmaf(c, c, var);
    }
    var *= (1.f / D);
    float rs = 1.f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int o = 0; o < D; ++o) y[o] = fmaf((y[o] - mu) * rs, W[L::LN_G + o], W[L::LN_B + o]);
  }
  store10(out + n * D, y);
}

// gather / scatter of node rows between the caller's numbering and the plan order
__global__ void k_permute_rows(int64_t N, int cols, const int32_t* __restrict__ map, const float* __restrict__ src,
                               float* __restrict__ dst, int scatter) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cols) return;
  int64_t r = i / cols, c = i - r * cols;
  int64_t m = map[r];
  if (scatter) dst[m * cols + c] = src[i];
  else dst[i] = src[m * cols + c];
}

// ------------------------------------------------------------------------------------------ host
// All tensors in plan order.  h = hbase + (*hsel) * hstride when hsel != NULL.
int psignn_f_tile_forward(const psignn_plan* p, const float* W, int nl, const float* h, const int32_t* hsel,
                          int64_t hstride, const float* h0, const float* prb, const float* nrm, float* out,
                          float* work, hipStream_t st) {
  ARG_CHECK(p && p->tiled, "plan has no tile structures");
  ARG_CHECK(W && h && h0 && prb && out, "NULL argument");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = (unsigned)(chunk * 8);
  if (p->mixed) {
    using L = WLayout<3>;
    size_t lds = (size_t)p->max_rows * TileRow<true>::RS * 4;
    LAUNCH("k_f_tile", st, (k_f_tile<3, true><<<grid, 256, lds, st>>>(
        (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg,
        p->ell_idx, p->ell_attr, p->flags_p, W, L::layer(nl - 1), L::phi_neu(nl), L::upd_neu(nl), 1, h, hsel, hstride,
        h0, prb, nrm, out)));
  } else {
    using L = WLayout<2>;
    size_t lds = (size_t)p->max_rows * TileRow<false>::RS * 4;
    ARG_CHECK(nl == 1 || work, "multi-layer evaluation needs a workspace");
    float* pp[2] = {work, work ? work + p->N * D : nullptr};
    const float* cur = h;
    for (int l = 0; l < nl; ++l) {
      float* dst = (l == nl - 1) ? out : pp[l & 1];
      LAUNCH("k_f_tile", st, (k_f_tile<2, false><<<grid, 256, lds, st>>>(
          (int)p->n_tiles, chunk, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg,
          p->ell_idx, p->ell_attr, p->flags_p, W, L::layer(l), 0, 0, l == nl - 1, cur, l == 0 ? hsel : nullptr, hstride,
          h0, prb, nrm, dst)));
      cur = dst;
    }
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// dst[new] = src[perm[new]]  (to_plan = 1)   or   dst[old] = src[inv[old]]  (to_plan = 0); rows of `cols` floats
extern "C" int psignn_plan_permute(const psignn_plan_t* p, const float* src, int cols, float* dst, int to_plan, void* stream) {
  ARG_CHECK(p && src && dst && cols > 0, "bad arguments");
  ARG_CHECK(src != dst, "in-place permutation is not supported");
  hipStream_t st = (hipStream_t)stream;
  if (!p->tiled) {
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)p->N * cols * 4, hipMemcpyDeviceToDevice, st));
    return PSIGNN_OK;
  }
  int64_t n = p->N * cols;
  k_permute_rows<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(p->N, cols, to_plan ? p->perm : p->inv, src, dst, 0);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
