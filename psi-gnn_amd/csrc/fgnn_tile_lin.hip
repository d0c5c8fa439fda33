// Linearised GNN block for Newton-Krylov (BASELINE config 5: 1M nodes, GMRES on J_f(h) - I; SURVEY section 8a "JVP needed for
// config 5 ... dm_e = W2(1[z>0] (W1i v_i + W1j v_j)) ...", section 8d "B_jvp (+ N 40 if h-dependent masks are recomputed rather
// than stored)").  One Newton step runs tens of GMRES iterations at a FIXED state h: every J_f(h) v of those re-derives, in
// k_jvp_tile, the whole value path of f -- pre-activations of 12 M edge directions, gate, update MLP, LayerNorm statistics -- only
// to obtain the relu masks and a few per-node scalars, which do not depend on v.  Here they are computed once per Newton step and
// stored (psignn_lin_build), and the product is the LINEAR operator they define (psignn_lin_jvp):
//
//   per slot row s (one ELL row of one 64-lane slice) and lane l, ONE dword instead of the 16-byte slot record:
//     slot[s][l] = LDS row of the neighbour (bits 0..9; 0 for an empty slot, whose masks are 0)
//                | 1[slot carries an in-edge  and z_to[o]   > 0] << (10 + o)      o < 10   (Phi_to pre-activations)
//                | 1[slot carries an out-edge and z_from[o] > 0] << (20 + o)               (Phi_from)
//   per node (24 floats): alpha, alpha (1 - alpha), 1 / sqrt(var + eps), bit mask of the update MLP's hidden relu, update[10], y_hat[10]
//
//   J v = LN'( v + alpha (1 - alpha) (w_a . dc) update + alpha U2 (1[q>0] (U1 dc)) ),  dc = [v, G_to dS_to, G_fr dS_fr],
//   dS_dir[o] = sum over the node's slots of mask (dPi_dir[o] + dPj_dir[o]),  dP = W1 v  (second Phi layer folded as in k_f_tile)
//
// Against k_jvp_tile at 1M nodes: ~0.4 x the VALU instructions per wave, 80-byte LDS rows (tangent projections only: six
// workgroups per CU instead of three), about the same bytes (the 96-byte node record replaces the h row, the 16-byte slot records
// shrink to 4 bytes).  Dirichlet plans with a single-layer block; mixed plans: the tiles WITHOUT Neumann nodes (all but the boundary
// tiles) go through the stored linearisation, the few tiles holding Neumann nodes through k_jvp_tile at the state kept from the build
// (a Neumann row needs a third mask set per slot; not worth a second record format for ~2 % of the tiles).
// weight loads of mv2 pinned chunk by chunk (tile_helpers.h; A/B in profiles/r3_ab_mv2.txt: k_jvp_lin 53 -> 48.5 us)
#ifndef MV2_LAUNDER
#define MV2_LAUNDER 2
#endif
#include "tile_helpers.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#define LIN_REC 24            // floats per node record
#define LIN_CHUNK 8           // slot dwords requested at once by the product kernel (a slice has ~6 slot rows)

struct psignn_lin {
  const psignn_plan* plan = nullptr;
  uint32_t* slot = nullptr;     // (ell_rows, 64)
  float* rec = nullptr;         // (N, LIN_REC)
  float *h = nullptr, *prb = nullptr, *nrm = nullptr;   // mixed plans: the state of the last build (k_jvp_tile on the Neumann tiles)
  size_t bytes = 0;
  int built = 0;
};

// Stage 1 of both kernels: rows [Pj_to | Pj_from] = W1j_{to,from} src[node] of the tile's own and halo nodes -> 80-byte LDS rows
// (own row by its lane; halo rows as HALF rows over all four waves, as in k_f_tile).  Returns the lane's own src row in x.
template <int P>
__device__ __forceinline__ void lin_stage1(const float* __restrict__ T, const float* __restrict__ src, const int32_t t0, const int n_t,
                                           const int n_h, const int32_t* __restrict__ hl, float* __restrict__ lds, float* x) {
  using L = WLayout<P>;
  constexpr int RS = 20;
  const int tid = threadIdx.x;
  const int32_t hidx_w = (tid >> 6 & 1) * 64 + (tid & 63);
  int32_t hnode = 0;
  if (hidx_w < n_h) hnode = hl[hidx_w];
  float xr[D], xh[D];
  if (tid < n_t) load10(src + (int64_t)(t0 + tid) * D, xr);
  if (hidx_w < n_h) load10(src + (int64_t)hnode * D, xh);
  if (tid < n_t) {
#pragma unroll
    for (int o = 0; o < D; ++o) x[o] = xr[o];
    v2f ta[5], tb[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ta[p] = tb[p] = splat(0.f);
    PHASE();
    mv2<D>(T + L::T_W1J_TO, xr, ta);
    PHASE();
    mv2<D>(T + L::T_W1J_FR, xr, tb);
    float4* q = reinterpret_cast<float4*>(lds + tid * RS);
    q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
    q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
    q[2] = make_float4(ta[4].x, ta[4].y, tb[0].x, tb[0].y);
    q[3] = make_float4(tb[1].x, tb[1].y, tb[2].x, tb[2].y);
    q[4] = make_float4(tb[3].x, tb[3].y, tb[4].x, tb[4].y);
  }
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);   // 0: Phi_to columns, 1: Phi_from columns
  for (int hb = 0; hb < n_h; hb += 128) {
    const int idx = hb + hidx_w;
    if (hb > 0 && idx < n_h) hnode = hl[idx];
    if (idx < n_h) {
      float xq[D];
      if (hb == 0) {
#pragma unroll
        for (int o = 0; o < D; ++o) xq[o] = xh[o];
      } else {
        load10(src + (int64_t)hnode * D, xq);
      }
      v2f ta[5];
#pragma unroll
      for (int p = 0; p < 5; ++p) ta[p] = splat(0.f);
      PHASE();
      float* rowp = lds + (n_t + idx) * RS;
      if (half == 0) {
        mv2<D>(T + L::T_W1J_TO, xq, ta);
        float4* q = reinterpret_cast<float4*>(rowp);
        q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
        q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
        reinterpret_cast<float2*>(rowp + 8)[0] = make_float2(ta[4].x, ta[4].y);
      } else {
        mv2<D>(T + L::T_W1J_FR, xq, ta);
        reinterpret_cast<float2*>(rowp + 10)[0] = make_float2(ta[0].x, ta[0].y);
        float4* q = reinterpret_cast<float4*>(rowp + 12);
        q[0] = make_float4(ta[1].x, ta[1].y, ta[2].x, ta[2].y);
        q[1] = make_float4(ta[3].x, ta[3].y, ta[4].x, ta[4].y);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Build: the value path of f at h (same formulas and operation order as k_jvp_tile's value half), storing what J_f(h) needs.
// ------------------------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(TILE_THREADS) void k_lin_build(int n_tiles, int chunk, const int32_t* __restrict__ tile_list, const TileCtx C,
                                                            const float* __restrict__ W,
                                                            int lofs, int tofs, const float* __restrict__ h,
                                                            const float* __restrict__ prb, uint32_t* __restrict__ slot,
                                                            float* __restrict__ rec) {
  using L = WLayout<P>;
  constexpr int RS = 20;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int slot_ = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot_ >= n_tiles) return;
  const int tile = tile_list ? tile_list[slot_] : slot_;   // mixed plans: the tiles without Neumann nodes
  const int tid = threadIdx.x;
  const int tn = C.tile_nodes;
  const int32_t t0 = tn ? tile * tn : C.tile_ptr[tile];
  const int n_t = tn ? min(tn, C.n_nodes - t0) : C.tile_ptr[tile + 1] - t0;
  const int n_h = C.halo_cnt[tile];
  const int32_t* hl = C.halo + (int64_t)tile * HALO_CAP;
  const float* T = W + tofs;
  float x[D];
#pragma unroll
  for (int o = 0; o < D; ++o) x[o] = 0.f;
  lin_stage1<P>(T, h, t0, n_t, n_h, hl, lds, x);
  __syncthreads();
  if (tid >= n_t) return;
  const int lane = tid & 63;
  const int64_t n = (int64_t)t0 + tid;
  if (C.flags_p[n] & FLAG_DIRICHLET) return;   // (the product kernel writes zeros for such a row without looking at its slots)
  const int slice = __builtin_amdgcn_readfirstlane((tn ? tile * (tn >> 6) : C.tile_slice[tile]) + (tid >> 6));
  const int srow0 = C.slice_off[slice];
  const int nslots = C.slice_deg[slice];
  const uint4* slots = C.ell + (int64_t)srow0 * 64 + lane;
  uint32_t* so = slot + (int64_t)srow0 * 64 + lane;
  // ---- neighbour sums and masks
  v2f S_to[5], S_fr[5], pt[5], pf[5], wt[15], wf[15];
  {
    v2f Pi[5], Pi2[5];
    ld5(T + L::T_B1_TO, Pi);
    ld5(T + L::T_B1_FR, Pi2);
    PHASE();
    mv2<D>(T + L::T_W1I_TO, x, Pi);
    PHASE();
    mv2<D>(T + L::T_W1I_FR, x, Pi2);
    const v2f sc = splat(RELU_SCALE);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      pt[p] = Pi[p] * sc;
      pf[p] = Pi2[p] * sc;
      S_to[p] = S_fr[p] = splat(0.f);
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      wt[i] = reinterpret_cast<const v2f*>(T + L::T_A_TO)[i];
      wf[i] = reinterpret_cast<const v2f*>(T + L::T_A_FR)[i];
    }
  }
  float deg_in = 0.f, deg_out = 0.f;
  const v2f sc = splat(RELU_SCALE);
  uint4 c0 = nslots > 0 ? slots[0] : make_uint4(ELL_EMPTY, 0u, 0u, 0u);
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 1, nslots - 1) * 64];
    const unsigned w = c0.x;
    unsigned word = 0;
    if ((w & 0xFFFFu) != ELL_EMPTY) {
      word = w & 0xFFFFu;
      const v2f a01 = (v2f){__uint_as_float(c0.y), __uint_as_float(c0.z)} * sc;
      const v2f a2 = (v2f){__uint_as_float(c0.w) * RELU_SCALE, 0.f};
      const float4* rp = reinterpret_cast<const float4*>(lds + (int)word * RS);
      const float4 v0 = rp[0], v1 = rp[1], v2 = rp[2], v3 = rp[3], v4 = rp[4];
      if (w & SLOT_IN) {
        v2f z[5] = {(v2f){v0.x, v0.y}, (v2f){v0.z, v0.w}, (v2f){v1.x, v1.y}, (v2f){v1.z, v1.w}, (v2f){v2.x, v2.y}};
        deg_in += 1.f;
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(z[p], sc, pt[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_lo(wt[p], a01, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_hi(wt[5 + p], a01, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_lo_clamp(wt[10 + p], a2, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) {   // clamp(2^-40 z) > 0  <=>  z > 0, as k_jvp_tile's mask
          S_to[p] += z[p];
          word |= (z[p].x > 0.f ? 1u : 0u) << (10 + 2 * p);
          word |= (z[p].y > 0.f ? 1u : 0u) << (11 + 2 * p);
        }
      }
      if (w & SLOT_OUT) {
        v2f z[5] = {(v2f){v2.z, v2.w}, (v2f){v3.x, v3.y}, (v2f){v3.z, v3.w}, (v2f){v4.x, v4.y}, (v2f){v4.z, v4.w}};
        deg_out += 1.f;
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(z[p], sc, pf[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_lo(wf[p], a01, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_hi(wf[5 + p], a01, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = pk_fma_lo_clamp(wf[10 + p], a2, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) {
          S_fr[p] += z[p];
          word |= (z[p].x > 0.f ? 1u : 0u) << (20 + 2 * p);
          word |= (z[p].y > 0.f ? 1u : 0u) << (21 + 2 * p);
        }
      }
    }
    so[(int64_t)r * 64] = word;
    c0 = nx;
  }
  const v2f us = splat(RELU_UNSCALE);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    S_to[p] *= us;
    S_fr[p] *= us;
  }
  // ---- gate and update MLP (second Phi layer folded), LayerNorm statistics: as k_jvp_tile
  const float* Wf = W + lofs + L::L_FOLD;
  const float* Wa = W + L::AL_W;
  const float* sto = reinterpret_cast<const float*>(S_to);
  const float* sfr = reinterpret_cast<const float*>(S_fr);
  float pq[P];
#pragma unroll
  for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
  PHASE();
  float al = fmaf(deg_in, Wf[L::F_ABTO], fmaf(deg_out, Wf[L::F_ABFR], W[L::AL_B]));
#pragma unroll
  for (int k = 0; k < D; ++k) al = fmaf(Wa[k], x[k], al);
#pragma unroll
  for (int k = 0; k < D; ++k) al = fmaf(Wf[L::F_ATO + k], sto[k], al);
#pragma unroll
  for (int k = 0; k < D; ++k) al = fmaf(Wf[L::F_AFR + k], sfr[k], al);
#pragma unroll
  for (int k = 0; k < P; ++k) al = fmaf(Wa[3 * D + k], pq[k], al);
  al = 1.f / (1.f + expf(-al));
  v2f q[5], g1[5], g2[5], upd[5];
  ld5(T + L::T_HB, q);
  ld5(T + L::T_gTO, g1);
  ld5(T + L::T_gFR, g2);
#pragma unroll
  for (int p = 0; p < 5; ++p)
    q[p] = __builtin_elementwise_fma(splat(deg_in), g1[p], __builtin_elementwise_fma(splat(deg_out), g2[p], q[p]));
  PHASE();
  mv2<D>(T + L::T_U1H, x, q);
  PHASE();
  mv2<D>(T + L::T_GTO, sto, q);
  PHASE();
  mv2<D>(T + L::T_GFR, sfr, q);
  mv2<P>(T + L::T_U1P, pq, q);
  unsigned hm = 0;
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    hm |= (q[p].x > 0.f ? 1u : 0u) << (2 * p);
    hm |= (q[p].y > 0.f ? 1u : 0u) << (2 * p + 1);
    q[p] = __builtin_elementwise_max(q[p], splat(0.f));
  }
  ld5(T + L::T_C2, upd);
  PHASE();
  mv2<D>(T + L::T_U2, reinterpret_cast<const float*>(q), upd);
  const float* u = reinterpret_cast<const float*>(upd);
  float y[D], mu = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = fmaf(al, u[o], x[o]);
    mu += y[o];
  }
  mu *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    const float c = y[o] - mu;
    var = fmaf(c, c, var);
  }
  var *= (1.f / D);
  const float rs = 1.f / sqrtf(var + 1e-5f);
  float4* rp = reinterpret_cast<float4*>(rec + n * LIN_REC);
  float yh[D];
#pragma unroll
  for (int o = 0; o < D; ++o) yh[o] = (y[o] - mu) * rs;
  rp[0] = make_float4(al, al * (1.f - al), rs, __uint_as_float(hm));
  rp[1] = make_float4(u[0], u[1], u[2], u[3]);
  rp[2] = make_float4(u[4], u[5], u[6], u[7]);
  rp[3] = make_float4(u[8], u[9], yh[0], yh[1]);
  rp[4] = make_float4(yh[2], yh[3], yh[4], yh[5]);
  rp[5] = make_float4(yh[6], yh[7], yh[8], yh[9]);
}

// ------------------------------------------------------------------------------------------------------------------
// Product: out = J_f(h) v from the stored linearisation
// ------------------------------------------------------------------------------------------------------------------
#ifndef LIN_PRIO
#define LIN_PRIO 0   // 1: s_setprio(3) through stage 1 and the slot walk, 0 from the node update on (as k_f_tile's TILE_PRIO = 15)
#endif
#ifndef LIN_REC_EARLY
#define LIN_REC_EARLY 0   // 1: node record requested before the slot walk (127 VGPRs, four waves per SIMD)
#endif
#ifndef LIN_WAVES
#define LIN_WAVES 0   // > 0: hold the register allocator to this many waves per SIMD
#endif
#if LIN_WAVES > 0
#define LIN_OCC __attribute__((amdgpu_waves_per_eu(LIN_WAVES, LIN_WAVES)))
#else
#define LIN_OCC
#endif
template <int P>
__global__ __launch_bounds__(TILE_THREADS) LIN_OCC void k_jvp_lin(int n_tiles, int chunk, const int32_t* __restrict__ tile_list,
                                                                  const TileCtx C, const float* __restrict__ W,
                                                                  int lofs, int tofs, const uint32_t* __restrict__ slot,
                                                                  const float* __restrict__ rec,
                                                                  const float* __restrict__ tv, float* __restrict__ out) {
  using L = WLayout<P>;
  constexpr int RS = 20;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int slot_ = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot_ >= n_tiles) return;
  const int tile = tile_list ? tile_list[slot_] : slot_;
  const int tid = threadIdx.x;
  const int tn = C.tile_nodes;
  const int32_t t0 = tn ? tile * tn : C.tile_ptr[tile];
  const int n_t = tn ? min(tn, C.n_nodes - t0) : C.tile_ptr[tile + 1] - t0;
  const int n_h = C.halo_cnt[tile];
  const int32_t* hl = C.halo + (int64_t)tile * HALO_CAP;
  const float* T = W + tofs;
  float dx[D];
  if (LIN_PRIO) __builtin_amdgcn_s_setprio(3);
  lin_stage1<P>(T, tv, t0, n_t, n_h, hl, lds, dx);
  __syncthreads();
  if (tid >= n_t) return;
  const int64_t n = (int64_t)t0 + tid;
  float dy[D];
  if (C.flags_p[n] & FLAG_DIRICHLET) {
#pragma unroll
    for (int o = 0; o < D; ++o) dy[o] = 0.f;
    store10(out + n * D, dy);
    return;
  }
  const int lane = tid & 63;
  // (wave-uniform by construction; said so to the compiler: scalar loads, scalar branches in the walk)
  const int slice = __builtin_amdgcn_readfirstlane((tn ? tile * (tn >> 6) : C.tile_slice[tile]) + (tid >> 6));
  const int srow0 = C.slice_off[slice];
  const int nslots = C.slice_deg[slice];
  const uint32_t* si = slot + (int64_t)srow0 * 64 + lane;
  // all slot dwords of the node are requested at once (clamped index: unconditional loads), ahead of the own-side projections
  uint32_t sw[LIN_CHUNK];
#pragma unroll
  for (int i = 0; i < LIN_CHUNK; ++i) sw[i] = 0u;
  if (nslots > 0) {
#pragma unroll
    for (int i = 0; i < LIN_CHUNK; ++i) sw[i] = si[(int64_t)min(i, nslots - 1) * 64];
  }
  const float4* rp = reinterpret_cast<const float4*>(rec + n * LIN_REC);
#if LIN_REC_EARLY
  const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3], r4 = rp[4], r5 = rp[5];   // requested ahead of the walk
#endif
  v2f dPt[5], dPf[5], dS_to[5], dS_fr[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) dPt[p] = dPf[p] = dS_to[p] = dS_fr[p] = splat(0.f);
  PHASE();
  mv2<D>(T + L::T_W1I_TO, dx, dPt);
  PHASE();
  mv2<D>(T + L::T_W1I_FR, dx, dPf);
  for (int r0 = 0; r0 < nslots; r0 += LIN_CHUNK) {
#pragma unroll
    for (int i = 0; i < LIN_CHUNK; ++i) {
      if (r0 + i < nslots) {   // wave-uniform
        const uint32_t w = sw[i];
        const float4* q = reinterpret_cast<const float4*>(lds + (int)(w & 1023u) * RS);
        const float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
        const float d[20] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y,
                             v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int p = 0; p < 5; ++p) {
          const v2f a = (v2f){d[2 * p], d[2 * p + 1]} + dPt[p];
          const v2f b = (v2f){d[10 + 2 * p], d[11 + 2 * p]} + dPf[p];
          // (v_bfe_i32: the mask bit as 0 / ~0, and-ed into the addend)
          const int m0 = __builtin_amdgcn_sbfe(w, 10 + 2 * p, 1), m1 = __builtin_amdgcn_sbfe(w, 11 + 2 * p, 1);
          const int m2 = __builtin_amdgcn_sbfe(w, 20 + 2 * p, 1), m3 = __builtin_amdgcn_sbfe(w, 21 + 2 * p, 1);
          dS_to[p] += (v2f){__int_as_float(__float_as_int(a.x) & m0), __int_as_float(__float_as_int(a.y) & m1)};
          dS_fr[p] += (v2f){__int_as_float(__float_as_int(b.x) & m2), __int_as_float(__float_as_int(b.y) & m3)};
        }
      }
    }
    if (r0 + LIN_CHUNK < nslots) {   // (rare: a slice with more than LIN_CHUNK slot rows)
#pragma unroll
      for (int i = 0; i < LIN_CHUNK; ++i) sw[i] = si[(int64_t)min(r0 + LIN_CHUNK + i, nslots - 1) * 64];
    }
  }
  // ---- tangent of the gate, the update MLP and LayerNorm
  if (LIN_PRIO) __builtin_amdgcn_s_setprio(0);
#if !LIN_REC_EARLY
  PHASE();   // the node record and v's own row are not needed during the walk: read them here instead of holding 34 VGPRs across it
  const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3], r4 = rp[4], r5 = rp[5];
  load10(tv + n * D, dx);
#endif
  const float* Wf = W + lofs + L::L_FOLD;
  const float* Wa = W + L::AL_W;
  const float* dsto = reinterpret_cast<const float*>(dS_to);
  const float* dsfr = reinterpret_cast<const float*>(dS_fr);
  PHASE();
  float dal = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) dal = fmaf(Wa[k], dx[k], dal);
#pragma unroll
  for (int k = 0; k < D; ++k) dal = fmaf(Wf[L::F_ATO + k], dsto[k], dal);
#pragma unroll
  for (int k = 0; k < D; ++k) dal = fmaf(Wf[L::F_AFR + k], dsfr[k], dal);
  const float al = r0.x;
  dal *= r0.y;
  const unsigned hm = __float_as_uint(r0.w);
  v2f dq[5], dupd[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) dq[p] = dupd[p] = splat(0.f);
  PHASE();
  mv2<D>(T + L::T_U1H, dx, dq);
  PHASE();
  mv2<D>(T + L::T_GTO, dsto, dq);
  PHASE();
  mv2<D>(T + L::T_GFR, dsfr, dq);
#pragma unroll
  for (int p = 0; p < 5; ++p)
    dq[p] = (v2f){(hm >> (2 * p)) & 1u ? dq[p].x : 0.f, (hm >> (2 * p + 1)) & 1u ? dq[p].y : 0.f};
  PHASE();
  mv2<D>(T + L::T_U2, reinterpret_cast<const float*>(dq), dupd);
  const float* du = reinterpret_cast<const float*>(dupd);
  const float u[D] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y};
  const float yh[D] = {r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w};
  float dm = 0.f, yd = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    dy[o] = dx[o] + dal * u[o] + al * du[o];
    dm += dy[o];
    yd = fmaf(yh[o], dy[o], yd);
  }
  dm *= (1.f / D);
  yd *= (1.f / D);
  const float rs = r0.z;
#pragma unroll
  for (int o = 0; o < D; ++o) dy[o] = W[L::LN_G + o] * rs * (dy[o] - dm - yh[o] * yd);
  store10(out + n * D, dy);
}

// ------------------------------------------------------------------------------------------------------------------ host
int psignn_f_tile_jvp_groups(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                             const float* v, float* out, int groups, hipStream_t st);

extern "C" int psignn_lin_create(psignn_lin_t** out, const psignn_plan_t* p) {
  ARG_CHECK(out && p, "NULL argument");
  ARG_CHECK(p->tiled, "linearised JVP: tiled plans (other plans use psignn_f_jvp)");
  ARG_CHECK(p->max_rows <= 1024, "tile + halo rows exceed the 10-bit row field of the stored slots");
  psignn_lin* s = new psignn_lin();
  s->plan = p;
  const size_t b_slot = (size_t)(p->ell_rows + 1) * 64 * 4, b_rec = (size_t)p->N * LIN_REC * 4;
  const size_t b_state = p->mixed ? (size_t)p->N * (D + 3 + 2) * 4 : 0;
  bool ok = hipMalloc((void**)&s->slot, b_slot) == hipSuccess && hipMalloc((void**)&s->rec, b_rec) == hipSuccess;
  if (ok && p->mixed)
    ok = hipMalloc((void**)&s->h, (size_t)p->N * D * 4) == hipSuccess && hipMalloc((void**)&s->prb, (size_t)p->N * 3 * 4) == hipSuccess &&
         hipMalloc((void**)&s->nrm, (size_t)p->N * 2 * 4) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    psignn_lin_destroy(s);
    psignn_set_error("psignn_lin_create: out of device memory (%zu bytes)", b_slot + b_rec + b_state);
    return PSIGNN_ENOMEM;
  }
  s->bytes = b_slot + b_rec + b_state;
  *out = s;
  return PSIGNN_OK;
}

extern "C" void psignn_lin_destroy(psignn_lin_t* s) {
  if (!s) return;
  for (void* q : {(void*)s->slot, (void*)s->rec, (void*)s->h, (void*)s->prb, (void*)s->nrm})
    if (q) (void)hipFree(q);
  delete s;
}

extern "C" size_t psignn_lin_bytes(const psignn_lin_t* s) { return s ? s->bytes : 0; }

// h, prb (and, mixed plans, the unit normals) in PLAN order; dirichlet: single-layer block; mixed: any depth (the iterated layer is
// the last one, as in psignn_f_jvp)
extern "C" int psignn_lin_build(psignn_lin_t* s, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                                void* stream) {
  ARG_CHECK(s && W && h && prb, "NULL argument");
  const psignn_plan* p = s->plan;
  ARG_CHECK(p->mixed ? nl >= 1 : nl == 1, "linearised JVP: single-layer blocks (mixed plans: the last layer)");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)p->max_rows * 20 * 4;
  if (p->mixed) {
    using L = WLayout<3>;
    const int na = (int)p->n_tiles_plain;
    HIP_TRY(hipMemcpyAsync(s->h, h, (size_t)p->N * D * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->prb, prb, (size_t)p->N * 3 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->nrm, nrm, (size_t)p->N * 2 * 4, hipMemcpyDeviceToDevice, st));
    if (na > 0) {
      const int chunk = (int)cdiv(na, 8);
      PROF_BYTES((62 * p->N + 20 * p->Ep) + (int64_t)p->N * LIN_REC * 4 + (int64_t)p->ell_rows * 64 * 4);
      LAUNCH("k_lin_build", st, (k_lin_build<3><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
          na, chunk, p->tile_order, p->h_ctx, W, L::layer(nl - 1), L::tp_layer(nl, true, nl - 1), h, prb, s->slot, s->rec)));
    }
  } else {
    using L = WLayout<2>;
    const int chunk = (int)cdiv(p->n_tiles, 8);
    // B_f's reads (h, prb, flags, slot records) + the stored linearisation
    PROF_BYTES((49 * p->N + 20 * p->Ep) + (int64_t)p->N * LIN_REC * 4 + (int64_t)p->ell_rows * 64 * 4);
    LAUNCH("k_lin_build", st, (k_lin_build<2><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
        (int)p->n_tiles, chunk, nullptr, p->h_ctx, W, L::layer(0), L::tp_layer(nl, false, 0), h, prb, s->slot, s->rec)));
  }
  HIP_TRY(hipGetLastError());
  s->built = 1;
  return PSIGNN_OK;
}

// v, out in PLAN order: out = J_f(h) v for the h of the last psignn_lin_build
extern "C" int psignn_lin_jvp(const psignn_lin_t* s, const float* W, int nl, const float* v, float* out, void* stream) {
  ARG_CHECK(s && W && v && out, "NULL argument");
  ARG_CHECK(s->built, "psignn_lin_build has not run");
  ARG_CHECK(v != out, "in-place product is not supported");
  const psignn_plan* p = s->plan;
  ARG_CHECK(p->mixed ? nl >= 1 : nl == 1, "linearised JVP: single-layer blocks (mixed plans: the last layer)");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = std::max((size_t)p->max_rows * 20 * 4, tile_lds_min());
  if (p->mixed) {
    using L = WLayout<3>;
    const int na = (int)p->n_tiles_plain;
    if (na > 0) {
      const int chunk = (int)cdiv(na, 8);
      PROF_BYTES((int64_t)p->N * (81 + LIN_REC * 4) + (int64_t)p->ell_rows * 64 * 4);
      LAUNCH("k_jvp_lin", st, (k_jvp_lin<3><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
          na, chunk, p->tile_order, p->h_ctx, W, L::layer(nl - 1), L::tp_layer(nl, true, nl - 1), s->slot, s->rec, v, out)));
    }
    int rc = psignn_f_tile_jvp_groups(p, W, nl, s->h, s->prb, s->nrm, v, out, 2, st);   // the tiles holding Neumann nodes
    if (rc) return rc;
  } else {
    using L = WLayout<2>;
    const int chunk = (int)cdiv(p->n_tiles, 8);
    // v, out (40 N each), flags (N), node records, slot dwords
    PROF_BYTES((int64_t)p->N * (81 + LIN_REC * 4) + (int64_t)p->ell_rows * 64 * 4);
    LAUNCH("k_jvp_lin", st, (k_jvp_lin<2><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
        (int)p->n_tiles, chunk, nullptr, p->h_ctx, W, L::layer(0), L::tp_layer(nl, false, 0), s->slot, s->rec, v, out)));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
