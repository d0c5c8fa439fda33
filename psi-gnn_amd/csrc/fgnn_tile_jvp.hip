// Tiled Jacobian-vector product of f_theta (both families; the layer a solver iterates: single layer, or the last layer of
// the mixed family's loop, which reads the original h -- mixed/psignn/model.py:221-245), plan order (gfx950).
//
// out = J_f(h) v: what the Newton-Krylov solver of BASELINE configs[4] needs once per inner iteration (the reference only
// imports scipy's newton_krylov, utilities/solver.py:6; its finite-difference JVPs do not converge in fp32).
// Same mathematics as the global-gather JVP of fgnn.hip (k_project / k_node with JVP = true) on the tile structures:
//   stage 1  projects BOTH the state row and the tangent row of every tile + halo node with the neighbour-side weights
//            -> 160-byte LDS row [Pj_to | Pj_from | dPj_to | dPj_from];
//   stage 2  per node and direction, one walk over the pair-merged slots:  z = Pi + Pj + A a,  S += relu(z),
//            dS += 1[z > 0] (dPi + dPj);  then the tangent of the folded gate / update MLP and of LayerNorm.
// Dirichlet rows of f are constants: their tangent is 0.  Mixed family: a Neumann row is update_neumann([h | Phi_neumann(h)
// | prb | normal]) (it REPLACES the row, mixed/psignn/model.py:236,241) -- its tangent comes from a third slot walk over the
// Phi_neumann columns, which tiles WITH Neumann nodes stage as 20 more floats per LDS row (tiles without run the 160-byte
// form in a launch of their own, like k_f_tile).
#include "tile_helpers.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#ifndef JVP_STAGE1_DEFAULT_MFMA
#define JVP_STAGE1_DEFAULT_MFMA 0   // A/B at 1M nodes: 114 us (mfma) vs 100 us (valu)
#endif

__device__ __forceinline__ void lds_row10(const float* __restrict__ row, v2f* r) {  // 16-byte aligned
  float4 v0 = reinterpret_cast<const float4*>(row)[0], v1 = reinterpret_cast<const float4*>(row)[1];
  float2 v2 = reinterpret_cast<const float2*>(row)[4];
  r[0] = (v2f){v0.x, v0.y}; r[1] = (v2f){v0.z, v0.w}; r[2] = (v2f){v1.x, v1.y}; r[3] = (v2f){v1.z, v1.w};
  r[4] = (v2f){v2.x, v2.y};
}
__device__ __forceinline__ void lds_row10u(const float* __restrict__ row, v2f* r) {  // 8 mod 16
  float2 v0 = reinterpret_cast<const float2*>(row)[0];
  float4 v1 = reinterpret_cast<const float4*>(row + 2)[0], v2 = reinterpret_cast<const float4*>(row + 2)[1];
  r[0] = (v2f){v0.x, v0.y}; r[1] = (v2f){v1.x, v1.y}; r[2] = (v2f){v1.z, v1.w}; r[3] = (v2f){v2.x, v2.y};
  r[4] = (v2f){v2.z, v2.w};
}

// S[o] += relu(z), dS[o] += 1[z > 0] (dPi[o] + row[DCOL + o]) over the slots carrying MASK; z = Pi + row[COL..] + AT . a
template <int RS, int COL, int DCOL, unsigned MASK>
__device__ __forceinline__ float edge_pass_jvp(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                               const float* __restrict__ AT, const v2f* Pi, const v2f* dPi, v2f* S, v2f* dS) {
  float deg = 0.f;
  v2f wa[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) wa[i] = reinterpret_cast<const v2f*>(AT)[i];
  if (nslots <= 0) return deg;
  uint4 c0 = slots[0];
  uint4 c1 = slots[(int64_t)min(1, nslots - 1) * 64];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 2, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY && (w & MASK)) {
      const v2f a0 = splat(__uint_as_float(c0.y)), a1 = splat(__uint_as_float(c0.z)), a2 = splat(__uint_as_float(c0.w));
      const float* row = lds + (int)(w & 0xFFFFu) * RS;
      v2f pj[5], dpj[5], z[5];
      if (COL % 4 == 0) lds_row10(row + COL, pj); else lds_row10u(row + COL, pj);
      if (DCOL % 4 == 0) lds_row10(row + DCOL, dpj); else lds_row10u(row + DCOL, dpj);
      deg += 1.f;
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = Pi[p] + pj[p];
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[p], a0, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[5 + p], a1, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[10 + p], a2, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        S[p] += __builtin_elementwise_max(z[p], splat(0.f));
        const v2f t = dPi[p] + dpj[p];
        dS[p] += (v2f){z[p].x > 0.f ? t.x : 0.f, z[p].y > 0.f ? t.y : 0.f};
      }
    }
    c0 = c1;
    c1 = nx;
  }
  return deg;
}

#ifndef JVP_WAVES
#define JVP_WAVES 0   // > 0: hold the register allocator to this many waves per SIMD (A/B in DESIGN.md)
#endif
#if JVP_WAVES > 0
#define JVP_OCC __attribute__((amdgpu_waves_per_eu(JVP_WAVES, JVP_WAVES)))
#else
#define JVP_OCC
#endif
#ifndef JVP_PRIO
#define JVP_PRIO 1   // 1: s_setprio(3) through stage 1 and the slot walk, 0 from the node update on (86.4 -> 83.6 us; k_jvp_lin: 49.2 -> 50.7, left off; profiles/r3_ab_prio_jvp.txt)
#endif
#ifndef JVP_XRELOAD
#define JVP_XRELOAD 1
#endif
__device__ __forceinline__ v2f pk_mul_clamp(v2f a, v2f b) {   // clamp(a * b, 0, 1)
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Both directions of the neighbour sums and of their tangents in one walk over the pair-merged slots (the f kernel's
// edge_pass_both_clamp plus the tangent): the neighbour's LDS rows [Pj_to | Pj_fr] and [dPj_to | dPj_fr] (doff floats apart) are read once.
// relu(z) = 2^40 clamp(2^-40 z) as there; the mask 1[z > 0] = clamp(2^126 * clamp(2^-40 z)) is a packed multiply instead of two
// compares and two selects per pair (exact for |z| >= 2^-86; below that the f kernel's relu has already flushed the edge to 0
// while this factor is a fraction -- a pre-activation that small does not occur on finite inputs of O(1) weights).
template <int RS>
__device__ __forceinline__ void edge_pass_jvp_both(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                                   const float* __restrict__ AT_to, const float* __restrict__ AT_fr,
                                                   const v2f* Pi_to, const v2f* dPi_to, const v2f* Pi_fr, const v2f* dPi_fr,
                                                   v2f* S_to, v2f* dS_to, v2f* S_fr, v2f* dS_fr, float& deg_in, float& deg_out,
                                                   const int doff) {
  v2f wt[15], wf[15], pt[5], pf[5];
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    wt[i] = reinterpret_cast<const v2f*>(AT_to)[i];
    wf[i] = reinterpret_cast<const v2f*>(AT_fr)[i];
  }
  const v2f sc = splat(RELU_SCALE), big = splat(8.507059173023462e37f);   // 2^-40, 2^126
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    pt[p] = Pi_to[p] * sc;
    pf[p] = Pi_fr[p] * sc;
  }
  deg_in = deg_out = 0.f;
  if (nslots <= 0) return;
  uint4 c0 = slots[0];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 1, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY) {
      const v2f a01 = (v2f){__uint_as_float(c0.y), __uint_as_float(c0.z)} * sc;
      const v2f a2 = (v2f){__uint_as_float(c0.w) * RELU_SCALE, 0.f};
      const float4* row = reinterpret_cast<const float4*>(lds + (int)(w & 0xFFFFu) * RS);
      const float4* rowd = reinterpret_cast<const float4*>(lds + (int)(w & 0xFFFFu) * RS + doff);   // tangent row (see k_jvp_tile)
      if (w & SLOT_IN) {
        const float4 v0 = row[0], v1 = row[1], v2 = row[2], v5 = rowd[0], v6 = rowd[1], v7 = rowd[2];
        v2f z[5] = {(v2f){v0.x, v0.y}, (v2f){v0.z, v0.w}, (v2f){v1.x, v1.y}, (v2f){v1.z, v1.w}, (v2f){v2.x, v2.y}};
        const v2f d[5] = {(v2f){v5.x, v5.y}, (v2f){v5.z, v5.w}, (v2f){v6.x, v6.y}, (v2f){v6.z, v6.w}, (v2f){v7.x, v7.y}};
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
        for (int p = 0; p < 5; ++p) {
          S_to[p] += z[p];
          dS_to[p] = __builtin_elementwise_fma(dPi_to[p] + d[p], pk_mul_clamp(z[p], big), dS_to[p]);
        }
      }
      if (w & SLOT_OUT) {
        const float4 v2 = row[2], v3 = row[3], v4 = row[4], v7 = rowd[2], v8 = rowd[3], v9 = rowd[4];
        v2f z[5] = {(v2f){v2.z, v2.w}, (v2f){v3.x, v3.y}, (v2f){v3.z, v3.w}, (v2f){v4.x, v4.y}, (v2f){v4.z, v4.w}};
        const v2f d[5] = {(v2f){v7.z, v7.w}, (v2f){v8.x, v8.y}, (v2f){v8.z, v8.w}, (v2f){v9.x, v9.y}, (v2f){v9.z, v9.w}};
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
          dS_fr[p] = __builtin_elementwise_fma(dPi_fr[p] + d[p], pk_mul_clamp(z[p], big), dS_fr[p]);
        }
      }
    }
    c0 = nx;
  }
  const v2f us = splat(RELU_UNSCALE);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    S_to[p] *= us;
    S_fr[p] *= us;
  }
}

template <int P, bool MIXED, bool MFMA1>
__global__ __launch_bounds__(TILE_THREADS) JVP_OCC void k_jvp_tile(int n_tiles, int chunk, const int32_t* __restrict__ tile_list,
                                                           const int32_t* __restrict__ tile_ptr,
                                                           const int32_t* __restrict__ tile_slice,
                                                           const int32_t* __restrict__ halo, const int32_t* __restrict__ halo_cnt,
                                                           const int32_t* __restrict__ slice_off,
                                                           const uint8_t* __restrict__ slice_deg, const uint4* __restrict__ ell,
                                                           const uint8_t* __restrict__ flags, const float* __restrict__ W, int lofs,
                                                           int tofs, int tnofs, const float* __restrict__ h,
                                                           const float* __restrict__ prb, const float* __restrict__ nrm,
                                                           const float* __restrict__ tv, float* __restrict__ out) {
  using L = WLayout<P>;
  // LDS rows.  Tiles with Neumann nodes: one 240-byte row [Pj_to | Pj_from | dPj_to | dPj_from | Pj_neu | dPj_neu] per staged node.
  // All other tiles: TWO arrays of 80-byte rows, values [Pj_to | Pj_from] and, behind them, tangents [dPj_to | dPj_from].  A
  // ds_read_b128 is served in groups of 16 lanes over 16 slots of 16 bytes: with 160-byte rows a row starts on an EVEN slot
  // (10 r mod 16), neighbouring lanes' (mostly consecutive) rows share slots and 73 % of the LDS cycles of the launch were bank
  // conflicts (SQ_LDS_BANK_CONFLICT 21.1 M of SQ_LDS_IDX_ACTIVE 29.0 M cycles, LDS pipe 47 % busy at 1M nodes); 80-byte rows start on
  // 5 r mod 16 -- every slot, sixteen consecutive rows conflict-free -- as in k_f_tile.
  constexpr int RS = MIXED ? 60 : 20;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int slot_ = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot_ >= n_tiles) return;
  const int tile = tile_list ? tile_list[slot_] : slot_;
  const float* TN = W + tnofs;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  const float* T = W + tofs;
  const int doff = MIXED ? 2 * D : (n_t + n_h) * RS;   // floats from a node's value row to its tangent row
  // ---- stage 1
  if (JVP_PRIO) __builtin_amdgcn_s_setprio(3);
  float x[D], dx[D];
  if constexpr (MFMA1) {
    // Both dense products of stage 1 -- state rows and tangent rows times the neighbour-side weights -- on the matrix
    // cores, exactly as in k_f_tile's MFMA stage 1 (weights = A operand, 16 node rows = B, a lane receives four
    // consecutive outputs of one row = one 16-byte LDS store; the f32 MFMA sums k in order: same bits as the VALU form).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, g = lane >> 4, c = lane & 15, wave = tid >> 6;
    float wa[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int sk = 0; sk < 3; ++sk) {
        const int o = 16 * mt + c, k = 4 * sk + g;
        float v = 0.f;
        if (k < D && o < 2 * D) v = (o < D ? T + L::T_W1J_TO : T + L::T_W1J_FR)[k * D + (o % D)];
        wa[mt][sk] = v;
      }
    const int rows = n_t + n_h;
    for (int nt = wave; nt * 16 < rows; nt += TILE_THREADS / 64) {
      const int row = nt * 16 + c;
      const bool ok = row < rows;
      const int64_t node = !ok ? (int64_t)t0 : (row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t]);
#pragma unroll
      for (int src = 0; src < 2; ++src) {   // 0: state rows -> columns 0..19, 1: tangent rows -> columns 20..39
        const float* base = src ? tv : h;
        float xb[3];
#pragma unroll
        for (int sk = 0; sk < 3; ++sk) {
          const int k = 4 * sk + g;
          xb[sk] = (ok && k < D) ? base[node * D + k] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int sk = 0; sk < 3; ++sk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][sk], xb[sk], acc, 0, 0, 0);
          const int o0 = 16 * mt + 4 * g;
          if (ok && o0 < 2 * D)
            *reinterpret_cast<float4*>(lds + row * RS + (src ? doff : 0) + o0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
      }
    }
    if (tid < n_t) {
      load10(h + (int64_t)(t0 + tid) * D, x);
      load10(tv + (int64_t)(t0 + tid) * D, dx);
    }
  } else {
  for (int row = tid; row < n_t + n_h; row += TILE_THREADS) {
    const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
    float xr[D], vr[D];
    load10(h + node * D, xr);
    load10(tv + node * D, vr);
    if (row == tid) {
#pragma unroll
      for (int o = 0; o < D; ++o) {
        x[o] = xr[o];
        dx[o] = vr[o];
      }
    }
    v2f ta[5], tb[5], da[5], db[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ta[p] = tb[p] = da[p] = db[p] = splat(0.f);
    PHASE();
    mv2<D>(T + L::T_W1J_TO, xr, ta);
    mv2<D>(T + L::T_W1J_TO, vr, da);
    PHASE();
    mv2<D>(T + L::T_W1J_FR, xr, tb);
    mv2<D>(T + L::T_W1J_FR, vr, db);
    float4* q = reinterpret_cast<float4*>(lds + row * RS);
    q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
    q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
    q[2] = make_float4(ta[4].x, ta[4].y, tb[0].x, tb[0].y);
    q[3] = make_float4(tb[1].x, tb[1].y, tb[2].x, tb[2].y);
    q[4] = make_float4(tb[3].x, tb[3].y, tb[4].x, tb[4].y);
    float4* qd = reinterpret_cast<float4*>(lds + row * RS + doff);
    qd[0] = make_float4(da[0].x, da[0].y, da[1].x, da[1].y);
    qd[1] = make_float4(da[2].x, da[2].y, da[3].x, da[3].y);
    qd[2] = make_float4(da[4].x, da[4].y, db[0].x, db[0].y);
    qd[3] = make_float4(db[1].x, db[1].y, db[2].x, db[2].y);
    qd[4] = make_float4(db[3].x, db[3].y, db[4].x, db[4].y);
    if (MIXED) {   // Phi_neumann columns of the state and the tangent row
#pragma unroll
      for (int p = 0; p < 5; ++p) ta[p] = da[p] = splat(0.f);
      PHASE();
      mv2<D>(TN + L::N_W1J, xr, ta);
      mv2<D>(TN + L::N_W1J, vr, da);
      q[10] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
      q[11] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
      q[12] = make_float4(ta[4].x, ta[4].y, da[0].x, da[0].y);
      q[13] = make_float4(da[1].x, da[1].y, da[2].x, da[2].y);
      q[14] = make_float4(da[3].x, da[3].y, da[4].x, da[4].y);
    }
  }
  }
  __syncthreads();
  if (tid >= n_t) return;
  const int64_t n = (int64_t)t0 + tid;
  float dy[D];
  const uint8_t fl = flags[n];
  if (fl & FLAG_DIRICHLET) {
#pragma unroll
    for (int o = 0; o < D; ++o) dy[o] = 0.f;
    store10(out + n * D, dy);
    return;
  }
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (tid >> 6);
  const uint4* slots = ell + (int64_t)slice_off[slice] * 64 + lane;
  const int nslots = slice_deg[slice];
  float y[D], mu = 0.f;
  if (MIXED && (fl & FLAG_NEUMANN)) {
    // ---- Neumann row: y = N2 relu(q) + nb2, q = nb1 + deg gN + N1h x + Gn S_n + N1p [prb | normal]  (second Phi_neumann
    // layer folded into Gn / gN like the interior path's), tangent through the same chain
    v2f Pi[5], dPi[5], S_n[5], dS_n[5];
    ld5(TN + L::N_B1, Pi);
#pragma unroll
    for (int p = 0; p < 5; ++p) S_n[p] = dS_n[p] = dPi[p] = splat(0.f);
    PHASE();
    mv2<D>(TN + L::N_W1I, x, Pi);
    mv2<D>(TN + L::N_W1I, dx, dPi);
    const float deg_out = edge_pass_jvp<RS, 4 * D, 5 * D, SLOT_OUT>(slots, nslots, lds, TN + L::N_A, Pi, dPi, S_n, dS_n);
    v2f q[5], dq[5], gN[5], y2[5], dy2[5];
    ld5(TN + L::N_NB1, q);
    ld5(TN + L::N_gN, gN);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      q[p] = __builtin_elementwise_fma(splat(deg_out), gN[p], q[p]);
      dq[p] = dy2[p] = splat(0.f);
    }
    PHASE();
    mv2<D>(TN + L::N_N1H, x, q);
    mv2<D>(TN + L::N_N1H, dx, dq);
    PHASE();
    mv2<D>(TN + L::N_GN, reinterpret_cast<const float*>(S_n), q);
    mv2<D>(TN + L::N_GN, reinterpret_cast<const float*>(dS_n), dq);
    float pq[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    pq[P] = nrm[n * 2];
    pq[P + 1] = nrm[n * 2 + 1];
    PHASE();
    mv2<P + 2>(TN + L::N_N1P, pq, q);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      dq[p] = (v2f){q[p].x > 0.f ? dq[p].x : 0.f, q[p].y > 0.f ? dq[p].y : 0.f};
      q[p] = __builtin_elementwise_max(q[p], splat(0.f));
    }
    ld5(TN + L::N_NB2, y2);
    PHASE();
    mv2<D>(TN + L::N_N2, reinterpret_cast<const float*>(q), y2);
    mv2<D>(TN + L::N_N2, reinterpret_cast<const float*>(dq), dy2);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      y[2 * p] = y2[p].x; y[2 * p + 1] = y2[p].y;
      dy[2 * p] = dy2[p].x; dy[2 * p + 1] = dy2[p].y;
    }
#pragma unroll
    for (int o = 0; o < D; ++o) mu += y[o];
  } else {
  // ---- stage 2: neighbour sums and their tangents
  v2f S_to[5], S_fr[5], dS_to[5], dS_fr[5];
  float deg_in, deg_out;
#pragma unroll
  for (int p = 0; p < 5; ++p) S_to[p] = S_fr[p] = dS_to[p] = dS_fr[p] = splat(0.f);
  {
    v2f Pi[5], dPi[5], Pi2[5], dPi2[5];
    ld5(T + L::T_B1_TO, Pi);
    ld5(T + L::T_B1_FR, Pi2);
#pragma unroll
    for (int p = 0; p < 5; ++p) dPi[p] = dPi2[p] = splat(0.f);
    PHASE();
    mv2<D>(T + L::T_W1I_TO, x, Pi);
    mv2<D>(T + L::T_W1I_TO, dx, dPi);
    PHASE();
    mv2<D>(T + L::T_W1I_FR, x, Pi2);
    mv2<D>(T + L::T_W1I_FR, dx, dPi2);
    edge_pass_jvp_both<RS>(slots, nslots, lds, T + L::T_A_TO, T + L::T_A_FR, Pi, dPi, Pi2, dPi2, S_to, dS_to, S_fr, dS_fr, deg_in,
                           deg_out, doff);
  }
#if JVP_XRELOAD
  PHASE();   // x / dx are not needed during the walk: re-read them (L2 hits) instead of holding 20 VGPRs across it
  load10(h + n * D, x);
  load10(tv + n * D, dx);
#endif
  // ---- gate and update MLP (second Phi layer folded), values and tangents
  if (JVP_PRIO) __builtin_amdgcn_s_setprio(0);
  const float* Wf = W + lofs + L::L_FOLD;
  const float* Wa = W + L::AL_W;
  const float* sto = reinterpret_cast<const float*>(S_to);
  const float* sfr = reinterpret_cast<const float*>(S_fr);
  const float* dsto = reinterpret_cast<const float*>(dS_to);
  const float* dsfr = reinterpret_cast<const float*>(dS_fr);
  float pq[P];
#pragma unroll
  for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
  PHASE();
  float al = fmaf(deg_in, Wf[L::F_ABTO], fmaf(deg_out, Wf[L::F_ABFR], W[L::AL_B])), dal = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    al = fmaf(Wa[k], x[k], al);
    dal = fmaf(Wa[k], dx[k], dal);
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    al = fmaf(Wf[L::F_ATO + k], sto[k], al);
    dal = fmaf(Wf[L::F_ATO + k], dsto[k], dal);
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    al = fmaf(Wf[L::F_AFR + k], sfr[k], al);
    dal = fmaf(Wf[L::F_AFR + k], dsfr[k], dal);
  }
#pragma unroll
  for (int k = 0; k < P; ++k) al = fmaf(Wa[3 * D + k], pq[k], al);
  al = 1.f / (1.f + expf(-al));
  dal *= al * (1.f - al);
  v2f q[5], dq[5], g1[5], g2[5], upd[5], dupd[5];
  ld5(T + L::T_HB, q);
  ld5(T + L::T_gTO, g1);
  ld5(T + L::T_gFR, g2);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    q[p] = __builtin_elementwise_fma(splat(deg_in), g1[p], __builtin_elementwise_fma(splat(deg_out), g2[p], q[p]));
    dq[p] = dupd[p] = splat(0.f);
  }
  PHASE();
  mv2<D>(T + L::T_U1H, x, q);
  mv2<D>(T + L::T_U1H, dx, dq);
  PHASE();
  mv2<D>(T + L::T_GTO, sto, q);
  mv2<D>(T + L::T_GTO, dsto, dq);
  PHASE();
  mv2<D>(T + L::T_GFR, sfr, q);
  mv2<D>(T + L::T_GFR, dsfr, dq);
  mv2<P>(T + L::T_U1P, pq, q);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    dq[p] = (v2f){q[p].x > 0.f ? dq[p].x : 0.f, q[p].y > 0.f ? dq[p].y : 0.f};
    q[p] = __builtin_elementwise_max(q[p], splat(0.f));
  }
  ld5(T + L::T_C2, upd);
  PHASE();
  mv2<D>(T + L::T_U2, reinterpret_cast<const float*>(q), upd);
  mv2<D>(T + L::T_U2, reinterpret_cast<const float*>(dq), dupd);
  const float* u = reinterpret_cast<const float*>(upd);
  const float* du = reinterpret_cast<const float*>(dupd);
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = fmaf(al, u[o], x[o]);
    dy[o] = dx[o] + dal * u[o] + al * du[o];
    mu += y[o];
  }
  }
  // ---- LayerNorm tangent (eps 1e-5, biased variance)
  mu *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    const float c = y[o] - mu;
    var = fmaf(c, c, var);
  }
  var *= (1.f / D);
  const float rs = 1.f / sqrtf(var + 1e-5f);
  float dm = 0.f, yd = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = (y[o] - mu) * rs;
    dm += dy[o];
    yd = fmaf(y[o], dy[o], yd);
  }
  dm *= (1.f / D);
  yd *= (1.f / D);
#pragma unroll
  for (int o = 0; o < D; ++o) dy[o] = W[L::LN_G + o] * rs * (dy[o] - dm - y[o] * yd);
  store10(out + n * D, dy);
}

// h, prb, nrm (mixed plans), v, out in PLAN order.
// groups (mixed plans): bit 0 = tiles without Neumann nodes, bit 1 = tiles holding Neumann nodes (fgnn_tile_lin.hip applies the
// stored linearisation on the first group and this kernel on the second)
int psignn_f_tile_jvp_groups(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                             const float* v, float* out, int groups, hipStream_t st);
int psignn_f_tile_jvp(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                      const float* v, float* out, hipStream_t st) {
  return psignn_f_tile_jvp_groups(p, W, nl, h, prb, nrm, v, out, 3, st);
}
int psignn_f_tile_jvp_groups(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                             const float* v, float* out, int groups, hipStream_t st) {
  ARG_CHECK(p && p->tiled && (nl == 1 || p->mixed), "tiled JVP: tiled plans, single-layer blocks (mixed: any depth, last layer)");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
#define JVP_TILE_ARGS p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell, p->flags_p
  if (p->mixed) {
    using L = WLayout<3>;
    const int lofs = L::layer(nl - 1), tofs = L::tp_layer(nl, true, nl - 1), tnofs = L::tp_neu(nl);
    const int na = (int)p->n_tiles_plain, nb = (int)(p->n_tiles - p->n_tiles_plain);
    ARG_CHECK((size_t)p->max_rows * 60 * 4 <= 160 * 1024, "tile + halo rows exceed the LDS budget of the tiled JVP");
    if (na > 0 && (groups & 1)) {   // tiles without Neumann nodes: 160-byte LDS rows, no Neumann branch
      const int chunk = (int)cdiv(na, 8);
      LAUNCH("k_jvp_tile", st, (k_jvp_tile<3, false, false><<<(unsigned)(chunk * 8), TILE_THREADS, (size_t)p->max_rows * 40 * 4, st>>>(
          na, chunk, p->tile_order, JVP_TILE_ARGS, W, lofs, tofs, tnofs, h, prb, nrm, v, out)));
    }
    if (nb > 0 && (groups & 2)) {
      const int chunk = (int)cdiv(nb, 8);
      LAUNCH("k_jvp_tile", st, (k_jvp_tile<3, true, false><<<(unsigned)(chunk * 8), TILE_THREADS, (size_t)p->max_rows * 60 * 4, st>>>(
          nb, chunk, p->tile_order + na, JVP_TILE_ARGS, W, lofs, tofs, tnofs, h, prb, nrm, v, out)));
    }
    HIP_TRY(hipGetLastError());
    return PSIGNN_OK;
  }
  using L = WLayout<2>;
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const size_t lds = std::max((size_t)p->max_rows * 40 * 4, tile_lds_min());
  ARG_CHECK(lds <= 160 * 1024, "tile + halo rows exceed the LDS budget of the tiled JVP");
  // stage-1 form: PSIGNN_JVP_STAGE1 = mfma | valu (default: see the A/B in DESIGN.md)
  KNOB_INT(use_mfma, [] {
    const char* e = getenv("PSIGNN_JVP_STAGE1");
    return e ? (int)(strcmp(e, "mfma") == 0) : (int)JVP_STAGE1_DEFAULT_MFMA;
  }());
  if (use_mfma)
    LAUNCH("k_jvp_tile", st, (k_jvp_tile<2, false, true><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
        (int)p->n_tiles, chunk, nullptr, JVP_TILE_ARGS, W, L::layer(0), L::tp_layer(nl, false, 0), 0, h, prb, nrm, v, out)));
  else
    LAUNCH("k_jvp_tile", st, (k_jvp_tile<2, false, false><<<(unsigned)(chunk * 8), TILE_THREADS, lds, st>>>(
        (int)p->n_tiles, chunk, nullptr, JVP_TILE_ARGS, W, L::layer(0), L::tp_layer(nl, false, 0), 0, h, prb, nrm, v, out)));
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
