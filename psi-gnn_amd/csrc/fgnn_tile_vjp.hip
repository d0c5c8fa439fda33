// Tiled vector-Jacobian product of f_theta (single-layer dirichlet blocks; mixed blocks of any depth, whose iterated layer is
// the last one and reads the original h -- mixed/psignn/model.py:221-245), plan order (gfx950).
//
// Same mathematics as fgnn_vjp.hip (two gather passes, no atomics; reference: autograd.grad(new_H, H, v),
// dirichlet/psignn/model.py:210-223,416-452), on the tile structures of the mesh plan:
//   pass A  (k_vjp_tile_a)  = the forward tile kernel (stage 1: neighbour projections -> LDS; stage 2: pair-merged
//           slots) that additionally counts, per output, the edges whose pre-activation is positive.  The masked
//           cotangent sum over a node's OWN edges is then just  dS * count  (dS_to[n] is per node, not per edge),
//           so no second sweep over the slots is needed.  Backward through LayerNorm, the folded gate / update
//           MLP; writes the node-local result and B[n] = { Pt[n], dS_to[n], Pf[n], dS_fr[n] } (40 floats).
//   pass B  (k_vjp_tile_b)  stages B[tile + halo] in LDS (one 80-byte half row at a time) and, for node u, walks the same slots
//           from the neighbour's side:  OUT slot (edge u -> n): acc_t += dS_to[n] * 1[Pt[n] + Pjt[u] + At a > 0]
//                                       IN  slot (edge n -> u): acc_f += dS_fr[n] * 1[Pf[n] + Pjf[u] + Af m(a) > 0]
//           out[u] += W1j_to^T acc_t + W1j_fr^T acc_f.
// Mixed family: a Neumann row n is update_neumann([h | Phi_neumann(h) | prb | normal]) with Phi_neumann summing over the
// node's OUT edges (mixed/psignn/model.py:225,233-236,241).  Pass A runs the tiles that hold Neumann nodes with a third
// projection column in LDS (128-byte rows; the other tiles in a launch of their own, like k_f_tile) and writes
// B[n] = { 0, 0, Pn[n], dS_n[n] } for such a row -- the Phi_from half, its dS_to is zero.  Pass B keeps the rows' Neumann
// flag beside the B rows in LDS: an IN slot (edge n -> u) whose sender n is a Neumann row uses Phi_neumann's weights and
// accumulates into a third sum, out[u] += W1j_neu^T acc_n.  A tile without Neumann rows among tile + halo skips all of it.
// All tensors in plan order.  The parameter-gradient records (PG) exist for the dirichlet family only.
#include "fgnn_common.h"

#define SLOT_IN 0x10000u
#define SLOT_OUT 0x20000u
#define VT 256
#define PHASE() asm volatile("" ::: "memory")   // bounds scalar-load hoisting (see tile_helpers.h)
#define PGREC 320   // floats per node of a parameter-gradient record: 20 groups of 16 (layout in k_vjp_tile_a)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }

// acc[p] += WT[k][2p..2p+1] * x[k]   (transposed weight section: [in k][out o], o fastest)
template <int K>
__device__ __forceinline__ void mvf(const float* __restrict__ WT, const float* x, v2f* acc) {
  const v2f* w = reinterpret_cast<const v2f*>(WT);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const v2f xs = splat2(x[k]);
#pragma unroll
    for (int p = 0; p < 5; ++p) acc[p] = __builtin_elementwise_fma(w[k * 5 + p], xs, acc[p]);
  }
}
// acc[p] += W[o*ld + off + 2p..2p+1] * g[o]   (ORIGINAL (out,in) layout = the transposed product, k pairs adjacent)
template <int NO>
__device__ __forceinline__ void mvb(const float* __restrict__ W, int ld, int off, const float* g, v2f* acc) {
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const v2f gs = splat2(g[o]);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const v2f w = (v2f){W[o * ld + off + 2 * p], W[o * ld + off + 2 * p + 1]};
      acc[p] = __builtin_elementwise_fma(w, gs, acc[p]);
    }
  }
}
__device__ __forceinline__ void ldu5(const float* __restrict__ p, v2f* r) {
  const v2f* q = reinterpret_cast<const v2f*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) r[i] = q[i];
}
__device__ __forceinline__ void row10(const float* __restrict__ row, v2f* pj) {  // 10 floats at a 16-byte aligned LDS address
  float4 v0 = reinterpret_cast<const float4*>(row)[0], v1 = reinterpret_cast<const float4*>(row)[1];
  float2 v2 = reinterpret_cast<const float2*>(row)[4];
  pj[0] = (v2f){v0.x, v0.y}; pj[1] = (v2f){v0.z, v0.w}; pj[2] = (v2f){v1.x, v1.y}; pj[3] = (v2f){v1.z, v1.w};
  pj[4] = (v2f){v2.x, v2.y};
}
__device__ __forceinline__ void row10u(const float* __restrict__ row, v2f* pj) {  // 8-byte aligned start
  float2 v0 = reinterpret_cast<const float2*>(row)[0];
  float4 v1 = reinterpret_cast<const float4*>(row + 2)[0], v2 = reinterpret_cast<const float4*>(row + 2)[1];
  pj[0] = (v2f){v0.x, v0.y}; pj[1] = (v2f){v1.x, v1.y}; pj[2] = (v2f){v1.z, v1.w}; pj[3] = (v2f){v2.x, v2.y};
  pj[4] = (v2f){v2.z, v2.w};
}

// ---------------------------------------------------------------------------------------------- pass A
// forward sum S[o] += relu(z) and cnt[o] += 1[z > 0] over the slots carrying MASK; z = Pi + row[COL..] + AT . a
// PG (parameter-gradient mode): also m[c*5 + p] += 1[z > 0] * a_c, the attr moments the W1 attr-block gradient needs.
template <int RS, int COL, unsigned MASK, bool PG>
__device__ __forceinline__ float pass_fwd(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                          const float* __restrict__ AT, const v2f* Pi, v2f* S, v2f* cnt, v2f* m) {
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
      const v2f a0 = splat2(__uint_as_float(c0.y)), a1 = splat2(__uint_as_float(c0.z)), a2 = splat2(__uint_as_float(c0.w));
      const float* row = lds + (int)(w & 0xFFFFu) * RS + COL;
      v2f pj[5], z[5];
      if (COL % 4 == 0) row10(row, pj); else row10u(row, pj);
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
        S[p] += __builtin_elementwise_max(z[p], splat2(0.f));
        const v2f mk = (v2f){z[p].x > 0.f ? 1.f : 0.f, z[p].y > 0.f ? 1.f : 0.f};
        cnt[p] += mk;
        if (PG) {
          m[p] = __builtin_elementwise_fma(mk, a0, m[p]);
          m[5 + p] = __builtin_elementwise_fma(mk, a1, m[5 + p]);
          m[10 + p] = __builtin_elementwise_fma(mk, a2, m[10 + p]);
        }
      }
    }
    c0 = c1;
    c1 = nx;
  }
  return deg;
}

// one 16-float group of a parameter-gradient record: v[0..n) then `tail`, then zeros
__device__ __forceinline__ void rec_group(float* __restrict__ g, const float* v, int n, float t0 = 0.f, float t1 = 0.f,
                                          float t2 = 0.f, float t3 = 0.f, float t4 = 0.f) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    r[i] = i < n ? v[i] : (i == n ? t0 : (i == n + 1 ? t1 : (i == n + 2 ? t2 : (i == n + 3 ? t3 : (i == n + 4 ? t4 : 0.f)))));
  float4* q = reinterpret_cast<float4*>(g);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
}

// LayerNorm forward + backward at one row: y -> normalised y (in place), dy = d loss / d y for the cotangent w of LN(y)
template <class L>
__device__ __forceinline__ void ln_fwd_bwd(const float* __restrict__ W, float mu, float* y, const float* w, float* dy) {
  mu *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    float c = y[o] - mu;
    var = fmaf(c, c, var);
  }
  var *= (1.f / D);
  const float rs = 1.f / sqrtf(var + 1e-5f);
  float dyh[D], m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = (y[o] - mu) * rs;
    dyh[o] = w[o] * W[L::LN_G + o];
    m1 += dyh[o];
    m2 = fmaf(dyh[o], y[o], m2);
  }
  m1 *= (1.f / D);
  m2 *= (1.f / D);
#pragma unroll
  for (int o = 0; o < D; ++o) dy[o] = rs * (dyh[o] - m1 - y[o] * m2);
}

#ifndef VJPA_WAVES
#define VJPA_WAVES 0   // > 0: hold the register allocator of pass A to this many waves per SIMD
#endif
#if VJPA_WAVES > 0
#define VJPA_OCC __attribute__((amdgpu_waves_per_eu(VJPA_WAVES, VJPA_WAVES)))
#else
#define VJPA_OCC
#endif
template <int P, bool MIXED, bool PG>
__global__ __launch_bounds__(VT) VJPA_OCC void k_vjp_tile_a(int n_tiles, int chunk, const int32_t* __restrict__ tile_list,
                                                   const int32_t* __restrict__ tile_ptr,
                                                   const int32_t* __restrict__ tile_slice, const int32_t* __restrict__ halo,
                                                   const int32_t* __restrict__ halo_cnt, const int32_t* __restrict__ slice_off,
                                                   const uint8_t* __restrict__ slice_deg, const uint4* __restrict__ ell,
                                                   const uint8_t* __restrict__ flags, const float* __restrict__ W, int nl,
                                                   int lofs, int tofs, int tnofs, const float* __restrict__ h,
                                                   const float* __restrict__ prb, const float* __restrict__ nrm,
                                                   const float* __restrict__ wv, float* __restrict__ B,
                                                   float* __restrict__ out, float* __restrict__ rec) {
  using L = WLayout<P>;
  // record stride: 20 groups (dirichlet plans) / 30 (mixed plans, P = 3: the Neumann groups 20..29 of fgnn_vjp.hip's PgRec follow)
  constexpr int REC = P == 3 ? 480 : PGREC;
  constexpr int RS = MIXED ? 32 : 20;   // [Pj_to 10 | Pj_from 10 (| Pj_neu 10 | pad 2)]
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int slot_ = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot_ >= n_tiles) return;
  const int tile = tile_list ? tile_list[slot_] : slot_;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  const float* T = W + tofs;
  // ---- stage 1: neighbour-side projections of tile + halo rows -> LDS
  float x[D];
  for (int row = tid; row < n_t + n_h; row += VT) {
    const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
    float xr[D];
    load10(h + node * D, xr);
    if (row == tid) {
#pragma unroll
      for (int o = 0; o < D; ++o) x[o] = xr[o];
    }
    v2f ta[5], tb[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ta[p] = tb[p] = splat2(0.f);
    PHASE();
    mvf<D>(T + L::T_W1J_TO, xr, ta);
    PHASE();
    mvf<D>(T + L::T_W1J_FR, xr, tb);
    float4* q = reinterpret_cast<float4*>(lds + row * RS);
    q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
    q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
    q[2] = make_float4(ta[4].x, ta[4].y, tb[0].x, tb[0].y);
    q[3] = make_float4(tb[1].x, tb[1].y, tb[2].x, tb[2].y);
    q[4] = make_float4(tb[3].x, tb[3].y, tb[4].x, tb[4].y);
    if (MIXED) {
#pragma unroll
      for (int p = 0; p < 5; ++p) ta[p] = splat2(0.f);
      PHASE();
      mvf<D>(W + tnofs + L::N_W1J, xr, ta);
      q[5] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
      q[6] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
      q[7] = make_float4(ta[4].x, ta[4].y, 0.f, 0.f);
    }
  }
  __syncthreads();
  if (tid >= n_t) return;
  const int64_t n = (int64_t)t0 + tid;
  float4* Bn = reinterpret_cast<float4*>(B + n * 4 * D);
  const uint8_t fl = flags[n];
  if (fl & FLAG_DIRICHLET) {  // constant row: sends nothing
#pragma unroll
    for (int i = 0; i < 10; ++i) Bn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float zero[D];
#pragma unroll
    for (int o = 0; o < D; ++o) zero[o] = 0.f;
    store10(out + n * D, zero);
    if (PG) {  // a constant row still acts as a neighbour: its (x, 1) group feeds the W1j products of pass B
      float* r = rec + n * REC;
      rec_group(r, x, D, 1.f);
      for (int gI = 1; gI < REC / 16; ++gI)
        if (gI != 12 && gI != 13 && gI != 27) rec_group(r + 16 * gI, zero, 0);   // (12, 13, 27: pass B)
    }
    return;
  }
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (tid >> 6);
  const uint4* slots = ell + (int64_t)slice_off[slice] * 64 + lane;
  const int nslots = slice_deg[slice];
  if (MIXED && (fl & FLAG_NEUMANN)) {
    // ---- Neumann row: y = LN(N2 relu(q) + nb2), q = nb1 + deg gN + N1h x + Gn S_n + N1p [prb | normal]; the row was
    // REPLACED, so there is no residual path
    const float* TN = W + tnofs;
    const float* Un = W + L::upd_neu(nl);
    v2f Pn[5], S_n[5], c_n[5], m_n[PG ? 15 : 1];
    if (PG) {
#pragma unroll
      for (int i = 0; i < 15; ++i) m_n[PG ? i : 0] = splat2(0.f);
    }
    ldu5(TN + L::N_B1, Pn);
#pragma unroll
    for (int p = 0; p < 5; ++p) S_n[p] = c_n[p] = splat2(0.f);
    PHASE();
    mvf<D>(TN + L::N_W1I, x, Pn);
    const float deg_out = pass_fwd<RS, 2 * D, SLOT_OUT, PG>(slots, nslots, lds, TN + L::N_A, Pn, S_n, c_n, m_n);
    v2f q2[5], gN[5], hid2[5], y2[5];
    ldu5(TN + L::N_NB1, q2);
    ldu5(TN + L::N_gN, gN);
#pragma unroll
    for (int p = 0; p < 5; ++p) q2[p] = __builtin_elementwise_fma(splat2(deg_out), gN[p], q2[p]);
    PHASE();
    mvf<D>(TN + L::N_N1H, x, q2);
    PHASE();
    mvf<D>(TN + L::N_GN, reinterpret_cast<const float*>(S_n), q2);
    float pq[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    pq[P] = nrm[n * 2];
    pq[P + 1] = nrm[n * 2 + 1];
    PHASE();
    mvf<P + 2>(TN + L::N_N1P, pq, q2);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid2[p] = __builtin_elementwise_max(q2[p], splat2(0.f));
    ldu5(TN + L::N_NB2, y2);
    PHASE();
    mvf<D>(TN + L::N_N2, reinterpret_cast<const float*>(hid2), y2);
    float y[D], w[D], dy[D], mu = 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) {
      y[o] = reinterpret_cast<const float*>(y2)[o];
      mu += y[o];
    }
    load10(wv + n * D, w);
    ln_fwd_bwd<L>(W, mu, y, w, dy);
    v2f dq2[5], g[5], dS_n[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) dq2[p] = g[p] = dS_n[p] = splat2(0.f);
    PHASE();
    mvb<D>(Un + L::NEU_W2, D, 0, dy, dq2);
    float dq[D];
#pragma unroll
    for (int o = 0; o < D; ++o) dq[o] = reinterpret_cast<const float*>(q2)[o] > 0.f ? reinterpret_cast<const float*>(dq2)[o] : 0.f;
    PHASE();
    mvb<D>(Un + L::NEU_W1, L::NEU_CAT, 0, dq, g);
    PHASE();
    mvb<D>(W + L::nfold(nl) + L::NF_G, D, 0, dq, dS_n);   // dS_n[k] = sum_o Gn[o][k] dq[o]
    Bn[0] = Bn[1] = Bn[2] = Bn[3] = Bn[4] = make_float4(0.f, 0.f, 0.f, 0.f);   // sends nothing through Phi_to
    Bn[5] = make_float4(Pn[0].x, Pn[0].y, Pn[1].x, Pn[1].y);
    Bn[6] = make_float4(Pn[2].x, Pn[2].y, Pn[3].x, Pn[3].y);
    Bn[7] = make_float4(Pn[4].x, Pn[4].y, dS_n[0].x, dS_n[0].y);
    Bn[8] = make_float4(dS_n[1].x, dS_n[1].y, dS_n[2].x, dS_n[2].y);
    Bn[9] = make_float4(dS_n[3].x, dS_n[3].y, dS_n[4].x, dS_n[4].y);
    float gn[D], go[D];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const v2f a = dS_n[p] * c_n[p];
      gn[2 * p] = a.x; gn[2 * p + 1] = a.y;
    }
    PHASE();
    mvb<D>(W + L::phi_neu(nl) + L::PHI_W1, L::EIN, 0, gn, g);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      go[2 * p] = g[p].x;
      go[2 * p + 1] = g[p].y;
    }
    store10(out + n * D, go);
    if (PG) {
      // ---- parameter-gradient record of a Neumann row (groups as fgnn_vjp.hip's PgRec): the interior branch's factors are zero
      float* r = rec + n * REC;
      const float* Wn = W + L::phi_neu(nl);
      float zero[D], mp[D], t[D];
#pragma unroll
      for (int o = 0; o < D; ++o) zero[o] = 0.f;
      rec_group(r, x, D, 1.f);                                                          // 0: x, 1
      for (int gI = 1; gI < 12; ++gI) rec_group(r + 16 * gI, zero, 0);
#pragma unroll
      for (int o = 0; o < D; ++o) t[o] = w[o] * y[o];
      rec_group(r + 14 * 16, t, D);                                                     // 14: w * yhat
      rec_group(r + 15 * 16, w, D);                                                     // 15: w
      for (int gI = 16; gI < 20; ++gI) rec_group(r + 16 * gI, zero, 0);
#pragma unroll
      for (int o = 0; o < D; ++o) mp[o] = deg_out * Wn[L::PHI_B2 + o];
      matvec10<D, true>(Wn + L::PHI_W2, D, 0, reinterpret_cast<const float*>(S_n), mp);
      rec_group(r + 20 * 16, mp, D, pq[0], pq[1], pq[2], pq[3], pq[4]);                 // 20: mp_n, prb, normal
      rec_group(r + 21 * 16, reinterpret_cast<const float*>(S_n), D, deg_out);          // 21: S_n, deg_out
      rec_group(r + 22 * 16, reinterpret_cast<const float*>(hid2), D, 1.f);             // 22: hid_n, 1
      rec_group(r + 23 * 16, dq, D);                                                    // 23: dq_n
      rec_group(r + 24 * 16, gn, D);                                                    // 24: sum of masked cotangents, own out-edges
      v2f dm[5];
#pragma unroll
      for (int p = 0; p < 5; ++p) dm[p] = splat2(0.f);
      PHASE();
      mvb<D>(Un + L::NEU_W1, L::NEU_CAT, D, dq, dm);
      rec_group(r + 25 * 16, reinterpret_cast<const float*>(dm), D);                    // 25: d mp_n
      rec_group(r + 26 * 16, dy, D);                                                    // 26: dy_n      (27: pass B)
      float da[32];
      const float* dsn = reinterpret_cast<const float*>(dS_n);
      const float* mn = reinterpret_cast<const float*>(m_n);
#pragma unroll
      for (int o = 0; o < D; ++o) {
        da[o * 3] = dsn[o] * mn[PG ? o : 0];
        da[o * 3 + 1] = dsn[o] * mn[PG ? 10 + o : 0];
        da[o * 3 + 2] = dsn[o] * mn[PG ? 20 + o : 0];
      }
      da[30] = da[31] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)                                                       // 28..29: dS_n (.) attr moments
        reinterpret_cast<float4*>(r + 28 * 16)[i] = make_float4(da[4 * i], da[4 * i + 1], da[4 * i + 2], da[4 * i + 3]);
    }
    return;
  }
  // ---- forward with activity counts
  v2f Pt[5], Pf[5], S_to[5], S_fr[5], c_to[5], c_fr[5];
  v2f m_to[PG ? 15 : 1], m_fr[PG ? 15 : 1];
  if (PG) {
#pragma unroll
    for (int i = 0; i < 15; ++i) m_to[PG ? i : 0] = m_fr[PG ? i : 0] = splat2(0.f);
  }
  ldu5(T + L::T_B1_TO, Pt);
  ldu5(T + L::T_B1_FR, Pf);
#pragma unroll
  for (int p = 0; p < 5; ++p) S_to[p] = S_fr[p] = c_to[p] = c_fr[p] = splat2(0.f);
  PHASE();
  mvf<D>(T + L::T_W1I_TO, x, Pt);
  const float deg_in = pass_fwd<RS, 0, SLOT_IN, PG>(slots, nslots, lds, T + L::T_A_TO, Pt, S_to, c_to, m_to);
  PHASE();
  mvf<D>(T + L::T_W1I_FR, x, Pf);
  const float deg_out = pass_fwd<RS, D, SLOT_OUT, PG>(slots, nslots, lds, T + L::T_A_FR, Pf, S_fr, c_fr, m_fr);
  const float* Wf = W + lofs + L::L_FOLD;
  const float* Wu = W + lofs + L::L_UPD;
  const float* Wa = W + L::AL_W;
  const float* sto = reinterpret_cast<const float*>(S_to);
  const float* sfr = reinterpret_cast<const float*>(S_fr);
  float pq[P];
#pragma unroll
  for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
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
  v2f q2[5], g1[5], g2[5], upd2[5], hid2[5];
  ldu5(T + L::T_HB, q2);
  ldu5(T + L::T_gTO, g1);
  ldu5(T + L::T_gFR, g2);
#pragma unroll
  for (int p = 0; p < 5; ++p)
    q2[p] = __builtin_elementwise_fma(splat2(deg_in), g1[p], __builtin_elementwise_fma(splat2(deg_out), g2[p], q2[p]));
  PHASE();
  mvf<D>(T + L::T_U1H, x, q2);
  PHASE();
  mvf<D>(T + L::T_GTO, sto, q2);
  PHASE();
  mvf<D>(T + L::T_GFR, sfr, q2);
  PHASE();
  mvf<P>(T + L::T_U1P, pq, q2);
#pragma unroll
  for (int p = 0; p < 5; ++p) hid2[p] = __builtin_elementwise_max(q2[p], splat2(0.f));
  ldu5(T + L::T_C2, upd2);
  PHASE();
  mvf<D>(T + L::T_U2, reinterpret_cast<const float*>(hid2), upd2);
  const float* qf = reinterpret_cast<const float*>(q2);
  const float* upd = reinterpret_cast<const float*>(upd2);
  float y[D], mu = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = fmaf(al, upd[o], x[o]);
    mu += y[o];
  }
  // ---- backward: LayerNorm
  float w[D], dy[D];
  load10(wv + n * D, w);
  ln_fwd_bwd<L>(W, mu, y, w, dy);
  float dal = 0.f, dupd[D];
  v2f g[5];
#pragma unroll
  for (int o = 0; o < D; ++o) {
    dal = fmaf(dy[o], upd[o], dal);
    dupd[o] = al * dy[o];
  }
#pragma unroll
  for (int p = 0; p < 5; ++p) g[p] = (v2f){dy[2 * p], dy[2 * p + 1]};  // residual path y = x + ...
  dal *= al * (1.f - al);
  // ---- update MLP, gate, folded second Phi layer
  v2f dq2[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) dq2[p] = splat2(0.f);
  PHASE();
  mvb<D>(Wu + L::UPD_W2, D, 0, dupd, dq2);
  float dq[D];
#pragma unroll
  for (int o = 0; o < D; ++o) dq[o] = qf[o] > 0.f ? reinterpret_cast<const float*>(dq2)[o] : 0.f;
  PHASE();
  mvb<D>(Wu + L::UPD_W1, L::CAT, 0, dq, g);
  v2f dS_to[5], dS_fr[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    g[p] = __builtin_elementwise_fma((v2f){Wa[2 * p], Wa[2 * p + 1]}, splat2(dal), g[p]);
    dS_to[p] = (v2f){Wf[L::F_ATO + 2 * p], Wf[L::F_ATO + 2 * p + 1]} * splat2(dal);
    dS_fr[p] = (v2f){Wf[L::F_AFR + 2 * p], Wf[L::F_AFR + 2 * p + 1]} * splat2(dal);
  }
  PHASE();
  mvb<D>(Wf + L::F_GTO, D, 0, dq, dS_to);   // dS_to[k] = sum_o G_to[o][k] dq[o] + a_to[k] dal
  PHASE();
  mvb<D>(Wf + L::F_GFR, D, 0, dq, dS_fr);
  Bn[0] = make_float4(Pt[0].x, Pt[0].y, Pt[1].x, Pt[1].y);
  Bn[1] = make_float4(Pt[2].x, Pt[2].y, Pt[3].x, Pt[3].y);
  Bn[2] = make_float4(Pt[4].x, Pt[4].y, dS_to[0].x, dS_to[0].y);
  Bn[3] = make_float4(dS_to[1].x, dS_to[1].y, dS_to[2].x, dS_to[2].y);
  Bn[4] = make_float4(dS_to[3].x, dS_to[3].y, dS_to[4].x, dS_to[4].y);
  Bn[5] = make_float4(Pf[0].x, Pf[0].y, Pf[1].x, Pf[1].y);
  Bn[6] = make_float4(Pf[2].x, Pf[2].y, Pf[3].x, Pf[3].y);
  Bn[7] = make_float4(Pf[4].x, Pf[4].y, dS_fr[0].x, dS_fr[0].y);
  Bn[8] = make_float4(dS_fr[1].x, dS_fr[1].y, dS_fr[2].x, dS_fr[2].y);
  Bn[9] = make_float4(dS_fr[3].x, dS_fr[3].y, dS_fr[4].x, dS_fr[4].y);
  // ---- the node's own edges: masked cotangent sum = dS * (number of active edges per output)
  float gt[D], gf[D];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const v2f a = dS_to[p] * c_to[p], b = dS_fr[p] * c_fr[p];
    gt[2 * p] = a.x; gt[2 * p + 1] = a.y;
    gf[2 * p] = b.x; gf[2 * p + 1] = b.y;
  }
  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  PHASE();
  mvb<D>(Wto + L::PHI_W1, L::EIN, 0, gt, g);
  PHASE();
  mvb<D>(Wfr + L::PHI_W1, L::EIN, 0, gf, g);
  float go[D];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    go[2 * p] = g[p].x;
    go[2 * p + 1] = g[p].y;
  }
  store10(out + n * D, go);
  if (PG) {
    // ---- parameter-gradient record (fgnn_pgrad.hip reduces sum_n A_n (x) B_n over these 16-float groups)
    float* r = rec + n * REC;
    float mp[D], t[D];
    rec_group(r, x, D, 1.f);                                                   // 0: x, 1
#pragma unroll
    for (int o = 0; o < D; ++o) mp[o] = deg_in * Wto[L::PHI_B2 + o];
    matvec10<D, true>(Wto + L::PHI_W2, D, 0, sto, mp);
    rec_group(r + 16, mp, D, pq[0], pq[1], P == 3 ? pq[P - 1] : 0.f);         // 1: mp_to, prb
#pragma unroll
    for (int o = 0; o < D; ++o) mp[o] = deg_out * Wfr[L::PHI_B2 + o];
    matvec10<D, true>(Wfr + L::PHI_W2, D, 0, sfr, mp);
    rec_group(r + 32, mp, D);                                                  // 2: mp_from
    rec_group(r + 48, sto, D, deg_in);                                         // 3: S_to, deg_in
    rec_group(r + 64, sfr, D, deg_out);                                        // 4: S_from, deg_out
    rec_group(r + 80, reinterpret_cast<const float*>(hid2), D, 1.f);           // 5: hid, 1
    rec_group(r + 96, dq, D, dal);                                             // 6: dq, ds
    rec_group(r + 112, gt, D);                                                 // 7: sum of masked cotangents, in-edges
    rec_group(r + 128, gf, D);                                                 // 8: same, out-edges
    v2f dm[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) dm[p] = (v2f){Wa[D + 2 * p], Wa[D + 2 * p + 1]} * splat2(dal);
    PHASE();
    mvb<D>(Wu + L::UPD_W1, L::CAT, D, dq, dm);
    rec_group(r + 144, reinterpret_cast<const float*>(dm), D);                 // 9: d mp_to
#pragma unroll
    for (int p = 0; p < 5; ++p) dm[p] = (v2f){Wa[2 * D + 2 * p], Wa[2 * D + 2 * p + 1]} * splat2(dal);
    PHASE();
    mvb<D>(Wu + L::UPD_W1, L::CAT, 2 * D, dq, dm);
    rec_group(r + 160, reinterpret_cast<const float*>(dm), D);                 // 10: d mp_from
    rec_group(r + 176, dupd, D);                                               // 11: d upd0   (12, 13: pass B)
#pragma unroll
    for (int o = 0; o < D; ++o) t[o] = w[o] * y[o];
    rec_group(r + 224, t, D);                                                  // 14: w * yhat
    rec_group(r + 240, w, D);                                                  // 15: w
    // 16..19: dS[o] * (attr moments), index o*3 + c as in the W1 attr block; in-edges carry the mirrored attr
    float da[64];
    const float* dst = reinterpret_cast<const float*>(dS_to);
    const float* dsf = reinterpret_cast<const float*>(dS_fr);
    const float* mt = reinterpret_cast<const float*>(m_to);
    const float* mf = reinterpret_cast<const float*>(m_fr);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      da[o * 3] = -dst[o] * mt[o];
      da[o * 3 + 1] = -dst[o] * mt[10 + o];
      da[o * 3 + 2] = dst[o] * mt[20 + o];
      da[30 + o * 3] = dsf[o] * mf[o];
      da[30 + o * 3 + 1] = dsf[o] * mf[10 + o];
      da[30 + o * 3 + 2] = dsf[o] * mf[20 + o];
    }
    da[60] = da[61] = da[62] = da[63] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      reinterpret_cast<float4*>(r + 256)[i] = make_float4(da[4 * i], da[4 * i + 1], da[4 * i + 2], da[4 * i + 3]);
    if (P == 3) {   // mixed plan, interior row: no Neumann factors (27: pass B)
#pragma unroll
      for (int o = 0; o < D; ++o) t[o] = 0.f;
      for (int gI = 20; gI < 30; ++gI)
        if (gI != 27) rec_group(r + 16 * gI, t, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------- pass B
// acc[o] += 1[Pi[o] + row[PC + o] + AT . (sg a0, sg a1, a2) > 0] * row[DC + o]  over the slots carrying MASK.
// The transposed attr blocks store rows 0,1 NEGATED for the in-direction (T_A_TO) and plain for T_A_FR; the caller
// picks `flip` so that the effective attr is the edge's own: OUT slot of u = in-edge of n with attr a (no mirror).
template <int RS, int PC, int DC, unsigned MASK>
__device__ __forceinline__ void pass_rev(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                         const float* __restrict__ AT, float flip, const v2f* Pj, v2f* acc) {
  v2f wa[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) wa[i] = reinterpret_cast<const v2f*>(AT)[i];
  if (nslots <= 0) return;
  uint4 c0 = slots[0];
  uint4 c1 = slots[(int64_t)min(1, nslots - 1) * 64];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 2, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY && (w & MASK)) {
      const v2f a0 = splat2(flip * __uint_as_float(c0.y)), a1 = splat2(flip * __uint_as_float(c0.z));
      const v2f a2 = splat2(__uint_as_float(c0.w));
      const float* row = lds + (int)(w & 0xFFFFu) * RS;
      v2f pi[5], ds[5], z[5];
      if (PC % 4 == 0) row10(row + PC, pi); else row10u(row + PC, pi);
      if (DC % 4 == 0) row10(row + DC, ds); else row10u(row + DC, ds);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = pi[p] + Pj[p];
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[p], a0, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[5 + p], a1, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[10 + p], a2, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) acc[p] += (v2f){z[p].x > 0.f ? ds[p].x : 0.f, z[p].y > 0.f ? ds[p].y : 0.f};
    }
    c0 = c1;
    c1 = nx;
  }
}

// IN slots of a mixed plan: the sender n of edge (n -> u) is an interior row (Phi_from: weights AF, projection Pjf, sum af) or
// a Neumann row (Phi_neumann: AN, Pjn, an) -- nflag[row] tells; both kinds keep Pi and dS in the second half of their B row.
// one IN slot seen from the receiver: acc[o] += 1[row[o] + Pj[o] + AT . (-a0, -a1, a2) > 0] * row[10 + o]   (half-row [P | dS])
__device__ __forceinline__ void rev_slot_in(const float* __restrict__ row, const float* __restrict__ AT, const uint4& c, const v2f* Pj,
                                            v2f* acc) {
  const v2f* wa = reinterpret_cast<const v2f*>(AT);   // wave-uniform address: the weights stay in scalar registers
  const v2f a0 = splat2(-__uint_as_float(c.y)), a1 = splat2(-__uint_as_float(c.z)), a2 = splat2(__uint_as_float(c.w));
  v2f pi[5], ds[5], z[5];
  row10(row, pi);
  row10u(row + D, ds);
#pragma unroll
  for (int p = 0; p < 5; ++p) z[p] = pi[p] + Pj[p];
#pragma unroll
  for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[p], a0, z[p]);
#pragma unroll
  for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[5 + p], a1, z[p]);
#pragma unroll
  for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[10 + p], a2, z[p]);
#pragma unroll
  for (int p = 0; p < 5; ++p) acc[p] += (v2f){z[p].x > 0.f ? ds[p].x : 0.f, z[p].y > 0.f ? ds[p].y : 0.f};
}

// IN slots of a mixed plan: the sender n of edge (n -> u) is an interior row (Phi_from: weights AF, projection Pjf, sum af) or
// a Neumann row (Phi_neumann: AN, Pjn, an) -- nflag[row] tells; both kinds keep Pi and dS in the second half of their B row.
template <int RS>
__device__ __forceinline__ void pass_rev_in_mixed(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                                  const int32_t* __restrict__ nflag, const float* __restrict__ AF,
                                                  const float* __restrict__ AN, const v2f* Pjf, const v2f* Pjn, v2f* af,
                                                  v2f* an) {
  if (nslots <= 0) return;
  uint4 c0 = slots[0];
  uint4 c1 = slots[(int64_t)min(1, nslots - 1) * 64];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 2, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY && (w & SLOT_IN)) {
      const int ri = (int)(w & 0xFFFFu);
      const float* row = lds + ri * RS;
      if (nflag[ri] != 0) rev_slot_in(row, AN, c0, Pjn, an);   // (rare: boundary rows only)
      else rev_slot_in(row, AF, c0, Pjf, af);
    }
    c0 = c1;
    c1 = nx;
  }
}

template <int P, bool MIXED, bool PG>
__global__ __launch_bounds__(VT) void k_vjp_tile_b(int n_tiles, int chunk, const int32_t* __restrict__ tile_ptr,
                                                   const int32_t* __restrict__ tile_slice, const int32_t* __restrict__ halo,
                                                   const int32_t* __restrict__ halo_cnt, const int32_t* __restrict__ slice_off,
                                                   const uint8_t* __restrict__ slice_deg, const uint4* __restrict__ ell,
                                                   const uint8_t* __restrict__ flags, const float* __restrict__ W, int nl,
                                                   int lofs, int tofs, int tnofs, const float* __restrict__ h,
                                                   const float* __restrict__ B, float* __restrict__ out,
                                                   float* __restrict__ rec) {
  using L = WLayout<P>;
  constexpr int REC = P == 3 ? 480 : PGREC;
  // B row in memory: [Pt 10 | dS_to 10 | Pf 10 | dS_fr 10].  The OUT slots need the first half of the senders' rows, the IN slots the
  // second: the halves are staged one after the other into 80-byte LDS rows -- 26 KB per workgroup instead of 52, so six workgroups
  // fit a CU where three did (the kernel needs < 96 VGPRs: LDS was what held it at three waves per SIMD); same bytes from memory.
  constexpr int RS = 20;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int tid = threadIdx.x;
  const int32_t t0 = tile_ptr[tile];
  const int n_t = tile_ptr[tile + 1] - t0;
  const int n_h = halo_cnt[tile];
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  const float* T = W + tofs;
  auto stage = [&](int half) {   // float4 units: 5 per half row
    for (int i = tid; i < (n_t + n_h) * 5; i += VT) {
      const int row = i / 5, c = i - row * 5;
      const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
      reinterpret_cast<float4*>(lds)[row * 5 + c] = reinterpret_cast<const float4*>(B + node * 4 * D)[half * 5 + c];
    }
  };
  stage(0);
  int32_t* nflag = reinterpret_cast<int32_t*>(lds + (n_t + n_h) * RS);
  bool tile_neu = false;
  if (MIXED) {
    int any = 0;
    for (int row = tid; row < n_t + n_h; row += VT) {
      const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
      const int f = flags[node] & FLAG_NEUMANN;
      nflag[row] = f;
      any |= f;
    }
    tile_neu = __syncthreads_or(any) != 0;   // (also the barrier in front of the first walk)
  } else {
    __syncthreads();
  }
  const bool active = tid < n_t;            // (every thread stays for the second staging pass and its barriers)
  const int64_t u = (int64_t)t0 + min(tid, n_t - 1);
  const int lane = tid & 63;
  const int slice = tile_slice[tile] + (min(tid, n_t - 1) >> 6);
  const uint4* slots = ell + (int64_t)slice_off[slice] * 64 + lane;
  const int nslots = active ? slice_deg[slice] : 0;
  float x[D];
  load10(h + u * D, x);
  v2f Pj[5], at[5], af[5], an[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) Pj[p] = at[p] = af[p] = an[p] = splat2(0.f);
  // OUT slots: edge (u -> n) is an in-edge of n (Phi_to of n): Pt[n], dS_to[n] = first half row, attr = the slot's own
  // (T_A_TO holds the mirrored rows -> flip = -1 restores the plain attr weights)
  PHASE();
  mvf<D>(T + L::T_W1J_TO, x, Pj);
  pass_rev<RS, 0, D, SLOT_OUT>(slots, nslots, lds, T + L::T_A_TO, -1.f, Pj, at);
  __syncthreads();   // every wave is done with the first halves
  stage(1);
  __syncthreads();
  // IN slots: edge (n -> u) is an out-edge of n (Phi_from of n): Pf[n], dS_fr[n] = second half row, attr = mirror of the slot's
#pragma unroll
  for (int p = 0; p < 5; ++p) Pj[p] = splat2(0.f);
  PHASE();
  mvf<D>(T + L::T_W1J_FR, x, Pj);
  if (MIXED && tile_neu) {
    v2f Pjn[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) Pjn[p] = splat2(0.f);
    PHASE();
    mvf<D>(W + tnofs + L::N_W1J, x, Pjn);
    pass_rev_in_mixed<RS>(slots, nslots, lds, nflag, T + L::T_A_FR, W + tnofs + L::N_A, Pj, Pjn, af, an);
  } else {
    pass_rev<RS, 0, D, SLOT_IN>(slots, nslots, lds, T + L::T_A_FR, -1.f, Pj, af);
  }
  if (!active) return;
  if (PG) {  // neighbour-side cotangent sums: W1j gradients are sum_u acc[u] (x) x[u]
    rec_group(rec + u * REC + 192, reinterpret_cast<const float*>(at), D);
    rec_group(rec + u * REC + 208, reinterpret_cast<const float*>(af), D);
    if (P == 3) rec_group(rec + u * REC + 27 * 16, reinterpret_cast<const float*>(an), D);   // 27: neighbour-side sums through Phi_neumann
  }
  float go[D];
  load10(out + u * D, go);
  v2f g[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) g[p] = (v2f){go[2 * p], go[2 * p + 1]};
  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  PHASE();
  mvb<D>(Wto + L::PHI_W1, L::EIN, D, reinterpret_cast<const float*>(at), g);
  PHASE();
  mvb<D>(Wfr + L::PHI_W1, L::EIN, D, reinterpret_cast<const float*>(af), g);
  if (MIXED && tile_neu) {
    PHASE();
    mvb<D>(W + L::phi_neu(nl) + L::PHI_W1, L::EIN, D, reinterpret_cast<const float*>(an), g);
  }
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    go[2 * p] = g[p].x;
    go[2 * p + 1] = g[p].y;
  }
  store10(out + u * D, go);
}

// ---------------------------------------------------------------------------------------------- host
static int tile_vjp_launch(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                           const float* w, float* out, float* work, float* rec, hipStream_t st) {
  ARG_CHECK(p && p->tiled && (p->mixed || nl == 1), "tiled VJP: single-layer dirichlet plans, mixed plans");
  ARG_CHECK(!p->mixed || nrm, "mixed plan: needs unit normals");
  ARG_CHECK(!(p->mixed && rec) || nl == 1, "parameter-gradient records: single-layer blocks");
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = (unsigned)(chunk * 8);
  const size_t lds_b = (size_t)p->max_rows * (20 + (p->mixed ? 1 : 0)) * 4;
  ARG_CHECK(lds_b <= 160 * 1024, "tile + halo rows exceed the LDS budget of the tiled VJP");
#define VJP_PLAN p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell, p->flags_p
  if (p->mixed) {
    using L = WLayout<3>;
    const int lofs = L::layer(nl - 1), tofs = L::tp_layer(nl, true, nl - 1), tnofs = L::tp_neu(nl);
    const int na = (int)p->n_tiles_plain, nb = (int)(p->n_tiles - p->n_tiles_plain);
    if (na > 0) {   // tiles without Neumann nodes of their own
      const int ch = (int)cdiv(na, 8);
      if (rec)
        LAUNCH("k_pgrad_tile_a", st, (k_vjp_tile_a<3, false, true><<<(unsigned)(ch * 8), VT, (size_t)p->max_rows * 20 * 4, st>>>(
            na, ch, p->tile_order, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, prb, nrm, w, work, out, rec)));
      else
        LAUNCH("k_vjp_tile_a", st, (k_vjp_tile_a<3, false, false><<<(unsigned)(ch * 8), VT, (size_t)p->max_rows * 20 * 4, st>>>(
            na, ch, p->tile_order, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, prb, nrm, w, work, out, rec)));
    }
    if (nb > 0) {
      const int ch = (int)cdiv(nb, 8);
      if (rec)
        LAUNCH("k_pgrad_tile_a", st, (k_vjp_tile_a<3, true, true><<<(unsigned)(ch * 8), VT, (size_t)p->max_rows * 32 * 4, st>>>(
            nb, ch, p->tile_order + na, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, prb, nrm, w, work, out, rec)));
      else
        LAUNCH("k_vjp_tile_a", st, (k_vjp_tile_a<3, true, false><<<(unsigned)(ch * 8), VT, (size_t)p->max_rows * 32 * 4, st>>>(
            nb, ch, p->tile_order + na, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, prb, nrm, w, work, out, rec)));
    }
    if (rec)
      LAUNCH("k_pgrad_tile_b", st, (k_vjp_tile_b<3, true, true><<<grid, VT, lds_b, st>>>(
          (int)p->n_tiles, chunk, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, work, out, rec)));
    else
      LAUNCH("k_vjp_tile_b", st, (k_vjp_tile_b<3, true, false><<<grid, VT, lds_b, st>>>(
          (int)p->n_tiles, chunk, VJP_PLAN, W, nl, lofs, tofs, tnofs, h, work, out, rec)));
    HIP_TRY(hipGetLastError());
    return PSIGNN_OK;
  }
  using L = WLayout<2>;
  const size_t lds_a = (size_t)p->max_rows * 20 * 4;
#define VJP_ARGS_A (int)p->n_tiles, chunk, nullptr, VJP_PLAN, W, nl, L::layer(0), L::tp_layer(nl, false, 0), 0, h, prb, nrm, w, work, out, rec
#define VJP_ARGS_B (int)p->n_tiles, chunk, VJP_PLAN, W, nl, L::layer(0), L::tp_layer(nl, false, 0), 0, h, work, out, rec
  if (rec) {
    LAUNCH("k_pgrad_tile_a", st, (k_vjp_tile_a<2, false, true><<<grid, VT, lds_a, st>>>(VJP_ARGS_A)));
    LAUNCH("k_pgrad_tile_b", st, (k_vjp_tile_b<2, false, true><<<grid, VT, lds_b, st>>>(VJP_ARGS_B)));
  } else {
    LAUNCH("k_vjp_tile_a", st, (k_vjp_tile_a<2, false, false><<<grid, VT, lds_a, st>>>(VJP_ARGS_A)));
    LAUNCH("k_vjp_tile_b", st, (k_vjp_tile_b<2, false, false><<<grid, VT, lds_b, st>>>(VJP_ARGS_B)));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// h, prb, nrm (mixed plans), w, out in PLAN order; work: (N, 40) floats for B.
int psignn_f_tile_vjp(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                      const float* w, float* out, float* work, hipStream_t st) {
  return tile_vjp_launch(p, W, nl, h, prb, nrm, w, out, work, nullptr, st);
}
// same, additionally filling the parameter-gradient records rec: (N, 320) floats, mixed plans (N, 480) (fgnn_pgrad.hip)
int psignn_f_tile_vjp_rec(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                          const float* w, float* out, float* work, float* rec, hipStream_t st) {
  ARG_CHECK(rec, "NULL record buffer");
  return tile_vjp_launch(p, W, nl, h, prb, nrm, w, out, work, rec, st);
}
