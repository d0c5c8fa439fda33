// Vector-Jacobian product of f_theta:  out = w^T (d f / d h)  (gfx950), both families (a multi-layer mixed block only
// applies its last layer, mixed/psignn/model.py:221-245; multi-layer dirichlet blocks are not supported).
//
// This is what the reference obtains from autograd -- torch.autograd.grad(new_H, H, v) -- inside the implicit
// backward hook (dirichlet/psignn/model.py:210-223: Broyden on y = J^T y + grad), the Hutchinson Jacobian
// regulariser (jac_loss_estimate, model.py:416-435) and the power method (model.py:437-452).  SURVEY §8f-1.
//
// Autograd scatters every edge's cotangent to the neighbour with index_add.  Here the transpose is written as
// two GATHER passes over the plan's CSR/CSC lists, so there are no atomics and the result is reproducible:
//   pass 1 (node n): back through LayerNorm, the gated update MLP and the second Phi layer; keeps
//           B[n] = { Pt[n], Pf[n] (target-side projections incl. bias), dS_to[n], dS_fr[n] [, Pn[n], dS_n[n]] } and
//           writes the node-local part of the result:
//           dy + U1_h^T dq + w_alpha,h * dalpha + W1i_to^T sum_e g_e + W1i_fr^T sum_e g'_e
//           with g_e = dS_to[n] * 1[z_e > 0] over n's in-edges, g'_e likewise over its out-edges;
//   pass 2 (node u): the contributions u receives as somebody's neighbour:
//           for u's out-edges (u -> n), which are in-edges of n:  acc_t += dS_to[n] * 1[Pt[n] + Pjt[u] + At a > 0]
//           for u's in-edges  (n -> u), which are out-edges of n: acc_f += dS_fr[n] * 1[Pf[n] + Pjf[u] + Af a > 0]
//           out[u] += W1j_to^T acc_t + W1j_fr^T acc_f.
// Dirichlet rows of f are constants (their Jacobian rows vanish): they send nothing, but still receive.
// Mixed family: a Neumann row is REPLACED by update_neumann([h, Phi_neumann(h), prb, normal]) before LayerNorm
// (mixed/psignn/model.py:236,241), so its cotangent flows through that branch only (Phi_neumann is of the
// Phi_from type: out-edges of n; pass 2 adds acc_n over u's in-edges).
#include "fgnn_common.h"
#include <string.h>
#include <stdlib.h>

// The kernels below read ~1 500 wave-uniform weights.  Fully unrolled, the compiler hoists all their scalar loads to
// the top and then spills > 1 000 SGPRs into VGPR lanes; in the record-writing (PG) instantiation of the mixed family
// that went wrong on ROCm 7.2 (per-lane values came back as spilled weights).  A compiler-level memory barrier between
// the phases keeps each phase's scalar loads next to their use (<= ~100 live SGPRs): no spills, and faster.
#define PHASE() asm volatile("" ::: "memory")

// out[k] (+)= sum_o W[o*ld + off + k] * g[o]   (transposed product, W wave-uniform)
template <int K, bool ACC>
__device__ __forceinline__ void matvecT(const float* __restrict__ W, int ld, int off, const float* g, float* out) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = ACC ? out[k] : 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) s = fmaf(W[o * ld + off + k], g[o], s);
    out[k] = s;
  }
}

// z[o] = Pi[o] + pj[o] + W1[o, 20:23] . a   (pre-activation of one edge; W1 = first Phi layer, ld = 23)
__device__ __forceinline__ void edge_z(const float* __restrict__ W1, const float* Pi, const float* pj, float a0, float a1,
                                       float a2, float* z) {
  constexpr int EIN = 2 * D + 3;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    float t = Pi[o] + pj[o];
    t = fmaf(W1[o * EIN + 2 * D], a0, t);
    t = fmaf(W1[o * EIN + 2 * D + 1], a1, t);
    t = fmaf(W1[o * EIN + 2 * D + 2], a2, t);
    z[o] = t;
  }
}

// neighbour-side projections Pj[n] = { W1j_to h_n, W1j_fr h_n [, W1j_neu h_n] }
template <int P, bool MIXED>
__global__ __launch_bounds__(256) void k_vjp_project(int64_t N, const float* __restrict__ W, int lofs, int nofs,
                                                     const float* __restrict__ h, float* __restrict__ Pj) {
  using L = WLayout<P>;
  constexpr int NP = MIXED ? 3 : 2;
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float x[D], t[D];
  load10(h + n * D, x);
  matvec10<D, false>(W + lofs + L::L_TO + L::PHI_W1, L::EIN, D, x, t);
  store10(Pj + n * NP * D, t);
  matvec10<D, false>(W + lofs + L::L_FROM + L::PHI_W1, L::EIN, D, x, t);
  store10(Pj + n * NP * D + D, t);
  if (MIXED) {
    matvec10<D, false>(W + nofs + L::PHI_W1, L::EIN, D, x, t);
    store10(Pj + n * NP * D + 2 * D, t);
  }
}

// Parameter-gradient records (PG mode; reduced by fgnn_pgrad.hip): 16-float groups per node.  Groups 0..19 as written by
// the tiled kernels (fgnn_tile_vjp.hip); the mixed family appends 20: mp_n | prb | normal, 21: S_n | deg_out,
// 22: hid_n | 1, 23: dq_n, 24: gn, 25: d mp_n, 26: dy_n, 27: acc_n (pass 2), 28..29: dS_n (.) attr.
template <bool MIXED>
struct PgRec {
  static constexpr int NG = MIXED ? 30 : 20, SZ = 16 * NG;
};
__device__ __forceinline__ void pg_group(float* __restrict__ g, const float* v, int n, float t0 = 0.f, float t1 = 0.f,
                                         float t2 = 0.f, float t3 = 0.f, float t4 = 0.f) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    r[i] = i < n ? v[i] : (i == n ? t0 : (i == n + 1 ? t1 : (i == n + 2 ? t2 : (i == n + 3 ? t3 : (i == n + 4 ? t4 : 0.f)))));
  float4* q = reinterpret_cast<float4*>(g);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
}
__device__ __forceinline__ void pg_zero(float* __restrict__ g, int first, int last) {  // groups [first, last)
  for (int i = first * 4; i < last * 4; ++i) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int P, bool MIXED, bool PG>
__global__ __launch_bounds__(256) void k_vjp_local(int64_t N, const float* __restrict__ W, int lofs, int nofs, int unofs,
                                                   const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_nbr,
                                                   const float* __restrict__ csr_attr, const int32_t* __restrict__ csc_ptr,
                                                   const int32_t* __restrict__ csc_nbr, const float* __restrict__ csc_attr,
                                                   const uint8_t* __restrict__ flags, const float* __restrict__ h,
                                                   const float* __restrict__ prb, const float* __restrict__ nrm,
                                                   const float* __restrict__ wv, const float* __restrict__ Pj,
                                                   float* __restrict__ B, float* __restrict__ out,
                                                   float* __restrict__ rec) {
  using L = WLayout<P>;
  using R = PgRec<MIXED>;
  constexpr int NP = MIXED ? 3 : 2;   // Pj row = NP * 10 floats
  constexpr int NB = MIXED ? 6 : 4;   // B row  = NB * 10 floats
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  const float* Wn = W + nofs;
  const float* Wu = W + lofs + L::L_UPD;
  const float* Wa = W + L::AL_W;
  const uint8_t fl = flags[n];
  float x[D], Pt[D], Pf[D], Pn[D], zero[D];
  load10(h + n * D, x);
#pragma unroll
  for (int o = 0; o < D; ++o) {
    Pt[o] = Wto[L::PHI_B1 + o];
    Pf[o] = Wfr[L::PHI_B1 + o];
    zero[o] = 0.f;
  }
  PHASE();
  matvec10<D, true>(Wto + L::PHI_W1, L::EIN, 0, x, Pt);
  PHASE();
  matvec10<D, true>(Wfr + L::PHI_W1, L::EIN, 0, x, Pf);
  float* Bn = B + n * NB * D;
  store10(Bn, Pt);
  store10(Bn + D, Pf);
  if (MIXED) {
#pragma unroll
    for (int o = 0; o < D; ++o) Pn[o] = Wn[L::PHI_B1 + o];
    PHASE();
    matvec10<D, true>(Wn + L::PHI_W1, L::EIN, 0, x, Pn);
    store10(Bn + 4 * D, Pn);
  }
  if (fl & FLAG_DIRICHLET) {  // constant row: sends nothing
    store10(Bn + 2 * D, zero);
    store10(Bn + 3 * D, zero);
    if (MIXED) store10(Bn + 5 * D, zero);
    store10(out + n * D, zero);
    if (PG) {  // a constant row still acts as a neighbour: its (x, 1) group feeds the W1j products of pass 2
      float* r = rec + n * R::SZ;
      pg_group(r, x, D, 1.f);
      pg_zero(r, 1, 12);
      pg_zero(r, 14, MIXED ? 27 : 20);
      if (MIXED) pg_zero(r, 28, 30);
    }
    return;
  }
  const int32_t ib = csc_ptr[n], ie = csc_ptr[n + 1], ob = csr_ptr[n], oe = csr_ptr[n + 1];
  float w[D], g[D];
  load10(wv + n * D, w);

  if (MIXED && (fl & FLAG_NEUMANN)) {
    // ------------------------------------------------------------------ Neumann row: y = update_neumann(cat_n)
    const float* Un = W + unofs;
    float S_n[D], z[D], pj[D];
#pragma unroll
    for (int o = 0; o < D; ++o) S_n[o] = 0.f;
    PHASE();
    for (int32_t i = ob; i < oe; ++i) {
      load10(Pj + (int64_t)csr_nbr[i] * NP * D + 2 * D, pj);
      edge_z(Wn + L::PHI_W1, Pn, pj, csr_attr[3 * (int64_t)i], csr_attr[3 * (int64_t)i + 1], csr_attr[3 * (int64_t)i + 2], z);
#pragma unroll
      for (int o = 0; o < D; ++o) S_n[o] += fmaxf(z[o], 0.f);
    }
    float mp_n[D], q[D], hid[D], y[D];
#pragma unroll
    for (int o = 0; o < D; ++o) {
      mp_n[o] = (float)(oe - ob) * Wn[L::PHI_B2 + o];
      q[o] = Un[L::NEU_B1 + o];
    }
    PHASE();
    matvec10<D, true>(Wn + L::PHI_W2, D, 0, S_n, mp_n);
    PHASE();
    matvec10<D, true>(Un + L::NEU_W1, L::NEU_CAT, 0, x, q);
    PHASE();
    matvec10<D, true>(Un + L::NEU_W1, L::NEU_CAT, D, mp_n, q);
    float pq[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    pq[P] = nrm[n * 2];
    pq[P + 1] = nrm[n * 2 + 1];
    PHASE();
    matvec10<P + 2, true>(Un + L::NEU_W1, L::NEU_CAT, 2 * D, pq, q);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      hid[o] = fmaxf(q[o], 0.f);
      y[o] = Un[L::NEU_B2 + o];
    }
    PHASE();
    matvec10<D, true>(Un + L::NEU_W2, D, 0, hid, y);
    // LayerNorm backward
    float mu = 0.f, var = 0.f;
#pragma unroll
    for (int o = 0; o < D; ++o) mu += y[o];
    mu *= (1.f / D);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      float c = y[o] - mu;
      var = fmaf(c, c, var);
    }
    var *= (1.f / D);
    const float rs = 1.f / sqrtf(var + 1e-5f);
    float dyh[D], dy[D], m1 = 0.f, m2 = 0.f;
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
    float dq[D], dmp_n[D], dS_n[D];
    PHASE();
    matvecT<D, false>(Un + L::NEU_W2, D, 0, dy, dq);
#pragma unroll
    for (int o = 0; o < D; ++o) dq[o] = q[o] > 0.f ? dq[o] : 0.f;
    PHASE();
    matvecT<D, false>(Un + L::NEU_W1, L::NEU_CAT, 0, dq, g);        // no residual path: the row was replaced
    PHASE();
    matvecT<D, false>(Un + L::NEU_W1, L::NEU_CAT, D, dq, dmp_n);
    PHASE();
    matvecT<D, false>(Wn + L::PHI_W2, D, 0, dmp_n, dS_n);
    store10(Bn + 2 * D, zero);
    store10(Bn + 3 * D, zero);
    store10(Bn + 5 * D, dS_n);
    if (PG && MIXED) {  // everything but gn / the attr sums is known here: written before the second edge loop
      float* r = rec + n * R::SZ;
      float t[D];
      pg_group(r, x, D, 1.f);
      pg_zero(r, 1, 12);                                        // the interior branch's factors
#pragma unroll
      for (int o = 0; o < D; ++o) t[o] = w[o] * y[o];
      pg_group(r + 14 * 16, t, D);
      pg_group(r + 15 * 16, w, D);
      pg_zero(r, 16, 20);
      pg_group(r + 20 * 16, mp_n, D, pq[0], pq[1], pq[2], pq[3], pq[4]);
      pg_group(r + 21 * 16, S_n, D, (float)(oe - ob));
      pg_group(r + 22 * 16, hid, D, 1.f);
      pg_group(r + 23 * 16, dq, D);
      pg_group(r + 25 * 16, dmp_n, D);
      pg_group(r + 26 * 16, dy, D);
    }
    float gn[D], dsa[PG ? 32 : 1];
#pragma unroll
    for (int o = 0; o < D; ++o) gn[o] = 0.f;
    if (PG) {
#pragma unroll
      for (int i = 0; i < 32; ++i) dsa[PG ? i : 0] = 0.f;
    }
    PHASE();
    for (int32_t i = ob; i < oe; ++i) {
      const float a0 = csr_attr[3 * (int64_t)i], a1 = csr_attr[3 * (int64_t)i + 1], a2 = csr_attr[3 * (int64_t)i + 2];
      load10(Pj + (int64_t)csr_nbr[i] * NP * D + 2 * D, pj);
      edge_z(Wn + L::PHI_W1, Pn, pj, a0, a1, a2, z);
#pragma unroll
      for (int o = 0; o < D; ++o) {
        const float m = z[o] > 0.f ? dS_n[o] : 0.f;
        gn[o] += m;
        if (PG) {
          dsa[PG ? o * 3 : 0] = fmaf(m, a0, dsa[PG ? o * 3 : 0]);
          dsa[PG ? o * 3 + 1 : 0] = fmaf(m, a1, dsa[PG ? o * 3 + 1 : 0]);
          dsa[PG ? o * 3 + 2 : 0] = fmaf(m, a2, dsa[PG ? o * 3 + 2 : 0]);
        }
      }
    }
    PHASE();
    matvecT<D, true>(Wn + L::PHI_W1, L::EIN, 0, gn, g);
    store10(out + n * D, g);
    if (PG && MIXED) {
      float* r = rec + n * R::SZ;
      pg_group(r + 24 * 16, gn, D);
      dsa[PG ? 30 : 0] = dsa[PG ? 31 : 0] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        reinterpret_cast<float4*>(r + 28 * 16)[i] = make_float4(dsa[PG ? 4 * i : 0], dsa[PG ? 4 * i + 1 : 0], dsa[PG ? 4 * i + 2 : 0], dsa[PG ? 4 * i + 3 : 0]);
    }
    return;
  }

  // ---------------------------------------------------------------------- interior row
  float S_to[D], S_fr[D], z[D], pj[D];
#pragma unroll
  for (int o = 0; o < D; ++o) S_to[o] = S_fr[o] = 0.f;
  PHASE();
  for (int32_t i = ib; i < ie; ++i) {
    load10(Pj + (int64_t)csc_nbr[i] * NP * D, pj);
    edge_z(Wto + L::PHI_W1, Pt, pj, csc_attr[3 * (int64_t)i], csc_attr[3 * (int64_t)i + 1], csc_attr[3 * (int64_t)i + 2], z);
#pragma unroll
    for (int o = 0; o < D; ++o) S_to[o] += fmaxf(z[o], 0.f);
  }
  PHASE();
  for (int32_t i = ob; i < oe; ++i) {
    load10(Pj + (int64_t)csr_nbr[i] * NP * D + D, pj);
    edge_z(Wfr + L::PHI_W1, Pf, pj, csr_attr[3 * (int64_t)i], csr_attr[3 * (int64_t)i + 1], csr_attr[3 * (int64_t)i + 2], z);
#pragma unroll
    for (int o = 0; o < D; ++o) S_fr[o] += fmaxf(z[o], 0.f);
  }
  float mp_to[D], mp_fr[D], pq[P];
#pragma unroll
  for (int o = 0; o < D; ++o) {
    mp_to[o] = (float)(ie - ib) * Wto[L::PHI_B2 + o];
    mp_fr[o] = (float)(oe - ob) * Wfr[L::PHI_B2 + o];
  }
  PHASE();
  matvec10<D, true>(Wto + L::PHI_W2, D, 0, S_to, mp_to);
  PHASE();
  matvec10<D, true>(Wfr + L::PHI_W2, D, 0, S_fr, mp_fr);
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
  float q[D], upd[D], y[D], hid[D];
#pragma unroll
  for (int o = 0; o < D; ++o) q[o] = Wu[L::UPD_B1 + o];
  PHASE();
  matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 0, x, q);
  PHASE();
  matvec10<D, true>(Wu + L::UPD_W1, L::CAT, D, mp_to, q);
  PHASE();
  matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 2 * D, mp_fr, q);
  PHASE();
  matvec10<P, true>(Wu + L::UPD_W1, L::CAT, 3 * D, pq, q);
#pragma unroll
  for (int o = 0; o < D; ++o) {
    hid[o] = fmaxf(q[o], 0.f);
    upd[o] = Wu[L::UPD_B2 + o];
  }
  PHASE();
  matvec10<D, true>(Wu + L::UPD_W2, D, 0, hid, upd);
  float mu = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = fmaf(al, upd[o], x[o]);
    mu += y[o];
  }
  mu *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    float c = y[o] - mu;
    var = fmaf(c, c, var);
  }
  var *= (1.f / D);
  const float rs = 1.f / sqrtf(var + 1e-5f);
  // ---- backward: LayerNorm
  float dyh[D], dy[D];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) {
    y[o] = (y[o] - mu) * rs;  // normalised
    dyh[o] = w[o] * W[L::LN_G + o];
    m1 += dyh[o];
    m2 = fmaf(dyh[o], y[o], m2);
  }
  m1 *= (1.f / D);
  m2 *= (1.f / D);
  float dal = 0.f, dupd[D];
#pragma unroll
  for (int o = 0; o < D; ++o) {
    dy[o] = rs * (dyh[o] - m1 - y[o] * m2);
    dal = fmaf(dy[o], upd[o], dal);
    dupd[o] = al * dy[o];
    g[o] = dy[o];  // accumulates the node-local result; starts with the residual path y = x + ...
  }
  dal *= al * (1.f - al);
  // ---- update MLP and gate
  float dq[D], dmp_to[D], dmp_fr[D];
  PHASE();
  matvecT<D, false>(Wu + L::UPD_W2, D, 0, dupd, dq);
#pragma unroll
  for (int o = 0; o < D; ++o) dq[o] = q[o] > 0.f ? dq[o] : 0.f;
  PHASE();
  matvecT<D, true>(Wu + L::UPD_W1, L::CAT, 0, dq, g);
  PHASE();
  matvecT<D, false>(Wu + L::UPD_W1, L::CAT, D, dq, dmp_to);
  PHASE();
  matvecT<D, false>(Wu + L::UPD_W1, L::CAT, 2 * D, dq, dmp_fr);
#pragma unroll
  for (int k = 0; k < D; ++k) {
    g[k] = fmaf(Wa[k], dal, g[k]);
    dmp_to[k] = fmaf(Wa[D + k], dal, dmp_to[k]);
    dmp_fr[k] = fmaf(Wa[2 * D + k], dal, dmp_fr[k]);
  }
  // ---- second Phi layer
  float dS_to[D], dS_fr[D];
  PHASE();
  matvecT<D, false>(Wto + L::PHI_W2, D, 0, dmp_to, dS_to);
  PHASE();
  matvecT<D, false>(Wfr + L::PHI_W2, D, 0, dmp_fr, dS_fr);
  store10(Bn + 2 * D, dS_to);
  store10(Bn + 3 * D, dS_fr);
  if (MIXED) store10(Bn + 5 * D, zero);
  // ---- target-side projections: sum of the masked cotangents over the node's own edges
  float gt[D], gf[D], dsa[PG ? 64 : 1];
#pragma unroll
  for (int o = 0; o < D; ++o) gt[o] = gf[o] = 0.f;
  if (PG) {
#pragma unroll
    for (int i = 0; i < 64; ++i) dsa[PG ? i : 0] = 0.f;
  }
  PHASE();
  for (int32_t i = ib; i < ie; ++i) {
    const float a0 = csc_attr[3 * (int64_t)i], a1 = csc_attr[3 * (int64_t)i + 1], a2 = csc_attr[3 * (int64_t)i + 2];
    load10(Pj + (int64_t)csc_nbr[i] * NP * D, pj);
    edge_z(Wto + L::PHI_W1, Pt, pj, a0, a1, a2, z);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      const float m = z[o] > 0.f ? dS_to[o] : 0.f;
      gt[o] += m;
      if (PG) {
        dsa[PG ? o * 3 : 0] = fmaf(m, a0, dsa[PG ? o * 3 : 0]);
        dsa[PG ? o * 3 + 1 : 0] = fmaf(m, a1, dsa[PG ? o * 3 + 1 : 0]);
        dsa[PG ? o * 3 + 2 : 0] = fmaf(m, a2, dsa[PG ? o * 3 + 2 : 0]);
      }
    }
  }
  PHASE();
  for (int32_t i = ob; i < oe; ++i) {
    const float a0 = csr_attr[3 * (int64_t)i], a1 = csr_attr[3 * (int64_t)i + 1], a2 = csr_attr[3 * (int64_t)i + 2];
    load10(Pj + (int64_t)csr_nbr[i] * NP * D + D, pj);
    edge_z(Wfr + L::PHI_W1, Pf, pj, a0, a1, a2, z);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      const float m = z[o] > 0.f ? dS_fr[o] : 0.f;
      gf[o] += m;
      if (PG) {
        dsa[PG ? 30 + o * 3 : 0] = fmaf(m, a0, dsa[PG ? 30 + o * 3 : 0]);
        dsa[PG ? 31 + o * 3 : 0] = fmaf(m, a1, dsa[PG ? 31 + o * 3 : 0]);
        dsa[PG ? 32 + o * 3 : 0] = fmaf(m, a2, dsa[PG ? 32 + o * 3 : 0]);
      }
    }
  }
  PHASE();
  matvecT<D, true>(Wto + L::PHI_W1, L::EIN, 0, gt, g);
  PHASE();
  matvecT<D, true>(Wfr + L::PHI_W1, L::EIN, 0, gf, g);
  store10(out + n * D, g);
  if (PG) {
    float* r = rec + n * R::SZ;
    float t[D];
    pg_group(r, x, D, 1.f);
    pg_group(r + 16, mp_to, D, pq[0], pq[1], P > 2 ? pq[P - 1] : 0.f);
    pg_group(r + 2 * 16, mp_fr, D);
    pg_group(r + 3 * 16, S_to, D, (float)(ie - ib));
    pg_group(r + 4 * 16, S_fr, D, (float)(oe - ob));
    pg_group(r + 5 * 16, hid, D, 1.f);
    pg_group(r + 6 * 16, dq, D, dal);
    pg_group(r + 7 * 16, gt, D);
    pg_group(r + 8 * 16, gf, D);
    pg_group(r + 9 * 16, dmp_to, D);
    pg_group(r + 10 * 16, dmp_fr, D);
    pg_group(r + 11 * 16, dupd, D);
#pragma unroll
    for (int o = 0; o < D; ++o) t[o] = w[o] * y[o];
    pg_group(r + 14 * 16, t, D);
    pg_group(r + 15 * 16, w, D);
    dsa[PG ? 60 : 0] = dsa[PG ? 61 : 0] = dsa[PG ? 62 : 0] = dsa[PG ? 63 : 0] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      reinterpret_cast<float4*>(r + 16 * 16)[i] = make_float4(dsa[PG ? 4 * i : 0], dsa[PG ? 4 * i + 1 : 0], dsa[PG ? 4 * i + 2 : 0], dsa[PG ? 4 * i + 3 : 0]);
    if (MIXED) {
      pg_zero(r, 20, 27);
      pg_zero(r, 28, 30);
    }
  }
}

template <int P, bool MIXED, bool PG>
__global__ __launch_bounds__(256) void k_vjp_remote(int64_t N, const float* __restrict__ W, int lofs, int nofs,
                                                    const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_nbr,
                                                    const float* __restrict__ csr_attr, const int32_t* __restrict__ csc_ptr,
                                                    const int32_t* __restrict__ csc_nbr, const float* __restrict__ csc_attr,
                                                    const float* __restrict__ Pj, const float* __restrict__ B,
                                                    float* __restrict__ out, float* __restrict__ rec) {
  using L = WLayout<P>;
  constexpr int NP = MIXED ? 3 : 2;
  constexpr int NB = MIXED ? 6 : 4;
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  const float* Wn = W + nofs;
  float pjt[D], pjf[D], pjn[D], at[D], af[D], an[D], z[D];
  load10(Pj + u * NP * D, pjt);
  load10(Pj + u * NP * D + D, pjf);
  if (MIXED) load10(Pj + u * NP * D + 2 * D, pjn);
#pragma unroll
  for (int o = 0; o < D; ++o) at[o] = af[o] = an[o] = 0.f;
  // u's out-edges (u -> n) are in-edges of n: Phi_to terms of n that read h[u]
  PHASE();
  for (int32_t i = csr_ptr[u]; i < csr_ptr[u + 1]; ++i) {
    const float* Bn = B + (int64_t)csr_nbr[i] * NB * D;
    float pt[D], ds[D];
    load10(Bn, pt);
    load10(Bn + 2 * D, ds);
    edge_z(Wto + L::PHI_W1, pt, pjt, csr_attr[3 * (int64_t)i], csr_attr[3 * (int64_t)i + 1], csr_attr[3 * (int64_t)i + 2], z);
#pragma unroll
    for (int o = 0; o < D; ++o) at[o] += z[o] > 0.f ? ds[o] : 0.f;
  }
  // u's in-edges (n -> u) are out-edges of n: Phi_from (and Phi_neumann) terms of n that read h[u]
  PHASE();
  for (int32_t i = csc_ptr[u]; i < csc_ptr[u + 1]; ++i) {
    const float* Bn = B + (int64_t)csc_nbr[i] * NB * D;
    float pf[D], ds[D];
    const float a0 = csc_attr[3 * (int64_t)i], a1 = csc_attr[3 * (int64_t)i + 1], a2 = csc_attr[3 * (int64_t)i + 2];
    load10(Bn + D, pf);
    load10(Bn + 3 * D, ds);
    edge_z(Wfr + L::PHI_W1, pf, pjf, a0, a1, a2, z);
#pragma unroll
    for (int o = 0; o < D; ++o) af[o] += z[o] > 0.f ? ds[o] : 0.f;
    if (MIXED) {
      load10(Bn + 4 * D, pf);
      load10(Bn + 5 * D, ds);
      edge_z(Wn + L::PHI_W1, pf, pjn, a0, a1, a2, z);
#pragma unroll
      for (int o = 0; o < D; ++o) an[o] += z[o] > 0.f ? ds[o] : 0.f;
    }
  }
  if (PG) {  // neighbour-side cotangent sums: W1j gradients are sum_u acc[u] (x) x[u]
    float* r = rec + u * PgRec<MIXED>::SZ;
    pg_group(r + 12 * 16, at, D);
    pg_group(r + 13 * 16, af, D);
    if (MIXED) pg_group(r + 27 * 16, an, D);
  }
  float g[D];
  load10(out + u * D, g);
  PHASE();
  matvecT<D, true>(Wto + L::PHI_W1, L::EIN, D, at, g);
  PHASE();
  matvecT<D, true>(Wfr + L::PHI_W1, L::EIN, D, af, g);
  if (MIXED) matvecT<D, true>(Wn + L::PHI_W1, L::EIN, D, an, g);
  store10(out + u * D, g);
}

template <int P, bool MIXED, bool PG = false>
static void launch_vjp(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                       const float* w, float* out, float* work, hipStream_t st, float* rec = nullptr) {
  using L = WLayout<P>;
  const int layer = MIXED ? nl - 1 : 0;
  const int lofs = L::layer(layer), nofs = L::phi_neu(nl), unofs = L::upd_neu(nl);
  const unsigned grid = (unsigned)cdiv(p->N, 256);
  float* Pj = work;                            // (N, 20 | 30)
  float* B = work + p->N * (MIXED ? 3 : 2) * D;  // (N, 40 | 60)
  LAUNCH("k_vjp_project", st, (k_vjp_project<P, MIXED><<<grid, 256, 0, st>>>(p->N, W, lofs, nofs, h, Pj)));
  LAUNCH(PG ? "k_pgrad_local" : "k_vjp_local", st, (k_vjp_local<P, MIXED, PG><<<grid, 256, 0, st>>>(
      p->N, W, lofs, nofs, unofs, p->csr_ptr, p->csr_nbr, p->csr_attr, p->csc_ptr, p->csc_nbr, p->csc_attr, p->flags, h, prb,
      nrm, w, Pj, B, out, rec)));
  LAUNCH(PG ? "k_pgrad_remote" : "k_vjp_remote", st, (k_vjp_remote<P, MIXED, PG><<<grid, 256, 0, st>>>(
      p->N, W, lofs, nofs, p->csr_ptr, p->csr_nbr, p->csr_attr, p->csc_ptr, p->csc_nbr, p->csc_attr, Pj, B, out, rec)));
}

// global-gather VJP that also fills the parameter-gradient records (caller's numbering); work: N * 90 floats, rec: N * 320 | 480
int psignn_f_gather_vjp_rec(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm,
                            const float* w, float* out, float* work, float* rec, hipStream_t st) {
  if (p->mixed)
    launch_vjp<3, true, true>(p, W, nl, h, prb, nrm, w, out, work, st, rec);
  else
    launch_vjp<2, false, true>(p, W, nl, h, prb, nrm, w, out, work, st, rec);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

int psignn_f_tile_vjp(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm, const float* w,
                      float* out, float* work, hipStream_t st);

// plan-order VJP: tiled kernels where the plan has tiles (dirichlet, single layer), gather kernels otherwise
extern "C" int psignn_f_vjp_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                              const float* nrm, const float* w, float* out, float* work, void* stream) {
  ARG_CHECK(p && W && h && prb && w && out && work, "NULL argument");
  ARG_CHECK(out != w && out != h, "out must not alias its inputs");
  if (p->tiled && (p->mixed || nl == 1)) return psignn_f_tile_vjp(p, W, nl, h, prb, nrm, w, out, work, (hipStream_t)stream);
  ARG_CHECK(!p->tiled, "plan-order VJP of a tiled multi-layer dirichlet plan is not available");
  return psignn_f_vjp(p, W, nl, h, prb, nrm, w, out, work, stream);
}

extern "C" int psignn_f_vjp(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                            const float* nrm, const float* w, float* out, float* work, void* stream) {
  ARG_CHECK(p && W && h && prb && w && out && work, "NULL argument");
  ARG_CHECK(p->mixed || nl == 1, "VJP of a multi-layer dirichlet block is not implemented");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  ARG_CHECK(out != w && out != h, "out must not alias its inputs");
  hipStream_t st = (hipStream_t)stream;
  KNOB_INT(mixed_tiled, [] { const char* e = getenv("PSIGNN_MIXED_VJP"); return (int)!(e && strcmp(e, "gather") == 0); }());
  if (p->tiled && (p->mixed ? mixed_tiled : nl == 1)) {
    // caller numbering -> plan order -> tiled kernels -> caller numbering
    const int64_t N = p->N;
    const int P = p->mixed ? 3 : 2;
    float* Bw = work;                 // (N, 40)
    float* hp = Bw + N * 4 * D;
    float* wp = hp + N * D;
    float* op = wp + N * D;
    float* pp = op + N * D;           // (N, P)
    float* np = pp + N * 3;           // (N, 2) unit normals of a mixed plan
    int rc;
    if ((rc = psignn_plan_permute(p, h, D, hp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, w, D, wp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, prb, P, pp, 1, stream))) return rc;
    if (p->mixed && (rc = psignn_plan_permute(p, nrm, 2, np, 1, stream))) return rc;
    if ((rc = psignn_f_tile_vjp(p, W, nl, hp, pp, p->mixed ? np : nullptr, wp, op, Bw, st))) return rc;
    return psignn_plan_permute(p, op, D, out, 0, stream);
  }
  if (p->mixed)
    launch_vjp<3, true>(p, W, nl, h, prb, nrm, w, out, work, st);
  else
    launch_vjp<2, false>(p, W, nl, h, prb, nrm, w, out, work, st);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
