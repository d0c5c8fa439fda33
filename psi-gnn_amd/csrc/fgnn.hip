// f_theta: the PSI-GNN message-passing block, and its analytic Jacobian-vector product (gfx950).
//
// Reference semantics: Function.forward (dirichlet/psignn/model.py:279-300; mixed/psignn/model.py:216-245),
// Phi_to / Phi_from (model.py:334-368), MLP (model.py:316-332).
//
// Algebra used (exact up to fp32 re-association):
//   W1 [x_i; x_j; a] + b1 = W1i x_i + W1j x_j + W1a a + b1      -> project h once per node
//   sum_e (W2 z_e + b2)   = W2 (sum_e z_e) + deg * b2           -> second layer once per node
// Kernel 1 (k_project) writes the neighbour-side projections Pj[v] = {W1j_to h_v, W1j_from h_v[, W1j_neu h_v]}.
// Kernel 2 (k_node) owns one node per lane: target-side projection, segment sums over the node's
// CSC (Phi_to) and CSR (Phi_from, Phi_neumann) neighbour lists — no atomics, fixed order,
// bitwise reproducible — then gate, update MLP, LayerNorm and Dirichlet/Neumann row handling.
#include "common.h"

#include "fgnn_common.h"
#include <string.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// Kernel 1: neighbour-side projections.  NPH = number of Phi modules (2 dirichlet, 3 mixed).
// Pj row layout: [to(10) | from(10) | neu(10)]; with JVP the tangent rows follow at +N*NPH*10.
// ------------------------------------------------------------------------------------------
template <int P, bool MIXED, bool JVP>
__global__ __launch_bounds__(256) void k_project(int64_t N, const float* __restrict__ W, int lofs, int nofs,
                                                 const float* __restrict__ h, const int32_t* __restrict__ hsel,
                                                 int64_t hstride, const float* __restrict__ v,
                                                 float* __restrict__ Pj) {
  using L = WLayout<P>;
  constexpr int NPH = MIXED ? 3 : 2;
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (hsel) h += (int64_t)(*hsel) * hstride;  // iterate buffer chosen on the device (solver.hip)
  float x[D], t[D];
  load10(h + n * D, x);
  float* row = Pj + n * (NPH * D);
  matvec10<D, false>(W + lofs + L::L_TO + L::PHI_W1, L::EIN, D, x, t);
  store10(row, t);
  matvec10<D, false>(W + lofs + L::L_FROM + L::PHI_W1, L::EIN, D, x, t);
  store10(row + D, t);
  if (MIXED) {
    matvec10<D, false>(W + nofs + L::PHI_W1, L::EIN, D, x, t);
    store10(row + 2 * D, t);
  }
  if (JVP) {
    load10(v + n * D, x);
    float* drow = Pj + (N + n) * (NPH * D);
    matvec10<D, false>(W + lofs + L::L_TO + L::PHI_W1, L::EIN, D, x, t);
    store10(drow, t);
    matvec10<D, false>(W + lofs + L::L_FROM + L::PHI_W1, L::EIN, D, x, t);
    store10(drow + D, t);
    if (MIXED) {
      matvec10<D, false>(W + nofs + L::PHI_W1, L::EIN, D, x, t);
      store10(drow + 2 * D, t);
    }
  }
}

// Segment sum over one neighbour list of node n for one Phi module.
//   S[o]  = sum_e relu(Pi[o] + Pj[u_e][col..] + W1a a_e)
//   dS[o] = sum_e 1[z>0] (dPi[o] + dPj[u_e][col..])                       (JVP only)
template <int NPH, bool JVP>
__device__ __forceinline__ void seg_sum(int32_t beg, int32_t end, const int32_t* __restrict__ nbr,
                                        const float* __restrict__ attr, const float* __restrict__ W1, int ld,
                                        const float* __restrict__ Pj, int64_t N, int col, const float* Pi,
                                        const float* dPi, float* S, float* dS) {
#pragma unroll
  for (int o = 0; o < D; ++o) {
    S[o] = 0.f;
    if (JVP) dS[o] = 0.f;
  }
  for (int32_t i = beg; i < end; ++i) {
    int64_t u = nbr[i];
    float a0 = attr[3 * (int64_t)i], a1 = attr[3 * (int64_t)i + 1], a2 = attr[3 * (int64_t)i + 2];
    float pj[D], dpj[D];
    load10(Pj + u * (NPH * D) + col, pj);
    if (JVP) load10(Pj + (N + u) * (NPH * D) + col, dpj);
#pragma unroll
    for (int o = 0; o < D; ++o) {
      float z = Pi[o] + pj[o];
      z = fmaf(W1[o * ld + 2 * D + 0], a0, z);
      z = fmaf(W1[o * ld + 2 * D + 1], a1, z);
      z = fmaf(W1[o * ld + 2 * D + 2], a2, z);
      S[o] += fmaxf(z, 0.f);
      if (JVP) dS[o] += z > 0.f ? dPi[o] + dpj[o] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 2: one node per lane.
// mode bits: 1 = apply LayerNorm (last layer)
// ------------------------------------------------------------------------------------------
template <int P, bool MIXED, bool JVP>
__global__ __launch_bounds__(256) void k_node(int64_t N, const float* __restrict__ W, int lofs, int nofs, int unofs,
                                              int apply_ln, const int32_t* __restrict__ csr_ptr,
                                              const int32_t* __restrict__ csr_nbr, const float* __restrict__ csr_attr,
                                              const int32_t* __restrict__ csc_ptr, const int32_t* __restrict__ csc_nbr,
                                              const float* __restrict__ csc_attr, const uint8_t* __restrict__ flags,
                                              const float* __restrict__ h, const int32_t* __restrict__ hsel,
                                              int64_t hstride, const float* __restrict__ h0,
                                              const float* __restrict__ prb, const float* __restrict__ nrm,
                                              const float* __restrict__ v, const float* __restrict__ Pj,
                                              float* __restrict__ out, float* __restrict__ mp_out, int mp_which) {
  using L = WLayout<P>;
  constexpr int NPH = MIXED ? 3 : 2;
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const uint8_t fl = flags[n];
  if (!mp_out && (fl & FLAG_DIRICHLET)) {
    // Dirichlet rows <- h_initial rows (model.py:298); their Jacobian rows are zero.
    float r[D];
    if (JVP) {
#pragma unroll
      for (int o = 0; o < D; ++o) r[o] = 0.f;
    } else {
      load10(h0 + n * D, r);
    }
    store10(out + n * D, r);
    return;
  }
  if (hsel) h += (int64_t)(*hsel) * hstride;
  float x[D], dx[D];
  load10(h + n * D, x);
  if (JVP) load10(v + n * D, dx);

  const float* Wto = W + lofs + L::L_TO;
  const float* Wfr = W + lofs + L::L_FROM;
  float Pi[D], dPi[D], S[D], dS[D];
  float mp_to[D], mp_fr[D], dmp_to[D], dmp_fr[D];

  // ---- Phi_to: aggregate at the column index over the node's in-edges (CSC)
  {
#pragma unroll
    for (int o = 0; o < D; ++o) Pi[o] = Wto[L::PHI_B1 + o];
    matvec10<D, true>(Wto + L::PHI_W1, L::EIN, 0, x, Pi);
    if (JVP) matvec10<D, false>(Wto + L::PHI_W1, L::EIN, 0, dx, dPi);
    int32_t b = csc_ptr[n], e = csc_ptr[n + 1];
    seg_sum<NPH, JVP>(b, e, csc_nbr, csc_attr, Wto + L::PHI_W1, L::EIN, Pj, N, 0, Pi, dPi, S, dS);
    float deg = (float)(e - b);
#pragma unroll
    for (int o = 0; o < D; ++o) mp_to[o] = deg * Wto[L::PHI_B2 + o];
    matvec10<D, true>(Wto + L::PHI_W2, D, 0, S, mp_to);
    if (JVP) matvec10<D, false>(Wto + L::PHI_W2, D, 0, dS, dmp_to);
  }
  // ---- Phi_from: aggregate at the row index over the node's out-edges (CSR)
  const int32_t rb = csr_ptr[n], re = csr_ptr[n + 1];
  {
#pragma unroll
    for (int o = 0; o < D; ++o) Pi[o] = Wfr[L::PHI_B1 + o];
    matvec10<D, true>(Wfr + L::PHI_W1, L::EIN, 0, x, Pi);
    if (JVP) matvec10<D, false>(Wfr + L::PHI_W1, L::EIN, 0, dx, dPi);
    seg_sum<NPH, JVP>(rb, re, csr_nbr, csr_attr, Wfr + L::PHI_W1, L::EIN, Pj, N, D, Pi, dPi, S, dS);
    float deg = (float)(re - rb);
#pragma unroll
    for (int o = 0; o < D; ++o) mp_fr[o] = deg * Wfr[L::PHI_B2 + o];
    matvec10<D, true>(Wfr + L::PHI_W2, D, 0, S, mp_fr);
    if (JVP) matvec10<D, false>(Wfr + L::PHI_W2, D, 0, dS, dmp_fr);
  }
  if (mp_out && mp_which == 0) { store10(mp_out + n * D, mp_to); return; }
  if (mp_out && mp_which == 1) { store10(mp_out + n * D, mp_fr); return; }

  float y[D], dy[D];
  bool neumann = MIXED && (fl & FLAG_NEUMANN);
  if (MIXED && (neumann || mp_out)) {
    // ---- Phi_neumann (Phi_from type) + update_neumann: the row is REPLACED (mixed/psignn/model.py:236,241)
    const float* Wn = W + nofs;
    const float* Un = W + unofs;
    float mp_n[D], dmp_n[D];
#pragma unroll
    for (int o = 0; o < D; ++o) Pi[o] = Wn[L::PHI_B1 + o];
    matvec10<D, true>(Wn + L::PHI_W1, L::EIN, 0, x, Pi);
    if (JVP) matvec10<D, false>(Wn + L::PHI_W1, L::EIN, 0, dx, dPi);
    seg_sum<NPH, JVP>(rb, re, csr_nbr, csr_attr, Wn + L::PHI_W1, L::EIN, Pj, N, 2 * D, Pi, dPi, S, dS);
    float deg = (float)(re - rb);
#pragma unroll
    for (int o = 0; o < D; ++o) mp_n[o] = deg * Wn[L::PHI_B2 + o];
    matvec10<D, true>(Wn + L::PHI_W2, D, 0, S, mp_n);
    if (JVP) matvec10<D, false>(Wn + L::PHI_W2, D, 0, dS, dmp_n);
    if (mp_out) { store10(mp_out + n * D, mp_n); return; }
    // cat_n = [h | mp_neu | prb(3) | normal(2)]
    float hid[D], dhid[D];
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
    if (JVP) {
      matvec10<D, false>(Un + L::NEU_W1, L::NEU_CAT, 0, dx, dhid);
      matvec10<D, true>(Un + L::NEU_W1, L::NEU_CAT, D, dmp_n, dhid);
#pragma unroll
      for (int o = 0; o < D; ++o) dhid[o] = hid[o] > 0.f ? dhid[o] : 0.f;
      matvec10<D, false>(Un + L::NEU_W2, D, 0, dhid, dy);
    }
#pragma unroll
    for (int o = 0; o < D; ++o) {
      hid[o] = fmaxf(hid[o], 0.f);
      y[o] = Un[L::NEU_B2 + o];
    }
    matvec10<D, true>(Un + L::NEU_W2, D, 0, hid, y);
  } else {
    // ---- gate + update MLP on cat = [h | mp_to | mp_from | prb]
    const float* Wu = W + lofs + L::L_UPD;
    const float* Wa = W + L::AL_W;
    float pq[P];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    float al = W[L::AL_B], dal = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[k], x[k], al);
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[D + k], mp_to[k], al);
#pragma unroll
    for (int k = 0; k < D; ++k) al = fmaf(Wa[2 * D + k], mp_fr[k], al);
#pragma unroll
    for (int k = 0; k < P; ++k) al = fmaf(Wa[3 * D + k], pq[k], al);
    al = 1.f / (1.f + expf(-al));
    float hid[D], dhid[D], upd[D];
#pragma unroll
    for (int o = 0; o < D; ++o) hid[o] = Wu[L::UPD_B1 + o];
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 0, x, hid);
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, D, mp_to, hid);
    matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 2 * D, mp_fr, hid);
    matvec10<P, true>(Wu + L::UPD_W1, L::CAT, 3 * D, pq, hid);
    if (JVP) {
#pragma unroll
      for (int k = 0; k < D; ++k) dal = fmaf(Wa[k], dx[k], dal);
#pragma unroll
      for (int k = 0; k < D; ++k) dal = fmaf(Wa[D + k], dmp_to[k], dal);
#pragma unroll
      for (int k = 0; k < D; ++k) dal = fmaf(Wa[2 * D + k], dmp_fr[k], dal);
      dal *= al * (1.f - al);
      matvec10<D, false>(Wu + L::UPD_W1, L::CAT, 0, dx, dhid);
      matvec10<D, true>(Wu + L::UPD_W1, L::CAT, D, dmp_to, dhid);
      matvec10<D, true>(Wu + L::UPD_W1, L::CAT, 2 * D, dmp_fr, dhid);
#pragma unroll
      for (int o = 0; o < D; ++o) dhid[o] = hid[o] > 0.f ? dhid[o] : 0.f;
    }
#pragma unroll
    for (int o = 0; o < D; ++o) {
      hid[o] = fmaxf(hid[o], 0.f);
      upd[o] = Wu[L::UPD_B2 + o];
    }
    matvec10<D, true>(Wu + L::UPD_W2, D, 0, hid, upd);
#pragma unroll
    for (int o = 0; o < D; ++o) y[o] = fmaf(al, upd[o], x[o]);
    if (JVP) {
      float dupd[D];
      matvec10<D, false>(Wu + L::UPD_W2, D, 0, dhid, dupd);
#pragma unroll
      for (int o = 0; o < D; ++o) dy[o] = dx[o] + dal * upd[o] + al * dupd[o];
    }
  }
  // ---- LayerNorm(10), eps 1e-5, biased variance, affine (model.py:293)
  if (apply_ln) {
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
    if (!JVP) {
#pragma unroll
      for (int o = 0; o < D; ++o) y[o] = fmaf((y[o] - mu) * rs, W[L::LN_G + o], W[L::LN_B + o]);
    } else {
      float dm = 0.f, yd = 0.f;
#pragma unroll
      for (int o = 0; o < D; ++o) {
        y[o] = (y[o] - mu) * rs;  // normalised
        dm += dy[o];
        yd = fmaf(y[o], dy[o], yd);
      }
      dm *= (1.f / D);
      yd *= (1.f / D);
#pragma unroll
      for (int o = 0; o < D; ++o) dy[o] = W[L::LN_G + o] * rs * (dy[o] - dm - y[o] * yd);
    }
  }
  store10(out + n * D, JVP ? dy : y);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern "C" int64_t psignn_weights_size(int mixed, int n_layers) {
  if (n_layers < 1) return -1;
  return mixed ? WLayout<3>::total(n_layers, true) : WLayout<2>::total(n_layers, false);
}

extern "C" int64_t psignn_f_workspace_floats(const psignn_plan_t* p) {
  if (!p) return -1;
  // Pj (value + tangent rows, up to 3 Phi modules) + two (N,10) ping-pong buffers for n_layers > 1;
  // the VJP keeps Pj (30) + B (60) rows per node
  return p->N * (10 * D);
}

template <int P, bool MIXED, bool JVP>
static void launch_layer(const psignn_plan* p, const float* W, int n_layers, int layer, int apply_ln,
                         const float* h, const float* h0, const float* prb, const float* nrm, const float* v,
                         float* out, float* work, float* mp_out, int mp_which, hipStream_t st,
                         const int32_t* hsel = nullptr, int64_t hstride = 0) {
  using L = WLayout<P>;
  int lofs = L::layer(layer), nofs = L::phi_neu(n_layers), unofs = L::upd_neu(n_layers);
  unsigned grid = (unsigned)cdiv(p->N, 256);
  LAUNCH(JVP ? "k_project_jvp" : "k_project", st,
         (k_project<P, MIXED, JVP><<<grid, 256, 0, st>>>(p->N, W, lofs, nofs, h, hsel, hstride, v, work)));
  LAUNCH(JVP ? "k_node_jvp" : "k_node", st, (k_node<P, MIXED, JVP><<<grid, 256, 0, st>>>(p->N, W, lofs, nofs, unofs, apply_ln, p->csr_ptr, p->csr_nbr,
                                               p->csr_attr, p->csc_ptr, p->csc_nbr, p->csc_attr, p->flags, h, hsel,
                                               hstride, h0, prb, nrm, v, work, out, mp_out, mp_which)));
}

static int f_args_ok(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                     const float* nrm, const float* out, const float* work) {
  ARG_CHECK(p && W && h && prb && out && work, "NULL argument");
  ARG_CHECK(nl >= 1 && nl <= 64, "n_layers out of range");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  ARG_CHECK(out != h, "out must not alias h");
  return 0;
}

// h = hbase + (*d_sel) * stride when d_sel != NULL (iterate buffer selected by the device-side
// Broyden status, solver.hip); plain hbase otherwise.
int psignn_f_forward_sel(const psignn_plan_t* p, const float* W, int nl, const float* h, const int32_t* hsel,
                         int64_t hstride, const float* h0, const float* prb, const float* nrm, float* out,
                         float* work, hipStream_t st) {
  int rc = f_args_ok(p, W, nl, h, prb, nrm, out, work);
  if (rc) return rc;
  ARG_CHECK(h0 != nullptr, "h_initial is NULL");
  if (p->mixed) {
    // The reference's mixed loop never reassigns h: every layer reads the ORIGINAL h and only the
    // last layer's result is returned (mixed/psignn/model.py:221-245) -> evaluate the last layer only.
    launch_layer<3, true, false>(p, W, nl, nl - 1, 1, h, h0, prb, nrm, nullptr, out, work, nullptr, 0, st, hsel,
                                 hstride);
  } else {
    float* pp[2] = {work + p->N * (2 * 3 * D), work + p->N * (2 * 3 * D + D)};
    const float* cur = h;
    for (int l = 0; l < nl; ++l) {
      float* dst = (l == nl - 1) ? out : pp[l & 1];
      launch_layer<2, false, false>(p, W, nl, l, l == nl - 1, cur, h0, prb, nrm, nullptr, dst, work, nullptr, 0, st,
                                    l == 0 ? hsel : nullptr, hstride);
      cur = dst;
    }
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

int psignn_f_tile_forward(const psignn_plan* p, const float* W, int nl, const float* h, const int32_t* hsel,
                          int64_t hstride, const float* h0, const float* prb, const float* nrm, float* out,
                          float* work, hipStream_t st);

// Evaluation with every node tensor in PLAN order (the solver's internal numbering): tiled kernel when the
// plan has tile structures, global-gather kernels otherwise (then plan order == caller order).
int psignn_f_eval_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const int32_t* hsel,
                    int64_t hstride, const float* h0, const float* prb, const float* nrm, float* out, float* work,
                    hipStream_t st) {
  ARG_CHECK(p != nullptr, "plan is NULL");
  if (p->tiled) return psignn_f_tile_forward(p, W, nl, h, hsel, hstride, h0, prb, nrm, out, work, st);
  return psignn_f_forward_sel(p, W, nl, h, hsel, hstride, h0, prb, nrm, out, work, st);
}

extern "C" int psignn_f_forward_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* h0,
                                  const float* prb, const float* nrm, float* out, float* work, void* stream) {
  ARG_CHECK(out != h, "out must not alias h");
  return psignn_f_eval_p(p, W, nl, h, nullptr, 0, h0, prb, nrm, out, work, (hipStream_t)stream);
}

// n successive applications x <- f(x) in plan order (the core of forward_iteration, utilities/solver.py:301-341,
// without its per-step norms): d_x holds x_0 on entry and x_n on return; d_tmp is a second (N, d) buffer.
extern "C" int psignn_picard_p(const psignn_plan_t* p, const float* W, int nl, float* x, float* tmp, const float* h0,
                               const float* prb, const float* nrm, float* work, int n, void* stream) {
  ARG_CHECK(p && x && tmp && x != tmp && n >= 0, "bad arguments");
  float* cur = x;
  float* nxt = tmp;
  for (int i = 0; i < n; ++i) {
    int rc = psignn_f_eval_p(p, W, nl, cur, nullptr, 0, h0, prb, nrm, nxt, work, (hipStream_t)stream);
    if (rc) return rc;
    float* t = cur; cur = nxt; nxt = t;
  }
  if (cur != x) HIP_TRY(hipMemcpyAsync(x, cur, (size_t)p->N * D * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PSIGNN_OK;
}

extern "C" int psignn_f_forward(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* h0,
                                const float* prb, const float* nrm, float* out, float* work, void* stream) {
  ARG_CHECK(p != nullptr, "plan is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (!p->tiled) return psignn_f_forward_sel(p, W, nl, h, nullptr, 0, h0, prb, nrm, out, work, st);
  // caller numbering -> plan order -> tile kernel -> caller numbering (convenience path; callers that
  // iterate should permute once with psignn_plan_permute and use psignn_f_forward_p)
  int rc = f_args_ok(p, W, nl, h, prb, nrm, out, work);
  if (rc) return rc;
  ARG_CHECK(h0 != nullptr, "h_initial is NULL");
  const int64_t N = p->N;
  const int P = p->mixed ? 3 : 2;
  float* hp = work;
  float* h0p = hp + N * D;
  float* outp = h0p + N * D;
  float* prbp = outp + N * D;
  float* nrmp = prbp + N * 3;
  float* rest = nrmp + N * 2;
  if ((rc = psignn_plan_permute(p, h, D, hp, 1, stream))) return rc;
  if ((rc = psignn_plan_permute(p, h0, D, h0p, 1, stream))) return rc;
  if ((rc = psignn_plan_permute(p, prb, P, prbp, 1, stream))) return rc;
  if (p->mixed && (rc = psignn_plan_permute(p, nrm, 2, nrmp, 1, stream))) return rc;
  if ((rc = psignn_f_tile_forward(p, W, nl, hp, nullptr, 0, h0p, prbp, p->mixed ? nrmp : nullptr, outp, rest, st))) return rc;
  return psignn_plan_permute(p, outp, D, out, 0, stream);
}

extern "C" int psignn_phi(const psignn_plan_t* p, const float* W, int nl, int layer, int which, const float* h,
                          float* out, float* work, void* stream) {
  ARG_CHECK(p && W && h && out && work, "NULL argument");
  ARG_CHECK(layer >= 0 && layer < nl, "layer out of range");
  ARG_CHECK(which >= 0 && which <= (p->mixed ? 2 : 1), "which out of range");
  hipStream_t st = (hipStream_t)stream;
  // prb / normals / h0 are not read on the mp_out path
  if (p->mixed)
    launch_layer<3, true, false>(p, W, nl, layer, 0, h, h, h, h, nullptr, out, work, out, which, st);
  else
    launch_layer<2, false, false>(p, W, nl, layer, 0, h, h, h, h, nullptr, out, work, out, which, st);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

int psignn_f_tile_jvp(const psignn_plan* p, const float* W, int nl, const float* h, const float* prb, const float* nrm, const float* v,
                      float* out, hipStream_t st);

// plan-order JVP (tiled single-layer dirichlet plans): the form a Krylov solver that keeps its vectors in plan order uses
extern "C" int psignn_f_jvp_p(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                              const float* nrm, const float* v, float* out, void* stream) {
  ARG_CHECK(p && W && h && prb && v && out, "NULL argument");
  ARG_CHECK(out != v && out != h, "out must not alias its inputs");
  return psignn_f_tile_jvp(p, W, nl, h, prb, nrm, v, out, (hipStream_t)stream);
}

extern "C" int psignn_f_jvp(const psignn_plan_t* p, const float* W, int nl, const float* h, const float* prb,
                            const float* nrm, const float* v, float* out, float* work, void* stream) {
  int rc = f_args_ok(p, W, nl, h, prb, nrm, out, work);
  if (rc) return rc;
  ARG_CHECK(v != nullptr && out != v, "v is NULL or aliases out");
  ARG_CHECK(p->mixed || nl == 1, "JVP of a multi-layer dirichlet block is not implemented");
  hipStream_t st = (hipStream_t)stream;
  KNOB_INT(mixed_tiled, [] { const char* e = getenv("PSIGNN_MIXED_JVP"); return (int)!(e && strcmp(e, "gather") == 0); }());
  if (p->tiled && (p->mixed ? mixed_tiled : nl == 1)) {  // caller numbering -> plan order -> tiled kernel -> caller numbering
    const int64_t N = p->N;
    const int P = p->mixed ? 3 : 2;
    float* hp = work;
    float* vp = hp + N * D;
    float* op = vp + N * D;
    float* pp = op + N * D;  // (N, P)
    float* np = pp + N * 3;  // (N, 2) unit normals of a mixed plan
    if ((rc = psignn_plan_permute(p, h, D, hp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, v, D, vp, 1, stream))) return rc;
    if ((rc = psignn_plan_permute(p, prb, P, pp, 1, stream))) return rc;
    if (p->mixed && (rc = psignn_plan_permute(p, nrm, 2, np, 1, stream))) return rc;
    if ((rc = psignn_f_tile_jvp(p, W, nl, hp, pp, p->mixed ? np : nullptr, vp, op, st))) return rc;
    return psignn_plan_permute(p, op, D, out, 0, stream);
  }
  if (p->mixed)
    launch_layer<3, true, true>(p, W, nl, nl - 1, 1, h, h, prb, nrm, v, out, work, nullptr, 0, st);
  else
    launch_layer<2, false, true>(p, W, nl, 0, 1, h, h, prb, nrm, v, out, work, nullptr, 0, st);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
