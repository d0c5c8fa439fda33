// f_theta on the tiled plan: one workgroup per mesh tile, neighbour projections staged in LDS (gfx950).
//
// Same function as fgnn.hip (reference: dirichlet/psignn/model.py:279-300, mixed/psignn/model.py:216-245),
// different data movement:
//   stage 1  every lane takes one row of the tile + halo (<= 256 + HALO_CAP rows): loads h (tile rows are
//            one coalesced 40 B/lane stream, halo rows a short indexed gather), projects it with the
//            neighbour-side weights W1j_{to,from[,neu]} and parks the 80-byte result row in LDS
//   stage 2  every lane owns one tile node and walks its pair-merged ELL slots (tiles.hip): one coalesced
//            16-byte load {LDS row, IN/OUT, edge_attr} and one LDS row read (5 x ds_read_b128) serve BOTH
//            directions of a neighbour; relu terms are summed in the plan's canonical order (no atomics,
//            bitwise reproducible), then gate / update MLP / LayerNorm / boundary rows.
// The second Phi layer (W2 S + deg b2) is linear and is folded into the consumers' first-layer weights on
// the host (WLayout fold block).  All node tensors are in PLAN order; the solver keeps its state there.
// weight loads of mv2 pinned chunk by chunk (tile_helpers.h; A/B in profiles/r3_ab_mv2.txt: fused step 75.8 -> 73.4 us, plain f unchanged)
#ifndef MV2_LAUNDER
#define MV2_LAUNDER 2
#endif
#ifndef MV2_CH
#define MV2_CH 10
#endif
#include "tile_helpers.h"
#ifndef TILE_WPE
#define TILE_WPE 0    // > 0: __attribute__((amdgpu_waves_per_eu(TILE_WPE, TILE_WPE))) on k_f_tile: the register allocator is held to 96 VGPRs (5 waves per SIMD)
#endif
#ifndef FUSED_WPE
#define FUSED_WPE 0   // > 0: the FUSED instantiations only are held to this many waves per SIMD (6 = 80 VGPRs; they need 83 - 86)
#endif
#if FUSED_WPE
#define TILE_WPE_ATTR __attribute__((amdgpu_waves_per_eu(FUSED ? FUSED_WPE : 1, FUSED ? FUSED_WPE : 8)))
#elif TILE_WPE
#define TILE_WPE_ATTR __attribute__((amdgpu_waves_per_eu(TILE_WPE, TILE_WPE)))
#else
#define TILE_WPE_ATTR
#endif
#ifndef X_RELOAD
#define X_RELOAD 1    // 1: park the node state in memory during the slot walk (A/B: scripts/ab_edge.sh)
#endif
#ifndef EDGE_CLAMP
#define EDGE_CLAMP 1  // relu folded into the clamp bit of the last edge fma (tile_helpers.h: edge_pass_both_clamp)
#endif
#ifndef EDGE_BOTH
#define EDGE_BOTH 1   // both edge directions in one slot walk (edge_pass_both); 0: one walk per direction.  A/B on one
                      // box, 1M nodes: plain f 61.5 vs 62.9 us, fused Broyden step 99.5 vs 104 us
#endif

template <bool MIXED>
struct TileRow {
  static constexpr int RS = MIXED ? 32 : 20;  // floats per LDS row: [to 10 | from 10] (| neu 10 | pad 2)
};

__device__ __forceinline__ void lds_store20(float* __restrict__ p, const float* __restrict__ a, const float* __restrict__ b) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(a[0], a[1], a[2], a[3]);
  q[1] = make_float4(a[4], a[5], a[6], a[7]);
  q[2] = make_float4(a[8], a[9], b[0], b[1]);
  q[3] = make_float4(b[2], b[3], b[4], b[5]);
  q[4] = make_float4(b[6], b[7], b[8], b[9]);
}
__device__ __forceinline__ void lds_load20(const float* __restrict__ p, float* __restrict__ a, float* __restrict__ b) {
  const float4* q = reinterpret_cast<const float4*>(p);
  float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];
  a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
  a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
  a[8] = v2.x; a[9] = v2.y; b[0] = v2.z; b[1] = v2.w;
  b[2] = v3.x; b[3] = v3.y; b[4] = v3.z; b[5] = v3.w;
  b[6] = v4.x; b[7] = v4.y; b[8] = v4.z; b[9] = v4.w;
}
__device__ __forceinline__ void lds_store10at(float* __restrict__ p, const float* __restrict__ t) {  // 8-byte aligned
  float2* q = reinterpret_cast<float2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = make_float2(t[2 * i], t[2 * i + 1]);
}
__device__ __forceinline__ void lds_load10at(const float* __restrict__ p, float* __restrict__ t) {
  const float2* q = reinterpret_cast<const float2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float2 v = q[i];
    t[2 * i] = v.x;
    t[2 * i + 1] = v.y;
  }
}

// Stage 1 exists in two forms, chosen per launch (env PSIGNN_STAGE1 = mfma | valu overrides the default):
// matrix cores (v_mfma_f32_16x16x4_f32) or packed VALU.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
// Default form: packed VALU everywhere.  A/Bs on one box (1M-node mesh) after the scalar-load phase barriers removed the
// SGPR spills of the VALU form: dirichlet plain f 64.6 us (valu) vs 67.7 us (mfma); fused Broyden step 102 us (valu) vs
// 122-125 us (mfma: it has to re-load x and update for stage 2, which the VALU form gets for free from its stage-1
// registers); mixed plain f 102.9 (valu) vs 102.2 (mfma) in one launch, 81.8 us (valu) once the tiles without Neumann
// nodes run with 80-byte LDS rows (launch_mixed).  Before the barriers the VALU form spilled ~200 SGPRs and lost: 68.2 vs
// 66.3 us.  The MFMA form stays selectable (PSIGNN_STAGE1=mfma) for A/B runs.
static int stage1_mfma(bool fused, bool mixed) {
  KNOB_INT(forced, [] {
    const char* e = getenv("PSIGNN_STAGE1");
    return !e ? -1 : (strcmp(e, "mfma") == 0 ? 1 : 0);
  }());
  if (forced >= 0) return forced;
  (void)fused;
  (void)mixed;
  return 0;   // the MFMA form is kept for A/B runs (PSIGNN_STAGE1=mfma); see the comment above
}

// Broyden fusion (solver.hip): the kernel forms x_next = x_cur + update while loading, and its epilogue
// writes x_next, g_new = f(x_next) - x_next, dg = g_new - g_old and the per-wave partials of |g_new|^2, |f|^2
// -- the work of k_xnext and k_resid without their extra passes over the state vectors.
struct FuseArgs {
  const float* upd;   // update vector, plan order
  float* gnew;        // out: g = f(x_next) - x_next  (g of the previous iterate stays in the solver's other g buffer: the
                      // update kernels form dg = g_new - g_old themselves; round 2 read g_old and wrote dg here, +80 MB/step at 1M nodes)
  float* xbuf;        // iterate buffers (stride M)
  const int32_t* st;  // device status block (int32 view)
  int off_done, off_cur, off_nxt;
  int64_t M;
  float* part;
  int npart;
  long long* stamps;  // diagnostics (psignn_prof_tile_stamps): per tile and wave 8 constant-rate (100 MHz) clock stamps, else NULL
};

// Wave priority per phase: TILE_PRIO = p0 + 4 p1 + 16 p2 + 64 p3 sets s_setprio(p) at the start of stage 1 / after the barrier (slot
// walk) / at the node update / at the epilogue; 0 = never touched.  15: the phases that ISSUE memory requests (stage 1, slot walk)
// go ahead of the purely arithmetic node update of other waves -- plain f 52.0 -> 50.4 us over four interleaved runs each, fused
// step unchanged (profiles/r3_ab_prio.txt); the other placements measured there are within noise.
#ifndef TILE_PRIO
#define TILE_PRIO 15
#endif
#define PRIO_AT(ph) do { if (TILE_PRIO) __builtin_amdgcn_s_setprio((TILE_PRIO >> (2 * (ph))) & 3); } while (0)
// In-kernel phase stamps (diagnostics only; one lane per wave writes s_memtime values)
#ifndef TILE_STAMPS
#define TILE_STAMPS 0   // build with -DTILE_STAMPS=1 for scripts/tile_phases.py (the stamps cost registers and issue slots)
#endif
#if !TILE_STAMPS
#define STAMP(i) do { } while (0)
#else
#define STAMP(i)                                                                                        \
  do {                                                                                                  \
    if (fa.stamps && (threadIdx.x & 63) == 0)                                                           \
      fa.stamps[((int64_t)tile * 4 + (threadIdx.x >> 6)) * 8 + (i)] = (long long)wall_clock64(); \
  } while (0)
#endif
static long long* g_tile_stamps = nullptr;
extern "C" void psignn_prof_tile_stamps(void* d_buf) { g_tile_stamps = (long long*)d_buf; }

// wave sum of the fused epilogue's norm partials: DPP adds inside the rows of 16, row sums through v_readlane (vec_helpers.h)
template <int CTRL>
__device__ __forceinline__ float dpp_perm_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_perm_f<0xB1>(v);
  v += dpp_perm_f<0x4E>(v);
  v += dpp_perm_f<0x141>(v);
  v += dpp_perm_f<0x140>(v);
  const int vi = __builtin_bit_cast(int, v);   // (the builtin is typed int: a float argument would be converted by value)
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}

// One tile: body of k_f_tile.  `slot` = position in the launch's tile list.  Every thread of the workgroup reaches the
// stage-1 barrier; returns happen after it (the caller's loop barrier is outside).
template <int P, bool MIXED, bool FUSED, bool MFMA1>
__device__ __forceinline__ void f_tile_body(const FuseArgs& fa, const int slot, const int32_t* __restrict__ tile_list,
                                            const TileCtx* __restrict__ C, const float* __restrict__ W, int lofs, int tofs,
                                            int tnofs, int apply_ln, const float* __restrict__ h,
                                            const int32_t* __restrict__ hsel, int64_t hstride,
                                            const float* __restrict__ h0, const float* __restrict__ prb,
                                            const float* __restrict__ nrm, float* __restrict__ out, float* __restrict__ lds) {
  using L = WLayout<P>;
  constexpr int RS = TileRow<MIXED>::RS;
  const int tile = tile_list ? tile_list[slot] : slot;   // mixed plans: a sub-list of the tiles (see launch_mixed)
  STAMP(0);
  PRIO_AT(0);
  const int tid = threadIdx.x;
  // tile -> node range: arithmetic when the plan's tiles are uniform chunks (always, since round 2; the table stays for
  // tile sizes that are not a multiple of 64)
  const int tn = C->tile_nodes;
  const int32_t t0 = tn ? tile * tn : C->tile_ptr[tile];
  const int n_t = tn ? min(tn, C->n_nodes - t0) : C->tile_ptr[tile + 1] - t0;
  const int n_h = C->halo_cnt[tile];
  const int32_t* hl = C->halo + (int64_t)tile * HALO_CAP;
  if (hsel) h += (int64_t)(*hsel) * hstride;
  if (FUSED) h = fa.xbuf + (int64_t)fa.st[fa.off_cur] * fa.M;

  const float* T = W + tofs;    // transposed section of this layer
  const float* TN = W + tnofs;  // transposed Neumann blocks (mixed)

  // ---- stage 1: neighbour-side projections of tile + halo rows -> LDS
  float x[D];
  uint4 slot0 = make_uint4(ELL_EMPTY, 0u, 0u, 0u);   // slot row 0 of this lane / prefetch witness (SLOT_PREFETCH below)
  unsigned slot_touch = 0;
  (void)slot0;
  (void)slot_touch;
  if constexpr (MFMA1) {
    // Dense node-feature x weight product on the matrix cores: out[row][o] = sum_k x[row][k] W1j[o][k] as
    // v_mfma_f32_16x16x4_f32 tiles with the WEIGHTS as the A operand (A[i = output][k]) and the node rows as B
    // (B[k][j = row]): a lane then receives D[i = 4 (lane>>4) + r][j = lane&15], r = 0..3 -- four consecutive
    // outputs of ONE row -- which is a single 16-byte LDS store.  The f32 MFMA accumulates k in order as an fma
    // chain starting from 0 (MI355X guide, 'FP32-input MFMA'), i.e. the same sum as the VALU form.  K = 10 is
    // padded to 12 (three k-steps), the 20 / 30 outputs to 32 (two M-tiles); 16 rows per N-tile, N-tiles dealt
    // round-robin to the 4 waves.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NOUT = MIXED ? 30 : 20;
    const int lane = tid & 63, g = lane >> 4, c = lane & 15, wave = tid >> 6;
    float wa[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int o = 16 * mt + c, k = 4 * s + g;
        float v = 0.f;
        if (k < D && o < NOUT) {
          const float* blk = o < D ? T + L::T_W1J_TO : (o < 2 * D ? T + L::T_W1J_FR : TN + L::N_W1J);
          v = blk[k * D + (o % D)];
        }
        wa[mt][s] = v;
      }
    const int rows = n_t + n_h;
    for (int nt = wave; nt * 16 < rows; nt += TILE_THREADS / 64) {
      const int row = nt * 16 + c;
      const bool ok = row < rows;
      const int64_t node = !ok ? (int64_t)t0 : (row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t]);
      float xb[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int k = 4 * s + g;
        float v = 0.f;
        if (ok && k < D) {
          v = h[node * D + k];
          if (FUSED) v += fa.upd[node * D + k];  // x_next = x_cur + update
        }
        xb[s] = v;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][s], xb[s], acc, 0, 0, 0);
        const int o0 = 16 * mt + 4 * g;
        if (ok && o0 < (MIXED ? RS : NOUT))
          *reinterpret_cast<float4*>(lds + row * RS + o0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
    if (tid < n_t) {  // this lane's own node state for stage 2 (L1/L2-hot: the rows were just read above)
      load10(h + (int64_t)(t0 + tid) * D, x);
      if (FUSED) {
        float ur[D];
        load10(fa.upd + (int64_t)(t0 + tid) * D, ur);
#pragma unroll
        for (int o = 0; o < D; ++o) x[o] += ur[o];
      }
    }
  } else {
  // mixed family: the Phi_neumann projections are only read by Neumann lanes, i.e. in the few boundary tiles -- every
  // other tile skips a third of its stage-1 work
  bool tile_neu = false;
  if (MIXED) tile_neu = __syncthreads_or(tid < n_t ? (C->flags_p[t0 + tid] & FLAG_NEUMANN) : 0) != 0;
#ifndef HALO_SPLIT
#define HALO_SPLIT 1   // 1: halo rows of stage 1 shared out as half rows over all four waves (see below); 0: round 1's loop
#endif
#ifndef SLOT_PREFETCH
#define SLOT_PREFETCH 0   // 0 (default): stage 2 requests its slot rows itself; 2: flag byte and slot row 0 requested before the
                          // stage-1 barrier (measured: plain f 53.6 vs 53.2 us, fused step 95.0 vs 90.1 us -- no gain); 1: the wave's slot rows are requested before stage 1 (row 0 kept, the rest pulled towards L2).
                          // Measured (1M nodes, plain f): 64.7 - 65.2 us with, 57.5 - 58.2 us without -- the extra pass over the
                          // slot records costs more than the walk's misses; kept for A/B runs
#endif
#if HALO_SPLIT
#if SLOT_PREFETCH == 1
  // (experiment, off) every wave requests ALL its slot rows ahead of the h rows: row 0 stays in registers for the walk, the
  // others are only pulled towards L2 / L1 (their first words are folded into a value that is looked at once and never acted
  // on).  Measured: 64.7 - 65.2 us vs 57.5 - 58.2 us without -- the second pass over the slot records costs more than the
  // walk's misses.
  if ((tid & ~63) < n_t) {
    const int pslice = (tn ? tile * (tn >> 6) : C->tile_slice[tile]) + (tid >> 6);
    const uint4* pslots = C->ell + (int64_t)C->slice_off[pslice] * 64 + (tid & 63);
    const int pn = C->slice_deg[pslice];
    if (pn > 0) {   // (the fused Broyden step has no four registers to spare across stage 1: it only touches row 0 as well)
      if (FUSED) slot_touch ^= pslots[0].x;
      else slot0 = pslots[0];
    }
#pragma unroll
    for (int r = 1; r < 8; ++r)
      if (r < pn) slot_touch ^= pslots[(int64_t)r * 64].x;
  }
#endif
  // own row first: every lane's loads are in flight before any projection starts
  const int32_t hidx_w = (tid >> 6 & 1) * 64 + (tid & 63);   // this lane's halo slot inside a 128-row batch
  int32_t hnode = 0;
  if (hidx_w < n_h) hnode = hl[hidx_w];                        // halo index of the first batch, ahead of its use
  float xr[D], xh[D];
  if (tid < n_t) load10(h + (int64_t)(t0 + tid) * D, xr);
  // the first batch's halo row is requested before the own row is projected (the phase barriers below would otherwise keep
  // its load behind that arithmetic): one more memory round trip off stage 1's critical path
  if (hidx_w < n_h) {
    load10(h + (int64_t)hnode * D, xh);
    if (FUSED) {
      float uh[D];
      load10(fa.upd + (int64_t)hnode * D, uh);
#pragma unroll
      for (int o = 0; o < D; ++o) xh[o] += uh[o];
    }
  }
  if (tid < n_t) {
    if (FUSED) {  // x_next = x_cur + update (line_search with on=False: step 1, solver.py:85-94)
      float ur[D];
      load10(fa.upd + (int64_t)(t0 + tid) * D, ur);
#pragma unroll
      for (int o = 0; o < D; ++o) xr[o] += ur[o];
    }
#pragma unroll
    for (int o = 0; o < D; ++o) x[o] = xr[o];
#if SLOT_PREFETCH
    // the prefetch loads were issued before the h row and return in order: looking at their witness here costs no extra
    // wait and frees its register before the projections (never true: slot words are < 2^18)
    if (slot_touch == 0xDEADBEEFu) xr[0] = 0.f;
#endif
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
    if (MIXED && tile_neu) {
#pragma unroll
      for (int p = 0; p < 5; ++p) ta[p] = splat(0.f);
      PHASE();
      mv2<D>(TN + L::N_W1J, xr, ta);
      q[5] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
      q[6] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
      reinterpret_cast<float2*>(q + 7)[0] = make_float2(ta[4].x, ta[4].y);
    }
  }
  // Halo rows.  A tile of 256 nodes has ~110 of them: as whole rows they are a second round for waves 0 and 1 only, with
  // the other two waves parked at the barrier (stamps: 4.0 vs 2.5 us in stage 1, 1.7 us of barrier wait).  Shared out as
  // HALF rows instead -- waves 0, 1 project the Phi_to half of halo rows [64 (w & 1), +64), waves 2, 3 the Phi_from half --
  // every wave does one 10 x 10 block per 128 halo rows: the same number of wave instructions, half the critical path.
  {
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);   // wave-uniform: 0 Phi_to columns, 1 Phi_from columns
    for (int hb = 0; hb < n_h; hb += 128) {
      const int idx = hb + hidx_w;
      if (hb > 0 && idx < n_h) hnode = hl[idx];
      if (idx < n_h) {
        float xr[D];
        if (hb == 0) {
#pragma unroll
          for (int o = 0; o < D; ++o) xr[o] = xh[o];
        } else {
          load10(h + (int64_t)hnode * D, xr);
          if (FUSED) {
            float ur[D];
            load10(fa.upd + (int64_t)hnode * D, ur);
#pragma unroll
            for (int o = 0; o < D; ++o) xr[o] += ur[o];
          }
        }
        v2f ta[5];
#pragma unroll
        for (int p = 0; p < 5; ++p) ta[p] = splat(0.f);
        PHASE();
        float* rowp = lds + (n_t + idx) * RS;
        if (half == 0) {
          mv2<D>(T + L::T_W1J_TO, xr, ta);
          float4* q = reinterpret_cast<float4*>(rowp);                 // floats 0..9: b128, b128, b64
          q[0] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
          q[1] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
          reinterpret_cast<float2*>(rowp + 8)[0] = make_float2(ta[4].x, ta[4].y);
        } else {
          mv2<D>(T + L::T_W1J_FR, xr, ta);
          reinterpret_cast<float2*>(rowp + 10)[0] = make_float2(ta[0].x, ta[0].y);   // floats 10..19: b64, b128, b128
          float4* q = reinterpret_cast<float4*>(rowp + 12);
          q[0] = make_float4(ta[1].x, ta[1].y, ta[2].x, ta[2].y);
          q[1] = make_float4(ta[3].x, ta[3].y, ta[4].x, ta[4].y);
        }
      }
    }
    if (MIXED && tile_neu) {   // boundary tiles of the mixed family: Phi_neumann columns of the halo rows
      for (int idx = tid; idx < n_h; idx += TILE_THREADS) {
        const int64_t node = hl[idx];
        float xr[D];
        load10(h + node * D, xr);
        if (FUSED) {
          float ur[D];
          load10(fa.upd + node * D, ur);
#pragma unroll
          for (int o = 0; o < D; ++o) xr[o] += ur[o];
        }
        v2f ta[5];
#pragma unroll
        for (int p = 0; p < 5; ++p) ta[p] = splat(0.f);
        PHASE();
        mv2<D>(TN + L::N_W1J, xr, ta);
        float4* q = reinterpret_cast<float4*>(lds + (n_t + idx) * RS);
        q[5] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
        q[6] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
        reinterpret_cast<float2*>(q + 7)[0] = make_float2(ta[4].x, ta[4].y);
      }
    }
  }
#else
  for (int row = tid; row < n_t + n_h; row += TILE_THREADS) {
    const int64_t node = row < n_t ? (int64_t)(t0 + row) : (int64_t)hl[row - n_t];
    float xr[D];
    load10(h + node * D, xr);
    if (FUSED) {  // x_next = x_cur + update (line_search with on=False: step 1, solver.py:85-94)
      float ur[D];
      load10(fa.upd + node * D, ur);
#pragma unroll
      for (int o = 0; o < D; ++o) xr[o] += ur[o];
    }
    if (row == tid) {
#pragma unroll
      for (int o = 0; o < D; ++o) x[o] = xr[o];
    }
    v2f ta[5], tb[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) ta[p] = tb[p] = splat(0.f);
    PHASE();
    mv2<D>(T + L::T_W1J_TO, xr, ta);
    PHASE();
    mv2<D>(T + L::T_W1J_FR, xr, tb);
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
      mv2<D>(TN + L::N_W1J, xr, ta);
      q[5] = make_float4(ta[0].x, ta[0].y, ta[1].x, ta[1].y);
      q[6] = make_float4(ta[2].x, ta[2].y, ta[3].x, ta[3].y);
      reinterpret_cast<float2*>(q + 7)[0] = make_float2(ta[4].x, ta[4].y);
    }
  }
#endif
  }
  // stage 2's first loads -- the node's flag byte and slot row 0 -- are requested BEFORE the barrier: both sit on the
  // critical path right behind it (flag -> Dirichlet branch -> slot pointer -> first record), an HBM round trip each
  const bool active = tid < n_t;
  const int64_t n = (int64_t)t0 + (active ? tid : 0);
  const uint8_t fl = C->flags_p[n];
#if HALO_SPLIT && SLOT_PREFETCH == 2
  if (!MFMA1 && active) {
    const int pslice = (tn ? tile * (tn >> 6) : C->tile_slice[tile]) + __builtin_amdgcn_readfirstlane(tid >> 6);
    if (C->slice_deg[pslice] > 0) slot0 = C->ell[(int64_t)C->slice_off[pslice] * 64 + (tid & 63)];
  }
#endif
  STAMP(1);
  __syncthreads();
  STAMP(2);
  PRIO_AT(1);
  if (!FUSED && tid >= n_t) return;

  // ---- stage 2: one tile node per lane
  float y[D];
  const bool dirichlet = fl & FLAG_DIRICHLET;
  if (dirichlet) {  // Dirichlet rows <- h_initial rows (model.py:298)
    load10(h0 + n * D, y);
    if (!FUSED) {
      store10(out + n * D, y);
      return;
    }
  }
  if (!dirichlet && active) {
  const int lane = tid & 63;
  // the wave index is wave-uniform: saying so keeps the slice's slot count and the whole slot-loop control in scalar registers
  const int slice = (tn ? tile * (tn >> 6) : C->tile_slice[tile]) + __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint4* slots = C->ell + (int64_t)C->slice_off[slice] * 64 + lane;
  const int nslots = C->slice_deg[slice];

  // target-side projection (bias included) + neighbour sum
  v2f Pi[5], S_to[5], S_fr[5];
  float deg_in, deg_out;
#pragma unroll
  for (int p = 0; p < 5; ++p) S_to[p] = S_fr[p] = splat(0.f);
#if X_RELOAD
  // the node state is not needed during the slot walk: park it in memory (the fused step writes x_next to its slot of the
  // iterate buffer anyway) and read it back afterwards -- ten VGPRs less in the loop
  const float* xsrc = h + n * D;
  if (FUSED) {
    float* xn = fa.xbuf + (int64_t)fa.st[fa.off_nxt] * fa.M + n * D;
    store10(xn, x);
    xsrc = xn;
  }
#endif
#if EDGE_BOTH
  {
    v2f Pi2[5];
    ld5(T + L::T_B1_TO, Pi);
    PHASE();
    mv2<D>(T + L::T_W1I_TO, x, Pi);
    ld5(T + L::T_B1_FR, Pi2);
    PHASE();
    mv2<D>(T + L::T_W1I_FR, x, Pi2);
    PHASE();
#if EDGE_CLAMP
#if HALO_SPLIT && SLOT_PREFETCH
    edge_pass_both_clamp<RS>(slots, nslots, lds, T + L::T_A_TO, T + L::T_A_FR, Pi, Pi2, S_to, S_fr, deg_in, deg_out,
                             (MFMA1 || (FUSED && SLOT_PREFETCH == 1)) ? nullptr : &slot0);
#else
    edge_pass_both_clamp<RS>(slots, nslots, lds, T + L::T_A_TO, T + L::T_A_FR, Pi, Pi2, S_to, S_fr, deg_in, deg_out);
#endif
#else
    edge_pass_both<RS>(slots, nslots, lds, T + L::T_A_TO, T + L::T_A_FR, Pi, Pi2, S_to, S_fr, deg_in, deg_out);
#endif
    PHASE();
  }
  STAMP(3);
  PRIO_AT(2);
#if X_RELOAD
  load10(xsrc, x);
#endif
#else
  ld5(T + L::T_B1_TO, Pi);
  PHASE();
  mv2<D>(T + L::T_W1I_TO, x, Pi);
  deg_in = edge_pass<RS, 0, SLOT_IN>(slots, nslots, lds, T + L::T_A_TO, Pi, S_to);
  ld5(T + L::T_B1_FR, Pi);
  PHASE();
  mv2<D>(T + L::T_W1I_FR, x, Pi);
  deg_out = edge_pass<RS, D, SLOT_OUT>(slots, nslots, lds, T + L::T_A_FR, Pi, S_fr);
#endif

  v2f y2[5];
  if (MIXED && (fl & FLAG_NEUMANN)) {
    // Phi_neumann (Phi_from type: out-edges) + update_neumann: the row is REPLACED (mixed/psignn/model.py:236,241)
    v2f S_n[5], hid[5], gN[5];
    ld5(TN + L::N_B1, Pi);
#pragma unroll
    for (int p = 0; p < 5; ++p) S_n[p] = splat(0.f);
    PHASE();
    mv2<D>(TN + L::N_W1I, x, Pi);
    edge_pass<RS, 2 * D, SLOT_OUT>(slots, nslots, lds, TN + L::N_A, Pi, S_n);
    ld5(TN + L::N_NB1, hid);
    ld5(TN + L::N_gN, gN);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid[p] = __builtin_elementwise_fma(splat(deg_out), gN[p], hid[p]);
    PHASE();
    mv2<D>(TN + L::N_N1H, x, hid);
    PHASE();
    mv2<D>(TN + L::N_GN, reinterpret_cast<const float*>(S_n), hid);
    float pq[P + 2];
#pragma unroll
    for (int k = 0; k < P; ++k) pq[k] = prb[n * P + k];
    pq[P] = nrm[n * 2];
    pq[P + 1] = nrm[n * 2 + 1];
    PHASE();
    mv2<P + 2>(TN + L::N_N1P, pq, hid);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid[p] = __builtin_elementwise_max(hid[p], splat(0.f));
    ld5(TN + L::N_NB2, y2);
    PHASE();
    mv2<D>(TN + L::N_N2, reinterpret_cast<const float*>(hid), y2);
  } else {
    // gate + update MLP on cat = [h | mp_to | mp_from | prb], with mp_* = W2 S + deg b2 folded in
    const float* Wf = W + lofs + L::L_FOLD;  // this layer's fold vectors of the (shared) alpha gate
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
    v2f hid[5], g1[5], g2[5], upd[5];
    ld5(T + L::T_HB, hid);
    ld5(T + L::T_gTO, g1);
    ld5(T + L::T_gFR, g2);
#pragma unroll
    for (int p = 0; p < 5; ++p)
      hid[p] = __builtin_elementwise_fma(splat(deg_in), g1[p], __builtin_elementwise_fma(splat(deg_out), g2[p], hid[p]));
    PHASE();
    mv2<D>(T + L::T_U1H, x, hid);
    PHASE();
    mv2<D>(T + L::T_GTO, sto, hid);
    PHASE();
    mv2<D>(T + L::T_GFR, sfr, hid);
    PHASE();
    mv2<P>(T + L::T_U1P, pq, hid);
#pragma unroll
    for (int p = 0; p < 5; ++p) hid[p] = __builtin_elementwise_max(hid[p], splat(0.f));
    ld5(T + L::T_C2, upd);
    PHASE();
    mv2<D>(T + L::T_U2, reinterpret_cast<const float*>(hid), upd);
#pragma unroll
    for (int p = 0; p < 5; ++p) y2[p] = __builtin_elementwise_fma(splat(al), upd[p], (v2f){x[2 * p], x[2 * p + 1]});
  }
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    y[2 * p] = y2[p].x;
    y[2 * p + 1] = y2[p].y;
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
  }  // !dirichlet && active
  STAMP(4);
  PRIO_AT(3);
  if (!FUSED) {
    store10(out + n * D, y);
    STAMP(5);
    return;
  }
  // ---- fused Broyden epilogue
  float sg = 0.f, sf = 0.f;
  if (active) {
    float gn[D];
#pragma unroll
    for (int o = 0; o < D; ++o) {
      gn[o] = y[o] - x[o];
      sg = fmaf(gn[o], gn[o], sg);
      sf = fmaf(y[o], y[o], sf);
    }
    store10(fa.gnew + n * D, gn);
    if (!X_RELOAD || dirichlet) store10(fa.xbuf + (int64_t)fa.st[fa.off_nxt] * fa.M + n * D, x);
  }
  // one partial pair per tile: wave shuffles, then the 4 wave sums through LDS in a fixed order
  sg = wave_sum_f(sg);
  sf = wave_sum_f(sf);
  __shared__ float red[2][4];
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = sg;
    red[1][tid >> 6] = sf;
  }
  __syncthreads();
  if (tid == 0) {
    fa.part[tile] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    fa.part[fa.npart + tile] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Kernel: one workgroup per tile.  blocks b and b+8 share an XCD (round-robin dispatch): each XCD owns a contiguous run of
// `chunk` tiles, so that neighbouring tiles' halo rows hit the same L2.  Speed only; any mapping is correct.
// (Round 2 tried persistent workgroups -- a grid of 5 per CU striding through the XCD's run, and the same with a per-XCD
// atomic work queue whose next index is fetched during the current tile -- to fill the tail the phase stamps show: 60.9 - 62.2
// and 62.6 - 64.8 us against 57.7 - 58.7 us for this form on the 1M-node mesh; the loop state also cost the fused variant its
// fifth wave.  Removed; profiles/r2_f_tile_ab_runs.txt keeps the runs.)
#ifndef TILE_CTX_VALUE
#define TILE_CTX_VALUE 1   // 1: the tile pointers travel as a by-value struct in the kernel arguments; 0: behind a device
                           // pointer (scalar loads next to each use).  Measured, 1M nodes: plain f 52.6 - 53.9 us by value vs
                           // 54.9 - 57.7 us by pointer, fused step 94.0 vs 97.7 us
#endif
#if TILE_CTX_VALUE
#define TILE_CTX_PARAM const TileCtx Cv
#define TILE_CTX_USE const TileCtx* __restrict__ C = &Cv;
#else
#define TILE_CTX_PARAM const TileCtx* __restrict__ C
#define TILE_CTX_USE
#endif
template <int P, bool MIXED, bool FUSED, bool MFMA1>
__global__ __launch_bounds__(TILE_THREADS) TILE_WPE_ATTR void k_f_tile(FuseArgs fa, int n_tiles, int chunk, const int32_t* __restrict__ tile_list,
                                                TILE_CTX_PARAM, const float* __restrict__ W, int lofs, int tofs,
                                                int tnofs, int apply_ln, const float* __restrict__ h,
                                                const int32_t* __restrict__ hsel, int64_t hstride,
                                                const float* __restrict__ h0, const float* __restrict__ prb,
                                                const float* __restrict__ nrm, float* __restrict__ out) {
  TILE_CTX_USE
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (FUSED && fa.st[fa.off_done]) return;
  const int slot = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot >= n_tiles) return;
  f_tile_body<P, MIXED, FUSED, MFMA1>(fa, slot, tile_list, C, W, lofs, tofs, tnofs, apply_ln, h, hsel, hstride, h0, prb, nrm, out,
                                      lds);
}

static unsigned tile_grid(int chunk);

// Batched fused Broyden step (solver.hip psignn_broyden_solve_batch; dirichlet, single layer): ONE launch evaluates the
// fused step of every mesh of a shard.  The tiles of all meshes form one list (mesh m owns slots [tile_base[m],
// tile_base[m] + n_tiles[m])); a workgroup looks its slot's mesh up, loads that mesh's pointers from its descriptor and runs
// the same tile body as the single-mesh kernel -- same arithmetic per node, same per-tile norm partials.  A mesh whose stop
// test has fired is skipped tile by tile.
#ifndef BATCH_WPE
#define BATCH_WPE 0   // > 0: the batched fused kernel is held to this many waves per SIMD (A/B in profiles/r2_f_tile_ab_runs.txt)
#endif
#if BATCH_WPE
#define BATCH_WPE_ATTR __attribute__((amdgpu_waves_per_eu(BATCH_WPE, BATCH_WPE)))
#else
#define BATCH_WPE_ATTR
#endif
template <int P, bool MIXED>
__global__ __launch_bounds__(TILE_THREADS) BATCH_WPE_ATTR void k_f_tile_batch(const BatchDesc* __restrict__ descs, int n_mesh, int n_slots, int chunk,
                                                              int off_done, int off_cur, int off_nxt, int par,
                                                              const float* __restrict__ W, int lofs, int tofs, int tnofs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int slot = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (slot >= n_slots) return;
  int m = 0;
  while (m + 1 < n_mesh && descs[m + 1].tile_base <= slot) ++m;   // wave-uniform scalar walk (a shard has few meshes)
  const BatchDesc& d = descs[m];
  if (d.st[off_done]) return;
  FuseArgs fa{d.upd, par ? d.g0 : d.g1, d.xbuf, d.st, off_done, off_cur, off_nxt, d.M, d.nrm_part, d.n_tiles, nullptr};
  f_tile_body<P, MIXED, true, false>(fa, slot - d.tile_base, nullptr, d.ctx, W, lofs, tofs, tnofs, 1, d.xbuf, nullptr, 0, d.h0p, d.prbp,
                                     d.nrmp, nullptr, lds);
}

// descs: device array of n_mesh descriptors; max_rows: largest tile + halo row count over the meshes (LDS size).
// mixed shards: every tile runs the full kernel (128-byte LDS rows, Neumann branch for the lanes that need it) in tile order --
// what the single-mesh fused step does below 2 048 plain tiles, and bit-identical to its two-group launch above that
// (tests/test_gpu_configs.py::test_mixed_two_group_launch_at_natural_size).
int psignn_f_tile_fused_batch(const BatchDesc* d_descs, int n_mesh, int n_slots, int max_rows, const float* W, int mixed,
                              int off_done, int off_cur, int off_nxt, int par, hipStream_t st) {
  const int chunk = (int)cdiv(n_slots, 8);
  if (mixed) {
    using L = WLayout<3>;
    LAUNCH("k_f_tile_fused", st, (k_f_tile_batch<3, true><<<tile_grid(chunk), TILE_THREADS, (size_t)max_rows * TileRow<true>::RS * 4, st>>>(
        d_descs, n_mesh, n_slots, chunk, off_done, off_cur, off_nxt, par, W, L::layer(0), L::tp_layer(1, true, 0), L::tp_neu(1))));
  } else {
    using L = WLayout<2>;
    LAUNCH("k_f_tile_fused", st, (k_f_tile_batch<2, false><<<tile_grid(chunk), TILE_THREADS, (size_t)max_rows * TileRow<false>::RS * 4, st>>>(
        d_descs, n_mesh, n_slots, chunk, off_done, off_cur, off_nxt, par, W, L::layer(0), L::tp_layer(1, false, 0), 0)));
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// gather of node rows between the caller's numbering and the plan order: dst[i] = src[map[i]]
__global__ void k_permute_rows(int64_t N, int cols, const int32_t* __restrict__ map, const float* __restrict__ src,
                               float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * cols) return;
  int64_t r = i / cols, c = i - r * cols;
  dst[i] = src[(int64_t)map[r] * cols + c];
}

// ------------------------------------------------------------------------------------------ host
#if TILE_CTX_VALUE
#define TILE_ARGS p->h_ctx
#else
#define TILE_ARGS p->d_ctx
#endif

// Grid of a tile-kernel launch over `chunk` tiles per XCD: one workgroup per tile, a multiple of 8.
static unsigned tile_grid(int chunk) { return (unsigned)(chunk * 8); }

// ---- experiments behind run-time knobs ("the bound of k_f_tile"; defaults leave the launch unchanged)
// PSIGNN_TILE_ORDER = cost: inside each XCD's run of tiles, launch the tiles with the most work first (stage-1 rows + slot
// rows), so that the launch ends on its shortest tiles.  Results are tile-order invariant (norm partials are stored per tile).
static const int32_t* tile_cost_order(const psignn_plan* p, int chunk, hipStream_t st) {
  KNOB_INT(mode, [] { const char* e = getenv("PSIGNN_TILE_ORDER"); return (int)(e && strcmp(e, "cost") == 0); }());
  if (!mode || p->mixed) return nullptr;
  if (p->tile_order_cost) return p->tile_order_cost;
  const int64_t nt = p->n_tiles;
  std::vector<int32_t> tptr(nt + 1), tsl(nt + 1), hc(nt), order(nt);
  std::vector<uint8_t> deg(p->n_slices);
  if (hipMemcpy(tptr.data(), p->tile_ptr, (nt + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(tsl.data(), p->tile_slice, (nt + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(hc.data(), p->halo_cnt, nt * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(deg.data(), p->slice_deg, p->n_slices, hipMemcpyDeviceToHost) != hipSuccess)
    return nullptr;
  std::vector<int64_t> cost(nt);
  for (int64_t t = 0; t < nt; ++t) {
    int dmax = 0;
    for (int s = tsl[t]; s < tsl[t + 1]; ++s) dmax = std::max(dmax, (int)deg[s]);
    cost[t] = (int64_t)(tptr[t + 1] - tptr[t] + hc[t]) + 16 * (int64_t)dmax;   // a slot row costs ~ 60 instructions, a staged row ~ 4 per lane
  }
  for (int64_t t = 0; t < nt; ++t) order[t] = (int32_t)t;
  for (int x = 0; x < 8; ++x) {
    const int64_t a = std::min<int64_t>(nt, (int64_t)x * chunk), b = std::min<int64_t>(nt, (int64_t)(x + 1) * chunk);
    std::stable_sort(order.begin() + a, order.begin() + b, [&](int32_t u, int32_t v) { return cost[u] > cost[v]; });
  }
  int32_t* d = nullptr;
  if (hipMalloc((void**)&d, nt * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(d, order.data(), nt * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
  p->tile_order_cost = d;
  (void)st;
  return d;
}

static FuseArgs plain_args() {
  FuseArgs a{};
  a.stamps = g_tile_stamps;
  return a;
}

// Mixed family: two launches over disjoint tile groups.  Tiles without Neumann nodes (all but the boundary tiles) run the
// kernel WITHOUT the Neumann branch and with 80-byte LDS rows (the Phi_neumann columns are only read by Neumann lanes);
// the few tiles with Neumann nodes run the full kernel with 128-byte rows.  Same arithmetic per node either way.
template <bool FUSED>
static void launch_mixed(const psignn_plan* p, const FuseArgs& fa, const char* name, const float* W, int nl, const float* h,
                         const int32_t* hsel, int64_t hstride, const float* h0, const float* prb, const float* nrm,
                         float* out, hipStream_t st) {
  using L = WLayout<3>;
  const int lofs = L::layer(nl - 1), tofs = L::tp_layer(nl, true, nl - 1), tnofs = L::tp_neu(nl);
  int na = (int)p->n_tiles_plain, nb = (int)(p->n_tiles - p->n_tiles_plain);
  // the second launch costs ~10 us of latency: worth it only when the first group is long (1M nodes: plain f 99 -> 82 us;
  // 100 k nodes: 3 % slower per Broyden iteration) -- below 2 048 plain tiles everything runs in the full kernel
  const char* e = getenv("PSIGNN_MIXED_SPLIT_MIN");   // tests force the two-group path on small meshes with 0
  if (na < (e ? atoi(e) : 2048)) {
    na = 0;
    nb = (int)p->n_tiles;
  }
  const int32_t* list_b = na > 0 ? p->tile_order + na : nullptr;
  if (na > 0) {
    const int chunk = (int)cdiv(na, 8);
    LAUNCH(name, st, (k_f_tile<3, false, FUSED, false><<<tile_grid(chunk), TILE_THREADS, (size_t)p->max_rows * TileRow<false>::RS * 4, st>>>(
        fa, na, chunk, p->tile_order, TILE_ARGS, W, lofs, tofs, tnofs, 1, h, hsel, hstride, h0, prb, nrm, out)));
  }
  if (nb > 0) {
    const int chunk = (int)cdiv(nb, 8);
    LAUNCH(name, st, (k_f_tile<3, true, FUSED, false><<<tile_grid(chunk), TILE_THREADS, (size_t)p->max_rows * TileRow<true>::RS * 4, st>>>(
        fa, nb, chunk, list_b, TILE_ARGS, W, lofs, tofs, tnofs, 1, h, hsel, hstride, h0, prb, nrm, out)));
  }
}

// All tensors in plan order.  h = hbase + (*hsel) * hstride when hsel != NULL.
int psignn_f_tile_forward(const psignn_plan* p, const float* W, int nl, const float* h, const int32_t* hsel,
                          int64_t hstride, const float* h0, const float* prb, const float* nrm, float* out,
                          float* work, hipStream_t st) {
  ARG_CHECK(p && p->tiled, "plan has no tile structures");
  ARG_CHECK(W && h && h0 && prb && out, "NULL argument");
  ARG_CHECK(!p->mixed || nrm, "mixed plan needs unit normals");
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = tile_grid(chunk);
  // B_f: h, h' (40 N each), prb (8 / 12 N), flags and, mixed, normals; 20 bytes per directed non-self edge
  PROF_BYTES(((p->mixed ? 102 : 89) * p->N + 20 * p->Ep) * (p->mixed ? 1 : nl));
  if (p->mixed) {
    using L = WLayout<3>;
    if (stage1_mfma(false, true)) {  // single launch, MFMA stage 1 (PSIGNN_STAGE1=mfma)
      size_t lds = (size_t)p->max_rows * TileRow<true>::RS * 4;
      LAUNCH("k_f_tile", st, (k_f_tile<3, true, false, true><<<grid, TILE_THREADS, lds, st>>>(
        plain_args(), (int)p->n_tiles, chunk, nullptr, TILE_ARGS, W, L::layer(nl - 1), L::tp_layer(nl, true, nl - 1), L::tp_neu(nl), 1,
        h, hsel, hstride, h0, prb, nrm, out)));
    } else {
      launch_mixed<false>(p, plain_args(), "k_f_tile", W, nl, h, hsel, hstride, h0, prb, nrm, out, st);
    }
  } else {
    using L = WLayout<2>;
    size_t lds = std::max((size_t)p->max_rows * TileRow<false>::RS * 4, tile_lds_min());
    const int32_t* tlist = tile_cost_order(p, chunk, st);
    ARG_CHECK(nl == 1 || work, "multi-layer evaluation needs a workspace");
    float* pp[2] = {work, work ? work + p->N * D : nullptr};
    const float* cur = h;
    for (int l = 0; l < nl; ++l) {
      float* dst = (l == nl - 1) ? out : pp[l & 1];
      if (stage1_mfma(false, false))
        LAUNCH("k_f_tile", st, (k_f_tile<2, false, false, true><<<grid, TILE_THREADS, lds, st>>>(
            plain_args(), (int)p->n_tiles, chunk, tlist, TILE_ARGS, W, L::layer(l), L::tp_layer(nl, false, l), 0, l == nl - 1, cur,
            l == 0 ? hsel : nullptr, hstride, h0, prb, nrm, dst)));
      else
        LAUNCH("k_f_tile", st, (k_f_tile<2, false, false, false><<<grid, TILE_THREADS, lds, st>>>(
            plain_args(), (int)p->n_tiles, chunk, tlist, TILE_ARGS, W, L::layer(l), L::tp_layer(nl, false, l), 0, l == nl - 1, cur,
            l == 0 ? hsel : nullptr, hstride, h0, prb, nrm, dst)));
      cur = dst;
    }
  }
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}

// Fused Broyden step (single-layer models): x_next = x_cur + upd, f(x_next), g_new / x_next / norm partials.
// Returns the number of partial entries per norm (n_tiles), or a negative error.
int psignn_f_tile_fused(const psignn_plan* p, const float* W, int nl, float* xbuf, int64_t M, const int32_t* st_words,
                        int off_done, int off_cur, int off_nxt, const float* upd, float* gnew,
                        const float* h0, const float* prb, const float* nrm, float* part, hipStream_t st) {
  ARG_CHECK(p && p->tiled, "plan has no tile structures");
  ARG_CHECK(nl == 1 || p->mixed, "fused step supports single-layer evaluation");
  const int chunk = (int)cdiv(p->n_tiles, 8);
  const unsigned grid = tile_grid(chunk);
  const int npart = (int)p->n_tiles;
  FuseArgs fa{upd, gnew, xbuf, st_words, off_done, off_cur, off_nxt, M, part, npart, g_tile_stamps};
  PROF_BYTES((p->mixed ? 102 : 89) * p->N + 20 * p->Ep + 8 * M);   // B_f + update read, g_new written (x_next replaces f(x))
  if (p->mixed) {
    using L = WLayout<3>;
    if (stage1_mfma(true, true)) {
      size_t lds = (size_t)p->max_rows * TileRow<true>::RS * 4;
      LAUNCH("k_f_tile_fused", st, (k_f_tile<3, true, true, true><<<grid, TILE_THREADS, lds, st>>>(
        fa, (int)p->n_tiles, chunk, nullptr, TILE_ARGS, W, L::layer(nl - 1), L::tp_layer(nl, true, nl - 1), L::tp_neu(nl), 1, xbuf,
        nullptr, 0, h0, prb, nrm, nullptr)));
    } else {
      launch_mixed<true>(p, fa, "k_f_tile_fused", W, nl, xbuf, nullptr, 0, h0, prb, nrm, nullptr, st);
    }
  } else {
    using L = WLayout<2>;
    size_t lds = std::max((size_t)p->max_rows * TileRow<false>::RS * 4, tile_lds_min());
    const int32_t* tlist = tile_cost_order(p, chunk, st);
    if (stage1_mfma(true, false))
      LAUNCH("k_f_tile_fused", st, (k_f_tile<2, false, true, true><<<grid, TILE_THREADS, lds, st>>>(
        fa, (int)p->n_tiles, chunk, tlist, TILE_ARGS, W, L::layer(0), L::tp_layer(nl, false, 0), 0, 1, xbuf, nullptr, 0, h0, prb, nrm,
        nullptr)));
    else
      LAUNCH("k_f_tile_fused", st, (k_f_tile<2, false, true, false><<<grid, TILE_THREADS, lds, st>>>(
        fa, (int)p->n_tiles, chunk, tlist, TILE_ARGS, W, L::layer(0), L::tp_layer(nl, false, 0), 0, 1, xbuf, nullptr, 0, h0, prb, nrm,
        nullptr)));
  }
  HIP_TRY(hipGetLastError());
  return npart;
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
  k_permute_rows<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(p->N, cols, to_plan ? p->perm : p->inv, src, dst);
  HIP_TRY(hipGetLastError());
  return PSIGNN_OK;
}
