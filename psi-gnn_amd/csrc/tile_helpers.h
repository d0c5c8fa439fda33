// Device helpers shared by the tile kernels (fgnn_tile.hip, dsgps_tile.hip): packed-fp32 matvecs on the transposed
// weight blocks and the pair-merged slot walk of one edge direction (gfx950).
#pragma once
#include "fgnn_common.h"

#define SLOT_IN 0x10000u
#define SLOT_OUT 0x20000u

// Packed fp32: CDNA issues one VALU instruction per wave every 4 cycles; v_pk_fma_f32 / v_pk_add_f32 carry two
// floats per lane in that slot, so the FP32 peak (and this kernel, which is VALU-issue bound) needs them.  All
// matrices are read from the transposed weight section ([in k][out o], o fastest): the outputs (2p, 2p+1) of one
// input k sit in one SGPR pair.  Each output is still the same k-ordered fma chain as in fgnn.hip.
// Workgroup = 4 waves = TILE_MAX lanes.  (A fifth wave that only helps with the halo rows of stage 1 was tried:
// 79 us instead of 67 us per 1M-node evaluation -- it costs a wave slot per workgroup for the whole residency.)
#define TILE_THREADS 256
#ifndef EDGE_PD
#define EDGE_PD 2   // prefetch distance of the slot loads in edge_pass
#endif

// Compiler-level memory barrier between the phases of a kernel.  The tile kernels read > 1 000 wave-uniform weights;
// fully unrolled, the compiler hoists all their scalar loads to the top and then spills hundreds of SGPRs into VGPR
// lanes (v_writelane / v_readlane = VALU issue slots in VALU-issue-bound kernels).  The barrier keeps each phase's
// scalar loads next to their use.
#ifndef PHASE
#define PHASE() asm volatile("" ::: "memory")
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }

// acc[p] += WT[k][2p..2p+1] * x[k],  p < 5, k < K
// A 10 x 10 block is 100 wave-uniform weights -- with the wave's other live SGPRs more than the 102 there are, and the
// compiler, which loads a whole block ahead of its first use, then parks the excess in VGPR lanes (v_writelane /
// v_readlane: 36 VALU slots per stage-1 round in round 1's build).  A compiler barrier every MV2_CH inputs keeps at most
// MV2_CH x 10 weights in flight.
#ifndef MV2_CH
#define MV2_CH 5
#endif
// MV2_LAUNDER (round 3): the barrier alone does not hold the loads back -- the weights sit behind a `const __restrict__` kernel
// argument, their loads are invariant for the optimiser and were still hoisted over it: the fused f step and k_jvp_lin carried
// 220 / 204 v_writelane / v_readlane per wave (a sixth of their VALU instructions) for parked weights.  Adding an
// offset that went through an empty asm (always 0) to the block's pointer per chunk makes the chunk's loads data-dependent on
// that statement: they cannot move above it.  (Laundering the pointer itself loses its no-alias property: vector loads, 256 VGPRs.)
#ifndef MV2_LAUNDER
#define MV2_LAUNDER 0   // 0: barrier only; 1: offset re-made per chunk; 2: ... by a statement that also waits for the chunk before.  Chosen per
                        // file (fgnn_tile.hip, fgnn_tile_lin.hip: 2); which form leaves fewest parked SGPRs differs from kernel to kernel
#endif
template <int K>
__device__ __forceinline__ void mv2(const float* __restrict__ WT, const float* x, v2f* acc, const float after = 0.f) {
  const v2f* w = reinterpret_cast<const v2f*>(WT);
#if MV2_LAUNDER
  // An offset the optimiser cannot see through (always 0; the POINTER stays derived from the restrict argument: scalar loads),
  // re-made per chunk by a statement that also consumes the accumulators of the chunk before -- the chunk's loads can neither be
  // hoisted over it nor can the statement itself float above the arithmetic it waits for.  `after`: a value of the caller's that
  // the first chunk has to wait for (e.g. the last output of an independent product issued just before).
  int zero = 0;
#if MV2_LAUNDER == 2
  asm volatile("" : "+s"(zero) : "v"(acc[0]), "v"(acc[4]), "v"(after));
#else
  asm volatile("" : "+s"(zero));
#endif
  w += zero;
#endif
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (MV2_CH > 0 && k > 0 && k % MV2_CH == 0) {
      PHASE();
#if MV2_LAUNDER == 2
      asm volatile("" : "+s"(zero) : "v"(acc[0]), "v"(acc[4]));
      w += zero;
#elif MV2_LAUNDER
      asm volatile("" : "+s"(zero));
      w += zero;
#endif
    }
    const v2f xs = splat(x[k]);
#pragma unroll
    for (int p = 0; p < 5; ++p) acc[p] = __builtin_elementwise_fma(w[k * 5 + p], xs, acc[p]);
  }
}
__device__ __forceinline__ void ld5(const float* __restrict__ p, v2f* r) {  // 10 wave-uniform floats (8-byte aligned)
  const v2f* q = reinterpret_cast<const v2f*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) r[i] = q[i];
}

// One direction of the neighbour sum for this lane's node:
//   S[o] += relu(Pi[o] + row[COL + o] + AT[:, o] . (a0, a1, a2))   over the slots that carry `MASK`
// (AT = W1[:, 20:23]^T; for in-edges its first two rows are stored negated, because an in-edge's attr is the
// mirror (-a0, -a1, a2) of the slot's attr).  The two directions are separate passes on purpose: one pass needs
// 30 wave-uniform weights, which stay in SGPRs across the loop; with 60 the compiler re-issues the scalar loads
// and their waits in every iteration.  The second pass re-reads the 16-byte slots from L1/L2.
template <int RS, int COL, unsigned MASK>
__device__ __forceinline__ float edge_pass(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                           const float* __restrict__ AT, const v2f* Pi, v2f* S) {
  float deg = 0.f;
  v2f wa[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) wa[i] = reinterpret_cast<const v2f*>(AT)[i];
  auto one = [&](const uint4 s) {
    const unsigned w = s.x;
    if ((w & 0xFFFFu) != ELL_EMPTY && (w & MASK)) {
      const v2f a0 = splat(__uint_as_float(s.y)), a1 = splat(__uint_as_float(s.z)), a2 = splat(__uint_as_float(s.w));
      const float* row = lds + (int)(w & 0xFFFFu) * RS + COL;
      v2f pj[5];
      if (COL % 4 == 0) {  // 16-byte aligned start: b128, b128, b64
        float4 v0 = reinterpret_cast<const float4*>(row)[0], v1 = reinterpret_cast<const float4*>(row)[1];
        float2 v2 = reinterpret_cast<const float2*>(row)[4];
        pj[0] = (v2f){v0.x, v0.y}; pj[1] = (v2f){v0.z, v0.w}; pj[2] = (v2f){v1.x, v1.y}; pj[3] = (v2f){v1.z, v1.w};
        pj[4] = (v2f){v2.x, v2.y};
      } else {             // start at 8 mod 16: b64, b128, b128
        float2 v0 = reinterpret_cast<const float2*>(row)[0];
        float4 v1 = reinterpret_cast<const float4*>(row + 2)[0], v2 = reinterpret_cast<const float4*>(row + 2)[1];
        pj[0] = (v2f){v0.x, v0.y}; pj[1] = (v2f){v1.x, v1.y}; pj[2] = (v2f){v1.z, v1.w}; pj[3] = (v2f){v2.x, v2.y};
        pj[4] = (v2f){v2.z, v2.w};
      }
      deg += 1.f;
      // five independent chains, written stage by stage so that dependent packed ops are never back to back
      v2f z[5];
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = Pi[p] + pj[p];
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[p], a0, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[5 + p], a1, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wa[10 + p], a2, z[p]);
#pragma unroll
      for (int p = 0; p < 5; ++p) S[p] += __builtin_elementwise_max(z[p], splat(0.f));
    }
  };
  // software pipeline, distance EDGE_PD: the 16-byte slot loads are unconditional (index clamped, never branched
  // on) so the compiler keeps them whole and places their waits EDGE_PD - 1 iterations later
  if (nslots <= 0) return deg;
  uint4 c[EDGE_PD];
#pragma unroll
  for (int i = 0; i < EDGE_PD; ++i) c[i] = slots[(int64_t)min(i, nslots - 1) * 64];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + EDGE_PD, nslots - 1) * 64];
    one(c[0]);
#pragma unroll
    for (int i = 0; i + 1 < EDGE_PD; ++i) c[i] = c[i + 1];
    c[EDGE_PD - 1] = nx;
  }
  return deg;
}


// ---- relu folded into the last fma of an edge pre-activation -----------------------------------------------------
// v_max_f32 is not packed on gfx950, so  S += relu(z)  costs two v_max + one v_pk_add per output pair -- 10 of the 35
// VALU slots of one edge direction.  The VOP3P clamp bit clamps a result to [0, 1] for free; computing the
// pre-activation SCALED by 2^-40 (every term times a power of two: bit-exact, fp32 rounding commutes with it unless a
// value underflows below 2^-126, i.e. |z| < 2^-86) turns relu(z) = 2^40 clamp(2^-40 z) for every |z| < 2^40 ~ 1e12.
// NaN clamps to 0, as v_max(NaN, 0) did.  The attr values a0, a1, a2 are broadcast from the halves of two register
// pairs with op_sel (the compiler would spend six v_mov per slot on duplicating them).
#ifndef CLAMP_PD
#define CLAMP_PD 1    // slot records loaded this many rounds ahead in edge_pass_both_clamp (1 or 2)
#endif
#define RELU_SCALE 9.094947017729282e-13f   // 2^-40
#define RELU_UNSCALE 1099511627776.f        // 2^40
__device__ __forceinline__ v2f pk_fma_lo(v2f w, v2f a, v2f z) {  // z + w * (a.x, a.x)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "s"(w), "v"(a), "v"(z));
  return r;
}
__device__ __forceinline__ v2f pk_fma_hi(v2f w, v2f a, v2f z) {  // z + w * (a.y, a.y)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "s"(w), "v"(a), "v"(z));
  return r;
}
__device__ __forceinline__ v2f pk_fma_lo_clamp(v2f w, v2f a, v2f z) {  // clamp(z + w * (a.x, a.x), 0, 1)
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "s"(w), "v"(a), "v"(z));
  return r;
}

// edge_pass_both with the clamp form: S_to / S_fr come back UNSCALED (multiplied by 2^40 at the end).
template <int RS>
__device__ __forceinline__ void edge_pass_both_clamp(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                                     const float* __restrict__ AT_to, const float* __restrict__ AT_fr,
                                                     const v2f* Pi_to, const v2f* Pi_fr, v2f* S_to, v2f* S_fr,
                                                     float& deg_in, float& deg_out, const uint4* first = nullptr) {
  v2f wt[15], wf[15], pt[5], pf[5];
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    wt[i] = reinterpret_cast<const v2f*>(AT_to)[i];
    wf[i] = reinterpret_cast<const v2f*>(AT_fr)[i];
  }
  const v2f sc = splat(RELU_SCALE);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    pt[p] = Pi_to[p] * sc;
    pf[p] = Pi_fr[p] * sc;
  }
  deg_in = deg_out = 0.f;
  if (nslots <= 0) return;
  uint4 c0 = first ? *first : slots[0];   // `first`: slot row 0, loaded by the caller ahead of time
#if CLAMP_PD == 2
  uint4 c1 = slots[(int64_t)min(1, nslots - 1) * 64];
#endif
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + CLAMP_PD, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY) {
      const v2f a01 = (v2f){__uint_as_float(c0.y), __uint_as_float(c0.z)} * sc;
      const v2f a2 = (v2f){__uint_as_float(c0.w) * RELU_SCALE, 0.f};
      const float4* row = reinterpret_cast<const float4*>(lds + (int)(w & 0xFFFFu) * RS);
      const float4 v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3], v4 = row[4];
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
        for (int p = 0; p < 5; ++p) S_to[p] += z[p];
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
        for (int p = 0; p < 5; ++p) S_fr[p] += z[p];
      }
    }
#if CLAMP_PD == 2
    c0 = c1;
    c1 = nx;
#else
    c0 = nx;
#endif
  }
  const v2f us = splat(RELU_UNSCALE);
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    S_to[p] *= us;
    S_fr[p] *= us;
  }
}

// Both directions of the neighbour sum in ONE walk over the slots: a pair-merged slot is decoded once, its 80-byte LDS row
// [to | from] is read once, and the IN half (Phi_to, mirrored attr weights) and the OUT half (Phi_from) are evaluated back
// to back.  Needs both attr blocks (60 wave-uniform floats) in SGPRs for the whole loop -- affordable once the phase
// barriers keep every other scalar load out of the loop's live range.
template <int RS>
__device__ __forceinline__ void edge_pass_both(const uint4* __restrict__ slots, int nslots, const float* __restrict__ lds,
                                               const float* __restrict__ AT_to, const float* __restrict__ AT_fr,
                                               const v2f* Pi_to, const v2f* Pi_fr, v2f* S_to, v2f* S_fr, float& deg_in,
                                               float& deg_out) {
  v2f wt[15], wf[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    wt[i] = reinterpret_cast<const v2f*>(AT_to)[i];
    wf[i] = reinterpret_cast<const v2f*>(AT_fr)[i];
  }
  deg_in = deg_out = 0.f;
  if (nslots <= 0) return;
  uint4 c0 = slots[0];
  uint4 c1 = slots[(int64_t)min(1, nslots - 1) * 64];
  for (int r = 0; r < nslots; ++r) {
    const uint4 nx = slots[(int64_t)min(r + 2, nslots - 1) * 64];
    const unsigned w = c0.x;
    if ((w & 0xFFFFu) != ELL_EMPTY) {
      const v2f a0 = splat(__uint_as_float(c0.y)), a1 = splat(__uint_as_float(c0.z)), a2 = splat(__uint_as_float(c0.w));
      const float4* row = reinterpret_cast<const float4*>(lds + (int)(w & 0xFFFFu) * RS);
      const float4 v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3], v4 = row[4];
      if (w & SLOT_IN) {
        v2f z[5] = {(v2f){v0.x, v0.y}, (v2f){v0.z, v0.w}, (v2f){v1.x, v1.y}, (v2f){v1.z, v1.w}, (v2f){v2.x, v2.y}};
        deg_in += 1.f;
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] += Pi_to[p];
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wt[p], a0, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wt[5 + p], a1, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wt[10 + p], a2, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) S_to[p] += __builtin_elementwise_max(z[p], splat(0.f));
      }
      if (w & SLOT_OUT) {
        v2f z[5] = {(v2f){v2.z, v2.w}, (v2f){v3.x, v3.y}, (v2f){v3.z, v3.w}, (v2f){v4.x, v4.y}, (v2f){v4.z, v4.w}};
        deg_out += 1.f;
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] += Pi_fr[p];
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wf[p], a0, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wf[5 + p], a1, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) z[p] = __builtin_elementwise_fma(wf[10 + p], a2, z[p]);
#pragma unroll
        for (int p = 0; p < 5; ++p) S_fr[p] += __builtin_elementwise_max(z[p], splat(0.f));
      }
    }
    c0 = c1;
    c1 = nx;
  }
}
