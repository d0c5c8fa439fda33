// Small per-lane helpers shared by the GNN-block kernels (gfx950).
#pragma once
#include "common.h"

__device__ __forceinline__ void load10(const float* __restrict__ p, float* __restrict__ r) {
  const float2* q = reinterpret_cast<const float2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float2 t = q[i];
    r[2 * i] = t.x;
    r[2 * i + 1] = t.y;
  }
}
__device__ __forceinline__ void store10(float* __restrict__ p, const float* __restrict__ r) {
  float2* q = reinterpret_cast<float2*>(p);
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = make_float2(r[2 * i], r[2 * i + 1]);
}

// out[o] (+)= sum_k W[o*ld + off + k] * x[k],  o < 10, k < K   (W wave-uniform -> scalar loads)
template <int K, bool ACC>
__device__ __forceinline__ void matvec10(const float* __restrict__ W, int ld, int off, const float* x, float* out) {
#pragma unroll
  for (int o = 0; o < D; ++o) {
    float s = ACC ? out[o] : 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s = fmaf(W[o * ld + off + k], x[k], s);
    out[o] = s;
  }
}

