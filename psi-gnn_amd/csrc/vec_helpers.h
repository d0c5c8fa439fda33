// Vector-kernel helpers shared by the solver kernels (solver.hip, fpiter.hip, krylov.hip): thread -> element mapping,
// coalesced float4 loads / stores, fixed-shape reductions (bitwise reproducible results).  gfx950.
#pragma once
#include "common.h"

// The vector kernels are templated on VEC = elements per thread (16 for long vectors, 4 when N*d is small so that
// the grid still covers the 256 CUs; chosen at solver creation).
#define TB 256           // threads per block

// Sum over the 64 lanes of a wave, the same value in every lane, fixed association (bitwise reproducible):
// quad pairs, quads, half rows, rows of 16 as DPP operands of the adds (no LDS-path permutes), then the four row sums
// through v_readlane.  The xor-shuffle form (`__shfl_xor` = ds_bpermute_b32, six dependent LDS round trips per value) held the
// dots pass 10 % behind the axpy pass; -DWAVE_SUM_SHFL=1 rebuilds it for A/B runs.
#ifndef WAVE_SUM_SHFL
#define WAVE_SUM_SHFL 0
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
#if WAVE_SUM_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
#else
  v += dpp_perm<0xB1>(v);    // quad_perm:[1,0,3,2]
  v += dpp_perm<0x4E>(v);    // quad_perm:[2,3,0,1]
  v += dpp_perm<0x141>(v);   // row_half_mirror
  v += dpp_perm<0x140>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
  const int vi = __builtin_bit_cast(int, v);   // (the builtin is typed int: a float argument would be converted by value)
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
#endif
}

// Two per-lane values -> ONE pair per block (wave shuffles, then the 4 wave sums through LDS, fixed order).
__device__ __forceinline__ void block_pair_store(float a, float b, float* __restrict__ part, int n) {
  __shared__ float red[2][TB / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[n + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}


// Thread -> element mapping of every vector kernel: a wave owns 64*VEC contiguous floats and reads them as VEC/4
// fully coalesced float4 rows (lane l takes floats [256*i + 4*l, +4) of the wave's span), so one load instruction
// covers 8 whole 128-byte lines instead of a quarter of 32 lines.  elem0 is the thread's lowest element.
template <int VEC>
__device__ __forceinline__ int64_t elem0() {
  return ((int64_t)blockIdx.x * TB + (threadIdx.x & ~63)) * VEC + (threadIdx.x & 63) * 4;
}
template <int VEC>
__device__ __forceinline__ void ldv(const float* __restrict__ p, int64_t e0, int64_t M, float* r) {
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const int64_t o = e0 + i * 256;
    if (o + 4 <= M) {
      float4 t = *reinterpret_cast<const float4*>(p + o);
      r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    } else {  // tail
#pragma unroll
      for (int c = 0; c < 4; ++c) r[4 * i + c] = (o + c < M) ? p[o + c] : 0.f;
    }
  }
}
// Loads of the U / V sweeps.  Non-temporal loads were measured and REJECTED: although every byte is read once
// per pass, k_dots / k_axpy at N = 1M, k = 0..49 took 557 / 550 us per launch with nt against 365 / 382 us with
// default-policy loads (round 1, MI355X).  -DPSIGNN_NT_SWEEPS=1 rebuilds the nt variant for A/B timing.
#ifndef PSIGNN_NT_SWEEPS
#define PSIGNN_NT_SWEEPS 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void ldv_stream(const float* __restrict__ p, int64_t e0, int64_t M, float* r) {
#if PSIGNN_NT_SWEEPS
  if (e0 + (VEC / 4 - 1) * 256 + 4 <= M) {
#pragma unroll
    for (int i = 0; i < VEC / 4; ++i) {
      f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + e0 + i * 256));
      r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    }
    return;
  }
#endif
  ldv<VEC>(p, e0, M, r);
}

template <int VEC>
__device__ __forceinline__ void stv(float* __restrict__ p, int64_t e0, int64_t M, const float* r) {
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const int64_t o = e0 + i * 256;
    if (o + 4 <= M) {
      *reinterpret_cast<float4*>(p + o) = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (o + c < M) p[o + c] = r[4 * i + c];
    }
  }
}

template <int STRIDE = 1>
__device__ inline double block_sum_partials(const float* __restrict__ p, int n, double* sh) {
  // fixed summation shape (lane-strided, 4 independent accumulators, then a tree): reproducible, and the
  // loads of one lane do not wait on each other (a dependent scalar loop here cost 20-30 us per call)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * TB < n; i += 4 * TB) {
    float a = p[(int64_t)i * STRIDE], b = p[(int64_t)(i + TB) * STRIDE], c = p[(int64_t)(i + 2 * TB) * STRIDE], d = p[(int64_t)(i + 3 * TB) * STRIDE];
    s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
  }
  for (; i < n; i += TB) s0 += (double)p[(int64_t)i * STRIDE];
  double s = (s0 + s1) + (s2 + s3);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

// Sum of n CONTIGUOUS float partials (n a multiple of 4, 16-byte aligned), fixed shape: a thread takes every TB-th quad, four quads in
// flight per trip (tens of thousands of partials per coefficient: the trip count, not the bytes, is what such a reduction waits for).
__device__ inline double block_sum_contig(const float* __restrict__ p, int n, double* sh) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const int nq = n >> 2;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * TB < nq; i += 4 * TB) {
    const float4 a = q[i], b = q[i + TB], c = q[i + 2 * TB], d = q[i + 3 * TB];
    s0 += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
    s1 += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
    s2 += ((double)c.x + (double)c.y) + ((double)c.z + (double)c.w);
    s3 += ((double)d.x + (double)d.y) + ((double)d.z + (double)d.w);
  }
  for (; i < nq; i += TB) {
    const float4 a = q[i];
    s0 += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// ---- dot-product partials as coalesced per-block rows (solver.hip, krylov.hip) ---------------------------------------------------
// The lead lane of every wave stashes its wave sums of stored vector j in LDS; after every 64 vectors (and after the last) the
// block's first 64 threads add the four waves' values in a fixed order and store ONE contiguous row segment -- instead of one
// 4-byte store per wave and vector, which costs a streaming sweep 8 - 16 % of its rate (profiles/r3_ubench_sweep2.txt).
template <int NV>
struct PairStash {
  float v[NV][TB / 64][64];
};
// every thread of the block calls this with the same j, cnt (1..64 pairs stashed at slots 0..cnt-1) -- it contains barriers
template <int NV>
__device__ __forceinline__ void stash_flush(PairStash<NV>& sh, int cnt, float* const (&row)[NV]) {
  __syncthreads();
  const int t = threadIdx.x;
  if (t < cnt) {
#pragma unroll
    for (int q = 0; q < NV; ++q) row[q][t] = (sh.v[q][0][t] + sh.v[q][1][t]) + (sh.v[q][2][t] + sh.v[q][3][t]);
  }
  __syncthreads();
}
// sum of n values p[b * stride], b = 0 .. n-1 (one column of a partials matrix; stride 1: a contiguous list), fixed shape for a given
// block size: thread-strided, 4 loads in flight, tree
__device__ inline double block_sum_col(const float* __restrict__ p, int n, int64_t stride, double* sh) {
  const int T = blockDim.x;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * T < n; i += 4 * T) {
    const float a = p[(int64_t)i * stride], b = p[(int64_t)(i + T) * stride], c = p[(int64_t)(i + 2 * T) * stride],
                d = p[(int64_t)(i + 3 * T) * stride];
    s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
  }
  for (; i < n; i += T) s0 += (double)p[(int64_t)i * stride];
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// two contiguous lists of n values at once (p and q), each summed in exactly block_sum_col's shape (thread-strided, four partial
// sums per thread, the same tree): one set of barriers for both
__device__ inline void block_sum_col2(const float* __restrict__ p, const float* __restrict__ q, int n, double* sh, double* sh2,
                                      double& sp, double& sq) {
  const int T = blockDim.x;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * T < n; i += 4 * T) {
    const float x0 = p[i], x1 = p[i + T], x2 = p[i + 2 * T], x3 = p[i + 3 * T];
    const float y0 = q[i], y1 = q[i + T], y2 = q[i + 2 * T], y3 = q[i + 3 * T];
    a0 += (double)x0; a1 += (double)x1; a2 += (double)x2; a3 += (double)x3;
    b0 += (double)y0; b1 += (double)y1; b2 += (double)y2; b3 += (double)y3;
  }
  for (; i < n; i += T) {
    a0 += (double)p[i];
    b0 += (double)q[i];
  }
  sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
  sh2[threadIdx.x] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[threadIdx.x] += sh[threadIdx.x + o];
      sh2[threadIdx.x] += sh2[threadIdx.x + o];
    }
    __syncthreads();
  }
  sp = sh[0];
  sq = sh2[0];
  __syncthreads();
}

// launch a VEC-templated kernel with the solver's vector width
#define VPLAIN(vec, kern, cfg, ...)                                     \
  do {                                                                  \
    if ((vec) == 16) kern<16><<<VCFG cfg>>>(__VA_ARGS__);               \
    else kern<4><<<VCFG cfg>>>(__VA_ARGS__);                            \
  } while (0)
#define VCFG(...) __VA_ARGS__
#define VLAUNCH(name, st, vec, kern, cfg, ...) LAUNCH(name, st, VPLAIN(vec, kern, cfg, __VA_ARGS__))

