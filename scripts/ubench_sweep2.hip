// What separates the solver's U / V sweeps (0.63 - 0.65 of 8 TB/s) from a bare kernel of their access shape (0.775)?  (gfx950)
// Every variant streams k stored vectors of M floats, a block owning 256 * VEC consecutive elements of each (csrc/solver.hip):
//   bare16   loads + fma, nothing else, 16 floats per lane                  (profiles/r2_ubench_sweep.txt, layout A, one array)
//   bare4    the same at 4 floats per lane (shape of k_sweep_u2d / u2r)
//   dot16    + one wave_sum and one 4-byte partial store per vector          (k_sweep_u1)
//   dot4     the same at 4 floats per lane
//   coef16   + two scalar coefficient loads per vector, two accumulators     (k_sweep_u2)
// each on ZERO data and on RANDOM data (HBM and fabric power depend on the bits that toggle).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_sweep2 scripts/ubench_sweep2.hip && scripts/ubench_sweep2 [M] [k]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int TB = 256;
template <int CTRL> __device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_perm<0xB1>(v); v += dpp_perm<0x4E>(v); v += dpp_perm<0x141>(v); v += dpp_perm<0x140>(v);
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}
// MODE 0 bare, 1 dot (wave_sum + partial store per vector), 2 coef (scalar coefficient loads, two accumulators),
// 3 dot without the store, 4 the one-lane store without the reduction, 5 reduction + a full-wave 256-byte store
template <int VEC, int MODE, int UNR>
__global__ __launch_bounds__(TB) void k_sweep(const float* __restrict__ U, long long ld, int k, const float* __restrict__ coef,
                                              float* __restrict__ part, int npart, float* __restrict__ out) {
  const long long e0 = ((long long)blockIdx.x * TB + (threadIdx.x & ~63)) * VEC + (threadIdx.x & 63) * 4;
  float a1[VEC], a2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a1[i] = 1.f + i; a2[i] = 0.f; }
  const int w = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const bool lead = (threadIdx.x & 63) == 0;
  __shared__ float stash[TB / 64][64];
  for (int j0 = 0; j0 < k; j0 += UNR) {
    float u[UNR][VEC];
#pragma unroll
    for (int q = 0; q < UNR; ++q)
      if (j0 + q < k) {
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) {
          const float4 t = *reinterpret_cast<const float4*>(U + (long long)(j0 + q) * ld + e0 + i * 256);
          u[q][4 * i] = t.x; u[q][4 * i + 1] = t.y; u[q][4 * i + 2] = t.z; u[q][4 * i + 3] = t.w;
        }
      }
#pragma unroll
    for (int q = 0; q < UNR; ++q)
      if (j0 + q < k) {
        if (MODE == 6) {   // wave sums stashed in LDS, combined per block and stored as one coalesced row per 64 vectors
          float sa = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) sa = fmaf(a1[i], u[q][i], sa);
          sa = wave_sum(sa);
          const int j = j0 + q;
          if (lead) stash[threadIdx.x >> 6][j & 63] = sa;
          if ((j & 63) == 63 || j == k - 1) {
            __syncthreads();
            const int cnt = (j & 63) + 1;
            if (threadIdx.x < cnt)
              part[(long long)blockIdx.x * 64 * ((k + 63) / 64) + (j & ~63) + threadIdx.x] =
                  (stash[0][threadIdx.x] + stash[1][threadIdx.x]) + (stash[2][threadIdx.x] + stash[3][threadIdx.x]);
            __syncthreads();
          }
        } else if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
          float sa = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) sa = fmaf(a1[i], u[q][i], sa);
          if (MODE != 4) sa = wave_sum(sa);                                          // 4: store without the reduction
          if (MODE == 3) a2[0] += sa;                                                 // 3: reduction without the store
          else if (MODE == 5) part[(long long)(j0 + q) * npart * 64 + w * 64 + (threadIdx.x & 63)] = sa;   // 5: full-wave 256-byte store
          else if (lead) part[((long long)(j0 + q) * npart + w) * 4] = sa;
        } else {
          const float c0 = MODE == 2 ? coef[j0 + q] : 0.5f, c1 = MODE == 2 ? coef[k + j0 + q] : 0.25f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) { a1[i] = fmaf(-c0, u[q][i], a1[i]); a2[i] = fmaf(-c1, u[q][i], a2[i]); }
        }
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s += a1[i] + a2[i];
  if (s == 123.456f) out[0] = s;
}
__global__ void k_fill(float* p, size_t n, int rnd) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = rnd ? ((x & 0xFFFFFF) / 16777216.f - 0.5f) : 0.f;
  }
}
template <int VEC, int MODE, int UNR>
static void run(const float* U, long long M, int k, const float* coef, float* part, float* out, const char* name) {
  const int nblk = (int)(M / (TB * VEC));
  const long long ld = M;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k_sweep<VEC, MODE, UNR><<<nblk, TB>>>(U, ld, k, coef, part, nblk * 4, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k_sweep<VEC, MODE, UNR><<<nblk, TB>>>(U, ld, k, coef, part, nblk * 4, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)k * nblk * TB * VEC * 4, gbs = bytes / (ms / reps * 1e-3) / 1e9;
  printf("%-28s %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms / reps * 1e3, gbs, gbs / 8000.0);
}
int main(int argc, char** argv) {
  long long M = argc > 1 ? atoll(argv[1]) : 10005190; const int k = argc > 2 ? atoi(argv[2]) : 25;
  M = M / 4096 * 4096;
  float *U, *coef, *part, *out;
  CHECK(hipMalloc(&U, (size_t)k * M * 4)); CHECK(hipMalloc(&coef, 2 * k * 4 + 16)); CHECK(hipMalloc(&out, 16));
  CHECK(hipMalloc(&part, (size_t)k * (M / 1024) * 4 * 64 * 4 + 64));
  CHECK(hipMemset(coef, 0, 2 * k * 4));
  for (int rnd = 1; rnd < 2; ++rnd) {
    k_fill<<<4096, 256>>>(U, (size_t)k * M, rnd); CHECK(hipDeviceSynchronize());
    printf("M = %lld floats, k = %d vectors, %.2f GB per sweep, data: %s\n", M, k, (double)k * M * 4 / 1e9, rnd ? "random" : "zeros");
    run<16, 0, 1>(U, M, k, coef, part, out, "bare16 unroll 1");
    run<16, 0, 2>(U, M, k, coef, part, out, "bare16 unroll 2");
    run<4, 0, 1>(U, M, k, coef, part, out, "bare4 unroll 1");
    run<4, 0, 8>(U, M, k, coef, part, out, "bare4 unroll 8");
    run<16, 1, 1>(U, M, k, coef, part, out, "dot16 unroll 1");
    run<16, 1, 2>(U, M, k, coef, part, out, "dot16 unroll 2");
    run<4, 1, 8>(U, M, k, coef, part, out, "dot4 unroll 8");
    run<4, 1, 1>(U, M, k, coef, part, out, "dot4 unroll 1");
    run<16, 3, 1>(U, M, k, coef, part, out, "dot16 no store");
    run<4, 3, 8>(U, M, k, coef, part, out, "dot4 no store unroll 8");
    run<16, 4, 1>(U, M, k, coef, part, out, "dot16 store, no wave_sum");
    run<4, 4, 8>(U, M, k, coef, part, out, "dot4 store, no wave_sum");
    run<16, 5, 1>(U, M, k, coef, part, out, "dot16 full-wave store");
    run<16, 6, 1>(U, M, k, coef, part, out, "dot16 block rows");
    run<16, 6, 2>(U, M, k, coef, part, out, "dot16 block rows unroll 2");
    run<4, 6, 8>(U, M, k, coef, part, out, "dot4 block rows unroll 8");
    run<4, 6, 1>(U, M, k, coef, part, out, "dot4 block rows unroll 1");
    run<16, 2, 1>(U, M, k, coef, part, out, "coef16 unroll 1");
    run<16, 2, 2>(U, M, k, coef, part, out, "coef16 unroll 2");
    run<4, 2, 8>(U, M, k, coef, part, out, "coef4 unroll 8");
  }
  return 0;
}
