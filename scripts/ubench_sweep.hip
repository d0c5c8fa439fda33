// Access-pattern microbenchmark for the U/V sweeps of the Broyden solver (gfx950): k stored pairs of M floats each, every block
// owns 256 * VEC consecutive elements of every vector and walks the pairs j = 0 .. k-1 (the shape of k_axpy in csrc/solver.hip).
//   A  two arrays U[j][M], V[j][M]                      (the solver's layout)
//   B  one array  UV[j][block][2][256 * VEC]            (a block's two chunks adjacent in memory)
//   C  one array  UV[j][2][M]                           (a pair's two vectors adjacent)
//   S  plain streaming read of the same number of bytes (every block a contiguous slab)
// Prints GB/s per layout; decides whether re-laying U/V out is worth a refactor.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_sweep scripts/ubench_sweep.hip && /tmp/ubench_sweep [M=10005190] [k=25]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                      \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

constexpr int TB = 256, VEC = 16, CH = TB * VEC;   // elements per block and vector

__device__ __forceinline__ void ld16(const float* __restrict__ p, float* r) {   // lane l: floats [256 i + 4 l, +4) of the wave's span
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(p + i * 256);
    r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
  }
}

// MODE 0: A   1: B   2: C   3: S
template <int MODE>
__global__ __launch_bounds__(TB) void k_sweep(const float* __restrict__ U, const float* __restrict__ V, long long ld, int k,
                                              float c0, float* __restrict__ out) {
  const long long e0 = ((long long)blockIdx.x * TB + (threadIdx.x & ~63)) * VEC + (threadIdx.x & 63) * 4;   // as elem0<16>()
  const long long in_blk = (long long)(threadIdx.x & ~63) * VEC + (threadIdx.x & 63) * 4;
  float a1[VEC], a2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) a1[i] = a2[i] = 0.f;
  for (int j = 0; j < k; ++j) {
    float u[VEC], v[VEC];
    if (MODE == 0) {
      ld16(U + (long long)j * ld + e0, u);
      ld16(V + (long long)j * ld + e0, v);
    } else if (MODE == 1) {
      const float* base = U + ((long long)j * gridDim.x + blockIdx.x) * 2 * CH;
      ld16(base + in_blk, u);
      ld16(base + CH + in_blk, v);
    } else if (MODE == 2) {
      ld16(U + (long long)j * 2 * ld + e0, u);
      ld16(U + (long long)j * 2 * ld + ld + e0, v);
    } else {
      const float* base = U + (long long)blockIdx.x * 2 * CH * k + (long long)j * 2 * CH;
      ld16(base + in_blk, u);
      ld16(base + CH + in_blk, v);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a1[i] = fmaf(c0, v[i], a1[i]);
      a2[i] = fmaf(-c0, u[i], a2[i]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s += a1[i] + a2[i];
  if (s == 123.456f) out[0] = s;   // keep the loads
}

template <int MODE>
static double run(const float* U, const float* V, long long ld, int k, int nblk, float* out, const char* name) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k_sweep<MODE><<<nblk, TB>>>(U, V, ld, k, 0.5f, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) k_sweep<MODE><<<nblk, TB>>>(U, V, ld, k, 0.5f, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 2.0 * k * (double)nblk * CH * 4;
  const double gbs = bytes / (ms / reps * 1e-3) / 1e9;
  printf("%-44s %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms / reps * 1e3, gbs, gbs / 8000.0);
  return gbs;
}

int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 10005190;
  const int k = argc > 2 ? atoi(argv[2]) : 25;
  const int nblk = (int)(M / CH);               // whole blocks only: no tails in the probe
  const long long ld = (long long)nblk * CH;
  const size_t bytes = (size_t)2 * k * ld * 4;
  float *U, *V, *out;
  CHECK(hipMalloc(&U, bytes));                   // one allocation serves every layout (B, C, S index it as a single array)
  V = U + (size_t)k * ld;
  CHECK(hipMalloc(&out, 16));
  CHECK(hipMemset(U, 0, bytes));
  printf("M = %lld floats per vector (%d blocks of %d), k = %d stored pairs, %.2f GB per sweep\n", ld, nblk, CH, k, bytes / 1e9);
  run<0>(U, V, ld, k, nblk, out, "A  U[j][M], V[j][M] (solver layout)");
  run<1>(U, V, ld, k, nblk, out, "B  UV[j][block][2][4096]");
  run<2>(U, V, ld, k, nblk, out, "C  UV[j][2][M]");
  run<3>(U, V, ld, k, nblk, out, "S  per-block contiguous slab");
  run<0>(U, V, ld, k, nblk, out, "A  again");
  return 0;
}
