// Issue-rate microbenchmark for gfx950 (MI355X): how many cycles a SIMD spends per wave64 instruction of the kinds
// the tile kernels are made of, as a function of resident waves per SIMD, and whether the matrix pipe co-issues with
// packed VALU work of other waves.  Decides what "VALU-bound" means for k_f_tile.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_valu scripts/ubench_valu.hip && ./scripts/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));             \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int ITERS = 2000;

// mode 0: 32 x v_fma_f32 (VGPR operands)      1: 16 x v_pk_fma_f32 (VGPR)     2: 16 x v_pk_fma_f32 (SGPR src0)
// mode 3: 4 x v_mfma_f32_16x16x4_f32          4: waves alternate: even = mode 3, odd = mode 2
// mode 5: per iteration 2 MFMA + 8 pk_fma in ONE wave (does a wave's own MFMA hide its VALU?)
// mode 6: 16 x v_pk_add_f32                   7: 32 x v_add_f32
template <int MODE>
__global__ __launch_bounds__(256) void k_issue(float* out, long long* cyc, float s0, float s1) {
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = (float)threadIdx.x * 1e-3f + i;
  v2f p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const v2f sv = (v2f){s0, s1};
  const int wave = threadIdx.x >> 6;
  const bool mf = MODE == 3 || (MODE == 4 && (wave & 1) == 0);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s0));
    } else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
    } else if (MODE == 1) {
      v2f q = sv;
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(q));
    } else if (MODE == 6) {
      v2f q = sv;
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
    } else if (MODE == 2 || (MODE == 4 && !mf)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "s"(sv));
    } else if (mf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], acc[i], 0, 0, 0);
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "s"(sv));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) r += a[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) r += p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int per_iter, float* out, long long* cyc) {
  for (int wps : {1, 2, 3, 4, 5, 8}) {
    const int grid = 256 * wps;   // one 4-wave block = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_issue<MODE><<<grid, 256>>>(out, cyc, 1.0001f, 0.9999f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_issue<MODE><<<grid, 256>>>(out, cyc, 1.0001f, 0.9999f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= h.size();
    // cycles of wave lifetime per instruction, and SIMD cycles per instruction = that / waves per SIMD
    const double per_wave = avg / ((double)ITERS * per_iter);
    printf("%-34s waves/SIMD %d  wall %.3f ms  counter ticks/instr/wave %.2f  -> ticks per instr per SIMD %.2f  (wall: %.2f ns per instr per SIMD)\n",
           name, wps, ms, per_wave, per_wave / wps, ms * 1e6 / ((double)ITERS * per_iter * wps));
  }
}

int main() {
  float* out;
  long long* cyc;
  CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
  CHECK(hipMalloc(&cyc, (size_t)256 * 8 * 4 * 8));
  run<0>("v_fma_f32 x32", 32, out, cyc);
  run<7>("v_add_f32 x32", 32, out, cyc);
  run<1>("v_pk_fma_f32 x16 (vgpr)", 16, out, cyc);
  run<2>("v_pk_fma_f32 x16 (sgpr src0)", 16, out, cyc);
  run<6>("v_pk_add_f32 x16", 16, out, cyc);
  run<3>("v_mfma_f32_16x16x4_f32 x4", 4, out, cyc);
  run<4>("even waves mfma x4 / odd pk_fma x16", 10, out, cyc);
  run<5>("one wave: 2 mfma + 8 pk_fma", 10, out, cyc);
  return 0;
}
