// Micro-measurement (dev tool): how many independent vector instructions fit behind one v_mfma_f32_16x16x32_bf16 for
// free on gfx950, when they are interleaved in ONE wave's instruction stream (order pinned by inline asm) -- with one
// and with two waves per SIMD.  Prints cycles per MFMA for K = 0..5 v_fma_f32 behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(512) void k_coissue(float* out, int iters, float seed) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  b[0] += threadIdx.x;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk = {seed, seed};
  float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, c = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      // one MFMA on accumulator m & 3, then K independent fmas, all in one asm block: the order is exactly this
      if (K == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
      if (K == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_fma_f32 %1, %1, %4, %1" : "+v"(acc[m & 3]), "+v"(v0) : "v"(a), "v"(b), "v"(c));
      if (K == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_fma_f32 %1, %1, %5, %1\n v_fma_f32 %2, %2, %5, %2" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1) : "v"(a), "v"(b), "v"(c));
      if (K == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_fma_f32 %1, %1, %6, %1\n v_fma_f32 %2, %2, %6, %2\n v_fma_f32 %3, %3, %6, %3" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1), "+v"(v2) : "v"(a), "v"(b), "v"(c));
      if (K == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n v_fma_f32 %1, %1, %7, %1\n v_fma_f32 %2, %2, %7, %2\n v_fma_f32 %3, %3, %7, %3\n v_fma_f32 %4, %4, %7, %4" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b), "v"(c));
      if (K == 5) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %6, %7, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4) : "v"(a), "v"(b), "v"(c));
      if (K == 6) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_perm_b32 %1, %1, %2, %5\n v_perm_b32 %2, %2, %1, %5" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1) : "v"(a), "v"(b), "s"(0x07060302));
      if (K == 7) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n s_waitcnt lgkmcnt(7)\n v_perm_b32 %1, %1, %2, %5\n v_perm_b32 %2, %2, %1, %5" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1) : "v"(a), "v"(b), "s"(0x07060302));
      if (K == 8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_and_b32 %1, 0xffff0000, %1\n v_sub_f32 %2, %2, %1" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1) : "v"(a), "v"(b));
      if (K == 9) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(acc[m & 3]), "+v"(pk) : "v"(a), "v"(b), "v"(c));
      if (K == 10) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_sin_f32 %1, %1\n v_fma_f32 %2, %2, %5, %2" : "+v"(acc[m & 3]), "+v"(v0), "+v"(v1) : "v"(a), "v"(b), "v"(c));
    }
  }
  float s = v0 + v1 + v2 + v3 + v4 + pk.x + pk.y;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
static void run(float* out, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_coissue<K>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double waves_per_simd = threads / 256.0;
  const double mfma_per_simd = waves_per_simd * iters * 8.0;
  printf("  K=%d vector instr per MFMA: %7.3f ms  -> %5.1f cycles per MFMA at 2.4 GHz\n", K, best,
         best * 1e-3 * 2.4e9 / mfma_per_simd);
}


template <int NACC>
__global__ __launch_bounds__(512) void k_chain(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  b[0] += threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
  }
  float s = seed;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run_chain(float* out, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("  %d independent accumulators per wave: %5.1f cycles per MFMA at 2.4 GHz\n", NACC,
         best * 1e-3 * 2.4e9 / ((threads / 256.0) * iters * 8.0));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    printf("%d waves per SIMD:\n", threads / 256);
    run<0>(out, threads); run<1>(out, threads); run<2>(out, threads); run<3>(out, threads);
    printf("  dependent MFMA chains (no vector work):\n");
    run_chain<1>(out, threads); run_chain<2>(out, threads); run_chain<4>(out, threads); run_chain<8>(out, threads);
    run<4>(out, threads); run<5>(out, threads);
    printf("  (6: 2 v_perm  7: s_waitcnt + 2 v_perm  8: v_and + v_sub  9: 1 v_pk_fma_f32  10: v_sin + v_fma)\n");
    run<6>(out, threads); run<7>(out, threads); run<8>(out, threads); run<9>(out, threads); run<10>(out, threads);
  }
  return 0;
}
