// Micro-measurement (dev tool): v_mfma_f32_32x32x16_bf16 with K independent v_fma_f32 behind every MFMA in ONE wave's stream
// (order pinned by inline asm), one and two waves per SIMD, one accumulation chain or two alternating -- shader cycles per
// MFMA per SIMD from s_memtime.  Question behind it: csrc/onf_x32.hip runs 5 vector instructions per
// MFMA at 48 cycles per MFMA per SIMD; what does the bare instruction mix cost?
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma32_valu_coissue.hip -o tools/micro/mfma32_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FMA(R) "v_fma_f32 %" #R ", %" #R ", %[c], %" #R "\n\t"
#define MF0 "v_mfma_f32_32x32x16_bf16 %[acc0], %[a], %[b], %[acc0]\n\t"
#define MF1 "v_mfma_f32_32x32x16_bf16 %[acc1], %[a], %[b], %[acc1]\n\t"

// EXTRA (K = 5 only): 1 two satisfied s_waitcnt, 2 two s_add_u32, 3 two s_nop 0, 4 one ds_read_b128 that nobody waits for,
// 5 one v_sin_f32, 6 two ds_read_b64_tr_b16 that nobody waits for
template <int EXTRA>
__global__ __launch_bounds__(512) void kx(unsigned long long* cyc, float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = seed * i;
  __syncthreads();
  f32x16 acc0;
  for (int i = 0; i < 16; ++i) acc0[i] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u}, b = a;
  b[0] += threadIdx.x;
  float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = 0.1f * seed;
  const float c = 1.0001f;
  u32x4 l0 = {0, 0, 0, 0};
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  u32x2 l1 = {0, 0}, l2 = {0, 0};
  const unsigned addr = (unsigned)(size_t)(buf + 4 * (threadIdx.x & 63));
  int sreg = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#define BASE "v_mfma_f32_32x32x16_bf16 %[acc0], %[a], %[b], %[acc0]\n\tv_fma_f32 %[v0], %[v0], %[c], %[v0]\n\tv_fma_f32 %[v1], %[v1], %[c], %[v1]\n\tv_fma_f32 %[v2], %[v2], %[c], %[v2]\n\tv_fma_f32 %[v3], %[v3], %[c], %[v3]\n\tv_fma_f32 %[v4], %[v4], %[c], %[v4]\n\t"
#define XOPS [acc0] "+v"(acc0), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [v4] "+v"(v4), [v5] "+v"(v5), [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [sr] "+s"(sreg) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [ad] "v"(addr)
      if (EXTRA == 0) asm volatile(BASE : XOPS);
      if (EXTRA == 1) asm volatile(BASE "s_waitcnt vmcnt(63) lgkmcnt(15)\n\ts_waitcnt vmcnt(63) lgkmcnt(15)\n\t" : XOPS);
      if (EXTRA == 2) asm volatile(BASE "s_add_u32 %[sr], %[sr], 1\n\ts_add_u32 %[sr], %[sr], 3\n\t" : XOPS);
      if (EXTRA == 3) asm volatile(BASE "s_nop 0\n\ts_nop 0\n\t" : XOPS);
      if (EXTRA == 4) asm volatile(BASE "ds_read_b128 %[l0], %[ad]\n\t" : XOPS);
      if (EXTRA == 5) asm volatile(BASE "v_sin_f32 %[v5], %[v5]\n\t" : XOPS);
      if (EXTRA == 6) asm volatile(BASE "ds_read_b64_tr_b16 %[l1], %[ad]\n\tds_read_b64_tr_b16 %[l2], %[ad] offset:1024\n\t" : XOPS);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + (float)sreg + (float)l0[0] + (float)l1[0] + (float)l2[1];
  for (int i = 0; i < 16; ++i) s += acc0[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int EXTRA>
static void runx(unsigned long long* cyc, float* out, int threads, const char* what) {
  const int iters = 5000, nw = 256 * threads / 64;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((kx<EXTRA>), dim3(256), dim3(threads), 0, 0, cyc, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nw);
  (void)hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per_wave = (double)h[nw / 2] / (iters * 4.0);
  printf("  5 fma + %-44s: %6.1f cycles per MFMA per wave, %6.1f per MFMA per SIMD\n", what, per_wave, per_wave / (threads / 256.0));
}

template <int K, int NACC, int LDSR>
__global__ __launch_bounds__(512) void k(unsigned long long* cyc, float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = seed * i;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a = {0x3f803f80u, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u}, b = a;
  b[0] += threadIdx.x;
  float v[10];
  for (int i = 0; i < 10; ++i) v[i] = seed + i;
  const float c = 1.0001f;
  u32x4 l0 = {0, 0, 0, 0};
  const unsigned addr = (unsigned)(size_t)(buf + 4 * (threadIdx.x & 63));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#define OPS [acc0] "+v"(acc0), [acc1] "+v"(acc1), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]) : [a] "v"(a), [b] "v"(b), [c] "v"(c)
      // operands 2..11 are the filler registers
      if (LDSR) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(addr));
      if ((m & 1) == 0 || NACC == 1) {
        if (K == 0) asm volatile(MF0 : OPS);
        if (K == 2) asm volatile(MF0 FMA(2) FMA(3) : OPS);
        if (K == 4) asm volatile(MF0 FMA(2) FMA(3) FMA(4) FMA(5) : OPS);
        if (K == 5) asm volatile(MF0 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) : OPS);
        if (K == 6) asm volatile(MF0 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) : OPS);
        if (K == 8) asm volatile(MF0 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) : OPS);
        if (K == 10) asm volatile(MF0 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11) : OPS);
      } else {
        if (K == 0) asm volatile(MF1 : OPS);
        if (K == 2) asm volatile(MF1 FMA(2) FMA(3) : OPS);
        if (K == 4) asm volatile(MF1 FMA(2) FMA(3) FMA(4) FMA(5) : OPS);
        if (K == 5) asm volatile(MF1 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) : OPS);
        if (K == 6) asm volatile(MF1 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) : OPS);
        if (K == 8) asm volatile(MF1 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) : OPS);
        if (K == 10) asm volatile(MF1 FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11) : OPS);
      }
      if (LDSR) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %1" : "+v"(b[1]) : "v"(l0[0]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 10; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)l0[1];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int NACC, int LDSR>
static void run(unsigned long long* cyc, float* out, int threads) {
  const int iters = 5000, nw = 256 * threads / 64;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<K, NACC, LDSR>), dim3(256), dim3(threads), 0, 0, cyc, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nw);
  (void)hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per_wave = (double)h[nw / 2] / (iters * 4.0);         // cycles per MFMA of ONE wave's stream
  const double per_simd = per_wave / (threads / 256.0);              // two waves share the SIMD
  printf("  K=%2d fma per MFMA, %d chain(s)%s: %6.1f cycles per MFMA per wave, %6.1f per MFMA per SIMD\n", K, NACC,
         LDSR ? ", + ds_read_b128 & wait" : "", per_wave, per_simd);
}

int main() {
  unsigned long long* cyc; float* out;
  (void)hipMalloc(&cyc, 256 * 8 * 8); (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    printf("%d wave(s) per SIMD:\n", threads / 256);
    run<0, 1, 0>(cyc, out, threads); run<2, 1, 0>(cyc, out, threads); run<4, 1, 0>(cyc, out, threads); run<5, 1, 0>(cyc, out, threads);
    run<6, 1, 0>(cyc, out, threads); run<8, 1, 0>(cyc, out, threads); run<10, 1, 0>(cyc, out, threads);
    run<5, 2, 0>(cyc, out, threads); run<8, 2, 0>(cyc, out, threads);
    run<5, 1, 1>(cyc, out, threads);
    runx<0>(cyc, out, threads, "nothing"); runx<1>(cyc, out, threads, "2 satisfied s_waitcnt");
    runx<2>(cyc, out, threads, "2 s_add_u32"); runx<3>(cyc, out, threads, "2 s_nop 0");
    runx<4>(cyc, out, threads, "1 ds_read_b128 (not waited for)"); runx<5>(cyc, out, threads, "1 v_sin_f32");
    runx<6>(cyc, out, threads, "2 ds_read_b64_tr_b16 (not waited for)");
  }
  return 0;
}
