// Micro-experiment (dev tool): the residual of a bf16 level by v_dot2c_f32_bf16 instead of v_and + v_sub.
// Exact split x = hi + mid + lo: level = top 16 bits of the running residual (v_perm packs the levels of two values),
// residual r = x - level.  One v_dot2c_f32_bf16 computes r0 = x0 + (-1) * level0 + 0 * level1 from the PACKED word, so a pair
// costs perm + 2 dot2 = 3 instructions per level instead of and, and, perm, sub, sub = 5.  Checks bit-exactness against the
// mask-and-subtract form on random and edge-case inputs, and times both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_ref(float x0, float x1, unsigned (&w)[3]) {
  for (int l = 0; l < 3; ++l) {
    const unsigned t0 = __float_as_uint(x0) & 0xffff0000u, t1 = __float_as_uint(x1) & 0xffff0000u;
    w[l] = __builtin_amdgcn_perm(t1, t0, 0x07060302);
    x0 -= __uint_as_float(t0); x1 -= __uint_as_float(t1);
  }
}
__device__ __forceinline__ void split_dot(float x0, float x1, unsigned (&w)[3]) {
  // the selectors go through registers: hipcc 7.2 encodes (-1.0 | 0) as the INLINE constant -1.0, which the instruction
  // reads as the 32-bit pattern 0xbf800000 = (0 | -1.0) -- the wrong half (seen on MI355X: r0 came out as x0 - level1)
  unsigned c0 = 0x0000bf80u, c1 = 0xbf800000u;
  asm volatile("" : "+v"(c0), "+v"(c1));
  const bf16x2 m0 = __builtin_bit_cast(bf16x2, c0), m1 = __builtin_bit_cast(bf16x2, c1);
  for (int l = 0; l < 3; ++l) {
    w[l] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302);
    if (l < 2) {
      const bf16x2 a = __builtin_bit_cast(bf16x2, w[l]);
      x0 = __builtin_amdgcn_fdot2_f32_bf16(a, m0, x0, false);
      x1 = __builtin_amdgcn_fdot2_f32_bf16(a, m1, x1, false);
    }
  }
}
__global__ void check(const float* x, int n, unsigned* bad, unsigned* first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned a[3], b[3];
  split_ref(x[2 * i], x[2 * i + 1], a);
  split_dot(x[2 * i], x[2 * i + 1], b);
  if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) {
    if (atomicAdd(bad, 1u) == 0) *first = (unsigned)i;
  }
}
template <int FORM>
__global__ void rate(float* out, int iters) {
  float x[16];
  for (int k = 0; k < 16; ++k) x[k] = 1.0f + k + threadIdx.x * 1e-3f;
  unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // eight independent pairs per iteration
      unsigned w[3];
      if (FORM == 0) split_ref(x[2 * k], x[2 * k + 1], w); else split_dot(x[2 * k], x[2 * k + 1], w);
      acc[k] = w[0] ^ w[1] ^ w[2];
      asm volatile("" : "+v"(acc[k]), "+v"(x[2 * k]), "+v"(x[2 * k + 1]));
    }
  }
  unsigned r = 0;
  for (int k = 0; k < 8; ++k) r ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(r);
}

int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned bits = s;
    if ((i & 7) == 0) bits = (s & 0x807fffffu) | ((100u + (s >> 24) % 60u) << 23);   // moderate exponents
    float f; memcpy(&f, &bits, 4);
    if (!std::isfinite(f)) f = 1.5f;
    if (std::fabs(f) < 1e-30f) f = (i & 1) ? 3.0e-20f : -7.7e-3f;                      // no subnormal-range inputs here
    h[i] = f;
  }
  h[0] = 0.f; h[1] = -0.f; h[2] = 1.f; h[3] = -1.f; h[4] = 3.4e38f; h[5] = 1.17549435e-38f; h[6] = 1e-37f; h[7] = 65504.f;
  float *dx, *dout; unsigned *dbad;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, 256 * 1024 * 4); hipMalloc(&dbad, 8);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 8);
  hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dbad, dbad + 1);
  unsigned res[2];
  hipMemcpy(res, dbad, 8, hipMemcpyDeviceToHost);
  printf("pairs checked %d, mismatching %u", n / 2, res[0]);
  if (res[0]) printf(" (first pair %u: %g %g)", res[1], h[2 * res[1]], h[2 * res[1] + 1]);
  printf("\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; ++form) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (form == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, dout, 1000);
      else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, dout, 1000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.3f ms for 1024 x 256 threads x 8000 pair-splits (11 vs 7 instructions each)\n", form ? "perm + dot2c" : "and + sub + perm", best);
  }
  return 0;
}
