// Micro-experiment (dev tool): is  v_mfma_f32_32x32x2_f32  D = C + A[:,0] B[0,:] + A[:,1] B[1,:]  bit for bit the fp32 chain
// fmaf(a1, b1, fmaf(a0, b0, c))  (k = 0 first)?  If so the encoding arguments  w_y u_y + b, then + w_x u_x  of K1's feature
// evaluation can come off the matrix pipe with the vector ALU's own rounding ("what is left").
// Operands: lane l holds A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31]; D[i] of lane l = row 8 (i >> 2) + 4 (l >> 5) + (i & 3), col l & 31.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f32_chain.hip -o tools/micro/mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, const float* C, float* D, int ntiles) {
  const int lane = threadIdx.x & 63, t = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const float a = A[(t * 64) + lane], b = B[(t * 64) + lane];
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = C[(t * 64 + lane) * 16 + i];
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[(t * 64 + lane) * 16 + i] = c[i];
}

int main() {
  const int nt = 4096;
  std::vector<float> A(nt * 64), B(nt * 64), C(nt * 64 * 16), D(nt * 64 * 16);
  srand(7);
  auto rnd = [](float s) { return s * (2.0f * rand() / RAND_MAX - 1.0f); };
  for (auto& x : A) x = rnd(3.0f);          // weights ~ N(0,1)-ish
  for (auto& x : B) x = rnd(40.0f);         // scaled coordinates
  for (auto& x : C) x = rnd(3.14159f);      // encoding bias
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nt / 4), dim3(256), 0, 0, dA, dB, dC, dD, nt);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  long long n = 0, bad01 = 0, bad10 = 0, badsum = 0;
  for (int t = 0; t < nt; ++t)
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 16; ++i) {
        const int row = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), col = l & 31;
        const float a0 = A[t * 64 + row], a1 = A[t * 64 + 32 + row], b0 = B[t * 64 + col], b1 = B[t * 64 + 32 + col];
        const float c = C[(t * 64 + l) * 16 + i], d = D[(t * 64 + l) * 16 + i];
        const float k01 = fmaf(a1, b1, fmaf(a0, b0, c)), k10 = fmaf(a0, b0, fmaf(a1, b1, c));
        const float ex = (float)((double)c + (double)a0 * b0 + (double)a1 * b1);
        ++n; bad01 += d != k01; bad10 += d != k10; badsum += d != ex;
      }
  printf("%lld results: differ from fmaf(a1,b1,fmaf(a0,b0,c)) in %lld, from fmaf(a0,b0,fmaf(a1,b1,c)) in %lld, from the correctly rounded sum in %lld\n",
         n, bad01, bad10, badsum);
  return 0;
}
