// Micro-measurement (dev tool): accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) on gfx950,
// and of a split-precision revolution reduction in front of it.  Prints max |error| against double precision.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_sin_rev(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_sinf(x[i]);
}
__global__ void k_cos_rev(const float* x, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_cosf(x[i]);
}
// sin(a) for a in radians: revolutions = a/(2 pi) computed as hi + lo, fract(hi) + lo -> v_sin
__global__ void k_sin_rad(const float* a, float* y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float INV_HI = 0.15915494309189535f;             // fp32(1/(2pi))
  const float INV_LO = (float)(0.15915494309189533576888 - (double)0.15915494309189535f);
  float v = a[i];
  float hi = v * INV_HI;
  float lo = fmaf(v, INV_HI, -hi);
  lo = fmaf(v, INV_LO, lo);
  float f = __builtin_amdgcn_fractf(hi) + lo;
  y[i] = __builtin_amdgcn_sinf(f);
}

int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n), hy(n);
  float *dx, *dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  for (int pass = 0; pass < 4; ++pass) {
    double lo = pass == 0 ? 0.0 : (pass == 1 ? -0.5 : -16.0), hi = pass == 0 ? 1.0 : (pass == 1 ? 0.5 : 16.0);
    if (pass == 3) { lo = -60; hi = 60; }
    for (int i = 0; i < n; ++i) hx[i] = (float)(lo + (hi - lo) * ((i + 0.37) / n));
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    if (pass < 3) {
      k_sin_rev<<<n / 256, 256>>>(dx, dy, n);
      hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
      double es = 0; for (int i = 0; i < n; ++i) es = fmax(es, fabs((double)hy[i] - sin(2 * M_PI * (double)hx[i])));
      k_cos_rev<<<n / 256, 256>>>(dx, dy, n);
      hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
      double ec = 0; for (int i = 0; i < n; ++i) ec = fmax(ec, fabs((double)hy[i] - cos(2 * M_PI * (double)hx[i])));
      printf("revolutions in [%g,%g]: max|v_sin - sin| = %.3e   max|v_cos - cos| = %.3e\n", lo, hi, es, ec);
    } else {
      k_sin_rad<<<n / 256, 256>>>(dx, dy, n);
      hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
      double es = 0; for (int i = 0; i < n; ++i) es = fmax(es, fabs((double)hy[i] - sin((double)hx[i])));
      printf("radians in [%g,%g] via split reduction + v_sin: max abs err = %.3e\n", lo, hi, es);
    }
  }
  return 0;
}
