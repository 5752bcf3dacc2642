// Micro-experiment (dev tool): does v_pk_fma_f32 honour op_sel / op_sel_hi on gfx950 the way the compiler assumes?
// Round 2 found hipcc fusing two scalar fmas into  v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]  (both lanes x b.hi) with a wrong
// LOW lane in onf_split.hip; this isolates the instruction: alone, after a VALU write of its operands, after a trans op,
// and with dst == src0 as in that loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const f32x2* a, const f32x2* b, const f32x2* c, f32x2* out) {
  const int i = threadIdx.x;
  f32x2 x = a[i], y = b[i], z = c[i], d0, d1, d2, d3;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d0) : "v"(x), "v"(y), "v"(z));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d1) : "v"(x), "v"(y), "v"(z));
  // operands written by the VALU instruction just before, dst == src0
  f32x2 t = x;
  asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_fma_f32 %0, %0, %2, %3 op_sel:[0,1,0]" : "+v"(t) : "v"(y), "v"(y), "v"(z));
  d2 = t;
  // operand pair assembled by two v_mov (as the compiler does) right before
  f32x2 u;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(u.x), "=&v"(u.y) : "v"(x.y), "v"(x.x));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d3) : "v"(u), "v"(y), "v"(z));
  out[4 * i + 0] = d0; out[4 * i + 1] = d1; out[4 * i + 2] = d2; out[4 * i + 3] = d3;
}

int main() {
  const int n = 64;
  std::vector<f32x2> a(n), b(n), c(n), o(4 * n);
  for (int i = 0; i < n; ++i) {
    a[i] = {1.0f + i, 100.0f + i}; b[i] = {2.0f, 3.0f + 0.5f * i}; c[i] = {0.25f, 0.5f};
  }
  f32x2 *da, *db, *dc, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(n), 0, 0, da, db, dc, dout);
  hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const f32x2 x = a[i], y = b[i], z = c[i];
    const f32x2 e0 = {fmaf(x.x, y.y, z.x), fmaf(x.y, y.y, z.y)};
    const f32x2 e1 = {fmaf(x.x, y.x, z.x), fmaf(x.y, y.x, z.y)};
    const f32x2 e2 = {fmaf(x.x * y.x, y.y, z.x), fmaf(x.y * y.y, y.y, z.y)};
    const f32x2 e3 = {fmaf(x.y, y.y, z.x), fmaf(x.x, y.y, z.y)};
    const f32x2 e[4] = {e0, e1, e2, e3};
    for (int k = 0; k < 4; ++k) {
      const f32x2 g = o[4 * i + k];
      if (g.x != e[k].x || g.y != e[k].y) {
        if (bad[k]++ == 0) printf("case %d lane %d: got (%g, %g) expected (%g, %g)\n", k, i, g.x, g.y, e[k].x, e[k].y);
      }
    }
  }
  printf("mismatches: op_sel:[0,1,0] %d | op_sel_hi:[1,0,1] %d | after pk_mul, dst==src0 %d | after v_mov pair %d\n", bad[0], bad[1],
         bad[2], bad[3]);
  return 0;
}
