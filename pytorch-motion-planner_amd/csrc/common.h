// Shared host/device helpers of libnfopp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "nfopp_hip.h"

namespace nfopp {

// ---- error plumbing (host) -------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define NFOPP_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      ::nfopp::set_error(__VA_ARGS__);    \
      return NFOPP_ERR_ARG;               \
    }                                     \
  } while (0)

#define NFOPP_HIP(call)                                          \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return ::nfopp::hip_fail(e__, #call); \
  } while (0)

// ---- per-device launch state (host) ------------------------------------------------------------------------------
// One process may drive several GPUs: the dynamic-LDS attribute of a kernel and the CU count belong to the CURRENT
// device's copy of the code object, so "already set" flags are kept per device (csrc/runtime.hip).
constexpr int MAX_DEVICES = 64;
int current_device();   // hipGetDevice, -1 on failure (error string set)
// Raises the dynamic-LDS limit of `kernel` on the current device the first time it is launched there.
// `flags` is the caller's static bool[MAX_DEVICES] for that kernel instantiation.
int ensure_dynamic_lds(const void* kernel, size_t bytes, bool* flags);
// Content version the caller registered for a parameter buffer on the current device (nfopp_onf_params_version), 0 = none.
unsigned long long onf_params_version_of(const float* params_dev);
void onf_params_invalidate(const float* params_dev);

// ---- ONF parameter buffer geometry (state_dict order, include/nfopp_hip.h) --------------------------------------
struct OnfGeom {
  int n_enc, n_sin, ang_dim, n_ang, fin, point_dim;
  int off_ang_b, off_ang_f, off_w1, off_b1, off_w2, off_b2, off_w3, off_b3, off_we, off_be;  // -1 = absent
  int n_params;
  float mean, sigma;
};

inline bool make_geom(const nfopp_onf_config* c, OnfGeom* g) {
  if (!c) return false;
  if (c->angle_dim < 0 || c->angle_dim > 16 || !(c->sigma != 0.0f)) return false;
  g->n_enc = c->use_cos ? 200 : 100;
  g->n_sin = 100;
  g->ang_dim = c->angle_dim;
  g->n_ang = 2 * c->angle_dim;
  g->fin = g->n_enc + g->n_ang;
  g->point_dim = c->angle_dim > 0 ? 3 : 2;
  g->mean = c->mean;
  g->sigma = c->sigma;
  int o = 0;
  g->off_ang_b = g->off_ang_f = -1;
  if (g->n_ang) {
    g->off_ang_b = o; o += g->n_ang;
    g->off_ang_f = o; o += g->n_ang;
  }
  g->off_w1 = o; o += NFOPP_HIDDEN * g->fin;
  g->off_b1 = o; o += NFOPP_HIDDEN;
  g->off_w2 = o; o += NFOPP_HIDDEN * NFOPP_HIDDEN;
  g->off_b2 = o; o += NFOPP_HIDDEN;
  g->off_w3 = o; o += NFOPP_HIDDEN + g->fin;
  g->off_b3 = o; o += 1;
  g->off_we = o; o += 2 * g->n_enc;
  g->off_be = -1;
  if (c->has_bias) { g->off_be = o; o += g->n_enc; }
  g->n_params = o;
  return true;
}

// ---- device math shared by the kernels ---------------------------------------------------------------------------
#define NFOPP_PI_F 3.14159274101257324f      /* fp32(pi)   */
#define NFOPP_PI_D 3.14159265358979323846    /* float64 pi (path post-processing) */
#define NFOPP_TWO_PI_F 6.28318548202514648f  /* fp32(2 pi) */

// nfop/torch_math.py:5-7: (a + pi) % (2 pi) - pi, remainder with the divisor's sign, all in fp32.
// r = x - floor(x / 2pi) * 2pi through one fma: the true remainder is exactly representable, so the fma returns the
// same correctly rounded value torch's fmod-then-adjust produces; the two fix-ups only fire when x / 2pi rounds
// across an integer (remainder within 1 ulp of 0 or 2 pi).
__device__ __forceinline__ float wrap_angle(float a) {
  const float x = a + NFOPP_PI_F;
  const float k = floorf(x * 0.159154943f);
  float r = fmaf(-k, NFOPP_TWO_PI_F, x);
  if (r < 0.0f) r += NFOPP_TWO_PI_F;
  if (r >= NFOPP_TWO_PI_F) r -= NFOPP_TWO_PI_F;
  return r - NFOPP_PI_F;
}

// a + b * c and p * a + q * b with every product and sum rounded on its own, as a chain of separate torch ops is on the
// CPU (hipcc would contract the plain expressions to fused multiply-adds)
// (HIP's __fmul_rn / __fadd_rn are plain * and + and contract like them: the pragma is what holds)
__device__ __forceinline__ float add_mul_unfused(float a, float b, float c) {
#pragma clang fp contract(off)
  const float p = b * c;
  return a + p;
}
__device__ __forceinline__ float mix_unfused(float p, float a, float q, float b) {
#pragma clang fp contract(off)
  const float pa = p * a;
  const float qb = q * b;
  return pa + qb;
}

// sin(x + q*pi/2) for q in Z: 3-term Cody-Waite reduction to [-pi/4, pi/4] + minimax polynomials
// (<= 1.5 ulp for |x| < 1e5; checked against float64 in tests/test_host_logic.py through an fp32 emulation).
__device__ __forceinline__ float sin_quadrant(float x, int q) {
  float j = rintf(x * 0.636619772f);
  float r = fmaf(j, -1.57079601e+00f, x);
  r = fmaf(j, -3.13916473e-07f, r);
  r = fmaf(j, -5.39030253e-15f, r);
  int n = (int)j + q;
  float s = r * r;
  float ps = 2.86567956e-6f;
  ps = fmaf(ps, s, -1.98559923e-4f);
  ps = fmaf(ps, s, 8.33338592e-3f);
  ps = fmaf(ps, s, -1.66666672e-1f);
  float sv = fmaf(ps, r * s, r);
  float pc = 2.44677067e-5f;
  pc = fmaf(pc, s, -1.38877297e-3f);
  pc = fmaf(pc, s, 4.16666567e-2f);
  pc = fmaf(pc, s, -5.0e-1f);
  float cv = fmaf(pc, s, 1.0f);
  float res = (n & 1) ? cv : sv;
  return __int_as_float(__float_as_int(res) ^ ((n & 2) << 30));
}

// Packed pair of sin(x + q*pi/2), q = 2*qh in {0, 1, 2, 3} (qh = "half turns of pi"): the transcendental of the
// fused ONF kernel, written on 2-vectors so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.
//   t = x/pi + qh;  j = rint(t) via the 1.5*2^23 magic add (its low mantissa bit is the parity of j);
//   r = x - (j - qh)*pi  in [-pi/2, pi/2] (fma with a hi/lo split of pi);  result = (-1)^j * sin(r),
// sin(r) = r + r^3 P(r^2), P minimax of degree 4 (max abs error 1.2e-7 for |x| < 400, checked in
// tests/test_host_logic.py through an fp32 emulation against float64).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// scalar form of the same routine (bitwise identical arithmetic per element)
__device__ __forceinline__ float sin_halfturns1(float x, float qh) {
  const float t = fmaf(x, 0.318309886f, qh);
  const float jm = t + 12582912.0f;
  const float jh = (jm - 12582912.0f) - qh;
  float r = fmaf(jh, -3.14159274101257324f, x);
  r = fmaf(jh, 8.74227766e-08f, r);
  r = __uint_as_float(__float_as_uint(r) ^ (__float_as_uint(jm) << 31));
  const float s = r * r;
  float p = fmaf(-2.3909535684651928e-08f, s, 2.7526637040864443e-06f);
  p = fmaf(p, s, -0.00019840894674416631f);
  p = fmaf(p, s, 0.008333330973982811f);
  p = fmaf(p, s, -0.1666666716337204f);
  return fmaf(p, r * s, r);
}

// Hardware path: exact Cody-Waite reduction of x + qh*pi modulo 2 pi (fma with a hi/lo split of 2 pi), then
// v_sin_f32 on the remainder expressed in revolutions (|f| <= 0.5).  Measured on gfx950 (tools/micro/
// vsin_accuracy.hip): max |v_sin_f32(f) - sin(2 pi f)| = 1.25e-7; with the 3e-8-revolution rounding of f the
// feature error stays <= 3e-7 absolute for any |x| < 1e5, at 5 VALU + 1 transcendental op instead of 15 VALU.
__device__ __forceinline__ float sin_halfturns_hw(float x, float qq) {  // qq = offset in REVOLUTIONS (q/4)
  const float jm = fmaf(x, 0.159154943f, 12582912.0f);  // 1.5*2^23 + x/2pi: the add rounds to the nearest turn
  const float j = jm - 12582912.0f;                      // whole turns, exact
  float r = fmaf(j, -6.28318548202514648f, x);           // x - j * 2pi  in [-pi, pi] (product exact in the fma)
  r = fmaf(j, 1.74845553e-07f, r);
  return __builtin_amdgcn_sinf(fmaf(r, 0.159154943f, qq));  // |revolutions| <= 0.75
}

#ifndef NFOPP_TRIG_MODE
#define NFOPP_TRIG_MODE 2  /* 0 = scalar polynomial, 1 = packed polynomial, 2 = reduction + v_sin_f32 */
#endif
// sin_halfturns2(x, q * NFOPP_Q_UNIT) = sin(x + q*pi/2): the quadrant offset is pre-scaled to the unit the
// selected routine works in (half turns of pi for the polynomials, revolutions for v_sin_f32)
#if NFOPP_TRIG_MODE == 2
#define NFOPP_Q_UNIT 0.25f
#else
#define NFOPP_Q_UNIT 0.5f
#endif

__device__ __forceinline__ f32x2 sin_halfturns2(f32x2 x, f32x2 qh) {
#if NFOPP_TRIG_MODE == 0
  return f32x2{sin_halfturns1(x.x, qh.x), sin_halfturns1(x.y, qh.y)};
#elif NFOPP_TRIG_MODE == 2
  return f32x2{sin_halfturns_hw(x.x, qh.x), sin_halfturns_hw(x.y, qh.y)};
#endif
  const f32x2 t = fma2(x, splat2(0.318309886f), qh);
  const f32x2 jm = t + splat2(12582912.0f);
  const f32x2 jh = (jm - splat2(12582912.0f)) - qh;
  f32x2 r = fma2(jh, splat2(-3.14159274101257324f), x);
  r = fma2(jh, splat2(8.74227766e-08f), r);
  r.x = __uint_as_float(__float_as_uint(r.x) ^ (__float_as_uint(jm.x) << 31));
  r.y = __uint_as_float(__float_as_uint(r.y) ^ (__float_as_uint(jm.y) << 31));
  const f32x2 s = r * r;
  f32x2 p = fma2(splat2(-2.3909535684651928e-08f), s, splat2(2.7526637040864443e-06f));
  p = fma2(p, s, splat2(-0.00019840894674416631f));
  p = fma2(p, s, splat2(0.008333330973982811f));
  p = fma2(p, s, splat2(-0.1666666716337204f));
  return fma2(p, r * s, r);
}

// Philox4x32-10, first output word -> uniform [0,1) with 24 random bits (the same u32->float map torch uses).
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned long long ctr_lo,
                                                unsigned long long ctr_hi) {
  unsigned int c0 = (unsigned int)ctr_lo, c1 = (unsigned int)(ctr_lo >> 32);
  unsigned int c2 = (unsigned int)ctr_hi, c3 = (unsigned int)(ctr_hi >> 32);
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace nfopp
