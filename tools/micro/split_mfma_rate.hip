// Micro-measurement (dev tool): sustained rate of the bf16x3 split-precision MFMA scheme on gfx950.
//
// Models the L1-forward loop of the fused ONF kernel if fp32 operands were split into three bf16 levels
// (x = hi + mid + lo exactly) and the six significant partial products were issued on the bf16 matrix pipe:
// per k-block of 32 input features and per wave (NT = 2 tiles of 16 points):
//   VALU: 8 features per lane and tile (argument, range reduction, v_sin), 3-level split and packing
//   A operands (weights): hi, mid via ds_read_b128 from LDS, lo via a coalesced 16-byte global load (L2-resident)
//   7 hidden tiles x 6 products x NT MFMAs (v_mfma_f32_16x16x32_bf16)
// Variants switch the operand traffic and the VALU work on/off to see what co-executes with the matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512, HT = 7, KB = 7, NT = 2;

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float sin_hw(float x, float qq) {
  const float jm = fmaf(x, 0.159154943f, 12582912.0f);
  const float j = jm - 12582912.0f;
  float r = fmaf(j, -6.28318548202514648f, x);
  r = fmaf(j, 1.74845553e-07f, r);
  return __builtin_amdgcn_sinf(fmaf(r, 0.159154943f, qq));
}

// x[0..7] -> three packed bf16 fragments (truncation split: every level keeps the top 16 bits, residuals are exact)
__device__ __forceinline__ void split8(const float x[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned a = __float_as_uint(x[2 * p]), b = __float_as_uint(x[2 * p + 1]);
    hi[p] = __builtin_amdgcn_perm(b, a, 0x07060302);
    const float ra = x[2 * p] - __uint_as_float(a & 0xffff0000u), rb = x[2 * p + 1] - __uint_as_float(b & 0xffff0000u);
    const unsigned ua = __float_as_uint(ra), ub = __float_as_uint(rb);
    mid[p] = __builtin_amdgcn_perm(ub, ua, 0x07060302);
    const float la = ra - __uint_as_float(ua & 0xffff0000u), lb = rb - __uint_as_float(ub & 0xffff0000u);
    lo[p] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302);
  }
}

template <int VARIANT>
__global__ __launch_bounds__(THREADS, 2) void rate_kernel(const u32x4* __restrict__ lo_blob,
                                                           const u32x4* __restrict__ himid_blob,
                                                           const float* __restrict__ table, const float* pts,
                                                           float* out, int chunks) {
  constexpr bool LOADS = VARIANT & 1, VALU = VARIANT & 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* img = reinterpret_cast<u32x4*>(smem);                       // [2][KB][HT][64] lane-linear fragments
  float* tab = reinterpret_cast<float*>(img + 2 * KB * HT * 64);     // [KB*32][4] wx wy b q
  for (int k = threadIdx.x; k < 2 * KB * HT * 64; k += THREADS) img[k] = himid_blob[k];
  for (int k = threadIdx.x; k < KB * 32 * 4; k += THREADS) tab[k] = table[k];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  float total = 0.f;
  for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    float ux[NT], uy[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const int p = ((chunk * 8 + wave) * NT + tl) * 16 + (lane & 15);
      ux[tl] = pts[2 * (p & 0xffff)]; uy[tl] = pts[2 * (p & 0xffff) + 1];
    }
    f32x4 acc[NT][HT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) acc[tl][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      u32x4 bh[NT], bm[NT], bl[NT];
      if (VALU) {
        const float* e = tab + (kb * 32 + 8 * g) * 4;
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(e + 4 * j);
            f[j] = sin_hw(fmaf(t.x, ux[tl], fmaf(t.y, uy[tl], t.z)), t.w);
          }
          split8(f, bh[tl], bm[tl], bl[tl]);
        }
      } else {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          const unsigned v = __float_as_uint(ux[tl]) + kb;
          bh[tl] = u32x4{v, v, v, v}; bm[tl] = bh[tl]; bl[tl] = bh[tl];
        }
      }
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) {
        u32x4 ah, am, al;
        if (LOADS) {
          ah = img[(kb * HT + mt) * 64 + lane];
          am = img[((KB + kb) * HT + mt) * 64 + lane];
          al = lo_blob[(kb * HT + mt) * 64 + lane];
        } else {
          const unsigned v = 0x3f803f80u + mt;
          ah = u32x4{v, v, v, v}; am = ah; al = ah;
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          f32x4 c = acc[tl][mt];
          c = mfma(ah, bh[tl], c);
          c = mfma(ah, bm[tl], c);
          c = mfma(am, bh[tl], c);
          c = mfma(am, bm[tl], c);
          c = mfma(ah, bl[tl], c);
          c = mfma(al, bh[tl], c);
          acc[tl][mt] = c;
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) total += acc[tl][mt][0] + acc[tl][mt][1] + acc[tl][mt][2] + acc[tl][mt][3];
  }
  out[blockIdx.x * THREADS + threadIdx.x] = total;
}

template <int VARIANT>
static void run(const char* name, const u32x4* lo, const u32x4* hm, const float* tab, const float* pts, float* out) {
  const int chunks = 256 * 64;   // 64 chunks of 8 waves x 32 points per workgroup
  const size_t lds = size_t(2) * KB * HT * 64 * 16 + KB * 32 * 4 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<VARIANT>, dim3(256), dim3(THREADS), lds, 0, lo, hm, tab, pts, out, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  // per SIMD: 2 waves, each 64 chunks x KB x HT x 6 x NT MFMAs
  const double mfma_per_simd = 2.0 * 64 * KB * HT * 6 * NT;
  const double cyc = best * 1e-3 * 2.4e9;
  printf("%-34s %8.3f ms   %6.1f cycles per MFMA slot at 2.4 GHz   (%.0f points/us)\n", name, best, cyc / mfma_per_simd,
         256.0 * 64 * 8 * 32 / (best * 1e3));
}

int main() {
  const size_t nfrag = size_t(KB) * HT * 64;
  std::vector<unsigned> h(nfrag * 4 * 2), l(nfrag * 4);
  for (auto& v : h) v = 0x3c003c00u + (rand() & 0x00ff00ff);
  for (auto& v : l) v = 0x38003800u + (rand() & 0x00ff00ff);
  std::vector<float> tab(KB * 32 * 4), pts(2 * 65536);
  for (auto& v : tab) v = (rand() % 2000 - 1000) * 1e-3f;
  for (auto& v : pts) v = (rand() % 2000 - 1000) * 3e-3f;
  u32x4 *dlo, *dhm; float *dtab, *dpts, *dout;
  hipMalloc(&dlo, l.size() * 4); hipMalloc(&dhm, h.size() * 4); hipMalloc(&dtab, tab.size() * 4);
  hipMalloc(&dpts, pts.size() * 4); hipMalloc(&dout, 256 * THREADS * 4);
  hipMemcpy(dlo, l.data(), l.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dhm, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dpts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  run<0>("MFMA only (operands in registers)", dlo, dhm, dtab, dpts, dout);
  run<1>("+ A from LDS (2) and L2 (1)", dlo, dhm, dtab, dpts, dout);
  run<2>("+ features and split on VALU", dlo, dhm, dtab, dpts, dout);
  run<3>("+ both", dlo, dhm, dtab, dpts, dout);
  return 0;
}
