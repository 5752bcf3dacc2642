// Fused ONF forward + input-gradient kernel for gfx950 with every GEMM issued on the bf16 matrix pipe as an EXACT
// three-level split of the fp32 operands ("bf16x3").
//
// Why: on gfx950 the fp32 MFMA shares the vector ALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for onf_fused.hip: matrix and
// vector work serialise), and runs at 64 FLOP/clk/SIMD.  v_mfma_f32_16x16x32_bf16 runs at 1024 FLOP/clk/SIMD on its own
// pipe and leaves half of its 16 cycles to vector issue.  Every fp32 value x is split without error into
//      x = hi + mid + lo,   each level the top 16 bits (a bf16) of the running residual (8 significant bits a level),
// and a product a*b is accumulated in fp32 as  a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1.  The three dropped partial
// products are below 2^-24 |ab| -- smaller than the rounding of an fp32 multiply -- so the result is fp32-faithful
// (measured: 10x closer to float64 than a sequential fp32 dot product, tools/micro + tests/test_gpu_split_path.py),
// at 6/16 of the fp32 MFMA time.
//
// What changes against onf_fused.hip (same points-on-N / features-on-M orientation, same LDS images and layouts P/Q,
// same accumulator-as-next-operand chaining, same feature evaluation and epilogues):
//  * an LDS weight word holds (bf16 hi | bf16 mid) of the weight instead of its fp32 value: the images keep their
//    size, their [row][col] addressing and therefore every conflict-free read pattern of the fp32 kernel;
//  * the third level does not fit in LDS (6 bytes per weight): a small prep kernel writes it to a per-device blob in
//    exactly the order the lanes consume it (16 bytes per lane and step, coalesced, L2-resident, 160 KB per ONF);
//  * one MFMA step covers 32 k values = 8 per lane group: the lane gathers its 8 weight words with ds_read_b32, packs
//    the hi and the mid fragments with 8 v_perm_b32, fetches the lo fragment with one global load, and issues the 6
//    products per point tile; activations are split by the vector ALU (5.5 instructions per element) which now
//    overlaps the matrix pipe.
// k blocks pair the accumulator tiles (2kb, 2kb+1): element j of a lane's fragment is register j&3 of tile 2kb+(j>>2).
#include <stdlib.h>

#include <mutex>

#include "onf_layout.h"

namespace nfopp {

// Fences between MFMA steps pin the MEMORY instructions of a step (so that hipcc does not hoist every load of the
// unrolled loops to the top and spill) but let vector, scalar and matrix instructions cross, so that the packing and
// splitting of the next step can be interleaved with the MFMAs of this one.
#ifndef NFOPP_FENCE
#define NFOPP_FENCE 0x40E   /* may cross: VALU 0x2 | SALU 0x4 | MFMA 0x8 | transcendental 0x400 */
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfmab(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc += A * B with both operands in three levels (the six partial products above 2^-24)
__device__ __forceinline__ f32x4 mfma6(const u32x4& ah, const u32x4& am, const u32x4& al, const u32x4& bh,
                                       const u32x4& bm, const u32x4& bl, f32x4 c) {
  c = mfmab(al, bh, c);   // smallest terms first
  c = mfmab(ah, bl, c);
  c = mfmab(am, bm, c);
  c = mfmab(am, bh, c);
  c = mfmab(ah, bm, c);
  c = mfmab(ah, bh, c);
  return c;
}

// 8 fp32 values -> three packed bf16 fragments (element j in half-word j)
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

// The last hidden k block holds only the four hidden units 96..99 (one per lane group): instead of six 32-deep bf16
// products with 28 empty k slots it is issued as ONE fp32 MFMA 16x16x4 on the exact fp32 weight, rebuilt from its
// (hi | mid) word and the third level of the blob fragment (hi + mid is exact in fp32, + lo gives the weight back).
__device__ __forceinline__ float rebuild_weight(float word, unsigned lo_half) {
  const unsigned w = __float_as_uint(word);
  return (__uint_as_float(w & 0xffff0000u) + __uint_as_float(w << 16)) + __uint_as_float(lo_half << 16);
}
constexpr int TAIL_KB = 3;   // hidden k block issued on the fp32 matrix instruction

// 8 (hi | mid) weight words -> the hi and the mid fragment
__device__ __forceinline__ void pack_words(const float w[8], u32x4& hi, u32x4& mid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned a = __float_as_uint(w[2 * p]), b = __float_as_uint(w[2 * p + 1]);
    hi[p] = __builtin_amdgcn_perm(b, a, 0x07060302);
    mid[p] = __builtin_amdgcn_perm(b, a, 0x05040100);
  }
}

// One MFMA step (6 partial products per point tile) with the 8 packing permutes of the NEXT step placed between the
// MFMAs -- two behind every third MFMA (NT = 2) -- and a full scheduling fence after each MFMA so that hipcc keeps this
// order: behind a v_mfma_f32_16x16x32_bf16 two simple vector instructions issue for free (tools/micro/
// mfma_valu_coissue.hip).  ACC_(tl) names the accumulator of point tile tl; WN_ are the raw (hi | mid) words of the next
// step, AHN_/AMN_ receive its packed fragments.
#define NFOPP_NOWORK(q_)
#define NFOPP_STEP(ACC_, BH_, BM_, BL_, AH_, AM_, AL_, PACK_, WN_, AHN_, AMN_) \
  NFOPP_STEP_W(ACC_, BH_, BM_, BL_, AH_, AM_, AL_, PACK_, WN_, AHN_, AMN_, NFOPP_NOWORK)
// ... and WORK_(q) behind MFMA number q of the step (q = 0 .. 6 NT - 1) when that slot carries no permutes
#define NFOPP_STEP_W(ACC_, BH_, BM_, BL_, AH_, AM_, AL_, PACK_, WN_, AHN_, AMN_, WORK_)                       \
  _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) {                                                          \
    _Pragma("unroll") for (int tl_ = 0; tl_ < NT; ++tl_) {                                                     \
      const u32x4 a_ = k_ == 0 ? AL_ : ((k_ == 2 || k_ == 3) ? AM_ : AH_);                                     \
      const u32x4 b_ = k_ == 1 ? BL_[tl_] : ((k_ == 2 || k_ == 4) ? BM_[tl_] : BH_[tl_]);                      \
      ACC_(tl_) = mfmab(a_, b_, ACC_(tl_));                                                                    \
      const int q_ = k_ * NT + tl_;                                                                            \
      const int p_ = NT == 2 ? (q_ % 3 == 1 ? q_ / 3 : -1) : (q_ == 0 || q_ == 1 ? q_ : (q_ == 3 || q_ == 4 ? q_ - 1 : -1)); \
      if ((PACK_) && p_ >= 0 && p_ < 4) {                                                                      \
        const unsigned x_ = __float_as_uint(WN_[2 * p_]), y_ = __float_as_uint(WN_[2 * p_ + 1]);              \
        AHN_[p_] = __builtin_amdgcn_perm(y_, x_, 0x07060302);                                                  \
        AMN_[p_] = __builtin_amdgcn_perm(y_, x_, 0x05040100);                                                  \
      }                                                                                                        \
      if (p_ < 0) { WORK_(q_) }                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  }

// Development build (make EXTRA=-DNFOPP_PHASE_PROFILE): wave 0 of every workgroup accumulates clock ticks per phase of
// the chunk loop; launch_split_t prints the shares every tenth launch of the <.., 2, 0> kernel.
#ifdef NFOPP_PHASE_PROFILE
#define NFOPP_TICK(SLOT)                                           \
  {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    phase_ticks[SLOT] += (float)(now_ - phase_t0);                 \
    phase_t0 = now_;                                               \
  }
#else
#define NFOPP_TICK(SLOT)
#endif

// ---- third-level blob: [gemm][step][lane] 16 bytes --------------------------------------------------------------
template <int NKT>
struct Blob {
  static constexpr int NKB = (NKT + 1) / 2;   // input k blocks (pairs of input tiles)
  static constexpr int HKB = 4;               // hidden k blocks: tiles (0,1) (2,3) (4,5) (6,-)
  static constexpr int L1 = 0;                // step = kb * HT + mt
  static constexpr int L2 = L1 + NKB * HT;    // step = kb * HT + mt
  static constexpr int L2T = L2 + HKB * HT;   // step = kb * HT + mt
  static constexpr int L1T = L2T + HKB * HT;  // step = mt * HKB + kb
  static constexpr int STEPS = L1T + NKT * HKB;
  static constexpr size_t BYTES = size_t(STEPS + 2) * 64 * 16;   // + the two-step look-ahead past the last step
};

struct LaneGeom {
  int i, g, gi, ri, colP, colQ, rowposP, rowposQ;
  __device__ explicit LaneGeom(int lane) {
    i = lane & 15; g = lane >> 4; gi = i >> 2; ri = i & 3;
    colP = 16 * (g & 1) + 4 * (g >> 1); colQ = 8 * (g & 1) + 4 * (g >> 1);
    rowposP = 16 * (gi & 1) + 4 * (gi >> 1) + ri; rowposQ = 8 * (gi & 1) + 4 * (gi >> 1) + ri;
  }
};

// the hidden index a lane group contributes as k element j of hidden k block kb, under layout Q (h1) or P (h2); -1: none
__device__ __forceinline__ int hidden_k(int kb, int j, int g, bool layout_p, int colP, int colQ) {
  const int t = 2 * kb + (j >> 2), r = j & 3;
  if (t < 6) return layout_p ? base_p(t) + r + colP : 16 * t + r + colQ;
  return (t == 6 && r == 0) ? 96 + g : -1;
}

template <int NKT>
__global__ __launch_bounds__(64) void split_prep_kernel(const OnfGeom geo, const float* __restrict__ P, u32x4* blob) {
  using B = Blob<NKT>;
  const int step = blockIdx.x, lane = threadIdx.x;
  const LaneGeom q(lane);
  auto w1 = [&](int row, int col) { return (row >= 0 && col < geo.fin) ? P[geo.off_w1 + row * geo.fin + col] : 0.0f; };
  auto w2 = [&](int row, int col) { return (row >= 0 && col >= 0) ? P[geo.off_w2 + row * H + col] : 0.0f; };
  float w[8];
  if (step < B::L2) {                     // L1: a1 = W1 in
    const int kb = (step - B::L1) / HT, mt = (step - B::L1) % HT;
    const int row = mt < 6 ? 16 * mt + q.rowposQ : 96 + q.gi;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = w1(row, base_p(2 * kb + (j >> 2)) + q.colP + (j & 3));
  } else if (step < B::L2T) {             // L2: a2 = W2 h1
    const int kb = (step - B::L2) / HT, mt = (step - B::L2) % HT;
    const int row = mt < 6 ? base_p(mt) + q.rowposP : 96 + q.gi;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = w2(row, hidden_k(kb, j, q.g, false, q.colP, q.colQ));
  } else if (step < B::L1T) {             // L2T: dh1 = W2^T dh2
    const int kb = (step - B::L2T) / HT, mt = (step - B::L2T) % HT;
    const int col = mt < 6 ? 16 * mt + q.rowposQ : 96 + q.gi;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = w2(hidden_k(kb, j, q.g, true, q.colP, q.colQ), col);
  } else {                                // L1T: din = W1^T dh1
    const int mt = (step - B::L1T) / B::HKB, kb = (step - B::L1T) % B::HKB;
    const int col = base_p(mt) + q.rowposP;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = w1(hidden_k(kb, j, q.g, false, q.colP, q.colQ), col);
  }
  u32x4 out;
#pragma unroll
  for (int p = 0; p < 4; ++p) out[p] = lo_level(w[2 * p]) | (lo_level(w[2 * p + 1]) << 16);
  blob[step * 64 + lane] = out;
}

// MODE 0: forward + input gradient (planner step)   1: training pass (factors for the weight-gradient GEMMs, as
// onf_fused.hip's TRAIN mode: same stores, same record)   2: forward only (logits)
template <int NKT, int NT, int MODE>
__global__ __launch_bounds__(THREADS, THREADS / 256) void onf_split_kernel(const OnfKernelArgs a, const u32x4* __restrict__ blob) {
  constexpr bool FWD_ONLY = MODE == 2;
  constexpr bool TRAIN = MODE == 1;
  using L = Lds<NKT>;
  using B = Blob<NKT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds<NKT, TRAIN, true>(lds, a);
  __syncthreads();
#ifndef NFOPP_NO_PRIO_YOUNG
  // the second-dispatched half of the workgroup loses the vector-issue arbitration (age); one static priority for that
  // half, set once, evens it out (MI355X_MICROARCH.md "static priority for the younger half"; -0.4 % in a same-process A/B)
  if ((threadIdx.x >> 6) >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
#endif

  const OnfGeom& geo = a.geom;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, ri = i & 3;
  int gi = i >> 2;
  int colP = 16 * (g & 1) + 4 * (g >> 1), colQ = 8 * (g & 1) + 4 * (g >> 1);
  int rowposP = 16 * (gi & 1) + 4 * (gi >> 1) + ri, rowposQ = 8 * (gi & 1) + 4 * (gi >> 1) + ri;
  // Every LDS address below derives from these five lane constants.  Left to itself hipcc computes all ~40 addresses
  // once, in front of the persistent loop, and spills them; an empty asm makes them opaque at the start of each GEMM so
  // that they are re-derived (a few integer ops) where they are used.
#define NFOPP_REDERIVE() asm volatile("" : "+v"(colP), "+v"(colQ), "+v"(rowposP), "+v"(rowposQ), "+v"(gi))
  const int first_angle_kb = geo.n_ang ? geo.n_enc / 32 : B::NKB;   // first k block that can hold an angle feature

  const float* W1 = lds + L::W1;
  const float* W2 = lds + L::W2;
  constexpr int S1 = L::S1;
  constexpr int CH = WAVES * 16 * NT;
  const long long n_work = work_points(a);
  const long long n_chunks = (n_work + CH - 1) / CH;
  const float b3 = a.params[geo.off_b3];
  constexpr int WIN = 16 * NKT, WH = 16 * HT;   // row lengths of the stored factors (TRAIN)
  float loss_acc = 0.f;
  f32x4 g4_acc[HT];   // TRAIN: running sum_p rho_p * relu(a2_p) of this lane's (hidden row, point column) cells
#pragma unroll
  for (int mt = 0; mt < HT; ++mt) g4_acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // third-level fragments through a buffer resource: address = base (4 scalar registers, built once) + 16 * lane (ONE
  // vector register for the whole kernel) + the step's byte offset (a scalar operand) -- no per-step 64-bit vector
  // address, no FLAT load (which would count against lgkmcnt and stall every LDS wait on the global latency)
  const unsigned lane16 = lane * 16;
  const __amdgpu_buffer_rsrc_t blob_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(blob), 0, (int)B::BYTES, 0x00020000);
  auto lo_frag = [&](int step) __attribute__((always_inline)) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(blob_rsrc, lane16, step * 1024, 0));
  };

#ifdef NFOPP_PHASE_PROFILE
  float phase_ticks[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned long long phase_t0 = __builtin_readcyclecounter();
#endif
  for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // ---------------------------------------------------------------- sample / load the wave's points
    float ux[NT], uy[NT], th[NT];
    long long pidx[NT];
#ifndef NFOPP_NO_SHARED_SAMPLING
    if constexpr (NT == 2 && !TRAIN) {
      // the four lane groups of a point column would each draw and interpolate BOTH tiles' samples (Philox, wrap, lerp):
      // lane group g evaluates the sample of tile g & 1 once and the groups exchange the results
      const int mine = g & 1;
      float x, y, ang;
      const long long row = load_point(a, n_work, chunk * CH + (wave * NT + mine) * 16 + i, g >> 1, x, y, ang);
      const float uxm = (x - geo.mean) / geo.sigma, uym = (y - geo.mean) / geo.sigma;
      const int lo = (int)row, hi = (int)(row >> 32);
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        const int from = i + 16 * tl;   // a lane of group tl (g = tl: tile tl)
        ux[tl] = __shfl(uxm, from); uy[tl] = __shfl(uym, from); th[tl] = __shfl(ang, from);
        pidx[tl] = ((long long)__shfl(hi, from) << 32) | (unsigned)__shfl(lo, from);
      }
    } else
#endif
    {
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        float x, y, ang;
        pidx[tl] = load_point(a, n_work, chunk * CH + (wave * NT + tl) * 16 + i, g, x, y, ang);
        ux[tl] = (x - geo.mean) / geo.sigma;
        uy[tl] = (y - geo.mean) / geo.sigma;
        th[tl] = ang;
        if (TRAIN && g == 0 && pidx[tl] < a.n_points)
          *reinterpret_cast<f32x4*>(a.ws_u + pidx[tl] * 12) = f32x4{ux[tl], uy[tl], 1.0f, ang};
      }
    }

    NFOPP_TICK(0)   // sampling
    // ---------------------------------------------------------------- L1: a1 = W1 in + b1, features just-in-time
    NFOPP_REDERIVE();
    f32x4 acc1[NT][HT];
    float skip[NT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L::B1 + (mt < 6 ? 16 * mt + colQ : 96 + 4 * g));
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) acc1[tl][mt] = bias;
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) skip[tl] = 0.f;

    const float* w1a = W1 + rowposQ * S1 + colP;
    const float* w1b = w1a + 64 * S1;
    const float* w1c = W1 + (96 + gi) * S1 + colP;
    const float* ftl = lds + L::ft(colP);   // lane part of every feature-table address (+ L::ft_rel(block or tile base))
    const float* isl = lds + L::ISA + colP;
    const float* fcl = lds + L::fc(colP);   // the same for the compact table (+ L::fc_rel(...))
    const float* w3l = lds + L::W3B + colP;

    u32x4 q0 = lo_frag(B::L1), q1 = lo_frag(B::L1 + 1);   // third-level fragments of the next two steps
    // raw (hi | mid) words: wa1 = the NEXT step (packed between this step's MFMAs), wb1 = the step after (in flight);
    // ah1 / am1 = the packed fragments of the CURRENT step.  Steps run on over the k-block boundary.
    float wa1[8], wb1[8];
    u32x4 ah1, am1, ahn1 = {0, 0, 0, 0}, amn1 = {0, 0, 0, 0};
    auto fetch1 = [&](int kb, int mt, float (&dst)[8]) __attribute__((always_inline)) {
      const float* pa = w1a + 32 * kb;
      const float* pb = w1b + 32 * kb;
      const float* pc = w1c + 32 * kb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * (j >> 2) + (j & 3);
        dst[j] = mt < 4 ? pa[mt * 16 * S1 + c] : (mt < 6 ? pb[(mt - 4) * 16 * S1 + c] : pc[c]);
      }
    };
    fetch1(0, 0, wa1);
    pack_words(wa1, ah1, am1);
    fetch1(0, 1, wa1);
    auto l1_block = [&](auto ang_c, int kb) __attribute__((always_inline)) {
      constexpr bool ANG = decltype(ang_c)::value;
      const int off = 32 * kb;   // base_p(2 kb); the second tile of the pair sits 8 columns further
      // the lane's 8 features of this k block for every point tile: element j = feature off + 8 (j >> 2) + colP + (j & 3)
      u32x4 bh[NT], bm[NT], bl[NT];
      {
        float fv[NT][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float* fte = ftl + L::ft_rel(off + 8 * half);
          if (NT == 2) {
            const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
            f32x2 sk = {skip[0], skip[NT - 1]};
            f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
            if (ANG) isa4 = *reinterpret_cast<const f32x4*>(isl + off + 8 * half);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);
              const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);
              const f32x4 e2 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 8);
              const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
              const f32x2 qh = {e2.x, e2.y}, w3 = {e2.z, e2.w};
              const f32x2 v = features2<ANG, false>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
              sk = fma2(w3, v, sk);
              fv[0][4 * half + r] = v.x; fv[NT - 1][4 * half + r] = v.y;
            }
            skip[0] = sk.x; skip[NT - 1] = sk.y;
          } else {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const float* ea = fte + L::FTS * r;
              const float* eb = ea + L::FTS;
              const f32x2 ux2 = splat2(ux[0]), uy2 = splat2(uy[0]), th2 = splat2(th[0]);
              const f32x2 wx = {ea[0], eb[0]}, wy = {ea[2], eb[2]}, bb = {ea[4], eb[4]}, fr = {ea[6], eb[6]};
              const f32x2 qh = {ea[8], eb[8]}, isa = {isl[off + 8 * half + r], isl[off + 8 * half + r + 1]};
              const f32x2 v = features2<ANG, false>(wx, wy, bb, fr, qh, isa, ux2, uy2, th2);
              skip[0] = fmaf(ea[10], v.x, skip[0]);
              skip[0] = fmaf(eb[10], v.y, skip[0]);
              fv[0][4 * half + r] = v.x; fv[0][4 * half + r + 1] = v.y;
            }
          }
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) split8(fv[tl], bh[tl], bm[tl], bl[tl]);
      }
      const int lo_step = B::L1 + kb * HT;
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) {
        const u32x4 al = q0;
        q0 = q1;
        q1 = lo_frag(lo_step + mt + 2);   // runs on into the next k block (and, at the end, into L2's first steps)
        fetch1(mt + 2 < HT ? kb : kb + 1, (mt + 2) % HT, wb1);   // past the last block: a harmless in-image read
        __builtin_amdgcn_sched_barrier(0);
#define NFOPP_ACC(tl) acc1[tl][mt]
        NFOPP_STEP(NFOPP_ACC, bh, bm, bl, ah1, am1, al, true, wa1, ahn1, amn1)
#undef NFOPP_ACC
        ah1 = ahn1; am1 = amn1;
#pragma unroll
        for (int j = 0; j < 8; ++j) wa1[j] = wb1[j];
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (NT == 2 && B::NKB == 7) {
      // F = 200..224 at two tiles per wave: the features of block kb+1 (blocks 1..5: no angle features) are evaluated
      // and split IN THE MFMA SHADOWS of block kb -- one evaluation and one pair-split in flight at a time, cut into
      // single instructions and dealt five to a slot behind the MFMAs that carry no packing permutes (work item w of
      // 232: items 0..143 = evaluation w/9 (feature j = e>>1 of point tile e&1), step w%9; items 144..231 = split of
      // pair (w-144)/11, step (w-144)%11).  Same arithmetic, same order as l1_block's up-front evaluation.
      u32x4 bhc[NT], bmc[NT], blc[NT], bhn[NT], bmn[NT], bln[NT];
      float fvn[NT][8];
      float e_arg = 0.f, e_j = 0.f, e_r = 0.f, e_t = 0.f, e_v = 0.f;
      float s_ra = 0.f, s_rb = 0.f, s_la = 0.f, s_lb = 0.f;
      unsigned s_ta = 0, s_tb = 0;
      f32x4 tw[2];          // compact table entry of feature j (ping-pong on j & 1): (wx, wy, b, qh)
      float tz[2];          // its skip weight
      int nxt_off = 0;      // 32 (kb + 1): feature offset of the block being prepared
      auto table_load = [&](int j) __attribute__((always_inline)) {
        const int x = nxt_off + 8 * (j >> 2) + (j & 3);
        tw[j & 1] = *reinterpret_cast<const f32x4*>(fcl + L::fc_rel(x));
        tz[j & 1] = w3l[x];
      };
      auto work_item = [&](int w) __attribute__((always_inline)) {
        if (w < 144) {
          const int e = w / 9, u = w % 9, j = e >> 1, tl = e & 1, b = j & 1;
          if (u == 0) {
            e_arg = fmaf(tw[b].y, uy[tl], tw[b].z);
            if (tl == 0 && j + 1 < 8) table_load(j + 1);
          }
          if (u == 1) e_arg = fmaf(tw[b].x, ux[tl], e_arg);
          if (u == 2) e_t = fmaf(e_arg, 0.159154943f, 12582912.0f);
          if (u == 3) e_j = e_t - 12582912.0f;
          if (u == 4) e_r = fmaf(e_j, -6.28318548202514648f, e_arg);
          if (u == 5) e_r = fmaf(e_j, 1.74845553e-07f, e_r);
          if (u == 6) e_v = fmaf(e_r, 0.159154943f, tw[b].w);
          if (u == 7) e_v = __builtin_amdgcn_sinf(e_v);   // the argument is formed in e_v (source = destination)
          if (u == 8) { skip[tl] = fmaf(tz[b], e_v, skip[tl]); fvn[tl][j] = e_v; }
        } else if (w < 232) {
          const int sp = (w - 144) / 11, v = (w - 144) % 11, tl = sp >> 2, pp = sp & 3;
          const float x0 = fvn[tl][2 * pp], x1 = fvn[tl][2 * pp + 1];
          if (v == 0) bhn[tl][pp] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302);
          if (v == 1) s_ta = __float_as_uint(x0) & 0xffff0000u;
          if (v == 2) s_tb = __float_as_uint(x1) & 0xffff0000u;
          if (v == 3) s_ra = x0 - __uint_as_float(s_ta);
          if (v == 4) s_rb = x1 - __uint_as_float(s_tb);
          if (v == 5) bmn[tl][pp] = __builtin_amdgcn_perm(__float_as_uint(s_rb), __float_as_uint(s_ra), 0x07060302);
          if (v == 6) s_ta = __float_as_uint(s_ra) & 0xffff0000u;
          if (v == 7) s_tb = __float_as_uint(s_rb) & 0xffff0000u;
          if (v == 8) s_la = s_ra - __uint_as_float(s_ta);
          if (v == 9) s_lb = s_rb - __uint_as_float(s_tb);
          if (v == 10) bln[tl][pp] = __builtin_amdgcn_perm(__float_as_uint(s_lb), __float_as_uint(s_la), 0x07060302);
        }
      };
      // MFMA steps of block kb on the fragments in bhc/bmc/blc; HOOK: prepare block kb + 1 behind them
      auto l1_steps = [&](auto hook_c, int kb) __attribute__((always_inline)) {
        constexpr bool HOOK = decltype(hook_c)::value;
        const int lo_step = B::L1 + kb * HT;
        if (HOOK) { nxt_off = 32 * (kb + 1); table_load(0); }
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          const u32x4 al = q0;
          q0 = q1;
          q1 = lo_frag(lo_step + mt + 2);
          fetch1(mt + 2 < HT ? kb : kb + 1, (mt + 2) % HT, wb1);
          __builtin_amdgcn_sched_barrier(0);
#define NFOPP_ACC(tl) acc1[tl][mt]
#define NFOPP_L1_WORK(q_)                                                                     \
          if (HOOK) {                                                                          \
            const int slot_ = mt * 8 + (q_) - ((q_) + 2) / 3;                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) work_item(5 * slot_ + i_);       \
          }
          NFOPP_STEP_W(NFOPP_ACC, bhc, bmc, blc, ah1, am1, al, true, wa1, ahn1, amn1, NFOPP_L1_WORK)
#undef NFOPP_L1_WORK
#undef NFOPP_ACC
          ah1 = ahn1; am1 = amn1;
#pragma unroll
          for (int j = 0; j < 8; ++j) wa1[j] = wb1[j];
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto upfront = [&](auto ang_c, int kb) __attribute__((always_inline)) {   // l1_block's evaluation, no steps
        constexpr bool ANG = decltype(ang_c)::value;
        const int off = 32 * kb;
        float fv[NT][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float* fte = ftl + L::ft_rel(off + 8 * half);
          const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
          f32x2 sk = {skip[0], skip[NT - 1]};
          f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
          if (ANG) isa4 = *reinterpret_cast<const f32x4*>(isl + off + 8 * half);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);
            const f32x4 e2 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 8);
            const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
            const f32x2 qh = {e2.x, e2.y}, w3 = {e2.z, e2.w};
            const f32x2 v = features2<ANG, false>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
            sk = fma2(w3, v, sk);
            fv[0][4 * half + r] = v.x; fv[NT - 1][4 * half + r] = v.y;
          }
          skip[0] = sk.x; skip[NT - 1] = sk.y;
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) split8(fv[tl], bhc[tl], bmc[tl], blc[tl]);
      };
      upfront(std::false_type{}, 0);
#pragma unroll 1
      for (int kb = 0; kb < 5; ++kb) {
        l1_steps(std::true_type{}, kb);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) { bhc[tl] = bhn[tl]; bmc[tl] = bmn[tl]; blc[tl] = bln[tl]; }
      }
      l1_steps(std::false_type{}, 5);
      upfront(std::true_type{}, 6);
      l1_steps(std::false_type{}, 6);
    } else {
#pragma unroll 1
      for (int kb = 0; kb < first_angle_kb; ++kb) l1_block(std::false_type{}, kb);
#pragma unroll 1
      for (int kb = first_angle_kb; kb < B::NKB; ++kb) l1_block(std::true_type{}, kb);
    }

    if (TRAIN) {  // h1 (layout Q slots), ones at slot (tile 6, g = 0, r = 1)
#pragma unroll
      for (int tl = 0; tl < NT; ++tl)
        if (pidx[tl] < a.n_points) {
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            f32x4 v = {relu1(acc1[tl][t][0]), relu1(acc1[tl][t][1]), relu1(acc1[tl][t][2]), relu1(acc1[tl][t][3])};
            if (t == 6) v = f32x4{v[0], g == 0 ? 1.0f : 0.0f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(a.ws_h1 + pidx[tl] * WH + 16 * t + 4 * g) = v;
          }
        }
    }
    NFOPP_TICK(1)   // L1 (features + steps)
    // ---------------------------------------------------------------- L2: a2 = W2 relu(a1) + b2
    NFOPP_REDERIVE();
    f32x4 acc2[NT][HT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + L::B2 + (mt < 6 ? base_p(mt) + colP : 96 + 4 * g));
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) acc2[tl][mt] = bias;
    }
    {
      int rowoff2[HT];  // rows in h2 layout P
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) rowoff2[mt] = (mt < 6 ? base_p(mt) + rowposP : 96 + gi) * S2;
      // raw (hi | mid) words: wa = the NEXT step (packed between this step's MFMAs), wb = the step after (in flight)
      float wa[8], wb[8];
      auto fetch = [&](int kb, int mt, float (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = 2 * kb + (j >> 2), r = j & 3;
          dst[j] = t < 6 ? W2[rowoff2[mt] + 16 * t + r + colQ] : ((t == 6 && r == 0) ? W2[rowoff2[mt] + 96 + g] : 0.0f);
        }
      };
      constexpr int NSTEP = B::HKB * HT;
      u32x4 ah, am, ahn = {0, 0, 0, 0}, amn = {0, 0, 0, 0};
      fetch(0, 0, wa);
      pack_words(wa, ah, am);
      float raw0 = wa[0];
      fetch(0, 1, wa);
      u32x4 q0 = lo_frag(B::L2), q1 = lo_frag(B::L2 + 1);
#pragma unroll
      for (int kb = 0; kb < B::HKB; ++kb) {
        u32x4 bh[NT], bm[NT], bl[NT];
        float tail_b[NT];   // k block 3: the fp32 value of hidden unit 96 + g
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          tail_b[tl] = relu1(acc1[tl][6][0]);
          if (kb == TAIL_KB) continue;
          float hv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int t = 2 * kb + (j >> 2), r = j & 3;
            hv[j] = (t < 6 || (t == 6 && r == 0)) ? relu1(acc1[tl][t < HT ? t : 0][r]) : 0.0f;
          }
          split8(hv, bh[tl], bm[tl], bl[tl]);
        }
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          const int st = kb * HT + mt;
          const u32x4 al = q0;
          q0 = q1;
          q1 = lo_frag(B::L2 + st + 2);
          if (st + 2 < NSTEP) fetch((st + 2) / HT, (st + 2) % HT, wb);
          __builtin_amdgcn_sched_barrier(0);
          if (kb == TAIL_KB) {
            const float wf = rebuild_weight(raw0, al[0] & 0xffffu);
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) acc2[tl][mt] = mfma4(wf, tail_b[tl], acc2[tl][mt]);
          } else {
            const bool next_split = st + 1 < NSTEP && (st + 1) / HT < TAIL_KB;
#define NFOPP_ACC(tl) acc2[tl][mt]
            NFOPP_STEP(NFOPP_ACC, bh, bm, bl, ah, am, al, next_split, wa, ahn, amn)
#undef NFOPP_ACC
          }
          ah = ahn; am = amn; raw0 = wa[0];
#pragma unroll
          for (int j = 0; j < 8; ++j) wa[j] = wb[j];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    NFOPP_TICK(2)   // L2
    // ---------------------------------------------------------------- logit and dh2 = W3a * [a2 > 0]
    float logit[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) logit[tl] = skip[tl];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
      const f32x4 w3a = *reinterpret_cast<const f32x4*>(lds + L::W3A + (mt < 6 ? base_p(mt) + colP : 96 + 4 * g));
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a2 = acc2[tl][mt][r];
          const float d2 = a2 > 0.0f ? w3a[r] : 0.0f;   // dh2 = W3a * [a2 > 0]
          logit[tl] = fmaf(d2, a2, logit[tl]);          // = W3a * relu(a2)
          if (!TRAIN) acc2[tl][mt][r] = d2;             // TRAIN keeps a2 until rho is known (dW3[:100] below)
        }
      }
    }
    float rho[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      logit[tl] += __shfl_xor(logit[tl], 16);
      logit[tl] += __shfl_xor(logit[tl], 32);
      logit[tl] += b3;
      rho[tl] = 1.0f;
      if (TRAIN) {   // as onf_fused.hip: BCE-with-logits (nerf:25,88), rho = (sigmoid(l) - y) / count
        const bool valid = pidx[tl] < a.n_points;
        const float y = valid ? a.labels[pidx[tl]] : 0.0f;
        const float l = logit[tl];
        const float lp = fmaxf(l, 0.0f) - l * y + log1pf(expf(-fabsf(l)));
        rho[tl] = valid ? (1.0f / (1.0f + expf(-l)) - y) * a.inv_count : 0.0f;
        if (valid && g == 0) loss_acc += lp * a.inv_count;
        unsigned a2_mask = 0;
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          const f32x4 w3a = *reinterpret_cast<const f32x4*>(lds + L::W3A + (mt < 6 ? base_p(mt) + colP : 96 + 4 * g));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a2 = acc2[tl][mt][r];
            const bool on = a2 > 0.0f;
            g4_acc[mt][r] = fmaf(rho[tl], relu1(a2), g4_acc[mt][r]);
            a2_mask |= (on ? 1u : 0u) << (4 * mt + r);
            acc2[tl][mt][r] = (on ? w3a[r] : 0.0f) * rho[tl];
          }
        }
        if (valid) {
          if (g == 0) *reinterpret_cast<f32x4*>(a.ws_u + pidx[tl] * 12 + 4) = f32x4{rho[tl], 0.f, 0.f, 0.f};
          reinterpret_cast<unsigned*>(a.ws_u)[pidx[tl] * 12 + 8 + g] = a2_mask;
        }
      }
    }
    if (FWD_ONLY) {
#pragma unroll
      for (int tl = 0; tl < NT; ++tl)
        if (g == 0 && pidx[tl] < a.n_points)
          *reinterpret_cast<f32x4*>(a.out4 + pidx[tl] * 4) = f32x4{logit[tl], 0.f, 0.f, 0.f};
      continue;
    }

    NFOPP_TICK(3)   // logit
    // ---------------------------------------------------------------- L2T: dh1 = (W2^T dh2) * [a1 > 0]
    NFOPP_REDERIVE();
    {
      // the sign pattern of a1 is all L2T needs from it: 28 bits per point tile free its 28 registers
      unsigned mask1[NT];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        mask1[tl] = 0;
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mask1[tl] |= (acc1[tl][mt][r] > 0.0f ? 1u : 0u) << (4 * mt + r);
      }
      f32x4 accd[NT][HT];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) accd[tl][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      int coloff[HT];  // output rows = h1 layout Q -> column of W2
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) coloff[mt] = mt < 6 ? 16 * mt + rowposQ : 96 + gi;
      // raw (hi | mid) words: wa = the NEXT step (packed between this step's MFMAs), wb = the step after (in flight)
      float wa[8], wb[8];
      auto fetch = [&](int kb, int mt, float (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = 2 * kb + (j >> 2), r = j & 3;
          dst[j] = t < 6 ? W2[(base_p(t) + r + colP) * S2 + coloff[mt]]
                        : ((t == 6 && r == 0) ? W2[(96 + g) * S2 + coloff[mt]] : 0.0f);
        }
      };
      constexpr int NSTEP = B::HKB * HT;
      u32x4 ah, am, ahn = {0, 0, 0, 0}, amn = {0, 0, 0, 0};
      fetch(0, 0, wa);
      pack_words(wa, ah, am);
      float raw0 = wa[0];
      fetch(0, 1, wa);
      u32x4 q0 = lo_frag(B::L2T), q1 = lo_frag(B::L2T + 1);
#pragma unroll
      for (int kb = 0; kb < B::HKB; ++kb) {
        u32x4 bh[NT], bm[NT], bl[NT];
        float tail_b[NT];   // k block 3: the fp32 value of hidden unit 96 + g
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
          tail_b[tl] = acc2[tl][6][0];
          if (kb == TAIL_KB) continue;
          float hv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int t = 2 * kb + (j >> 2), r = j & 3;
            hv[j] = (t < 6 || (t == 6 && r == 0)) ? acc2[tl][t < HT ? t : 0][r] : 0.0f;
          }
          split8(hv, bh[tl], bm[tl], bl[tl]);
        }
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          const int st = kb * HT + mt;
          const u32x4 al = q0;
          q0 = q1;
          q1 = lo_frag(B::L2T + st + 2);
          if (st + 2 < NSTEP) fetch((st + 2) / HT, (st + 2) % HT, wb);
          __builtin_amdgcn_sched_barrier(0);
          if (kb == TAIL_KB) {
            const float wf = rebuild_weight(raw0, al[0] & 0xffffu);
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) accd[tl][mt] = mfma4(wf, tail_b[tl], accd[tl][mt]);
          } else {
            const bool next_split = st + 1 < NSTEP && (st + 1) / HT < TAIL_KB;
#define NFOPP_ACC(tl) accd[tl][mt]
            NFOPP_STEP(NFOPP_ACC, bh, bm, bl, ah, am, al, next_split, wa, ahn, amn)
#undef NFOPP_ACC
          }
          ah = ahn; am = amn; raw0 = wa[0];
#pragma unroll
          for (int j = 0; j < 8; ++j) wa[j] = wb[j];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc1[tl][mt][r] = ((mask1[tl] >> (4 * mt + r)) & 1u) ? accd[tl][mt][r] : 0.0f;  // dh1
      if (TRAIN) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          if (pidx[tl] < a.n_points) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              f32x4 v = acc1[tl][t];
              if (t == 6) v = f32x4{v[0], g == 0 ? rho[tl] : 0.0f, 0.f, 0.f};
              *reinterpret_cast<f32x4*>(a.ws_dh1 + pidx[tl] * WH + 16 * t + 4 * g) = v;
            }
          }
      }
    }

    NFOPP_TICK(4)   // L2T
    // ---------------------------------------------------------------- L1T: din = W1^T dh1 + W3b, then the chain rule
    NFOPP_REDERIVE();
    float gx[NT], gy[NT], gt[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) gx[tl] = gy[tl] = gt[tl] = 0.f;
    // dh1 in three levels, once for all output tiles: 4 k blocks x NT
    u32x4 dh[TAIL_KB][NT], dm[TAIL_KB][NT], dl[TAIL_KB][NT];
    float tail_d[NT];   // dh1 of hidden unit 96 + g (fp32 tail step)
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) tail_d[tl] = acc1[tl][6][0];
#pragma unroll
    for (int kb = 0; kb < TAIL_KB; ++kb)
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = 2 * kb + (j >> 2), r = j & 3;
          hv[j] = (t < 6 || (t == 6 && r == 0)) ? acc1[tl][t < HT ? t : 0][r] : 0.0f;
        }
        split8(hv, dh[kb][tl], dm[kb][tl], dl[kb][tl]);
      }
    const int rowkQ = colQ * S1;
    u32x4 q0t = lo_frag(B::L1T), q1t = lo_frag(B::L1T + 1);

    // step state of L1T (as in L1): steps run on from one output tile to the next
    float wat[8], wbt[8], raw0t;
    u32x4 aht, amt, ahnt = {0, 0, 0, 0}, amnt = {0, 0, 0, 0};
    auto fetcht = [&](int mt, int kb, float (&dst)[8]) __attribute__((always_inline)) {
      const int colA = base_p(mt) + rowposP;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 2 * kb + (j >> 2), r = j & 3;
        dst[j] = t < 6 ? W1[rowkQ + (16 * t + r) * S1 + colA] : ((t == 6 && r == 0) ? W1[(96 + g) * S1 + colA] : 0.0f);
      }
    };
    fetcht(0, 0, wat);
    pack_words(wat, aht, amt);
    raw0t = wat[0];
    fetcht(0, 1, wat);
    auto l1t_tile = [&](auto ang_c, int mt) __attribute__((always_inline)) {
      constexpr bool ANG = decltype(ang_c)::value;
      const int fbase = base_p(mt) + colP;
      const f32x4 w3b = *reinterpret_cast<const f32x4*>(lds + L::W3B + fbase);
      f32x4 acc[NT];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) acc[tl] = TRAIN ? w3b * rho[tl] : w3b;
      const int lo_step = B::L1T + mt * B::HKB;
#pragma unroll
      for (int kb = 0; kb < B::HKB; ++kb) {
        const u32x4 al = q0t;
        q0t = q1t;
        q1t = lo_frag(lo_step + kb + 2);
        fetcht(kb + 2 < B::HKB ? mt : mt + 1, (kb + 2) % B::HKB, wbt);   // past the last tile: a harmless in-image read
        __builtin_amdgcn_sched_barrier(0);
        if (kb == TAIL_KB) {
          const float wf = rebuild_weight(raw0t, al[0] & 0xffffu);
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma4(wf, tail_d[tl], acc[tl]);
          pack_words(wat, ahnt, amnt);   // next step: k block 0 of the next output tile
        } else {
#define NFOPP_ACC(tl) acc[tl]
          NFOPP_STEP(NFOPP_ACC, dh[kb < TAIL_KB ? kb : 0], dm[kb < TAIL_KB ? kb : 0], dl[kb < TAIL_KB ? kb : 0], aht, amt, al,
                     kb + 1 < TAIL_KB, wat, ahnt, amnt)
#undef NFOPP_ACC
        }
        aht = ahnt; amt = amnt; raw0t = wat[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) wat[j] = wbt[j];
        __builtin_amdgcn_sched_barrier(0);
      }
      const float* fte = lds + L::ft(fbase);
      if (NT == 2) {
        const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
        f32x2 gx2 = {gx[0], gx[NT - 1]}, gy2 = {gy[0], gy[NT - 1]}, gt2 = {gt[0], gt[NT - 1]};
        f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
        if (ANG) isa4 = *reinterpret_cast<const f32x4*>(lds + L::ISA + fbase);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);
          const f32x2 qh = *reinterpret_cast<const f32x2*>(fte + L::FTS * r + 8);
          const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
          const f32x2 cof = features2<ANG, true>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
          const f32x2 de = f32x2{acc[0][r], acc[NT - 1][r]} * cof;
          gx2 = fma2(de, wx, gx2);
          gy2 = fma2(de, wy, gy2);
          if (ANG) gt2 = fma2(de, fr, gt2);
        }
        gx[0] = gx2.x; gx[NT - 1] = gx2.y; gy[0] = gy2.x; gy[NT - 1] = gy2.y; gt[0] = gt2.x; gt[NT - 1] = gt2.y;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const float* ea = fte + L::FTS * r;
          const float* eb = ea + L::FTS;
          const f32x2 wx = {ea[0], eb[0]}, wy = {ea[2], eb[2]}, bb = {ea[4], eb[4]}, fr = {ea[6], eb[6]};
          const f32x2 qh = {ea[8], eb[8]}, isa = {lds[L::ISA + fbase + r], lds[L::ISA + fbase + r + 1]};
          const f32x2 ux2 = splat2(ux[0]), uy2 = splat2(uy[0]), th2 = splat2(th[0]);
          const f32x2 cof = features2<ANG, true>(wx, wy, bb, fr, qh, isa, ux2, uy2, th2);
          const f32x2 de = f32x2{acc[0][r], acc[0][r + 1]} * cof;
          if (TRAIN) { acc[0][r] = de.x; acc[0][r + 1] = de.y; }
          if (!TRAIN) {   // the fit needs de only, not d logit / d pose
            gx[0] = fmaf(de.x, wx.x, gx[0]); gx[0] = fmaf(de.y, wx.y, gx[0]);
            gy[0] = fmaf(de.x, wy.x, gy[0]); gy[0] = fmaf(de.y, wy.y, gy[0]);
            if (ANG) { gt[0] = fmaf(de.x, fr.x, gt[0]); gt[0] = fmaf(de.y, fr.y, gt[0]); }
          }
        }
      }
      if (TRAIN) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          if (pidx[tl] < a.n_points)
            *reinterpret_cast<f32x4*>(a.ws_de + pidx[tl] * WIN + 16 * mt + 4 * g) = acc[tl];
      }
    };
    const int first_angle_kt = 2 * first_angle_kb < NKT ? 2 * first_angle_kb : NKT;
    if constexpr (NT == 2 && B::NKB == 7) {
      // As in L1: the chain-rule epilogue of output tile mt-1 (tiles without angle features) runs in the MFMA shadows of
      // tile mt's steps.  92 work items per tile: evaluation e = w / 11 (feature row r = e >> 1 of point tile e & 1),
      // step w % 11; four to a slot.  Same arithmetic and order as l1t_tile's epilogue.
      f32x4 acc_prev[NT], acc_cur[NT];
      f32x4 de_prev[NT];   // TRAIN: de of the tile whose epilogue is in flight
      float d_arg = 0.f, d_j = 0.f, d_r = 0.f, d_t = 0.f, d_de = 0.f;
      // compact entry (wx, wy, b, qh) of feature row r, ping-pong on r & 1.  Kept as scalars: with the entry held as one
      // f32x4, hipcc 7.2's SLP pass fused the two tiles' gy updates into v_pk_fma_f32 ... op_sel:[0,1,0] and the low lane
      // (tile 0) came out wrong on the GPU -- the instruction alone is fine (tools/micro/pk_opsel.hip); DESIGN.md has the
      // account.  tests/test_gpu_split_path.py is the guard.
      float dwx[2], dwy[2], db[2], dq[2];   // dq: quadrant offset + a quarter turn (derivative)
      int pm_base = 0;      // base_p(mt - 1) + colP: first feature of the tile whose epilogue is in flight
      auto dtable_load = [&](int r) __attribute__((always_inline)) {
        const f32x4 e = *reinterpret_cast<const f32x4*>(lds + L::fc(pm_base + r));
        dwx[r & 1] = e.x; dwy[r & 1] = e.y; db[r & 1] = e.z;
        dq[r & 1] = e.w + NFOPP_Q_UNIT;
      };
      auto epi_item = [&](int w) __attribute__((always_inline)) {
        if (w >= 88) return;
        const int e = w / 11, u = w % 11, r = e >> 1, tl = e & 1, b = r & 1;
        if (u == 0) {
          d_arg = fmaf(dwy[b], uy[tl], db[b]);
          if (tl == 0 && r + 1 < 4) dtable_load(r + 1);
        }
        if (u == 1) d_arg = fmaf(dwx[b], ux[tl], d_arg);
        if (u == 2) d_t = fmaf(d_arg, 0.159154943f, 12582912.0f);
        if (u == 3) d_j = d_t - 12582912.0f;
        if (u == 4) d_r = fmaf(d_j, -6.28318548202514648f, d_arg);
        if (u == 5) d_r = fmaf(d_j, 1.74845553e-07f, d_r);
        if (u == 6) d_t = fmaf(d_r, 0.159154943f, dq[b]);
        if (u == 7) d_t = __builtin_amdgcn_sinf(d_t);
        if (u == 8) d_de = acc_prev[tl][r] * d_t;
        if constexpr (TRAIN) {   // the fit needs de itself (stored per tile below), not d logit / d pose
          if (u == 9) de_prev[tl][r] = d_de;
        } else {
          if (u == 9) gx[tl] = fmaf(d_de, dwx[b], gx[tl]);
          if (u == 10) gy[tl] = fmaf(d_de, dwy[b], gy[tl]);
        }
      };
      auto store_de = [&](int mt, const f32x4 (&de)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
          if (pidx[tl] < a.n_points) *reinterpret_cast<f32x4*>(a.ws_de + pidx[tl] * WIN + 16 * mt + 4 * g) = de[tl];
      };
      auto l1t_steps = [&](auto hook_c, int mt, f32x4 (&acc)[NT]) __attribute__((always_inline)) {
        constexpr bool HOOK = decltype(hook_c)::value;
        const int fbase = base_p(mt) + colP;
        const f32x4 w3b = *reinterpret_cast<const f32x4*>(lds + L::W3B + fbase);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = TRAIN ? w3b * rho[tl] : w3b;
        const int lo_step = B::L1T + mt * B::HKB;
        if (HOOK) { pm_base = base_p(mt - 1) + colP; dtable_load(0); }
#pragma unroll
        for (int kb = 0; kb < B::HKB; ++kb) {
          const u32x4 al = q0t;
          q0t = q1t;
          q1t = lo_frag(lo_step + kb + 2);
          fetcht(kb + 2 < B::HKB ? mt : mt + 1, (kb + 2) % B::HKB, wbt);
          __builtin_amdgcn_sched_barrier(0);
          if (kb == TAIL_KB) {
            const float wf = rebuild_weight(raw0t, al[0] & 0xffffu);
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma4(wf, tail_d[tl], acc[tl]);
            pack_words(wat, ahnt, amnt);
          } else {
#define NFOPP_ACC(tl) acc[tl]
#define NFOPP_L1T_WORK(q_)                                                                    \
            if (HOOK) {                                                                        \
              const int slot_ = kb * 8 + (q_) - ((q_) + 2) / 3;                                \
              _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) epi_item(4 * slot_ + i_);      \
            }
            NFOPP_STEP_W(NFOPP_ACC, dh[kb < TAIL_KB ? kb : 0], dm[kb < TAIL_KB ? kb : 0], dl[kb < TAIL_KB ? kb : 0], aht, amt,
                         al, kb + 1 < TAIL_KB, wat, ahnt, amnt, NFOPP_L1T_WORK)
#undef NFOPP_L1T_WORK
#undef NFOPP_ACC
          }
          aht = ahnt; amt = amnt; raw0t = wat[0];
#pragma unroll
          for (int j = 0; j < 8; ++j) wat[j] = wbt[j];
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto epilogue = [&](auto ang_c, int mt, const f32x4 (&acc)[NT]) __attribute__((always_inline)) {   // as l1t_tile
        constexpr bool ANG = decltype(ang_c)::value;
        const int fbase = base_p(mt) + colP;
        const float* fte = lds + L::ft(fbase);
        const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
        f32x2 gx2 = {gx[0], gx[NT - 1]}, gy2 = {gy[0], gy[NT - 1]}, gt2 = {gt[0], gt[NT - 1]};
        f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
        if (ANG) isa4 = *reinterpret_cast<const f32x4*>(lds + L::ISA + fbase);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);
          const f32x2 qh = *reinterpret_cast<const f32x2*>(fte + L::FTS * r + 8);
          const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
          const f32x2 cof = features2<ANG, true>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
          const f32x2 de = f32x2{acc[0][r], acc[NT - 1][r]} * cof;
          if constexpr (TRAIN) {
            de_prev[0][r] = de.x; de_prev[NT - 1][r] = de.y;
          } else {
            gx2 = fma2(de, wx, gx2);
            gy2 = fma2(de, wy, gy2);
            if (ANG) gt2 = fma2(de, fr, gt2);
          }
        }
        if constexpr (TRAIN) store_de(mt, de_prev);
        gx[0] = gx2.x; gx[NT - 1] = gx2.y; gy[0] = gy2.x; gy[NT - 1] = gy2.y; gt[0] = gt2.x; gt[NT - 1] = gt2.y;
      };
      // tiles 0 .. first_angle_kt-1 carry no angle features: their epilogues hide behind the next tile's steps
      const int plain_tiles = first_angle_kt < NKT ? first_angle_kt : NKT - 1;   // epilogues that get a host tile
      l1t_steps(std::false_type{}, 0, acc_prev);
#pragma unroll 1
      for (int mt = 1; mt <= plain_tiles; ++mt) {
        l1t_steps(std::true_type{}, mt, acc_cur);
        if constexpr (TRAIN) store_de(mt - 1, de_prev);   // the epilogue of tile mt - 1 ran behind tile mt's steps
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc_prev[tl] = acc_cur[tl];
      }
      // what is left: the epilogue of tile plain_tiles (angle tile or last tile) and the angle tiles behind it
      if (plain_tiles >= first_angle_kt) epilogue(std::true_type{}, plain_tiles, acc_prev);
      else epilogue(std::false_type{}, plain_tiles, acc_prev);
#pragma unroll 1
      for (int mt = plain_tiles + 1; mt < NKT; ++mt) {
        l1t_steps(std::false_type{}, mt, acc_cur);
        epilogue(std::true_type{}, mt, acc_cur);
      }
    } else {
#pragma unroll 1
      for (int mt = 0; mt < first_angle_kt; ++mt) l1t_tile(std::false_type{}, mt);
#pragma unroll 1
      for (int mt = first_angle_kt; mt < NKT; ++mt) l1t_tile(std::true_type{}, mt);
    }
    NFOPP_TICK(5)   // L1T (steps + epilogues)
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      gx[tl] += __shfl_xor(gx[tl], 16); gx[tl] += __shfl_xor(gx[tl], 32);
      gy[tl] += __shfl_xor(gy[tl], 16); gy[tl] += __shfl_xor(gy[tl], 32);
      gt[tl] += __shfl_xor(gt[tl], 16); gt[tl] += __shfl_xor(gt[tl], 32);
      if (a.out4 && g == 0 && pidx[tl] < a.n_points) {
        f32x4 o = {logit[tl], gx[tl] / geo.sigma, gy[tl] / geo.sigma, gt[tl]};
        *reinterpret_cast<f32x4*>(a.out4 + pidx[tl] * 4) = o;
      }
    }
    NFOPP_TICK(6)   // output
  }
  if (TRAIN) {  // fixed-order partials, as onf_fused.hip: loss per wave, dW3[:100] per wave in h2 slot order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
    if (lane == 0) a.loss_partial[blockIdx.x * WAVES + wave] = loss_acc;
    float* g4 = a.g4_partial + (size_t)(blockIdx.x * WAVES + wave) * (16 * HT);
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
      f32x4 v = g4_acc[mt];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], o);
      if (i == 0) *reinterpret_cast<f32x4*>(g4 + 16 * mt + 4 * g) = v;
    }
  }
#ifdef NFOPP_PHASE_PROFILE
  if (MODE == 0 && a.ws_u && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(a.ws_u + k, phase_ticks[k]);
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------
// -1: read NFOPP_MATRIX_PATH on first use; 0: fp32 MFMA; 1: bf16x3 split (default): the 32x32x16 kernel of csrc/onf_x32.hip
// at EVERY launch size -- results must not depend on how a batch is sharded -- and this file's 16x16x32 kernel for the
// training pass and the feature dimensions the other does not cover; 2: bf16x3 split, this file's kernel everywhere
static int g_split_mode = -1;

static int split_mode() {
  if (g_split_mode < 0) {
    const char* e = getenv("NFOPP_MATRIX_PATH");
    g_split_mode = !e ? 1 : (e[0] == 'f' || e[0] == '0') ? 0 : e[0] == '2' ? 2 : 1;
  }
  return g_split_mode;
}

bool onf_split_enabled() { return split_mode() >= 1; }

// the 32x32x16 kernel takes the launch whenever it covers the feature dimension
bool onf_use_x32(const OnfGeom& g) { return split_mode() == 1 && onf_x32_supports(g); }

// Third-level blobs.  split_prep_kernel rewrites the blob in front of every launch ON THE LAUNCH STREAM, so launches of
// one stream are ordered by the stream itself; two streams of one device (two planners with different fields) must not
// share a blob, so blobs are keyed by (device, stream).  A handful of streams per device is the realistic case.
constexpr int MAX_BLOBS = 16;
struct BlobSlot { hipStream_t stream; void* ptr; size_t bytes; bool used; unsigned long long stamp; };
static unsigned long long g_blob_stamp = 0;
static BlobSlot g_blobs[MAX_DEVICES][MAX_BLOBS] = {};
static std::mutex g_blob_mutex;

static int blob_for_stream(size_t bytes, hipStream_t stream, u32x4** out) {
  const int dev = current_device();
  if (dev < 0) return NFOPP_ERR_HIP;
  std::lock_guard<std::mutex> lock(g_blob_mutex);
  BlobSlot* slot = nullptr;
  for (int k = 0; k < MAX_BLOBS && !slot; ++k)
    if (g_blobs[dev][k].used && g_blobs[dev][k].stream == stream) slot = &g_blobs[dev][k];
  for (int k = 0; k < MAX_BLOBS && !slot; ++k)
    if (!g_blobs[dev][k].used) { slot = &g_blobs[dev][k]; slot->used = true; }
  if (!slot) {   // every slot taken (PyTorch's pool alone hands out 32 streams per priority): reuse the least recently used one;
                 // its stream may still be running on the blob, so wait for the device once
    slot = &g_blobs[dev][0];
    for (int k = 1; k < MAX_BLOBS; ++k)
      if (g_blobs[dev][k].stamp < slot->stamp) slot = &g_blobs[dev][k];
    NFOPP_HIP(hipDeviceSynchronize());
  }
  slot->stream = stream;
  slot->stamp = ++g_blob_stamp;
  if (slot->bytes < bytes) {
    if (slot->ptr) NFOPP_HIP(hipFree(slot->ptr));
    slot->ptr = nullptr; slot->bytes = 0;
    NFOPP_HIP(hipMalloc(&slot->ptr, bytes));
    slot->bytes = bytes;
  }
  *out = reinterpret_cast<u32x4*>(slot->ptr);
  return NFOPP_OK;
}

template <int NKT, int NT, int MODE>
static int launch_split_t(const OnfKernelArgs& a, hipStream_t stream, int* grid_out = nullptr) {
  using L = Lds<NKT>;
  using B = Blob<NKT>;
  static bool attr_set[MAX_DEVICES] = {};
  auto kern = onf_split_kernel<NKT, NT, MODE>;
  int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), L::BYTES, attr_set);
  if (rc != NFOPP_OK) return rc;
  u32x4* blob = nullptr;
  rc = blob_for_stream(B::BYTES, stream, &blob);
  if (rc != NFOPP_OK) return rc;
  // third weight level in consumption order (the parameters may have changed since the last call: always rebuilt)
  hipLaunchKernelGGL(split_prep_kernel<NKT>, dim3(B::STEPS), dim3(64), 0, stream, a.geom, a.params, blob);
  NFOPP_HIP(hipGetLastError());
  constexpr int CH = WAVES * 16 * NT;
  long long n_chunks = (a.n_points + CH - 1) / CH;
  long long grid = query_cus();
  if (grid > n_chunks) grid = n_chunks;
  if (grid_out) *grid_out = (int)grid;
#ifdef NFOPP_PHASE_PROFILE
  if (MODE == 0 && NT == 2) {   // development only: synchronous, prints to stderr
    static float* dbg = nullptr;
    if (!dbg) NFOPP_HIP(hipMalloc(&dbg, 64));
    NFOPP_HIP(hipMemsetAsync(dbg, 0, 64, stream));
    OnfKernelArgs b = a;
    b.ws_u = dbg;   // unused by this mode: carries the tick buffer
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), L::BYTES, stream, b, (const u32x4*)blob);
    float h[8];
    NFOPP_HIP(hipMemcpyAsync(h, dbg, 32, hipMemcpyDeviceToHost, stream));
    NFOPP_HIP(hipStreamSynchronize(stream));
    static int calls = 0;
    if (++calls % 10 == 0) {
      float tot = 0;
      for (int k = 0; k < 7; ++k) tot += h[k];
      const char* names[7] = {"sampling", "L1", "L2", "logit", "L2T", "L1T", "output"};
      fprintf(stderr, "[phase profile] wave 0 of %lld workgroups, share of the chunk loop:\n", grid);
      for (int k = 0; k < 7; ++k) fprintf(stderr, "   %-10s %5.1f %%\n", names[k], 100.0f * h[k] / tot);
    }
    return NFOPP_OK;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), L::BYTES, stream, a, (const u32x4*)blob);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

template <int MODE>
static int launch_split_mode(const OnfKernelArgs& a, hipStream_t stream) {
  const int nkt = (a.geom.fin + 15) / 16;
#ifdef NFOPP_FORCE_NT1
  const bool small = true;   // development A/B: one point tile per wave at any size
#else
  const bool small = a.n_points < (long long)query_cus() * WAVES * 16 * 2;
#endif
  switch (nkt) {
    case 14: return small ? launch_split_t<14, 1, MODE>(a, stream) : launch_split_t<14, 2, MODE>(a, stream);
    case 13: return small ? launch_split_t<13, 1, MODE>(a, stream) : launch_split_t<13, 2, MODE>(a, stream);
    case 8: return launch_split_t<8, 1, MODE>(a, stream);
    case 7: return launch_split_t<7, 1, MODE>(a, stream);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

// training pass on the split path: two point tiles per wave on the shadow steps for F = 200..224 (spills 344 B of registers and
// is still 9 % faster than one tile per wave), one tile per wave otherwise
int launch_onf_split_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out) {
  const int nkt = (a.geom.fin + 15) / 16;
  switch (nkt) {
#ifdef NFOPP_TRAIN_NT1   /* development A/B: one point tile per wave at any size */
    case 14: return launch_split_t<14, 1, 1>(a, stream, grid_out);
#else
    // two tiles per wave once every CU gets a full workgroup of them (as the evaluation kernel decides); smaller fits fill more
    // CUs with one tile per wave (4096 samples: 57 vs 85 us)
    case 14:
      return a.n_points < (long long)query_cus() * WAVES * 16 * 2 ? launch_split_t<14, 1, 1>(a, stream, grid_out)
                                                                   : launch_split_t<14, 2, 1>(a, stream, grid_out);
#endif
    case 13: return launch_split_t<13, 1, 1>(a, stream, grid_out);
    case 8: return launch_split_t<8, 1, 1>(a, stream, grid_out);
    case 7: return launch_split_t<7, 1, 1>(a, stream, grid_out);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

int launch_onf_split_kernel(const OnfKernelArgs& a, hipStream_t stream, bool forward_only) {
  if (a.n_points <= 0) return NFOPP_OK;
  if (onf_use_x32(a.geom)) return launch_onf_x32_kernel(a, stream, forward_only);
  return forward_only ? launch_split_mode<2>(a, stream) : launch_split_mode<0>(a, stream);
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_set_matrix_path(int32_t path) {
  NFOPP_REQUIRE(path >= 0 && path <= 2,
                "matrix path must be 0 (fp32 MFMA), 1 (bf16x3 split MFMA) or 2 (bf16x3 split MFMA, 16x16x32 kernels only)");
  g_split_mode = path;
  return NFOPP_OK;
}

extern "C" int nfopp_get_matrix_path(void) { return split_mode(); }
