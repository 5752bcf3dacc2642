// (implementation header: the kernels and their launch templates; csrc/onf_x32.hip holds the per-stream images and the
// entry points, csrc/onf_x32_k{14,13,8,7}.hip instantiate one feature dimension each so that the build runs in parallel)
// K1 on 32x32x16 tiles: fused collision sampling + ONF forward + input gradient for gfx950, every GEMM on the bf16 matrix
// pipe as an exact three-level split of the fp32 operands (the arithmetic of csrc/onf_split.hip: x = hi + mid + lo, six
// partial products per multiply, fp32 accumulation).  Reference: nfop/onf_model.py:33-50 + autograd, nfop/angle_encoder.py:
// 15-18, the collision sampling of nfop/constrained_nerf_opt_planner.py:78-81 (csrc/onf_layout.h: load_point).
//
// Why a second formulation (round 3).  onf_split.hip (16x16x32 tiles, two 16-point tiles per wave) ran at 49 % of the
// matrix pipe with 4.2 vector instructions per MFMA: a v_mfma_f32_16x16x32_bf16 holds the SIMD's vector issue port for 8 of
// its 16 cycles, every weight fragment cost 8 ds_read_b32 + 8 v_perm_b32, and biases / skip connection / ReLU masks were
// vector work.  Here
//  * one wave owns ONE tile of 32 samples; v_mfma_f32_32x32x16_bf16 holds the issue port for 8 of its 32 cycles, so the
//    vector work of a step (feature evaluation, operand splitting, chain-rule epilogue) fits behind its MFMAs;
//  * the hi and the mid level of the weights are two bf16 MATRICES per layer in LDS, M[output position][input slot]
//    (144 KB), read as finished A fragments: ds_read_b128 in the forward GEMMs, ds_read_b64_tr_b16 in the transposed ones,
//    both conflict-free on the same swizzled image (tools/x32/lds_search.py); no permutes, 2 (4) LDS instructions per step;
//  * b1, b2, b3 and the skip connection W3[100:] ride in spare rows / columns of the padded matrices: input position `fin`
//    is a constant-one feature (column = b1), hidden row 100 is the skip row (W3b | b3 -> its pre-activation IS the skip
//    part of the logit, and dh1[100] := 1 adds W3b to the input gradient), hidden row 101 regenerates the constant one
//    (column 101 of W2 = b2);
//  * the third level stays in an L2-resident blob in consumption order (one 16-byte buffer load per lane and step).
// Accumulator chaining as before: a 32x32 result tile is the next GEMM's B operand with the k order permuted inside a
// 16-block -- slot 16 kb + 8 g + e  <->  position 16 kb + 8 (e >> 2) + 4 g + (e & 3) -- and the images are stored in slot
// order, so L1 -> L2 -> L2^T -> L1^T never leaves registers.  tools/x32/emulate_x32.py checks every address formula
// below, lane by lane, against a plain MLP (CPU).
//
// Modes: 0 forward + input gradient, 2 forward only, 1 the training pass of the ONF fit at scale (pass 1 of
// csrc/onf_wgrad.hip; reference: loss.backward() of nfop/nerf_opt_planner.py:83-89).  The training pass runs the SAME four
// GEMMs with dh2 = W3a [a2 > 0] unscaled -- everything behind it is linear in rho = (sigmoid(logit) - y) / count, which is
// only known once the logit is complete -- and multiplies rho in where the factors are stored:  h1 | rho dh1 | rho de  by
// feature / hidden-unit index ("x32 order": the accumulator layout gives every lane four consecutive positions per
// register quad, i.e. one 16-byte store) plus the 48-byte record (u, 1, theta | rho | sign words of a2).  dW3[:100] =
// sum_p rho_p relu(a2_p) is NOT accumulated here: it falls out of G2 in the gather kernel (WgradArgs::x32_order).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#pragma once
#include "onf_layout.h"

namespace nfopp {
namespace x32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int V> using ic = std::integral_constant<int, V>;
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    sfor<I + 1, N>(f);
  }
}

// ---- geometry of the LDS image (bytes) ---------------------------------------------------------------------------
constexpr int RS1 = 448, RS2 = 256;            // row strides of the W1 / W2 images (224 / 112 bf16 slots + swizzle room)
constexpr int W1_ROWS = 103, W2_ROWS = 101;    // hidden rows 0..99 | 100 skip | 101 ones | zero row;  0..99 | zero row
constexpr int W1_ZERO = 102, W2_ZERO = 100;
constexpr int SKIP = 100, ONES = 101;
constexpr int HK = 7;                          // hidden k steps of 16 slots (positions 0..111)
constexpr int up256(int x) { return (x + 255) / 256 * 256; }
// every image starts on a multiple of 256 bytes: the W2 read addresses are formed with XORs on the low 8 bits
constexpr int O_W1H = 0;
constexpr int O_W1M = up256(O_W1H + W1_ROWS * RS1);
constexpr int O_W2H = up256(O_W1M + W1_ROWS * RS1);
constexpr int O_W2M = up256(O_W2H + W2_ROWS * RS2);
constexpr int O_FT = up256(O_W2M + W2_ROWS * RS2);    // [224] (c0, c1, b, q): forward feature table by input position
constexpr int O_FTD = O_FT + 224 * 16;         // the same with q + a quarter turn: d feature / d argument
constexpr int O_ISA = O_FTD + 224 * 16;        // [224] 1.0 = angle feature
constexpr int O_W3A = O_ISA + 224 * 4;         // [128] fp32 W3[:100] by hidden position
constexpr int O_W3L = O_W3A + 128 * 4;         // [7 kb][2 g][3 levels][4 words]: W3[:100] pre-split in B-fragment order
constexpr int IMG_BYTES = O_W3L + HK * 2 * 3 * 16;
static_assert(IMG_BYTES % 16 == 0 && IMG_BYTES <= 160 * 1024, "LDS image");

__host__ __device__ constexpr int pos_of_slot(int s) {
  return 16 * (s >> 4) + 8 * ((s >> 2) & 1) + 4 * ((s >> 3) & 1) + (s & 3);
}
__host__ __device__ constexpr int swz1(int row) { return (row >> 2) & 3; }
__host__ __device__ constexpr int swz2(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int NKB>
struct Cfg {
  static constexpr int NMT = (NKB + 1) / 2;        // 32-position output tiles of L1^T
  static constexpr int FS = NKB >= 13 ? 12 : 6;    // n_enc / 16: first input k block that can hold angle / ones / pad features
  static constexpr int S_L1 = 0, S_L2 = S_L1 + 4 * NKB, S_L2T = S_L2 + 4 * HK, S_L1T = S_L2T + 4 * HK;
  static constexpr int STEPS = S_L1T + HK * NMT;
  static constexpr size_t BLOB_BYTES = size_t(STEPS + 4) * 1024;   // + the three-step look-ahead past the last step
};

// ---- extended weight matrices (the folded biases / skip row / ones unit) ---------------------------------------
__device__ __forceinline__ float w1ext(const OnfGeom& g, const float* P, int row, int f) {
  if (row < H) return f < g.fin ? P[g.off_w1 + row * g.fin + f] : (f == g.fin ? P[g.off_b1 + row] : 0.0f);
  if (row == SKIP) return f < g.fin ? P[g.off_w3 + H + f] : (f == g.fin ? P[g.off_b3] : 0.0f);
  if (row == ONES) return f == g.fin ? 1.0f : 0.0f;
  return 0.0f;
}
__device__ __forceinline__ float w2ext(const OnfGeom& g, const float* P, int row, int hp) {
  if (row >= H) return 0.0f;
  if (hp < H) return P[g.off_w2 + row * H + hp];
  return hp == ONES ? P[g.off_b2 + row] : 0.0f;
}
__device__ __forceinline__ unsigned level_of(float w, int lvl) {
  unsigned r = split_level(w);
  if (lvl >= 1) r = split_level(w);
  if (lvl >= 2) r = __float_as_uint(w) >> 16;
  return r;
}
__device__ __forceinline__ u32x4 pack_level(const float (&w)[8], int lvl) {
  u32x4 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) o[p] = level_of(w[2 * p], lvl) | (level_of(w[2 * p + 1], lvl) << 16);
  return o;
}

// One thread per 16-byte piece of the LDS image and of the third-level blob.
template <int NKB>
__global__ __launch_bounds__(256) void x32_prep_kernel(const OnfGeom geo, const float* __restrict__ P, u32x4* __restrict__ img,
                                                       u32x4* __restrict__ blob) {
  using C = Cfg<NKB>;
  constexpr int N_IMG = IMG_BYTES / 16, N_BLOB = (C::STEPS + 4) * 64;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N_IMG + N_BLOB) return;
  if (idx >= N_IMG) {   // ---- blob: step (gemm, kb, mt), lane -> the lane's 8 weights, third level
    const int b = idx - N_IMG, step = b >> 6, lane = b & 63, i = lane & 31, g = lane >> 5;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = 0.0f;
    if (step < C::STEPS) {
      int kb, mt, kind;
      if (step < C::S_L2) { kind = 0; kb = step >> 2; mt = step & 3; }
      else if (step < C::S_L2T) { kind = 1; kb = (step - C::S_L2) >> 2; mt = (step - C::S_L2) & 3; }
      else if (step < C::S_L1T) { kind = 2; kb = (step - C::S_L2T) >> 2; mt = (step - C::S_L2T) & 3; }
      else { kind = 3; mt = (step - C::S_L1T) / HK; kb = (step - C::S_L1T) % HK; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = pos_of_slot(16 * kb + 8 * g + e), m = 32 * mt + i;
        w[e] = kind == 0 ? w1ext(geo, P, m, p) : kind == 1 ? w2ext(geo, P, m, p) : kind == 2 ? w2ext(geo, P, p, m)
                                                                                            : w1ext(geo, P, p, m);
      }
    }
    blob[b] = pack_level(w, 2);
    return;
  }
  const int byte = idx * 16;
  u32x4 out = {0u, 0u, 0u, 0u};
  if (byte < O_W2H) {          // W1 images: physical chunk c of row r holds logical chunk c ^ swz1(r)
    const int lvl = byte >= O_W1M ? 1 : 0, rel = byte - (lvl ? O_W1M : O_W1H), row = rel / RS1, ch = ((rel % RS1) >> 4) ^ swz1(row);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = row < W1_ROWS ? w1ext(geo, P, row, pos_of_slot(8 * ch + e)) : 0.0f;
    out = pack_level(w, lvl);
  } else if (byte < O_FT) {    // W2 images
    const int lvl = byte >= O_W2M ? 1 : 0, rel = byte - (lvl ? O_W2M : O_W2H), row = rel / RS2, ch = ((rel % RS2) >> 4) ^ swz2(row);
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (row < W2_ROWS && (8 * ch + e) < 16 * HK) ? w2ext(geo, P, row, pos_of_slot(8 * ch + e)) : 0.0f;
    out = pack_level(w, lvl);
  } else if (byte < O_ISA) {   // feature tables: sin(arg + q pi/2), q in revolutions (v_sin_f32 unit); FTD: + a quarter turn
    const bool deriv = byte >= O_FTD;
    const int f = (byte - (deriv ? O_FTD : O_FT)) >> 4;
    float c0 = 0.f, c1 = 0.f, b = 0.f, q = 0.f;
    if (f < geo.n_enc) {         // encoding_layer: W_e u + b_e (onf_model.py:39), cosine half: onf_model.py:41
      c0 = P[geo.off_we + 2 * f]; c1 = P[geo.off_we + 2 * f + 1];
      b = geo.off_be >= 0 ? P[geo.off_be + f] : 0.0f;
      q = (geo.n_enc > geo.n_sin && f >= geo.n_sin) ? 0.25f : 0.0f;
    } else if (f < geo.fin) {    // angle_encoder.py:16: (theta + b) * f
      const int k = f - geo.n_enc;
      c0 = P[geo.off_ang_f + k]; b = P[geo.off_ang_b + k];
      q = k >= geo.ang_dim ? 0.25f : 0.0f;
    } else if (f == geo.fin) {
      q = 0.25f;                 // the ones feature (its value is forced to exactly 1 where it is evaluated)
    }
    if (deriv) q += 0.25f;
    out = u32x4{__float_as_uint(c0), __float_as_uint(c1), __float_as_uint(b), __float_as_uint(q)};
  } else if (byte < O_W3A) {
    const int f0 = (byte - O_ISA) >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = __float_as_uint((f0 + k >= geo.n_enc && f0 + k < geo.fin) ? 1.0f : 0.0f);
  } else if (byte < O_W3L) {
    const int p0 = (byte - O_W3A) >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = __float_as_uint(p0 + k < H ? P[geo.off_w3 + p0 + k] : 0.0f);
  } else {
    const int t = (byte - O_W3L) >> 4, lvl = t % 3, g = (t / 3) & 1, kb = t / 6;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = pos_of_slot(16 * kb + 8 * g + e);
      w[e] = p < H ? P[geo.off_w3 + p] : 0.0f;
    }
    out = pack_level(w, lvl);
  }
  img[idx] = out;
}

// ---- device helpers ----------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const unsigned char* lds, int byte) {
  return *reinterpret_cast<const u32x4*>(lds + byte);
}
__device__ __forceinline__ f32x4 lds128f(const unsigned char* lds, int byte) {
  return *reinterpret_cast<const f32x4*>(lds + byte);
}
// two transposing reads = the 8 k values of a transposed A fragment (rows 8 apart in k: byte distance `d8`)
__device__ __forceinline__ u32x4 lds_tr(const unsigned char* lds, int byte0, int byte1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + byte0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + byte1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(u32x4, s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
}

// 6 partial products of one step (smallest terms first); WORK(slot) runs behind MFMA number slot - SLOT0
// FIRST: the accumulator starts here -- the first product takes a literal zero as its C operand (no register zeroing)
template <int SLOT0, bool FIRST = false, class W>
__device__ __forceinline__ void step6(f32x16& acc, const u32x4& ah, const u32x4& am, const u32x4& al, const u32x4 (&b)[3],
                                      W&& work_in) {
#ifdef X32_ABL_NOWORK   /* timing-only ablation (results wrong): the MFMA steps without the hooked vector work */
  auto work = [](auto) {};
  (void)work_in;
#else
  auto& work = work_in;
#endif
  if constexpr (FIRST) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = mfma32(al, b[0], zero);
  } else {
    acc = mfma32(al, b[0], acc);
  }
#if defined(X32_GROUP2)   /* development A/B: MFMAs in pairs, the work of both slots behind the pair */
  acc = mfma32(ah, b[2], acc); work(ic<SLOT0 + 0>{}); work(ic<SLOT0 + 1>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(am, b[1], acc);
  acc = mfma32(am, b[0], acc); work(ic<SLOT0 + 2>{}); work(ic<SLOT0 + 3>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(ah, b[1], acc);
  acc = mfma32(ah, b[0], acc); work(ic<SLOT0 + 4>{}); work(ic<SLOT0 + 5>{}); __builtin_amdgcn_sched_barrier(0);
#elif defined(X32_NO_FENCE)   /* development A/B: hipcc places the work items */
  work(ic<SLOT0 + 0>{});
  acc = mfma32(ah, b[2], acc); work(ic<SLOT0 + 1>{});
  acc = mfma32(am, b[1], acc); work(ic<SLOT0 + 2>{});
  acc = mfma32(am, b[0], acc); work(ic<SLOT0 + 3>{});
  acc = mfma32(ah, b[1], acc); work(ic<SLOT0 + 4>{});
  acc = mfma32(ah, b[0], acc); work(ic<SLOT0 + 5>{});
#else
  work(ic<SLOT0 + 0>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(ah, b[2], acc); work(ic<SLOT0 + 1>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(am, b[1], acc); work(ic<SLOT0 + 2>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(am, b[0], acc); work(ic<SLOT0 + 3>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(ah, b[1], acc); work(ic<SLOT0 + 4>{}); __builtin_amdgcn_sched_barrier(0);
  acc = mfma32(ah, b[0], acc); work(ic<SLOT0 + 5>{}); __builtin_amdgcn_sched_barrier(0);
#endif
}

// one instruction of the exact pair split (11 per pair): x0, x1 -> one word of each level
struct SplitState { unsigned ta, tb; float ra, rb, la, lb; };
template <int U>
__device__ __forceinline__ void split_item(SplitState& s, float x0, float x1, u32x4 (&out)[3], int p) {
  if constexpr (U == 0) out[0][p] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302);
  if constexpr (U == 1) s.ta = __float_as_uint(x0) & 0xffff0000u;
  if constexpr (U == 2) s.tb = __float_as_uint(x1) & 0xffff0000u;
  if constexpr (U == 3) s.ra = x0 - __uint_as_float(s.ta);
  if constexpr (U == 4) s.rb = x1 - __uint_as_float(s.tb);
  if constexpr (U == 5) out[1][p] = __builtin_amdgcn_perm(__float_as_uint(s.rb), __float_as_uint(s.ra), 0x07060302);
  if constexpr (U == 6) s.ta = __float_as_uint(s.ra) & 0xffff0000u;
  if constexpr (U == 7) s.tb = __float_as_uint(s.rb) & 0xffff0000u;
  if constexpr (U == 8) s.la = s.ra - __uint_as_float(s.ta);
  if constexpr (U == 9) s.lb = s.rb - __uint_as_float(s.tb);
  if constexpr (U == 10) out[2][p] = __builtin_amdgcn_perm(__float_as_uint(s.lb), __float_as_uint(s.la), 0x07060302);
}
__device__ __forceinline__ void split_pair(float x0, float x1, u32x4 (&out)[3], int p) {
  SplitState s;
  sfor<0, 11>([&](auto u) { split_item<decltype(u)::value>(s, x0, x1, out, p); });
}

// one instruction of a feature evaluation sin(arg + q) (hardware path of common.h: sin_halfturns_hw), 8 per feature
struct EvalState { float arg, t, j, r, v; };
template <int U>
__device__ __forceinline__ void eval_item(EvalState& s, const f32x4& tw, float ux, float uy) {
  if constexpr (U == 0) s.arg = fmaf(tw.y, uy, tw.z);
  if constexpr (U == 1) s.arg = fmaf(tw.x, ux, s.arg);
  if constexpr (U == 2) s.t = fmaf(s.arg, 0.159154943f, 12582912.0f);
  if constexpr (U == 3) s.j = s.t - 12582912.0f;
  if constexpr (U == 4) s.r = fmaf(s.j, -6.28318548202514648f, s.arg);
  if constexpr (U == 5) s.r = fmaf(s.j, 1.74845553e-07f, s.r);
  if constexpr (U == 6) s.v = fmaf(s.r, 0.159154943f, tw.w);
#ifdef X32_ABL_NOSIN   /* timing-only ablation (results wrong): a plain vector instruction in place of v_sin_f32 in the hooks */
  if constexpr (U == 7) s.v = s.v * 0.5f;
#else
  if constexpr (U == 7) s.v = __builtin_amdgcn_sinf(s.v);
#endif
}

// Workgroup shapes: XT = 512 threads (two waves per SIMD, 256 samples per pass) fills the chip at scale; XT = 256 (one wave per
// SIMD, 128 samples per pass) takes the launches that cannot give every CU a 256-sample chunk: twice the CUs take part and a
// wave alone on its SIMD finishes its tile in about half the time (B = 1 drop-in latency).  A tile's arithmetic does not
// depend on the shape, so results are bit-identical either way (sharding-independent).

// Development build (make EXTRA=-DX32_PHASE_PROFILE): waves 0 and 4 of every workgroup accumulate clock ticks per phase of the
// chunk loop; launch_t prints the shares every tenth launch of the mode-0 kernel (synchronous, stderr).
#ifdef X32_PHASE_PROFILE
#define X32_TICK(SLOT)                                             \
  {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    phase_ticks[SLOT] += (float)(now_ - phase_t0);                 \
    phase_t0 = now_;                                               \
  }
#else
#define X32_TICK(SLOT)
#endif

// Third-level (blob) fragments are fetched LOOK steps ahead into a ring of RL registers sets
#ifndef X32_LOOK
#define X32_LOOK 3   /* development A/B: 6 = a ring of 8 */
#endif
constexpr int LOOK = X32_LOOK, RL = LOOK <= 3 ? 4 : 8;
static_assert(LOOK == 3 || LOOK == 6, "look-ahead of the third-level fragments");

// One MFMA step for NT point tiles that share the weight fragments: the six partial products in the order of step6, each
// issued for tile 0 .. NT-1 in turn (independent accumulators alternate on the matrix pipe); WORK(slot) runs behind MFMA
// number slot - SLOT0 (6 NT slots per step).
#ifdef X32_ABL_TILE3_HALF   /* 1: L1's fourth output tile, 2: L1, L2 and L2^T */
#define X32_HALF_L1(mt) ((mt) == 3)
#define X32_HALF_L2(mt) ((mt) == 3 && X32_ABL_TILE3_HALF >= 2)
#else
#define X32_HALF_L1(mt) false
#define X32_HALF_L2(mt) false
#endif
// HALF (timing-only ablation X32_ABL_TILE3_HALF, results wrong): the step's six products as v_mfma_f32_16x16x32_bf16 -- the same
// number of matrix instructions and hook slots at half the pipe cycles each: what a 16-row remainder tile could save at best
template <int NT, int SLOT0, bool FIRST, bool HALF = false, class W>
__device__ __forceinline__ void stepN(f32x16 (&acc)[NT], const u32x4& ah, const u32x4& am, const u32x4& al,
                                      const u32x4 (&b)[NT][3], W&& work) {
  sfor<0, 6>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    sfor<0, NT>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      const u32x4& a_ = p == 0 ? al : ((p == 2 || p == 3) ? am : ah);
      const u32x4& b_ = b[T][p == 1 ? 2 : ((p == 2 || p == 4) ? 1 : 0)];
      if constexpr (HALF) {
        f32x4 c4 = {acc[T][0], acc[T][1], acc[T][2], acc[T][3]};
        if constexpr (FIRST && p == 0) c4 = f32x4{0.f, 0.f, 0.f, 0.f};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c4, 0, 0, 0);
        acc[T][0] = c4[0]; acc[T][1] = c4[1]; acc[T][2] = c4[2]; acc[T][3] = c4[3];
        if constexpr (FIRST && p == 0) {
#pragma unroll
          for (int r = 4; r < 16; ++r) acc[T][r] = 0.0f;
        }
      } else if constexpr (FIRST && p == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[T] = mfma32(a_, b_, zero);
      } else {
        acc[T] = mfma32(a_, b_, acc[T]);
      }
      work(ic<SLOT0 + p * NT + T>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  });
}

// one instruction of the evaluation of a feature of ANY kind (positional / angle / ones / pad; the arithmetic of
// feature_any in the kernel): 12 per feature.  isa != 0: angle feature (angle_encoder.py:16); is_one: the ones feature.
struct EvalAnyState { float arg, za, t, j, r, v; };
template <int U>
__device__ __forceinline__ void eval_any_item(EvalAnyState& s, const f32x4& tw, float isa, float ux, float uy, float th, bool is_one) {
  if constexpr (U == 0) s.arg = fmaf(tw.y, uy, tw.z);
  if constexpr (U == 1) s.arg = fmaf(tw.x, ux, s.arg);
  if constexpr (U == 2) s.za = th + tw.z;
  if constexpr (U == 3) s.za = s.za * tw.x;
  if constexpr (U == 4) s.arg = isa != 0.0f ? s.za : s.arg;
  if constexpr (U == 5) s.t = fmaf(s.arg, 0.159154943f, 12582912.0f);
  if constexpr (U == 6) s.j = s.t - 12582912.0f;
  if constexpr (U == 7) s.r = fmaf(s.j, -6.28318548202514648f, s.arg);
  if constexpr (U == 8) s.r = fmaf(s.j, 1.74845553e-07f, s.r);
  if constexpr (U == 9) s.v = fmaf(s.r, 0.159154943f, tw.w);
  if constexpr (U == 10) s.v = __builtin_amdgcn_sinf(s.v);
  if constexpr (U == 11) s.v = is_one ? 1.0f : s.v;
}

// NT = 32-sample tiles per wave.  Shapes in use (x32::launch_t): <512 threads, NT 1> two waves per SIMD; <256, 2> one wave
// per SIMD with two tiles sharing every weight fragment (half the LDS / L2 fragment traffic per sample, two independent
// accumulation and hook chains in one stream); <256, 1> for small launches.  A tile's arithmetic is the same in all of them.
template <int NKB, int MODE, int XT, int NT>
__global__ __launch_bounds__(XT, XT / 256) void onf_x32_kernel(const OnfKernelArgs a, const u32x4* __restrict__ img,
                                                         const u32x4* __restrict__ blob) {
  using C = Cfg<NKB>;
  constexpr bool FWD_ONLY = MODE == 2, TRAIN = MODE == 1;
  static_assert(!TRAIN || NT == 1, "the training pass is written for one tile per wave");
  constexpr int CH = (XT / 64) * 32 * NT;   // samples per workgroup pass: NT 32-sample tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  {   // image -> LDS, four 16-byte pieces per thread in flight
    constexpr int N16 = IMG_BYTES / 16;
    for (int k0 = threadIdx.x; k0 < N16; k0 += 4 * XT) {
      u32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (k0 + XT * q < N16) v[q] = img[k0 + XT * q];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (k0 + XT * q < N16) reinterpret_cast<u32x4*>(lds)[k0 + XT * q] = v[q];
    }
  }
  __syncthreads();
  // static priority for the younger half (MI355X_MICROARCH.md); the condition must be provably wave-uniform, or hipcc
  // lowers it to an exec mask around an UNCONDITIONAL s_setprio
#ifndef X32_NO_PRIO
  if (XT == 512 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif

  const OnfGeom& geo = a.geom;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  // ---- lane parts of every LDS address (formulas: tools/x32/emulate_x32.py) ----
  // Left to itself hipcc forms every (lane base + image offset + tile offset) once, in front of the persistent loop --
  // some 200 registers -- and spills them.  So each GEMM derives its few bases from an OPAQUE copy of the lane index where
  // it starts (a dozen integer instructions; the same device as onf_split.hip's NFOPP_REDERIVE), and nothing but the lane
  // index itself stays live across the chunk.
  int lane_v = lane;
#define X32_OPAQUE2(A) asm volatile("" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]))
  // forward W1: + 64 (kb >> 1) + 32 * RS1 * mt; [parity of kb][tile 3 ?]
  auto bases_w1f = [&](int (&w1f)[2][2]) __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane_v));
    const int jj = lane_v & 31, gg = lane_v >> 5, xs = (jj >> 2) & 3;
    const int lowE = 16 * ((gg ^ xs) & 3), lowO = 16 * (((2 | gg) ^ xs) & 3), row31 = min(96 + jj, W1_ZERO);
    w1f[0][0] = jj * RS1 + lowE; w1f[0][1] = row31 * RS1 + lowE; w1f[1][0] = jj * RS1 + lowO; w1f[1][1] = row31 * RS1 + lowO;
  };
  // forward W2: (base ^ (kb << 5)) + 32 * RS2 * mt; [tile 3 ?]
  auto bases_w2f = [&](int (&w2f)[2]) __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane_v));
    const int jj = lane_v & 31, gg = lane_v >> 5, sw2 = 16 * swz2(jj), row32 = min(96 + jj, W2_ZERO);
    w2f[0] = (jj * RS2 + sw2) ^ (gg << 4); w2f[1] = (row32 * RS2 + sw2) ^ (gg << 4);
  };
  // transposed reads: lane = (g, a, q, p); [eh][kb == 6 ?]
  //   W2: (base ^ (mt << 6)) + 16 * RS2 * kb;   W1: base + 16 * RS1 * kb + 64 * mt
  auto bases_tr = [&](auto w1_c, int (&t)[2][2]) __attribute__((always_inline)) {
    constexpr bool W1 = decltype(w1_c)::value;
    asm volatile("" : "+v"(lane_v));
    const int gg = lane_v >> 5, ta = (lane_v >> 4) & 1, tq = (lane_v >> 2) & 3, tp = lane_v & 3;
#pragma unroll
    for (int eh = 0; eh < 2; ++eh) {
      const int low = 16 * (((2 * ta + (tp & 1)) ^ (2 * eh + gg)) & 3) + 8 * (tp >> 1);
      const int r = 8 * eh + 4 * gg + tq;
      if (W1) { t[eh][0] = r * RS1 + low; t[eh][1] = min(96 + r, W1_ZERO) * RS1 + low; }
      else { t[eh][0] = r * RS2 + 64 * tq + low; t[eh][1] = min(96 + r, W2_ZERO) * RS2 + 64 * tq + low; }
    }
  };
  int ftl = O_FT + 64 * g, ftdl = O_FTD + 64 * g, isl = O_ISA + 16 * g, w3al = O_W3A + 16 * g, w3ll = O_W3L + 48 * g;
  int fin_rel = geo.fin - 4 * g;   // position == fin  <=>  16 kb + 8 (e >> 2) + (e & 3) == fin_rel

  const unsigned lane16 = lane * 16;
  const __amdgpu_buffer_rsrc_t blob_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(blob), 0, (int)C::BLOB_BYTES, 0x00020000);
  auto lo_frag = [&](int step) __attribute__((always_inline)) {
#ifdef X32_ABL_NOLO   /* timing-only ablation (results wrong): no third-level loads */
    return u32x4{(unsigned)step, lane16, 0u, 0u};
#elif defined(X32_ABL_LO_L1)   /* timing-only ablation (results wrong): the same loads from a 7 KB window -- served by the CU's L1 */
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(blob_rsrc, lane16, (step % 7) * 1024, 0));
#else
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(blob_rsrc, lane16, step * 1024, 0));
#endif
  };

  const long long n_work = TRAIN ? a.n_points : work_points(a);
  const long long n_chunks = (n_work + CH - 1) / CH;

#ifdef X32_PHASE_PROFILE
  float phase_ticks[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned long long phase_t0 = __builtin_readcyclecounter();
#endif
  if (n_chunks <= (long long)blockIdx.x) return;   // (an empty live list: nothing to do)
  // third-level fragments: ring of 4, three steps ahead, running on across the GEMMs (blob steps are consecutive) and,
  // at the end of a chunk, on into the first steps of the next
  u32x4 fl[RL];
#pragma unroll
  for (int q = 0; q < LOOK; ++q) fl[q] = lo_frag(q);
  // Samples.  Both lane halves of a wave need each tile's 32 poses, so forming them in all 64 lanes would do every draw,
  // interpolation and normalisation twice.  Each pose is formed by exactly ONE lane (the arithmetic of load_point: same
  // bits) and the halves exchange (5 cross-half moves).  NT = 2: the lower half forms tile 0's poses, the upper half tile
  // 1's.  NT = 1: every OTHER chunk the lower half forms this chunk's poses and the upper half the next chunk's.
  float nux = 0.f, nuy = 0.f, nth = 0.f;
  long long npidx = 0;
  bool have_next = false;
  float loss_acc = 0.f;   // TRAIN
  // TRAIN: 16-byte store of four consecutive positions of this lane's sample row (rows past P: dropped by the range check)
  auto st4 = [](const __amdgpu_buffer_rsrc_t& r, int voff, float x0, float x1, float x2, float x3) __attribute__((always_inline)) {
#ifdef X32_ABL_NOSTORE   /* timing-only ablation: the training pass without its factor stores (values kept alive) */
    asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(voff));
    return;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)}, r, voff, 0, 0);
  };
#ifdef X32_STAGGER   /* development A/B (MI355X_MICROARCH.md, two waves per SIMD, item 9): the second-dispatched half of the workgroup
                        starts X32_STAGGER x 8 k cycles late, so that SIMD partners are out of phase at the GEMM boundaries */
  if (XT == 512 && wave >= 4) {
#pragma unroll 1
    for (int q = 0; q < X32_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
  }
#endif
  for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    float ux[NT], uy[NT], th[NT];
    long long pidx[NT];
    // table bases: opaque per chunk, so that (base + small constant) stays an immediate offset of the LDS read instead
    // of one hoisted register per constant
    asm volatile("" : "+v"(ftl), "+v"(ftdl), "+v"(isl), "+v"(w3al), "+v"(w3ll), "+v"(fin_rel));
    if (NT == 1 && have_next) {
      ux[0] = nux; uy[0] = nuy; th[0] = nth; pidx[0] = npidx;
      have_next = false;
    } else {
      float x, y, ang;
      // NT = 1: this chunk (lower half) / the next one (upper half; past the last chunk: padding lanes, nothing stored)
      const long long p = NT == 1 ? (chunk + (g ? (long long)gridDim.x : 0)) * CH + wave * 32 + j : chunk * CH + wave * 64 + lane;
      long long row;
      if constexpr (TRAIN) {   // the fit's samples are explicit poses: the trajectory half of load_point (and the dozen scalar
                               // registers of its arguments) is not part of this kernel
        const bool ok = p < n_work;
        const float* q = a.points + (ok ? p : n_work - 1) * a.geom.point_dim;
        x = q[0]; y = q[1]; ang = a.geom.point_dim == 3 ? q[2] : 0.0f;
        row = ok ? p : a.n_points;
      } else {
        row = load_point(a, n_work, p, 0, x, y, ang);
      }
      const float sx = (x - geo.mean) / geo.sigma, sy = (y - geo.mean) / geo.sigma;
      const int rlo = (int)row, rhi = (int)(row >> 32);
      const float ox = __shfl_xor(sx, 32), oy = __shfl_xor(sy, 32), oa = __shfl_xor(ang, 32);
      const int olo = __shfl_xor(rlo, 32), ohi = __shfl_xor(rhi, 32);
      const long long orow = ((long long)ohi << 32) | (unsigned)olo;
      ux[0] = g ? ox : sx; uy[0] = g ? oy : sy; th[0] = g ? oa : ang; pidx[0] = g ? orow : row;      // the lower half's poses
      if constexpr (NT == 2) {
        ux[NT - 1] = g ? sx : ox; uy[NT - 1] = g ? sy : oy; th[NT - 1] = g ? ang : oa; pidx[NT - 1] = g ? row : orow;
      } else {
        nux = g ? sx : ox; nuy = g ? sy : oy; nth = g ? ang : oa; npidx = g ? row : orow;
        have_next = true;
      }
    }
    // TRAIN: one buffer resource per stored array over the wave's 32 rows (scalar base, byte offsets j * row + 16 g + a
    // compile-time constant), the label, and the sign words of a2 (positions 16 kb + 8 h + 4 g + r -> word 2 h + g, bit 4 kb + r)
    // Each resource is built (a dozen scalar instructions, from an opaque copy of the wave index) where its phase starts:
    // four descriptors alive across the chunk cost 16 scalar registers the kernel does not have.
    int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto train_rsrc = [&](float* base, int row_floats) __attribute__((always_inline)) {
      asm volatile("" : "+s"(wave_s));
      long long rows = a.n_points - (chunk * CH + wave_s * 32);
      rows = rows > 32 ? 32 : (rows < 0 ? 0 : rows);
#ifdef X32_ABL_STORE_SAME   /* timing-only ablation (results wrong): every wave stores to the same few rows -- no HBM traffic */
      const long long p0 = wave_s * 32;
#else
      const long long p0 = rows > 0 ? chunk * CH + wave_s * 32 : 0;
#endif
      return __builtin_amdgcn_make_buffer_rsrc(base + p0 * row_floats, 0, (int)(rows * row_floats * 4), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc((float*)nullptr, 0, 0, 0x00020000), r_dh1 = r_h1, r_de = r_h1, r_rec = r_h1;
    // (the lane's byte offsets are formed from an opaque copy of the lane index where a phase needs them, like the LDS bases)
    int vo_h = 0, vo_de = 0, vo_rec = 0;
    auto train_offsets = [&]() __attribute__((always_inline)) {
      asm volatile("" : "+v"(lane_v));
      const int jj = lane_v & 31, gg = lane_v >> 5;
      vo_h = jj * 448 + 16 * gg; vo_de = jj * (64 * NKB) + 16 * gg; vo_rec = jj * 48;
    };
    float yv = 0.f, rho[NT];
    unsigned sgn[NT][2];
#pragma unroll
    for (int T = 0; T < NT; ++T) { rho[T] = 1.0f; sgn[T][0] = sgn[T][1] = 0u; }
    if constexpr (TRAIN) {
      train_offsets();
      r_rec = train_rsrc(a.ws_u, 12);
      if (g == 0) st4(r_rec, vo_rec, ux[0], uy[0], 1.0f, th[0]);
    }
    u32x4 fh[2], fm[2];   // hi / mid fragments: this step and the next

    // generic evaluation of one input feature (any kind), used for the first block and the angle / ones / pad blocks
    auto feature_any = [&](auto special_c, int T, int kb, int e) __attribute__((always_inline)) {
      constexpr bool SPECIAL = decltype(special_c)::value;   // the block may hold angle / ones / pad positions
      const int off = 256 * kb + 128 * (e >> 2) + 16 * (e & 3);
      const f32x4 tw = lds128f(lds, ftl + off);
      float arg = fmaf(tw.x, ux[T], fmaf(tw.y, uy[T], tw.z));
      if constexpr (SPECIAL) {
        const float isa = *reinterpret_cast<const float*>(lds + isl + (off >> 2));
        const float za = (th[T] + tw.z) * tw.x;
        arg = isa != 0.0f ? za : arg;
      }
      const float v = sin_halfturns_hw(arg, tw.w);
      if constexpr (SPECIAL) return (16 * kb + 8 * (e >> 2) + (e & 3)) == fin_rel ? 1.0f : v;
      else return v;
    };
    auto features_upfront = [&](auto special_c, int kb, u32x4 (&out)[NT][3]) __attribute__((always_inline)) {
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int p = 0; p < 4; ++p)
          split_pair(feature_any(special_c, T, kb, 2 * p), feature_any(special_c, T, kb, 2 * p + 1), out[T], p);
    };

    X32_TICK(0)   // sampling
    // ================================================================ L1: a1 = W1ext in
    f32x16 acc1[4][NT];
    {
      int w1f[2][2];
      bases_w1f(w1f);
      // (the mid image's offset plus a tile offset does not fit the 16-bit immediate of an LDS read: its own bases)
      int w1m[2][2] = {{w1f[0][0] + O_W1M, w1f[0][1] + O_W1M}, {w1f[1][0] + O_W1M, w1f[1][1] + O_W1M}};
      X32_OPAQUE2(w1m);
      u32x4 bA[NT][3], bB[NT][3];
      features_upfront(std::false_type{}, 0, bA);
      // hooked preparation of the NEXT block's fragments: per tile 4 pairs x (16 evaluation + 11 split) instructions, 5 per
      // slot, the tiles' items alternating.  Schedule of a tile (item w): E0 E1 S0 E2 S1 E3 S2 S3; the table entries of
      // pair p+1 (shared by the tiles) are loaded at the start of E(p).
      f32x4 tw[2][2];
      float fv[NT][8];
      EvalState es[NT][2];
      SplitState ss[NT];
      int nxt_ft = 0;   // table byte address of the block being prepared
      auto load_pair = [&](int p) __attribute__((always_inline)) {
        const int off = nxt_ft + 128 * ((2 * p) >> 2) + 16 * ((2 * p) & 3);
        tw[p & 1][0] = lds128f(lds, off);
        tw[p & 1][1] = lds128f(lds, off + 16);
      };
      auto l1_item = [&](auto wc, u32x4 (&out)[NT][3]) __attribute__((always_inline)) {
        constexpr int T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
        constexpr int seg = w < 16 ? 0 : w < 32 ? 1 : w < 43 ? 2 : w < 59 ? 3 : w < 70 ? 4 : w < 86 ? 5 : w < 97 ? 6 : w < 108 ? 7 : 8;
        constexpr int start[9] = {0, 16, 32, 43, 59, 70, 86, 97, 108};
        constexpr int evp[9] = {0, 1, -1, 2, -1, 3, -1, -1, -1}, spp[9] = {-1, -1, 0, -1, 1, -1, 2, 3, -1};
        constexpr int u = w - start[seg];
        if constexpr (evp[seg] >= 0) {
          constexpr int p = evp[seg], which = u & 1, st = u >> 1;
          if constexpr (u == 0 && p < 3 && T == 0) load_pair(p + 1);
          eval_item<st>(es[T][which], tw[p & 1][which], ux[T], uy[T]);
          if constexpr (st == 7) fv[T][2 * p + which] = es[T][which].v;
        } else if constexpr (spp[seg] >= 0) {
          constexpr int p = spp[seg];
          split_item<u>(ss[T], fv[T][2 * p], fv[T][2 * p + 1], out[T], p);
        }
      };
      // the same for a block that may hold angle / ones / pad positions: per tile 4 pairs x (24 evaluation + 11 split)
      // instructions, 6 per slot; the arithmetic (and its order) is feature_any's
      EvalAnyState eas[NT][2];
      float isa2[2][2];
      int nxt_is = 0, nxt_pos = 0;
      auto load_pair_any = [&](int p) __attribute__((always_inline)) {
        load_pair(p);
        const int off = nxt_is + 32 * ((2 * p) >> 2) + 4 * ((2 * p) & 3);
        const f32x2 fl2 = *reinterpret_cast<const f32x2*>(lds + off);
        isa2[p & 1][0] = fl2.x; isa2[p & 1][1] = fl2.y;
      };
      auto l1_item_any = [&](auto wc, u32x4 (&out)[NT][3]) __attribute__((always_inline)) {
        constexpr int T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
        constexpr int seg = w < 24 ? 0 : w < 48 ? 1 : w < 59 ? 2 : w < 83 ? 3 : w < 94 ? 4 : w < 118 ? 5 : w < 129 ? 6 : w < 140 ? 7 : 8;
        constexpr int start[9] = {0, 24, 48, 59, 83, 94, 118, 129, 140};
        constexpr int evp[9] = {0, 1, -1, 2, -1, 3, -1, -1, -1}, spp[9] = {-1, -1, 0, -1, 1, -1, 2, 3, -1};
        constexpr int u = w - start[seg];
        if constexpr (evp[seg] >= 0) {
          constexpr int p = evp[seg], which = u & 1, st = u >> 1, e = 2 * p + which;
          if constexpr (u == 0 && p < 3 && T == 0) load_pair_any(p + 1);
          eval_any_item<st>(eas[T][which], tw[p & 1][which], isa2[p & 1][which], ux[T], uy[T], th[T],
                            (nxt_pos + 8 * (e >> 2) + (e & 3)) == fin_rel);
          if constexpr (st == 11) fv[T][e] = eas[T][which].v;
        } else if constexpr (spp[seg] >= 0) {
          constexpr int p = spp[seg];
          split_item<u>(ss[T], fv[T][2 * p], fv[T][2 * p + 1], out[T], p);
        }
      };
      // MFMA steps of block kb (fragments in bc), preparing block kb + 1 into bn when HOOK (1: a plain block, 2: any kind)
      auto l1_block = [&](auto hook_c, auto par_c, int kb, const u32x4 (&bc)[NT][3], u32x4 (&bn)[NT][3], auto first_c) __attribute__((always_inline)) {
        constexpr int HOOK = decltype(hook_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;   // block 0: the accumulators start here
        constexpr int PAR = decltype(par_c)::value;   // parity of kb
        const int kq = 64 * (kb >> 1);
        if constexpr (HOOK == 1) { nxt_ft = ftl + 256 * (kb + 1); load_pair(0); }
        if constexpr (HOOK == 2) { nxt_ft = ftl + 256 * (kb + 1); nxt_is = isl + 64 * (kb + 1); nxt_pos = 16 * (kb + 1); load_pair_any(0); }
        sfor<0, 4>([&](auto mtc) {
          constexpr int mt = decltype(mtc)::value;
          // this step's hi / mid were fetched one step ago; fetch the next step's (past the last block: a harmless
          // in-image read)
          if constexpr (mt < 3) {
            const int off = kq + (mt + 1 < 3 ? 32 * RS1 * (mt + 1) : 0);
            fh[(mt + 1) & 1] = lds128(lds, O_W1H + w1f[PAR][mt + 1 == 3] + off);
            fm[(mt + 1) & 1] = lds128(lds, w1m[PAR][mt + 1 == 3] + off);
          } else {
            const int off = 64 * ((kb + 1) >> 1);
            fh[0] = lds128(lds, O_W1H + w1f[PAR ^ 1][0] + off); fm[0] = lds128(lds, w1m[PAR ^ 1][0] + off);
          }
          fl[(4 * PAR + mt + LOOK) & (RL - 1)] = lo_frag(C::S_L1 + 4 * kb + mt + LOOK);
          __builtin_amdgcn_sched_barrier(0);
          stepN<NT, 6 * NT * mt, FIRST, X32_HALF_L1(mt)>(acc1[mt], fh[mt & 1], fm[mt & 1], fl[(4 * PAR + mt) & (RL - 1)], bc, [&](auto slot) {
            if constexpr (HOOK == 1) sfor<0, 5>([&](auto i) { l1_item(ic<5 * decltype(slot)::value + decltype(i)::value>{}, bn); });
            if constexpr (HOOK == 2) sfor<0, 6>([&](auto i) { l1_item_any(ic<6 * decltype(slot)::value + decltype(i)::value>{}, bn); });
          });
        });
      };
      constexpr int FS = C::FS;
      fh[0] = lds128(lds, O_W1H + w1f[0][0]); fm[0] = lds128(lds, w1m[0][0]);
      l1_block(ic<1>{}, ic<0>{}, 0, bA, bB, std::true_type{});
      l1_block(ic<1>{}, ic<1>{}, 1, bB, bA, std::false_type{});
#pragma unroll 1
      for (int kp = 1; kp < FS / 2 - 1; ++kp) {
        l1_block(ic<1>{}, ic<0>{}, 2 * kp, bA, bB, std::false_type{});
        l1_block(ic<1>{}, ic<1>{}, 2 * kp + 1, bB, bA, std::false_type{});
      }
      l1_block(ic<1>{}, ic<0>{}, FS - 2, bA, bB, std::false_type{});
      // blocks FS .. NKB-1 can hold angle / ones / pad features: each is prepared behind the block in front of it
      l1_block(ic<2>{}, ic<1>{}, FS - 1, bB, bA, std::false_type{});
      sfor<FS, NKB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int HK2 = kb + 1 < NKB ? 2 : 0;
        if constexpr ((kb - FS) % 2 == 0) l1_block(ic<HK2>{}, ic<(kb & 1)>{}, kb, bA, bB, std::false_type{});
        else l1_block(ic<HK2>{}, ic<(kb & 1)>{}, kb, bB, bA, std::false_type{});
      });
    }
    X32_TICK(1)   // L1
    float skipv[NT];   // position 100 = tile 3, g = 1, register 0: W3b . in + b3 (lanes g = 1)
#pragma unroll
    for (int T = 0; T < NT; ++T) skipv[T] = acc1[3][T][0];

    // ================================================================ L2: a2 = W2ext relu(a1)
    f32x16 acc2[4][NT];
    unsigned m1w[NT][2];   // [a1 > 0]: block kb, element e -> word kb >> 2, pushed from the low end in order
#pragma unroll
    for (int T = 0; T < NT; ++T) m1w[T][0] = m1w[T][1] = 0u;
    int w2f[2];
    bases_w2f(w2f);
    if constexpr (TRAIN) { train_offsets(); r_h1 = train_rsrc(a.ws_h1, 112); }
    {
      u32x4 bb[2][NT][3];
      float hv[NT][4];   // (TRAIN: the pairs alternate between [0..1] and [2..3]; every second pair completes a 16-byte store of h1)
      SplitState ss[NT];
      // relu + sign + split of one pair of block kb: 6 + 11 instructions per tile, the tiles' items alternating
      auto h1_item = [&](auto kbc, auto wc, u32x4 (&out)[NT][3]) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value, T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
        if constexpr (w < 68) {
          constexpr int p = w / 17, u = w % 17, t = kb >> 1, r0 = 8 * (kb & 1) + 2 * p, h0 = TRAIN ? 2 * (p & 1) : 0;
          if constexpr (u == 0) hv[T][h0] = relu1(acc1[t][T][r0]);
          if constexpr (u == 1) hv[T][h0 + 1] = relu1(acc1[t][T][r0 + 1]);
          if constexpr (u == 2) m1w[T][kb >> 2] = __builtin_amdgcn_alignbit(m1w[T][kb >> 2], 0u - __float_as_uint(hv[T][h0]), 31);
          if constexpr (u == 3) m1w[T][kb >> 2] = __builtin_amdgcn_alignbit(m1w[T][kb >> 2], 0u - __float_as_uint(hv[T][h0 + 1]), 31);
          if constexpr (u >= 4 && u < 15) split_item<u - 4>(ss[T], hv[T][h0], hv[T][h0 + 1], out[T], p);
          if constexpr (TRAIN && u == 15 && (p & 1)) st4(r_h1, vo_h + 64 * kb + 32 * (p >> 1), hv[T][0], hv[T][1], hv[T][2], hv[T][3]);
        }
      };
      sfor<0, 68 * NT>([&](auto w) { h1_item(ic<0>{}, w, bb[0]); });
      sfor<0, HK>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        const int kx = kb << 5;
        sfor<0, 4>([&](auto mtc) {
          constexpr int mt = decltype(mtc)::value;
          if constexpr (mt == 0 && kb == 0) {
            fh[0] = lds128(lds, O_W2H + w2f[0]); fm[0] = lds128(lds, O_W2M + w2f[0]);
          }
          if constexpr (mt < 3) {
            const int ad = (w2f[mt + 1 == 3] ^ kx) + (mt + 1 < 3 ? 32 * RS2 * (mt + 1) : 0);
            fh[(mt + 1) & 1] = lds128(lds, O_W2H + ad); fm[(mt + 1) & 1] = lds128(lds, O_W2M + ad);
          } else if constexpr (kb + 1 < HK) {
            const int ad = w2f[0] ^ ((kb + 1) << 5);
            fh[0] = lds128(lds, O_W2H + ad); fm[0] = lds128(lds, O_W2M + ad);
          }
          fl[(C::S_L2 + 4 * kb + mt + LOOK) & (RL - 1)] = lo_frag(C::S_L2 + 4 * kb + mt + LOOK);
          __builtin_amdgcn_sched_barrier(0);
          stepN<NT, 6 * NT * mt, kb == 0, X32_HALF_L2(mt)>(acc2[mt], fh[mt & 1], fm[mt & 1], fl[(C::S_L2 + 4 * kb + mt) & (RL - 1)], bb[kb & 1], [&](auto slot) {
            if constexpr (kb + 1 < HK)
              sfor<0, 3>([&](auto i) { h1_item(ic<kb + 1>{}, ic<3 * decltype(slot)::value + decltype(i)::value>{}, bb[(kb + 1) & 1]); });
          });
        });
      });
    }

    X32_TICK(2)   // L2
    if constexpr (FWD_ONLY) {
      // the summation order of the full kernel (two chains over the element parity, blocks in order): same logits bit for bit
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        float lgs[2] = {0.0f, 0.0f};
        sfor<0, HK>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          const f32x4 w3f[2] = {lds128f(lds, w3al + 64 * kb), lds128f(lds, w3al + 64 * kb + 32)};
#pragma unroll
          for (int e = 0; e < 8; ++e) lgs[e & 1] = fmaf(w3f[e >> 2][e & 3], relu1(acc2[kb >> 1][T][8 * (kb & 1) + e]), lgs[e & 1]);
        });
        float lg = lgs[0] + lgs[1] + (g == 1 ? skipv[T] : 0.0f);
        lg += __shfl_xor(lg, 32);
        if (g == 0 && pidx[T] < a.n_points) *reinterpret_cast<f32x4*>(a.out4 + pidx[T] * 4) = f32x4{lg, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < LOOK; ++q) fl[q] = lo_frag(q);
      continue;
    }

    // ================================================================ L2^T: W2ext^T dh2,  dh2 = W3a [a2 > 0]; the logit on the way
    f32x16 accd[4][NT];
    float lgs[NT][2];
#pragma unroll
    for (int T = 0; T < NT; ++T) lgs[T][0] = lgs[T][1] = 0.0f;
    int t2[2][2];
    bases_tr(std::false_type{}, t2);
    if constexpr (TRAIN) yv = pidx[0] < a.n_points ? a.labels[pidx[0]] : 0.0f;   // (used behind this GEMM)
    {
      constexpr int DH2_IP = TRAIN ? 14 : 12, DH2_PS = TRAIN ? 3 : 2;   // items per pair, per slot
      u32x4 bb[2][NT][3];
      f32x4 w3f[2];     // fp32 W3a of the block's 8 positions
      u32x4 w3c[3];     // its pre-split levels, B-fragment order
      float hv[NT][2];
      unsigned mk[NT][2];
      auto dh2_load = [&](int kb) __attribute__((always_inline)) {
        w3f[0] = lds128f(lds, w3al + 64 * kb); w3f[1] = lds128f(lds, w3al + 64 * kb + 32);
        w3c[0] = lds128(lds, w3ll + 96 * kb); w3c[1] = lds128(lds, w3ll + 96 * kb + 16); w3c[2] = lds128(lds, w3ll + 96 * kb + 32);
      };
      // one pair of block kb: relu, sign mask, logit terms, masked level words: 12 instructions per tile, tiles alternating
      // (TRAIN: + 2, the sign bits of a2 for the record, in front of the mask merge)
      auto dh2_item = [&](auto kbc, auto wc, u32x4 (&out)[NT][3]) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value, T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
        if constexpr (w < 4 * DH2_IP) {
          constexpr int p = w / DH2_IP, u0 = w % DH2_IP, t = kb >> 1, r0 = 8 * (kb & 1) + 2 * p, e0 = 2 * p;
          constexpr int u = !TRAIN ? u0 : (u0 < 8 ? u0 : (u0 < 10 ? 92 + u0 : u0 - 2));
          // (the 28 bit constants live in scalar registers and push a few of those into spill lanes; pushing the bits in with
          // v_alignbit instead frees them but moves the pressure to the vector file: 51 instead of 35 spilled registers, +3 %)
#ifndef X32_SGN_PUSH
          if constexpr (u == 100) sgn[T][p >> 1] |= mk[T][0] & (1u << (4 * kb + (e0 & 3)));
          if constexpr (u == 101) sgn[T][p >> 1] |= mk[T][1] & (1u << (4 * kb + (e0 & 3) + 1));
#else   /* development A/B: bits pushed in from the low end, element (kb, r) ends at bit 27 - (4 kb + r) */
          if constexpr (u == 100) sgn[T][p >> 1] = __builtin_amdgcn_alignbit(sgn[T][p >> 1], mk[T][0], 31);
          if constexpr (u == 101) sgn[T][p >> 1] = __builtin_amdgcn_alignbit(sgn[T][p >> 1], mk[T][1], 31);
#endif
          if constexpr (u == 0) hv[T][0] = relu1(acc2[t][T][r0]);
          if constexpr (u == 1) hv[T][1] = relu1(acc2[t][T][r0 + 1]);
          if constexpr (u == 2) mk[T][0] = 0u - __float_as_uint(hv[T][0]);
          if constexpr (u == 3) mk[T][1] = 0u - __float_as_uint(hv[T][1]);
          if constexpr (u == 4) mk[T][0] = (unsigned)((int)mk[T][0] >> 31);
          if constexpr (u == 5) mk[T][1] = (unsigned)((int)mk[T][1] >> 31);
          if constexpr (u == 6) lgs[T][0] = fmaf(w3f[e0 >> 2][e0 & 3], hv[T][0], lgs[T][0]);
          if constexpr (u == 7) lgs[T][1] = fmaf(w3f[(e0 + 1) >> 2][(e0 + 1) & 3], hv[T][1], lgs[T][1]);
          if constexpr (u == 8) mk[T][0] = __builtin_amdgcn_perm(mk[T][1], mk[T][0], 0x07060302);
          if constexpr (u == 9) out[T][0][p] = w3c[0][p] & mk[T][0];
          if constexpr (u == 10) out[T][1][p] = w3c[1][p] & mk[T][0];
          if constexpr (u == 11) out[T][2][p] = w3c[2][p] & mk[T][0];
        }
      };
      dh2_load(0);
      sfor<0, 4 * DH2_IP * NT>([&](auto w) { dh2_item(ic<0>{}, w, bb[0]); });
      sfor<0, HK>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int Z = kb == 6;
        if constexpr (kb + 1 < HK) dh2_load(kb + 1);
        sfor<0, 4>([&](auto mtc) {
          constexpr int mt = decltype(mtc)::value;
          auto fetch = [&](int kbn, int mtn, int zn, int slot) __attribute__((always_inline)) {
            const int a0 = (t2[0][zn] ^ (mtn << 6)) + (zn ? 0 : 16 * RS2 * kbn);
            const int a1 = (t2[1][zn] ^ (mtn << 6)) + (zn ? 0 : 16 * RS2 * kbn);
            fh[slot] = lds_tr(lds, O_W2H + a0, O_W2H + a1);
            fm[slot] = lds_tr(lds, O_W2M + a0, O_W2M + a1);
          };
          if constexpr (mt == 0 && kb == 0) fetch(0, 0, 0, 0);
          if constexpr (mt < 3) fetch(kb, mt + 1, Z, (mt + 1) & 1);
          else if constexpr (kb + 1 < HK) fetch(kb + 1, 0, kb + 1 == 6, 0);
          fl[(C::S_L2T + 4 * kb + mt + LOOK) & (RL - 1)] = lo_frag(C::S_L2T + 4 * kb + mt + LOOK);
          __builtin_amdgcn_sched_barrier(0);
          stepN<NT, 6 * NT * mt, kb == 0, X32_HALF_L2(mt)>(accd[mt], fh[mt & 1], fm[mt & 1], fl[(C::S_L2T + 4 * kb + mt) & (RL - 1)], bb[kb & 1], [&](auto slot) {
            if constexpr (kb + 1 < HK)
              sfor<0, DH2_PS>([&](auto i) { dh2_item(ic<kb + 1>{}, ic<DH2_PS * decltype(slot)::value + decltype(i)::value>{}, bb[(kb + 1) & 1]); });
          });
        });
      });
    }
    float logit[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      logit[T] = lgs[T][0] + lgs[T][1] + (g == 1 ? skipv[T] : 0.0f);
      logit[T] += __shfl_xor(logit[T], 32);
    }
    if constexpr (TRAIN) {   // BCE with logits (nerf_opt_planner.py:25,88): rho = (sigmoid(l) - y) / count; the record
      train_offsets();
      r_rec = train_rsrc(a.ws_u, 12);
      r_dh1 = train_rsrc(a.ws_dh1, 112);
      const bool t_valid = pidx[0] < a.n_points;
      const float l = logit[0];
      // e = exp(-|l|) on the hardware exponential (relative error ~1e-7 |l|, where e matters |l| is small); sigmoid and
      // softplus from it without cancellation -- the library expf / log1pf here cost two dozen spilled registers
      const float e = __builtin_amdgcn_exp2f(-1.44269504f * fabsf(l));
      const float inv = __builtin_amdgcn_rcpf(1.0f + e);
      const float sg = l >= 0.0f ? inv : e * inv;
      const float lp = fmaxf(l, 0.0f) - l * yv + 0.693147181f * __builtin_amdgcn_logf(1.0f + e);
      rho[0] = t_valid ? (sg - yv) * a.inv_count : 0.0f;
      if (t_valid && g == 0) loss_acc += lp * a.inv_count;
      if (g == 0) st4(r_rec, vo_rec + 16, rho[0], 0.0f, 0.0f, 0.0f);
#ifdef X32_SGN_PUSH
      sgn[0][0] = __builtin_bitreverse32(sgn[0][0]) >> 4; sgn[0][1] = __builtin_bitreverse32(sgn[0][1]) >> 4;
#endif
      __builtin_amdgcn_raw_buffer_store_b32(sgn[0][0], r_rec, vo_rec + 32 + 4 * g, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(sgn[0][1], r_rec, vo_rec + 40 + 4 * g, 0, 0);
    }
    X32_TICK(3)   // L2^T

    // ================================================================ dh1 = accd * [a1 > 0], dh1[skip row] = 1; three levels
    // Block 0 here; blocks 1..6 behind the steps of L1^T's first output tile (each step kb prepares block kb + 1).
    u32x4 dhl[HK][NT][3];
    float dv[NT][2], dq[NT][4];
    SplitState dss[NT];
    constexpr int DH1_IP = TRAIN ? 19 : 16, DH1_PS = TRAIN ? 13 : 11;   // items per pair; per slot behind L1^T's first tile
    // one pair of block kb: mask bits, masked values, (the skip row's constant), split: 16 instructions per tile
    // (TRAIN: + 3, rho * dh1 and every second pair its 16-byte store; the GEMM itself goes on with the unscaled dh1)
    auto dh1_item = [&](auto kbc, auto wc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
      if constexpr (w < 4 * DH1_IP) {
        constexpr int p = w / DH1_IP, u = w % DH1_IP, t = kb >> 1, e0 = 2 * p;
        if constexpr (TRAIN && u == 16) dq[T][2 * (p & 1)] = dv[T][0] * rho[T];
        if constexpr (TRAIN && u == 17) dq[T][2 * (p & 1) + 1] = dv[T][1] * rho[T];
        if constexpr (TRAIN && u == 18 && (p & 1)) st4(r_dh1, vo_h + 64 * kb + 32 * (p >> 1), dq[T][0], dq[T][1], dq[T][2], dq[T][3]);
        constexpr int nbits = (kb >> 2) == 0 ? 32 : 8 * (HK - 4);   // pushes into this mask word
        constexpr int k0 = 8 * (kb & 3) + e0;
        if constexpr (u == 0) dv[T][0] = __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)m1w[T][kb >> 2], nbits - 1 - k0, 1));
        if constexpr (u == 1) dv[T][1] = __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)m1w[T][kb >> 2], nbits - 2 - k0, 1));
        if constexpr (u == 2) dv[T][0] = __uint_as_float(__float_as_uint(accd[t][T][8 * (kb & 1) + e0]) & __float_as_uint(dv[T][0]));
        if constexpr (u == 3) dv[T][1] = __uint_as_float(__float_as_uint(accd[t][T][8 * (kb & 1) + e0 + 1]) & __float_as_uint(dv[T][1]));
        if constexpr (u == 4 && kb == 6 && p == 0) dv[T][0] = g == 1 ? 1.0f : dv[T][0];   // position 100: d logit / d skip
        if constexpr (u >= 5 && u < 16) split_item<u - 5>(dss[T], dv[T][0], dv[T][1], dhl[kb][T], p);
      }
    };
    sfor<0, 4 * DH1_IP * NT>([&](auto w) { dh1_item(ic<0>{}, w); });

    X32_TICK(4)   // dh1
    // ================================================================ L1^T: din = W1ext^T dh1, then the chain rule
    float gxs[NT][2], gys[NT][2], gt[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) { gxs[T][0] = gxs[T][1] = gys[T][0] = gys[T][1] = 0.f; gt[T] = 0.f; }
    int t1[2][2];
    bases_tr(std::true_type{}, t1);
    int t1m[2][2] = {{t1[0][0] + O_W1M, t1[0][1] + O_W1M}, {t1[1][0] + O_W1M, t1[1][1] + O_W1M}};
    X32_OPAQUE2(t1m);
    {
      u32x4 fl7[7];   // third level: ring of 7 = the steps of one output tile (LOOK steps ahead)
#pragma unroll
      for (int q = 0; q < LOOK; ++q) fl7[q] = fl[(C::S_L1T + q) & (RL - 1)];
      f32x16 accp[NT], accc[NT];
      u32x4 fhn, fmn;   // hi / mid fragments of the next tile's first step
      f32x4 tw[2][2];
      EvalState es[NT][2];
      float de[NT][2], eq[NT][4];
      int ep_ft = 0, ep_vo = 0;   // (TRAIN: byte offset of the epilogue tile in the sample's row of de)
      auto ep_load = [&](int pr) __attribute__((always_inline)) {   // pair pr = registers 2 pr, 2 pr + 1
        const int off = ep_ft + 128 * ((2 * pr) >> 2) + 16 * ((2 * pr) & 3);
        tw[pr & 1][0] = lds128f(lds, off);
        tw[pr & 1][1] = lds128f(lds, off + 16);
      };
      // chain-rule epilogue of the previous tile (plain positional features): per point tile 8 pairs x 22 instructions, 5 per
      // slot, the point tiles' items alternating (the table entries are shared)
      auto ep_item = [&](auto wc) __attribute__((always_inline)) {
        constexpr int T = decltype(wc)::value % NT, w = decltype(wc)::value / NT;
        if constexpr (w < 176) {
          constexpr int pr = w / 22, u = w % 22, which = u & 1, st = u >> 1, r = 2 * pr + which;
          if constexpr (u == 0 && pr < 7 && T == 0) ep_load(pr + 1);
          if constexpr (st < 8) eval_item<st>(es[T][which], tw[pr & 1][which], ux[T], uy[T]);
          if constexpr (st == 8) de[T][which] = accp[T][r] * es[T][which].v;
          if constexpr (!TRAIN) {
            if constexpr (st == 9) gxs[T][which] = fmaf(de[T][which], tw[pr & 1][which].x, gxs[T][which]);
            if constexpr (st == 10) gys[T][which] = fmaf(de[T][which], tw[pr & 1][which].y, gys[T][which]);
          } else {   // the fit needs rho * de itself, not d logit / d pose
            if constexpr (st == 9) eq[T][2 * (pr & 1) + which] = de[T][which] * rho[T];
            if constexpr (st == 10 && which == 1 && (pr & 1)) st4(r_de, ep_vo + 32 * (pr >> 1), eq[T][0], eq[T][1], eq[T][2], eq[T][3]);
          }
        }
      };
      // HOOK 1: the epilogue of tile mt - 1 behind this tile's steps;  2 (tile 0): dh1 block kb + 1 behind step kb
      auto l1t_tile = [&](auto hook_c, int mt, f32x16 (&acc)[NT]) __attribute__((always_inline)) {
        constexpr int HOOK = decltype(hook_c)::value;
        if constexpr (HOOK == 1) { ep_ft = ftdl + 512 * (mt - 1); ep_vo = vo_de + 128 * (mt - 1); ep_load(0); }
        auto fetch = [&](int kbn, int mtn, int zn, u32x4& oh, u32x4& om) __attribute__((always_inline)) {
          const int off = 64 * mtn + (zn ? 0 : 16 * RS1 * kbn);
          oh = lds_tr(lds, O_W1H + t1[0][zn] + off, O_W1H + t1[1][zn] + off);
          om = lds_tr(lds, t1m[0][zn] + off, t1m[1][zn] + off);
        };
        fh[0] = fhn; fm[0] = fmn;   // fetched during the previous tile's last step
        sfor<0, HK>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          if constexpr (kb + 1 < HK) fetch(kb + 1, mt, kb + 1 == 6, fh[(kb + 1) & 1], fm[(kb + 1) & 1]);
          else fetch(0, mt + 1, 0, fhn, fmn);   // past the last tile: a harmless in-image read
          {   // past the last step of the chunk: the next chunk's first steps
            const int st = C::S_L1T + HK * mt + kb + LOOK;
            fl7[(kb + LOOK) % 7] = lo_frag(st >= C::STEPS ? st - C::STEPS : st);
          }
          __builtin_amdgcn_sched_barrier(0);
          stepN<NT, 6 * NT * kb, kb == 0>(acc, fh[kb & 1], fm[kb & 1], fl7[kb], dhl[kb], [&](auto slot) {
            if constexpr (HOOK == 1) sfor<0, 5>([&](auto i) { ep_item(ic<5 * decltype(slot)::value + decltype(i)::value>{}); });
            if constexpr (HOOK == 2 && kb + 1 < HK)
              sfor<0, DH1_PS>([&](auto i) { dh1_item(ic<kb + 1>{}, ic<DH1_PS * (decltype(slot)::value - 6 * NT * kb) + decltype(i)::value>{}); });
          });
        });
      };
      // epilogue of a tile that can hold angle / ones / pad positions: evaluated after its steps
      auto ep_any = [&](int mt, const f32x16 (&acc)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = 512 * mt + 128 * (r >> 2) + 16 * (r & 3);
            const f32x4 e = lds128f(lds, ftdl + off);
            const float isa = *reinterpret_cast<const float*>(lds + isl + (off >> 2));
            float arg = fmaf(e.x, ux[T], fmaf(e.y, uy[T], e.z));
            const float za = (th[T] + e.z) * e.x;
            arg = isa != 0.0f ? za : arg;
            const float d = acc[T][r] * sin_halfturns_hw(arg, e.w);
            if constexpr (TRAIN) {
              eq[T][r & 3] = d * rho[T];
              // (an odd number of input blocks: the row of de ends in the middle of the last tile)
              if ((r & 3) == 3 && 2 * mt + (r >> 3) < NKB) st4(r_de, vo_de + 128 * mt + 32 * (r >> 2), eq[T][0], eq[T][1], eq[T][2], eq[T][3]);
            } else {
              gxs[T][r & 1] = fmaf(d, isa != 0.0f ? 0.0f : e.x, gxs[T][r & 1]);
              gys[T][r & 1] = fmaf(d, e.y, gys[T][r & 1]);
              gt[T] = fmaf(d, isa != 0.0f ? e.x : 0.0f, gt[T]);
            }
          }
      };
      {
        fhn = lds_tr(lds, O_W1H + t1[0][0], O_W1H + t1[1][0]);
        fmn = lds_tr(lds, t1m[0][0], t1m[1][0]);
      }
      l1t_tile(ic<2>{}, 0, accp);
      if constexpr (TRAIN) { train_offsets(); r_de = train_rsrc(a.ws_de, 16 * NKB); }
#pragma unroll 1
      for (int mt = 1; mt < C::NMT; ++mt) {
        l1t_tile(ic<1>{}, mt, accc);
#pragma unroll
        for (int T = 0; T < NT; ++T) accp[T] = accc[T];
      }
      X32_TICK(5)   // L1^T steps + hooked epilogues
      ep_any(C::NMT - 1, accp);
#pragma unroll
      for (int q = 0; q < LOOK; ++q) fl[q] = fl7[q];   // the first steps of the next chunk (fetched during the last tile)
    }
#pragma unroll
    for (int T = 0; T < NT && !TRAIN; ++T) {
      float gx = gxs[T][0] + gxs[T][1], gy = gys[T][0] + gys[T][1], gth = gt[T];
      gx += __shfl_xor(gx, 32); gy += __shfl_xor(gy, 32); gth += __shfl_xor(gth, 32);
      if (a.out4 && g == 0 && pidx[T] < a.n_points)
        *reinterpret_cast<f32x4*>(a.out4 + pidx[T] * 4) = f32x4{logit[T], gx / geo.sigma, gy / geo.sigma, gth};
    }
    X32_TICK(6)   // last epilogue + output
  }
  if constexpr (TRAIN) {   // loss per wave, fixed order; the host sums 8 rows per workgroup whatever the shape
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
    if (lane == 0) {
      a.loss_partial[blockIdx.x * 8 + wave] = loss_acc;
      if (XT == 256) a.loss_partial[blockIdx.x * 8 + 4 + wave] = 0.0f;
    }
  }
#ifdef X32_PHASE_PROFILE
  if (MODE == 0 && a.ws_u && (threadIdx.x == 0 || threadIdx.x == 256))
    for (int k = 0; k < 8; ++k) atomicAdd(a.ws_u + k + (threadIdx.x ? 8 : 0), phase_ticks[k]);
#endif
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Image + blob per (device, stream), kept by csrc/onf_x32.hip: the prep kernel rewrites them on the launch stream in front of
// every launch (the parameters may have changed), so launches of one stream are ordered by the stream itself.
struct Slot {
  hipStream_t stream; void* ptr; size_t bytes; bool used; unsigned long long stamp;
  // what the image in `ptr` was built from: parameter buffer, its registered content version (0 = unknown), geometry
  const float* params; unsigned long long version; OnfGeom geom; int nkb;
};
int buffers_for_stream(size_t bytes, hipStream_t stream, void** out, Slot** slot_out);
void slot_built(Slot* slot, const float* params, unsigned long long version, const OnfGeom& geom, int nkb);
// one feature dimension each (csrc/onf_x32_k*.hip): mode 0 logits + input gradient, 1 training pass, 2 logits only
int launch_nkb14(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out);
int launch_nkb13(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out);
int launch_nkb8(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out);
int launch_nkb7(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out);

template <int NKB, int MODE, int XT, int NT>
static int launch_shape(const OnfKernelArgs& a, hipStream_t stream, int* grid_out = nullptr) {
  using C = Cfg<NKB>;
  constexpr int CH = (XT / 64) * 32 * NT;
  static bool attr_set[MAX_DEVICES] = {};
  auto kern = onf_x32_kernel<NKB, MODE, XT, NT>;
  int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), IMG_BYTES, attr_set);
  if (rc != NFOPP_OK) return rc;
  void* buf = nullptr;
  Slot* slot = nullptr;
  rc = buffers_for_stream(IMG_BYTES + C::BLOB_BYTES, stream, &buf, &slot);
  if (rc != NFOPP_OK) return rc;
  u32x4* img = reinterpret_cast<u32x4*>(buf);
  u32x4* blob = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(buf) + IMG_BYTES);
  // The image is a function of the parameter values.  It is rebuilt in front of every launch unless the caller vouches
  // for the buffer's content with a version (nfopp_onf_params_version) and this stream's image was built from that very
  // (buffer, version, geometry): a frozen field then costs no prep launch.
  const unsigned long long ver = onf_params_version_of(a.params);
  const bool fresh = ver != 0 && slot->version == ver && slot->params == a.params && slot->nkb == NKB &&
                     memcmp(&slot->geom, &a.geom, sizeof(OnfGeom)) == 0;
  if (!fresh) {
    constexpr int N_PIECES = IMG_BYTES / 16 + (C::STEPS + 4) * 64;
    hipLaunchKernelGGL(x32_prep_kernel<NKB>, dim3((N_PIECES + 255) / 256), dim3(256), 0, stream, a.geom, a.params, img, blob);
    NFOPP_HIP(hipGetLastError());
    slot_built(slot, a.params, ver, a.geom, NKB);
  }
  const long long n_chunks = (a.n_points + CH - 1) / CH;
  long long grid = query_cus();
  if (grid > n_chunks) grid = n_chunks;
  if (grid_out) *grid_out = (int)grid;
#ifdef X32_PHASE_PROFILE
  if (MODE == 0) {   // development only: synchronous, prints to stderr
    static float* dbg = nullptr;
    if (!dbg) NFOPP_HIP(hipMalloc(&dbg, 64));
    NFOPP_HIP(hipMemsetAsync(dbg, 0, 64, stream));
    OnfKernelArgs b = a;
    b.ws_u = dbg;   // unused by this mode: carries the tick buffer
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(XT), IMG_BYTES, stream, b, (const u32x4*)img, (const u32x4*)blob);
    float h[16];
    NFOPP_HIP(hipMemcpyAsync(h, dbg, 64, hipMemcpyDeviceToHost, stream));
    NFOPP_HIP(hipStreamSynchronize(stream));
    static int calls = 0;
    if (++calls % 10 == 0) {
      const char* names[7] = {"sampling", "L1", "L2", "L2T+logit", "dh1", "L1T", "tail+out"};
      for (int w = 0; w < 2; ++w) {
        float tot = 0;
        for (int k = 0; k < 7; ++k) tot += h[8 * w + k];
        fprintf(stderr, "[x32 phase profile] wave %d of %lld workgroups: %.0f ticks per workgroup\n", 4 * w, grid, tot / grid);
        for (int k = 0; k < 7; ++k) fprintf(stderr, "   %-10s %5.1f %%\n", names[k], 100.0f * h[8 * w + k] / tot);
      }
    }
    return NFOPP_OK;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(XT), IMG_BYTES, stream, a, (const u32x4*)img, (const u32x4*)blob);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

template <int NKB, int MODE>
static int launch_t(const OnfKernelArgs& a, hipStream_t stream, int* grid_out = nullptr) {
#ifdef X32_THREADS   /* development A/B: one shape at every size (X32_TILES = 32-sample tiles per wave) */
#ifndef X32_TILES
#define X32_TILES 1
#endif
  return launch_shape<NKB, MODE, X32_THREADS, MODE == 1 ? 1 : X32_TILES>(a, stream, grid_out);
#else
  return a.n_points < (long long)query_cus() * 256 ? launch_shape<NKB, MODE, 256, 1>(a, stream, grid_out)
                                                   : launch_shape<NKB, MODE, 512, 1>(a, stream, grid_out);
#endif
}

// the three modes of one feature dimension (used by csrc/onf_x32_k*.hip)
template <int NKB>
static int launch_modes(const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out) {
  return mode == 0 ? launch_t<NKB, 0>(a, stream, grid_out) : mode == 1 ? launch_t<NKB, 1>(a, stream, grid_out)
                                                                        : launch_t<NKB, 2>(a, stream, grid_out);
}

}  // namespace x32
}  // namespace nfopp
