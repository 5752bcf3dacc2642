// LDS image layout, staging and feature evaluation shared by the fused ONF kernels: csrc/onf_fused.hip (fp32 MFMA)
// and csrc/onf_split.hip (bf16x3 split-precision MFMA).  See the header comment of onf_fused.hip for the design.
#pragma once
#include <type_traits>

#include "onf_kernel.h"

namespace nfopp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = NFOPP_HIDDEN;
constexpr int HT = 7;     // hidden tiles of 16 (tile 6 carries features 96..99 in rows (g, r = 0))
constexpr int S2 = 129;   // LDS row stride of W2 (floats), = 1 mod 32
#ifndef NFOPP_THREADS
#define NFOPP_THREADS 512   /* development: 256 = one wave per SIMD (tools/split_speed.py A/B) */
#endif
constexpr int THREADS = NFOPP_THREADS;
constexpr int WAVES = THREADS / 64;
constexpr int KSTEPS = 25;  // hidden k-steps: ks -> tile ks>>2, register ks&3 (tile 6 only register 0)


constexpr bool BYTES_OK(int floats) { return size_t(floats) * 4 <= 160 * 1024; }

template <int NKT>
struct Lds {
  // layout P spreads a pair of input tiles over a block of 32 features, so an odd NKT still indexes a full block
  static constexpr int NF = 32 * ((NKT + 1) / 2);           // input-feature slots (>= fin, zero padded)
  static constexpr int S1 = (NF > 128) ? 225 : 129;         // = 1 mod 32, > NF
  static constexpr int W1 = 0;
  static constexpr int W2 = W1 + H * S1;
  static constexpr int FT = ((W2 + H * S2 + 3) / 4) * 4;  // feature table, FTS floats per input feature
  static constexpr int FTS = 12;                          // (wx wx wy wy | b b fr fr | qh qh w3b w3b): packed-math pairs
  // The table is read with 16-byte accesses by lanes whose feature indices differ by 16 (lane group bit g & 1) and by 4
  // (g >> 1).  ds_read_b128 serves lanes of BOTH values of g & 1 in one cycle group, and 16 entries = 192 floats = 0 mod
  // 64 banks: a plain [feature][FTS] table is 2-way conflicted on every read (22 % of the kernel's LDS cycles in round
  // 1's counters).  So the table is stored in two halves by bit 4 of the feature index, the second half 32 banks further.
  static constexpr int FT_HALF = ((NF / 2) * FTS / 64) * 64 + 64 + 32;   // float offset of the second half, = 32 mod 64
  static constexpr int FT_FLOATS = FT_HALF + (NF / 2) * FTS;
  // entry of feature f, floats from the start of LDS;  ft_rel(x): the part that does not depend on the lane, for
  // x = 32 K + c with c < 16 (block / tile bases and row offsets: bit 4 clear)
#ifndef NFOPP_FT_PLAIN
  __host__ __device__ static constexpr int ft(int f) { return FT + FT_HALF * ((f >> 4) & 1) + FTS * (((f >> 5) << 4) | (f & 15)); }
  __host__ __device__ static constexpr int ft_rel(int x) { return FTS * (x - 16 * (x >> 5)); }
#else   /* development A/B: the plain [feature][FTS] table of round 1 (2-way conflicted 16-byte reads) */
  __host__ __device__ static constexpr int ft(int f) { return FT + FTS * f; }
  __host__ __device__ static constexpr int ft_rel(int x) { return FTS * x; }
#endif
  static constexpr int B1 = FT + ((FT_FLOATS + 3) / 4) * 4;   // 112 each, D-layout indexable (see fill)
  static constexpr int B2 = B1 + 16 * HT;
  static constexpr int W3A = B2 + 16 * HT;
  static constexpr int W3B = W3A + 16 * HT;               // NF skip weights
  static constexpr int ISA = W3B + NF;                    // NF flags: 1.0 = angle feature
  // compact copy of the table for the paths that evaluate ONE feature per lane behind MFMAs (onf_split.hip's shadow
  // work): (wx, wy, b, qh) in one 16-byte entry, the same two-halves placement
  static constexpr int FC = ((ISA + NF + 3) / 4) * 4;
  static constexpr int FC_HALF = ((NF / 2) * 4 / 64) * 64 + 64 + 32;     // = 32 mod 64
  __host__ __device__ static constexpr int fc(int f) { return FC + FC_HALF * ((f >> 4) & 1) + 4 * (((f >> 5) << 4) | (f & 15)); }
  __host__ __device__ static constexpr int fc_rel(int x) { return 4 * (x - 16 * (x >> 5)); }
  static constexpr int TOTAL = FC + FC_HALF + (NF / 2) * 4;
  static_assert(BYTES_OK(TOTAL), "LDS image exceeds 160 KB");
  static constexpr size_t BYTES = size_t(TOTAL) * 4;
};

__device__ __forceinline__ int base_p(int t) { return 32 * (t >> 1) + 8 * (t & 1); }

// max(x, 0) as ONE integer instruction: for IEEE floats max_i32(bits, 0) clears every negative value (and -0)
// and keeps every positive one.  (fmaxf lowers to canonicalize + max; an inline-asm v_max_f32 would hide the
// VALU->MFMA operand hazard from hipcc.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- exact three-level bf16 split: x = hi + mid + lo, every level the top 16 bits of the running residual --------
// (truncation keeps 8 significant bits per level; the residual subtractions are exact in fp32)
__host__ __device__ __forceinline__ unsigned split_level(float& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned top = __float_as_uint(x) & 0xffff0000u;
  x = x - __uint_as_float(top);
#else
  unsigned bits; __builtin_memcpy(&bits, &x, 4);
  const unsigned top = bits & 0xffff0000u;
  float t; __builtin_memcpy(&t, &top, 4);
  x = x - t;
#endif
  return top >> 16;
}
__device__ __forceinline__ float pack_hi_mid(float w) {
  const unsigned hi = split_level(w), mid = split_level(w);
  return __uint_as_float((hi << 16) | mid);
}
__device__ __forceinline__ unsigned lo_level(float w) {
  split_level(w); split_level(w);
  return __float_as_uint(w) >> 16;   // third level: exactly representable, low half-word is zero
}

// ---- stage the flat parameter buffer into LDS ---------------------------------------------------------------
// PACKED: weight words hold (bf16 hi | bf16 mid) of the weight instead of its fp32 value (onf_split.hip)
template <int NKT, bool TRAIN, bool PACKED = false>
__device__ void fill_lds(float* lds, const OnfKernelArgs& a) {
  using L = Lds<NKT>;
  const OnfGeom& g = a.geom;
  const float* P = a.params;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < H * L::S1; idx += THREADS) {
    int row = idx / L::S1, col = idx - row * L::S1;
    const float w = col < g.fin ? P[g.off_w1 + row * g.fin + col] : 0.0f;
    lds[L::W1 + idx] = PACKED ? pack_hi_mid(w) : w;
  }
  for (int idx = tid; idx < H * S2; idx += THREADS) {
    int row = idx / S2, col = idx - row * S2;
    const float w = col < H ? P[g.off_w2 + row * H + col] : 0.0f;
    lds[L::W2 + idx] = PACKED ? pack_hi_mid(w) : w;
  }
  for (int f = tid; f < L::NF; f += THREADS) {
    // feature = sin(arg + q*pi/2), q = 1 for cosine features, stored as qh = q * NFOPP_Q_UNIT; every scalar is
    // stored twice (a packed-math pair)
    float wx = 0.f, wy = 0.f, b = 0.f, fr = 0.f, w3b = 0.f, qh = 0.f, is_angle = 0.f;
    if (f < g.n_enc) {
      wx = P[g.off_we + 2 * f];
      wy = P[g.off_we + 2 * f + 1];
      b = g.off_be >= 0 ? P[g.off_be + f] : 0.0f;
      qh = (g.n_enc > g.n_sin && f >= g.n_sin) ? NFOPP_Q_UNIT : 0.0f;
      w3b = P[g.off_w3 + H + f];
    } else if (f < g.fin) {
      int k = f - g.n_enc;
      b = P[g.off_ang_b + k];
      fr = P[g.off_ang_f + k];
      qh = k >= g.ang_dim ? NFOPP_Q_UNIT : 0.0f;
      is_angle = 1.0f;
      w3b = P[g.off_w3 + H + f];
    } else if (TRAIN && f == a.aug_feature) {
      qh = NFOPP_Q_UNIT;  // all weights zero: sin(0 + pi/2) = 1
    }
    float* e = lds + L::ft(f);
    e[0] = e[1] = wx; e[2] = e[3] = wy; e[4] = e[5] = b; e[6] = e[7] = fr;
    e[8] = e[9] = qh; e[10] = e[11] = w3b;
    float* c = lds + L::fc(f);
    c[0] = wx; c[1] = wy; c[2] = b; c[3] = qh;
    lds[L::W3B + f] = w3b;
    lds[L::ISA + f] = is_angle;
  }
  // hidden-indexed vectors: entries 0..95 natural; entries 96 + 4g + r hold feature 96+g for r == 0, else 0
  for (int k = tid; k < 16 * HT; k += THREADS) {
    int h = k < 96 ? k : (((k - 96) & 3) == 0 ? 96 + ((k - 96) >> 2) : -1);
    lds[L::B1 + k] = h >= 0 ? P[g.off_b1 + h] : 0.0f;
    lds[L::B2 + k] = h >= 0 ? P[g.off_b2 + h] : 0.0f;
    lds[L::W3A + k] = h >= 0 ? P[g.off_w3 + h] : 0.0f;
  }
}

// Two input features at once (packed fp32): table entries e0 = (wx, wy, b, fr), (qh, is_angle) per slot,
// points (ux, uy, th) per slot; DERIV adds half a turn of pi/2... i.e. evaluates d feature / d arg (L1T epilogue).
template <bool MAY_BE_ANGLE, bool DERIV>
__device__ __forceinline__ f32x2 features2(f32x2 wx, f32x2 wy, f32x2 b, f32x2 fr, f32x2 qh, f32x2 is_angle,
                                           f32x2 ux, f32x2 uy, f32x2 th) {
  f32x2 arg = fma2(wx, ux, fma2(wy, uy, b));  // encoding_layer: W_e u + b_e (onf_model.py:39)
  if (MAY_BE_ANGLE) {
    const f32x2 za = (th + b) * fr;            // (theta + b) * f (angle_encoder.py:16)
    arg.x = is_angle.x != 0.0f ? za.x : arg.x;
    arg.y = is_angle.y != 0.0f ? za.y : arg.y;
  }
  return sin_halfturns2(arg, DERIV ? qh + splat2(NFOPP_Q_UNIT) : qh);
}

// ---- the wave's points: explicit poses or collision samples of a trajectory batch --------------------------------
// Number of samples this launch walks: all of them, or only those of the live trajectories (early stop).
__device__ __forceinline__ long long work_points(const OnfKernelArgs& a) {
  return a.live ? (long long)a.live[0] * (a.n_way - 1) : a.n_points;
}

// Pose of work item p (padding lanes past n_work repeat the last item), in two halves so that a persistent kernel can
// issue the loads of its NEXT chunk early: point_fetch issues the global loads and point_finish does the arithmetic.
// load_point = both.  Returns the row the item's outputs belong to -- a.n_points for a padding lane, so
// `row < a.n_points` is the store predicate.  Trajectory mode draws / reads the interpolation parameter and forms the
// collision sample: constrained:78-81 (SE(2)) / nerf:113-117 (2-D).
struct RawPoint {
  float qa[3], qb[3];       // explicit mode: qa = the pose;  trajectory mode: traj[:-1][j], traj[1:][j]
  float t;                  // t_mode 0: the injected draw
  long long row;            // output row, a.n_points for a padding lane
  unsigned long long gp;    // global sample index (Philox counter)
  bool valid;
};

__device__ __forceinline__ void point_fetch(const OnfKernelArgs& a, long long n_work, long long p, RawPoint& r) {
  r.valid = p < n_work;
  if (!r.valid) p = n_work - 1;
  r.qa[2] = r.qb[0] = r.qb[1] = r.qb[2] = 0.f;
  r.t = 0.f;
  r.gp = 0;
  if (a.points) {
    const float* q = a.points + p * a.geom.point_dim;
    r.qa[0] = q[0]; r.qa[1] = q[1];
    if (a.geom.point_dim == 3) r.qa[2] = q[2];
    r.row = r.valid ? p : a.n_points;
    return;
  }
  const int nseg = a.n_way - 1;
  long long b;
  int j;
  if (a.n_points < 0x7fffffffLL) {  // wave-uniform: 32-bit division for every realistic batch
    const unsigned b32 = (unsigned)p / (unsigned)nseg;
    b = b32;
    j = (int)((unsigned)p - b32 * (unsigned)nseg);
  } else {
    b = p / nseg;
    j = (int)(p - b * nseg);
  }
  if (a.live) b = a.live[1 + b];
  const long long row = b * nseg + j;
  if (a.t_mode == 0) r.t = a.t[row];
  else r.gp = (unsigned long long)((a.traj_index_offset + b) * nseg + j);
  const float* qa = a.traj + (b * a.n_way + j) * a.dim;  // traj[:-1]
  const float* qb = qa + a.dim;                           // traj[1:]
  r.qa[0] = qa[0]; r.qa[1] = qa[1]; r.qb[0] = qb[0]; r.qb[1] = qb[1];
  if (a.dim == 3) { r.qa[2] = qa[2]; r.qb[2] = qb[2]; }
  r.row = row;
}

__device__ __forceinline__ long long point_finish(const OnfKernelArgs& a, const RawPoint& r, int g, float& x, float& y,
                                                   float& ang) {
  ang = 0.f;
  if (a.points) {
    x = r.qa[0]; y = r.qa[1];
    if (a.geom.point_dim == 3) ang = r.qa[2];
    return r.row;
  }
  float tt = r.t;
  if (a.t_mode != 0) {
    tt = philox_uniform(a.seed, r.gp, a.rng_offset);
    if (g == 0 && r.valid) a.t[r.row] = tt;
  }
  // every product and sum rounded on its own, as the reference's separate torch ops do (no fused multiply-add): the
  // sample is then bit-identical to the oracle's for the same t, and K1 sees the reference's inputs
  if (a.dim == 3) {
    // constrained:79-81  p = traj[1:] + t * wrap-theta(traj[:-1] - traj[1:])
    float dx = r.qa[0] - r.qb[0], dy = r.qa[1] - r.qb[1], dth = wrap_angle(r.qa[2] - r.qb[2]);
    x = add_mul_unfused(r.qb[0], tt, dx); y = add_mul_unfused(r.qb[1], tt, dy); ang = add_mul_unfused(r.qb[2], tt, dth);
  } else {
    // nerf:117  p = traj[1:] * (1 - t) + traj[:-1] * t
    float omt = 1.0f - tt;
    x = mix_unfused(r.qb[0], omt, r.qa[0], tt); y = mix_unfused(r.qb[1], omt, r.qa[1], tt);
  }
  return r.valid ? r.row : a.n_points;
}

__device__ __forceinline__ long long load_point(const OnfKernelArgs& a, long long n_work, long long p, int g, float& x,
                                                float& y, float& ang) {
  RawPoint r;
  point_fetch(a, n_work, p, r);
  return point_finish(a, r, g, x, y, ang);
}

}  // namespace nfopp
