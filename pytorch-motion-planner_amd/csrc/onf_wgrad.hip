// ONF fitting step at scale: weight gradients as fp32-MFMA GEMMs over the sample axis.
//
// Replaces (reference, PyTorch-CPU + autograd): `loss.backward()` of `_optimize_collision_model`
// nfop/nerf_opt_planner.py:83-89 when the number of samples is large (batched planning with a shared field:
// P = B * (N + 109) samples per step, BASELINE config 5).  Small P keeps the per-sample path of csrc/onf_train.hip.
//
// Pipeline (all reductions in a fixed order -> bitwise reproducible, no float atomics):
//   1. pass 1, by matrix path: onf_x32_kernel<.., MODE 1> (csrc/onf_x32.hip; default, factors stored by index: "x32 order"),
//      onf_split_kernel<.., MODE 1> or onf_fwd_bwd_kernel<.., TRAIN> (csrc/onf_split.hip / onf_fused.hip; factors in their
//      slot orders): forward + backward per sample on the MFMA chain.  It writes
//      only the factors pass 2 cannot rebuild cheaply --  h1 | dh1 | de  (slot order, with a ones column and a rho row
//      so that bias and W3 gradients fall out of the same GEMMs) and a 48-byte record (u, rho, sign bits of a2) -- plus
//      per-wave loss partials and per-wave partials of dW3[:100] = sum_p rho_p h2_p (h2 never leaves that kernel);
//      1.84 KB per sample instead of the 3.6 KB of a full factor dump;
//   2. onf_wgrad_kernel: persistent workgroups, 16-sample chunks staged into LDS as one combined row per sample
//      (stride = 16 mod 32 -> conflict-free operand reads); the input features `in` are RE-EVALUATED from u (same
//      arithmetic as pass 1) and dh2 = rho * W3a * [a2 > 0] is rebuilt from the record; 8*NKT+49 output tiles over 8 waves:
//        G1 = dh1^T in   (dW1, db1, dW3[100:], db3)     G2 = dh2^T h1 (dW2, db2)     G3 = de^T u (dWe, dbe, angle grads)
//      next chunk prefetched into registers while the current one multiplies; per-workgroup partial tiles to HBM;
//   3. onf_wgrad_reduce_kernel (sum over workgroups / waves) and onf_wgrad_gather_kernel (slot -> parameter index).
#include "onf_layout.h"

namespace nfopp {

constexpr int WG_THREADS = 512;
constexpr int WG_WAVES = 8;
constexpr int KC = 16;  // samples per LDS chunk (4 k-steps: every operand offset stays a ds_read immediate)
constexpr int HS = 112; // hidden-side row length (7 tiles)

// out[e] = sum over rows of partial[row][e] for MANY rows (per-wave partials): one block per element; thread t adds rows
// t, t + 256, ... in ascending order, then a fixed binary tree over the 256 threads.  Deterministic for a given row count.
__global__ __launch_bounds__(256) void onf_rows_reduce_kernel(const float* partial, float* out, int n_elems, int n_rows) {
  __shared__ float tree[256];
  const int e = blockIdx.x, t = threadIdx.x;
  float s = 0.f;
  for (int r = t; r < n_rows; r += 256) s += partial[(long long)r * n_elems + e];
  tree[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) tree[t] += tree[t + o];
    __syncthreads();
  }
  if (t == 0) out[e] = tree[0];
}

// ---- slot maps (see csrc/onf_fused.hip for the layouts) --------------------------------------------------------
__host__ __device__ inline int slot_layout_p(int f) {  // input features, h2 / dh2 (features < 96)
  const int o = f & 31, tpar = (o >> 3) & 1, op = o - 8 * tpar, ap = op & ~3, r = op & 3;
  const int g = ((ap >> 4) & 1) | (((ap >> 2) & 1) << 1);
  return 16 * (2 * (f >> 5) + tpar) + 4 * g + r;
}
__host__ __device__ inline int slot_layout_q(int f) {  // h1 / dh1 (features < 96)
  const int o = f & 15, aq = o & 12, r = o & 3;
  const int g = ((aq >> 3) & 1) | (((aq >> 2) & 1) << 1);
  return 16 * (f >> 4) + 4 * g + r;
}
__host__ __device__ inline int hidden_slot(int h, bool layout_q) {
  if (h >= 96) return 96 + 4 * (h - 96);
  return layout_q ? slot_layout_q(h) : slot_layout_p(h);
}
constexpr int AUG_HIDDEN_SLOT = 97;  // tile 6, g = 0, r = 1

template <int NKT>
struct WgLayout {
  static constexpr int WIN = 16 * NKT;
  // combined LDS row of one sample; +16 pad keeps the stride = 16 mod 32
  static constexpr int C_DH1 = 0, C_IN = HS, C_DH2 = C_IN + WIN, C_H1 = C_DH2 + HS, C_DE = C_H1 + HS, C_U = C_DE + WIN,
                       STRIDE = C_U + 16 + 16;
  static_assert(STRIDE % 32 == 16, "operand reads are conflict-free for a row stride of 16 mod 32");
  static constexpr int NTILES = 8 * NKT + 49;
  static constexpr int TPW = (NTILES + WG_WAVES - 1) / WG_WAVES;  // tiles per wave
  static constexpr int F4_PER_SAMPLE = (2 * HS + WIN + 12) / 4;    // stored: dh1 | h1 | de | record
  static constexpr int F4_PER_THREAD = (KC * F4_PER_SAMPLE + WG_THREADS - 1) / WG_THREADS;
  static constexpr int REBUILD_F4 = (HS + WIN) / 4;                // rebuilt per sample: dh2 | in
  // LDS: two buffers of { KC combined rows | record tail: rho [KC], a2 sign words [KC][4] } (chunk k multiplies from
  //      one while chunk k+1 is committed and rebuilt in the other) | feature table as six planes [6][WIN] (wx | wy | b | fr | qh | is_angle,
  //      slot order: a lane reads four consecutive slots of a plane with ONE 16-byte access) | W3a [HS] (slot order)
  static constexpr int B_RHO = KC * STRIDE, B_MASK = B_RHO + KC, BUF = B_MASK + 4 * KC;
  static constexpr int L_FT = 2 * BUF, L_W3A = L_FT + 6 * WIN, L_TOTAL = L_W3A + HS;
  static constexpr size_t LDS_BYTES = size_t(L_TOTAL) * 4;
};

struct WgradArgs {
  OnfGeom geom;
  const float* params;
  int aug_feature;
  // 0: pass 1 was onf_fused.hip / onf_split.hip (factors in their slot orders, layouts P and Q);  1: pass 1 was onf_x32.hip:
  // a factor's slot IS the feature / hidden-unit index, ones feature = geom.fin, h1's ones unit = 101, rho row of dh1 = 100,
  // and dh2 is rebuilt WITHOUT W3a (G2 = sum_p rho_p [a2_p > 0] h1_p^T; the gather kernel applies W3a)
  int x32_order;
  // stored factors, back to back in this order (carve_wgrad):  h1 [P,112] | dh1 [P,112] | de [P,WIN] | record [P,12]
  const float* ws;
  long long P;
  float* partial;  // [grid][NTILES][256]
};

// float4 number c (0 .. F4_PER_SAMPLE) of a sample: workspace offset of its source and its column in the combined LDS
// row (col < 0: record words 4..11 -> rho / sign-bit arrays).  Branch-free selects over compile-time constants, because
// the lanes of a wave straddle the segment boundaries.
template <int NKT>
__device__ __forceinline__ void f4_source(long long P, long long p, int c, long long* src_off, int* col) {
  using W = WgLayout<NKT>;
  constexpr int h4 = HS / 4, w4 = W::WIN / 4;
  constexpr int t1 = h4, t2 = t1 + h4, t3 = t2 + w4;
  // (prefix = floats per sample stored before this array, row length, first c, LDS column)
  int prefix = HS, row = HS, start = 0, lcol = W::C_DH1;                                  // dh1
  if (c >= t1) { prefix = 0; row = HS; start = t1; lcol = W::C_H1; }                       // h1
  if (c >= t2) { prefix = 2 * HS; row = W::WIN; start = t2; lcol = W::C_DE; }              // de
  if (c >= t3) { prefix = 2 * HS + W::WIN; row = 12; start = t3; lcol = W::C_U; }          // record: u | rho | sign words
  *src_off = P * prefix + p * row + 4 * (c - start);
  *col = lcol + 4 * (c - start);
  if (c > t3) *col = -(c - t3);   // -1: rho quad, -2: the four sign words
}

// Four input features of one sample from its pose, scalar form of features2<true, false> (same operations in the same order,
// so the values are the ones pass 1 used).  Scalar on purpose: the pose has just been read from LDS, and the packed form let
// hipcc fold the splat of u_y into  v_pk_fma_f32 ... op_sel:[0,1,0]  on the freshly loaded register pair, whose LOW lane then
// sometimes saw the previous sample's u_y on MI355X ("A hazard hipcc does not pad").
__device__ __forceinline__ f32x4 features4_scalar(const f32x4& wx, const f32x4& wy, const f32x4& b, const f32x4& fr, const f32x4& qh,
                                                  const f32x4& isa, float ux, float uy, float th) {
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float arg = fmaf(wx[k], ux, fmaf(wy[k], uy, b[k]));   // encoding_layer: W_e u + b_e (onf_model.py:39)
    const float za = (th + b[k]) * fr[k];                  // (theta + b) * f (angle_encoder.py:16)
    arg = isa[k] != 0.0f ? za : arg;
    v[k] = sin_halfturns_hw(arg, qh[k]);
  }
  return v;
}

// the same for four POSITIONAL features (sin / cos of the encoding layer): no angle argument, no select -- the values of
// features4_scalar for isa == 0, bit for bit
__device__ __forceinline__ f32x4 features4_positional(const f32x4& wx, const f32x4& wy, const f32x4& b, const f32x4& qh, float ux, float uy) {
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = sin_halfturns_hw(fmaf(wx[k], ux, fmaf(wy[k], uy, b[k])), qh[k]);
  return v;
}

template <int NKT>
__global__ __launch_bounds__(WG_THREADS, 2) void onf_wgrad_kernel(const WgradArgs a) {
  using W = WgLayout<NKT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const long long n_chunks = (a.P + KC - 1) / KC;

  // operand pointers of this wave's tiles: row g of the chunk, column of the tile + lane; k-step s adds the immediate
  // 4 * s * STRIDE
  int pa[W::TPW], pb[W::TPW];  // 32-bit LDS float offsets
#pragma unroll
  for (int j = 0; j < W::TPW; ++j) {
    int T = wave + WG_WAVES * j;
    if (T >= W::NTILES) T = W::NTILES - 1;  // idle slot: recomputes the last tile, never written
    int ac, bc;
    if (T < 7 * NKT) { ac = W::C_DH1 + 16 * (T / NKT); bc = W::C_IN + 16 * (T % NKT); }
    else if (T < 7 * NKT + 49) { const int q = T - 7 * NKT; ac = W::C_DH2 + 16 * (q / 7); bc = W::C_H1 + 16 * (q % 7); }
    else { ac = W::C_DE + 16 * (T - 7 * NKT - 49); bc = W::C_U; }
    pa[j] = g * W::STRIDE + ac + i;
    pb[j] = g * W::STRIDE + bc + i;
  }
  f32x4 acc[W::TPW];
#pragma unroll
  for (int j = 0; j < W::TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // zero the unused columns 4..15 of the u tile once (both buffers)
  for (int k = tid; k < 2 * KC * 12; k += WG_THREADS)
    lds[(k / (KC * 12)) * W::BUF + ((k / 12) % KC) * W::STRIDE + W::C_U + 4 + (k % 12)] = 0.0f;
  // tables in SLOT order for the rebuilt factors: feature table of `in` (pads evaluate to sin(0) = 0, the pad feature
  // `aug_feature` to cos(0) = 1: the ones column, as in fill_lds) and W3a for dh2
  {
    const OnfGeom& g = a.geom;
    const float* P = a.params;
    for (int k = tid; k < 6 * W::WIN; k += WG_THREADS) lds[W::L_FT + k] = 0.0f;
    for (int k = tid; k < HS; k += WG_THREADS) lds[W::L_W3A + k] = 0.0f;
    __syncthreads();
    for (int f = tid; f < 32 * ((NKT + 1) / 2); f += WG_THREADS) {
      const int slot = slot_layout_p(f);
      if (slot >= W::WIN) continue;
      float wx = 0.f, wy = 0.f, b = 0.f, fr = 0.f, qh = 0.f, isa = 0.f;
      if (f < g.n_enc) {
        wx = P[g.off_we + 2 * f]; wy = P[g.off_we + 2 * f + 1];
        b = g.off_be >= 0 ? P[g.off_be + f] : 0.0f;
        qh = (g.n_enc > g.n_sin && f >= g.n_sin) ? NFOPP_Q_UNIT : 0.0f;
      } else if (f < g.fin) {
        const int k = f - g.n_enc;
        b = P[g.off_ang_b + k]; fr = P[g.off_ang_f + k];
        qh = k >= g.ang_dim ? NFOPP_Q_UNIT : 0.0f;
        isa = 1.0f;
      } else if (f == a.aug_feature) {
        qh = NFOPP_Q_UNIT;
      }
      float* e = lds + W::L_FT + slot;
      e[0] = wx; e[W::WIN] = wy; e[2 * W::WIN] = b; e[3 * W::WIN] = fr; e[4 * W::WIN] = qh; e[5 * W::WIN] = isa;
    }
    for (int h = tid; h < NFOPP_HIDDEN; h += WG_THREADS) lds[W::L_W3A + hidden_slot(h, false)] = P[g.off_w3 + h];
  }

  // Per-thread staging descriptors, computed ONCE: float4 number tid + k * WG_THREADS of a chunk always comes from the same
  // array / column of sample q_k and goes to the same LDS place; from chunk to chunk only the sample index moves, by a
  // fixed stride per array.  (Recomputing them per chunk -- 64-bit offsets, divisions by 115 -- was ~360 vector instructions
  // per wave and chunk, and the fp32 MFMA shares the vector ALU: a quarter of the kernel.)
  f32x4 stage[W::F4_PER_THREAD];
  const float* src[W::F4_PER_THREAD];   // source of this thread's k-th float4 for the NEXT prefetch
  int src_step[W::F4_PER_THREAD];       // floats per chunk stride (KC * gridDim.x samples of that array's row length)
  int dst[W::F4_PER_THREAD];            // LDS float offset inside a buffer; -1: nothing to stage
  int qk[W::F4_PER_THREAD];             // sample of the chunk (for the ragged last chunk)
#pragma unroll
  for (int k = 0; k < W::F4_PER_THREAD; ++k) {
    const int idx = tid + k * WG_THREADS;
    src[k] = a.ws; src_step[k] = 0; dst[k] = -1; qk[k] = 0;
    if (idx < KC * W::F4_PER_SAMPLE) {
      const int q = idx / W::F4_PER_SAMPLE, c = idx - q * W::F4_PER_SAMPLE;
      long long off; int col;
      f4_source<NKT>(a.P, (long long)blockIdx.x * KC + q, c, &off, &col);
      long long off1; int col1;
      f4_source<NKT>(a.P, (long long)blockIdx.x * KC + q + 1, c, &off1, &col1);
      src[k] = a.ws + off;
      src_step[k] = (int)(off1 - off) * KC * (int)gridDim.x;   // row length of the array times samples per step
      qk[k] = q;
      dst[k] = col >= 0 ? q * W::STRIDE + col : (col == -1 ? W::B_RHO + q : W::B_MASK + 4 * q);
      if (col == -1) dst[k] |= 1 << 30;                        // scalar store (rho)
    }
  }
  auto prefetch = [&](long long chunk) {
    const long long left = a.P - chunk * KC;   // samples from the start of this chunk to the end
#pragma unroll
    for (int k = 0; k < W::F4_PER_THREAD; ++k) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (dst[k] >= 0 && qk[k] < left) v = *reinterpret_cast<const f32x4*>(src[k]);
      stage[k] = v;
      src[k] += src_step[k];
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int k = 0; k < W::F4_PER_THREAD; ++k) {
      if (dst[k] < 0) continue;
      if (dst[k] & (1 << 30)) buf[dst[k] & ~(1 << 30)] = stage[k][0];
      else *reinterpret_cast<f32x4*>(buf + dst[k]) = stage[k];
    }
  };
  // factors pass 1 did not store: in = features(u) with pass 1's arithmetic (features2), dh2 = rho * W3a * [a2 > 0].
  // Item idx = tid + k * WG_THREADS (sample q, float4 c) is fixed per thread: its offsets are computed once.
  constexpr int RB_ITEMS = (KC * W::REBUILD_F4 + WG_THREADS - 1) / WG_THREADS;
  int rb_row[RB_ITEMS], rb_c[RB_ITEMS], rb_q[RB_ITEMS];
#pragma unroll
  for (int k = 0; k < RB_ITEMS; ++k) {
    const int idx = tid + k * WG_THREADS;
    const int q = idx / W::REBUILD_F4;
    rb_q[k] = idx < KC * W::REBUILD_F4 ? q : -1;
    rb_c[k] = idx - q * W::REBUILD_F4;
    rb_row[k] = q * W::STRIDE;
  }
  auto rebuild = [&](float* buf) {
#pragma unroll
    for (int k = 0; k < RB_ITEMS; ++k) {
      if (rb_q[k] < 0) continue;
      const int q = rb_q[k], c = rb_c[k];
      float* row = buf + rb_row[k];
      if (c < HS / 4) {                       // dh2 slots 4c .. 4c+3: tile c >> 2, lane group c & 3
        const unsigned bits = __float_as_uint(buf[W::B_MASK + 4 * q + (c & 3)]) >> (4 * (c >> 2));
        const float rho = buf[W::B_RHO + q];
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + W::L_W3A + 4 * c);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((bits >> r) & 1u) ? w[r] * rho : 0.0f;
        *reinterpret_cast<f32x4*>(row + W::C_DH2 + 4 * c) = v;
      } else {                                // in slots 4s .. 4s+3
        const int s4 = c - HS / 4;
        const float ux = row[W::C_U], uy = row[W::C_U + 1], th = row[W::C_U + 3];
        const float* e = lds + W::L_FT + 4 * s4;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(e), wy = *reinterpret_cast<const f32x4*>(e + W::WIN);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(e + 2 * W::WIN), fr = *reinterpret_cast<const f32x4*>(e + 3 * W::WIN);
        const f32x4 qh = *reinterpret_cast<const f32x4*>(e + 4 * W::WIN), isa = *reinterpret_cast<const f32x4*>(e + 5 * W::WIN);
        const f32x4 v = features4_scalar(wx, wy, bb, fr, qh, isa, ux, uy, th);
        *reinterpret_cast<f32x4*>(row + W::C_IN + 4 * s4) = v;
      }
    }
  };

  // Pipeline over this workgroup's chunks c0, c0 + grid, ...: while chunk k multiplies out of one LDS buffer, chunk k+1 is
  // committed (registers -> LDS) and rebuilt in the other, and chunk k+2 is in flight from HBM.  Two barriers per chunk,
  // and between any two of them every wave has MFMAs of chunk k to issue, so one wave's commit / rebuild arithmetic
  // overlaps the other waves' matrix work (the fp32 MFMA shares the vector ALU with its own wave's VALU work only).
  const long long c0 = blockIdx.x, step = gridDim.x;
  float* cur = lds;
  float* nxt = lds + W::BUF;
  if (c0 < n_chunks) {
    prefetch(c0);
    commit(cur);
    if (c0 + step < n_chunks) prefetch(c0 + step);
    __syncthreads();
    rebuild(cur);
    __syncthreads();
  }
  constexpr int GRP = 7;  // tiles whose operands are in flight together
  // operands of step (tile group, k-step) it+1 are read from LDS while the MFMAs of step it issue
  auto multiply = [&](int off, int j_lo, int j_hi) __attribute__((always_inline)) {
#ifndef NFOPP_ABL_NO_MFMA
    const int n_it = ((j_hi - j_lo + GRP - 1) / GRP) * (KC / 4);
    float av[2][GRP], bv[2][GRP];
    auto fetch = [&](int it, int buf) __attribute__((always_inline)) {
      const int j0 = j_lo + GRP * (it / (KC / 4)), s = it % (KC / 4);
#pragma unroll
      for (int j = j0; j < j0 + GRP && j < W::TPW; ++j) {
        av[buf][j - j0] = lds[off + pa[j] + 4 * s * W::STRIDE];
        bv[buf][j - j0] = lds[off + pb[j] + 4 * s * W::STRIDE];
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int it = 0; it < n_it; ++it) {
      if (it + 1 < n_it) fetch(it + 1, (it + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int j0 = j_lo + GRP * (it / (KC / 4));
#pragma unroll
      for (int j = j0; j < j0 + GRP && j < W::TPW; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[it & 1][j - j0], bv[it & 1][j - j0], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  };
  constexpr int SPLIT_AT = GRP * ((W::TPW + GRP - 1) / GRP / 2 + ((W::TPW + GRP - 1) / GRP) % 2);   // groups before barrier A
  for (long long chunk = c0; chunk < n_chunks; chunk += step) {
    const bool has_next = chunk + step < n_chunks;
    const int off = (int)(cur - lds);
    if (has_next) commit(nxt);                               // staged registers of chunk + step -> the free buffer
    if (chunk + 2 * step < n_chunks) prefetch(chunk + 2 * step);
    multiply(off, 0, SPLIT_AT);
    __syncthreads();                                          // A: `nxt` committed by everyone
#ifndef NFOPP_ABL_NO_REBUILD
    if (has_next) rebuild(nxt);
#endif
    multiply(off, SPLIT_AT, W::TPW);
    __syncthreads();                                          // B: `cur` consumed, `nxt` rebuilt
    float* t = cur; cur = nxt; nxt = t;
  }
#pragma unroll
  for (int j = 0; j < W::TPW; ++j) {
    const int T = wave + WG_WAVES * j;
    if (T < W::NTILES) {
      float* o = a.partial + ((long long)blockIdx.x * W::NTILES + T) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[64 * r] = acc[j][r];
    }
  }
}

// ---- pass 2 on the bf16 matrix pipe: three-level exact operand split, six partial products per fp32 multiply ---------
// (the split matrix path, csrc/onf_split.hip's arithmetic: x = hi + mid + lo, products hh, hm, mh, hl, lh, mm -- every
// dropped term is below 2^-24 of the product).  G1 = dh1^T in and G2 = dh2^T h1 carry 97 % of the flops, K = 32 samples per chunk.
// Factors in x32 order (pass 1 = csrc/onf_x32_impl.h, the default; round 4): v_mfma_f32_32x32x16_bf16, and G2 as
// mask^T (rho h1) -- the bare ReLU mask is exact in ONE level, so three products, nothing dropped; G3 = de^T u (4 useful
// columns) on v_mfma_f32_4x4x1_16b_f32.  Factors in the 16x16 kernels' slot order (matrix path 2): v_mfma_f32_16x16x32_bf16
// throughout, G3 likewise on the 4x4x1 form.
//
// LDS images.  Operands are stored SAMPLE-major as bf16, [32 samples][slots], one image per level, and read with the
// transposing ds_read_b64_tr_b16 (a 16-lane group covers 16 slots of 4 sample rows: lane i gets slot i of the four).
//   16x16x32: lane (i = l & 15, g = l >> 4): group g takes rows 16 (g >> 1) + 4 (g & 1) + {0..3} and + 8 -- the same for A
//             and B, so k pairs up -- which puts the 8 rows a 32-lane half reads at once next to each other;
//   32x32x16: lane (i = l & 31, g = l >> 5): the even rows of the half's eight samples in the first read, the odd ones in the second;
// with a row length of 8 * odd dwords both cover the 64 banks exactly once per 32-lane half (conflict-free).  Staging threads
// split each float4 once (11 vector instructions per pair) and store 8 bytes per level.
//
// Two phases per chunk, one barrier after each; each buffer is filled during the phase that multiplies out of the other:
//   phase A(k): multiply G1(k) from bufA, then (x32 order) write the mask plane of dh2(k)
//               | stage into bufB: h1(k) (x32 order: scaled by rho), de(k), u(k), (slot order: dh2(k) from the record);
//                 evaluate the input features of chunk k+1 into registers (record(k+1) was committed in phase B(k-1))
//   phase B(k): multiply G2(k), G3(k) from bufB | stage into bufA: dh1(k+1), the split features of chunk k+1; commit record(k+2)
// One register set per staged array: a chunk's rows are committed at the start of their phase and the same registers re-loaded
// with the next chunk's rows right behind (a chunk time ahead of use).  Waves 0..3 multiply, waves 4..7 stage (w and w + 4
// share a SIMD: see the kernel); the staging waves' instruction total bounds the kernel.
constexpr int KS = 32;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int odd8(int dwords) { return ((dwords / 8) & 1) ? dwords : dwords + 8; }

template <int NKT>
struct WsLayout {   // all offsets in dwords
  static constexpr int WIN = 16 * NKT;
  static constexpr int R_H = odd8(HS / 2), R_IN = odd8(WIN / 2);      // row lengths of the bf16 images
  static constexpr int P_H = KS * R_H, P_IN = KS * R_IN;              // one level of one operand
  static constexpr int RS_DE = (WIN + 16) % 32 == 16 ? WIN + 16 : WIN + 32;   // fp32 rows  de | u tile, = 16 mod 32
  static constexpr int A_DH1 = 0, A_IN = 3 * P_H, BUF_A = 3 * P_H + 3 * P_IN;
  static constexpr int B_DH2 = BUF_A, B_H1 = B_DH2 + 3 * P_H, B_DE = B_H1 + 3 * P_H, BUF_END = B_DE + KS * RS_DE;
  static constexpr int REC = BUF_END;                                  // two record areas [KS][12]
  static constexpr int L_FT = REC + 2 * KS * 12, L_W3A = L_FT + 6 * WIN, L_TOTAL = L_W3A + HS;
  static constexpr size_t LDS_BYTES = size_t(L_TOTAL) * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS image of the split weight-gradient pass");
  static constexpr int NTILES = 8 * NKT + 49;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// x0, x1 -> (bf16 level of x0 | bf16 level of x1 << 16), residuals left in x0, x1
__device__ __forceinline__ unsigned take_level(float& x0, float& x1) {
  const unsigned t0 = __float_as_uint(x0) & 0xffff0000u, t1 = __float_as_uint(x1) & 0xffff0000u;
  const unsigned w = __builtin_amdgcn_perm(t1, t0, 0x07060302);
  x0 = x0 - __uint_as_float(t0);
  x1 = x1 - __uint_as_float(t1);
  return w;
}
// four consecutive slots of one sample -> 8 bytes per level at dword offset `at` of the three images starting at `img`
__device__ __forceinline__ void store_split4(float* lds, int img, int plane, int at, f32x4 v) {
  u32x2 h, m, l;
#ifdef NFOPP_ABL2_NO_SPLIT   /* development ablation: the three stores without the arithmetic */
  h.x = __float_as_uint(v.x); h.y = __float_as_uint(v.y); m = h; l.x = __float_as_uint(v.z); l.y = __float_as_uint(v.w);
  *reinterpret_cast<u32x2*>(lds + img + at) = h;
  *reinterpret_cast<u32x2*>(lds + img + plane + at) = m;
  *reinterpret_cast<u32x2*>(lds + img + 2 * plane + at) = l;
  return;
#endif
  float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
  h.x = take_level(x0, x1); h.y = take_level(x2, x3);
  m.x = take_level(x0, x1); m.y = take_level(x2, x3);
  l.x = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302);   // third level: exact
  l.y = __builtin_amdgcn_perm(__float_as_uint(x3), __float_as_uint(x2), 0x07060302);
  *reinterpret_cast<u32x2*>(lds + img + at) = h;
  *reinterpret_cast<u32x2*>(lds + img + plane + at) = m;
  *reinterpret_cast<u32x2*>(lds + img + 2 * plane + at) = l;
}

// fragment of one level: two transposing reads (sample rows +0 and +8 of this lane group's set)
template <int R>
__device__ __forceinline__ s16x8 read_frag(const float* lane_base, int dword_off) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#ifdef NFOPP_ABL2_NO_FRAGREAD   /* development ablation */
  return s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lane_base + dword_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lane_base + dword_off + 8 * R));
  return s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// Wait states behind a group of MFMAs before its fragment registers are reloaded (make EXTRA=-DNFOPP_MFMA_GUARD).  A precaution of
// round 2 against run-to-run differences that turned out to come from a packed fma with op_sel on freshly loaded LDS values
// (DESIGN.md K5); hipcc's hazard recogniser handles MFMA operand reuse for builtins.
__device__ __forceinline__ void mfma_guard() {
#ifdef NFOPP_MFMA_GUARD   /* round 4: OFF by default -- 24 guards x 64 wait states per chunk were 13 % of the kernel once the
                             multiplying waves had become its critical path; the bit-for-bit repeat tests pass without them */
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");   // 64 wait states: four MFMA issue slots
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// Development build (make EXTRA=-DNFOPP_WG_PROFILE): waves 0 and 4 of workgroup 0 print the clock ticks they spent in their
// work and at the barriers of the two phases.
#ifdef NFOPP_WG_PROFILE
#define WG_TICK(SLOT)                                              \
  {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    wg_ticks[SLOT] += (float)(now_ - wg_t0);                       \
    wg_t0 = now_;                                                  \
  }
#else
#define WG_TICK(SLOT)
#endif

// End of a pipeline phase.  The scheduling barrier matters: without it hipcc hoists the NEXT phase's register arithmetic (the
// splitting of operands whose loads were issued a moment ago) above the barrier, and the s_waitcnt vmcnt(0) that goes with
// it exposes the full HBM latency once per chunk (0.6 ms of the 1.7 ms the kernel took).
__device__ __forceinline__ void phase_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// Two tiles that share the A operand: the two accumulation chains are independent and alternate on the matrix pipe (a chain
// alone waits for its own previous result every time).
__device__ __forceinline__ void mfma_split_pair(const s16x8 (&a)[3], const s16x8 (&b0)[3], const s16x8 (&b1)[3], f32x4& c0,
                                                f32x4& c1) {
#ifdef NFOPP_ABL2_NO_MFMA
  asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]));
  return;
#endif
#ifdef NFOPP_ABL2_HALF_MFMA   /* development ablation (timing only): half the matrix instructions (what 32x32x16 tiles would issue) */
  c0 = mfma_bf16(a[2], b0[0], c0); c0 = mfma_bf16(a[0], b0[2], c0); c0 = mfma_bf16(a[1], b0[1], c0);
  c1 = mfma_bf16(a[1], b1[0], c1); c1 = mfma_bf16(a[0], b1[1], c1); c1 = mfma_bf16(a[0], b1[0], c1);
  return;
#endif
  c0 = mfma_bf16(a[2], b0[0], c0); c1 = mfma_bf16(a[2], b1[0], c1);
  c0 = mfma_bf16(a[0], b0[2], c0); c1 = mfma_bf16(a[0], b1[2], c1);
  c0 = mfma_bf16(a[1], b0[1], c0); c1 = mfma_bf16(a[1], b1[1], c1);
  c0 = mfma_bf16(a[1], b0[0], c0); c1 = mfma_bf16(a[1], b1[0], c1);
  c0 = mfma_bf16(a[0], b0[1], c0); c1 = mfma_bf16(a[0], b1[1], c1);
  c0 = mfma_bf16(a[0], b0[0], c0); c1 = mfma_bf16(a[0], b1[0], c1);
}
// The same with an A operand that is EXACT in one bf16 level (a 0 / 1 mask): three products, nothing dropped
__device__ __forceinline__ void mfma_mask_pair(const s16x8& a, const s16x8 (&b0)[3], const s16x8 (&b1)[3], f32x4& c0, f32x4& c1) {
#ifdef NFOPP_ABL2_NO_MFMA
  asm volatile("" :: "v"(a), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]));
  return;
#endif
  c0 = mfma_bf16(a, b0[2], c0); c1 = mfma_bf16(a, b1[2], c1);
  c0 = mfma_bf16(a, b0[1], c0); c1 = mfma_bf16(a, b1[1], c1);
  c0 = mfma_bf16(a, b0[0], c0); c1 = mfma_bf16(a, b1[0], c1);
}
// XO: the factors are in x32 order (WgradArgs::x32_order, a.x32_order == XO)
template <int NKT, bool XO>
__global__ __launch_bounds__(WG_THREADS, 2) void onf_wgrad_split_kernel(const WgradArgs a) {
  using L = WsLayout<NKT>;
  constexpr int WIN = L::WIN, H4 = HS / 4, W4 = WIN / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long n_chunks = (a.P + KS - 1) / KS;
  const long long c0 = blockIdx.x, step = gridDim.x;

  // ---- tables (slot order) and the constant parts of the images -----------------------------------------------------
  {
    const OnfGeom& g = a.geom;
    const float* P = a.params;
    for (int k = tid; k < L::L_TOTAL; k += WG_THREADS) lds[k] = 0.0f;   // also: u-tile columns 4..15, image pad columns
    __syncthreads();
    for (int f = tid; f < 32 * ((NKT + 1) / 2); f += WG_THREADS) {
      const int slot = XO ? f : slot_layout_p(f);
      if (slot >= WIN) continue;
      float wx = 0.f, wy = 0.f, b = 0.f, fr = 0.f, qh = 0.f, isa = 0.f;
      if (f < g.n_enc) {
        wx = P[g.off_we + 2 * f]; wy = P[g.off_we + 2 * f + 1];
        b = g.off_be >= 0 ? P[g.off_be + f] : 0.0f;
        qh = (g.n_enc > g.n_sin && f >= g.n_sin) ? NFOPP_Q_UNIT : 0.0f;
      } else if (f < g.fin) {
        const int k = f - g.n_enc;
        b = P[g.off_ang_b + k]; fr = P[g.off_ang_f + k];
        qh = k >= g.ang_dim ? NFOPP_Q_UNIT : 0.0f;
        isa = 1.0f;
      } else if (f == a.aug_feature) {
        qh = NFOPP_Q_UNIT;
      }
      float* e = lds + L::L_FT + slot;
      e[0] = wx; e[WIN] = wy; e[2 * WIN] = b; e[3 * WIN] = fr; e[4 * WIN] = qh; e[5 * WIN] = isa;
    }
    // factor of the rebuilt dh2: W3a in slot order, or 1 (x32 order: rho * [a2 > 0] alone, see WgradArgs)
    for (int h = tid; h < NFOPP_HIDDEN; h += WG_THREADS) lds[L::L_W3A + (XO ? h : hidden_slot(h, false))] = XO ? 1.0f : P[g.off_w3 + h];
  }
  __syncthreads();   // the tables are read into registers below

  // ---- two kinds of waves ---------------------------------------------------------------------------------------------
  // Waves w and w + 4 share a SIMD.  Waves 0..3 only multiply, waves 4..7 only stage: the SIMD's matrix pipe and its vector
  // ALU each have one wave that never waits for the other kind of work.  (With every wave doing both, the partners ran the
  // same kind of work at the same time -- MFMA groups against MFMA groups, splitting against splitting -- and the kernel
  // took the SUM of its matrix and vector time: 1.7 ms.)  Both kinds pass the same barriers.
  const float* const ws_h1 = a.ws;
  const float* const ws_dh1 = a.ws + a.P * HS;
  const float* const ws_de = a.ws + a.P * 2 * HS;
  const float* const ws_rec = a.ws + a.P * (2 * HS + WIN);

  if (wave >= WG_WAVES / 2) {
    // =============================== producers: 256 threads stage the operands ===========================================
    constexpr int PT = WG_THREADS / 2;
    const int pt = tid - PT;
    // item = one float4 of one sample of the chunk; array with row4 float4 per sample: idx = pt + j * 256 -> (q, c).
    // Surplus threads repeat the last item (same data, same place: no branches in the staging code).
    constexpr int N_H = (KS * H4 + PT - 1) / PT, N_W = (KS * W4 + PT - 1) / PT;
    // ONE register set per array: a chunk's rows are committed at the start of their phase and the same registers are re-loaded
    // with the next chunk's rows right behind the commit -- a whole chunk time ahead of their use.  (Round 3 kept two sets and
    // issued the loads in front of the commit; with the feature registers below that no longer fits 256 registers.)
    f32x4 st_dh1[N_H], st_h1[N_H], st_de[N_W], st_rec;
    int hqc[N_H], wqc[N_W];   // sample q << 8 | float4 c
#pragma unroll
    for (int j = 0; j < N_H; ++j) {
      const int idx = pt + j * PT, it = idx < KS * H4 ? idx : KS * H4 - 1;
      hqc[j] = ((it / H4) << 8) | (it % H4);
    }
#pragma unroll
    for (int j = 0; j < N_W; ++j) {
      const int idx = pt + j * PT, it = idx < KS * W4 ? idx : KS * W4 - 1;
      wqc[j] = ((it / W4) << 8) | (it % W4);
    }
    auto unpack = [](int qc, int& q, int& c) __attribute__((always_inline)) {
      q = qc >> 8;
      c = qc & 255;
    };
    const int rit = pt < KS * 3 ? pt : KS * 3 - 1, rq = rit / 3, rc = rit % 3;   // record: 3 float4 per sample
    // HBM loads as raw buffer loads: one resource per array and chunk (base = the chunk's first row, size = the rows that
    // exist), so the per-item offset q * row + 16 c is a 32-bit chunk-invariant and rows past P read as zeros without a branch
    auto chunk_rsrc = [&](const float* base, long long chunk, int row_floats) __attribute__((always_inline)) {
#ifdef NFOPP_WG_REVERSE   /* development A/B: walk the samples from the end (what pass 1 wrote last is still in the caches) */
      const long long p0 = chunk < n_chunks ? (n_chunks - 1 - chunk) * KS : a.P;
#else
      const long long p0 = chunk * KS;
#endif
      long long rows = a.P - p0;
      rows = rows > KS ? KS : (rows < 0 ? 0 : rows);
      const float* ptr = base + (rows > 0 ? p0 : 0) * row_floats;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, (int)(rows * row_floats * 4), 0x00020000);
    };
    auto load_h = [&](const float* base, f32x4 (&st)[N_H], long long chunk) __attribute__((always_inline)) {
      const auto rsrc = chunk_rsrc(base, chunk, HS);
#pragma unroll
      for (int j = 0; j < N_H; ++j) {
        int q, c;
        unpack(hqc[j], q, c);
#ifdef NFOPP_ABL2_NO_LOAD   /* development ablation (timing only): no HBM traffic, the arithmetic and LDS work stay */
        st[j] = f32x4{(float)q, (float)c, 1.0f + (float)chunk, 2.0f};
        asm volatile("" : "+v"(st[j]));
#else
        st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, q * (HS * 4) + 16 * c, 0, 0));
#endif
      }
    };
    auto load_de = [&](f32x4 (&st)[N_W], long long chunk) __attribute__((always_inline)) {
      const auto rsrc = chunk_rsrc(ws_de, chunk, WIN);
#pragma unroll
      for (int j = 0; j < N_W; ++j) {
        int q, c;
        unpack(wqc[j], q, c);
#ifdef NFOPP_ABL2_NO_LOAD
        st[j] = f32x4{(float)q, (float)c, 1.0f + (float)chunk, 2.0f};
        asm volatile("" : "+v"(st[j]));
#else
        st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, q * (WIN * 4) + 16 * c, 0, 0));
#endif
      }
    };
    auto load_rec = [&](f32x4& st, long long chunk) __attribute__((always_inline)) {
      const auto rsrc = chunk_rsrc(ws_rec, chunk, 12);
#ifdef NFOPP_ABL2_NO_LOAD
      st = f32x4{0.25f, 0.5f, 1.0f, 0.1f * (float)(chunk & 7)};
      asm volatile("" : "+v"(st));
#else
      st = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, rq * 48 + 16 * rc, 0, 0));
#endif
    };
    auto commit_rec = [&](f32x4 st, int parity) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(lds + L::REC + parity * KS * 12 + rq * 12 + 4 * rc) = st;
    };
    // rho_rec: x32 order, h1 only -- the rows are scaled by the sample's rho (record word 4) before the split, because G2's other
    // factor is then the bare mask [a2 > 0], exact in ONE bf16 level: S = sum_p [a2_p > 0] (rho_p h1_p), three partial products
    // instead of six and one image plane of dh2 instead of three (round 4; the same sum as (rho [a2 > 0]) h1, rounded at a
    // different place)
    auto commit_h = [&](int img, const f32x4 (&st)[N_H], const float* rho_rec = nullptr) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < N_H; ++j) {
        int q, c;
        unpack(hqc[j], q, c);
        f32x4 v = st[j];
        if (rho_rec) {   // (element by element: a vector-typed multiply becomes v_pk_mul_f32 ... op_sel, which the build check bans)
          const float rho = rho_rec[q * 12 + 4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * rho;
        }
#ifdef NFOPP_ABL2_NO_SPLIT_H   /* development ablation (timing only): h1 / dh1 committed without the split arithmetic */
        const u32x2 raw{__float_as_uint(st[j].x), __float_as_uint(st[j].z)};
        *reinterpret_cast<u32x2*>(lds + img + q * L::R_H + 2 * c) = raw;
        *reinterpret_cast<u32x2*>(lds + img + L::P_H + q * L::R_H + 2 * c) = raw;
        *reinterpret_cast<u32x2*>(lds + img + 2 * L::P_H + q * L::R_H + 2 * c) = u32x2{__float_as_uint(st[j].y), __float_as_uint(st[j].w)};
#else
        store_split4(lds, img, L::P_H, q * L::R_H + 2 * c, v);
#endif
      }
    };
    // Rebuilt operands.  A thread's items keep their columns from chunk to chunk, so the table entries they need sit in
    // registers:  dh2: item j of the hidden-side list (sample q, float4 c) -> W3a[4c .. 4c+3];
    //             in:  thread t < T_I owns slot quad s4 = t % W4 for the samples q = t / W4 + SP * j (one 24-float entry).
    f32x4 w3a_reg[N_H];
#pragma unroll
    for (int j = 0; j < N_H; ++j) w3a_reg[j] = *reinterpret_cast<const f32x4*>(lds + L::L_W3A + 4 * (hqc[j] & 255));
    constexpr int SP = PT / W4, T_I = SP * W4, N_I = (KS + SP - 1) / SP;
    const int it_in = pt < T_I ? pt : pt - T_I;       // surplus threads repeat other owners' work
    // x32 order (slots = feature indices): the positional quads are dealt to the first threads, the quads that can hold angle /
    // ones / pad slots to the last ones, so that whole WAVES evaluate positional features only and skip the angle argument and
    // the select (4 of 14 instructions per feature; the evaluation is 40 % of the staging waves' work)
#ifdef NFOPP_POSITIONAL_WAVES
    constexpr int NPQ = XO ? (NKT >= 13 ? 50 : 25) : 0, NSQ = W4 - NPQ;   // encoding-layer features / 4 (200 or 100 of them)
    const int in_s4 = !XO ? it_in % W4 : (it_in < NPQ * SP ? it_in % NPQ : NPQ + (it_in - NPQ * SP) % NSQ);
    const int in_q0 = !XO ? it_in / W4 : (it_in < NPQ * SP ? it_in / NPQ : (it_in - NPQ * SP) / NSQ);
#else
    const int in_s4 = it_in % W4, in_q0 = it_in / W4;
#endif
    // OFF by default (make EXTRA=-DNFOPP_POSITIONAL_WAVES): same-box A/B at P = 2.54 M, twice: 1.738 ms with it, 1.680 ms without
    // -- the instructions saved are fewer than what the changed slot-to-thread map costs the split stores.
#ifdef NFOPP_POSITIONAL_WAVES
    const bool pos_only = XO && (wave - WG_WAVES / 2) * 64 + 63 < NPQ * SP;   // wave-uniform
#else
    const bool pos_only = false;
#endif
    f32x4 t_wx, t_wy, t_b, t_fr, t_qh, t_isa;
    {
      const float* e = lds + L::L_FT + 4 * in_s4;
      t_wx = *reinterpret_cast<const f32x4*>(e); t_wy = *reinterpret_cast<const f32x4*>(e + WIN);
      t_b = *reinterpret_cast<const f32x4*>(e + 2 * WIN); t_fr = *reinterpret_cast<const f32x4*>(e + 3 * WIN);
      t_qh = *reinterpret_cast<const f32x4*>(e + 4 * WIN); t_isa = *reinterpret_cast<const f32x4*>(e + 5 * WIN);
    }
#ifdef NFOPP_WG_PROFILE
    float wg_ticks[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned long long wg_t0 = __builtin_readcyclecounter();
#endif
    // The input features of chunk k+1 are EVALUATED in phase A(k) -- into registers, from the record committed one phase
    // earlier -- and only split and stored in phase B(k).  (Round 3 evaluated and stored them in phase B: in-kernel stamps showed
    // phase A bound by the multiplying waves' G1 (6.8 k cycles, the staging waves idle for 2.5 k of them) and phase B by this
    // evaluation (4.8 k of 6.0 k, the multiplying waves idle for 3 k).  Same arithmetic, same values: bit-identical results.)
    f32x4 feat[N_I];
    auto eval_in = [&](const float* rec) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < N_I; ++j) {
        int q = in_q0 + SP * j;   // in slots 4 s4 .. 4 s4 + 3 of sample q = features(u), pass 1's arithmetic
        q = q < KS ? q : KS - 1;
        const float ux = rec[q * 12], uy = rec[q * 12 + 1], th = rec[q * 12 + 3];
#ifdef NFOPP_ABL2_NO_IN   /* development ablation (timing only): no feature evaluation, the split and the stores stay */
        feat[j] = f32x4{ux + t_wx.x, uy + t_wx.y, th + t_wx.z, ux + t_wx.w};
#else
        feat[j] = pos_only ? features4_positional(t_wx, t_wy, t_b, t_qh, ux, uy)
                           : features4_scalar(t_wx, t_wy, t_b, t_fr, t_qh, t_isa, ux, uy, th);
#endif
        // pinned: without this LLVM sinks the (pure) evaluation to its use behind the phase barrier, i.e. back into phase B
        asm volatile("" : "+v"(feat[j]));
      }
    };
    auto store_in = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < N_I; ++j) {
        int q = in_q0 + SP * j;
        q = q < KS ? q : KS - 1;
        store_split4(lds, L::A_IN, L::P_IN, q * L::R_IN + 2 * in_s4, feat[j]);
      }
    };
    // phase A(k)'s staging (register set s = record area of chunk k), into bufB: h1, de, u, dh2 of chunk k; then the features of
    // chunk k+1 from record area s^1 (committed in phase B(k-1)) into registers.  Loads first: h1, de of chunk k+1 into the other
    // set and the record of chunk k+2 (committed in phase B(k)).
    auto stage_b = [&](auto set_c, long long chunk) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      const float* rec = lds + L::REC + S * KS * 12;
      commit_h(L::B_H1, st_h1, XO ? rec : nullptr);
      load_h(ws_h1, st_h1, chunk + step);
      WG_TICK(5)
#pragma unroll
      for (int j = 0; j < N_W; ++j) {
        int q, c;
        unpack(wqc[j], q, c);
        *reinterpret_cast<f32x4*>(lds + L::B_DE + q * L::RS_DE + 4 * c) = st_de[j];
      }
      load_de(st_de, chunk + step);
      load_rec(st_rec, chunk + 2 * step);
      { const int t = pt & (KS * 4 - 1); lds[L::B_DE + (t >> 2) * L::RS_DE + WIN + (t & 3)] = rec[(t >> 2) * 12 + (t & 3)]; }
      WG_TICK(6)
#pragma unroll
      for (int j = 0; j < N_H; ++j) {   // dh2 slots 4c .. 4c+3 = rho * W3a * [a2 > 0]: tile c >> 2, lane group c & 3
        int q, c;
        unpack(hqc[j], q, c);
        if constexpr (XO) {   // the bare mask [a2 > 0] as bf16 1.0 / 0 (rho rides on h1, W3a is applied by the gather kernel)
#ifdef NFOPP_MASK_BY_STAGING_WAVES   /* A/B: round 4 first had the staging waves write it */
          const int word = (int)__float_as_uint(rec[q * 12 + 8 + (c & 3)]), sh = 4 * (c >> 2);
          const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(word, sh, 1), m1 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 1, 1);
          const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 3, 1);
          const unsigned k01 = __builtin_amdgcn_perm(m1, m0, 0x07060302), k23 = __builtin_amdgcn_perm(m3, m2, 0x07060302);
          *reinterpret_cast<u32x2*>(lds + L::B_DH2 + q * L::R_H + 2 * c) = u32x2{k01 & 0x3f803f80u, k23 & 0x3f803f80u};
#endif
          continue;   // (default: the MULTIPLYING waves write the mask plane behind G1, in the time they would wait at the barrier)
        }
        const unsigned bits = __float_as_uint(rec[q * 12 + 8 + (c & 3)]) >> (4 * (c >> 2));
        const float rho = rec[q * 12 + 4];
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((bits >> r) & 1u) ? w3a_reg[j][r] * rho : 0.0f;
        store_split4(lds, L::B_DH2, L::P_H, q * L::R_H + 2 * c, v);
      }
      WG_TICK(7)
      eval_in(lds + L::REC + (S ^ 1) * KS * 12);
    };
    // phase B(k)'s staging, into bufA: dh1 (set s) and the split features of chunk k+1; the record of chunk k+2 -> area s (last read
    // in phase A(k)); first the load of dh1 of chunk k+2 into the other set
    auto stage_a = [&](auto set_c, long long chunk) __attribute__((always_inline)) {
      constexpr int S = decltype(set_c)::value;
      commit_rec(st_rec, S);
      commit_h(L::A_DH1, st_dh1);
      load_h(ws_dh1, st_dh1, chunk + 2 * step);
      WG_TICK(9)
      store_in();
    };
    // prologue: record(c0) -> area 0, record(c0 + step) -> area 1, bufA(c0) in place; h1 / de (c0) and dh1(c0 + step) in set 0
    if (c0 < n_chunks) {
      f32x4 r0, r1;
      load_rec(r0, c0);
      load_h(ws_dh1, st_dh1, c0);
      load_rec(r1, c0 + step);                  // past P: zeros
      load_h(ws_h1, st_h1, c0);
      load_de(st_de, c0);
      commit_rec(r0, 0);
      commit_rec(r1, 1);
      phase_barrier();
      commit_h(L::A_DH1, st_dh1);
      load_h(ws_dh1, st_dh1, c0 + step);
      eval_in(lds + L::REC);
      store_in();
      phase_barrier();
    }
    auto one_chunk = [&](auto set_c, long long chunk) __attribute__((always_inline)) {
      stage_b(set_c, chunk);                    // phase A
      WG_TICK(0)
      phase_barrier();
      WG_TICK(1)
      stage_a(set_c, chunk);                    // phase B
      WG_TICK(2)
      phase_barrier();
      WG_TICK(3)
    };
    for (long long chunk = c0; chunk < n_chunks; chunk += 2 * step) {
      one_chunk(std::integral_constant<int, 0>{}, chunk);
      if (chunk + step < n_chunks) one_chunk(std::integral_constant<int, 1>{}, chunk + step);
    }
#ifdef NFOPP_WG_PROFILE
    if (blockIdx.x == 0 && tid == WG_THREADS / 2)
      printf("producer wave 4: phase A: loads %.0f, h1 %.0f, de + u %.0f, dh2 %.0f, features %.0f | barrier %.0f | phase B: load %.0f, "
             "rec + dh1 %.0f, in split %.0f | barrier %.0f ticks\n", wg_ticks[4], wg_ticks[5], wg_ticks[6], wg_ticks[7], wg_ticks[0],
             wg_ticks[1], wg_ticks[8], wg_ticks[9], wg_ticks[2], wg_ticks[3]);
#endif
    return;
  }

#ifndef NFOPP_WGRAD_16X16
  if constexpr (XO) {
    // ================== consumers, x32 order (round 4): waves 0..3 multiply on v_mfma_f32_32x32x16_bf16 ==================
    // The 16x16x32 form below issues 210 bf16 matrix instructions per wave and chunk, each holding the SIMD's vector issue port for 8
    // of its 16 cycles -- the port this wave shares with its staging partner, whose instruction total bounds the kernel (timing
    // ablation with half the matrix instructions: -8 %).  32x32x16 tiles need half as many issues for the same products.
    //   G1 = dh1^T in: 4 x 7 tiles of 32 x 32 (rows 112..127 are whatever lies behind the 112 slots of an image row: garbage that
    //        only reaches result rows that are never stored); wave w owns row tile w, all 7 column tiles, K = 2 steps of 16 samples.
    //   G2 = mask^T (rho h1): 4 x 4 tiles, wave w owns row tile w; the mask is exact in one level: 3 products per step.
    // Fragments by ds_read_b64_tr_b16 from the SAME sample-major images: lane (i = lane & 31, g = lane >> 5) holds slot 32 T + i for
    // the samples 16 ks + 8 g + {0..7}: two reads of four sample rows; a 16-lane group covers 16 slots (8 dwords) of 4 rows.
    // The first read takes the EVEN rows of the eight, the second the odd ones (A and B alike, so k pairs up): with rows of
    // 8 * odd dwords the eight (row, column half) pieces of a 32-lane half -- at 2 q R + 8 h dwords, q = 0..3, h = 0, 1 -- start at
    // eight different multiples of 8 banks (2 R / 8 = 2 or 6 mod 8): conflict-free; four CONSECUTIVE rows overlap pairwise.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    const int rowl = 8 * (lane >> 5) + 2 * qq, coll = 8 * ((lane >> 4) & 1) + 2 * pp;
    static_assert((L::R_H / 8) % 2 == 1 && (L::R_IN / 8) % 2 == 1 && L::R_H % 8 == 0 && L::R_IN % 8 == 0, "bank pattern of the 32-wide transposing reads");
    const float* const fA_dh1 = lds + L::A_DH1 + rowl * L::R_H + coll + 16 * wave;
    const float* const fA_in = lds + L::A_IN + rowl * L::R_IN + coll;
    const float* const fB_msk = lds + L::B_DH2 + rowl * L::R_H + coll + 16 * wave;
    const float* const fB_h1 = lds + L::B_H1 + rowl * L::R_H + coll;
    auto frag32 = [](const float* base, int dword_off, auto r_c) __attribute__((always_inline)) {
      constexpr int R = decltype(r_c)::value;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + dword_off));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + dword_off + R));
      return s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    };
    auto mm = [](const s16x8& x, const s16x8& y, f32x16 c) __attribute__((always_inline)) {
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };
    using ic_rh = std::integral_constant<int, L::R_H>;
    using ic_rin = std::integral_constant<int, L::R_IN>;
    constexpr int NCT = (WIN + 31) / 32;   // 32-column tiles of G1 (input slots)
    f32x16 acc1[NCT], acc2[4];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[t][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
    f32x4 acc3e = f32x4{0.f, 0.f, 0.f, 0.f}, acc3o = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* const g3_a = lds + L::B_DE + 64 * wave + lane;
    const float* const g3_b = lds + L::B_DE + WIN + (lane & 3);
    const bool g3_mine = 64 * wave < WIN;
    if (c0 < n_chunks) {
      phase_barrier();
      phase_barrier();
    }
#ifdef NFOPP_WG_PROFILE
    float wg_ticks[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long wg_t0 = __builtin_readcyclecounter();
#endif
    int par = 0;   // record area of the chunk being multiplied (the staging waves' register-set parity)
    for (long long chunk = c0; chunk < n_chunks; chunk += step) {
      // ---- phase A: G1 out of bufA.  Per k step of 16 samples: the A fragments once, the B fragments one column tile ahead
      {
        s16x8 af[3], bf[2][3];
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) bf[0][lv] = frag32(fA_in, lv * L::P_IN, ic_rin{});
        static_for<0, 2 * NCT>([&](auto sc) __attribute__((always_inline)) {
          constexpr int st = decltype(sc)::value, ks = st / NCT, ct = st % NCT;
          if constexpr (ct == 0) {
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) af[lv] = frag32(fA_dh1, lv * L::P_H + 16 * ks * L::R_H, ic_rh{});
          }
          if constexpr (st + 1 < 2 * NCT) {
            constexpr int ks1 = (st + 1) / NCT, ct1 = (st + 1) % NCT;
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) bf[(st + 1) & 1][lv] = frag32(fA_in, lv * L::P_IN + 16 * ct1 + 16 * ks1 * L::R_IN, ic_rin{});
          }
          __builtin_amdgcn_sched_barrier(0);
          const s16x8 (&b)[3] = bf[st & 1];
#ifndef NFOPP_ABL2_NO_MFMA
          acc1[ct] = mm(af[2], b[0], acc1[ct]);
          acc1[ct] = mm(af[0], b[2], acc1[ct]);
          acc1[ct] = mm(af[1], b[1], acc1[ct]);
          acc1[ct] = mm(af[1], b[0], acc1[ct]);
          acc1[ct] = mm(af[0], b[1], acc1[ct]);
          acc1[ct] = mm(af[0], b[0], acc1[ct]);
#endif
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      {   // the mask plane of dh2 for THIS chunk (see the 16x16x32 form below)
        const float* rec = lds + L::REC + par * KS * 12;
#pragma unroll
        for (int j = 0; j < (KS * H4 + WG_THREADS / 2 - 1) / (WG_THREADS / 2); ++j) {
          const int idx = tid + j * (WG_THREADS / 2), it = idx < KS * H4 ? idx : KS * H4 - 1;
          const int q = it / H4, c = it - q * H4;
          const int word = (int)__float_as_uint(rec[q * 12 + 8 + (c & 3)]), sh = 4 * (c >> 2);
          const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(word, sh, 1), m1 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 1, 1);
          const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 3, 1);
          const unsigned k01 = __builtin_amdgcn_perm(m1, m0, 0x07060302), k23 = __builtin_amdgcn_perm(m3, m2, 0x07060302);
          *reinterpret_cast<u32x2*>(lds + L::B_DH2 + q * L::R_H + 2 * c) = u32x2{k01 & 0x3f803f80u, k23 & 0x3f803f80u};
        }
        par ^= 1;
      }
      WG_TICK(0)
      phase_barrier();
      WG_TICK(1)
      // ---- phase B: G2 and G3 out of bufB
      {
        s16x8 am[2], bf[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) am[ks] = frag32(fB_msk, 16 * ks * L::R_H, ic_rh{});
#pragma unroll
        for (int lv = 0; lv < 3; ++lv) bf[0][lv] = frag32(fB_h1, lv * L::P_H, ic_rh{});
        static_for<0, 8>([&](auto sc) __attribute__((always_inline)) {
          constexpr int st = decltype(sc)::value, ct = st >> 1, ks = st & 1;
          if constexpr (st + 1 < 8) {
            constexpr int ct1 = (st + 1) >> 1, ks1 = (st + 1) & 1;
#pragma unroll
            for (int lv = 0; lv < 3; ++lv) bf[(st + 1) & 1][lv] = frag32(fB_h1, lv * L::P_H + 16 * ct1 + 16 * ks1 * L::R_H, ic_rh{});
          }
          __builtin_amdgcn_sched_barrier(0);
          const s16x8 (&b)[3] = bf[st & 1];
          // G3's operands of four samples: read here, multiplied behind this step's matrix instructions (the reads' latency
          // hides behind them; G3 after G2 in one block left this phase bound by these waves)
          float ga[4], gb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { ga[j] = g3_a[(4 * st + j) * L::RS_DE]; gb[j] = g3_b[(4 * st + j) * L::RS_DE]; }
#ifndef NFOPP_ABL2_NO_MFMA
          acc2[ct] = mm(am[ks], b[2], acc2[ct]);
          acc2[ct] = mm(am[ks], b[1], acc2[ct]);
          acc2[ct] = mm(am[ks], b[0], acc2[ct]);
#endif
          if (g3_mine) {
            acc3e = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[0], gb[0], acc3e, 0, 0, 0);
            acc3o = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[1], gb[1], acc3o, 0, 0, 0);
            acc3e = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[2], gb[2], acc3e, 0, 0, 0);
            acc3o = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[3], gb[3], acc3o, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      WG_TICK(2)
      phase_barrier();
      WG_TICK(3)
    }
#ifdef NFOPP_WG_PROFILE
    if (blockIdx.x == 0 && tid == 0)
      printf("consumer wave 0 (32x32x16): G1 %.0f | barrier %.0f | G2+G3 %.0f | barrier %.0f ticks\n", wg_ticks[0], wg_ticks[1], wg_ticks[2],
             wg_ticks[3]);
#endif
    // ---- per-workgroup partial tiles, in the 16 x 16 tile order the reduce / gather kernels read: result (row, col) -> tile
    // (row / 16, col / 16), element 64 (row & 3) + 16 ((row & 15) >> 2) + (col & 15).  A lane of a 32 x 32 accumulator holds column
    // lane & 31 and the rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3), r = 0..15.
    float* const part = a.partial + (long long)blockIdx.x * L::NTILES * 256;
    const int cj = lane & 31, g = lane >> 5;
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + 8 * (r >> 2) + 4 * g + (r & 3), col = 32 * t + cj;
        if (row < HS && col < WIN) part[((row >> 4) * NKT + (col >> 4)) * 256 + 64 * (row & 3) + 16 * ((row & 15) >> 2) + (col & 15)] = acc1[t][r];
      }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + 8 * (r >> 2) + 4 * g + (r & 3), col = 32 * t + cj;
        if (row < HS && col < HS) part[(7 * NKT + (row >> 4) * 7 + (col >> 4)) * 256 + 64 * (row & 3) + 16 * ((row & 15) >> 2) + (col & 15)] = acc2[t][r];
      }
    {
      const f32x4 g3 = acc3e + acc3o;
      const int rb = 4 * wave + (lane >> 4);
      float* o = part + (7 * NKT + 49 + rb) * 256;
      if (rb < NKT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[16 * ((lane >> 2) & 3) + (lane & 3) + 64 * i] = g3[i];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rz = 4 * wave + t;
        if (rz < NKT && (lane & 15) >= 4) {
          float* z = part + (7 * NKT + 49 + rz) * 256 + lane;
#pragma unroll
          for (int r = 0; r < 4; ++r) z[64 * r] = 0.0f;
        }
      }
    }
    return;
  }
#endif
  // ================================= consumers: waves 0..3 multiply =======================================================
  // G1 (7 x NKT tiles, dh1^T in): wave w owns the column blocks w, w + 4, w + 8, w + 12, all 7 row blocks each (a column block
  //   past NKT is multiplied on whatever the image holds there and never stored: 28 tiles for every wave).
  // G2 (7 x 7, dh2^T h1): w2 -> column blocks {2, 3}, w3 -> {4, 5}, w0 -> {0, 6}, w1 -> {1, 6}; block 6 is computed by both w0
  //   and w1 and stored by rows (w0: row blocks 0..3, w1: 4..6).   G3 (NKT x 1, de^T u, fp32): row blocks w, w+4, w+8, w+12.
  const int grp = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, i16 = lane & 15;
  const int row0 = 16 * (grp >> 1) + 4 * (grp & 1) + qq;   // lane part of every transposing read: sample row, dwords 2p
  const int lane_h = row0 * L::R_H + 2 * pp, lane_in = row0 * L::R_IN + 2 * pp;
  constexpr int R_F = L::R_H, LS_F = L::P_H;   // images of h1 / dh1: three planes [KS][R_H]
  const int lane_f = lane_h;
  const int g2_cb0 = wave < 2 ? wave : 2 * wave - 2, g2_cb1 = wave < 2 ? 6 : 2 * wave - 1;
  // one base register per image, with this wave's first column block folded in: every read offset below is a compile-time
  // constant under 64 KB and goes into the instruction's offset field
  const float* const bA_dh1 = lds + L::A_DH1 + lane_f;
  const float* const bA_in = lds + L::A_IN + lane_in + 8 * wave;          // + 32 dwords per further column block
  const float* const bB_dh2 = lds + L::B_DH2 + lane_h;
  const float* const bB_h1a = lds + L::B_H1 + lane_f + 8 * g2_cb0;
  const float* const bB_h1b = lds + L::B_H1 + lane_f + 8 * g2_cb1;
#ifdef NFOPP_G3_16X16
  const float* const rowk = lds + L::B_DE + grp * L::RS_DE + i16;
#endif
  f32x4 acc1[4][7], acc2[2][7], acc3[4];
  // G3 = de^T u has 4 useful columns (u_x, u_y, 1, theta): on v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4 x 4 blocks, one
  // rank-1 update each -- a lane feeds de[sample][64 w + lane] and u[sample][lane & 3], every output is a useful one and an
  // instruction takes 8 pipe cycles; the 16x16x4 form spent 32 cycles on tiles whose 12 other columns are zero, and an fp32 MFMA
  // blocks the SIMD's vector issue for its whole duration (no co-execution), i.e. the staging partner too.  Wave w owns rows
  // 64 w .. 64 w + 63; two alternating accumulators (even / odd samples), added at the end in a fixed order.
  f32x4 acc3e = f32x4{0.f, 0.f, 0.f, 0.f}, acc3o = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int G3_RA = L::RS_DE, G3_RB = L::RS_DE;   // row stride of the  de | u tile  rows
  const float* const g3_a = lds + L::B_DE + 64 * wave + lane;
  const float* const g3_b = lds + L::B_DE + WIN + (lane & 3);
  const bool g3_mine = 64 * wave < WIN;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 7; ++r) acc1[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 7; ++r) acc2[c][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 7 row blocks against one pair of column blocks; the A fragments ping-pong between two register sets (no copies)
  // (image geometry as compile-time constants: row length and level step of the A image and of the B image)
  // NA = levels of the A operand: 3, or 1 for an A that is exact in one bf16 level (the mask of G2 in x32 order)
  auto mul_pair = [&](const float* a_base, const float* b0_base, const float* b1_base, auto ra_c, auto la_c, auto rb_c, auto lb_c,
                      auto na_c, f32x4 (&c0)[7], f32x4 (&c1)[7]) __attribute__((always_inline)) {
    constexpr int RA = decltype(ra_c)::value, LA = decltype(la_c)::value, RB = decltype(rb_c)::value, LB = decltype(lb_c)::value;
    constexpr int NA = decltype(na_c)::value;
    s16x8 bf0[3], bf1[3], af[2][NA];
    mfma_guard();   // the fragment registers below were operands of the MFMAs just issued
#pragma unroll
    for (int lv = 0; lv < 3; ++lv) {
      bf0[lv] = read_frag<RB>(b0_base, lv * LB);
      bf1[lv] = read_frag<RB>(b1_base, lv * LB);
    }
#pragma unroll
    for (int lv = 0; lv < NA; ++lv) af[0][lv] = read_frag<RA>(a_base, lv * LA);
    static_for<0, 7>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      if constexpr (r + 1 < 7) {
        if constexpr (r > 0) mfma_guard();   // set (r + 1) & 1 was read by step r - 1's MFMAs, issued just before
#pragma unroll
        for (int lv = 0; lv < NA; ++lv) af[(r + 1) & 1][lv] = read_frag<RA>(a_base, lv * LA + 8 * (r + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NA == 3) mfma_split_pair(af[r & 1], bf0, bf1, c0[r], c1[r]);
      else mfma_mask_pair(af[r & 1][0], bf0, bf1, c0[r], c1[r]);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  if (c0 < n_chunks) {
    phase_barrier();
    phase_barrier();
  }
#ifdef NFOPP_WG_PROFILE
  float wg_ticks[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned long long wg_t0 = __builtin_readcyclecounter();
#endif
  int par = 0;   // record area of the chunk being multiplied (the staging waves' register-set parity)
  for (long long chunk = c0; chunk < n_chunks; chunk += step) {
    // phase A: G1 out of bufA
    using ic_rf = std::integral_constant<int, R_F>;
    using ic_lf = std::integral_constant<int, LS_F>;
    using ic_rin = std::integral_constant<int, L::R_IN>;
    using ic_pin = std::integral_constant<int, L::P_IN>;
    using ic_rh = std::integral_constant<int, L::R_H>;
    using ic_ph = std::integral_constant<int, L::P_H>;
    using ic3 = std::integral_constant<int, 3>;
    mul_pair(bA_dh1, bA_in, bA_in + 32, ic_rf{}, ic_lf{}, ic_rin{}, ic_pin{}, ic3{}, acc1[0], acc1[1]);
    mul_pair(bA_dh1, bA_in + 64, bA_in + 96, ic_rf{}, ic_lf{}, ic_rin{}, ic_pin{}, ic3{}, acc1[2], acc1[3]);
#ifndef NFOPP_MASK_BY_STAGING_WAVES
    if constexpr (XO) {
      // The mask plane of dh2 for THIS chunk (read by G2 in phase B), from the sign words of record area `par`: the staging waves
      // bound both phases (8.6 k cycles per chunk against 5.8 k here), these waves would wait ~1.5 k cycles at the barrier below,
      // and the item needs no registers to speak of (one LDS word in, one 8-byte store out).
      const float* rec = lds + L::REC + par * KS * 12;
#pragma unroll
      for (int j = 0; j < (KS * H4 + WG_THREADS / 2 - 1) / (WG_THREADS / 2); ++j) {
        const int idx = tid + j * (WG_THREADS / 2), it = idx < KS * H4 ? idx : KS * H4 - 1;   // surplus threads repeat the last item
        const int q = it / H4, c = it - q * H4;
        const int word = (int)__float_as_uint(rec[q * 12 + 8 + (c & 3)]), sh = 4 * (c >> 2);
        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(word, sh, 1), m1 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 1, 1);
        const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe(word, sh + 3, 1);
        const unsigned k01 = __builtin_amdgcn_perm(m1, m0, 0x07060302), k23 = __builtin_amdgcn_perm(m3, m2, 0x07060302);
        *reinterpret_cast<u32x2*>(lds + L::B_DH2 + q * L::R_H + 2 * c) = u32x2{k01 & 0x3f803f80u, k23 & 0x3f803f80u};
      }
      par ^= 1;
    }
#endif
    WG_TICK(0)
    phase_barrier();
    WG_TICK(1)
    // phase B: G2 and G3 out of bufB
    mul_pair(bB_dh2, bB_h1a, bB_h1b, ic_rh{}, ic_ph{}, ic_rf{}, ic_lf{}, std::integral_constant<int, XO ? 1 : 3>{}, acc2[0], acc2[1]);
#if !defined(NFOPP_ABL2_NO_G3) && !defined(NFOPP_G3_16X16)
    if (g3_mine) {
#pragma unroll
      for (int q = 0; q < KS; q += 2) {
        acc3e = __builtin_amdgcn_mfma_f32_4x4x1f32(g3_a[q * G3_RA], g3_b[q * G3_RB], acc3e, 0, 0, 0);
        acc3o = __builtin_amdgcn_mfma_f32_4x4x1f32(g3_a[(q + 1) * G3_RA], g3_b[(q + 1) * G3_RB], acc3o, 0, 0, 0);
      }
    }
#endif
#if !defined(NFOPP_ABL2_NO_G3) && defined(NFOPP_G3_16X16)
#pragma unroll
    for (int s = 0; s < KS / 4; ++s) {          // G3: four independent fp32 chains, k-step outer
      const float bu = rowk[4 * s * L::RS_DE + WIN];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rb = wave + 4 * j < NKT ? wave + 4 * j : NKT - 1;   // idle slot: repeats the last row block, never stored
        acc3[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(rowk[4 * s * L::RS_DE + 16 * rb], bu, acc3[j], 0, 0, 0);
      }
    }
#endif
    WG_TICK(2)
    phase_barrier();
    WG_TICK(3)
  }
#ifdef NFOPP_WG_PROFILE
  if (blockIdx.x == 0 && tid == 0)
    printf("consumer wave 0: G1 %.0f | barrier %.0f | G2+G3 %.0f | barrier %.0f ticks\n", wg_ticks[0], wg_ticks[1], wg_ticks[2],
           wg_ticks[3]);
#endif

  // ---- per-workgroup partial tiles (tile numbering and element order of the fp32 kernel) ---------------------------------
  auto put = [&](int T, const f32x4& v) __attribute__((always_inline)) {
    float* o = a.partial + ((long long)blockIdx.x * L::NTILES + T) * 256 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[64 * r] = v[r];
  };
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (wave + 4 * c < NKT)
#pragma unroll
      for (int r = 0; r < 7; ++r) put(r * NKT + wave + 4 * c, acc1[c][r]);
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    put(7 * NKT + r * 7 + g2_cb0, acc2[0][r]);
    const bool mine = wave >= 2 || (wave == 0 ? r < 4 : r >= 4);   // column block 6 is split by rows between waves 0 and 1
    if (mine) put(7 * NKT + r * 7 + g2_cb1, acc2[1][r]);
  }
#ifdef NFOPP_G3_16X16
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (wave + 4 * j < NKT) put(7 * NKT + 49 + wave + 4 * j, acc3[j]);
#else
  // G3 tiles in the 16 x 16 tile order the reduce / gather kernels read: row k, column c -> tile k / 16, element
  // 16 ((k % 16) / 4) + c + 64 (k % 4).  This lane holds rows 64 w + 4 (lane / 4) + i, column lane & 3; the columns 4..15 of
  // the tiles (the zero columns of the 16x16x4 form) are written as zeros by the lanes that own none of the others.
  {
    const f32x4 g3 = acc3e + acc3o;
    const int rb = 4 * wave + (lane >> 4);
    float* o = a.partial + ((long long)blockIdx.x * L::NTILES + 7 * NKT + 49 + rb) * 256;
    if (rb < NKT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[16 * ((lane >> 2) & 3) + (lane & 3) + 64 * i] = g3[i];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rz = 4 * wave + t;
      if (rz < NKT && (lane & 15) >= 4) {
        float* z = a.partial + ((long long)blockIdx.x * L::NTILES + 7 * NKT + 49 + rz) * 256 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[64 * r] = 0.0f;
      }
    }
  }
#endif
}

// reduced[e] = sum over workgroups of partial[wg][e], fixed order: 64 elements per block, 16 row groups per element (thread
// (e, r) adds rows r, r + 16, ... in ascending order, four loads in flight), then the 16 group sums in ascending r.  (One thread
// per element walked all 256 rows alone: 161 blocks on 256 CUs, 64 us of load latency per fit; this form: 11 us.)
constexpr int RED_GROUPS = 16;
__global__ __launch_bounds__(64 * RED_GROUPS) void onf_wgrad_reduce_kernel(const float* partial, float* reduced, int n_elems, int n_wg) {
  __shared__ float part[RED_GROUPS][64];
  const int el = threadIdx.x & 63, r = threadIdx.x >> 6, e = blockIdx.x * 64 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n_elems) {
    int w = r;
    for (; w + 3 * RED_GROUPS < n_wg; w += 4 * RED_GROUPS) {
      s0 += partial[(long long)w * n_elems + e];
      s1 += partial[(long long)(w + RED_GROUPS) * n_elems + e];
      s2 += partial[(long long)(w + 2 * RED_GROUPS) * n_elems + e];
      s3 += partial[(long long)(w + 3 * RED_GROUPS) * n_elems + e];
    }
    for (; w < n_wg; w += RED_GROUPS) s0 += partial[(long long)w * n_elems + e];
  }
  part[r][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (r == 0 && e < n_elems) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RED_GROUPS; ++k) s += part[k][el];
    reduced[e] = s;
  }
}

struct GatherArgs {
  OnfGeom geom;
  int nkt, aug_in_slot;
  int x32_order;              // WgradArgs::x32_order
  const float* params;
  const float* reduced;       // [NTILES][256]
  const float* g4;            // [112] dW3[:100] in h2 slot order (reduced per-wave partials of pass 1)
  const float* loss_partial;  // [1] summed loss
  float* grad;                // [n_params + 2]
  float count;
};

__device__ __forceinline__ float tile_elem(const float* reduced, int T, int rs, int cs) {
  return reduced[T * 256 + (rs & 3) * 64 + (rs >> 2) * 16 + cs];
}

__global__ __launch_bounds__(256) void onf_wgrad_gather_kernel(const GatherArgs a) {
  const OnfGeom& g = a.geom;
  const int NKT = a.nkt;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int T_G2 = 7 * NKT, T_G3 = 7 * NKT + 49;
  const bool xo = a.x32_order != 0;
  // slot of an input feature / of a hidden unit as a row or column of h1, dh1 (layout Q) / of h2, dh2 (layout P)
  auto in_slot = [&](int f) { return xo ? f : slot_layout_p(f); };
  auto h1_slot = [&](int h) { return xo ? h : hidden_slot(h, true); };
  auto h2_slot = [&](int h) { return xo ? h : hidden_slot(h, false); };
  const int ones_h1 = xo ? 101 : AUG_HIDDEN_SLOT, rho_row = xo ? 100 : AUG_HIDDEN_SLOT;
  auto g1 = [&](int rs, int cs) { return tile_elem(a.reduced, (rs / 16) * NKT + cs / 16, rs % 16, cs % 16); };
  auto g2 = [&](int rs, int cs) { return tile_elem(a.reduced, T_G2 + (rs / 16) * 7 + cs / 16, rs % 16, cs % 16); };
  auto g3 = [&](int s, int c) { return tile_elem(a.reduced, T_G3 + s / 16, s % 16, c); };
  if (o < g.n_params) {
    float v = 0.f;
    if (g.n_ang && o < g.off_ang_b + g.n_ang) {            // d/d angle bias = f_k * sum dz_k
      const int k = o - g.off_ang_b;
      v = a.params[g.off_ang_f + k] * g3(in_slot(g.n_enc + k), 2);
    } else if (g.n_ang && o < g.off_ang_f + g.n_ang) {     // d/d frequency = sum dz_k (theta + b_k)
      const int k = o - g.off_ang_f, s = in_slot(g.n_enc + k);
      v = g3(s, 3) + a.params[g.off_ang_b + k] * g3(s, 2);
    } else if (o < g.off_b1) {                             // W1[m][k]
      const int q = o - g.off_w1, m = q / g.fin, k = q - m * g.fin;
      v = g1(h1_slot(m), in_slot(k));
    } else if (o < g.off_w2) {                             // b1[m]
      v = g1(h1_slot(o - g.off_b1), a.aug_in_slot);
    } else if (o < g.off_b2) {                             // W2[m][k]
      const int q = o - g.off_w2, m = q / NFOPP_HIDDEN, k = q - m * NFOPP_HIDDEN;
      v = g2(h2_slot(m), h1_slot(k));
      if (xo) v *= a.params[g.off_w3 + m];
    } else if (o < g.off_w3) {                             // b2[m]
      const int m = o - g.off_b2;
      v = g2(h2_slot(m), ones_h1);
      if (xo) v *= a.params[g.off_w3 + m];
    } else if (o < g.off_b3) {                             // W3[j]
      const int j = o - g.off_w3;
      if (j < NFOPP_HIDDEN && !xo) {
        v = a.g4[hidden_slot(j, false)];
      } else if (j < NFOPP_HIDDEN) {
        // sum_p rho_p relu(a2_p)[j] = sum_p rho_p [a2_p[j] > 0] (W2[j] . h1_p + b2[j]) = W2[j] . S[j] + b2[j] S[j][ones]:
        // h2 never has to be summed where it is formed (fixed order: k ascending, the bias last)
        for (int k = 0; k < NFOPP_HIDDEN; ++k) v = fmaf(a.params[g.off_w2 + j * NFOPP_HIDDEN + k], g2(j, k), v);
        v = fmaf(a.params[g.off_b2 + j], g2(j, ones_h1), v);
      } else {
        v = g1(rho_row, in_slot(j - NFOPP_HIDDEN));
      }
    } else if (o == g.off_b3) {
      v = g1(rho_row, a.aug_in_slot);
    } else if (g.off_be < 0 || o < g.off_be) {             // We[f][c]
      const int q = o - g.off_we;
      v = g3(in_slot(q >> 1), q & 1);
    } else {                                               // be[f]
      v = g3(in_slot(o - g.off_be), 2);
    }
    a.grad[o] = v;
  }
  if (o == 0) {  // mean BCE loss: the per-wave partials were summed by onf_rows_reduce_kernel (fixed tree)
    a.grad[g.n_params] = a.loss_partial[0];
    a.grad[g.n_params + 1] = a.count;
  }
}

// smallest zero-weight pad feature that the input slot map covers (tiles < NKT): carries the "ones" column
static int find_aug_feature(int fin, int nkt) {
  for (int f = fin; f < 32 * ((nkt + 1) / 2); ++f)
    if (slot_layout_p(f) / 16 < nkt) return f;
  return -1;
}

struct WgradWs {  // float offsets into the workspace
  long long h1, dh1, de, rec, loss, loss_sum, g4_partial, g4, partial, reduced, total;
  int win, ntiles, grid_cap;
};

static WgradWs carve_wgrad(const OnfGeom& g, long long P, int nkt) {
  WgradWs w;
  const long long hrow = HS;
  w.win = 16 * nkt;
  w.ntiles = 8 * nkt + 49;
  w.grid_cap = onf_train_grid_upper_bound();
  long long o = 0;
  w.h1 = o; o += P * hrow;     // the kernel addresses these four back to back (f4_source): keep the order
  w.dh1 = o; o += P * hrow;
  w.de = o; o += P * w.win;
  w.rec = o; o += P * 12;
  w.loss = o; o += (long long)w.grid_cap * (WAVES > 8 ? WAVES : 8);   // pass 1 writes one row per wave (onf_x32.hip: always 8 per workgroup)
  w.loss_sum = o; o += 4;
  w.g4_partial = o; o += (long long)w.grid_cap * WAVES * HS;
  w.g4 = o; o += HS;
  w.partial = o; o += (long long)w.grid_cap * w.ntiles * 256;
  w.reduced = o; o += (long long)w.ntiles * 256;
  w.total = o;
  return w;
}

// sized for the longer of the two row lengths (x32 order reserves a pad position for the ones feature: (fin + 16) / 16 tiles),
// so the answer does not depend on the matrix path in force when the fit runs
size_t wgrad_workspace_bytes(const OnfGeom& g, long long P) { return (size_t)carve_wgrad(g, P, (g.fin + 16) / 16).total * sizeof(float); }

template <int NKT>
static int launch_wgrad(const WgradArgs& a, int grid, hipStream_t st) {
  using W = WgLayout<NKT>;
  static bool attr_set[MAX_DEVICES] = {};
  auto kern = onf_wgrad_kernel<NKT>;
  const int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), W::LDS_BYTES, attr_set);
  if (rc_attr != NFOPP_OK) return rc_attr;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG_THREADS), W::LDS_BYTES, st, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

template <int NKT, bool XO>
static int launch_wgrad_split_o(const WgradArgs& a, int grid, hipStream_t st) {
  using L = WsLayout<NKT>;
  static bool attr_set[MAX_DEVICES] = {};
  auto kern = onf_wgrad_split_kernel<NKT, XO>;
  const int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), L::LDS_BYTES, attr_set);
  if (rc_attr != NFOPP_OK) return rc_attr;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WG_THREADS), L::LDS_BYTES, st, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

template <int NKT>
static int launch_wgrad_split(const WgradArgs& a, int grid, hipStream_t st) {
  return a.x32_order ? launch_wgrad_split_o<NKT, true>(a, grid, st) : launch_wgrad_split_o<NKT, false>(a, grid, st);
}

// gradient of the mean BCE loss over `count` samples (inv_count = 1 / count) into grad[n_params + 2]
int onf_train_grad_mfma(const OnfGeom& g, const float* params, const float* samples, const float* labels, long long P,
                        float inv_count, float* grad, float* ws, hipStream_t st) {
  // pass 1 on the 32x32x16 kernel (matrix path 1, the feature dimensions it covers): factors in x32 order
  const bool xo = onf_use_x32(g);
  const int nkt = xo ? (g.fin + 16) / 16 : (g.fin + 15) / 16;
  const WgradWs w = carve_wgrad(g, P, nkt);
  const int aug = xo ? g.fin : find_aug_feature(g.fin, nkt);
  NFOPP_REQUIRE(aug >= 0, "no pad feature available for the ones column (fin = %d)", g.fin);
  OnfKernelArgs a = {};
  a.geom = g; a.params = params; a.points = samples; a.n_points = P; a.out4 = nullptr;
  a.labels = labels; a.inv_count = inv_count; a.aug_feature = aug;
  a.ws_h1 = ws + w.h1; a.ws_dh1 = ws + w.dh1; a.ws_de = ws + w.de; a.ws_u = ws + w.rec;
  a.loss_partial = ws + w.loss; a.g4_partial = ws + w.g4_partial;
  int grid_fwd = 0;
  int rc = NFOPP_OK;
#ifdef NFOPP_DEV_SKIP_PASS1   /* development timing: pass 2 alone on the factors of the first calls (env NFOPP_DEV_SKIP_PASS1) */
  static int dev_calls = 0;
  static int dev_grid = 0;
  if (getenv("NFOPP_DEV_SKIP_PASS1") && dev_calls++ >= 2) grid_fwd = dev_grid;
  else
#endif
  rc = xo ? launch_onf_x32_train_kernel(a, st, &grid_fwd) : launch_onf_train_kernel(a, st, &grid_fwd);
  if (rc) return rc;
#ifdef NFOPP_DEV_SKIP_PASS1
  dev_grid = grid_fwd;
#endif

  WgradArgs wa;
  wa.geom = g; wa.params = params; wa.aug_feature = aug; wa.x32_order = xo ? 1 : 0;
  wa.ws = ws + w.h1; wa.P = P; wa.partial = ws + w.partial;   // arrays back to back from w.h1 (see carve_wgrad)
  const bool split = onf_split_enabled();   // pass 2 follows pass 1's matrix path
  const int kc = split ? KS : KC;
  long long n_chunks = (P + kc - 1) / kc;
  int grid = (int)(n_chunks < w.grid_cap ? n_chunks : w.grid_cap);
  switch (nkt) {
    case 14: rc = split ? launch_wgrad_split<14>(wa, grid, st) : launch_wgrad<14>(wa, grid, st); break;
    case 13: rc = split ? launch_wgrad_split<13>(wa, grid, st) : launch_wgrad<13>(wa, grid, st); break;
    case 8: rc = split ? launch_wgrad_split<8>(wa, grid, st) : launch_wgrad<8>(wa, grid, st); break;
    case 7: rc = split ? launch_wgrad_split<7>(wa, grid, st) : launch_wgrad<7>(wa, grid, st); break;
    default: set_error("unsupported ONF feature dimension %d", g.fin); return NFOPP_ERR_ARG;
  }
  if (rc) return rc;
  const int n_elems = w.ntiles * 256;
  hipLaunchKernelGGL(onf_wgrad_reduce_kernel, dim3((n_elems + 63) / 64), dim3(64 * RED_GROUPS), 0, st, ws + w.partial,
                     ws + w.reduced, n_elems, grid);
  NFOPP_HIP(hipGetLastError());
  // loss and dW3[:100]: per-wave partials of pass 1
  hipLaunchKernelGGL(onf_rows_reduce_kernel, dim3(1), dim3(256), 0, st, ws + w.loss, ws + w.loss_sum, 1, grid_fwd * (xo ? 8 : WAVES));
  NFOPP_HIP(hipGetLastError());
  if (!xo) {   // (x32 order: dW3[:100] comes out of G2 in the gather kernel)
    hipLaunchKernelGGL(onf_rows_reduce_kernel, dim3(HS), dim3(256), 0, st, ws + w.g4_partial, ws + w.g4, HS, grid_fwd * WAVES);
    NFOPP_HIP(hipGetLastError());
  }
  GatherArgs ga;
  ga.x32_order = xo ? 1 : 0;
  ga.geom = g; ga.nkt = nkt; ga.aug_in_slot = xo ? aug : slot_layout_p(aug);   ga.params = params; ga.reduced = ws + w.reduced; ga.g4 = ws + w.g4; ga.loss_partial = ws + w.loss_sum; ga.grad = grad; ga.count = (float)P;
  hipLaunchKernelGGL(onf_wgrad_gather_kernel, dim3((g.n_params + 255) / 256), dim3(256), 0, st, ga);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

}  // namespace nfopp
