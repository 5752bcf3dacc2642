// ONF fitting step at scale: weight gradients as fp32-MFMA GEMMs over the sample axis.
//
// Replaces (reference, PyTorch-CPU + autograd): `loss.backward()` of `_optimize_collision_model`
// nfop/nerf_opt_planner.py:83-89 when the number of samples is large (batched planning with a shared field:
// P = B * (N + 109) samples per step, BASELINE config 5).  Small P keeps the per-sample path of csrc/onf_train.hip.
//
// Pipeline (all reductions in a fixed order -> bitwise reproducible, no float atomics):
//   1. onf_fwd_bwd_kernel<.., TRAIN> (csrc/onf_fused.hip): forward + backward per sample on the MFMA chain.  It writes
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
        const f32x2 ux = splat2(row[W::C_U]), uy = splat2(row[W::C_U + 1]), th = splat2(row[W::C_U + 3]);
        const float* e = lds + W::L_FT + 4 * s4;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(e), wy = *reinterpret_cast<const f32x4*>(e + W::WIN);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(e + 2 * W::WIN), fr = *reinterpret_cast<const f32x4*>(e + 3 * W::WIN);
        const f32x4 qh = *reinterpret_cast<const f32x4*>(e + 4 * W::WIN), isa = *reinterpret_cast<const f32x4*>(e + 5 * W::WIN);
        f32x4 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 o = features2<true, false>(f32x2{wx[2 * h], wx[2 * h + 1]}, f32x2{wy[2 * h], wy[2 * h + 1]},
                                                 f32x2{bb[2 * h], bb[2 * h + 1]}, f32x2{fr[2 * h], fr[2 * h + 1]},
                                                 f32x2{qh[2 * h], qh[2 * h + 1]}, f32x2{isa[2 * h], isa[2 * h + 1]}, ux, uy, th);
          v[2 * h] = o.x; v[2 * h + 1] = o.y;
        }
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

// reduced[e] = sum over workgroups of partial[wg][e], fixed order
__global__ __launch_bounds__(256) void onf_wgrad_reduce_kernel(const float* partial, float* reduced, int n_elems, int n_wg) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elems) return;
  float s = 0.f;
  for (int w = 0; w < n_wg; ++w) s += partial[(long long)w * n_elems + e];
  reduced[e] = s;
}

struct GatherArgs {
  OnfGeom geom;
  int nkt, aug_in_slot;
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
  if (o < g.n_params) {
    float v = 0.f;
    if (g.n_ang && o < g.off_ang_b + g.n_ang) {            // d/d angle bias = f_k * sum dz_k
      const int k = o - g.off_ang_b, s = slot_layout_p(g.n_enc + k);
      v = a.params[g.off_ang_f + k] * tile_elem(a.reduced, T_G3 + s / 16, s % 16, 2);
    } else if (g.n_ang && o < g.off_ang_f + g.n_ang) {     // d/d frequency = sum dz_k (theta + b_k)
      const int k = o - g.off_ang_f, s = slot_layout_p(g.n_enc + k);
      v = tile_elem(a.reduced, T_G3 + s / 16, s % 16, 3) +
          a.params[g.off_ang_b + k] * tile_elem(a.reduced, T_G3 + s / 16, s % 16, 2);
    } else if (o < g.off_b1) {                             // W1[m][k]
      const int q = o - g.off_w1, m = q / g.fin, k = q - m * g.fin;
      const int rs = hidden_slot(m, true), cs = slot_layout_p(k);
      v = tile_elem(a.reduced, (rs / 16) * NKT + cs / 16, rs % 16, cs % 16);
    } else if (o < g.off_w2) {                             // b1[m]
      const int rs = hidden_slot(o - g.off_b1, true), cs = a.aug_in_slot;
      v = tile_elem(a.reduced, (rs / 16) * NKT + cs / 16, rs % 16, cs % 16);
    } else if (o < g.off_b2) {                             // W2[m][k]
      const int q = o - g.off_w2, m = q / NFOPP_HIDDEN, k = q - m * NFOPP_HIDDEN;
      const int rs = hidden_slot(m, false), cs = hidden_slot(k, true);
      v = tile_elem(a.reduced, T_G2 + (rs / 16) * 7 + cs / 16, rs % 16, cs % 16);
    } else if (o < g.off_w3) {                             // b2[m]
      const int rs = hidden_slot(o - g.off_b2, false), cs = AUG_HIDDEN_SLOT;
      v = tile_elem(a.reduced, T_G2 + (rs / 16) * 7 + cs / 16, rs % 16, cs % 16);
    } else if (o < g.off_b3) {                             // W3[j]
      const int j = o - g.off_w3;
      if (j < NFOPP_HIDDEN) {
        v = a.g4[hidden_slot(j, false)];
      } else {
        const int cs = slot_layout_p(j - NFOPP_HIDDEN);
        v = tile_elem(a.reduced, 6 * NKT + cs / 16, 1, cs % 16);
      }
    } else if (o == g.off_b3) {
      v = tile_elem(a.reduced, 6 * NKT + a.aug_in_slot / 16, 1, a.aug_in_slot % 16);
    } else if (g.off_be < 0 || o < g.off_be) {             // We[f][c]
      const int q = o - g.off_we, f = q >> 1, s = slot_layout_p(f);
      v = tile_elem(a.reduced, T_G3 + s / 16, s % 16, q & 1);
    } else {                                               // be[f]
      const int s = slot_layout_p(o - g.off_be);
      v = tile_elem(a.reduced, T_G3 + s / 16, s % 16, 2);
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

static WgradWs carve_wgrad(const OnfGeom& g, long long P) {
  WgradWs w;
  const int nkt = (g.fin + 15) / 16;
  w.win = 16 * nkt;
  w.ntiles = 8 * nkt + 49;
  w.grid_cap = onf_train_grid_upper_bound();
  long long o = 0;
  w.h1 = o; o += P * HS;       // the kernel addresses these four back to back (f4_source): keep the order
  w.dh1 = o; o += P * HS;
  w.de = o; o += P * w.win;
  w.rec = o; o += P * 12;
  w.loss = o; o += (long long)w.grid_cap * 8;
  w.loss_sum = o; o += 4;
  w.g4_partial = o; o += (long long)w.grid_cap * 8 * HS;
  w.g4 = o; o += HS;
  w.partial = o; o += (long long)w.grid_cap * w.ntiles * 256;
  w.reduced = o; o += (long long)w.ntiles * 256;
  w.total = o;
  return w;
}

size_t wgrad_workspace_bytes(const OnfGeom& g, long long P) { return (size_t)carve_wgrad(g, P).total * sizeof(float); }

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

// gradient of the mean BCE loss over `count` samples (inv_count = 1 / count) into grad[n_params + 2]
int onf_train_grad_mfma(const OnfGeom& g, const float* params, const float* samples, const float* labels, long long P,
                        float inv_count, float* grad, float* ws, hipStream_t st) {
  const WgradWs w = carve_wgrad(g, P);
  const int nkt = (g.fin + 15) / 16;
  const int aug = find_aug_feature(g.fin, nkt);
  NFOPP_REQUIRE(aug >= 0, "no pad feature available for the ones column (fin = %d)", g.fin);
  OnfKernelArgs a = {};
  a.geom = g; a.params = params; a.points = samples; a.n_points = P; a.out4 = nullptr;
  a.labels = labels; a.inv_count = inv_count; a.aug_feature = aug;
  a.ws_h1 = ws + w.h1; a.ws_dh1 = ws + w.dh1; a.ws_de = ws + w.de; a.ws_u = ws + w.rec;
  a.loss_partial = ws + w.loss; a.g4_partial = ws + w.g4_partial;
  int grid_fwd = 0;
  int rc = launch_onf_train_kernel(a, st, &grid_fwd);
  if (rc) return rc;

  WgradArgs wa;
  wa.geom = g; wa.params = params; wa.aug_feature = aug;
  wa.ws = ws + w.h1; wa.P = P; wa.partial = ws + w.partial;   // arrays back to back from w.h1 (see carve_wgrad)
  long long n_chunks = (P + KC - 1) / KC;
  int grid = (int)(n_chunks < w.grid_cap ? n_chunks : w.grid_cap);
  switch (nkt) {
    case 14: rc = launch_wgrad<14>(wa, grid, st); break;
    case 13: rc = launch_wgrad<13>(wa, grid, st); break;
    case 8: rc = launch_wgrad<8>(wa, grid, st); break;
    case 7: rc = launch_wgrad<7>(wa, grid, st); break;
    default: set_error("unsupported ONF feature dimension %d", g.fin); return NFOPP_ERR_ARG;
  }
  if (rc) return rc;
  const int n_elems = w.ntiles * 256;
  hipLaunchKernelGGL(onf_wgrad_reduce_kernel, dim3((n_elems + 255) / 256), dim3(256), 0, st, ws + w.partial,
                     ws + w.reduced, n_elems, grid);
  NFOPP_HIP(hipGetLastError());
  // loss and dW3[:100]: per-wave partials of pass 1
  hipLaunchKernelGGL(onf_rows_reduce_kernel, dim3(1), dim3(256), 0, st, ws + w.loss, ws + w.loss_sum, 1, grid_fwd * 8);
  NFOPP_HIP(hipGetLastError());
  hipLaunchKernelGGL(onf_rows_reduce_kernel, dim3(HS), dim3(256), 0, st, ws + w.g4_partial, ws + w.g4, HS, grid_fwd * 8);
  NFOPP_HIP(hipGetLastError());
  GatherArgs ga;
  ga.geom = g; ga.nkt = nkt; ga.aug_in_slot = slot_layout_p(aug);   ga.params = params; ga.reduced = ws + w.reduced; ga.g4 = ws + w.g4; ga.loss_partial = ws + w.loss_sum; ga.grad = grad; ga.count = (float)P;
  hipLaunchKernelGGL(onf_wgrad_gather_kernel, dim3((g.n_params + 255) / 256), dim3(256), 0, st, ga);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

}  // namespace nfopp
