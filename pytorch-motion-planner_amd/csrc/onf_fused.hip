// Fused ONF forward + input-gradient kernel for gfx950 (MI355X), fp32 MFMA.
//
// Replaces (reference, PyTorch-CPU): ONF.forward nfop/onf_model.py:33-50 + AngleEncoder.forward
// nfop/angle_encoder.py:15-18 + autograd w.r.t. the input, and -- in trajectory mode -- the sampling of the
// collision points nfop/constrained_nerf_opt_planner.py:78-81 (SE(2)) / nfop/nerf_opt_planner.py:113-117 (2-D).
//
// Design (DESIGN.md section "K1"):
//  * one persistent 512-thread workgroup per CU; the ONF weights are staged ONCE per workgroup into LDS as plain
//    row-major images W1[100][S1], W2[100][129] (strides = 1 mod 32) and stay there for fwd AND bwd;
//  * each wave owns NT tiles of 16 points.  Points sit on the MFMA N axis (lane & 15), features on M/K:
//      D[feat][pt] += W[feat][k] * X[k][pt]        v_mfma_f32_16x16x4_f32, A = weights (ds_read_b32), B = activations
//    The accumulator layout (row = 4*(lane>>4)+reg, col = lane&15) IS the B-operand layout of the next layer once
//    the k index is permuted, so activations never leave registers between the four GEMMs
//      L1: a1 = W1 in        L2: a2 = W2 relu(a1)       L2T: dh1 = W2^T dh2      L1T: din = W1^T dh1
//    The permutations (layouts P for input features and h2, Q for h1 -- tools/emulate_mfma_layout.py) make every
//    weight read of all four GEMMs bank-conflict-free from the SAME two LDS images;
//  * input features sin/cos(W_e u + b_e), sin/cos((theta+b)f) are computed just-in-time as the L1 B operand and
//    re-derived (shifted by a quadrant) in the L1T epilogue, so neither `in` nor `din` is ever stored.
#include "onf_layout.h"

namespace nfopp {

// MODE 0: forward + input gradient (planner step)   1: training pass (factor matrices for the weight gradients)
// MODE 2: forward only (logits: pool-candidate weights, `ONF.forward`)
template <int NKT, int NT, int MODE>
__global__ __launch_bounds__(THREADS, THREADS / 256) void onf_fwd_bwd_kernel(const OnfKernelArgs a) {
  constexpr bool TRAIN = MODE == 1;
  constexpr bool FWD_ONLY = MODE == 2;
  using L = Lds<NKT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds<NKT, TRAIN>(lds, a);
  __syncthreads();

  const OnfGeom& geo = a.geom;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, gi = i >> 2, ri = i & 3;
  const int colP = 16 * (g & 1) + 4 * (g >> 1), colQ = 8 * (g & 1) + 4 * (g >> 1);
  const int rowposP = 16 * (gi & 1) + 4 * (gi >> 1) + ri, rowposQ = 8 * (gi & 1) + 4 * (gi >> 1) + ri;
  // first input tile that can hold an angle feature (layout P puts feature f in tile pair f/32)
  const int first_angle_kt = geo.n_ang ? 2 * (geo.n_enc / 32) : NKT;

  const float* W1 = lds + L::W1;
  const float* W2 = lds + L::W2;
  constexpr int S1 = L::S1;
  constexpr int CH = WAVES * 16 * NT;
  const long long n_work = work_points(a);
  const long long n_chunks = (n_work + CH - 1) / CH;
  const float b3 = a.params[geo.off_b3];
  constexpr int WIN = 16 * NKT;   // row length of the input-side factor matrices (TRAIN)
  constexpr int WH = 16 * HT;     // row length of the hidden-side factor matrices
  float loss_acc = 0.f;
  f32x4 g4_acc[HT];   // TRAIN: running sum_p rho_p * relu(a2_p) of this lane's (hidden row, point column) cells
#pragma unroll
  for (int mt = 0; mt < HT; ++mt) g4_acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // ---------------------------------------------------------------- sample / load the wave's points
    float ux[NT], uy[NT], th[NT];
    long long pidx[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      float x, y, ang;
      pidx[tl] = load_point(a, n_work, chunk * CH + (wave * NT + tl) * 16 + i, g, x, y, ang);
      ux[tl] = (x - geo.mean) / geo.sigma;  // onf_model.py:38
      uy[tl] = (y - geo.mean) / geo.sigma;
      th[tl] = ang;
      if (TRAIN && g == 0 && pidx[tl] < a.n_points)
        *reinterpret_cast<f32x4*>(a.ws_u + pidx[tl] * 12) = f32x4{ux[tl], uy[tl], 1.0f, ang};
    }

    // ---------------------------------------------------------------- L1: a1 = W1 in + b1, features just-in-time
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

    // A-operand bases of L1 (rows in h1 layout Q): tiles 0..3, 4..5 and 6 get one lane base each so that every
    // (mt, r) offset below is a compile-time immediate of the ds_read
    const float* w1a = W1 + rowposQ * S1 + colP;
    const float* w1b = w1a + 64 * S1;
    const float* w1c = W1 + (96 + gi) * S1 + colP;
    const float* ftl = lds + L::ft(colP);   // lane part of every feature-table address (+ L::ft_rel(block or tile base))
    const float* isl = lds + L::ISA + colP;

    auto l1_tile = [&](auto ang_c, int kt) __attribute__((always_inline)) {
      constexpr bool ANG = decltype(ang_c)::value;
      constexpr bool ang_tile = ANG;
      const int off = base_p(kt);
      const float* fte = ftl + L::ft_rel(off);
      const float* pa = w1a + off;
      const float* pb = w1b + off;
      const float* pc = w1c + off;
      float fv[4][NT];
      if (NT == 2) {  // pair = the two point tiles of this wave
        const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
        f32x2 sk = {skip[0], skip[NT - 1]};
        f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
        if (ang_tile) isa4 = *reinterpret_cast<const f32x4*>(isl + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);      // wx wx wy wy
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);  // b b fr fr
          const f32x4 e2 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 8);  // qh qh w3b w3b
          const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
          const f32x2 qh = {e2.x, e2.y}, w3 = {e2.z, e2.w};
          const f32x2 v = features2<ANG, false>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
          sk = fma2(w3, v, sk);
          fv[r][0] = v.x; fv[r][NT - 1] = v.y;
        }
        skip[0] = sk.x; skip[NT - 1] = sk.y;
      } else {        // pair = two consecutive k-steps of the single tile
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const float* ea = fte + L::FTS * r;
          const float* eb = ea + L::FTS;
          const f32x2 ux2 = splat2(ux[0]), uy2 = splat2(uy[0]), th2 = splat2(th[0]);
          const f32x2 wx = {ea[0], eb[0]}, wy = {ea[2], eb[2]}, bb = {ea[4], eb[4]}, fr = {ea[6], eb[6]};
          const f32x2 qh = {ea[8], eb[8]}, isa = {isl[off + r], isl[off + r + 1]};
          const f32x2 v = features2<ANG, false>(wx, wy, bb, fr, qh, isa, ux2, uy2, th2);
          skip[0] = fmaf(ea[10], v.x, skip[0]);
          skip[0] = fmaf(eb[10], v.y, skip[0]);
          fv[r][0] = v.x; fv[r + 1][0] = v.y;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) {
          const float wa = mt < 4 ? pa[mt * 16 * S1 + r] : (mt < 6 ? pb[(mt - 4) * 16 * S1 + r] : pc[r]);
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) acc1[tl][mt] = mfma4(wa, fv[r][tl], acc1[tl][mt]);
        }
      }
    };
#pragma unroll 1
    for (int kt = 0; kt < first_angle_kt; ++kt) l1_tile(std::false_type{}, kt);
#pragma unroll 1
    for (int kt = first_angle_kt; kt < NKT; ++kt) l1_tile(std::true_type{}, kt);

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

    // ---------------------------------------------------------------- L2: a2 = W2 relu(a1) + b2
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
      // 25 k-steps ks -> (t, r) = (ks >> 2, ks & 3); weights of step ks+1 are fetched while step ks multiplies
      float wa[2][HT];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) wa[0][mt] = W2[rowoff2[mt] + colQ];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int t = ks >> 2, r = ks & 3;
        if (ks + 1 < KSTEPS) {
          const int tn = (ks + 1) >> 2, rn = (ks + 1) & 3;
          const int col = tn < 6 ? 16 * tn + rn + colQ : 96 + g;  // k index in h1 layout Q
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) wa[(ks + 1) & 1][mt] = W2[rowoff2[mt] + col];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs
        float hb[NT];
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) hb[tl] = relu1(acc1[tl][t][r]);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) acc2[tl][mt] = mfma4(wa[ks & 1][mt], hb[tl], acc2[tl][mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

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
      if (TRAIN) {
        // BCE-with-logits, mean over the job's samples (nerf:25,88): rho = (sigmoid(l) - y) / count
        const bool valid = pidx[tl] < a.n_points;
        const float y = valid ? a.labels[pidx[tl]] : 0.0f;
        const float l = logit[tl];
        const float lp = fmaxf(l, 0.0f) - l * y + log1pf(expf(-fabsf(l)));
        rho[tl] = valid ? (1.0f / (1.0f + expf(-l)) - y) * a.inv_count : 0.0f;
        if (valid && g == 0) loss_acc += lp * a.inv_count;
        // dW3[:100] += rho * relu(a2) (per-lane running sums, reduced over the wave's point lanes at the end); the
        // sign pattern of a2 is all pass 2 needs to rebuild dh2 = rho * W3a * [a2 > 0]; then acc2 <- dh2
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

    // ---------------------------------------------------------------- L2T: dh1 = (W2^T dh2) * [a1 > 0]
    {
      f32x4 accd[NT][HT];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) accd[tl][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      int coloff[HT];  // output rows = h1 layout Q -> column of W2
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) coloff[mt] = mt < 6 ? 16 * mt + rowposQ : 96 + gi;
      float wa[2][HT];
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) wa[0][mt] = W2[colP * S2 + coloff[mt]];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int t = ks >> 2, r = ks & 3;
        if (ks + 1 < KSTEPS) {
          const int tn = (ks + 1) >> 2, rn = (ks + 1) & 3;
          const int row = (tn < 6 ? base_p(tn) + rn + colP : 96 + g) * S2;  // k index in h2 layout P
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) wa[(ks + 1) & 1][mt] = W2[row + coloff[mt]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < HT; ++mt)
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) accd[tl][mt] = mfma4(wa[ks & 1][mt], acc2[tl][t][r], accd[tl][mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < HT; ++mt)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc1[tl][mt][r] = acc1[tl][mt][r] > 0.0f ? accd[tl][mt][r] : 0.0f;  // dh1
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

    // ---------------------------------------------------------------- L1T: din = W1^T dh1 + W3b, then chain through the
    // encodings: d/dx = sum_k din_k * d(feature_k)/dx  (feature derivative = next quadrant of the same argument)
    float gx[NT], gy[NT], gt[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) gx[tl] = gy[tl] = gt[tl] = 0.f;
    const int rowkQ = colQ * S1;  // k-step ks reads W1 row 16*t + r + colQ (h1 layout Q), (6,0) -> row 96 + g

    auto l1t_tile = [&](auto ang_c, int mt) __attribute__((always_inline)) {
      constexpr bool ANG = decltype(ang_c)::value;
      const int fbase = base_p(mt) + colP;   // features of D rows (g, 0..3)
      const int colA = base_p(mt) + rowposP; // A-operand column of W1 (output row i in layout P)
      const f32x4 w3b = *reinterpret_cast<const f32x4*>(lds + L::W3B + fbase);
      f32x4 acc[NT];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) acc[tl] = TRAIN ? w3b * rho[tl] : w3b;
      // weights of this output tile: 25 values, fetched in groups of GRP k-steps one group ahead
      constexpr int GRP = 5;
      float wa[2][GRP];
#pragma unroll
      for (int k = 0; k < GRP; ++k) wa[0][k] = W1[rowkQ + (16 * (k >> 2) + (k & 3)) * S1 + colA];
#pragma unroll
      for (int grp = 0; grp < KSTEPS / GRP; ++grp) {
        if (grp + 1 < KSTEPS / GRP) {
#pragma unroll
          for (int k = 0; k < GRP; ++k) {
            const int ks = (grp + 1) * GRP + k, tn = ks >> 2, rn = ks & 3;
            wa[(grp + 1) & 1][k] = tn < 6 ? W1[rowkQ + (16 * tn + rn) * S1 + colA] : W1[(96 + g) * S1 + colA];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
          const int ks = grp * GRP + k, t = ks >> 2, r = ks & 3;
#pragma unroll
          for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma4(wa[grp & 1][k], acc1[tl][t][r], acc[tl]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // chain through the encodings: d feature / d arg = sin(arg + (qh + 0.5) pi); rows (g, r) <-> slots (4 mt + r, g)
      constexpr bool ang_tile = ANG;
      const float* fte = lds + L::ft(fbase);
      if (NT == 2) {
        const f32x2 ux2 = {ux[0], ux[NT - 1]}, uy2 = {uy[0], uy[NT - 1]}, th2 = {th[0], th[NT - 1]};
        f32x2 gx2 = {gx[0], gx[NT - 1]}, gy2 = {gy[0], gy[NT - 1]}, gt2 = {gt[0], gt[NT - 1]};
        f32x4 isa4 = {0.f, 0.f, 0.f, 0.f};
        if (ang_tile) isa4 = *reinterpret_cast<const f32x4*>(lds + L::ISA + fbase);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(fte + L::FTS * r + 4);
          const f32x2 qh = *reinterpret_cast<const f32x2*>(fte + L::FTS * r + 8);
          const f32x2 wx = {e0.x, e0.y}, wy = {e0.z, e0.w}, bb = {e1.x, e1.y}, fr = {e1.z, e1.w};
          const f32x2 cof = features2<ANG, true>(wx, wy, bb, fr, qh, splat2(isa4[r]), ux2, uy2, th2);
          const f32x2 de = f32x2{acc[0][r], acc[NT - 1][r]} * cof;
          if (TRAIN) { acc[0][r] = de.x; acc[NT - 1][r] = de.y; }
          gx2 = fma2(de, wx, gx2);
          gy2 = fma2(de, wy, gy2);
          if (ANG) gt2 = fma2(de, fr, gt2);   // spatial features have no theta dependence (fr = 0)
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
#pragma unroll 1
    for (int mt = 0; mt < first_angle_kt; ++mt) l1t_tile(std::false_type{}, mt);
#pragma unroll 1
    for (int mt = first_angle_kt; mt < NKT; ++mt) l1t_tile(std::true_type{}, mt);
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
  }
  if (TRAIN) {  // fixed-order loss partial: lanes of a wave (xor tree), then waves / workgroups in the final kernel
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
    if (lane == 0) a.loss_partial[blockIdx.x * WAVES + wave] = loss_acc;
    // dW3[:100] partial of this wave: sum over its 16 point lanes (xor tree, fixed order), h2 slot = 16 mt + 4 g + r
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
}

// ---- host side ----------------------------------------------------------------------------------------------
static int g_num_cus[MAX_DEVICES] = {};

int query_cus() {
  const int dev = current_device();
  if (dev < 0) return 256;
  if (g_num_cus[dev] > 0) return g_num_cus[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  g_num_cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return g_num_cus[dev];
}

template <int NKT, int NT, int MODE = 0>
static int launch_t(const OnfKernelArgs& a, hipStream_t stream, int* grid_out = nullptr) {
  using L = Lds<NKT>;
  static bool attr_set[MAX_DEVICES] = {};
  auto kern = onf_fwd_bwd_kernel<NKT, NT, MODE>;
  const int rc_attr = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), L::BYTES, attr_set);
  if (rc_attr != NFOPP_OK) return rc_attr;
  constexpr int CH = WAVES * 16 * NT;
  long long n_chunks = (a.n_points + CH - 1) / CH;
  long long grid = query_cus();
  if (grid > n_chunks) grid = n_chunks;
  if (grid_out) *grid_out = (int)grid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), L::BYTES, stream, a);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}

int launch_onf_kernel(const OnfKernelArgs& a, hipStream_t stream) {
  if (a.n_points <= 0) return NFOPP_OK;
  if (onf_split_enabled()) return launch_onf_split_kernel(a, stream, false);
  const int nkt = (a.geom.fin + 15) / 16;
  // small jobs: one tile per wave so that more CUs take part; large jobs: two tiles per wave (half the LDS reads)
  const bool small = a.n_points < (long long)query_cus() * WAVES * 16 * 2;
  switch (nkt) {
    case 14: return small ? launch_t<14, 1>(a, stream) : launch_t<14, 2>(a, stream);
    case 13: return small ? launch_t<13, 1>(a, stream) : launch_t<13, 2>(a, stream);
    case 8: return launch_t<8, 1>(a, stream);   // 7-8 input tiles: NT = 2 spills (S1 = 129 images), keep one tile
    case 7: return launch_t<7, 1>(a, stream);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

int launch_onf_logits_kernel(const OnfKernelArgs& a, hipStream_t stream) {
  if (a.n_points <= 0) return NFOPP_OK;
  if (onf_split_enabled()) return launch_onf_split_kernel(a, stream, true);
  const int nkt = (a.geom.fin + 15) / 16;
  const bool small = a.n_points < (long long)query_cus() * WAVES * 16 * 2;
  switch (nkt) {
    case 14: return small ? launch_t<14, 1, 2>(a, stream) : launch_t<14, 2, 2>(a, stream);
    case 13: return small ? launch_t<13, 1, 2>(a, stream) : launch_t<13, 2, 2>(a, stream);
    case 8: return launch_t<8, 1, 2>(a, stream);
    case 7: return launch_t<7, 1, 2>(a, stream);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

// training forward/backward pass (factor matrices for csrc/onf_wgrad.hip); one tile per wave: the factor stores
// need the registers the second tile would take
int launch_onf_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out) {
  if (onf_split_enabled()) return launch_onf_split_train_kernel(a, stream, grid_out);
  const int nkt = (a.geom.fin + 15) / 16;
  switch (nkt) {
    case 14: return launch_t<14, 1, 1>(a, stream, grid_out);
    case 13: return launch_t<13, 1, 1>(a, stream, grid_out);
    case 8: return launch_t<8, 1, 1>(a, stream, grid_out);
    case 7: return launch_t<7, 1, 1>(a, stream, grid_out);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

int onf_train_grid_upper_bound() { return query_cus(); }

// ---- early stop: stable compaction of the live trajectory indices (one workgroup; B is a few thousand per GPU) ------
// live[0] = count, live[1 + k] = index of the k-th trajectory with active[b] != 0, ascending.
constexpr int CP_THREADS = 1024;
__global__ __launch_bounds__(CP_THREADS) void compact_live_kernel(const unsigned char* active, long long batch, int* live) {
  __shared__ int wave_sum[CP_THREADS / 64];
  __shared__ int wave_off[CP_THREADS / 64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long per = (batch + CP_THREADS - 1) / CP_THREADS;
  const long long lo = tid * per, hi = lo + per < batch ? lo + per : batch;
  int mine = 0;
  for (long long b = lo; b < hi; ++b) mine += active[b] != 0;
  int scan = mine;   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(scan, o);
    if (lane >= o) scan += up;
  }
  if (lane == 63) wave_sum[wave] = scan;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < CP_THREADS / 64; ++w) { wave_off[w] = run; run += wave_sum[w]; }
    wave_off[CP_THREADS / 64] = run;
    live[0] = run;
  }
  __syncthreads();
  int pos = wave_off[wave] + scan - mine;
  for (long long b = lo; b < hi; ++b)
    if (active[b] != 0) live[1 + pos++] = (int)b;
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_onf_eval_points(const nfopp_onf_config* cfg, const float* params_dev, const float* points_dev,
                                     int64_t n_points, float* out4_dev, void* stream) {
  OnfKernelArgs a = {};
  NFOPP_REQUIRE(make_geom(cfg, &a.geom), "bad ONF configuration");
  NFOPP_REQUIRE(n_points >= 0, "negative point count");
  NFOPP_REQUIRE(n_points == 0 || (params_dev && points_dev && out4_dev), "null device pointer");
  a.params = params_dev;
  a.points = points_dev;
  a.n_points = n_points;
  a.out4 = out4_dev;
  return launch_onf_kernel(a, (hipStream_t)stream);
}

extern "C" int nfopp_onf_eval_logits(const nfopp_onf_config* cfg, const float* params_dev, const float* points_dev,
                                     int64_t n_points, float* out4_dev, void* stream) {
  OnfKernelArgs a = {};
  NFOPP_REQUIRE(make_geom(cfg, &a.geom), "bad ONF configuration");
  NFOPP_REQUIRE(n_points >= 0, "negative point count");
  NFOPP_REQUIRE(n_points == 0 || (params_dev && points_dev && out4_dev), "null device pointer");
  a.params = params_dev;
  a.points = points_dev;
  a.n_points = n_points;
  a.out4 = out4_dev;
  return launch_onf_logits_kernel(a, (hipStream_t)stream);
}

extern "C" int nfopp_traj_collision_eval(const nfopp_onf_config* cfg, const float* params_dev, const float* traj_dev,
                                         int64_t batch, int32_t n_waypoints, int32_t dim, float* t_dev,
                                         int32_t t_mode, uint64_t seed, uint64_t rng_offset,
                                         int64_t traj_index_offset, float* out4_dev, const uint8_t* active_dev,
                                         int32_t* live_ws_dev, void* stream) {
  OnfKernelArgs a = {};
  NFOPP_REQUIRE(make_geom(cfg, &a.geom), "bad ONF configuration");
  NFOPP_REQUIRE(batch >= 0 && n_waypoints >= 2, "need batch >= 0 and at least 2 waypoints");
  NFOPP_REQUIRE(batch == 0 || (params_dev && traj_dev && t_dev && out4_dev), "null device pointer");
  NFOPP_REQUIRE(dim == a.geom.point_dim, "trajectory dim %d does not match the ONF point dim %d", dim,
                a.geom.point_dim);
  NFOPP_REQUIRE(t_mode == 0 || t_mode == 1, "t_mode must be 0 (read) or 1 (Philox)");
  NFOPP_REQUIRE(batch <= 0x7fffffffLL, "batch too large for one launch");
  NFOPP_REQUIRE(!active_dev || live_ws_dev, "an active mask needs the live-list workspace (batch + 1 int32)");
  if (batch == 0) return NFOPP_OK;
  if (active_dev) {
    hipLaunchKernelGGL(compact_live_kernel, dim3(1), dim3(CP_THREADS), 0, (hipStream_t)stream, active_dev,
                       (long long)batch, live_ws_dev);
    NFOPP_HIP(hipGetLastError());
    a.live = live_ws_dev;
  }
  a.params = params_dev;
  a.traj = traj_dev;
  a.n_way = n_waypoints;
  a.dim = dim;
  a.t = t_dev;
  a.t_mode = t_mode;
  a.seed = seed;
  a.rng_offset = rng_offset;
  a.traj_index_offset = traj_index_offset;
  a.n_points = batch * (int64_t)(n_waypoints - 1);
  a.out4 = out4_dev;
  return launch_onf_kernel(a, (hipStream_t)stream);
}
