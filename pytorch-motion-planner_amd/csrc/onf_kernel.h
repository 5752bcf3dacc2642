// Argument block and launch entry points of the fused ONF kernels (csrc/onf_fused.hip), shared with the
// weight-gradient path (csrc/onf_wgrad.hip).
#pragma once
#include "common.h"

namespace nfopp {

struct OnfKernelArgs {
  OnfGeom geom;
  const float* params;
  // explicit-point mode
  const float* points;
  // trajectory mode (points == nullptr)
  const float* traj;
  int n_way, dim;
  float* t;
  int t_mode;
  unsigned long long seed, rng_offset;
  long long traj_index_offset;
  long long n_points;
  // early stop (trajectory mode): live[0] = number of live trajectories, live[1 + k] = index of the k-th one (ascending).
  // The kernel then walks live[0] * (n_way - 1) samples; t / out4 rows keep their place (trajectory * (n_way-1) + j).
  const int* live;
  float* out4;
  // training mode (TRAIN kernels only): labels, BCE normalisation, the per-sample factors the weight-gradient GEMMs
  // (csrc/onf_wgrad.hip) cannot rebuild cheaply, per-wave loss and dW3[:100] partials.  NOT stored (rebuilt in pass 2
  // from the 48-byte record): the input features `in` (a function of u) and dh2 (= rho * W3a * [a2 > 0]); h2 never
  // leaves the kernel (its only use, dW3[:100] = sum_p rho_p h2_p, is accumulated here).
  const float* labels;
  float inv_count;
  int aug_feature;   // zero-weight pad feature evaluated as cos(0) = 1: the "ones" column of the input matrix
  // (slot orders below: the 16x16 kernels'; csrc/onf_x32.hip stores by index -- ones unit 101, rho row 100, ones feature fin,
  //  rows of 16 * ((fin + 16) / 16) floats in ws_de, no g4 partials: csrc/onf_wgrad.hip, WgradArgs::x32_order)
  float* ws_h1;      // [P, 112]      relu(a1) in slot order (layout Q), ones at slot 16*6 + 1
  float* ws_dh1;     // [P, 112]      d loss / d a1, rho at slot 16*6 + 1
  float* ws_de;      // [P, 16*NKT]   d loss / d (encoding argument)
  float* ws_u;       // [P, 12]       (ux, uy, 1, theta | rho, 0, 0, 0 | 4 words: bit 4*tile + r of word g = [a2 > 0])
  float* g4_partial; // [grid * WAVES, 112]  per-wave sum_p rho_p * relu(a2_p) in h2 slot order (layout P)
  float* loss_partial;  // [grid * WAVES]
};

int query_cus();
int launch_onf_kernel(const OnfKernelArgs& a, hipStream_t stream);
// csrc/onf_split.hip: the same kernels with every GEMM issued as bf16x3 split-precision products (mode 0 / 2)
int launch_onf_split_kernel(const OnfKernelArgs& a, hipStream_t stream, bool forward_only);
int launch_onf_split_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out);
bool onf_split_enabled();
// csrc/onf_x32.hip: the bf16x3 split path on 32x32x16 tiles (mode 0 / 2), one 32-sample tile per wave
bool onf_x32_supports(const OnfGeom& g);
bool onf_use_x32(const OnfGeom& g);   // matrix path 1 and a feature dimension the 32x32x16 kernel covers
int launch_onf_x32_kernel(const OnfKernelArgs& a, hipStream_t stream, bool forward_only);
// training pass of csrc/onf_x32.hip: factors in x32 order (csrc/onf_wgrad.hip: WgradArgs::x32_order), loss partials [grid * 8]
int launch_onf_x32_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out);
int launch_onf_logits_kernel(const OnfKernelArgs& a, hipStream_t stream);
int launch_onf_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out);
int onf_train_grid_upper_bound();

// csrc/onf_wgrad.hip: MFMA weight-gradient path of the ONF fitting step
size_t wgrad_workspace_bytes(const OnfGeom& g, long long n_samples);
int onf_train_grad_mfma(const OnfGeom& g, const float* params, const float* samples, const float* labels,
                        long long n_samples, float inv_count, float* grad, float* ws, hipStream_t st);

}  // namespace nfopp
