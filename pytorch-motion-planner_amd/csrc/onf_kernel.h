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
  // training mode (TRAIN kernels only): labels, BCE normalisation, per-sample factor matrices for the weight-gradient
  // GEMMs (csrc/onf_wgrad.hip), per-wave loss partials
  const float* labels;
  float inv_count;
  int aug_feature;   // zero-weight pad feature evaluated as cos(0) = 1: the "ones" column of the input matrix
  float* ws_in;      // [P, 16*NKT]   input features in slot order, ones column at the slot of aug_feature
  float* ws_h1;      // [P, 112]      relu(a1) in slot order (layout Q), ones at slot 16*6 + 1
  float* ws_h2;      // [P, 112]      relu(a2) (layout P)
  float* ws_dh1;     // [P, 112]      d loss / d a1, rho at slot 16*6 + 1
  float* ws_dh2;     // [P, 112]      d loss / d a2, rho at slot 16*6 + 1
  float* ws_de;      // [P, 16*NKT]   d loss / d (encoding argument)
  float* ws_u;       // [P, 4]        (ux, uy, 1, theta)
  float* loss_partial;  // [grid * WAVES]
};

int query_cus();
int launch_onf_kernel(const OnfKernelArgs& a, hipStream_t stream);
// csrc/onf_split.hip: the same kernels with every GEMM issued as bf16x3 split-precision products (mode 0 / 2)
int launch_onf_split_kernel(const OnfKernelArgs& a, hipStream_t stream, bool forward_only);
bool onf_split_enabled();
int launch_onf_logits_kernel(const OnfKernelArgs& a, hipStream_t stream);
int launch_onf_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out);
int onf_train_grid_upper_bound();

// csrc/onf_wgrad.hip: MFMA weight-gradient path of the ONF fitting step
size_t wgrad_workspace_bytes(const OnfGeom& g, long long n_samples);
int onf_train_grad_mfma(const OnfGeom& g, const float* params, const float* samples, const float* labels,
                        long long n_samples, float inv_count, float* grad, float* ws, hipStream_t st);

}  // namespace nfopp
