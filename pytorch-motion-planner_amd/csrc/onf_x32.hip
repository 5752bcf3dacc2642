// K1 on 32x32x16 tiles (csrc/onf_x32_impl.h): per-stream weight images and the entry points.  The kernels themselves are
// instantiated per feature dimension in csrc/onf_x32_k{14,13,8,7}.hip.
#include "onf_x32_impl.h"

namespace nfopp {
namespace x32 {

constexpr int MAX_SLOTS = 16;
static Slot g_slots[MAX_DEVICES][MAX_SLOTS] = {};
static unsigned long long g_stamp = 0;
static std::mutex g_mutex;

int buffers_for_stream(size_t bytes, hipStream_t stream, void** out, Slot** slot_out) {
  const int dev = current_device();
  if (dev < 0) return NFOPP_ERR_HIP;
  std::lock_guard<std::mutex> lock(g_mutex);
  Slot* slot = nullptr;
  for (int k = 0; k < MAX_SLOTS && !slot; ++k)
    if (g_slots[dev][k].used && g_slots[dev][k].stream == stream) slot = &g_slots[dev][k];
  for (int k = 0; k < MAX_SLOTS && !slot; ++k)
    if (!g_slots[dev][k].used) { slot = &g_slots[dev][k]; slot->used = true; }
  if (!slot) {   // every slot taken: reuse the least recently used one (its stream's launches are done or ordered before ours
                 // only if it is idle, so wait for the device once; a 17th concurrent stream is not the realistic case)
    slot = &g_slots[dev][0];
    for (int k = 1; k < MAX_SLOTS; ++k)
      if (g_slots[dev][k].stamp < slot->stamp) slot = &g_slots[dev][k];
    NFOPP_HIP(hipDeviceSynchronize());
  }
  if (slot->stream != stream) slot->version = 0;   // another stream's image: never trusted
  slot->stream = stream;
  slot->stamp = ++g_stamp;
  if (slot->bytes < bytes) {
    if (slot->ptr) NFOPP_HIP(hipFree(slot->ptr));
    slot->ptr = nullptr; slot->bytes = 0; slot->version = 0;
    NFOPP_HIP(hipMalloc(&slot->ptr, bytes));
    slot->bytes = bytes;
  }
  *out = slot->ptr;
  *slot_out = slot;
  return NFOPP_OK;
}

void slot_built(Slot* slot, const float* params, unsigned long long version, const OnfGeom& geom, int nkb) {
  std::lock_guard<std::mutex> lock(g_mutex);
  slot->params = params; slot->version = version; slot->geom = geom; slot->nkb = nkb;
}

static int launch_nkb(int nkb, const OnfKernelArgs& a, hipStream_t stream, int mode, int* grid_out) {
  switch (nkb) {
    case 14: return launch_nkb14(a, stream, mode, grid_out);
    case 13: return launch_nkb13(a, stream, mode, grid_out);
    case 8: return launch_nkb8(a, stream, mode, grid_out);
    case 7: return launch_nkb7(a, stream, mode, grid_out);
    default:
      set_error("unsupported ONF feature dimension %d", a.geom.fin);
      return NFOPP_ERR_ARG;
  }
}

}  // namespace x32

// Feature dimensions the 32x32 kernel covers: one pad position must be free for the ones feature.
bool onf_x32_supports(const OnfGeom& g) {
  const int nkb = (g.fin + 16) >> 4;
  return (g.n_enc == 200 && (nkb == 14 || nkb == 13)) || (g.n_enc == 100 && (nkb == 8 || nkb == 7));
}

int launch_onf_x32_kernel(const OnfKernelArgs& a, hipStream_t stream, bool forward_only) {
  if (a.n_points <= 0) return NFOPP_OK;
  return x32::launch_nkb((a.geom.fin + 16) >> 4, a, stream, forward_only ? 2 : 0, nullptr);
}

// training pass (pass 1 of csrc/onf_wgrad.hip), factors in x32 order
int launch_onf_x32_train_kernel(const OnfKernelArgs& a, hipStream_t stream, int* grid_out) {
  return x32::launch_nkb((a.geom.fin + 16) >> 4, a, stream, 1, grid_out);
}

}  // namespace nfopp
