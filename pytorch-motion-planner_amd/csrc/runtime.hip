// Library-level plumbing: error strings, device query, the flat Adam update.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>

#include "common.h"

namespace nfopp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return e == hipErrorNoDevice ? NFOPP_ERR_NO_DEVICE : NFOPP_ERR_HIP;
}

int current_device() {
  int dev = -1;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { hip_fail(e, "hipGetDevice"); return -1; }
  if (dev < 0 || dev >= MAX_DEVICES) { set_error("device index %d out of range", dev); return -1; }
  return dev;
}

int ensure_dynamic_lds(const void* kernel, size_t bytes, bool* flags) {
  const int dev = current_device();
  if (dev < 0) return NFOPP_ERR_HIP;
  if (!flags[dev]) {
    NFOPP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    flags[dev] = true;
  }
  return NFOPP_OK;
}

// ---- content versions of ONF parameter buffers (nfopp_onf_params_version) -------------------------------------------
// (device, pointer) -> version the caller vouches for; 0 / absent = unknown (the split kernels then rebuild their pre-split
// weight image in front of every launch, as they always did).
static std::map<std::pair<int, const float*>, unsigned long long> g_param_versions;
static std::mutex g_param_mutex;

unsigned long long onf_params_version_of(const float* params_dev) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(g_param_mutex);
  auto it = g_param_versions.find({dev, params_dev});
  return it == g_param_versions.end() ? 0ull : it->second;
}

void onf_params_invalidate(const float* params_dev) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_param_mutex);
  g_param_versions.erase({dev, params_dev});
}

// torch.optim.Adam single-tensor path on a flat buffer (ONF weights: nfop/nerf_opt_planner.py:90)
__global__ void adam_flat_kernel(float* p, const float* g, float* m, float* v, long long n, float beta2, float omb1,
                                 float omb2, float eps, float step_size, float bc2_sqrt) {
  for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
    const float gk = g[k];
    float mk = m[k], vk = v[k];
    mk = mk + omb1 * (gk - mk);
    vk = vk * beta2 + (omb2 * gk) * gk;
    const float denom = sqrtf(vk) / bc2_sqrt + eps;
    p[k] = p[k] - step_size * (mk / denom);
    m[k] = mk;
    v[k] = vk;
  }
}

}  // namespace nfopp

using namespace nfopp;

extern "C" int nfopp_abi_version(void) { return NFOPP_ABI_VERSION; }

extern "C" const char* nfopp_last_error(void) { return g_err; }

extern "C" int nfopp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int64_t nfopp_onf_param_count(const nfopp_onf_config* cfg) {
  OnfGeom g;
  if (!make_geom(cfg, &g)) {
    set_error("bad ONF configuration");
    return NFOPP_ERR_ARG;
  }
  return g.n_params;
}

extern "C" int nfopp_onf_params_version(const float* params_dev, uint64_t version) {
  NFOPP_REQUIRE(params_dev, "null device pointer");
  const int dev = current_device();
  if (dev < 0) return NFOPP_ERR_HIP;
  std::lock_guard<std::mutex> lock(g_param_mutex);
  if (version) g_param_versions[{dev, params_dev}] = version;
  else g_param_versions.erase({dev, params_dev});
  return NFOPP_OK;
}

extern "C" int nfopp_adam_step(float* param_dev, const float* grad_dev, float* m_dev, float* v_dev, int64_t n,
                               float beta2, float omb1, float omb2, float eps, float step_size, float bc2_sqrt,
                               void* stream) {
  NFOPP_REQUIRE(param_dev && grad_dev && m_dev && v_dev, "null device pointer");
  NFOPP_REQUIRE(n >= 0, "negative length");
  if (n == 0) return NFOPP_OK;
  onf_params_invalidate(param_dev);   // the buffer changes: a version vouched for it (nfopp_onf_params_version) no longer holds
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param_dev, grad_dev,
                     m_dev, v_dev, (long long)n, beta2, omb1, omb2, eps, step_size, bc2_sqrt);
  NFOPP_HIP(hipGetLastError());
  return NFOPP_OK;
}
