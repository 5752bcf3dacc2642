// Micro-experiment (dev tool, run ONCE): the one precondition tools/micro/pk_opsel.hip never tested.
//
// Round 2's run-to-run differences in the weight-gradient pass were traced to this emitted sequence (pre-fix onf_wgrad.hip,
// commit d11dd17, `hipcc -S`; DESIGN.md K5 "A hazard hipcc does not pad"):
//     v_mfma_f32_16x16x32_bf16 v[54:57], v[150:153], v[154:157], v[54:57]     ; last of 12, A operand = v[150:153]
//     s_nop 15
//     v_mad_u64_u32 v[150:151], ...                                           ; LDS address
//     ds_read_b64  v[152:153], v150                                           ; (u_x, u_y) of the sample
//     ds_read_b32  v150, v150 offset:12
//     s_waitcnt lgkmcnt(1)
//     v_pk_fma_f32 v[206:207], v[14:15], v[152:153], v[18:19] op_sel:[0,1,0]  ; LOW lane takes the pair's HIGH dword
// The wait count is right for in-order LDS returns and the instruction alone honours op_sel (pk_opsel.hip), yet the low
// lane sometimes saw the PREVIOUS sample's u_y.  This kernel replays exactly that with pinned registers: per iteration
// every lane stores a fresh pair into its own LDS slot, runs 12 MFMAs whose A operand is v[100:103], then reads the
// pair back into v[102:103] (+ the trailing ds_read_b32 into the address register), waits with lgkmcnt(1) and issues the
// packed fma -- 512-thread workgroups on every CU (two waves per SIMD).  A stale read shows the previous iteration's
// value, which differs.  Variants:
//   0  the sequence as emitted                      1  + s_nop 1 behind the s_waitcnt
//   2  op_sel_hi:[1,0,1] (low lane <- LOW dword)    3  as 0 without the MFMA group in front
//   4  as 0 with s_waitcnt lgkmcnt(0)
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/pk_opsel_lds.hip -o tools/micro/pk_opsel_lds
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

#define MFMA "v_mfma_f32_16x16x32_bf16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t"
#define MFMA12 MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA "s_nop 15\n\t"
#define HEAD                                                                                                 \
  "v_mov_b32 v112, %[xx]\n\tv_mov_b32 v113, %[xy]\n\tv_mov_b32 v114, %[zx]\n\tv_mov_b32 v115, %[zy]\n\t"      \
  "v_mov_b32 v118, %[ylo]\n\tv_mov_b32 v119, %[yhi]\n\t"                                                     \
  "ds_write_b64 %[addr], v[118:119]\n\tds_write_b32 %[addr], v118 offset:12\n\t"                             \
  "v_mov_b32 v100, 0x3f803f80\n\tv_mov_b32 v101, 0x3f803f80\n\tv_mov_b32 v102, 0x3f803f80\n\tv_mov_b32 v103, 0x3f803f80\n\t" \
  "v_mov_b32 v104, 0x3f803f80\n\tv_mov_b32 v105, 0x3f803f80\n\tv_mov_b32 v106, 0x3f803f80\n\tv_mov_b32 v107, 0x3f803f80\n\t" \
  "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
#define READ                                                                                                  \
  "v_mov_b32 v100, %[addr]\n\tv_mov_b32 v101, 0\n\ts_nop 0\n\t"                                               \
  "ds_read_b64 v[102:103], v100\n\tds_read_b32 v100, v100 offset:12\n\t"
#define TAIL "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %[d0], v116\n\tv_mov_b32 %[d1], v117\n\tv_mov_b32 %[w], v100\n\t"
#define PK_HI "v_pk_fma_f32 v[116:117], v[112:113], v[102:103], v[114:115] op_sel:[0,1,0]\n\t"
#define PK_LO "v_pk_fma_f32 v[116:117], v[112:113], v[102:103], v[114:115] op_sel_hi:[1,0,1]\n\t"
#define CLOBBERS "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
                 "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void replay(int iters, unsigned* mismatches) {
  __shared__ __attribute__((aligned(16))) float slots[512 * 4];
  const int tid = threadIdx.x;
  const unsigned addr = (unsigned)(size_t)(slots + 4 * tid);   // LDS byte address of this lane's slot
  unsigned bad = 0;
  const float xx = 1.0f + 0.001f * tid, xy = 2.0f + 0.002f * tid, zx = 0.25f, zy = 0.5f;
  for (int it = 0; it < iters; ++it) {
    const float ylo = 3.0f + 0.5f * (float)(it & 1023) + 0.01f * tid, yhi = -7.0f - 0.25f * (float)(it & 511) - 0.02f * tid;
    float d0, d1, w;
#define OPERANDS : [d0] "=v"(d0), [d1] "=v"(d1), [w] "=v"(w) \
                 : [addr] "v"(addr), [xx] "v"(xx), [xy] "v"(xy), [ylo] "v"(ylo), [yhi] "v"(yhi), [zx] "v"(zx), [zy] "v"(zy) : CLOBBERS
    if (VARIANT == 0) asm volatile(HEAD MFMA12 READ "s_waitcnt lgkmcnt(1)\n\t" PK_HI TAIL OPERANDS);
    if (VARIANT == 1) asm volatile(HEAD MFMA12 READ "s_waitcnt lgkmcnt(1)\n\ts_nop 1\n\t" PK_HI TAIL OPERANDS);
    if (VARIANT == 2) asm volatile(HEAD MFMA12 READ "s_waitcnt lgkmcnt(1)\n\t" PK_LO TAIL OPERANDS);
    if (VARIANT == 3) asm volatile(HEAD READ "s_waitcnt lgkmcnt(1)\n\t" PK_HI TAIL OPERANDS);
    if (VARIANT == 4) asm volatile(HEAD MFMA12 READ "s_waitcnt lgkmcnt(0)\n\t" PK_HI TAIL OPERANDS);
    const float y = VARIANT == 2 ? ylo : yhi;
    if (d0 != fmaf(xx, y, zx) || d1 != fmaf(xy, y, zy) || w != ylo) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

template <int V>
static void run(const char* what) {
  unsigned* d;
  hipMalloc(&d, 4);
  hipMemset(d, 0, 4);
  const int iters = 20000;
  hipLaunchKernelGGL(replay<V>, dim3(1024), dim3(512), 0, 0, iters, d);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("variant %d (%s): %u mismatching packed fmas of %.3g\n", V, what, h, 1024.0 * 512 * iters);
  hipFree(d);
}

int main() {
  run<0>("as emitted: ds_read_b64 -> lgkmcnt(1) -> v_pk_fma op_sel:[0,1,0], MFMA group in front");
  run<1>("+ s_nop 1 behind the wait");
  run<2>("op_sel_hi:[1,0,1], low lane reads the low dword");
  run<3>("as emitted, no MFMA group in front");
  run<4>("as emitted with lgkmcnt(0)");
  return 0;
}
