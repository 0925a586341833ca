// v_lshlrev_b64 whose 32-bit shift amount sits in the LAST vector register of the wave's allocation
// (v255 of a 256-register kernel; hipcc's allocator puts values there like anywhere else).
// Found through the co-residency fault of DESIGN.md 3.2: in the failing wavefronts that instruction
// shifted by (v0 & 63) -- register 0 -- instead of by v255.
//   hipcc --offload-arch=gfx950 -O3 shift64_last_vgpr.hip -o shift64_last_vgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// Other instructions with a 64-bit result and a 32-bit vector operand, same placement (v255 of 256):
template <int MODE>
__global__ __launch_bounds__(256) void probe2(uint32_t* wrong, uint32_t* as_v0, int iters) {
  uint32_t bad = 0, like_v0 = 0;
  uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  for (int it = 0; it < iters; ++it) {
    seed = seed * 1664525u + 1013904223u;
    const uint32_t amt = (seed >> 9) & 31;
    const uint64_t x = ((uint64_t)seed << 32) | (seed * 7u + 3u);
    uint64_t r, want;
    uint32_t v0now;
    if (MODE == 0) {  // logical right shift
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_lshrrev_b64 %0, v255, %3\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt), "v"(x) : "v255");
      want = x >> amt;
      like_v0 += r != want && r == (x >> (v0now & 63));
    } else if (MODE == 1) {  // arithmetic right shift
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_ashrrev_i64 %0, v255, %3\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt), "v"(x) : "v255");
      want = (uint64_t)((int64_t)x >> amt);
      like_v0 += r != want && r == (uint64_t)((int64_t)x >> (v0now & 63));
    } else if (MODE == 2) {  // v_mad_u64_u32, first factor in v255
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_mad_u64_u32 %0, vcc, v255, %4, %3\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt), "v"(x), "v"(seed) : "v255", "vcc");
      want = (uint64_t)amt * seed + x;
      like_v0 += r != want && r == (uint64_t)v0now * seed + x;
    } else if (MODE == 3) {  // v_mad_u64_u32, second factor in v255
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_mad_u64_u32 %0, vcc, %4, v255, %3\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt), "v"(x), "v"(seed) : "v255", "vcc");
      want = (uint64_t)amt * seed + x;
      like_v0 += r != want && r == (uint64_t)v0now * seed + x;
    } else {  // v_lshl_add_u64, shift amount (0..4) in v255
      asm volatile("v_and_b32 v255, 3, %2\n\ts_nop 4\n\tv_lshl_add_u64 %0, %3, v255, %3\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt), "v"(x) : "v255");
      want = (x << (amt & 3)) + x;
      like_v0 += r != want && r == (x << (v0now & 7)) + x;
    }
    bad += r != want;
  }
  if (bad) atomicAdd(wrong, bad);
  if (like_v0) atomicAdd(as_v0, like_v0);
}

template <int MODE>  // 0: amount in v255 (last register); 1: amount in v254; 2: amount in v255, 32-bit shift (control)
__global__ __launch_bounds__(256) void probe(uint32_t* wrong, uint32_t* as_v0, int iters) {
  uint32_t bad = 0, like_v0 = 0;
  uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  for (int it = 0; it < iters; ++it) {
    seed = seed * 1664525u + 1013904223u;
    const uint32_t amt = (seed >> 9) & 31;
    uint64_t r;
    uint32_t r32, v0now;
    if (MODE == 0) {
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_lshlrev_b64 %0, v255, 1\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt) : "v255");
      bad += r != (1ull << amt);
      like_v0 += r != (1ull << amt) && r == (1ull << (v0now & 63));
    } else if (MODE == 1) {
      asm volatile("v_mov_b32 v254, %2\n\ts_nop 4\n\tv_lshlrev_b64 %0, v254, 1\n\tv_mov_b32 %1, v0" : "=&v"(r), "=&v"(v0now) : "v"(amt) : "v254", "v255");
      bad += r != (1ull << amt);
    } else {
      asm volatile("v_mov_b32 v255, %2\n\ts_nop 4\n\tv_lshlrev_b32 %0, v255, 1\n\tv_mov_b32 %1, v0" : "=&v"(r32), "=&v"(v0now) : "v"(amt) : "v255");
      bad += r32 != (1u << amt);
    }
  }
  if (bad) atomicAdd(wrong, bad);
  if (like_v0) atomicAdd(as_v0, like_v0);
}

template <int MODE>
void run(const char* name, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<MODE>), dim3(wgs), dim3(256), 0, 0, d, d + 1, iters);
    (void)hipDeviceSynchronize();
    uint32_t h[2] = {0, 0};
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%-50s %5d workgroups, %6d shifts per lane: %u wrong results (%u of them = 1 << (v0 & 63))\n", name, wgs, iters, h[0], h[1]);
  }
  (void)hipFree(d);
}

// Other allocation sizes: the amount in v167 of a 168-register kernel, v127 of 128, and v250 of a kernel that uses
// 251 registers (256 allocated: v250 is NOT the last allocated one).
#define PROBE_REG(NAME, REG, BOUNDS)                                                                              \
  __global__ __launch_bounds__(256, BOUNDS) void NAME(uint32_t* wrong, uint32_t* as_v0, int iters) {             \
    uint32_t bad = 0, like_v0 = 0;                                                                                \
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;                                          \
    for (int it = 0; it < iters; ++it) {                                                                          \
      seed = seed * 1664525u + 1013904223u;                                                                       \
      const uint32_t amt = (seed >> 9) & 31;                                                                      \
      uint64_t r;                                                                                                 \
      uint32_t v0now;                                                                                             \
      asm volatile("v_mov_b32 " REG ", %2\n\ts_nop 4\n\tv_lshlrev_b64 %0, " REG ", 1\n\tv_mov_b32 %1, v0"       \
                   : "=&v"(r), "=&v"(v0now) : "v"(amt) : REG);                                                    \
      bad += r != (1ull << amt);                                                                                  \
      like_v0 += r != (1ull << amt) && r == (1ull << (v0now & 63));                                               \
    }                                                                                                             \
    if (bad) atomicAdd(wrong, bad);                                                                               \
    if (like_v0) atomicAdd(as_v0, like_v0);                                                                       \
  }
PROBE_REG(probe_v167, "v167", 3)
PROBE_REG(probe_v127, "v127", 4)
PROBE_REG(probe_v250, "v250", 2)

template <typename K>
void run3(const char* name, K kernel, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  (void)hipMemset(d, 0, 8);
  hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, d, d + 1, iters);
  (void)hipDeviceSynchronize();
  uint32_t h[2] = {0, 0};
  (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("%-50s %5d workgroups, %6d per lane: %u wrong results (%u of them = 1 << (v0 & 63))\n", name, wgs, iters, h[0], h[1]);
  (void)hipFree(d);
}

template <int MODE>
void run2(const char* name, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  (void)hipMemset(d, 0, 8);
  hipLaunchKernelGGL((probe2<MODE>), dim3(wgs), dim3(256), 0, 0, d, d + 1, iters);
  (void)hipDeviceSynchronize();
  uint32_t h[2] = {0, 0};
  (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("%-50s %5d workgroups, %6d per lane: %u wrong results (%u of them as if the operand were v0)\n", name, wgs, iters, h[0], h[1]);
  (void)hipFree(d);
}

int main() {
  run<0>("v_lshlrev_b64, amount in v255 (last register)", 256, 2000);
  run<0>("v_lshlrev_b64, amount in v255 (last register)", 2048, 2000);
  run<1>("v_lshlrev_b64, amount in v254", 2048, 2000);
  run<2>("v_lshlrev_b32, amount in v255 (control)", 2048, 2000);
  run2<0>("v_lshrrev_b64, amount in v255", 2048, 2000);
  run2<1>("v_ashrrev_i64, amount in v255", 2048, 2000);
  run2<2>("v_mad_u64_u32, first factor in v255", 2048, 2000);
  run2<3>("v_mad_u64_u32, second factor in v255", 2048, 2000);
  run2<4>("v_lshl_add_u64, shift amount in v255", 2048, 2000);
  run3("v_lshlrev_b64, amount in v167 of 168 registers", probe_v167, 3072, 2000);
  run3("v_lshlrev_b64, amount in v127 of 128 registers", probe_v127, 4096, 2000);
  run3("v_lshlrev_b64, amount in v250 of 251 (256 allocated)", probe_v250, 2048, 2000);
  return 0;
}
