// v_lshlrev_b64 right after the vector instruction that wrote one of its operands (the shift amount,
// or the 64-bit value's halves): hipcc emits both back to back, e.g.
//     v_or_b32_e32 v1, 6, v167 ; v_lshlrev_b64 v[2:3], v1, 1
//     v_not_b32_e32 v241, v179 ; v_not_b32_e32 v240, v178 ; v_cndmask_b32 ... ; v_lshlrev_b64 v[240:241], v47, v[240:241]
// Does the shift always see the new operand, also with other waves of the SIMD issuing 64-bit shifts?
//   hipcc --offload-arch=gfx950 -O3 shift64_raw.hip -o shift64_raw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>  // 0: amount written by the previous instruction; 1: value halves written by the two previous ones; 2: mode 0 padded (control)
__global__ __launch_bounds__(256) void probe(uint32_t* wrong, int iters) {
  uint32_t bad = 0;
  uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  for (int it = 0; it < iters; ++it) {
    seed = seed * 1664525u + 1013904223u;
    const uint32_t amt = (seed >> 9) & 31, src = seed | 1u;
    uint64_t r;
    uint32_t t;
    if (MODE == 0) {
      asm volatile("v_and_b32 %1, 31, %2\n\tv_lshlrev_b64 %0, %1, 1" : "=&v"(r), "=&v"(t) : "v"(seed >> 9));
      bad += r != (1ull << amt);
    } else if (MODE == 2) {
      asm volatile("v_and_b32 %1, 31, %2\n\ts_nop 4\n\tv_lshlrev_b64 %0, %1, 1" : "=&v"(r), "=&v"(t) : "v"(seed >> 9));
      bad += r != (1ull << amt);
    } else {
      asm volatile("v_not_b32 v101, %2\n\tv_not_b32 v100, %1\n\tv_lshlrev_b64 %0, %3, v[100:101]"
                   : "=&v"(r) : "v"(src), "v"(~src), "v"(amt) : "v100", "v101");
      bad += r != ((((uint64_t)src << 32) | (uint32_t)~src) << amt);
    }
  }
  if (bad) atomicAdd(wrong, bad);
}

template <int MODE>
void run(const char* name, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<MODE>), dim3(wgs), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    uint32_t h = 0;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-62s %5d workgroups, %6d shifts per lane: %u wrong results\n", name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  run<0>("shift amount written by the previous instruction", 256, 20000);
  run<0>("shift amount written by the previous instruction", 4096, 20000);
  run<1>("both halves of the value written by the two previous ones", 4096, 20000);
  run<2>("shift amount, s_nop 4 between (control)", 4096, 20000);
  return 0;
}
