// A VALU instruction reads an SGPR pair as a mask operand and the NEXT instruction -- scalar ALU --
// overwrites that pair (hipcc emits this: "v_cndmask_b32_e64 v32, 0, 1, s[34:35]; s_cselect_b64 s[34:35], -1, 0").
// Does the VALU instruction always see the old value, also when the other waves of the SIMD keep the
// vector ALU busy with quarter-rate instructions?  Odd waves are the busy neighbours.
//   hipcc --offload-arch=gfx950 -O3 sgpr_war.hip -o sgpr_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* wrong, int iters) {
  uint32_t bad = 0;
  const bool busy = MODE != 2 && ((threadIdx.x >> 6) & 1);
  if (busy) {
    uint64_t x = threadIdx.x * 0x9E3779B97F4A7C15ull + 1;
    uint32_t y = threadIdx.x + 3;
    for (int it = 0; it < iters * 4; ++it) {  // 64-bit shifts and 32-bit multiplies: quarter rate
      x = (x << (y & 31)) ^ (x >> 7);
      y = y * 0x01000193u + (uint32_t)x;
    }
    if (x == 12345 && y == 7) wrong[1] = 1;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    uint32_t v;
    uint64_t m;
    if (MODE == 0 || MODE == 2)  // write-after-read: the mask is all ones when read, zeroed by the next instruction
      asm volatile("s_mov_b64 %1, -1\n\ts_nop 2\n\tv_cndmask_b32_e64 %0, 0, 1, %1\n\ts_mov_b64 %1, 0\n\ts_nop 2"
                   : "=v"(v), "=&s"(m) : : "memory");
    else  // the compiler's own sequence: s_cselect, three instructions, v_cndmask, s_cselect over the same pair
      asm volatile("s_cmp_eq_u32 0, 0\n\ts_cselect_b64 %1, -1, 0\n\ts_cmp_lg_u32 0, 0\n\ts_mul_i32 s90, s91, 12\n\t"
                   "v_cndmask_b32_e64 %0, 0, 1, %1\n\ts_cselect_b64 %1, -1, 0\n\ts_nop 2"
                   : "=v"(v), "=&s"(m) : : "memory", "scc", "s90");
    bad += v != 1u;
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
    printf("%-58s %5d workgroups, %6d tests per lane: %u lanes read the overwritten value\n", name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  run<2>("v_cndmask reads s[n:n+1], s_mov overwrites it (no neighbours)", 2048, 20000);
  run<0>("v_cndmask reads s[n:n+1], s_mov overwrites it (busy neighbours)", 2048, 20000);
  run<1>("the compiler's s_cselect / v_cndmask / s_cselect sequence", 2048, 20000);
  return 0;
}
