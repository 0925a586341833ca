// Does an SALU instruction that reads an SGPR pair see the value a VALU compare wrote in the
// instruction before it?  Each wave alternates the compared value, so the right answer of every
// test is known; a second (padded) form of the same sequence is the control.
//   hipcc --offload-arch=gfx950 -O3 valu_sgpr_salu.hip -o valu_sgpr_salu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PAD, int FILL>
__global__ __launch_bounds__(256) void probe(uint32_t* wrong, int iters) {
  uint32_t bad = 0;
  uint32_t filler = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const uint32_t val = (it & 1) ? 5u : 0u;  // every lane the same: any(val > 1) == (it & 1)
    uint32_t v = val + (filler & 0u);
    uint32_t r;
    uint64_t m;
    // some vector work in front, so the compare is issued into a busy pipe
#pragma unroll
    for (int f = 0; f < FILL; ++f) filler = filler * 0x9E3779B1u + 0x7F4A7C15u;
    if (PAD)
      asm volatile("v_cmp_lt_u32_e64 %1, 1, %2\n\ts_nop 4\n\ts_cmp_eq_u64 %1, 0\n\ts_cselect_b32 %0, 0, 1"
                   : "=s"(r), "=&s"(m) : "v"(v) : "scc");
    else
      asm volatile("v_cmp_lt_u32_e64 %1, 1, %2\n\ts_cmp_eq_u64 %1, 0\n\ts_cselect_b32 %0, 0, 1"
                   : "=s"(r), "=&s"(m) : "v"(v) : "scc");
    bad += r != (uint32_t)(it & 1);
  }
  if (bad && (threadIdx.x & 63) == 0) atomicAdd(wrong, bad);
  if (filler == 0x12345u) wrong[1] = filler;
}

template <int PAD, int FILL>
void run(const char* name, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<PAD, FILL>), dim3(wgs), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    uint32_t h = 0;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-34s %5d workgroups x 4 waves, %d tests per wave: %u wrong\n", name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  run<0, 0>("back to back, no filler", 256, 20000);
  run<0, 0>("back to back, no filler", 2048, 20000);
  run<0, 8>("back to back, 8 vector ops ahead", 256, 20000);
  run<0, 8>("back to back, 8 vector ops ahead", 2048, 20000);
  run<0, 8>("back to back, 8 vector ops ahead", 8192, 5000);
  run<1, 8>("s_nop 4 between (control)", 2048, 20000);
  run<1, 8>("s_nop 4 between (control)", 8192, 5000);
  return 0;
}
