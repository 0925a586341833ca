// scratch_store_dwordx2 / scratch_load_dwordx2 at an offset that is 4 mod 8 (hipcc emits these for
// 64-bit spills: "scratch_store_dwordx2 off, v[12:13], off offset:108").  Every lane writes four
// dwords, overwrites the middle two with one misaligned dwordx2 store, reads all four back singly
// and once more through a misaligned dwordx2 load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void probe(uint32_t* bad, int iters) {
  volatile uint32_t priv[48];
  priv[threadIdx.x % 48] = 1u;
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  uint32_t n_bad = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t a = id * 4u + 0x10000000u * (it & 7), b = a + 1, c = a + 2, d = a + 3;
    uint64_t mid = ((uint64_t)(0xC0000000u | id) << 32) | (0x80000000u | (id ^ it));
    uint32_t r0, r1, r2, r3;
    uint64_t rm;
    asm volatile(
        "scratch_store_dword off, %6, off offset:64\n\tscratch_store_dword off, %7, off offset:68\n\t"
        "scratch_store_dword off, %8, off offset:72\n\tscratch_store_dword off, %9, off offset:76\n\t"
        "scratch_store_dwordx2 off, %5, off offset:68\n\ts_waitcnt vmcnt(0)\n\t"
        "scratch_load_dword %0, off, off offset:64\n\tscratch_load_dword %1, off, off offset:68\n\t"
        "scratch_load_dword %2, off, off offset:72\n\tscratch_load_dword %3, off, off offset:76\n\t"
        "scratch_load_dwordx2 %4, off, off offset:68\n\ts_waitcnt vmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(rm)
        : "v"(mid), "v"(a), "v"(b), "v"(c), "v"(d)
        : "memory");
    n_bad += (r0 != a) + (r1 != (uint32_t)mid) + (r2 != (uint32_t)(mid >> 32)) + (r3 != d) + (rm != mid);
  }
  if (n_bad) atomicAdd(bad, n_bad);
  if (priv[(threadIdx.x + 3) % 48] == 77u) bad[1] = 1;
}

int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int wgs : {256, 768, 4096})
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(d, 0, 8);
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, d, 500);
      (void)hipDeviceSynchronize();
      uint32_t h = 0;
      (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
      printf("misaligned scratch dwordx2, %4d workgroups, 500 rounds per lane: %u wrong dwords\n", wgs, h);
    }
  return 0;
}
