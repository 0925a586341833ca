// Is the data register of a vector-memory STORE safe to overwrite in the very next instruction?
// (hipcc does exactly that after spill stores and after ordinary global stores.)  Every lane queues a
// burst of loads, stores register X (value A) to scratch or to global memory, overwrites X with B in
// the next instruction, then reads the location back.  Reading B back means the store picked its data
// up after the overwrite.
//   hipcc --offload-arch=gfx950 -O3 store_data_war.hip -o store_data_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int BURST>  // 0: scratch store, 1: global store
__global__ __launch_bounds__(256) void probe(const uint32_t* src, uint32_t* sink, uint64_t words, uint32_t* bad, int iters) {
  volatile uint32_t priv[32];
  priv[threadIdx.x & 31] = 1u;
  uint32_t n_bad = 0, acc = 0;
  const uint64_t gid = blockIdx.x * 256ull + threadIdx.x;
  uint64_t x = gid * 0x9E3779B97F4A7C15ull + 1;
  uint32_t* q = sink + gid * 16;
  for (int it = 0; it < iters; ++it) {
    const uint32_t* pa[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // BURST: four cache-missing loads are queued ahead of the store, inside the same asm block
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      pa[b] = src + (x >> 20) % words;
    }
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    uint32_t data = 0xA0000000u | (uint32_t)it, other = 0xB0000000u | (uint32_t)it, got;
    if (MODE == 0 && BURST)
      asm volatile("global_load_dword %4, %8, off\n\tglobal_load_dword %5, %9, off\n\tglobal_load_dword %6, %10, off\n\t"
                   "global_load_dword %7, %11, off\n\t"
                   "scratch_store_dword off, %1, off offset:64\n\tv_mov_b32 %1, %2\n\ts_waitcnt vmcnt(0)\n\t"
                   "scratch_load_dword %0, off, off offset:64\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(got), "+v"(data), "+v"(other), "+v"(q), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                   : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]) : "memory");
    else if (MODE == 0)
      asm volatile("scratch_store_dword off, %1, off offset:64\n\tv_mov_b32 %1, %2\n\ts_waitcnt vmcnt(0)\n\t"
                   "scratch_load_dword %0, off, off offset:64\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(got), "+v"(data) : "v"(other), "v"(q) : "memory");
    else if (BURST)
      asm volatile("global_load_dword %4, %8, off\n\tglobal_load_dword %5, %9, off\n\tglobal_load_dword %6, %10, off\n\t"
                   "global_load_dword %7, %11, off\n\t"
                   "global_store_dword %3, %1, off\n\tv_mov_b32 %1, %2\n\ts_waitcnt vmcnt(0)\n\t"
                   "global_load_dword %0, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(got), "+v"(data), "+v"(other), "+v"(q), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                   : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]) : "memory");
    else
      asm volatile("global_store_dword %3, %1, off\n\tv_mov_b32 %1, %2\n\ts_waitcnt vmcnt(0)\n\t"
                   "global_load_dword %0, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(got), "+v"(data) : "v"(other), "v"(q) : "memory");
    acc += l0 + l1 + l2 + l3;
    n_bad += got != (0xA0000000u | (uint32_t)it);
    acc += data;
  }
  if (n_bad) atomicAdd(bad, n_bad);
  if (acc == 0x1234567u || priv[(threadIdx.x + 5) & 31] == 99u) bad[1] = acc;
}

template <int MODE, int BURST>
void run(const char* name, const uint32_t* src, uint32_t* sink, uint64_t words, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<MODE, BURST>), dim3(wgs), dim3(256), 0, 0, src, sink, words, d, iters);
    (void)hipDeviceSynchronize();
    uint32_t h = 0;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-40s %5d workgroups, %4d stores per lane: %u stored the overwritten value\n", name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  const uint64_t words = 1ull << 27;
  uint32_t *src, *sink;
  (void)hipMalloc(&src, words * 4);
  (void)hipMemset(src, 0x11, words * 4);
  (void)hipMalloc(&sink, 8192ull * 256 * 16 * 4);
  run<0, 0>("scratch store, no load burst", src, sink, words, 2048, 500);
  run<0, 8>("scratch store behind 4 loads", src, sink, words, 256, 200);
  run<0, 8>("scratch store behind 4 loads", src, sink, words, 2048, 200);
  run<1, 0>("global store, no load burst", src, sink, words, 2048, 500);
  run<1, 8>("global store behind 4 loads", src, sink, words, 2048, 200);
  return 0;
}
