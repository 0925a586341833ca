// Do vector-memory LOADS and younger STORES of one wave retire in issue order on gfx950?
// hipcc's wait-count pass (ROCm 7.2) assumes so on this target: it waits for an older load with
// `s_waitcnt vmcnt(N)` where N counts the younger operations, stores included.  Here every lane
// issues a (cache-missing) global load, then a younger store -- to scratch or to global memory --
// then waits with vmcnt(1) and copies the load's destination register, which was preset to a
// sentinel.  A copied sentinel means the store retired first and the wait let the wave through
// before the load's data arrived.
//   hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t kSentinel = 0xDEADBEEFu;

template <int MODE>  // 0: younger scratch store, 1: younger global store, 2: control (vmcnt(0))
__global__ __launch_bounds__(256) void probe(const uint32_t* src, uint32_t* sink, uint64_t words, uint32_t* early, int iters) {
  volatile uint32_t priv[32];  // makes the kernel own a private segment (the asm below stores at its offset 0..)
  priv[threadIdx.x & 31] = 1u;
  uint32_t n_early = 0;
  uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  for (int it = 0; it < iters; ++it) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t* p = src + (x >> 20) % words;  // a different, cold line for every lane and iteration
    uint32_t* q = sink + ((blockIdx.x * 256ull + threadIdx.x) * 16 + (it & 15));
    uint32_t dst = kSentinel, out, val = (uint32_t)it;
    if (MODE == 0)
      asm volatile("global_load_dword %0, %2, off\n\tscratch_store_dword off, %3, off offset:64\n\ts_waitcnt vmcnt(1)\n\t"
                   "v_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                   : "+v"(dst), "=v"(out) : "v"(p), "v"(val), "v"(q) : "memory");
    else if (MODE == 1)
      asm volatile("global_load_dword %0, %2, off\n\tglobal_store_dword %4, %3, off\n\ts_waitcnt vmcnt(1)\n\t"
                   "v_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                   : "+v"(dst), "=v"(out) : "v"(p), "v"(val), "v"(q) : "memory");
    else
      asm volatile("global_load_dword %0, %2, off\n\tscratch_store_dword off, %3, off offset:64\n\ts_waitcnt vmcnt(0)\n\t"
                   "v_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                   : "+v"(dst), "=v"(out) : "v"(p), "v"(val), "v"(q) : "memory");
    n_early += out == kSentinel;
  }
  if (n_early) atomicAdd(early, n_early);
  if (priv[(threadIdx.x + 7) & 31] == 77u) early[1] = 1;
}

template <int MODE>
void run(const char* name, const uint32_t* src, uint32_t* sink, uint64_t words, int wgs, int iters) {
  uint32_t* d;
  (void)hipMalloc(&d, 8);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((probe<MODE>), dim3(wgs), dim3(256), 0, 0, src, sink, words, d, iters);
    (void)hipDeviceSynchronize();
    uint32_t h = 0;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-44s %5d workgroups, %4d loads per lane: %u of %llu loads read before their data arrived\n", name, wgs, iters, h,
           (unsigned long long)wgs * 256ull * iters);
  }
  (void)hipFree(d);
}

int main() {
  const uint64_t words = 1ull << 28;  // 1 GiB of source words (none equals the sentinel)
  uint32_t *src, *sink;
  (void)hipMalloc(&src, words * 4);
  (void)hipMemset(src, 0x11, words * 4);
  (void)hipMalloc(&sink, 8192ull * 256 * 16 * 4);
  run<0>("load, scratch store, vmcnt(1)", src, sink, words, 256, 200);
  run<0>("load, scratch store, vmcnt(1)", src, sink, words, 2048, 200);
  run<1>("load, global store, vmcnt(1)", src, sink, words, 256, 200);
  run<1>("load, global store, vmcnt(1)", src, sink, words, 2048, 200);
  run<2>("load, scratch store, vmcnt(0)  (control)", src, sink, words, 2048, 200);
  return 0;
}
