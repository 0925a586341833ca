// Multi-dword LDS reads (ds_read2_b32, ds_read_b64, ds_read_b128): is EVERY destination register
// written when s_waitcnt lgkmcnt(0) lets the wave through, also while other waves keep the LDS pipe
// busy with conflicted reads?  The destinations are preset to a sentinel and tested right after the wait.
//   hipcc --offload-arch=gfx950 -O3 lds_multi_dword.hip -o lds_multi_dword
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t kS = 0xDEADBEEFu;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* early, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[9216];
  for (int t = threadIdx.x; t < 9216; t += 256) lds[t] = 0x00010001u * (t & 0x7fff) + 1u;  // never the sentinel
  __syncthreads();
  uint32_t n_early = 0, acc = 0;
  const bool hammer = (threadIdx.x >> 6) & 1;  // every other wave only keeps the LDS pipe busy
  for (int it = 0; it < iters; ++it) {
    if (hammer) {
      acc += lds[((threadIdx.x & 63) * 32 + it * 5 + acc) % 9216];
      continue;
    }
    const uint32_t a = (((threadIdx.x & 63) * 48 + (it % 40)) % 2200) * 16;  // per-lane, 16-byte aligned, bank-conflicted
    if (MODE == 0) {
      uint64_t d = ((uint64_t)kS << 32) | kS;
      asm volatile("ds_read2_b32 %0, %1 offset0:9 offset1:21\n\ts_waitcnt lgkmcnt(0)" : "+v"(d) : "v"(a) : "memory");
      n_early += ((uint32_t)d == kS) + ((uint32_t)(d >> 32) == kS);
      acc += (uint32_t)d;
    } else if (MODE == 1) {
      uint64_t d = ((uint64_t)kS << 32) | kS;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(d) : "v"(a) : "memory");
      n_early += ((uint32_t)d == kS) + ((uint32_t)(d >> 32) == kS);
      acc += (uint32_t)d;
    } else {
      u32x4 d = {kS, kS, kS, kS};
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(d) : "v"(a) : "memory");
      n_early += (d.x == kS) + (d.y == kS) + (d.z == kS) + (d.w == kS);
      acc += d.x;
    }
  }
  if (n_early) atomicAdd(early, n_early);
  if (acc == 0x1234567u) early[1] = acc;
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
    printf("%-14s %5d workgroups (every other wave hammering the LDS), %5d reads per lane: %u destination dwords still unwritten after lgkmcnt(0)\n",
           name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  run<0>("ds_read2_b32", 256, 4000);
  run<0>("ds_read2_b32", 2048, 4000);
  run<1>("ds_read_b64", 2048, 4000);
  run<2>("ds_read_b128", 2048, 4000);
  return 0;
}
