// Private-memory (scratch) integrity under workgroup co-residency: every lane fills a dynamically
// indexed private array with a pattern, works for a while, reads it back.  Prints the number of
// mismatching dwords per launch for a few (workgroups, LDS bytes, private dwords) shapes.
//   hipcc --offload-arch=gfx950 -O3 scratch_check.hip -o scratch_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
  return x;
}

template <int N>
__global__ __launch_bounds__(256) void check(uint32_t* bad, uint32_t* first_bad, int iters, int lds_dwords) {
  extern __shared__ uint32_t lds[];
  for (int t = threadIdx.x; t < lds_dwords; t += 256) lds[t] = t;
  __syncthreads();
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  uint32_t priv[N];
  for (int j = 0; j < N; ++j) priv[(j + id) % N] = mix(id, j);  // dynamic index: the array lives in scratch
  uint32_t acc = id;
  for (int it = 0; it < iters; ++it) {  // keep the wave busy (LDS + ALU), like the walk
    acc = mix(acc, lds[(acc >> 7) % lds_dwords]);
    priv[acc % N] ^= 0u;  // touch scratch in the loop as well
  }
  uint32_t wrong = 0;
  for (int j = 0; j < N; ++j) wrong += priv[(j + id) % N] != mix(id, j);
  if (wrong) {
    atomicAdd(bad, wrong);
    atomicMin(first_bad, id);
  }
  if (acc == 0x12345678u) bad[1] = acc;
}

template <int N>
void run(int wgs, int lds_bytes, int iters) {
  uint32_t *bad, *fb;
  hipMalloc(&bad, 8); hipMalloc(&fb, 4);
  for (int rep = 0; rep < 4; ++rep) {
    uint32_t z[2] = {0, 0}, m = 0xFFFFFFFFu;
    hipMemcpy(bad, z, 8, hipMemcpyHostToDevice); hipMemcpy(fb, &m, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check<N>, dim3(wgs), dim3(256), lds_bytes, 0, bad, fb, iters, lds_bytes / 4);
    hipDeviceSynchronize();
    uint32_t h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&m, fb, 4, hipMemcpyDeviceToHost);
    printf("private %3d dwords/lane, %4d workgroups, %6d B LDS, %5d iters: %u wrong dwords (first bad lane %d)\n", N, wgs, lds_bytes,
           iters, h[0], h[0] ? (int)m : -1);
  }
  hipFree(bad); hipFree(fb);
}

int main() {
  (void)hipFuncSetAttribute((const void*)check<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)check<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run<100>(256, 37280, 2000);   // one workgroup per compute unit
  run<100>(768, 37280, 2000);   // three
  run<100>(2048, 37280, 2000);
  run<64>(768, 37280, 2000);
  run<64>(4096, 8192, 500);
  return 0;
}
