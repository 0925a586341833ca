// A kernel that only occupies the compute units: `wgs` workgroups of 256 threads, each holding
// `lds_bytes` of LDS, spinning for `cycles` clocks.  Loaded through ctypes by tools/coresidency_probe.py.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC occupy.hip -o liboccupy.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void occupy_kernel(long long cycles, int lds_dwords, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  for (int t = threadIdx.x; t < lds_dwords; t += 256) lds[t] = t;
  __syncthreads();
  const long long t0 = clock64();
  uint32_t acc = 0;
  while (clock64() - t0 < cycles) acc += lds[(acc + threadIdx.x) % lds_dwords];
  if (acc == 0x12345u) sink[0] = acc;
}

extern "C" int occupy(void* stream, int wgs, int lds_bytes, long long cycles, uint32_t* sink) {
  (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), lds_bytes, (hipStream_t)stream, cycles, lds_bytes / 4, sink);
  return (int)hipGetLastError();
}
