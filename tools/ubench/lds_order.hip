// Do LDS reads of one wave return in issue order?  hipcc waits for an older ds_read with
// s_waitcnt lgkmcnt(N), N = the number of younger LDS operations.  Here the older read is slow
// (64-way bank conflict) and the younger one fast (broadcast, other widths); the older read's
// destination is preset to a sentinel and copied after lgkmcnt(1).
//   hipcc --offload-arch=gfx950 -O3 lds_order.hip -o lds_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t kSentinel = 0xDEADBEEFu;

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* early, int iters) {
  __shared__ uint32_t lds[9216];  // 36 KiB: four workgroups per compute unit
  for (int t = threadIdx.x; t < 9216; t += 256) lds[t] = 0x01010101u * (t & 63);
  __syncthreads();
  uint32_t n_early = 0, acc = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t slow = (((threadIdx.x & 63) * 32 + (it & 31)) % 9216) * 4;  // one bank for all 64 lanes
    const uint32_t fast = ((it * 7) % 9216) * 4 & ~15u;                        // one address for all lanes
    uint32_t dst = kSentinel, out;
    uint32_t y0, y1, y2, y3;
    if (MODE == 0)
      asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %2, %7\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(dst), "=&v"(out), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(slow), "v"(fast) : "memory");
    else if (MODE == 1)
      asm volatile("ds_read_b32 %0, %6\n\tds_read_u8 %2, %7\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(dst), "=&v"(out), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(slow), "v"(fast) : "memory");
    else
      asm volatile("ds_read_b32 %0, %6\n\tds_read_u8 %2, %7\n\tds_read_u8 %3, %7 offset:1\n\tds_read_b32 %4, %7 offset:4\n\t"
                   "ds_read_u8 %5, %7 offset:9\n\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(dst), "=&v"(out), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(slow), "v"(fast) : "memory");
    n_early += out == kSentinel;
    acc += out + y0;
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
    printf("%-58s %5d workgroups, %5d rounds per lane: %u read before their data arrived\n", name, wgs, iters, h);
  }
  (void)hipFree(d);
}

int main() {
  run<0>("conflicted ds_read_b32, broadcast ds_read_b32, lgkmcnt(1)", 256, 4000);
  run<0>("conflicted ds_read_b32, broadcast ds_read_b32, lgkmcnt(1)", 2048, 4000);
  run<1>("conflicted ds_read_b32, broadcast ds_read_u8, lgkmcnt(1)", 2048, 4000);
  run<2>("conflicted ds_read_b32, four mixed reads, lgkmcnt(4)", 2048, 4000);
  return 0;
}
