// Memory-side microbenchmark of the step kernel: the SAME loads and stores per env-step as
// tetris_hip_step (10x20 board, packed planes, all outputs) with next to no arithmetic, in several
// layouts, to find out what the memory floor of one launch over B envs is and which layout
// choices move it.  Every variant reads its state, changes it a little (so nothing is elided and
// written lines differ from read ones) and writes every output.
//
//   V0  current layout: 8 u32 planes [p][B] + u64 meta [B]; outputs obs 2 x float4 per lane,
//       reward i32, action i32, four u8 arrays, one uint4 status slot per wave (read + write)
//   V1  V0 with the four u8 outputs merged into one u32 "flags" word per env
//   V2  V1 with meta folded into the state: 9 u32 planes, no u64 meta
//   V3  V2 with the state tile-major: [B/64][9][64] words (one contiguous 2,304 B record per wave)
//   V4  V3 + obs written transposed (each store instruction covers 1 KiB contiguous)
//   V5  V0 + the per-workgroup table staging of the real kernel (7 KiB + 2.4 KiB from L2 into LDS)
//   V6  plain float4 copy of the same number of bytes (upper bound of this chip for the byte count)
//   round 3 (the layout shipped in round 2 and the candidates for round 3):
//   V7  SHIPPED in round 2: 8 u32 planes tile-major [B/64][8][64] + separate u64 meta [B]; outputs as V0
//   V8  V7 with meta folded into the tile record as planes 8-9 (10 planes tile-major) and the four u8
//       outputs merged into one u32 flags word
//   V9  state = 8 planes tile-major INCLUDING piece / bag / valid mask (20-bit columns + 12 side bits per
//       word: 256 bits at 10x20), flags word, obs 2 x float4 per lane
//   V10 V9 + obs halves exchanged with v_permlane32_swap so that each store instruction covers 1 KiB
//       contiguous (lanes 0-31 low halves, lanes 32-63 high halves of the same 32 rows)
//   V11 V9 + obs transposed through ds_bpermute as V4 (lane L writes half L%2 of row L/2)
//   V12 V10 + per-wave done bitmask word (8 B per wave) and the counters written per wave without
//       reading the slot back (double-buffered payload of the done gather)
// BLK = 256 or 512 threads per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 step_traffic.hip -o step_traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

struct Ptrs {
  uint32_t* planes;   // 9 * B words (V0/V1 use 8)
  uint64_t* meta;
  float4* obs;        // 2 * B
  int32_t* reward;
  int32_t* action;
  uint8_t* done;
  uint8_t* lines;
  uint8_t* nvalid;
  uint8_t* piece;
  uint32_t* flags;
  uint4* status;
  unsigned long long* done_bits;
  const uint4* lut;   // 9.4 KiB of table bytes
  uint32_t B;
};

// V7: V0's traffic with NV extra vector instructions per lane between the loads and the stores (four
// independent chains of xor / add / shift): how memory time and VALU time combine for this access
// pattern -- max() if they overlap, sum if they do not.
template <int V, int BLK, int NV = 0>
__global__ __launch_bounds__(BLK) void traffic(const Ptrs p) {
  constexpr int NP = (V >= 2 && V <= 4) ? 9 : (V == 8 ? 10 : 8);
  constexpr bool TILE = V == 3 || V == 4 || V >= 7;
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i >= p.B) return;
  __shared__ uint4 lds[(V == 5) ? 602 : 1];
  uint32_t w[NP];
  const uint32_t wave = i >> 6, lane = i & 63;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (TILE)
      w[q] = p.planes[(size_t)wave * (NP * 64) + q * 64 + lane];
    else
      w[q] = p.planes[(size_t)q * p.B + i];
  }
  uint64_t meta = 0;
  if (V < 2 || (V >= 5 && V <= 7)) meta = p.meta[i];
  uint4 st = make_uint4(0, 0, 0, 0);
  if (V != 12) st = p.status[wave];
  uint32_t extra = 0;
  if (V == 5) {
    for (int t = threadIdx.x; t < 602; t += BLK) lds[t] = p.lut[t];
    __syncthreads();
    extra = lds[(w[0] ^ threadIdx.x) % 602].x;
  }
  uint32_t acc = (uint32_t)meta ^ (uint32_t)(meta >> 32) ^ extra;
  if (NV > 0) {
    uint32_t c0 = w[0], c1 = w[1], c2 = w[2], c3 = w[3];
#pragma unroll 16
    for (int t = 0; t < NV / 8; ++t) {  // 8 instructions per iteration
      c0 = (c0 ^ c1) + 0x9E3779B9u;
      c1 = (c1 >> 3) ^ c2;
      c2 = (c2 + c3) ^ 0x85EBCA6Bu;
      c3 = (c3 >> 5) + c0;
    }
    acc ^= c0 ^ c1 ^ c2 ^ c3;
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    acc = acc * 0x9E3779B1u + w[q];
    w[q] = w[q] * 5u + 1u;
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (TILE)
      p.planes[(size_t)wave * (NP * 64) + q * 64 + lane] = w[q];
    else
      p.planes[(size_t)q * p.B + i] = w[q];
  }
  if (V < 2 || (V >= 5 && V <= 7)) p.meta[i] = meta + acc;
  float f[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) f[q] = (float)((acc >> (3 * q)) & 31u);
  if (V == 10 || V == 12) {
    // halves exchanged between the two half-waves: lane i < 32 ends up with the low halves of rows i and
    // i + 32, lane 32 + i with their high halves (T21 of the CDNA guide)
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lo[q] = __float_as_uint(f[q]);
      hi[q] = __float_as_uint(f[q + 4]);
      auto r = __builtin_amdgcn_permlane32_swap(lo[q], hi[q], false, false);
      lo[q] = r[0];
      hi[q] = r[1];
    }
    // lanes < 32: lo = own low half (row lane), hi = low half of row lane + 32
    // lanes >= 32: lo = high half of row lane - 32, hi = own high half (row lane)
    const uint32_t row = lane & 31u, part = lane >> 5;
    float4* o = p.obs + (size_t)wave * 128;
    o[2 * row + part] = make_float4(__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(lo[2]), __uint_as_float(lo[3]));
    o[2 * (row + 32) + part] = make_float4(__uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(hi[2]), __uint_as_float(hi[3]));
  } else if (V == 4 || V == 11) {
    // transposed: instruction A covers rows 0..31 of the wave (lane L writes half L%2 of row L/2),
    // instruction B rows 32..63; values travel through ds_bpermute
    float4 a, b;
    {
      const int srcA = (int)(lane >> 1), srcB = 32 + (int)(lane >> 1);
      const bool hi = lane & 1;
      float ga[4], gb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float lo_a = __shfl(f[q], srcA), hi_a = __shfl(f[q + 4], srcA);
        const float lo_b = __shfl(f[q], srcB), hi_b = __shfl(f[q + 4], srcB);
        ga[q] = hi ? hi_a : lo_a;
        gb[q] = hi ? hi_b : lo_b;
      }
      a = make_float4(ga[0], ga[1], ga[2], ga[3]);
      b = make_float4(gb[0], gb[1], gb[2], gb[3]);
    }
    float4* o = p.obs + (size_t)wave * 128;
    o[lane] = a;
    o[64 + lane] = b;
  } else {
    p.obs[2 * (size_t)i] = make_float4(f[0], f[1], f[2], f[3]);
    p.obs[2 * (size_t)i + 1] = make_float4(f[4], f[5], f[6], f[7]);
  }
  p.reward[i] = (int32_t)(acc & 3u) - 1;
  p.action[i] = (int32_t)(acc >> 27);
  if (V == 0 || V == 5 || V == 7) {
    p.done[i] = (uint8_t)(acc & 1u);
    p.lines[i] = (uint8_t)((acc >> 1) & 3u);
    p.nvalid[i] = (uint8_t)((acc >> 3) & 31u);
    p.piece[i] = (uint8_t)((acc >> 8) & 1u);
  } else {
    p.flags[i] = acc & 0x01031F01u;
  }
  const unsigned long long m = __ballot(acc & 1u);
  if (lane == 0) {
    st.y += (unsigned)__popcll(m);
    st.w += 64;
    p.status[wave] = st;
    if (V == 12) p.done_bits[wave] = m;
  }
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

template <int V, int BLK, int NV = 0>
float run(const Ptrs& p, int reps) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  const int grid = (p.B + BLK - 1) / BLK;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((traffic<V, BLK, NV>), dim3(grid), dim3(BLK), 0, 0, p);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(s, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((traffic<V, BLK, NV>), dim3(grid), dim3(BLK), 0, 0, p);
    hipEventRecord(e, 0);
    hipEventSynchronize(e);
    float ms = 0;
    hipEventElapsedTime(&ms, s, e);
    if (ms / reps < best) best = ms / reps;
  }
  return best * 1e3f;
}

static double bytes_of(int V) {
  // per env: state r/w + outputs
  const double out_common = 32 + 4 + 4 + 4 + 0.5;  // obs, reward, action, 4 flag bytes, status r/w
  if (V >= 2 && V <= 4) return 2 * 36 + out_common;
  if (V == 8) return 2 * 40 + out_common;
  if (V >= 9 && V <= 11) return 2 * 32 + out_common;
  if (V == 12) return 2 * 32 + out_common - 0.25 + 0.125;
  return 2 * 32 + 2 * 8 + out_common;
}

int main(int argc, char** argv) {
  for (int pass = 0; pass < 3; ++pass) {
    const uint32_t B = pass == 0 ? (1u << 20) : (pass == 1 ? (1u << 22) : (1u << 16));
    Ptrs p;
    memset(&p, 0, sizeof(p));
    p.B = B;
#define ALLOC(field, bytes)                                               \
  if (hipMalloc((void**)&p.field, (bytes)) != hipSuccess) return 1;       \
  hipMemset(p.field, 1, (bytes));
    ALLOC(planes, (size_t)10 * B * 4)
    ALLOC(done_bits, (size_t)(B / 64) * 8)
    ALLOC(meta, (size_t)B * 8)
    ALLOC(obs, (size_t)B * 32)
    ALLOC(reward, (size_t)B * 4)
    ALLOC(action, (size_t)B * 4)
    ALLOC(done, (size_t)B)
    ALLOC(lines, (size_t)B)
    ALLOC(nvalid, (size_t)B)
    ALLOC(piece, (size_t)B)
    ALLOC(flags, (size_t)B * 4)
    ALLOC(status, (size_t)(B / 64) * 16)
    uint4* lut;
    if (hipMalloc((void**)&lut, 602 * 16) != hipSuccess) return 1;
    hipMemset(lut, 3, 602 * 16);
    p.lut = lut;
    const int reps = pass == 1 ? 60 : 200;
    printf("---- B = %u envs ----\n", B);
#define RUN(V, BLK)                                                                                        \
  {                                                                                                        \
    const float us = run<V, BLK>(p, reps);                                                                 \
    printf("V%d blk %3d: %7.2f us  %6.1f B/env  %.2f TB/s\n", V, BLK, us, bytes_of(V), bytes_of(V) * B / us / 1e6); \
  }
#define RUNW(V, NV)                                                                              \
  {                                                                                              \
    const float us = run<V, 512, NV>(p, reps);                                                   \
    printf("V%d blk 512 + %4d VALU/lane: %7.2f us\n", V, NV, us);                                \
  }
#define RUNV(NV)                                                                                 \
  {                                                                                              \
    const float us = run<0, 512, NV>(p, reps);                                                   \
    printf("V0 blk 512 + %4d VALU/lane: %7.2f us\n", NV, us);                                    \
  }
    RUNV(256) RUNV(512) RUNV(768) RUNV(1024) RUNV(1536) RUNV(2048)
    RUN(0, 256) RUN(0, 512) RUN(1, 256) RUN(1, 512) RUN(2, 256) RUN(2, 512) RUN(3, 256) RUN(3, 512) RUN(4, 256)
    RUN(4, 512) RUN(5, 256) RUN(5, 512)
    RUN(7, 256) RUN(7, 512) RUN(8, 256) RUN(8, 512) RUN(9, 256) RUN(9, 512) RUN(10, 256) RUN(10, 512) RUN(11, 256) RUN(11, 512)
    RUN(12, 256) RUN(12, 512)
    RUNW(7, 768) RUNW(7, 1024) RUNW(10, 512) RUNW(10, 768) RUNW(10, 1024) RUNW(12, 768)
    {
      const size_t n = (size_t)(bytes_of(0) * B / 2 / 16);  // read n float4 + write n float4 = the bytes of V0
      float4 *a, *b;
      if (hipMalloc((void**)&a, n * 16) != hipSuccess || hipMalloc((void**)&b, n * 16) != hipSuccess) return 1;
      hipMemset(a, 1, n * 16);
      hipEvent_t s, e;
      hipEventCreate(&s);
      hipEventCreate(&e);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy4, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
      hipEventRecord(s, 0);
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy4, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
      hipEventRecord(e, 0);
      hipEventSynchronize(e);
      float ms = 0;
      hipEventElapsedTime(&ms, s, e);
      printf("V6 float4 copy of the same bytes: %7.2f us  %.2f TB/s\n", ms / reps * 1e3, 2.0 * n * 16 / (ms / reps) / 1e9);
      hipFree(a);
      hipFree(b);
    }
    hipFree(p.planes); hipFree(p.meta); hipFree(p.obs); hipFree(p.reward); hipFree(p.action); hipFree(p.done);
    hipFree(p.lines); hipFree(p.nvalid); hipFree(p.piece); hipFree(p.flags); hipFree(p.status); hipFree(p.done_bits); hipFree(lut);
  }
  return 0;
}
