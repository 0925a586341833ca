// Store-path microbenchmark for the env-major afterstate matrix [B][36][8] f32: every wave owns 64
// consecutive envs (1152 B apart) and writes all their 32-byte rows, with G adjacent lanes
// cooperating on G*16 contiguous bytes per store instruction:
//   G = 1: a lane writes its own row as two float4 (what a plain per-lane store does)
//   G = 2: lane pairs write one 32-byte row per instruction
//   G = 4 / 8: 64 / 128 contiguous bytes (two / four consecutive rows of one env) per instruction
// No compute, values are lane ids.  build: hipcc --offload-arch=gfx950 -O3 row_store.hip -o row_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kRows = 36, kBlock = 256;

template <int G>
__global__ __launch_bounds__(kBlock) void store_rows(float4* out, int B) {
  const int lane = threadIdx.x & 63;
  const int wave_env0 = (blockIdx.x * kBlock + (threadIdx.x & ~63));
  const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
  // per env: kRows * 2 float4.  One instruction covers 64 / G envs... iterate so that every env gets all rows
  const int sub = lane % G;       // which 16-byte piece of the G*16-byte run
  const int grp = lane / G;       // which env group member (64 / G envs per instruction)
  constexpr int kPieces = kRows * 2;  // float4 pieces per env
  for (int e0 = 0; e0 < 64; e0 += 64 / G) {  // G passes over the wave's envs
    const int env = wave_env0 + e0 + grp;
    if (env >= B) continue;
    float4* dst = out + (size_t)env * kPieces;
    for (int p = 0; p < kPieces; p += G) dst[p + sub] = v;
  }
}

template <int G>
float run(float4* out, int B, int reps) {
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  const int grid = (B + kBlock - 1) / kBlock;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_rows<G>, dim3(grid), dim3(kBlock), 0, 0, out, B);
  hipEventRecord(s, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_rows<G>, dim3(grid), dim3(kBlock), 0, 0, out, B);
  hipEventRecord(e, 0);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  return ms / reps;
}

int main() {
  const int B = 1 << 20;
  const size_t bytes = (size_t)B * kRows * 32;
  float4* out;
  if (hipMalloc(&out, bytes) != hipSuccess) return 1;
  const float t1 = run<1>(out, B, 10), t2 = run<2>(out, B, 10), t4 = run<4>(out, B, 10), t8 = run<8>(out, B, 10);
  printf("matrix %.1f MB, rows of 32 B, envs 1152 B apart\n", bytes / 1e6);
  printf("G=1 (16 B per request): %.3f ms = %.2f TB/s\n", t1, bytes / t1 / 1e9);
  printf("G=2 (32 B per request): %.3f ms = %.2f TB/s\n", t2, bytes / t2 / 1e9);
  printf("G=4 (64 B per request): %.3f ms = %.2f TB/s\n", t4, bytes / t4 / 1e9);
  printf("G=8 (128 B per request): %.3f ms = %.2f TB/s\n", t8, bytes / t8 / 1e9);
  hipFree(out);
  return 0;
}
