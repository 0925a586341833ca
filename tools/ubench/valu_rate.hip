// Micro-benchmark: issue rate (cycles per wave64 instruction per SIMD) of the integer VALU ops the
// Tetris kernels lean on.  8 independent dependency chains per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITER 8192

#define KERNEL(NAME, BODY)                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {       \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, \
             a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;   \
    uint32_t b = seed | 1u, c = seed * 9u + 3u;                                          \
    for (int it = 0; it < ITER; ++it) {                                                  \
      BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)            \
    }                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;        \
  }

#define ASM1(op) asm volatile(op " %0, %0, %1" : "+v"(x) : "v"(b));
#define B_add(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_and(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_xor(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_lshl(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define B_lshr(x) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define B_bfe(x) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(x) : "v"(b));
#define B_bcnt(x) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_ffbh(x) asm volatile("v_ffbh_u32 %0, %0" : "+v"(x));
#define B_lshlor(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
#define B_andor(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_bitop3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(x) : "v"(b), "v"(c));
#define B_or3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_add3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_max(x) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_med3(x) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_cndmask(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : );
#define B_cmpcnd(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define B_cmpcnd_sgpr(x) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x) : "v"(b), "v"(c) : "s20", "s21");
#define B_mullo(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_mul24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_mad24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_alignbit(x) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_perm(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_fma(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_pkadd16(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_pkmax16(x) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_mov(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define B_sub(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_not(x) asm volatile("v_not_b32 %0, %0" : "+v"(x));
#define B_ffbl(x) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
#define B_bfm(x) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_bfi(x) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_sad(x) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_mbcnt(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(x) : "v"(b));

KERNEL(add, B_add) KERNEL(sub, B_sub) KERNEL(and, B_and) KERNEL(xor, B_xor) KERNEL(not, B_not) KERNEL(mov, B_mov)
KERNEL(lshl, B_lshl) KERNEL(lshr, B_lshr) KERNEL(bfe, B_bfe) KERNEL(bfm, B_bfm) KERNEL(bfi, B_bfi)
KERNEL(bcnt, B_bcnt) KERNEL(ffbh, B_ffbh) KERNEL(ffbl, B_ffbl) KERNEL(mbcnt, B_mbcnt)
KERNEL(lshlor, B_lshlor) KERNEL(andor, B_andor) KERNEL(bitop3, B_bitop3) KERNEL(or3, B_or3) KERNEL(add3, B_add3)
KERNEL(max, B_max) KERNEL(med3, B_med3) KERNEL(cndmask, B_cndmask) KERNEL(cmpcnd, B_cmpcnd)
KERNEL(cmpcnd_sgpr, B_cmpcnd_sgpr) KERNEL(mullo, B_mullo) KERNEL(mul24, B_mul24) KERNEL(mad24, B_mad24)
KERNEL(alignbit, B_alignbit) KERNEL(perm, B_perm) KERNEL(sad, B_sad) KERNEL(fma, B_fma)
KERNEL(pkadd16, B_pkadd16) KERNEL(pkmax16, B_pkmax16)

// 64-bit shift
__global__ __launch_bounds__(256) void k_lshr64(uint32_t* out, uint32_t seed) {
  uint64_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = (uint64_t)(threadIdx.x + seed) * (2 * i + 3) * 0x100000001ull;
  uint32_t b = (seed & 3) + 1;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
  }
  uint64_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int per_body; };

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  Entry es[] = {
#define E(n) {#n, k_##n, 1},
      E(add) E(sub) E(and) E(xor) E(not) E(mov) E(lshl) E(lshr) E(bfe) E(bfm) E(bfi) E(bcnt) E(ffbh) E(ffbl)
      E(mbcnt) E(lshlor) E(andor) E(bitop3) E(or3) E(add3) E(max) E(med3) E(cndmask)
      {"cmp+cndmask(vcc)", k_cmpcnd, 2}, {"cmp+nop+cndmask(sgpr)", k_cmpcnd_sgpr, 2},
      E(mullo) E(mul24) E(mad24) E(alignbit) E(perm) E(sad) E(fma) E(pkadd16) E(pkmax16) E(lshr64)
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("CUs %d, clock %d MHz (nominal)\n", cus, clk_khz / 1000);
  for (auto& e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 2u + rep);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // per SIMD: 8 waves x ITER x 8 chains x per_body instructions
    double instr = 8.0 * ITER * 8 * e.per_body;
    double cycles = best * 1e-3 * (clk_khz * 1e3);
    printf("%-24s %8.3f ms  %6.2f cycles/instr/SIMD (at nominal clock)\n", e.name, best, cycles / instr);
  }
  return 0;
}
