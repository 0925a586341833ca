// tetris_kernels.hip -- gfx950 kernels + the C-ABI of include/tetris_hip.h.
//
// One lane per env; column bitboards in tile-major HBM storage (tet::plane_index: the planes of a
// wavefront's 64 envs back to back, one contiguous record per wave; 256 contiguous bytes (u32) or 512
// (u64) per plane and wave).  No dense contraction anywhere -> no MFMA; the path is integer bit work
// (popcount / clz / shifts / byte-table reads from LDS) over state streamed from HBM, and it is bound by
// vector-instruction issue, not by bytes.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tetris_hip.h"
#include "tetris_core.hpp"
#include "tetris_table.hpp"

namespace {

using tet::SetTable;
using tet::StepCfg;
using tet::build_table;
using tet::n_placements;
using tet::check_desc;

constexpr int kBlock = 256;

// minimum waves per SIMD the step kernel is compiled for (bounds its VGPR budget).  u32 boards: 5
// -- the kernel needs 73 VGPRs (80 allocated: 6 waves per SIMD = three 512-env workgroups per CU),
// no spills.  u64 boards: 103-129 VGPRs (4 / 3 waves per SIMD).  Builds with 7 and 8 waves per SIMD
// (256-env tiles at 71 VGPRs; 512-env tiles squeezed to 64) were timed in round 2 and are not
// faster: profiles/r02_experiments/sweep_compacted_mask_and_occupancy_variants.txt.
#ifndef TET_STEP_WAVES
#define TET_STEP_WAVES 0
#endif
template <typename W, int CR = 12>
constexpr int step_waves() { return TET_STEP_WAVES ? TET_STEP_WAVES : (sizeof(W) == 4 ? 5 : (CR == 10 ? 4 : 3)); }

// envs per workgroup of the step kernel: the feature tables are staged once per workgroup, so a
// larger tile amortises that L2 -> LDS traffic over more envs.  u32 boards: 512 (8 waves = 2 per
// SIMD per workgroup, three workgroups per CU; 320, 640 and 1024 are slower: uneven waves per
// SIMD / one workgroup per CU; 256 is within 1 %).  u64 boards keep 256 (their registers allow
// 3-4 waves per SIMD = three or four 256-env workgroups).
#ifndef TET_STEP_BLOCK
#define TET_STEP_BLOCK 0
#endif
#ifndef TET_LUT10
#define TET_LUT10 1   // 0: always the 12-row-chunk tables (A/B timing)
#endif
template <typename W>
constexpr int step_block() { return TET_STEP_BLOCK ? TET_STEP_BLOCK : (sizeof(W) == 4 ? 512 : 256); }

// Diagnostic build only (-DTET_STAMPS=1, tools/timeline.py): every workgroup of the step kernel
// records when it started, when its loads had landed, when it began to store and when it ended
// (s_memrealtime, 100 MHz) plus where it ran.  Never compiled into the product library.
#ifndef TET_STAMPS
#define TET_STAMPS 0
#endif
#if TET_STAMPS
constexpr int kStampWgs = 16384, kStampWords = 20;  // (slots 6.. are per-tile stamps of looping variants; unused here)
__device__ uint64_t g_stamps[kStampWgs * kStampWords];
#define TET_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < kStampWgs) g_stamps[blockIdx.x * kStampWords + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TET_STAMP(slot) ((void)0)
#endif

// ---- kernels ------------------------------------------------------------------

// feature tables (tools/gen_feature_lut.py): byte tables for hole depth and wells (28 KiB), copied
// to LDS as one block by the kernels that compute features
struct alignas(16) FeatureLut {
  uint8_t bytes[tet::kFeatureLutBytes];
};
__device__ const FeatureLut kFeatureLut = {{
#include "tetris_feature_lut.inc"
}};
static_assert(sizeof(FeatureLut) == tet::kFeatureLutBytes, "layout assumed by col_wells");
// the same tables for 10-row chunks (7 KiB): stepping kernels on boards of up to 20 rows
struct alignas(16) FeatureLut10 {
  uint8_t bytes[tet::kFeatureLut10Bytes];
};
__device__ const FeatureLut10 kFeatureLut10 = {{
#include "tetris_feature_lut10.inc"
}};
// tables of the kernels that walk the afterstates (tet::AfterLut: hole tables, packed wells entries, select tables)
struct alignas(16) AfterLutData {
  uint8_t bytes[tet::kAfterLutBytes];
};
__device__ const AfterLutData kAfterLut = {{
#include "tetris_after_lut.inc"
}};
static_assert(tet::kAfterLutBytes % 16 == 0, "staged as uint4");
template <int CR>
__device__ __forceinline__ const uint4* feature_lut_src() {
  return CR == 10 ? reinterpret_cast<const uint4*>(&kFeatureLut10) : reinterpret_cast<const uint4*>(&kFeatureLut);
}
__device__ __forceinline__ void stage_after_lut(uint8_t* lds) {
  const uint4* src = reinterpret_cast<const uint4*>(&kAfterLut);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int t = threadIdx.x; t < tet::kAfterLutBytes / 16; t += blockDim.x) dst[t] = src[t];
}

// Everything a stepping workgroup keeps in LDS, as ONE object so that the placement table sits at
// LDS address 0: its reads then fit the 8-bit offsets of ds_read2 / the offset field of
// ds_read_b128 and need no per-read address arithmetic.
template <typename W, int C, int BLK, int CR, bool AFTER = false>
struct alignas(16) StepLds {
  SetTable tab;
  alignas(16) uint8_t lut[AFTER ? tet::kAfterLutBytes : tet::LutLayout<CR>::kBytes];  // AFTER: a tet::AfterLut
  W lane_cols[C][BLK];  // per-lane scratch for the runtime-indexed stamp (bank = lane)
};

template <int CR = 12>
__device__ __forceinline__ void stage_hole_lut(uint8_t* lds) {
  const uint4* src = feature_lut_src<CR>();
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int t = threadIdx.x; t < tet::LutLayout<CR>::kBytes / 16; t += blockDim.x) dst[t] = src[t];
}

__device__ __forceinline__ void stage_table(SetTable& lds, const SetTable& arg) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&arg);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&lds);
  for (int t = threadIdx.x; t < (int)(sizeof(SetTable) / 4); t += blockDim.x) dst[t] = src[t];
  __syncthreads();
}

// Per-wave counters without atomics: same-address atomics from 16k waves serialise
// at ~12 ns each (measured: 430 us per launch), so every wave owns one 16-byte slot
// of `status` and lane 0 read-modify-writes it (launches are stream-ordered).
__device__ __forceinline__ unsigned wave_sum(int value_bits, int v) {
  unsigned total = 0;
  for (int b = 0; b < value_bits; ++b) {
    unsigned long long m = __ballot((v >> b) & 1);
    total += (unsigned)__popcll(m) << b;
  }
  return total;
}

struct StepParams {
  void* cols;              // tile-major board words (tet::plane_index)
  uint64_t* meta;
  const int32_t* action;   // NULL: built-in uniform random policy
  int32_t* action_out;     // optional: the action each env played
  const uint8_t* stream;
  int32_t* cursor;
  int64_t stream_len;
  float* obs;
  int32_t* reward;
  uint8_t* done;
  uint8_t* lines;
  uint8_t* n_valid;
  uint8_t* piece_next;
  uint32_t* status;
  // payload of the done/reset gather, written by the step itself when given: one 64-bit done
  // mask per wavefront (bit = lane) and a copy of the wavefront's counter slot as of THIS step -- a
  // consistent snapshot in memory no later step touches, so the exchange can run beside the next steps
  unsigned long long* done_bits;
  uint32_t* status_snapshot;
  uint32_t B;
  uint32_t env_offset;     // global env index of env 0 (mod 2^32)
  // step index from device memory (HIP-graph replays: the kernel arguments of a captured launch are
  // frozen, the step's hash keys must not be): when set, the keys are derived in the kernel from
  // *step_counter + step_rel instead of taken from cfg
  const uint64_t* step_counter;
  uint64_t seed;
  uint32_t step_rel;
  StepCfg cfg;
  SetTable tab;
};

// Access element `byte_off / sizeof(T)` of a wave-uniform array through a 32-bit BYTE offset:
// base stays in SGPRs and the lane offset is one shared VGPR (global_load/store saddr form)
// instead of a 64-bit VGPR address per access.  Callers guarantee byte_off < 2^32.
template <typename T>
__device__ __forceinline__ T ld_off(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(T* base, uint32_t byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// The board of env i goes back to its planes (packed storage: tet::pack_board).
// byte offset of (env i, plane 0) in the tile-major storage: < 2^32 for every batch the C-ABI accepts
template <typename W, int NP>
__device__ __forceinline__ uint32_t record_off(uint32_t i) {
  return ((i >> 6) * (uint32_t)(NP * 64) + (i & 63u)) * (uint32_t)sizeof(W);
}
template <typename W, int C, bool PACK>
__device__ __forceinline__ void store_planes(const StepParams& p, uint32_t i, const W (&col)[C]) {
  constexpr int NP = tet::n_planes(C, PACK);
  W w[NP];
  tet::pack_board<W, C, PACK>(col, w);
  const uint32_t off = record_off<W, NP>(i);
#pragma unroll
  for (int q = 0; q < NP; ++q) st_off(static_cast<W*>(p.cols), off + (uint32_t)(q * 64 * sizeof(W)), w[q]);
}

// Everything one lane reads for one env.
template <typename W, int C>
struct StepInputs {
  W col[C];
  uint64_t meta;
  int action;
  int draw, draw_reset, cursor;
  bool exhausted;  // replay stream: this step could run past the last recorded row
  uint4 status;
};

template <typename W, int C, bool PACK>
__device__ __forceinline__ void load_inputs(const StepParams& p, uint32_t i, StepInputs<W, C>& in) {
  const uint32_t ii = i < p.B ? i : 0;
  constexpr int NP = tet::n_planes(C, PACK);
  W w[NP];
  const uint32_t off = record_off<W, NP>(ii);
#pragma unroll
  for (int q = 0; q < NP; ++q) w[q] = ld_off(static_cast<const W*>(p.cols), off + (uint32_t)(q * 64 * sizeof(W)));
  tet::unpack_board<W, C, PACK>(w, in.col);
  in.meta = ld_off(p.meta, ii * 8u);
  in.action = p.action ? ld_off(p.action, ii * 4u) : -1;
  in.draw = -1;
  in.draw_reset = -1;
  in.cursor = 0;
  in.exhausted = false;
  if (p.stream) {
    in.cursor = p.cursor[ii];
    // a step consumes one row, two when it ends the episode under auto-reset: an env whose stream
    // cannot cover that is counted as invalid and left untouched (never a silent replay of the last row)
    in.exhausted = (int64_t)in.cursor + (p.cfg.auto_reset ? 2 : 1) > p.stream_len || in.cursor < 0;
    int64_t r0 = in.cursor < p.stream_len ? in.cursor : p.stream_len - 1;
    int64_t r1 = in.cursor + 1 < p.stream_len ? in.cursor + 1 : p.stream_len - 1;
    in.draw = p.stream[r0 * p.B + ii];
    in.draw_reset = p.stream[r1 * p.B + ii];
  }
  in.status = make_uint4(0, 0, 0, 0);
  if (p.status) in.status = ld_off(reinterpret_cast<const uint4*>(p.status), (i >> 6) * 16u);  // this wave's slot
}

// One tile of envs per workgroup.  Persistent variants were measured in both rounds and are not
// faster: round 1's grid-stride loops (register prefetch, or 2 / 4 / 8 tiles per workgroup) were
// 3-20 % slower; round 2's per-wavefront loop with LDS-DMA prefetch of the next tile (no load is
// ever waited for, tables staged once per launch) ties at 1 Mi envs and loses elsewhere -- the SIMD
// arbitrates by age, so the oldest waves race ahead and the youngest finish the launch alone;
// evening that out with s_setprio recovers the tie, no more (profiles/r02_experiments/).
// NCH = number of 12-row chunks of the stored board, fixed at compile time for the common
// geometries (2: up to 24 stored rows, e.g. 10x20; 4: up to 48, e.g. 10x40), 0 = decide from R.
// BLK: envs per workgroup (step_block<W>(), or 256 for the small-batch variant of the headline geometry).
template <typename W, int C, int NCH, int CR, int BLK = step_block<W>()>
__global__ __launch_bounds__(BLK, (step_waves<W, CR>())) void step_kernel(const StepParams p) {
  constexpr int kBlock = BLK;  // shadows the file-wide tile size inside this kernel
  __shared__ StepLds<W, C, kBlock, CR> lds;
  SetTable& tab = lds.tab;
  uint8_t* const hole_lut = lds.lut;
  W (&lane_cols)[C][kBlock] = lds.lane_cols;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < p.B;
  // Issue every global load of this lane first (board, meta, action, counters, its share of the
  // two tables) so that one memory latency covers them all; only then fill LDS and barrier.
  constexpr bool PACK = NCH != 0 && !TET_NO_PACK;  // the launchers pick a counted-chunk variant exactly for packed boards
  TET_STAMP(0);
  StepInputs<W, C> in;
  load_inputs<W, C, PACK>(p, i, in);
  {
    constexpr int kLutVecs = tet::LutLayout<CR>::kBytes / 16, kLutPerLane = (kLutVecs + kBlock - 1) / kBlock;
    const uint4* lsrc = feature_lut_src<CR>();
    uint4 lv[kLutPerLane];
#pragma unroll
    for (int q = 0; q < kLutPerLane; ++q) {  // (clamped, not predicated: keeps lv[] in registers)
      const int t = (int)threadIdx.x + q * kBlock;
      lv[q] = lsrc[kLutVecs % kBlock == 0 || t < kLutVecs ? t : kLutVecs - 1];
    }
    constexpr int kTabWords = (int)(sizeof(SetTable) / 4), kTabPerLane = (kTabWords + kBlock - 1) / kBlock;
    const uint32_t* tsrc = reinterpret_cast<const uint32_t*>(&p.tab);
    uint32_t tw[kTabPerLane];
#pragma unroll
    for (int q = 0; q < kTabPerLane; ++q)
      tw[q] = (int)threadIdx.x + q * kBlock < kTabWords ? tsrc[threadIdx.x + q * kBlock] : 0u;
#pragma unroll
    for (int q = 0; q < kLutPerLane; ++q)
      if (!(TET_ABLATE & 128) && (kLutVecs % kBlock == 0 || (int)threadIdx.x + q * kBlock < kLutVecs))
        reinterpret_cast<uint4*>(hole_lut)[threadIdx.x + q * kBlock] = lv[q];
#pragma unroll
    for (int q = 0; q < kTabPerLane; ++q)
      if ((int)threadIdx.x + q * kBlock < kTabWords) reinterpret_cast<uint32_t*>(&tab)[threadIdx.x + q * kBlock] = tw[q];
    __syncthreads();
  }
  TET_STAMP(1);
  StepCfg cfg = p.cfg;
  if (p.step_counter) {  // wave-uniform: scalar registers
    const uint64_t c = *p.step_counter;
    const uint64_t idx = (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                          __builtin_amdgcn_readfirstlane((uint32_t)c)) + p.step_rel;
    cfg.key_step = tet::hash_key(p.seed, idx * 4u + 0u);
    cfg.key_policy = tet::hash_key(p.seed, idx * 4u + 3u);
  }
  int invalid = 0, done = 0, lines = 0, done_flag = 0;
  if (live) {
    tet::StepOut out;
    tet::env_step<W, C, NCH, CR>(in.col, in.meta, in.exhausted ? -1 : in.action, p.action == nullptr && !in.exhausted,
                             tab, hole_lut, &lane_cols[0][threadIdx.x], kBlock, cfg, p.env_offset + i, in.draw,
                             in.draw_reset, out);
    invalid = out.invalid;
    TET_STAMP(2);
    if (p.obs) {
      float4* o4 = reinterpret_cast<float4*>(p.obs);
      st_off(o4, i * 32u, make_float4(out.obs[0], out.obs[1], out.obs[2], out.obs[3]));
      st_off(o4, i * 32u + 16u, make_float4(out.obs[4], out.obs[5], out.obs[6], out.obs[7]));
    }
    if (!invalid) {
      store_planes<W, C, PACK>(p, i, in.col);
      st_off(p.meta, i * 8u, in.meta);
      done = out.done;
      lines = out.lines;
      if (p.stream) p.cursor[i] = in.cursor + 1 + ((out.done && cfg.auto_reset) ? 1 : 0);
    }
    done_flag = out.done;  // what the done array holds (an env that was already over reports done again)
    st_off(p.reward, i * 4u, (int32_t)out.reward);
    st_off(p.done, i, (uint8_t)out.done);
    st_off(p.lines, i, (uint8_t)out.lines);
    st_off(p.n_valid, i, (uint8_t)out.n_valid);
    if (p.piece_next) st_off(p.piece_next, i, (uint8_t)out.piece);
    if (p.action_out) st_off(p.action_out, i * 4u, (int32_t)out.action);
  }
  if (p.status || p.done_bits) {
    const unsigned long long done_mask = __ballot(done_flag != 0);  // == tetris_hip_pack_done_bits(done)
    const unsigned n_inv = wave_sum(1, invalid);
    const unsigned n_done = wave_sum(1, done);
    const unsigned n_lines = wave_sum(3, lines);
    const unsigned n_steps = wave_sum(1, (live && !invalid) ? 1 : 0);
    if ((threadIdx.x & 63) == 0) {
      uint4 v = in.status;
      v.x += n_inv;
      v.y += n_done;
      v.z += n_lines;
      v.w += n_steps;
      if (p.status) st_off(reinterpret_cast<uint4*>(p.status), (i >> 6) * 16u, v);
      if (p.status_snapshot) st_off(reinterpret_cast<uint4*>(p.status_snapshot), (i >> 6) * 16u, v);
      if (p.done_bits) st_off(p.done_bits, (i >> 6) * 8u, done_mask);
    }
  }
#if TET_STAMPS
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < kStampWgs) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
    g_stamps[blockIdx.x * kStampWords + 3] = __builtin_amdgcn_s_memrealtime();
    g_stamps[blockIdx.x * kStampWords + 4] = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID
    g_stamps[blockIdx.x * kStampWords + 5] = __builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID
  }
#endif
}

// Minimum waves per SIMD the kernels built on tet::afterstates_env are compiled for (= their VGPR budget),
// measured per kernel at 1 Mi envs (profiles/r03_experiments/afterstates_family_waves_sweep.txt).  32-bit
// boards: the afterstate matrix kernel is bound by its row stores and runs best unconstrained (2: ~200
// VGPRs, no spills: 0.43 ms against 0.49 / 0.73 at 3 / 4); the greedy policy kernel at 3 (0.255 ms against
// 0.31 / 0.29 at 2 / 4); the kernels that also step (K greedy steps per launch, rollouts) at 4 (128 VGPRs, a
// few spilled dwords: 0.27 ms per step against 0.33 / 0.29, 9.5 ms against 12.8 / 10.6).  64-bit boards
// (10x40): 2 everywhere -- at 3 the 168-VGPR budget spills 600-870 bytes and every kernel is 1.3-2 x slower.
#ifndef TET_AFTER_WAVES
#define TET_AFTER_WAVES 2
#endif
#ifndef TET_GREEDY_WAVES
#define TET_GREEDY_WAVES 3
#endif
#ifndef TET_STEP_GREEDY_WAVES
#define TET_STEP_GREEDY_WAVES 4
#endif
#ifndef TET_AFTER_WAVES64
#define TET_AFTER_WAVES64 2
#endif
// A hardware hazard met on the way (round 3): on gfx950 a 64-bit shift -- v_lshlrev_b64, v_lshrrev_b64,
// v_ashrrev_i64 -- whose 32-bit shift amount sits in the LAST vector register of the wave's allocation (v255
// of 256, v167 of 168, v127 of 128) shifts by (v0 & 63) instead whenever another wave shares the SIMD
// (tools/ubench/shift64_last_vgpr.hip: 0 wrong results at one workgroup per compute unit, 4-5 % at eight;
// v_lshlrev_b32, v_mad_u64_u32, v_lshl_add_u64 are not affected).  The kernels of this family fill their
// register budgets, all multiples of the allocation granule, so hipcc (ROCm 7.2) did place shift amounts there:
// get_best_policy / get_after_states / rollouts results were wrong only where two workgroups shared a compute
// unit.  Nothing in the source can keep the allocator off that register (amdgpu_num_vgpr has no effect on these
// kernels), so tetris_amd/build.py patches the device ASSEMBLY of every translation unit before it is assembled
// (patch_last_vgpr_shifts), and tools/check_last_vgpr.py -- run by the CPU tests on the built library --
// disassembles it and fails if one kernel still has the pattern.  DESIGN.md section 3.2.
template <typename W, int C>
constexpr int after_waves(int want) {
  return sizeof(W) == 4 ? want : (want > TET_AFTER_WAVES64 ? TET_AFTER_WAVES64 : want);
}

struct StepManyParams {
  StepParams one;        // pointers of step 0; per-step outputs advance by B elements per step
  int32_t n_steps;
  int32_t policy;        // 0 uniform random, 1 greedy on w
  uint64_t seed;
  uint64_t step_idx0;
  float w[8];
};

// K consecutive env-steps of every env in ONE launch, for policies that live in the kernel
// (uniform random / greedy linear): the board and meta stay in registers between steps, every
// step's outputs are written to trajectory buffers [K][B]...; bit-identical to K launches of
// step_kernel with step_idx0, step_idx0 + 1, ...  (the per-step keys are re-derived on device).
template <typename W, int C, int NCH, int POLICY, int CR>
__global__ __launch_bounds__(step_block<W>(), (POLICY == 0 ? step_waves<W, CR>() : after_waves<W, C>(TET_STEP_GREEDY_WAVES))) void step_many_kernel(const StepManyParams q) {
  static_assert(POLICY == 0 || CR == 12, "the greedy policy evaluates terminal afterstates too: 12-row chunks");
  constexpr int kBlock = step_block<W>();  // shadows the file-wide tile size inside this kernel
  const StepParams& p = q.one;
  constexpr bool AFTER = POLICY == 1;  // the greedy policy walks the afterstates: tet::AfterLut tables
  __shared__ StepLds<W, C, kBlock, CR, AFTER> lds;
  SetTable& tab = lds.tab;
  uint8_t* const hole_lut = lds.lut;
  W (&lane_cols)[C][kBlock] = lds.lane_cols;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < p.B;
  constexpr bool PACK = NCH != 0 && !TET_NO_PACK;
  StepInputs<W, C> in;
  load_inputs<W, C, PACK>(p, i, in);
  if (AFTER) stage_after_lut(hole_lut);
  else stage_hole_lut<CR>(hole_lut);
  stage_table(tab, p.tab);
  unsigned n_inv = 0, n_done = 0, n_lines = 0, n_steps = 0;
  StepCfg cfg = p.cfg;
#pragma unroll 1
  for (int k = 0; k < q.n_steps; ++k) {
    cfg.key_step = tet::hash_key(q.seed, (q.step_idx0 + (uint64_t)k) * 4u + 0u);
    cfg.key_policy = tet::hash_key(q.seed, (q.step_idx0 + (uint64_t)k) * 4u + 3u);
    int invalid = 0, done = 0, lines = 0;
    if (live) {
      int action = -1;
      bool use_policy = true;
      if (POLICY == 1) {  // greedy: first non-terminal action of maximal fitness
        float best = 0.f;
        int best_row = -1;
        tet::afterstates_env<W, C, NCH>(in.col, in.meta, tab, hole_lut, cfg.R, [&](bool has, int, int, float (&f)[8], int, int row, bool is_valid) {
          if (!has) return;
          if (is_valid) {
            const float v = tet::fitness_of(f, q.w);
            if (best_row < 0 || v > best || (v == best && row < best_row)) {
              best = v;
              best_row = row;
            }
          }
        });
        action = best_row;
        use_policy = false;
      }
      tet::StepOut out;
      tet::env_step<W, C, NCH, CR, AFTER>(in.col, in.meta, action, use_policy, tab, hole_lut, &lane_cols[0][threadIdx.x],
                                          kBlock, cfg, p.env_offset + i, -1, -1, out);
      invalid = out.invalid;
      const uint32_t e = (uint32_t)k * p.B + i;  // element index in the [K][B] trajectory buffers
      if (p.obs) {
        float4* o4 = reinterpret_cast<float4*>(p.obs);
        st_off(o4, e * 32u, make_float4(out.obs[0], out.obs[1], out.obs[2], out.obs[3]));
        st_off(o4, e * 32u + 16u, make_float4(out.obs[4], out.obs[5], out.obs[6], out.obs[7]));
      }
      if (!invalid) {
        done = out.done;
        lines = out.lines;
      }
      st_off(p.reward, e * 4u, (int32_t)out.reward);
      st_off(p.done, e, (uint8_t)out.done);
      st_off(p.lines, e, (uint8_t)out.lines);
      st_off(p.n_valid, e, (uint8_t)out.n_valid);
      if (p.piece_next) st_off(p.piece_next, e, (uint8_t)out.piece);
      if (p.action_out) st_off(p.action_out, e * 4u, (int32_t)out.action);
    }
    n_inv += wave_sum(1, invalid);
    n_done += wave_sum(1, done);
    n_lines += wave_sum(3, lines);
    n_steps += wave_sum(1, (live && !invalid) ? 1 : 0);
  }
  if (live) {
    store_planes<W, C, PACK>(p, i, in.col);
    st_off(p.meta, i * 8u, in.meta);
  }
  if (p.status && (threadIdx.x & 63) == 0) {
    uint4 v = in.status;
    v.x += n_inv;
    v.y += n_done;
    v.z += n_lines;
    v.w += n_steps;
    st_off(reinterpret_cast<uint4*>(p.status), (i >> 6) * 16u, v);
  }
}

struct ResetParams {
  void* cols;
  uint64_t* meta;
  uint32_t* status;        // per-wave counters or NULL: an env whose replay stream is exhausted is counted as invalid
  const uint8_t* reset_mask;
  uint8_t* piece_out;
  uint8_t* n_valid_out;
  const uint8_t* stream;
  int32_t* cursor;
  int64_t stream_len;
  int64_t B;
  int64_t env_offset;
  int32_t init_bag;
  int32_t n_pieces;
  int32_t R;
  uint32_t key;
  SetTable tab;
};

template <typename W, int C, bool PACK>
__global__ __launch_bounds__(kBlock) void reset_kernel(const ResetParams p) {
  __shared__ SetTable tab;
  stage_table(tab, p.tab);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool mine = i < p.B && !(p.reset_mask && !p.reset_mask[i]);
  // replay mode: a reset consumes one stream row; an env whose cursor is at or past the end is left
  // untouched and counted as invalid (as tetris_hip_step does), never continued on the last row
  const int cur = (mine && p.stream) ? p.cursor[i] : 0;
  const bool exhausted = mine && p.stream && (cur < 0 || cur >= p.stream_len);
  if (p.status) {
    const unsigned n_bad = (unsigned)__popcll(__ballot(exhausted));
    if ((threadIdx.x & 63) == 0 && n_bad) p.status[(i >> 6) * 4 + TETRIS_STATUS_INVALID] += n_bad;
  }
  if (!mine || exhausted) return;
  W* cols = static_cast<W*>(p.cols);
#pragma unroll
  for (int q = 0; q < tet::n_planes(C, PACK); ++q) cols[tet::plane_index(i, q, tet::n_planes(C, PACK))] = 0;  // game.py:55-58
  uint32_t bag = p.init_bag ? 0u : tet::meta_bag(p.meta[i]);
  int piece;
  if (p.stream) {
    piece = p.stream[(int64_t)cur * p.B + i];
    p.cursor[i] = cur + 1;
  } else {
    piece = tet::bag_draw(bag, p.n_pieces, tet::hash_env(p.key, (uint32_t)(p.env_offset + i)) >> 16);  // game.py:60
  }
  const uint64_t mask = tab.fullmask[piece];
  p.meta[i] = tet::meta_pack(mask, piece, bag);
  if (p.piece_out) p.piece_out[i] = (uint8_t)piece;
  if (p.n_valid_out) p.n_valid_out[i] = (uint8_t)tet::popc(mask);
}

struct RefreshParams {
  const void* cols;
  uint64_t* meta;
  uint8_t* n_valid_out;
  int64_t B;
  int32_t R;
  SetTable tab;
};

template <typename W, int C, bool PACK>
__global__ __launch_bounds__(kBlock) void refresh_kernel(const RefreshParams p) {
  __shared__ SetTable tab;
  stage_table(tab, p.tab);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.B) return;
  const W* cols = static_cast<const W*>(p.cols);
  W col[C];
  int h[C];
  tet::load_board<W, C, PACK>(cols, p.B, i, col);
  tet::heights_of<W, C>(col, h);
  const uint64_t meta = p.meta[i];
  const int piece = tet::meta_piece(meta);
  const uint64_t mask = tet::valid_mask<W, C>(col, h, tet::piece_entries(tab, piece), tab.fullmask[piece], p.R);
  p.meta[i] = tet::meta_pack(mask, piece, tet::meta_bag(meta));
  if (p.n_valid_out) p.n_valid_out[i] = (uint8_t)tet::popc(mask);
}

struct AfterParams {
  const void* cols;
  const uint64_t* meta;
  float* feats;
  uint8_t* n_valid;
  float* feats_all;
  uint8_t* n_all;
  int64_t B;
  int64_t env_stride;  // in floats: distance between the matrices of consecutive envs
  int64_t row_stride;  // in floats: distance between consecutive feature rows of one env
  int32_t R;
  int32_t a_max;
  int32_t has_direct_by;
  float direct_by[8];
  SetTable tab;
};

// game.py:67-80.  One lane per env walks the static slots in reference order;
// the k-th non-terminal placement lands in feats[i][k].


// Feature rows are 32 bytes and the rows of different envs are far apart, so a plain store of one
// row per lane is 64 separate 16-byte write requests per instruction -- the kernel was bound by
// the L2 write-request rate (75 M requests per launch), not by bytes.  Lanes therefore pair up:
// in two steps the even/odd lane of a pair write the two 16-byte halves of ONE row (first the even
// lane's row, then the odd lane's), so every request carries a whole 32-byte row.  The halves
// change lanes through DPP (quad_perm 1,0,3,2: neighbours swap) -- round 1 sent them through LDS,
// whose write -> wave barrier -> read round trip sat on the critical path of a kernel that is bound
// by latency.  All lanes of the wave must call this together (afterstates_env's emit is
// wave-uniform).
__device__ __forceinline__ uint32_t swap_neighbour(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void store_row_paired(float* base, bool has, uint32_t at4, const float (&f)[8]) {
  const bool odd = threadIdx.x & 1u;
  const uint32_t at_mine = has ? at4 : ~0u;
  const uint32_t at_other = swap_neighbour(at_mine);
  float r[4];  // the partner's half this lane stores: even lanes get the odd row's low half, odd lanes the even row's high half
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = __uint_as_float(swap_neighbour(__float_as_uint(odd ? f[q] : f[q + 4])));
  const uint32_t d_even = odd ? at_other : at_mine, d_odd = odd ? at_mine : at_other;  // rows of the pair's even / odd lane
  const float4 v_even = odd ? make_float4(r[0], r[1], r[2], r[3]) : make_float4(f[0], f[1], f[2], f[3]);
  const float4 v_odd = odd ? make_float4(f[4], f[5], f[6], f[7]) : make_float4(r[0], r[1], r[2], r[3]);
  float4* out = reinterpret_cast<float4*>(base);
  const uint32_t part = odd ? 1u : 0u;
  if (d_even != ~0u) out[(size_t)d_even + part] = v_even;
  if (d_odd != ~0u) out[(size_t)d_odd + part] = v_odd;
}

template <typename W, int C, int NCH>
__global__ __launch_bounds__(kBlock, (after_waves<W, C>(TET_AFTER_WAVES))) void afterstates_kernel(const AfterParams p) {
  __shared__ SetTable tab;
  __shared__ __attribute__((aligned(16))) uint8_t hole_lut[tet::kAfterLutBytes];
  stage_after_lut(hole_lut);
  stage_table(tab, p.tab);
  const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i0 < p.B;
  const int64_t i = live ? i0 : p.B - 1;  // lanes past the end keep pace on the last env and store nothing
  const W* cols = static_cast<const W*>(p.cols);
  W col[C];
  tet::load_board<W, C, (NCH != 0 && !TET_NO_PACK)>(cols, p.B, i, col);
  const uint64_t meta = p.meta[i];
  const int piece = tet::meta_piece(meta);
  const uint64_t full = tab.fullmask[piece];
  const uint64_t valid = tet::meta_mask(meta) & full;  // non-terminal slots (kept fresh by step/reset/refresh)
  // env-major ([B][a_max][8]: row_stride 8) keeps a wave's 36 rows x 64 envs inside one 72 KiB
  // span; action-major ([a_max][B][8]: env_stride 8) walks 36 regions 32 MiB apart and measured
  // 1.4x slower (TLB reach), so env-major is the default upstream.  Offsets are in float4 units
  // (the C-ABI checks that the whole matrix stays below 2^32 of them).
  const uint32_t es4 = (uint32_t)(p.env_stride / 4), rs4 = (uint32_t)(p.row_stride / 4);
  const uint32_t env4 = (uint32_t)i * es4;
  const int nv = tet::popc(valid), na = tet::popc(full);
  float sink = 0.f;
  tet::afterstates_env<W, C, NCH>(col, meta, tab, hole_lut, p.R, [&](bool has, int sk, int sc, float (&f)[8], int row_all, int row_valid, bool is_valid) {
    has = has && live;
    if (TET_ABLATE & 64) {  // timing experiment: no feature stores
      if (has) sink += f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + (float)(sk + sc);
      return;
    }
    if (p.has_direct_by) {
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] *= p.direct_by[q];
    }
    // the row of a placement follows the reference's enumeration order
    if (p.feats_all) store_row_paired(p.feats_all, has, env4 + (uint32_t)row_all * rs4, f);
    store_row_paired(p.feats, has && is_valid, env4 + (uint32_t)row_valid * rs4, f);  // game.py:69
  });
  if (TET_ABLATE & 64) {
    if (live) p.feats[i * p.env_stride] = sink;
    return;
  }
  // zero the rows past the last placement: the lanes of a pair write the two halves of one row per
  // instruction as above, but there is nothing to exchange -- only the partner's base, count and liveness,
  // fetched once
  {
    const bool odd = threadIdx.x & 1u;
    const uint32_t part = odd ? 1u : 0u;
    const uint32_t env4_o = swap_neighbour(env4);
    const int nv_o = (int)swap_neighbour((uint32_t)(live ? nv : 0x7FFFFFFF)), na_o = (int)swap_neighbour((uint32_t)(live ? na : 0x7FFFFFFF));
    const int nv_m = live ? nv : 0x7FFFFFFF, na_m = live ? na : 0x7FFFFFFF;
    // rows of the pair's even / odd lane
    const uint32_t base_e = odd ? env4_o : env4, base_o = odd ? env4 : env4_o;
    const int nv_e = odd ? nv_o : nv_m, nv_od = odd ? nv_m : nv_o, na_e = odd ? na_o : na_m, na_od = odd ? na_m : na_o;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* out = reinterpret_cast<float4*>(p.feats);
    float4* out_all = reinterpret_cast<float4*>(p.feats_all);
    for (int k = 0; k < p.a_max; ++k) {  // (wave-uniform loop)
      const uint32_t r4 = (uint32_t)k * rs4 + part;
      if (k >= nv_e) out[(size_t)base_e + r4] = z4;
      if (k >= nv_od) out[(size_t)base_o + r4] = z4;
      if (out_all) {
        if (k >= na_e) out_all[(size_t)base_e + r4] = z4;
        if (k >= na_od) out_all[(size_t)base_o + r4] = z4;
      }
    }
  }
  if (live) {
    p.n_valid[i] = (uint8_t)nv;
    if (p.n_all) p.n_all[i] = (uint8_t)na;
  }
}


struct GreedyParams {
  const void* cols;
  const uint64_t* meta;
  int32_t* best_action;
  float* best_value;
  float* fitness_all;
  int64_t B;
  int32_t R;
  int32_t a_max;
  float w[8];
  SetTable tab;
};

// Tetris.get_best_policy / fitness (game.py:102-120) for every env: the fitness of every
// placement (raw order, terminal included, like game.py:103) and the best NON-terminal action.
// The [B][a_max][8] feature matrix never touches HBM.
template <typename W, int C, int NCH>
__global__ __launch_bounds__(kBlock, (after_waves<W, C>(TET_GREEDY_WAVES))) void greedy_kernel(const GreedyParams p) {
  __shared__ SetTable tab;
  __shared__ __attribute__((aligned(16))) uint8_t hole_lut[tet::kAfterLutBytes];
  stage_after_lut(hole_lut);
  stage_table(tab, p.tab);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.B) return;
  const W* cols = static_cast<const W*>(p.cols);
  W col[C];
  tet::load_board<W, C, (NCH != 0 && !TET_NO_PACK)>(cols, p.B, i, col);
  const uint64_t meta = p.meta[i];
  const int piece = tet::meta_piece(meta);
  const uint64_t full = tab.fullmask[piece];
  const uint64_t valid = tet::meta_mask(meta) & full;
  float* fall = p.fitness_all ? p.fitness_all + i * (int64_t)p.a_max : nullptr;
  float best = 0.f;
  int best_row = -1;
  tet::afterstates_env<W, C, NCH>(col, meta, tab, hole_lut, p.R, [&](bool has, int, int, float (&f)[8], int row_all, int row, bool is_valid) {
    if (!has) return;
    const float v = tet::fitness_of(f, p.w);
    if (fall && !(TET_ABLATE & 2048)) fall[row_all] = v;
    if (is_valid) {
      if (best_row < 0 || v > best || (v == best && row < best_row)) {
        best = v;
        best_row = row;
      }
    }
  });
  if (fall)
    for (int k = tet::popc(full); k < p.a_max; ++k) fall[k] = 0.f;
  p.best_action[i] = best_row;
  if (p.best_value) p.best_value[i] = best;
}

struct RolloutParams {
  const void* cols;
  const uint64_t* meta;
  double* returns;  // [B][a_max]
  int64_t B;
  int64_t env_offset;
  int32_t R;
  int32_t a_max;
  int32_t n_pieces;
  int32_t length;
  int32_t n;
  int32_t policy;
  uint32_t key;
  const uint8_t* pieces;  // NULL, or uint8 [B][a_max][n][length]: the piece every step of every rollout draws
  float w[8];
  SetTable tab;
};

// Tetris.perform_rollouts (game.py:150-160) as a fan-out: one lane per (env, first action) runs its
// n rollouts back to back with the board in registers; nothing but the mean returns is written.
// 512 lanes per workgroup: with the 35 KiB of afterstate tables two workgroups = 16 waves fit a CU
constexpr int kRolloutBlock = 512;
template <typename W, int C, int NCH>
__global__ __launch_bounds__(kRolloutBlock, (after_waves<W, C>(TET_STEP_GREEDY_WAVES))) void rollouts_kernel(const RolloutParams p) {
  constexpr int kBlock = kRolloutBlock;  // shadows the file-wide tile size inside this kernel
  __shared__ StepLds<W, C, kBlock, 12, true> lds;
  SetTable& tab = lds.tab;
  uint8_t* const hole_lut = lds.lut;
  W (&lane_cols)[C][kBlock] = lds.lane_cols;
  stage_after_lut(hole_lut);
  stage_table(tab, p.tab);
  const int64_t id = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (id >= p.B * p.a_max) return;
  const int64_t i = id / p.a_max;
  const int a0 = (int)(id - i * p.a_max);
  const W* cols = static_cast<const W*>(p.cols);
  W col[C];
  tet::load_board<W, C, (NCH != 0 && !TET_NO_PACK)>(cols, p.B, i, col);
  const uint64_t meta = p.meta[i];
  const int nv = tet::popc(tet::meta_mask(meta));
  double mean = __longlong_as_double(0x7FF8000000000000ll);  // NaN: not an action of this env
  if (a0 < nv) {
    int sum = 0;
    for (int r = 0; r < p.n; ++r) {
      const uint64_t uid = ((uint64_t)(p.env_offset + i) * (uint64_t)p.a_max + (uint64_t)a0) * (uint64_t)p.n + r;
      const uint32_t key0 = tet::mix32(p.key ^ ((uint32_t)(uid >> 32) * 0x9E3779B1u));
      const uint64_t fed = ((uint64_t)(i * p.a_max + a0) * (uint64_t)p.n + (uint64_t)r) * (uint64_t)p.length;
      sum += tet::rollout_env<W, C, NCH>(col, meta, a0, p.length, p.policy, p.w, tab, hole_lut,
                                    &lane_cols[0][threadIdx.x], kBlock, p.R, p.n_pieces, key0, (uint32_t)uid,
                                    p.pieces ? p.pieces + fed : nullptr);
    }
    mean = (double)sum / (double)p.n;
  }
  p.returns[id] = mean;
}

__global__ void counter_add_kernel(uint64_t* counter, uint64_t n) { *counter += n; }

// done flags -> bitmask (bit i % 8 of byte i / 8): one ballot per wavefront, eight bytes per store.
// The payload of the done/reset gather, the path's only collective.
__global__ __launch_bounds__(kBlock) void pack_done_bits_kernel(const uint8_t* __restrict__ done,
                                                                unsigned long long* __restrict__ bits, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long m = __ballot(i < B && done[i] != 0);
  if ((threadIdx.x & 63) == 0 && (i & ~(int64_t)63) < B) bits[i >> 6] = m;
}

__global__ __launch_bounds__(kBlock) void policy_random_kernel(const uint8_t* __restrict__ n_valid,
                                                               int32_t* __restrict__ action, uint32_t key,
                                                               int64_t env_offset, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  action[i] = tet::policy_random(key, (uint32_t)(env_offset + i), n_valid[i]);
}

// The reference's piece sampler on NumPy's legacy global stream, reproduced on the device
// (tetromino.py:12-22 on top of np.random.seed / np.random.permutation): env i is
// seeded like `np.random.seed(seeds[i])` right before `game.Tetris(...)` is constructed, and row t of
// the stream is the list index its sampler hands out at its t-th call.  MT19937 (Matsumoto &
// Nishimura) with NumPy's init_genrand seeding; permutation(n) = Fisher-Yates from the top with
// masked rejection sampling on raw 32-bit outputs.  One lane per env, the 2.5 KB generator state in
// private memory: this is a set-up kernel for exact replays of seeded reference games (small to
// moderate B), not a hot path -- the counter-based bag in `meta` is the one for large batches.
__global__ __launch_bounds__(64) void numpy_bag_stream_kernel(const uint32_t* __restrict__ seeds, int n_pieces,
                                                             int64_t L, uint8_t* __restrict__ stream, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= B) return;
  uint32_t mt[624];
  mt[0] = seeds[i];
  for (int k = 1; k < 624; ++k) mt[k] = 1812433253U * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
  int pos = 624;
  auto next_u32 = [&]() -> uint32_t {
    if (pos >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000U) | (mt[k + 1 < 624 ? k + 1 : 0] & 0x7fffffffU);
        mt[k] = mt[k + 397 < 624 ? k + 397 : k + 397 - 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
      }
      pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= y >> 18;
    return y;
  };
  uint8_t bag[TETRIS_MAX_PIECES];
  int left = 0;
  for (int64_t t = 0; t < L; ++t) {
    if (left == 0) {  // tetromino.py:15,18-19: a fresh np.random.permutation(n)
      for (int k = 0; k < n_pieces; ++k) bag[k] = (uint8_t)k;
      for (int k = n_pieces - 1; k >= 1; --k) {
        uint32_t mask = (uint32_t)k;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        uint32_t v;
        do {
          v = next_u32() & mask;
        } while (v > (uint32_t)k);
        const uint8_t tmp = bag[k];
        bag[k] = bag[v];
        bag[v] = tmp;
      }
      left = n_pieces;
    }
    stream[t * B + i] = bag[n_pieces - left];  // tetromino.py:20-21: element 0, then delete it
    --left;
  }
}

template <typename W>
__global__ __launch_bounds__(kBlock) void decode_kernel(const W* __restrict__ cols, int8_t* __restrict__ cells,
                                                        int32_t* __restrict__ heights, int C, int rows, int64_t B,
                                                        bool packed) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  for (int c = 0; c < C; ++c) {
    const W x = tet::load_column_rt<W>(cols, B, i, c, C, packed);
    if (heights) heights[i * C + c] = tet::bitlen(x);
    if (cells)
      for (int r = 0; r < rows; ++r) cells[(i * rows + r) * C + c] = (int8_t)((x >> r) & 1);
  }
}

template <typename W>
__global__ __launch_bounds__(kBlock) void encode_kernel(const int8_t* __restrict__ cells, W* __restrict__ cols,
                                                        int C, int rows, int64_t B, bool packed) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  W col[tet::kMaxCols];
  for (int c = 0; c < C; ++c) {
    W x = 0;
    for (int r = 0; r < rows; ++r) x |= (W)(cells[(i * rows + r) * C + c] != 0) << r;
    col[c] = x;
  }
  tet::store_columns_rt<W>(cols, B, i, col, C, packed);
}

// ---- dispatch on (word, C) -------------------------------------------------------

inline dim3 grid_for(int64_t B) { return dim3((unsigned)((B + kBlock - 1) / kBlock)); }
inline dim3 rollout_grid(int64_t n) { return dim3((unsigned)((n + kRolloutBlock - 1) / kRolloutBlock)); }
constexpr uint32_t kSmallBatch = 65536;
template <typename W>
inline dim3 step_grid(int64_t B) { return dim3((unsigned)((B + step_block<W>() - 1) / step_block<W>())); }

// ---- translation units ------------------------------------------------------------------------
// The kernels are templates on the column count, and one hipcc process compiling all of them takes
// minutes, so the in-tree build (tetris_amd/build.py) compiles this file once per column count
// (-DTET_PART=<C>: the kernels of that count behind tetris_part_<C>) plus once as the main unit
// (-DTET_SPLIT_MAIN: the C-ABI, which reaches the column-specific launches through those entries),
// in parallel, and links the objects.  Compiled alone with neither macro the file is a complete
// single-unit library (what the tools' experiment builds do).
template <template <typename, int> class Launcher> struct LaunchId;
// Kernel variant of a geometry.  Packed boards (tet::board_packed: stored rows within three
// quarters of the word) always run the variants with a compile-time chunk count -- NCH = 2 on u32,
// 4 on u64 -- and those variants read / write the packed planes; everything else is NCH = 0 on
// one plane per column.
template <typename W>
inline bool packed_geometry(int R) { return R + 4 <= 6 * (int)sizeof(W); }  // == tet::board_packed (TET_NO_PACK builds keep the variant choice)
template <typename W>
constexpr int packed_chunks() { return sizeof(W) == 4 ? 2 : 4; }

template <typename W, int C>
struct LaunchStep {
  static void run(const StepParams& p, hipStream_t s) {
    // a step only evaluates the features of a NON-terminal board (cells below row R): up to R = 20
    // (u32) two 10-row chunks cover it, up to R = 40 (u64) four, and the tables are the 7 KiB
    // set; otherwise 12-row chunks
    constexpr int N = packed_chunks<W>();
    if (!packed_geometry<W>(p.cfg.R))
      hipLaunchKernelGGL((step_kernel<W, C, 0, 12>), step_grid<W>(p.B), dim3(step_block<W>()), 0, s, p);
    else if (p.cfg.R <= 10 * N && TET_LUT10) {
      // small batches (up to one wave per SIMD): 256-env tiles put a workgroup on every CU at 65,536 envs
      // and shorten its table staging: 6.75 us against 7.24 us per step there; 5 % slower at 131,072 envs, 1 % at
      // 1 Mi envs and beyond (profiles/r03_experiments/small_batch_variants.txt)
      if (step_block<W>() > 256 && p.B <= kSmallBatch)
        hipLaunchKernelGGL((step_kernel<W, C, N, 10, 256>), dim3((p.B + 255) / 256), dim3(256), 0, s, p);
      else
        hipLaunchKernelGGL((step_kernel<W, C, N, 10>), step_grid<W>(p.B), dim3(step_block<W>()), 0, s, p);
    } else
      hipLaunchKernelGGL((step_kernel<W, C, N, 12>), step_grid<W>(p.B), dim3(step_block<W>()), 0, s, p);
  }
};
template <typename W, int C>
struct LaunchStepMany {
  static void run(const StepManyParams& q, hipStream_t s) {
    constexpr int N = packed_chunks<W>();
    const int R = q.one.cfg.R;
    const bool packed = packed_geometry<W>(R);
    if (q.policy == 1 && packed)
      hipLaunchKernelGGL((step_many_kernel<W, C, N, 1, 12>), step_grid<W>(q.one.B), dim3(step_block<W>()), 0, s, q);
    else if (q.policy == 1)
      hipLaunchKernelGGL((step_many_kernel<W, C, 0, 1, 12>), step_grid<W>(q.one.B), dim3(step_block<W>()), 0, s, q);
    else if (!packed)
      hipLaunchKernelGGL((step_many_kernel<W, C, 0, 0, 12>), step_grid<W>(q.one.B), dim3(step_block<W>()), 0, s, q);
    else if (R <= 10 * N && TET_LUT10)
      hipLaunchKernelGGL((step_many_kernel<W, C, N, 0, 10>), step_grid<W>(q.one.B), dim3(step_block<W>()), 0, s, q);
    else
      hipLaunchKernelGGL((step_many_kernel<W, C, N, 0, 12>), step_grid<W>(q.one.B), dim3(step_block<W>()), 0, s, q);
  }
};
template <typename W, int C>
struct LaunchReset {
  static void run(const ResetParams& p, hipStream_t s) {
    if (tet::board_packed((int)sizeof(W), p.R))
      hipLaunchKernelGGL((reset_kernel<W, C, true>), grid_for(p.B), dim3(kBlock), 0, s, p);
    else
      hipLaunchKernelGGL((reset_kernel<W, C, false>), grid_for(p.B), dim3(kBlock), 0, s, p);
  }
};
template <typename W, int C>
struct LaunchRefresh {
  static void run(const RefreshParams& p, hipStream_t s) {
    if (tet::board_packed((int)sizeof(W), p.R))
      hipLaunchKernelGGL((refresh_kernel<W, C, true>), grid_for(p.B), dim3(kBlock), 0, s, p);
    else
      hipLaunchKernelGGL((refresh_kernel<W, C, false>), grid_for(p.B), dim3(kBlock), 0, s, p);
  }
};
template <typename W, int C>
struct LaunchRollouts {
  static void run(const RolloutParams& p, hipStream_t s) {
    if (packed_geometry<W>(p.R))
      hipLaunchKernelGGL((rollouts_kernel<W, C, packed_chunks<W>()>), rollout_grid(p.B * p.a_max), dim3(kRolloutBlock), 0, s, p);
    else
      hipLaunchKernelGGL((rollouts_kernel<W, C, 0>), rollout_grid(p.B * p.a_max), dim3(kRolloutBlock), 0, s, p);
  }
};
template <typename W, int C>
struct LaunchGreedy {
  static void run(const GreedyParams& p, hipStream_t s) {
    if (packed_geometry<W>(p.R))
      hipLaunchKernelGGL((greedy_kernel<W, C, packed_chunks<W>()>), grid_for(p.B), dim3(kBlock), 0, s, p);
    else
      hipLaunchKernelGGL((greedy_kernel<W, C, 0>), grid_for(p.B), dim3(kBlock), 0, s, p);
  }
};
template <typename W, int C>
struct LaunchAfter {
  static void run(const AfterParams& p, hipStream_t s) {
    if (packed_geometry<W>(p.R))
      hipLaunchKernelGGL((afterstates_kernel<W, C, packed_chunks<W>()>), grid_for(p.B), dim3(kBlock), 0, s, p);
    else
      hipLaunchKernelGGL((afterstates_kernel<W, C, 0>), grid_for(p.B), dim3(kBlock), 0, s, p);
  }
};


template <> struct LaunchId<LaunchStep> { static constexpr int value = 0; };
template <> struct LaunchId<LaunchStepMany> { static constexpr int value = 1; };
template <> struct LaunchId<LaunchReset> { static constexpr int value = 2; };
template <> struct LaunchId<LaunchRefresh> { static constexpr int value = 3; };
template <> struct LaunchId<LaunchRollouts> { static constexpr int value = 4; };
template <> struct LaunchId<LaunchGreedy> { static constexpr int value = 5; };
template <> struct LaunchId<LaunchAfter> { static constexpr int value = 6; };

template <template <typename, int> class Launcher, int CC, typename P>
inline void launch_words(int word_bytes, const void* p, hipStream_t s) {
  if (word_bytes == 4) Launcher<uint32_t, CC>::run(*static_cast<const P*>(p), s);
  else Launcher<uint64_t, CC>::run(*static_cast<const P*>(p), s);
}
// every launch of column count CC (instantiates all its kernels)
template <int CC>
int launch_part(int which, int word_bytes, const void* p, hipStream_t s) {
  switch (which) {
    case 0: launch_words<LaunchStep, CC, StepParams>(word_bytes, p, s); break;
    case 1: launch_words<LaunchStepMany, CC, StepManyParams>(word_bytes, p, s); break;
    case 2: launch_words<LaunchReset, CC, ResetParams>(word_bytes, p, s); break;
    case 3: launch_words<LaunchRefresh, CC, RefreshParams>(word_bytes, p, s); break;
    case 4: launch_words<LaunchRollouts, CC, RolloutParams>(word_bytes, p, s); break;
    case 5: launch_words<LaunchGreedy, CC, GreedyParams>(word_bytes, p, s); break;
    case 6: launch_words<LaunchAfter, CC, AfterParams>(word_bytes, p, s); break;
    default: return TETRIS_E_DESC;
  }
  return (int)hipGetLastError();
}

}  // namespace

#if defined(TET_PART)
#define TET_PART_NAME2(c) tetris_part_##c
#define TET_PART_NAME(c) TET_PART_NAME2(c)
extern "C" __attribute__((visibility("hidden"))) int TET_PART_NAME(TET_PART)(int which, int word_bytes, const void* p,
                                                                            hipStream_t s) {
  return launch_part<TET_PART>(which, word_bytes, p, s);
}
#elif defined(TET_SPLIT_MAIN)
#define X(CC) extern "C" __attribute__((visibility("hidden"))) int tetris_part_##CC(int, int, const void*, hipStream_t);
TET_COLUMNS(X)
#undef X
#endif

namespace {

template <template <typename, int> class Launcher, typename P>
int dispatch(const TetrisDesc* d, const P& p, hipStream_t s) {
  switch (d->num_columns) {
#if defined(TET_SPLIT_MAIN)
#define X(CC) case CC: return tetris_part_##CC(LaunchId<Launcher>::value, d->word_bytes, &p, s);
#else
#define X(CC) case CC: return launch_part<CC>(LaunchId<Launcher>::value, d->word_bytes, &p, s);
#endif
    TET_COLUMNS(X)
#undef X
    default:
      return TETRIS_E_COLUMNS;
  }
}

}  // namespace

#if !defined(TET_PART)
// ---- C-ABI ---------------------------------------------------------------------------
extern "C" {

int tetris_hip_version(void) { return TETRIS_HIP_ABI_VERSION; }

#if TET_STAMPS
int tetris_debug_read_stamps(uint64_t* dst, int n_wgs) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(uint64_t) * kStampWords * (size_t)n_wgs, 0,
                                  hipMemcpyDeviceToHost);
}
#endif

const char* tetris_hip_error_string(int code) {
  const char* own = tet::error_text(code);
  return own ? own : (code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error");
}

int tetris_hip_supported_columns(int32_t* out, int cap) {
  int n = 0;
#define X(CC) if (out && n < cap) out[n] = CC; ++n;
  TET_COLUMNS(X)
#undef X
  return n;
}

int tetris_hip_n_planes(const TetrisDesc* desc) {
  const int rc = check_desc(desc);
  if (rc) return rc;
  return tet::n_planes(desc->num_columns, tet::board_packed(desc->word_bytes, desc->num_rows));
}

int64_t tetris_hip_board_words(const TetrisDesc* desc, int64_t B) {
  const int rc = check_desc(desc);
  if (rc) return rc;
  if (B <= 0) return TETRIS_E_BATCH;
  return tet::board_words(B, tet::n_planes(desc->num_columns, tet::board_packed(desc->word_bytes, desc->num_rows)));
}

int64_t tetris_hip_status_words(int64_t B) {
  if (B <= 0) return 0;
  // one 16-byte slot per wavefront, rounded up to the largest tile a stepping kernel may use
  return 4 * (((B + 1023) / 1024) * 16);
}

int tetris_hip_n_placements(int32_t catalogue_id, int32_t num_columns) {
  if (catalogue_id < 0 || catalogue_id >= TETRIS_N_CATALOGUE) return TETRIS_E_PIECES;
  return n_placements(catalogue_id, num_columns);
}

int tetris_hip_desc_init(TetrisDesc* desc, int32_t num_columns, int32_t num_rows, const int32_t* piece_ids,
                         int32_t n_pieces, const float* direct_by) {
  return tet::desc_init(desc, num_columns, num_rows, piece_ids, n_pieces, direct_by);
}

int tetris_hip_reset(const TetrisDesc* desc, void* cols, uint64_t* meta, const uint8_t* reset_mask,
                     uint8_t* piece_out, uint8_t* n_valid_out, const uint8_t* stream, int32_t* cursor,
                     int64_t stream_len, uint32_t* status, int32_t init_bag, uint64_t seed, uint64_t step_idx,
                     int64_t env_offset, int64_t B, void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  if (stream && (!cursor || stream_len <= 0)) return TETRIS_E_STREAM;
  ResetParams p{};
  p.cols = cols;
  p.meta = meta;
  p.status = status;
  p.reset_mask = reset_mask;
  p.piece_out = piece_out;
  p.n_valid_out = n_valid_out;
  p.stream = stream;
  p.cursor = cursor;
  p.stream_len = stream_len;
  p.B = B;
  p.env_offset = env_offset;
  p.init_bag = init_bag;
  p.n_pieces = desc->n_pieces;
  p.R = desc->num_rows;
  p.key = tet::hash_key(seed, step_idx * 4u + 2u);
  build_table(desc, &p.tab);
  return dispatch<LaunchReset>(desc, p, (hipStream_t)hip_stream);
}

static int fill_step_params(StepParams& p, const TetrisDesc* desc, void* cols, uint64_t* meta, const int32_t* action,
                            int32_t* action_out, const uint8_t* stream, int32_t* cursor, int64_t stream_len,
                            float* obs, int32_t* reward, uint8_t* done, uint8_t* lines, uint8_t* n_valid_next,
                            uint8_t* piece_next, uint32_t* status, int32_t auto_reset, uint64_t seed,
                            uint64_t step_idx, int64_t env_offset, int64_t B, int64_t max_elems);

int tetris_hip_step_many(const TetrisDesc* desc, void* cols, uint64_t* meta, int32_t n_steps, int32_t policy,
                         const float* weights, int32_t* action_out, float* obs, int32_t* reward, uint8_t* done,
                         uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next, uint32_t* status,
                         int32_t auto_reset, uint64_t seed, uint64_t step_idx0, int64_t env_offset, int64_t B,
                         void* hip_stream) {
  if (n_steps < 1 || policy < 0 || policy > 1 || (policy == 1 && !weights)) return TETRIS_E_BATCH;
  StepManyParams q{};
  int rc = fill_step_params(q.one, desc, cols, meta, nullptr, action_out, nullptr, nullptr, 0, obs, reward, done,
                            lines, n_valid_next, piece_next, status, auto_reset, seed, step_idx0, env_offset, B,
                            (int64_t)B * n_steps);
  if (rc) return rc;
  q.n_steps = n_steps;
  q.policy = policy;
  q.seed = seed;
  q.step_idx0 = step_idx0;
  for (int i = 0; i < 8; ++i) q.w[i] = weights ? weights[i] : 0.f;
  return dispatch<LaunchStepMany>(desc, q, (hipStream_t)hip_stream);
}

int tetris_hip_step(const TetrisDesc* desc, void* cols, uint64_t* meta, const int32_t* action, int32_t* action_out,
                    const uint8_t* stream, int32_t* cursor, int64_t stream_len, float* obs, int32_t* reward,
                    uint8_t* done, uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next, uint32_t* status,
                    int32_t auto_reset, uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B,
                    void* hip_stream) {
  StepParams p{};
  int rc = fill_step_params(p, desc, cols, meta, action, action_out, stream, cursor, stream_len, obs, reward, done,
                            lines, n_valid_next, piece_next, status, auto_reset, seed, step_idx, env_offset, B, B);
  if (rc) return rc;
  return dispatch<LaunchStep>(desc, p, (hipStream_t)hip_stream);
}

static int fill_step_params(StepParams& p, const TetrisDesc* desc, void* cols, uint64_t* meta, const int32_t* action,
                            int32_t* action_out, const uint8_t* stream, int32_t* cursor, int64_t stream_len,
                            float* obs, int32_t* reward, uint8_t* done, uint8_t* lines, uint8_t* n_valid_next,
                            uint8_t* piece_next, uint32_t* status, int32_t auto_reset, uint64_t seed,
                            uint64_t step_idx, int64_t env_offset, int64_t B, int64_t max_elems) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta || !reward || !done || !lines || !n_valid_next) return TETRIS_E_NULL;
  if (B <= 0 || max_elems > 0x7FFFFFFF / 32) return TETRIS_E_BATCH;  // every byte offset (<= 32 B/element) fits 32 bits
  if ((B + 63) / 64 * 64 * (int64_t)tet::kMaxCols * 8 > 0xFFFFFFFFll) return TETRIS_E_BATCH;  // ... and so do the board records
  if (stream && (!cursor || stream_len <= 0)) return TETRIS_E_STREAM;
  p.cols = cols;
  p.meta = meta;
  p.action = action;
  p.action_out = action_out;
  p.stream = stream;
  p.cursor = cursor;
  p.stream_len = stream_len;
  p.obs = obs;
  p.reward = reward;
  p.done = done;
  p.lines = lines;
  p.n_valid = n_valid_next;
  p.piece_next = piece_next;
  p.status = status;
  p.done_bits = nullptr;        // (the gather payload is attached per call: tetris_hip_step_call_run_gather)
  p.status_snapshot = nullptr;
  p.B = (uint32_t)B;
  p.env_offset = (uint32_t)env_offset;
  p.step_counter = nullptr;
  p.seed = seed;
  p.step_rel = 0;
  p.cfg.R = desc->num_rows;
  p.cfg.n_pieces = desc->n_pieces;
  p.cfg.auto_reset = auto_reset;
  p.cfg.key_step = tet::hash_key(seed, step_idx * 4u + 0u);
  p.cfg.key_policy = tet::hash_key(seed, step_idx * 4u + 3u);
  p.cfg.compute_obs = obs != nullptr;
  p.cfg.has_direct_by = desc->has_direct_by;
  for (int i = 0; i < 8; ++i) p.cfg.direct_by[i] = desc->direct_by[i];
  build_table(desc, &p.tab);
  return TETRIS_OK;
}

// ---- bound step call: everything that does not change between steps is prepared once ------------
struct TetrisStepCall {
  uint64_t magic;
  StepParams p{};
  TetrisDesc desc;
  uint64_t seed;
  int32_t* action_out;  // written only when the built-in policy draws the action
};
constexpr uint64_t kStepCallMagic = 0x5445545249535343ull;  // "TETRISSC"

int64_t tetris_hip_step_call_size(void) { return (int64_t)sizeof(TetrisStepCall); }

int tetris_hip_step_call_init(void* call_, const TetrisDesc* desc, void* cols, uint64_t* meta, int32_t* action_out,
                              const uint8_t* stream, int32_t* cursor, int64_t stream_len, float* obs, int32_t* reward,
                              uint8_t* done, uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next,
                              uint32_t* status, int32_t auto_reset, uint64_t seed, int64_t env_offset, int64_t B) {
  if (!call_) return TETRIS_E_NULL;
  TetrisStepCall* call = static_cast<TetrisStepCall*>(call_);
  call->magic = 0;
  int rc = fill_step_params(call->p, desc, cols, meta, nullptr, nullptr, stream, cursor, stream_len, obs, reward, done,
                            lines, n_valid_next, piece_next, status, auto_reset, seed, 0, env_offset, B, B);
  if (rc) return rc;
  call->desc = *desc;
  call->seed = seed;
  call->action_out = action_out;
  call->magic = kStepCallMagic;
  return TETRIS_OK;
}

int tetris_hip_step_call_run(void* call_, const int32_t* action, uint64_t step_idx, void* hip_stream) {
  TetrisStepCall* call = static_cast<TetrisStepCall*>(call_);
  if (!call) return TETRIS_E_NULL;
  if (call->magic != kStepCallMagic) return TETRIS_E_DESC;
  StepParams& p = call->p;
  p.action = action;
  p.action_out = action ? nullptr : call->action_out;
  p.cfg.key_step = tet::hash_key(call->seed, step_idx * 4u + 0u);
  p.cfg.key_policy = tet::hash_key(call->seed, step_idx * 4u + 3u);
  return dispatch<LaunchStep>(&call->desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_step_call_run_gather(void* call_, const int32_t* action, uint64_t step_idx, uint64_t* done_bits,
                                    uint32_t* status_snapshot, void* hip_stream) {
  TetrisStepCall* call = static_cast<TetrisStepCall*>(call_);
  if (!call) return TETRIS_E_NULL;
  if (call->magic != kStepCallMagic) return TETRIS_E_DESC;
  StepParams p = call->p;  // (a copy: the bound call itself stays free of the payload pointers)
  p.action = action;
  p.action_out = action ? nullptr : call->action_out;
  p.cfg.key_step = tet::hash_key(call->seed, step_idx * 4u + 0u);
  p.cfg.key_policy = tet::hash_key(call->seed, step_idx * 4u + 3u);
  p.done_bits = reinterpret_cast<unsigned long long*>(done_bits);
  p.status_snapshot = status_snapshot;
  return dispatch<LaunchStep>(&call->desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_stream_link(void* from_stream, void* to_stream) {
  // `to_stream` waits for everything enqueued on `from_stream` so far.  The event releases to DEVICE scope:
  // producer and consumer run on the same GPU (the consumer is the RCCL kernel that reads the gather
  // payload), so the system-scope cache write-back a default event carries is not needed -- that write-back
  // is what made a recorded event cost the stepping stream tens of microseconds.
  hipEvent_t ev;
  hipError_t rc = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventReleaseToDevice);
  if (rc != hipSuccess) return (int)rc;
  rc = hipEventRecord(ev, (hipStream_t)from_stream);
  if (rc == hipSuccess) rc = hipStreamWaitEvent((hipStream_t)to_stream, ev, 0);
  const hipError_t rc2 = hipEventDestroy(ev);  // (released once the wait has consumed it)
  return (int)(rc != hipSuccess ? rc : rc2);
}

#ifndef TET_SRC_HASH
#define TET_SRC_HASH "unknown"
#endif
const char* tetris_hip_source_hash(void) { return TET_SRC_HASH; }

int tetris_hip_step_call_run_counted(void* call_, const int32_t* action, const uint64_t* step_counter, uint32_t step_rel,
                                     void* hip_stream) {
  TetrisStepCall* call = static_cast<TetrisStepCall*>(call_);
  if (!call || !step_counter) return TETRIS_E_NULL;
  if (call->magic != kStepCallMagic) return TETRIS_E_DESC;
  StepParams p = call->p;  // (a copy: the bound call itself stays usable for plain runs)
  p.action = action;
  p.action_out = action ? nullptr : call->action_out;
  p.step_counter = step_counter;
  p.step_rel = step_rel;
  return dispatch<LaunchStep>(&call->desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_pack_done_bits(const uint8_t* done, uint8_t* bits, int64_t B, void* hip_stream) {
  if (!done || !bits) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  hipLaunchKernelGGL(pack_done_bits_kernel, grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream, done,
                     reinterpret_cast<unsigned long long*>(bits), B);
  return (int)hipGetLastError();
}

int tetris_hip_counter_add(uint64_t* counter, uint64_t n, void* hip_stream) {
  if (!counter) return TETRIS_E_NULL;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, counter, n);
  return (int)hipGetLastError();
}

int tetris_hip_afterstates(const TetrisDesc* desc, const void* cols, const uint64_t* meta, float* feats,
                           uint8_t* n_valid, float* feats_all, uint8_t* n_all, int64_t env_stride,
                           int64_t row_stride, int64_t B, void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta || !feats || !n_valid) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  AfterParams p{};
  p.cols = cols;
  p.meta = meta;
  p.feats = feats;
  p.n_valid = n_valid;
  p.feats_all = feats_all;
  p.n_all = n_all;
  p.B = B;
  if (env_stride % 4 || row_stride % 4 || env_stride < 8 || row_stride < 8) return TETRIS_E_STRIDE;
  // the kernel addresses rows with 32-bit float4 indices
  if ((B - 1) * (env_stride / 4) + (int64_t)(desc->a_max - 1) * (row_stride / 4) + 2 >= 0xFFFFFFFFll) return TETRIS_E_STRIDE;
  p.env_stride = env_stride;
  p.row_stride = row_stride;
  p.R = desc->num_rows;
  p.a_max = desc->a_max;
  p.has_direct_by = desc->has_direct_by;
  for (int i = 0; i < 8; ++i) p.direct_by[i] = desc->direct_by[i];
  build_table(desc, &p.tab);
  return dispatch<LaunchAfter>(desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_policy_greedy(const TetrisDesc* desc, const void* cols, const uint64_t* meta, const float* weights,
                             int32_t* best_action, float* best_value, float* fitness_all, int64_t B,
                             void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta || !weights || !best_action) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  GreedyParams p{};
  p.cols = cols;
  p.meta = meta;
  p.best_action = best_action;
  p.best_value = best_value;
  p.fitness_all = fitness_all;
  p.B = B;
  p.R = desc->num_rows;
  p.a_max = desc->a_max;
  for (int i = 0; i < 8; ++i) p.w[i] = weights[i];
  build_table(desc, &p.tab);
  return dispatch<LaunchGreedy>(desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_rollouts(const TetrisDesc* desc, const void* cols, const uint64_t* meta, double* returns,
                        int32_t length, int32_t n, int32_t policy, const float* weights, const uint8_t* pieces,
                        uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B, void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta || !returns || (policy == 1 && !weights)) return TETRIS_E_NULL;
  if (B <= 0 || length < 1 || n < 1 || policy < 0 || policy > 1) return TETRIS_E_BATCH;
  RolloutParams p{};
  p.cols = cols;
  p.meta = meta;
  p.returns = returns;
  p.B = B;
  p.env_offset = env_offset;
  p.R = desc->num_rows;
  p.a_max = desc->a_max;
  p.n_pieces = desc->n_pieces;
  p.length = length;
  p.n = n;
  p.policy = policy;
  p.pieces = pieces;
  p.key = tet::hash_key(seed ^ 0x526F6C6C6F757473ull, step_idx);
  for (int i = 0; i < 8; ++i) p.w[i] = weights ? weights[i] : 0.f;
  build_table(desc, &p.tab);
  return dispatch<LaunchRollouts>(desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_refresh(const TetrisDesc* desc, const void* cols, uint64_t* meta, uint8_t* n_valid_out,
                       int64_t B, void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !meta) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  RefreshParams p{};
  p.cols = cols;
  p.meta = meta;
  p.n_valid_out = n_valid_out;
  p.B = B;
  p.R = desc->num_rows;
  build_table(desc, &p.tab);
  return dispatch<LaunchRefresh>(desc, p, (hipStream_t)hip_stream);
}

int tetris_hip_policy_random(const uint8_t* n_valid, int32_t* action, uint64_t seed, uint64_t step_idx,
                             int64_t env_offset, int64_t B, void* hip_stream) {
  if (!n_valid || !action) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  const uint32_t key = tet::hash_key(seed, step_idx * 4u + 3u);
  hipLaunchKernelGGL(policy_random_kernel, grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream, n_valid,
                     action, key, env_offset, B);
  return (int)hipGetLastError();
}

int tetris_hip_numpy_bag_stream(const uint32_t* seeds, int32_t n_pieces, int64_t L, uint8_t* stream, int64_t B,
                                void* hip_stream) {
  if (!seeds || !stream) return TETRIS_E_NULL;
  if (n_pieces < 1 || n_pieces > TETRIS_MAX_PIECES) return TETRIS_E_PIECES;
  if (B <= 0 || L <= 0) return TETRIS_E_BATCH;
  hipLaunchKernelGGL(numpy_bag_stream_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream,
                     seeds, n_pieces, L, stream, B);
  return (int)hipGetLastError();
}

int tetris_hip_decode(const TetrisDesc* desc, const void* cols, int8_t* cells, int32_t* heights, int64_t B,
                      void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  const int C = desc->num_columns, rows = desc->num_rows + 4;
  const bool packed = tet::board_packed(desc->word_bytes, desc->num_rows);
  if (desc->word_bytes == 4)
    hipLaunchKernelGGL((decode_kernel<uint32_t>), grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream,
                       static_cast<const uint32_t*>(cols), cells, heights, C, rows, B, packed);
  else
    hipLaunchKernelGGL((decode_kernel<uint64_t>), grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream,
                       static_cast<const uint64_t*>(cols), cells, heights, C, rows, B, packed);
  return (int)hipGetLastError();
}

int tetris_hip_encode(const TetrisDesc* desc, const int8_t* cells, void* cols, int64_t B, void* hip_stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!cols || !cells) return TETRIS_E_NULL;
  if (B <= 0) return TETRIS_E_BATCH;
  const int C = desc->num_columns, rows = desc->num_rows + 4;
  const bool packed = tet::board_packed(desc->word_bytes, desc->num_rows);
  if (desc->word_bytes == 4)
    hipLaunchKernelGGL((encode_kernel<uint32_t>), grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream, cells,
                       static_cast<uint32_t*>(cols), C, rows, B, packed);
  else
    hipLaunchKernelGGL((encode_kernel<uint64_t>), grid_for(B), dim3(kBlock), 0, (hipStream_t)hip_stream, cells,
                       static_cast<uint64_t*>(cols), C, rows, B, packed);
  return (int)hipGetLastError();
}

}  // extern "C"
#endif  // !TET_PART
