// tetris_core.hpp -- per-env bitboard logic of the placement-level Tetris step.
//
// Every function here is the body one GPU lane runs for one env.  The same
// source also compiles for the host (g++) ONLY so that tests/harness can run
// the lane logic under the CPU test-suite and sanitizers; the product path is
// the HIP kernels in tetris_kernels.hip (there is no CPU fallback).
//
// Board layout: column bitboards.  Bit r of col[c] = cell (row r, column c),
// row 0 = bottom; stored rows = R + 4 (reference: game.py:56, state.py:27-30).
// Semantics follow /root/reference (cited per function); the closed forms are
// SURVEY.md Appendix A/B.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TET_HD __host__ __device__ __forceinline__
#else
#define TET_HD inline
#endif

// Experiment switch for tools/ablate.py (timing-only builds with parts of the
// step removed; results are wrong when nonzero).  Always 0 in the product build.
#ifndef TET_ABLATE
#define TET_ABLATE 0
#endif
#ifndef TET_BAG_SMALL_PATH
#define TET_BAG_SMALL_PATH 1   // 0: always the three-group search of the bag draw (A/B timing)
#endif
#ifndef TET_RESCUE_SKIP
#define TET_RESCUE_SKIP 1   // 0: valid_mask always evaluates the rescue by cleared rows (A/B timing)
#endif

// wave-uniform "does any active lane want this": lets code that only some pieces need be
// skipped by the whole wavefront (one env per lane).  On the host build it is the lane itself.
// Fence for the instruction scheduler between the columns of board_features: without it the
// compiler hoists every table read of all ten columns to the top (hundreds of live registers on
// u64 boards -> spills or one wave per SIMD); latency is hidden by the other waves anyway.
#ifndef TET_FENCE_EVERY
#define TET_FENCE_EVERY 2
#endif
#ifndef TET_FENCE_MASK
#define TET_FENCE_MASK 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define TET_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// materialise an accumulator here: stops the compiler from re-associating the per-column sums into
// one reduction at the end of the loop, which keeps every table value it has read live until then
#define TET_PIN(x) asm volatile("" : "+v"(x))
#else
#define TET_SCHED_FENCE() ((void)0)
#define TET_PIN(x) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define TET_WAVE_ANY(x) (__ballot((x)) != 0ull)
#else
#define TET_WAVE_ANY(x) (x)
#endif

// OR into a per-lane scratch word (LDS on the device: a single ds_or, no return value)
#if defined(__HIP_DEVICE_COMPILE__)
#define TET_SCRATCH_OR(ptr, v) ((void)__hip_atomic_fetch_or((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#else
#define TET_SCRATCH_OR(ptr, v) (*(ptr) |= (v))
#endif

// float32 multiply / add that the compiler may not contract into an FMA (the reference's
// fitness sum rounds every product and every partial sum, game.py:109-118)
#if defined(__HIP_DEVICE_COMPILE__)
#define TET_FMUL(a, b) __fmul_rn((a), (b))
#define TET_FADD(a, b) __fadd_rn((a), (b))
#else
#define TET_FMUL(a, b) ((float)((volatile float)(a) * (b)))
#define TET_FADD(a, b) ((float)((volatile float)(a) + (b)))
#endif

namespace tet {

constexpr int kMaxPieces = 12;   // pieces in one set (bag is 12 bits of meta)
constexpr int kMaxCols = 12;     // 4 x 12-bit mask fields fill the 48-bit valid mask
constexpr int kNumCatalogue = 9;

// ---- packed orientation descriptor --------------------------------------
// bits 0-2 w | 3-5 H | for j<4: b_j at 6+5j (2 bits), n_j at 8+5j (3 bits) | 31 exists
// Footprint column j of a placement with left column c covers rows
// a+b_j .. a+b_j+n_j-1 of column c+j, a = max_j(h[c+j]-b_j)  (tetromino.py,
// e.g. :122-128).
struct Orient {
  int w, H, b[4], n[4];
  bool exists;
};

TET_HD Orient unpack_orient(uint32_t d) {
  Orient o;
  o.w = d & 7;
  o.H = (d >> 3) & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o.b[j] = (d >> (6 + 5 * j)) & 3;
    o.n[j] = (d >> (8 + 5 * j)) & 7;
  }
  o.exists = (d >> 31) != 0;
  return o;
}

// ---- storage of a board in HBM ---------------------------------------------
// Tile-major words: envs are grouped in tiles of 64 (one wavefront), and the planes of one tile are
// stored back to back, `words[tile][p][lane]` with tile = env / 64, lane = env % 64 -- so the board
// state of a wavefront is ONE contiguous record of n_planes * 64 words (2 KiB at 10x20): a wave
// streams it with full 256-byte rows per plane, from one DRAM page instead of n_planes pages
// millions of envs apart (measured on the step's traffic pattern at 4 Mi envs: 1.4x).  The
// last tile is padded to 64 lanes.  Unpacked: plane c = column c.  Packed (chosen whenever the
// stored rows R + 4 fit three quarters of the word: 24 bits of u32, 48 of u64): the columns are
// F = 24 / 48-bit fields of one bit string, four columns per three words, so ten columns take
// eight planes instead of ten -- a fifth fewer board bytes per env-step on the paper's 10x20 and
// 10x40 boards.  board_packed() is a function of the descriptor alone: every kernel, the host
// harness and the Python host agree on it through tetris_hip_n_planes().
#ifndef TET_NO_PACK
#define TET_NO_PACK 0   // 1: one plane per column always (A/B timing builds)
#endif
TET_HD constexpr bool board_packed(int word_bytes, int num_rows) { return !TET_NO_PACK && num_rows + 4 <= 6 * word_bytes; }
TET_HD constexpr int n_planes(int C, bool packed) { return packed ? (C / 4) * 3 + ((C % 4) * 3 + 3) / 4 : C; }
constexpr int kTileEnvs = 64;
// word index of (env i, plane p) in storage of NP planes
TET_HD constexpr int64_t plane_index(int64_t i, int p, int NP) { return ((i >> 6) * NP + p) * kTileEnvs + (i & 63); }
TET_HD constexpr int64_t board_words(int64_t B, int NP) { return ((B + kTileEnvs - 1) / kTileEnvs) * NP * kTileEnvs; }

template <typename W, int C, bool PACK>
TET_HD void unpack_board(const W (&w)[n_planes(C, PACK)], W (&col)[C]) {
  if (!PACK) {
#pragma unroll
    for (int c = 0; c < C; ++c) col[c] = w[c < n_planes(C, PACK) ? c : 0];
    return;
  }
  constexpr int Wb = 8 * (int)sizeof(W), F = Wb * 3 / 4, P = n_planes(C, PACK);
  const W M = (W)(((W)1 << F) - 1);
#pragma unroll
  for (int g = 0; 4 * g < C; ++g) {
    const int b = 3 * g, c0 = 4 * g;
    const W w0 = w[b], w1 = w[b + 1 < P ? b + 1 : b], w2 = w[b + 2 < P ? b + 2 : b];
    col[c0] = (W)(w0 & M);
    if (c0 + 1 < C) col[c0 + 1] = (W)(((w0 >> F) | (w1 << (Wb - F))) & M);
    if (c0 + 2 < C) col[c0 + 2] = (W)(((w1 >> (2 * F - Wb)) | (w2 << (2 * Wb - 2 * F))) & M);
    if (c0 + 3 < C) col[c0 + 3] = (W)(w2 >> (3 * F - 2 * Wb));
  }
}

// (columns must fit F bits: boards of R + 4 <= F stored rows)
template <typename W, int C, bool PACK>
TET_HD void pack_board(const W (&col)[C], W (&w)[n_planes(C, PACK)]) {
  if (!PACK) {
#pragma unroll
    for (int c = 0; c < n_planes(C, PACK); ++c) w[c] = col[c < C ? c : 0];
    return;
  }
  constexpr int Wb = 8 * (int)sizeof(W), F = Wb * 3 / 4, P = n_planes(C, PACK);
#pragma unroll
  for (int g = 0; 4 * g < C; ++g) {
    const int b = 3 * g, c0 = 4 * g;
    const W x0 = col[c0], x1 = c0 + 1 < C ? col[c0 + 1 < C ? c0 + 1 : c0] : (W)0;
    const W x2 = c0 + 2 < C ? col[c0 + 2 < C ? c0 + 2 : c0] : (W)0, x3 = c0 + 3 < C ? col[c0 + 3 < C ? c0 + 3 : c0] : (W)0;
    w[b] = (W)(x0 | (x1 << F));
    if (b + 1 < P) w[b + 1 < P ? b + 1 : b] = (W)((x1 >> (Wb - F)) | (x2 << (2 * F - Wb)));
    if (b + 2 < P) w[b + 2 < P ? b + 2 : b] = (W)((x2 >> (2 * Wb - 2 * F)) | (x3 << (3 * F - 2 * Wb)));
  }
}

// the same for memory (tile-major, plane_index); B is not needed for addressing
template <typename W, int C, bool PACK>
TET_HD void load_board(const W* planes, int64_t /*B*/, int64_t i, W (&col)[C]) {
  constexpr int NP = n_planes(C, PACK);
  W w[NP];
  const W* rec = planes + plane_index(i, 0, NP);
#pragma unroll
  for (int p = 0; p < NP; ++p) w[p] = rec[p * kTileEnvs];
  unpack_board<W, C, PACK>(w, col);
}
template <typename W, int C, bool PACK>
TET_HD void store_board(W* planes, int64_t /*B*/, int64_t i, const W (&col)[C]) {
  constexpr int NP = n_planes(C, PACK);
  W w[NP];
  pack_board<W, C, PACK>(col, w);
  W* rec = planes + plane_index(i, 0, NP);
#pragma unroll
  for (int p = 0; p < NP; ++p) rec[p * kTileEnvs] = w[p];
}

// run-time column count (codec kernels): column c of env i / all planes of env i from a column array
template <typename W>
TET_HD W load_column_rt(const W* planes, int64_t /*B*/, int64_t i, int c, int C, bool packed) {
  const int NP = n_planes(C, packed);
  if (!packed) return planes[plane_index(i, c, NP)];
  constexpr int Wb = 8 * (int)sizeof(W), F = Wb * 3 / 4;
  const W M = (W)(((W)1 << F) - 1);
  const int b = 3 * (c >> 2), k = c & 3;
  const W lo = planes[plane_index(i, b + (k == 0 ? 0 : k - 1), NP)];
  if (k == 0) return (W)(lo & M);
  if (k == 3) return (W)(lo >> (3 * F - 2 * Wb));
  const W hi = planes[plane_index(i, b + k, NP)];
  return k == 1 ? (W)(((lo >> F) | (hi << (Wb - F))) & M) : (W)(((lo >> (2 * F - Wb)) | (hi << (2 * Wb - 2 * F))) & M);
}
template <typename W>
TET_HD void store_columns_rt(W* planes, int64_t /*B*/, int64_t i, const W* col, int C, bool packed) {
  if (!packed) {
    for (int c = 0; c < C; ++c) planes[plane_index(i, c, C)] = col[c];
    return;
  }
  constexpr int Wb = 8 * (int)sizeof(W), F = Wb * 3 / 4;
  const int P = n_planes(C, true);
  for (int p = 0; p < P; ++p) {
    const int c0 = 4 * (p / 3), q = p % 3;
    const W a = c0 + q < C ? col[c0 + q] : (W)0, b = c0 + q + 1 < C ? col[c0 + q + 1] : (W)0;
    const W v = q == 0 ? (W)(a | (b << F))
                       : (q == 1 ? (W)((a >> (Wb - F)) | (b << (2 * F - Wb))) : (W)((a >> (2 * Wb - 2 * F)) | (b << (3 * F - 2 * Wb))));
    planes[plane_index(i, p, P)] = v;
  }
}

// ---- meta word ------------------------------------------------------------
// bits 0-47  valid mask: four 12-bit fields (the low C bits used), field k = 2L + o (loop L,
//            orientation o of tetromino.py's enumeration) at bit 12k, bit c of a field = left
//            column c.  The reference enumerates (L, c, o) in that order, so action k walks fields
//            0 and 1 interleaved by column, then fields 2 and 3 (slot_of_action).  The fixed
//            12-bit stride keeps every 4-column group of a field inside one nibble (table decode).
// bits 48-51 current piece (list index, game.py:38-39)
// bits 52-63 bag: list indices still to be drawn (tetromino.py:12-22)
constexpr uint64_t kMaskBits = (1ull << 48) - 1;
constexpr int kFieldStride = 12;
TET_HD constexpr int mask_bit(int k, int c) { return kFieldStride * k + c; }
TET_HD uint64_t meta_pack(uint64_t mask, int piece, uint32_t bag) {
  return (mask & kMaskBits) | ((uint64_t)(piece & 15) << 48) | ((uint64_t)(bag & 0xFFF) << 52);
}
TET_HD uint64_t meta_mask(uint64_t m) { return m & kMaskBits; }
TET_HD int meta_piece(uint64_t m) { return (int)((m >> 48) & 15); }
TET_HD uint32_t meta_bag(uint64_t m) { return (uint32_t)(m >> 52); }

// ---- word helpers -----------------------------------------------------------
TET_HD int popc(uint32_t x) { return __builtin_popcount(x); }
TET_HD int popc(uint64_t x) { return __builtin_popcountll(x); }
// bit length (index of the top set bit + 1; 0 for 0).  The top bit of a board word is never
// used (stored rows < bits(W)), so (x << 1) | 1 is a non-zero word one bit longer: no select.
TET_HD int bitlen(uint32_t x) { return 31 - __builtin_clz((x << 1) | 1u); }
TET_HD int bitlen(uint64_t x) { return 63 - __builtin_clzll((x << 1) | 1ull); }
// (1 << n) - 1 for 0 <= n < bits(W)   (stored rows < bits(W) by contract)
template <typename W>
TET_HD W lowmask(int n) { return (W)(((W)1 << n) - 1); }

// ---- counter-based bag (build design; same distribution as popping a fresh
// np.random.permutation front to back, tetromino.py:17-22) ------------------
TET_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// env-independent part: computed once per launch on the host (per step inside step_many)
TET_HD uint32_t hash_key(uint64_t seed, uint64_t counter) {
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9U);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ (uint32_t)counter);
  k = mix32(k ^ (uint32_t)(counter >> 32));
  return k;
}
// per-env part (two multiplies); env = global env index mod 2^32
TET_HD uint32_t hash_env(uint32_t key, uint32_t env) {
  uint32_t h = (key ^ env) * 0x9E3779B1U;
  h ^= h >> 16;
  h *= 0x85EBCA6BU;
  h ^= h >> 13;
  return h;
}
// uniform integer in [0, n) from 16 random bits (n <= 64)
TET_HD int scale16(uint32_t r16, int n) { return (int)((r16 * (uint32_t)n) >> 16); }

// position of the k-th (0-based) set bit of x, k < popc(x)
TET_HD int select_bit32(uint32_t v, int k) {
  int pos = 0;
  int c = popc(v & 0xFFFFu);
  if (k >= c) { k -= c; pos += 16; v >>= 16; }
  c = popc(v & 0xFFu);
  if (k >= c) { k -= c; pos += 8; v >>= 8; }
  c = popc(v & 0xFu);
  if (k >= c) { k -= c; pos += 4; v >>= 4; }
  c = popc(v & 0x3u);
  if (k >= c) { k -= c; pos += 2; v >>= 2; }
  c = (int)(v & 1u);
  if (k >= c) { pos += 1; }
  return pos;
}
// draw one piece from the bag with 16 random bits (bag <= 12 bits)
TET_HD int bag_draw(uint32_t& bag, int n_pieces, uint32_t r16) {
  if (bag == 0) bag = (1u << n_pieces) - 1u;
  const int k = scale16(r16, popc(bag));
  uint32_t v = bag;
  int pos = 0;
  int c = popc(v & 0xFFu);
  if (k >= c) { pos = 8; v >>= 8; }
  int kk = k >= c ? k - c : k;
  c = popc(v & 0xFu);
  if (kk >= c) { kk -= c; pos += 4; v >>= 4; }
  c = popc(v & 0x3u);
  if (kk >= c) { kk -= c; pos += 2; v >>= 2; }
  c = (int)(v & 1u);
  if (kk >= c) pos += 1;
  bag &= ~(1u << pos);
  return pos;
}

// the same draw through the nibble select table (LutLayout::kSelNib)
TET_HD int bag_draw_lut(uint32_t& bag, int n_pieces, uint32_t r16, const uint8_t* sel_nib) {
  if (bag == 0) bag = (1u << n_pieces) - 1u;
#if TET_BAG_SMALL_PATH
  if (n_pieces <= 4) {  // (uniform over the launch) the whole bag is one nibble: no group search
    const int k4 = scale16(r16, popc(bag));
    const int pos4 = (int)sel_nib[bag * 4u + (uint32_t)k4];
    bag &= ~(1u << pos4);
    return pos4;
  }
#endif
  int k = scale16(r16, popc(bag));
  const int c0 = popc(bag & 0xFu), c1 = c0 + popc(bag & 0xF0u);
  const int grp = (k >= c0) + (k >= c1);
  k -= grp == 0 ? 0 : (grp == 1 ? c0 : c1);
  const uint32_t nib = (bag >> (4 * grp)) & 15u;
  const int pos = 4 * grp + (int)sel_nib[nib * 4u + (uint32_t)k];
  bag &= ~(1u << pos);
  return pos;
}

// ---- per-set table staged in LDS -------------------------------------------
// One entry per (piece of the set, orientation slot k = 2L + o).  Every valid_mask field has a
// whole word to itself on purpose: the kernels are integer-VALU bound while the LDS pipe idles,
// so fields are fetched ready to use (wide LDS reads) instead of being shifted out of packed words.
struct alignas(16) OrientEntry {  // 48 bytes = three 16-byte LDS reads (field order matters for that)
  uint32_t sh[4];      // per footprint column j: 10 * (need_j - 1) + j, a shift into the level word
  uint32_t rj0[2];     // rescue rows t = 1, 2 (board rows R-3+t): first piece column of that row,
  uint32_t rj1[2];     //   last piece column; rj0 = 31 when the piece has no such row
  uint32_t vert4;      // all ones for the vertical Straight, else 0
  uint32_t desc;       // packed Orient descriptor (above)
  uint32_t pad_[2];
};
static_assert(sizeof(OrientEntry) == 48, "table staging and LDS reads assume 48-byte entries");

struct SetTable {
  OrientEntry orient[kMaxPieces][4];
  uint64_t fullmask[kMaxPieces];   // all existing placements (every one valid)
};

// The four entries of piece `np`.  The byte offset is laundered through a register so that the
// compiler addresses every field as ONE base register + immediate offsets (it otherwise folds the
// field offset into a separate multiply-add per LDS read).
TET_HD const OrientEntry* piece_entries(const SetTable& tab, int np) {
  uint32_t off = (uint32_t)np * (uint32_t)sizeof(tab.orient[0]);
  TET_PIN(off);
  return reinterpret_cast<const OrientEntry*>(reinterpret_cast<const char*>(&tab.orient[0][0]) + off);
}

// feature tables (tools/gen_feature_lut.py), staged in LDS by the kernels as ONE block of byte
// tables, one per field so nothing has to be shifted or masked out of a packed entry:
// hole_A, hole_u (index: a CR-row chunk of a column + the cell above), wells_S, wells_lead,
// wells_trail (index: a CR-row chunk of a column's well cells).  CR = 12 rows per chunk in general
// (28 KiB); the stepping kernels use CR = 10 (7 KiB) on boards of up to 20 rows, where two
// chunks cover every cell a non-terminal board can hold.
template <int CR>
struct LutLayout {
  static constexpr int kHoleEntries = 1 << (CR + 1), kWellsEntries = 1 << CR;
  static constexpr int kHoleA = 0, kHoleU = kHoleEntries, kWellsS = 2 * kHoleEntries;
  static constexpr int kWellsLead = kWellsS + kWellsEntries, kWellsTrail = kWellsLead + kWellsEntries;
  // select tables (the same in every set): k-th placement of a 4-column group of one loop's two
  // orientation fields in the reference's order; k-th set bit of a nibble
  static constexpr int kSelPair = kWellsTrail + kWellsEntries, kSelNib = kSelPair + 256 * 8;
  static constexpr int kBytes = kSelNib + 16 * 4;
};
constexpr int kFeatureLutBytes = LutLayout<12>::kBytes;
constexpr int kFeatureLut10Bytes = LutLayout<10>::kBytes;
// tables of the afterstate kernels (tetris_after_lut.inc): the hole tables for 12-row chunks at the
// offsets of LutLayout<12>, then ONE 32-bit entry per 12-row chunk of a column's well cells,
// S | trail << 16 | lead << 24.  The afterstate walk re-evaluates the wells of up to five columns per
// placement and is bound by the number of LDS reads (one pipe per CU, 3-4-way bank conflicts on
// random indices), so it fetches the three fields with one read; the stepping kernels are bound by
// vector instructions and keep the ready-to-use byte tables.
struct AfterLut {
  static constexpr int kHoleA = LutLayout<12>::kHoleA, kHoleU = LutLayout<12>::kHoleU;
  static constexpr int kWellsPack = 2 * LutLayout<12>::kHoleEntries;
  static constexpr int kSelPair = kWellsPack + 4 * LutLayout<12>::kWellsEntries, kSelNib = kSelPair + 256 * 8;
  static constexpr int kBytes = kSelNib + 16 * 4;
};
constexpr int kAfterLutBytes = AfterLut::kBytes;

// bit c -> bit 2c (c < 16)
TET_HD uint32_t spread2(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

template <int C>
TET_HD uint32_t mask_field(uint64_t mask, int k) { return (uint32_t)(mask >> (kFieldStride * k)) & ((1u << C) - 1u); }

// action k -> (field kk = 2L + o, column c), given the valid mask (game.py:69,83: index into
// the non-terminal placements in enumeration order loop, column, orientation).  Table form: pick
// the loop, then the 4-column group by two masked popcounts, then ONE byte of sel_pair decodes the
// position inside the group -- about a third of the vector instructions of the bit-interleaving
// form below (the kernels are VALU-bound and the LDS pipe has room).
template <int C>
TET_HD void slot_of_action_lut(uint64_t mask, int k, const uint8_t* sel_pair, int& kk, int& c) {
  static_assert(C <= 12, "three 4-column groups per 12-bit field");
  constexpr uint32_t kLoopBits = (1u << (2 * kFieldStride)) - 1u;
  const uint32_t w0 = (uint32_t)mask & kLoopBits, w1 = (uint32_t)(mask >> (2 * kFieldStride)) & kLoopBits;
  const int n0 = popc(w0);
  const bool second = k >= n0;
  const uint32_t w = second ? w1 : w0;
  int kr = second ? k - n0 : k;
  constexpr uint32_t kG0 = 0xFu | (0xFu << kFieldStride);  // columns 0-3 of both orientation fields
  const int c0 = popc(w & kG0), c1 = c0 + popc(w & (kG0 << 4));
  const int grp = (C > 4 ? (kr >= c0) : 0) + (C > 8 ? (kr >= c1) : 0);
  kr -= grp == 0 ? 0 : (grp == 1 ? c0 : c1);
  const uint32_t t = w >> (4 * grp);
  const uint32_t idx = (t & 15u) | (((t >> kFieldStride) & 15u) << 4);
  const uint32_t r = sel_pair[idx * 8u + (uint32_t)kr];
  c = 4 * grp + (int)(r & 3u);
  kk = (second ? 2 : 0) + (int)(r >> 2);
}
template <int C>
TET_HD void slot_of_action(uint64_t mask, int k, int& kk, int& c) {
  const uint32_t f0 = mask_field<C>(mask, 0), f1 = mask_field<C>(mask, 1);
  const uint32_t f2 = mask_field<C>(mask, 2), f3 = mask_field<C>(mask, 3);
  const int n0 = popc(f0) + popc(f1);
  const bool second = k >= n0;
  const uint32_t z = spread2(second ? f2 : f0) | (spread2(second ? f3 : f1) << 1);  // (c, o) order
  const int pos = select_bit32(z, second ? k - n0 : k);
  c = pos >> 1;
  kk = (second ? 2 : 0) + (pos & 1);
}
// row of placement (kk, c) in the list of the placements whose bits are set in `mask`
template <int C>
TET_HD int row_of_slot(uint64_t mask, int kk, int c) {
  const int L = kk >> 1;
  const uint32_t fa = mask_field<C>(mask, 2 * L), fb = mask_field<C>(mask, 2 * L + 1);
  const uint32_t below = (1u << c) - 1u;
  int row = popc(fa & below) + popc(fb & below) + ((kk & 1) ? (int)((fa >> c) & 1u) : 0);
  if (L) row += popc(mask_field<C>(mask, 0)) + popc(mask_field<C>(mask, 1));
  return row;
}

template <typename W, int C>
TET_HD void heights_of(const W (&col)[C], int (&h)[C]) {
#pragma unroll
  for (int c = 0; c < C; ++c) h[c] = bitlen(col[c]);  // state.py:162-172
}

// Landing row + stamp of orientation `o` at STATIC left column c.
// Returns anchor row a; writes the stamped columns into nb (copy of col).
template <typename W, int C>
TET_HD int stamp_static(const W (&col)[C], const int (&h)[C], int c, const Orient& o, W (&nb)[C],
                        W (&pbits)[4]) {
  int a = -64;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c + j < C) {
      int v = (j < o.w) ? h[c + j] - o.b[j] : -64;
      a = v > a ? v : a;
    }
  if (a < 0) a = 0;  // only for slots that do not exist (masked out by the caller)
#pragma unroll
  for (int i = 0; i < C; ++i) nb[i] = col[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int nj = (j < o.w) ? o.n[j] : 0;
    pbits[j] = (W)(lowmask<W>(nj) << (a + o.b[j]));
    if (c + j < C) nb[c + j] |= pbits[j];
  }
  return a;
}

// state.py:121-143 on bitboards.  F = rows full in every column; each column
// drops those bits (rows above shift down, zero rows enter at the top).
// Only rows from the anchor row `a` upwards are looked at: the reference tests its
// `changed_lines` (rows of the piece, state.py:33,123-126) and nothing else, so a full row that a
// board set from outside already holds stays where it is -- such a row lies below every piece cell
// (a column filled at row r has height > r, and the column with bottom offset 0 puts a > r).
// On boards a game can reach there is no such row and the restriction changes nothing.
// Returns n_cleared; *eroded = piece cells that sat in cleared rows
// (state.py:99: sum(cleared_rows * pieces_per_changed_row)).
template <typename W, int C>
TET_HD int clear_lines(W (&col)[C], const W (&pbits)[4], int a, int* eroded_cells) {
  W F = (W)((W)~(W)0 << a);
#pragma unroll
  for (int i = 0; i < C; ++i) F &= col[i];
  int k = popc(F);
  int er = 0;
  if (F != 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) er += popc((W)(F & pbits[j]));
    while (F != 0) {
      int p = bitlen(F) - 1;  // highest full row first: lower rows keep their index
      W low = lowmask<W>(p);
#pragma unroll
      for (int i = 0; i < C; ++i) col[i] = (W)((col[i] & low) | ((col[i] >> (p + 1)) << p));
      F = (W)(F & low);
    }
  }
  *eroded_cells = er;
  return k;
}

// ---- per-column pieces of state.py:175-280 (closed forms of SURVEY App. B) ------------
// Part that depends on the column alone: holes, column transitions, hole depth.
// Hole depth (state.py:200,216,239): the top hole of each vertical run counts the filled cells
// above it.  12-row chunks through the tables: u = the run tops inside the chunk, A = their
// filled cells above inside the chunk; the cells above the chunk count once per run top.
// Column transitions (state.py:206,219-220,242-243) below the column top come in pairs, one
// entering and one leaving every hole run (the floor counts as filled, the top cell is filled),
// so they are 2 * (number of run tops) -- the same u.
// NCH > 0: the board has exactly NCH 12-row chunks (compile-time); NCH = 0: decide from R.
// Columns hold no bits at or above row R + 4, so the top chunk of an NCH board needs no mask.
template <typename W, int NCH = 0, int CR = 12>
TET_HD void col_own(W x, int hi, int R, const uint8_t* lut, W& ho, int& nh, int& f1, int& f7) {
  ho = (W)(~x & lowmask<W>(hi));                   // holes (state.py:210-213)
  nh = popc(ho);
  const uint8_t* lut_a = lut + LutLayout<CR>::kHoleA;
  const uint8_t* lut_u = lut + LutLayout<CR>::kHoleU;
  int d7 = 0, u = 0;
  if (!(TET_ABLATE & 16)) {
#pragma unroll
    for (int k = 0; CR * k < (int)(8 * sizeof(W)) - 1; ++k) {
      // rows beyond the stored ones are zero: entry 0 adds nothing, so extra chunks are harmless
      if (NCH > 0 ? k < NCH : (k < 2 || CR * k < R + 4)) {
        const uint32_t up = (uint32_t)(x >> (CR * k));
        const uint32_t idx = (NCH > 0 && k == NCH - 1) ? up : (up & (uint32_t)(LutLayout<CR>::kHoleEntries - 1));
        const int uk = lut_u[idx];
        u += uk;
        d7 += lut_a[idx];
        const bool more = NCH > 0 ? k + 1 < NCH : (CR * (k + 1) < R + 4);  // rows above this chunk exist
        if (CR * (k + 1) < (int)(8 * sizeof(W)) && more) d7 += uk * popc((W)(x >> (CR * (k + 1))));
      }
    }
  }
  f1 = 2 * u;
  f7 = d7;
}

// Row transitions of one column against its LEFT neighbour (state.py:203-204,223-226,
// 246-248,253-254).  Empty column: the filled cells of the left neighbour = hL - its holes
// (:254); otherwise max(hL-h, 0).
template <typename W>
TET_HD int col_rowtrans(W x, W L, int hi, int hL, int nh_left) {
  const int dl = hL - hi;
  return popc((W)((x ^ L) & lowmask<W>(hi))) + (dl > 0 ? dl : 0) - ((hi == 0) ? nh_left : 0);
}

// Cumulative wells of one column (state.py:223-233 inside the column, :258-272 above it).
// A well cell is an empty cell whose two neighbours are filled: inside the column (rows < h) the
// walls count as filled on every stored row (state.py:177-178); above it only rows below
// min(hL, hR) count, with wall height R (state.py:179,258-261) -- neighbours have no cells at or
// above their own height, so for inner columns the set is simply ~x & L & R, and for the edge
// columns the wall side is cut at max(h, R).  Every maximal vertical run of k well cells adds
// k(k+1)/2: summed per 12-row chunk through the wells tables (S, lead, trail) with a carry for
// runs that cross chunk borders -- no data-dependent loop.  (A full chunk has trail = lead = 12.)
template <typename W, int NCH = 0, int CR = 12>
TET_HD int col_wells(W x, W L, W Rr, int hi, int R, bool left_wall, bool right_wall, const uint8_t* lut) {
  W w = (W)(~x & L & Rr);
  if (left_wall || right_wall) w = (W)(w & lowmask<W>(hi > R ? hi : R));
  const uint8_t* lut_s = lut + LutLayout<CR>::kWellsS;
  const uint8_t* lut_lead = lut + LutLayout<CR>::kWellsLead;
  const uint8_t* lut_trail = lut + LutLayout<CR>::kWellsTrail;
  int total = 0, carry = 0;
#pragma unroll
  for (int k = 0; CR * k < (int)(8 * sizeof(W)) - 1; ++k) {
    if (NCH > 0 ? k < NCH : (k < 2 || CR * k < R + 4)) {  // rows beyond the stored ones hold no well cells
      const uint32_t up = (uint32_t)(w >> (CR * k));
      const uint32_t idx = (NCH > 0 && k == NCH - 1) ? up : (up & (uint32_t)(LutLayout<CR>::kWellsEntries - 1));
      const bool last = NCH > 0 && k == NCH - 1;
      total += lut_s[idx];
      if (k == 0) {
        if (!last) carry = lut_trail[idx];
      } else {
        const int lead = lut_lead[idx];
        total += carry * lead;
        if (!last) {
          const int trail = lut_trail[idx];  // read unconditionally: a select, not a branch around the load
          carry = (lead == CR) ? carry + CR : trail;
        }
      }
    }
  }
  return (TET_ABLATE & 32) ? 0 : total;
}

// The same sum through the packed table of the afterstate kernels (AfterLut::kWellsPack: 12-row chunks,
// entry = S | trail << 16 | lead << 24): one LDS read per chunk.
template <typename W, int NCH = 0>
TET_HD int col_wells_packed(W x, W L, W Rr, int hi, int R, bool left_wall, bool right_wall, const uint32_t* pack) {
  constexpr int CR = 12;
  W w = (W)(~x & L & Rr);
  if (left_wall || right_wall) w = (W)(w & lowmask<W>(hi > R ? hi : R));
  uint32_t total = 0, carry = 0;
#pragma unroll
  for (int k = 0; CR * k < (int)(8 * sizeof(W)) - 1; ++k) {
    if (NCH > 0 ? k < NCH : (k < 2 || CR * k < R + 4)) {
      const uint32_t up = (uint32_t)(w >> (CR * k));
      const bool last = NCH > 0 && k == NCH - 1;
      const uint32_t e = pack[last ? up : (up & 4095u)];
      total += e & 0xFFFFu;
      const uint32_t lead = e >> 24, trail = (e >> 16) & 255u;
      if (k == 0) {
        if (!last) carry = trail;
      } else {
        total += carry * lead;
        if (!last) carry = (lead == (uint32_t)CR) ? carry + (uint32_t)CR : trail;
      }
    }
  }
  return (int)total;
}

// state.py:175-280.  out = f0,f1,f2,f4,f5,f7.
// PACKW: hole_lut is an AfterLut (packed wells entries) instead of a LutLayout<CR>
template <typename W, int C, int NCH = 0, int CR = 12, bool PACKW = false>
TET_HD void board_features(const W (&col)[C], const int (&h)[C], int R, const uint8_t* hole_lut,
                           int& rows_with_holes, int& col_trans, int& holes, int& wells, int& row_trans,
                           int& hole_depth) {
  const W wall = lowmask<W>(R + 4);  // walls of ones over every stored row (state.py:177-178)
  W hole_rows = 0;
  int f1 = C;                      // one unconditional transition per column (state.py:194)
  int f2 = 0, f4 = 0, f7 = 0;
  int f5 = R - popc(col[C - 1]);   // state.py:190
  int nh_left = 0;                 // holes of the left neighbour (the wall has none)
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const W L = (i == 0) ? wall : col[i - 1];
    const W Rr = (i == C - 1) ? wall : col[i + 1];
    const int hL = (i == 0) ? R : h[i - 1];       // state.py:179 wall height = num_rows
    W ho;
    int nh, d1, d7;
    col_own<W, NCH, CR>(col[i], h[i], R, hole_lut, ho, nh, d1, d7);
    f1 += d1;
    f2 += nh;
    f7 += d7;
    hole_rows |= ho;                              // state.py:215
    f5 += col_rowtrans<W>(col[i], L, h[i], hL, nh_left);
    nh_left = nh;
    if (PACKW)
      f4 += col_wells_packed<W, NCH>(col[i], L, Rr, h[i], R, i == 0, i == C - 1,
                                     reinterpret_cast<const uint32_t*>(hole_lut + AfterLut::kWellsPack));
    else
      f4 += col_wells<W, NCH, CR>(col[i], L, Rr, h[i], R, i == 0, i == C - 1, hole_lut);
    if (TET_FENCE_EVERY > 0 && i % TET_FENCE_EVERY == TET_FENCE_EVERY - 1 && i + 1 < C) {
      TET_PIN(f1);
      TET_PIN(f2);
      TET_PIN(f4);
      TET_PIN(f5);
      TET_PIN(f7);
      TET_SCHED_FENCE();
    }
  }
  rows_with_holes = popc(hole_rows);  // state.py:274-275
  col_trans = f1;
  holes = f2;
  wells = f4;
  row_trans = f5;
  hole_depth = f7;
}

// state.py:97-107: the eight BCTS features as float32
template <typename W, int C, int NCH = 0, int CR = 12, bool PACKW = false>
TET_HD void bcts_features(const W (&col)[C], const int (&h)[C], int R, const uint8_t* hole_lut, int anchor_row,
                          int H, int eroded_cells, int n_cleared, float (&f)[8]) {
  int f0, f1, f2, f4, f5, f7;
  board_features<W, C, NCH, CR, PACKW>(col, h, R, hole_lut, f0, f1, f2, f4, f5, f7);
  f[0] = (float)f0;
  f[1] = (float)f1;
  f[2] = (float)f2;
  f[3] = (float)anchor_row + 0.5f * (float)(H - 1) + 1.0f;  // state.py:102; bonus = (H-1)/2
  f[4] = (float)f4;
  f[5] = (float)f5;
  f[6] = (float)(eroded_cells * n_cleared);                 // state.py:99-101
  f[7] = (float)f7;
}

// ---- valid-placement mask ---------------------------------------------------
// A placement is terminal iff a cell remains in row R after the clear
// (state.py:33,36,111-117).  The piece spans rows a..a+H-1 with a cell in every
// one of them and every cleared row lies inside that span, so
//      terminal  <=>  e := a + H - R  >  n_cleared.
// All columns c of one orientation are evaluated at once on C-bit column sets:
//  * slack s_c = R - h_c; footprint column j needs s_{c+j} >= need_j = H - b_j.
//    B_l = {c : s_c < l} for l = 1..4 (built with byte-parallel compares of the packed
//    heights) sit in 10-bit fields of one 64-bit level word Z = [0 | B_1 | B_2 | B_3 | B_4];
//    Z >> (10 (need_j - 1) + j) has B_{need_j - 1} >> j in bits 0-9 and B_{need_j} >> j in bits
//    10-19, so ONE shift per footprint column yields both "pokes >= 1 row above R-1" (I1, OR
//    over j of bits 10-19) and "pokes >= 2 rows" (I2, bits 0-9; level 0 is the empty field).
//    Bits of c + j >= 10 spill over from the next field: those placements do not exist and are
//    removed by fullmask.
//  * e = 1 (I1 & ~I2): the anchor is exactly a = R+1-H, so the piece's row rho sits
//    in board row R-3+t, t = rho+4-H.  That row becomes full iff all its missing
//    columns lie inside the piece's (contiguous) run [c+j0, c+j1] of that row, i.e.
//    c in [hi_t - j1, lo_t - j0] with lo_t/hi_t the lowest/highest missing column:
//    with X_t = {c >= hi_t} and Y_t = {c <= lo_t} that set is (X_t >> j1) & (Y_t >> j0).
//    One full row rescues the placement (n_cleared >= 1 = e).  Only rows below R can
//    be full (no stack cell sits at row >= R), so t <= 2.
//  * e >= 2 can only be rescued when H = 4 (vertical Straight): rows R-2 and R-1 must
//    both miss exactly column c (X_t & Y_t is that single column, or empty).
// stride of the level fields of valid_mask's level word: 10 bits up to ten columns (the paper's boards: the
// shifts below then stay in the range the round-1/2 kernels were tuned on), 12 bits for 11 and 12 columns
TET_HD constexpr int level_stride(int C) { return C <= 10 ? 10 : 12; }
template <int C> struct MissBits { typedef uint32_t type; };   // 3 bits per column: 32 bits up to 10 columns,
template <> struct MissBits<11> { typedef uint64_t type; };    // 64 beyond
template <> struct MissBits<12> { typedef uint64_t type; };
TET_HD int ctz_any(uint32_t x) { return __builtin_ctz(x); }
TET_HD int ctz_any(uint64_t x) { return __builtin_ctzll(x); }

template <typename W, int C>
TET_HD uint64_t valid_mask(const W (&col)[C], const int (&h)[C], const OrientEntry* tab, uint64_t fullmask, int R) {
  static_assert(C <= 12, "four 12-bit level fields + one empty field fill the 64-bit level word; 12-bit mask fields");
  constexpr int LS = level_stride(C);
  typedef typename MissBits<C>::type FT;
  uint32_t P[3] = {0u, 0u, 0u};
#pragma unroll
  for (int c = 0; c < C; ++c) P[c >> 2] |= (uint32_t)h[c] << (8 * (c & 3));
  uint32_t lv[4];
#pragma unroll
  for (int l = 1; l <= 4; ++l) {
    // byte b of P + K has bit 7 set iff h_b > R - l  (h <= 63, K <= 127: no carry between bytes)
    const uint32_t K = (uint32_t)(127 - (R - l)) * 0x01010101u;
    uint32_t g = 0;
#pragma unroll
    for (int q = 0; q < (C + 3) / 4; ++q)
      g |= ((((P[q] + K) & 0x80808080u) * 0x00204081u) >> 28) << (4 * q);  // gather the four bit-7s
    lv[l - 1] = g;
  }
  const uint64_t Z = ((uint64_t)lv[0] << LS) | ((uint64_t)lv[1] << (2 * LS)) | ((uint64_t)lv[2] << (3 * LS)) |
                     ((uint64_t)lv[3] << (4 * LS));
  // A placement that pokes above row R - 1 is rescued only by a row among R-3 .. R-1 that the piece
  // completes, i.e. one that misses at most four cells.  A cell in row R-3 or above means h >= R - 2
  // (level set 3), so when fewer than C - 4 columns reach that height no such row exists: the whole
  // rescue evaluation (a third of this function) is skipped -- by the wavefront, when none of its
  // envs needs it, which is the rule for boards that are not stacked to the top.
  const bool rescue = !TET_RESCUE_SKIP || TET_WAVE_ANY(popc(lv[2]) >= C - 4);
  uint32_t X[3] = {0u, 0u, 0u}, Y[3] = {0u, 0u, 0u};
  uint32_t rv1 = 0, rv2 = 0;
  if (rescue) {
    FT Fall = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) Fall |= (FT)((uint32_t)(col[c] >> (R - 3)) & 7u) << (3 * c);  // cells of rows R-3..R-1
    const FT Mall = (FT)~Fall;  // missing cells, 3 bits per column
    constexpr FT kEveryThird = (FT)0x9249249249249249ull & (FT)(((FT)1 << (3 * C)) - 1);  // bit 3c
    constexpr FT kTop = (FT)1 << (8 * sizeof(FT) - 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const FT m = (FT)(Mall >> t) & kEveryThird;                     // bit 3c: column c misses row R-3+t
      const int lo = (ctz_any((FT)(m | kTop)) * 11) >> 5;              // / 3
      const int hi = ((bitlen((FT)(m | 1)) - 1) * 11) >> 5;            // (bitlen: the top bit of m is never set)
      X[t] = ~0u << hi;
      Y[t] = m ? (2u << lo) - 1u : 0u;  // a full row (only on boards that were set from outside) rescues nothing
    }
    const uint32_t s0 = X[0] & Y[0], s1 = X[1] & Y[1], s2 = X[2] & Y[2];
    rv1 = s0 | s1 | s2;  // vertical Straight: any of its three lower rows
    rv2 = s1 & s2;       //                    / both of R-2, R-1
  }
  const uint32_t cm = (1u << C) - 1u;
  uint64_t mask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const OrientEntry& e = tab[k];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) r |= (uint32_t)(Z >> e.sh[j]);
    const uint32_t i1 = r >> LS, i2 = r;
    uint32_t v = ~i1;
    if (rescue) {
      // rescue by one cleared row (e = 1)
      uint32_t r1 = ((X[1] >> e.rj1[0]) & (Y[1] >> e.rj0[0])) | ((X[2] >> e.rj1[1]) & (Y[2] >> e.rj0[1]));
      r1 = (rv1 & e.vert4) | (r1 & ~e.vert4);
      const uint32_t r2 = rv2 & e.vert4;
      v |= ~(i2 & ~r2) & r1;
    }
    mask |= (uint64_t)(v & cm) << (kFieldStride * k);
  }
  return mask & fullmask;
}

// ---- the chosen placement (left column only known at run time) --------------
// Returns anchor row; stamps the piece into col; pbits[j] = cells added to
// footprint column j.  Written with arithmetic selects only (no per-lane array
// indexing): heights travel as packed bytes, the stamp as masked ORs.
template <typename W, int C>
TET_HD int stamp_dynamic(W (&col)[C], const int (&h)[C], int c, uint32_t d, W (&pbits)[4]) {
  static_assert(C <= 12, "heights are packed into three 32-bit words");
  uint32_t P[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < C; ++i) P[i >> 2] |= (uint32_t)h[i] << (8 * (i & 3));
  const int kq = c >> 2;
  const uint32_t plo = kq == 0 ? P[0] : (kq == 1 ? P[1] : P[2]);
  const uint32_t phi = kq == 0 ? P[1] : (kq == 1 ? P[2] : P[3]);
  const uint32_t win = (uint32_t)((((uint64_t)phi << 32) | plo) >> (8 * (c & 3)));  // h[c..c+3]
  const int w = d & 7;
  int a = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bj = (d >> (6 + 5 * j)) & 3;
    const int v = (int)((win >> (8 * j)) & 255u) - bj;
    a = (j < w && v > a) ? v : a;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bj = (d >> (6 + 5 * j)) & 3;
    const int nj = (j < w) ? (int)((d >> (8 + 5 * j)) & 7) : 0;
    pbits[j] = (W)(lowmask<W>(nj) << (a + bj));
  }
#pragma unroll
  for (int i = 0; i < C; ++i) {
    W add = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i - j >= 0) add |= (W)(pbits[j] & (W)(0 - (W)(c == i - j)));
    col[i] |= add;
  }
  return a;
}

// The same placement through a per-lane scratch column array (LDS on the device; element i of
// this lane at scratch[i * sstride]).  The runtime column index then addresses memory instead
// of selecting among registers: 10 stores, 4 loads, 4 ORs, 10 loads replace ~150 VALU selects,
// and only the four footprint heights are computed.  (The kernels are integer-VALU bound and
// the LDS pipe is otherwise idle.)
template <typename W, int C>
TET_HD int stamp_scratch(W (&col)[C], W* scratch, int sstride, int c, uint32_t d, W (&pbits)[4]) {
#pragma unroll
  for (int i = 0; i < C; ++i) scratch[i * sstride] = col[i];
  const int w = d & 7;
  int a = 0;
  int idx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    idx[j] = (c + j < C ? c + j : C - 1) * sstride;
    const int bj = (d >> (6 + 5 * j)) & 3;
    const int v = bitlen(scratch[idx[j]]) - bj;
    a = (j < w && v > a) ? v : a;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bj = (d >> (6 + 5 * j)) & 3;
    const int nj = (int)((d >> (8 + 5 * j)) & 7);  // 0 for j >= w (pack_orient leaves those fields empty)
    pbits[j] = (W)(lowmask<W>(nj) << (a + bj));    // one v_bfm_b32 on 32-bit boards
    TET_SCRATCH_OR(&scratch[idx[j]], pbits[j]);    // j >= w ORs zero (index clamped in range)
  }
#pragma unroll
  for (int i = 0; i < C; ++i) col[i] = scratch[i * sstride];
  return a;
}

// ---- all afterstates of one env (game.py:67-80) -----------------------------------------
// emit(has, field k = 2L + o, column c, f[8], row_all, row_valid, is_valid) is called for every placement
// of the current piece with the BCTS features of its afterstate; row_all = its index among all placements
// of the piece, row_valid = among the non-terminal ones (= the action, game.py:69; meaningful when
// is_valid) -- running counts here, because the walk follows the reference's order.  On the device the calls are WAVE-UNIFORM: every lane
// of the wavefront reaches every call (so the caller may cooperate across lanes, e.g. to merge
// stores) and `has` tells whether this lane's env really has the placement (f is garbage
// otherwise).  Slots are walked in the reference's enumeration order (loop L, column c,
// orientation o) so that consecutive calls of emit produce consecutive feature rows.
//
// The per-column terms of the current board are computed once (chunk tables); a placement that
// clears no line changes the board in at most four adjacent columns, and its features are those
// sums plus differences:
//  * footprint column j (piece cells in rows a+b_j .. a+b_j+n_j-1 above a column of height h_j): the
//    g_j = a + b_j - h_j cells left empty under the piece are new holes -- one more hole run if
//    g_j > 0 -- and the n_j new cells lie above every hole run of the column (state.py:200-239):
//        holes += g_j,  column transitions += 2 [g_j > 0],  hole depth += (u_j + [g_j > 0]) n_j
//    with u_j the hole-run tops of the old column, hole mask |= rows h_j .. a+b_j-1.  NO table;
//  * row transitions of the footprint columns and of the column to their right: col_rowtrans on the
//    new columns (a popcount each, no table);
//  * wells of the footprint columns and of both neighbours: col_wells_packed on the new columns, one
//    LDS read per 12-row chunk.
// Rounds 1-2 re-read the hole tables for every footprint column and three byte tables per chunk for
// the wells: 28 LDS reads per placement with 3-4-way bank conflicts on random indices, on the one LDS
// pipe four SIMDs share -- that pipe, not the vector ALUs, bounded the walk; this form issues 8-10.
// (Tried and dropped in round 3, all measured: the wells difference from a 4-row window table -- one read
// per column but twice the vector instructions; one lane per (env, column) with LDS-staged contiguous
// stores -- 5-9 x the wave instructions per env; the column loop rolled at run time with the arrays read in
// GPR-index mode -- a third of the code and 134 VGPRs, but slower at every occupancy.
// profiles/r03_experiments/.)
// Placements that complete a row (about 1 %) take the full evaluation afterwards.
// `lut` is an AfterLut.
template <typename W, int C, int NCH, typename Emit>
TET_HD void afterstates_env(const W (&col)[C], uint64_t meta, const SetTable& tab, const uint8_t* lut, int R,
                            Emit&& emit) {
  const uint32_t* pack = reinterpret_cast<const uint32_t*>(lut + AfterLut::kWellsPack);
  int h[C];
  heights_of<W, C>(col, h);
  const W wall = lowmask<W>(R + 4);
  const int piece = meta_piece(meta);
  const uint64_t full = tab.fullmask[piece];
  // per-column terms of the current board:  D = u | nh << 5 | rt << 11 | wells << 18
  // (hole-run tops <= 22, holes <= 43, row-transition term <= 88, wells <= 990 on 44 stored rows)
  W HO[C];
  uint32_t D[C];
  int sF1 = C;  // one unconditional transition per column (state.py:194)
  int sHoles = 0, sF7 = 0, sWells = 0;
  int sRT = R - popc(col[C - 1]);  // state.py:190
  {
    int nh_left = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const W L = (i == 0) ? wall : col[i - 1];
      const W Rr = (i == C - 1) ? wall : col[i + 1];
      const int hL = (i == 0) ? R : h[i - 1];
      int nh, e1, e7;
      col_own<W, NCH, 12>(col[i], h[i], R, lut, HO[i], nh, e1, e7);
      const int e5 = col_rowtrans<W>(col[i], L, h[i], hL, nh_left);
      const int e4 = col_wells_packed<W, NCH>(col[i], L, Rr, h[i], R, i == 0, i == C - 1, pack);
      nh_left = nh;
      D[i] = (uint32_t)(e1 >> 1) | ((uint32_t)nh << 5) | ((uint32_t)e5 << 11) | ((uint32_t)e4 << 18);
      sF1 += e1;
      sHoles += nh;
      sF7 += e7;
      sRT += e5;
      sWells += e4;
    }
  }
  uint64_t slow = 0;  // placements that clear lines: evaluated in full below
  const uint64_t valid = meta_mask(meta) & full;
  int row_all = 0, row_valid = 0;  // running row indices (the walk is in the reference's order)
#pragma unroll 1
  for (int L = 0; L < 2; ++L) {
    const uint32_t d0 = tab.orient[piece][2 * L].desc, d1 = tab.orient[piece][2 * L + 1].desc;
    if (!TET_WAVE_ANY((d0 | d1) >> 31)) continue;
    // widest footprint of each orientation over the wave: narrower pieces skip the extra columns
    const int wu0_ = 1 + (TET_WAVE_ANY((d0 & 7u) > 1) ? 1 : 0) + (TET_WAVE_ANY((d0 & 7u) > 2) ? 1 : 0) +
                    (TET_WAVE_ANY((d0 & 7u) > 3) ? 1 : 0);
    const int wu1_ = 1 + (TET_WAVE_ANY((d1 & 7u) > 1) ? 1 : 0) + (TET_WAVE_ANY((d1 & 7u) > 2) ? 1 : 0) +
                    (TET_WAVE_ANY((d1 & 7u) > 3) ? 1 : 0);
    const int wu0 = (TET_ABLATE & 1024) ? 4 : wu0_, wu1 = (TET_ABLATE & 1024) ? 4 : wu1_;
    // the two 12-bit fields of this loop, as 32-bit words: every per-placement test below is then a
    // constant-position bit test (no 64-bit variable shifts in the walk)
    const uint32_t fe0 = mask_field<C>(full, 2 * L), fe1 = mask_field<C>(full, 2 * L + 1);
    const uint32_t fv0 = mask_field<C>(valid, 2 * L), fv1 = mask_field<C>(valid, 2 * L + 1);
    uint32_t sl0 = 0, sl1 = 0;  // placements of this loop that clear lines
#pragma unroll
    for (int c = 0; c < C; ++c) {
      // AND of the columns / OR of the hole masks outside the window c .. c+3 (shared by both orientations)
      W a_ex = (W)~(W)0, o_ex = 0;
#pragma unroll
      for (int i = 0; i < C; ++i)
        if (i < c || i > c + 3) {
          a_ex &= col[i];
          o_ex |= HO[i];
        }

#pragma unroll
      for (int oi = 0; oi < 2; ++oi) {
        const int k = 2 * L + oi;
        const uint32_t dsc = oi ? d1 : d0;
        const int wu = oi ? wu1 : wu0;  // wave-uniform
        const int wd = (int)(dsc & 7u), H = (int)((dsc >> 3) & 7u);
        const bool ex = ((oi ? fe1 : fe0) >> c) & 1u;  // this lane's piece has this placement
        if (!(TET_ABLATE & 512) && !TET_WAVE_ANY(ex)) continue;
        const bool is_valid = ((oi ? fv1 : fv0) >> c) & 1u;
        const int my_row_all = row_all, my_row_valid = row_valid;
        row_all += ex ? 1 : 0;
        row_valid += is_valid ? 1 : 0;
        int a = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < C) {
            const int v = h[c + j] - (int)((dsc >> (6 + 5 * j)) & 3u);
            a = (j < wd && v > a) ? v : a;
          }
        // window position q = column c - 1 + q: new columns, heights, hole counts
        W XN[6];
        int HN[6], NHN[5];
        {
          const int cm1 = c >= 1 ? c - 1 : 0, c4 = c + 4 < C ? c + 4 : C - 1;
          XN[0] = (c >= 1) ? col[cm1] : wall;
          HN[0] = (c >= 1) ? h[cm1] : R;
          NHN[0] = (c >= 1) ? (int)((D[cm1] >> 5) & 63u) : 0;
          XN[5] = (c + 4 < C) ? col[c4] : wall;
          HN[5] = (c + 4 < C) ? h[c4] : R;
        }
        W hrows = o_ex, F = a_ex;
        int dholes = 0, dtops = 0, df7 = 0, dlast = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          constexpr int kz = 0;
          const int i = c + j < C ? c + j : kz;
          if (c + j < C) {
            // (all four window columns, also beyond the widest footprint of the wavefront: bounding this loop by
            // the wave-uniform width saves 4 % of the vector instructions and costs the stepping kernels 8-30 %
            // -- more scalar branches, worse register allocation at 128 VGPRs; measured, round 3)
            const int bj = (int)((dsc >> (6 + 5 * j)) & 3u), nj = (int)((dsc >> (8 + 5 * j)) & 7u);  // nj = 0 beyond the piece
            const int bot = (j < wd) ? a + bj : h[i];
            const int g = bot - h[i];  // cells left empty under the piece in this column
            XN[j + 1] = (W)(col[i] | (W)(lowmask<W>(nj) << bot));
            HN[j + 1] = bot + nj;
            hrows |= (W)(HO[i] | (W)(lowmask<W>(g) << h[i]));
            const int top = g > 0 ? 1 : 0;
            dholes += g;
            dtops += top;
            df7 += ((int)(D[i] & 31u) + top) * nj;
            NHN[j + 1] = (int)((D[i] >> 5) & 63u) + g;
            if (c + j == C - 1) dlast = nj;  // the right wall's term counts the cells of the last column
          } else {
            XN[j + 1] = wall;
            HN[j + 1] = R;
            NHN[j + 1] = 0;
          }
          F &= XN[j + 1];
        }
        F &= (W)((W)~(W)0 << a);  // rows of the piece only (see clear_lines)
        const bool fast = ex && F == 0;
        if (ex && F != 0) (oi ? sl1 : sl0) |= 1u << c;
        if (!(TET_ABLATE & 512) && !TET_WAVE_ANY(fast)) continue;
        // row transitions of columns c .. c+wu (the left neighbour of c+wu may have changed)
        int drt = 0;
#pragma unroll
        for (int q = 1; q <= 5; ++q)
          if (c + q - 1 < C && q <= wu + 1) {
            constexpr int kz = 0;
            const int i = c + q - 1 < C ? c + q - 1 : kz;
            drt += col_rowtrans<W>(XN[q], XN[q - 1], HN[q], HN[q - 1], NHN[q - 1]) - (int)((D[i] >> 11) & 127u);
          }
        // wells of columns c-1 .. c+wu
        int dwells = 0;
#pragma unroll
        for (int q = 0; q <= 5; ++q)
          if (c + q - 1 >= 0 && c + q - 1 < C && q <= wu + 1) {
            constexpr int kz = 0;
            const int i = (c + q - 1 >= 0 && c + q - 1 < C) ? c + q - 1 : kz;
            const int il = (i - 1 >= 0) ? i - 1 : kz, ir = (i + 1 < C) ? i + 1 : C - 1;
            const W Lq = (q >= 1) ? XN[q - 1] : ((i >= 1) ? col[il] : wall);
            const W Rq = (q <= 4) ? XN[q + 1] : ((i + 1 < C) ? col[ir] : wall);
            dwells += col_wells_packed<W, NCH>(XN[q], Lq, Rq, HN[q], R, i == 0, i == C - 1, pack) - (int)(D[i] >> 18);
          }
        float f[8];
        f[0] = (float)popc(hrows);
        f[1] = (float)(sF1 + 2 * dtops);
        f[2] = (float)(sHoles + dholes);
        f[3] = (float)a + 0.5f * (float)(H - 1) + 1.0f;  // state.py:102 with the pre-clear anchor row
        f[4] = (float)(sWells + dwells);
        f[5] = (float)(sRT + drt - dlast);
        f[6] = 0.0f;
        f[7] = (float)(sF7 + df7);
        emit(fast, k, c, f, my_row_all, my_row_valid, is_valid);
        TET_SCHED_FENCE();  // one placement at a time: interleaving the unrolled columns only costs registers
      }
    }
    slow |= ((uint64_t)sl0 << mask_bit(2 * L, 0)) | ((uint64_t)sl1 << mask_bit(2 * L + 1, 0));
  }
  while (TET_WAVE_ANY(slow != 0) && !(TET_ABLATE & 256)) {  // line-clearing placements (rare): full evaluation, state.py:33 onwards
    const bool has = slow != 0;  // lanes that are done keep pace on placement 0 (it always exists)
    const int sb = has ? bitlen(slow) - 1 : 0;
    slow &= ~(1ull << sb);
    W fb[C];
#pragma unroll
    for (int i = 0; i < C; ++i) fb[i] = col[i];
    W pbits[4];
    int fh[C];
    const int sk = sb / kFieldStride, sc = sb - sk * kFieldStride;
    const uint32_t od = tab.orient[piece][sk].desc;
    const int aa = stamp_dynamic<W, C>(fb, h, sc, od, pbits);
    int eroded = 0;
    const int kk = clear_lines<W, C>(fb, pbits, aa, &eroded);
    heights_of<W, C>(fb, fh);
    float f[8];
    bcts_features<W, C, NCH, 12, true>(fb, fh, R, lut, aa, (int)((od >> 3) & 7u), eroded, kk, f);
    emit(has, sk, sc, f, row_of_slot<C>(full, sk, sc), row_of_slot<C>(valid, sk, sc), ((valid >> sb) & 1) != 0);
  }
}

// Tetris.fitness (game.py:107-120): linear evaluation of one feature vector, left to right,
// in float32 (NumPy >= 2 keeps float32 * python-float in float32; the golden vectors were
// recorded that way).
TET_HD float fitness_of(const float (&f)[8], const float (&w)[8]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  float acc = TET_FMUL(f[0], w[0]);
#pragma unroll
  for (int q = 1; q < 8; ++q) acc = TET_FADD(acc, TET_FMUL(f[q], w[q]));
  return acc;
}

// ---- one env step (game.py:82-92) --------------------------------------------
struct StepOut {
  float obs[8];
  int action;
  int reward;
  int done;
  int lines;
  int n_valid;
  int piece;
  int invalid;
};

struct StepCfg {
  int R;
  int n_pieces;
  int auto_reset;
  uint32_t key_step;    // hash_key(seed, 4*step_idx + 0): piece draws of this step
  uint32_t key_policy;  // hash_key(seed, 4*step_idx + 3): built-in uniform random policy
  int compute_obs;      // 0: the caller passed obs = NULL (it already holds the afterstate features)
  float direct_by[8];   // state.py:49-50
  int has_direct_by;
};

// uniform random valid action (the random-rollout policy; also tetris_hip_policy_random)
TET_HD int policy_random(uint32_t key_policy, uint32_t env, int n_valid) {
  return scale16(hash_env(key_policy, env) >> 16, n_valid);
}

// `action` < 0 with use_policy: draw it with policy_random.  `draw` = replay piece for the
// step draw (or -1: use the bag), `draw_reset` = replay piece for the reset draw (or -1).
// PACKW: hole_lut is an AfterLut (kernels whose policy walks the afterstates) instead of a LutLayout<CR>
template <typename W, int C, int NCH = 0, int CR = 12, bool PACKW = false>
TET_HD void env_step(W (&col)[C], uint64_t& meta, int action, bool use_policy, const SetTable& tab,
                     const uint8_t* hole_lut, W* scratch, int sstride, const StepCfg& cfg, uint32_t env, int draw,
                     int draw_reset, StepOut& out) {
  const int R = cfg.R;
  const uint64_t mask = meta_mask(meta);
  int piece = meta_piece(meta);
  uint32_t bag = meta_bag(meta);
  const int nv = popc(mask);
  if (use_policy) action = policy_random(cfg.key_policy, env, nv);
  out.action = action;
  out.invalid = (action < 0 || action >= nv) ? 1 : 0;
  if (out.invalid) {  // game.py:83 raises IndexError; the env is left untouched
#pragma unroll
    for (int k = 0; k < 8; ++k) out.obs[k] = 0.0f;
    out.reward = 0;
    out.done = (nv == 0) ? 1 : 0;
    out.lines = 0;
    out.n_valid = nv;
    out.piece = piece;
    return;
  }
  // decode action -> (orientation field, left column): game.py:69,83
  int sk, c;
  slot_of_action_lut<C>(mask, action, hole_lut + (PACKW ? AfterLut::kSelPair : LutLayout<CR>::kSelPair), sk, c);
  const uint32_t od = tab.orient[piece][sk].desc;
  const int oH = (od >> 3) & 7;

  int h[C];
  W pbits[4];
  const int a = stamp_scratch<W, C>(col, scratch, sstride, c, od, pbits);  // tetromino.py get_after_states
  int eroded = 0;
  const int k = (TET_ABLATE & 8) ? 0 : clear_lines<W, C>(col, pbits, a, &eroded);   // state.py:33
  heights_of<W, C>(col, h);
  if ((TET_ABLATE & 1) || !cfg.compute_obs) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out.obs[i] = (float)h[i < C ? i : C - 1];  // placeholder: obs is not stored
  } else
  bcts_features<W, C, NCH, CR, PACKW>(col, h, R, hole_lut, a, oH, eroded, k, out.obs);  // game.py:91
  if (cfg.has_direct_by) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out.obs[i] *= cfg.direct_by[i];
  }
  // game.py:87 next piece, :88 is_game_over for THAT piece.  One hash feeds both draws of
  // the step: high 16 bits the step draw, low 16 bits the reset draw.
  const uint32_t rnd = hash_env(cfg.key_step, env);
  const uint8_t* sel_nib = hole_lut + (PACKW ? AfterLut::kSelNib : LutLayout<CR>::kSelNib);
  int np = draw >= 0 ? draw : bag_draw_lut(bag, cfg.n_pieces, rnd >> 16, sel_nib);
  uint64_t nmask = (TET_ABLATE & 2) ? (tab.fullmask[np] ^ (uint64_t)h[0])
                                    : valid_mask<W, C>(col, h, piece_entries(tab, np), tab.fullmask[np], R);
  int nnv = popc(nmask);
  int done = nnv == 0;
  out.reward = k - 1 + (done ? -100 : 0);  // game.py:86,89-90 (rewards :34-35)
  out.done = done;
  out.lines = k;
  if (done && cfg.auto_reset) {  // game.py:53-63 on the caller's behalf; the bag survives
#pragma unroll
    for (int i = 0; i < C; ++i) col[i] = 0;
    np = draw_reset >= 0 ? draw_reset : bag_draw_lut(bag, cfg.n_pieces, rnd & 0xFFFFu, sel_nib);
    nmask = tab.fullmask[np];
    nnv = popc(nmask);
  }
  out.n_valid = nnv;
  out.piece = np;
  meta = meta_pack(nmask, np, bag);
}

// ---- rollouts (game.py:129-160) --------------------------------------------------------------
// One rollout of `length` steps from (col, meta) starting with first action `a0`:
// returns -1 if the env is already over, dies on the first step or on any later one, else the
// sum of the rewards of steps 2..length (game.py:133-146; the first reward is not counted).
// policy 0: uniform random valid action; 1: greedy on the linear fitness `w` (first maximum).
// Pieces come from a fork of the env's bag driven by hash(key0 + t, uid), or from `fed` (one list
// index per step: what the reference's sampler handed out in a recorded run); the env is untouched.
template <typename W, int C, int NCH = 0>
TET_HD int rollout_env(const W (&col0)[C], uint64_t meta0, int a0, int length, int policy, const float (&w)[8],
                       const SetTable& tab, const uint8_t* hole_lut, W* scratch, int sstride, int R, int n_pieces,
                       uint32_t key0, uint32_t uid, const uint8_t* fed = nullptr) {
  W col[C];
#pragma unroll
  for (int i = 0; i < C; ++i) col[i] = col0[i];
  uint64_t meta = meta0;
  if (popc(meta_mask(meta)) == 0) return -1;  // game.py:132-133
  StepCfg cfg;
  cfg.R = R;
  cfg.n_pieces = n_pieces;
  cfg.auto_reset = 0;
  cfg.has_direct_by = 0;
  cfg.compute_obs = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) cfg.direct_by[i] = 1.0f;
  int ret = 0;
#pragma unroll 1
  for (int t = 0; t < length; ++t) {
    cfg.key_step = mix32(key0 + 2u * (uint32_t)t);
    cfg.key_policy = mix32(key0 + 2u * (uint32_t)t + 1u);
    int action = a0;
    bool use_policy = false;
    if (t > 0) {
      if (policy == 0) {
        use_policy = true;
        action = -1;
      } else {  // greedy: first non-terminal action of maximal fitness (game.py:102-120 on valid ones)
        float best = 0.f;
        int best_row = -1;
        afterstates_env<W, C, NCH>(col, meta, tab, hole_lut, R, [&](bool has, int, int, float (&f)[8], int, int row, bool is_valid) {
          if (!has) return;
          if (is_valid) {
            const float v = fitness_of(f, w);
            if (best_row < 0 || v > best || (v == best && row < best_row)) {
              best = v;
              best_row = row;
            }
          }
        });
        action = best_row;
      }
    }
    StepOut out;
    // fed: the piece each step draws comes from the caller (a recorded reference run) instead of the bag fork
    env_step<W, C, NCH, 12, true>(col, meta, action, use_policy, tab, hole_lut, scratch, sstride, cfg, uid,
                                  fed ? (int)fed[t] : -1, -1, out);
    if (out.done || out.invalid) return -1;  // game.py:135-138,143-145
    if (t > 0) ret += out.reward;
  }
  return ret;
}

}  // namespace tet
