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

namespace tet {

constexpr int kMaxPieces = 12;   // pieces in one set (bag is 12 bits of meta)
constexpr int kMaxCols = 12;     // 4*C slot bits must fit the 48-bit mask
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

// ---- meta word ------------------------------------------------------------
// bits 0-47  valid mask over static slots s = L*2C + 2c + o (ascending s is
//            the reference enumeration order: loop, column, orientation)
// bits 48-51 current piece (list index, game.py:38-39)
// bits 52-63 bag: list indices still to be drawn (tetromino.py:12-22)
constexpr uint64_t kMaskBits = (1ull << 48) - 1;
TET_HD uint64_t meta_pack(uint64_t mask, int piece, uint32_t bag) {
  return (mask & kMaskBits) | ((uint64_t)(piece & 15) << 48) | ((uint64_t)(bag & 0xFFF) << 52);
}
TET_HD uint64_t meta_mask(uint64_t m) { return m & kMaskBits; }
TET_HD int meta_piece(uint64_t m) { return (int)((m >> 48) & 15); }
TET_HD uint32_t meta_bag(uint64_t m) { return (uint32_t)(m >> 52); }

// ---- word helpers -----------------------------------------------------------
TET_HD int popc(uint32_t x) { return __builtin_popcount(x); }
TET_HD int popc(uint64_t x) { return __builtin_popcountll(x); }
TET_HD int bitlen(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }
TET_HD int bitlen(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }
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
// env-independent part: computed once per launch on the host
inline uint32_t hash_key(uint64_t seed, uint64_t counter) {
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9U);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ (uint32_t)counter);
  k = mix32(k ^ (uint32_t)(counter >> 32));
  return k;
}
TET_HD uint32_t hash_env(uint32_t key, uint64_t env) {
  uint32_t h = mix32(key ^ (uint32_t)env);
  return mix32(h ^ (uint32_t)(env >> 32) ^ 0x85EBCA6BU);
}

// position of the k-th (0-based) set bit of x, k < popc(x)
TET_HD int select_bit(uint64_t x, int k) {
  int pos = 0;
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  int c = popc(lo);
  uint32_t v = lo;
  if (k >= c) { k -= c; pos = 32; v = hi; }
  c = popc(v & 0xFFFFu);
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

TET_HD int bag_draw(uint32_t& bag, int n_pieces, uint32_t key, uint64_t env) {
  if (bag == 0) bag = (1u << n_pieces) - 1u;
  int m = popc(bag);
  uint32_t r = hash_env(key, env);
  int k = (int)(((uint64_t)r * (uint64_t)(uint32_t)m) >> 32);
  int p = select_bit((uint64_t)bag, k);
  bag &= ~(1u << p);
  return p;
}

// ---- per-set table staged in LDS -------------------------------------------
struct SetTable {
  uint32_t orient[kMaxPieces][4];  // [list index][L*2+o]
  uint64_t fullmask[kMaxPieces];   // all existing slots (every placement valid)
};

template <typename W, int C>
struct Board {
  W col[C];
};

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
// Returns n_cleared; *eroded = piece cells that sat in cleared rows
// (state.py:99: sum(cleared_rows * pieces_per_changed_row)).
template <typename W, int C>
TET_HD int clear_lines(W (&col)[C], const W (&pbits)[4], int* eroded_cells) {
  W F = col[0];
#pragma unroll
  for (int i = 1; i < C; ++i) F &= col[i];
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

// state.py:175-280 in closed form.  out = f0,f1,f2,f4,f5,f7.
template <typename W, int C>
TET_HD void board_features(const W (&col)[C], const int (&h)[C], int R, int& rows_with_holes,
                           int& col_trans, int& holes, int& wells, int& row_trans, int& hole_depth) {
  const W wall = lowmask<W>(R + 4);  // walls of ones over every stored row (state.py:177-178)
  W hole_rows = 0;
  int f1 = 0, f2 = 0, f4 = 0, f7 = 0;
  int f5 = R - popc(col[C - 1]);  // state.py:190
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const W x = col[i];
    const int hi = h[i];
    const W mh = lowmask<W>(hi);
    const W L = (i == 0) ? wall : col[i - 1];
    const W Rr = (i == C - 1) ? wall : col[i + 1];
    const int hL = (i == 0) ? R : h[i - 1];       // state.py:179 wall height = num_rows
    const int hR = (i == C - 1) ? R : h[i + 1];
    const W ho = (W)(~x & mh);                    // holes (state.py:210-213)
    f2 += popc(ho);
    hole_rows |= ho;                              // state.py:215
    f1 += 1 + popc((W)((x ^ ((x << 1) | 1)) & mh));  // state.py:194,206,219-220,242-243
    // hole depth: the top hole of each vertical run counts the filled cells above it
    // (state.py:200,216,239)
    W T = (W)(ho & (x >> 1));
    while (T != 0) {
      int r1 = bitlen(T);  // (index of the top remaining hole) + 1
      f7 += popc((W)(x >> r1));
      T = (W)(T & lowmask<W>(r1 - 1));
    }
    // row transitions (state.py:203-204,223-226,246-248,253-254)
    f5 += popc((W)((x ^ L) & mh));
    int dl = hL - hi;
    f5 += (hi > 0) ? (dl > 0 ? dl : 0) : popc((W)(L & lowmask<W>(hL)));
    // wells (state.py:223-233 inside the column, :258-272 above it)
    int top = hL < hR ? hL : hR;
    W LR = (W)(L & Rr);
    W win = (W)(ho & LR);
    W open = (top > hi) ? (W)(lowmask<W>(top) & ~mh) : (W)0;
    W wopen = (W)(LR & open);
    W t;
    if (wopen == open) {  // both neighbours solid over the whole open range
      int d = top > hi ? top - hi : 0;
      f4 += (d * (d + 1)) >> 1;
      f4 += popc(win);
      t = (W)(win & (win >> 1));
    } else {
      W w = (W)(win | wopen);
      f4 += popc(w);
      t = (W)(w & (w >> 1));
    }
    while (t != 0) {  // runs of k consecutive rows add k(k+1)/2 in total
      f4 += popc(t);
      t = (W)(t & (t >> 1));
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
template <typename W, int C>
TET_HD void bcts_features(const W (&col)[C], const int (&h)[C], int R, int anchor_row, int H,
                          int eroded_cells, int n_cleared, float (&f)[8]) {
  int f0, f1, f2, f4, f5, f7;
  board_features<W, C>(col, h, R, f0, f1, f2, f4, f5, f7);
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
// one of them and all cleared rows lie inside that span, so
//      terminal  <=>  a + H - n_cleared > R.
// Fast form (exact whenever no row >= R-3 can become full, i.e. n_cleared = 0
// for every placement that pokes above R): valid <=> for all j: h[c+j] <= R-H+b_j.
template <typename W, int C>
TET_HD uint64_t valid_mask_fast(const int (&h)[C], const uint32_t (&d4)[4], int R) {
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int lo_ = 0; lo_ < 4; ++lo_) {
    const int L = lo_ >> 1, oi = lo_ & 1;
    const Orient o = unpack_orient(d4[lo_]);
    int thr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) thr[j] = (j < o.w) ? R - o.H + o.b[j] : 1000;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < C) ok = ok && (h[c + j] <= thr[j]);
      const int s = L * 2 * C + 2 * c + oi;
      if (s < 32) lo |= ok ? (1u << s) : 0u;
      else hi |= ok ? (1u << (s - 32)) : 0u;
    }
  }
  return ((uint64_t)hi << 32) | lo;
}

// Exact form: evaluates the clear for placements that poke above row R.
template <typename W, int C>
TET_HD uint64_t valid_mask_exact(const W (&col)[C], const int (&h)[C], const uint32_t (&d4)[4], int R) {
  uint64_t mask = 0;
#pragma unroll
  for (int lo_ = 0; lo_ < 4; ++lo_) {
    const int L = lo_ >> 1, oi = lo_ & 1;
    const Orient o = unpack_orient(d4[lo_]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      int a = -64;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < C) {
          int v = (j < o.w) ? h[c + j] - o.b[j] : -64;
          a = v > a ? v : a;
        }
      if (a < 0) a = 0;
      int top = a + o.H;
      bool ok = top <= R;
      if (!ok && top <= R + 4) {
        W F = (W)~(W)0;
#pragma unroll
        for (int i = 0; i < C; ++i) {
          W x = col[i];
          const int j = i - c;
          if (j >= 0 && j < 4) {
            int nj = (j < o.w) ? o.n[j] : 0;
            x |= (W)(lowmask<W>(nj) << (a + o.b[j]));
          }
          F &= x;
        }
        ok = (top - popc(F)) <= R;
      }
      const int s = L * 2 * C + 2 * c + oi;
      mask |= ok ? (1ull << s) : 0ull;
    }
  }
  return mask;
}

// lanes for which the fast form might be wrong: a row r >= R-3 can only become
// full if at least C-4 columns already reach above it.
template <typename W, int C>
TET_HD bool needs_exact_mask(const int (&h)[C], int R) {
  int tall = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) tall += (h[c] >= R - 2) ? 1 : 0;
  return tall >= C - 4;
}

template <typename W, int C>
TET_HD uint64_t valid_mask(const W (&col)[C], const int (&h)[C], const uint32_t (&d4)[4],
                           uint64_t fullmask, int R) {
  uint64_t m = valid_mask_fast<W, C>(h, d4, R);
  if (needs_exact_mask<W, C>(h, R)) m = valid_mask_exact<W, C>(col, h, d4, R);
  return m & fullmask;
}

// ---- the chosen placement (left column only known at run time) --------------
// Returns anchor row; stamps the piece into col; pbits[j] = cells added to
// footprint column j.
template <typename W, int C>
TET_HD int stamp_dynamic(W (&col)[C], const int (&h)[C], int c, const Orient& o, W (&pbits)[4]) {
  int a = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int j = i - c;
    int bj = (j == 0) ? o.b[0] : (j == 1) ? o.b[1] : (j == 2) ? o.b[2] : o.b[3];
    bool in = (j >= 0) && (j < o.w);
    int v = in ? h[i] - bj : 0;
    a = v > a ? v : a;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int nj = (j < o.w) ? o.n[j] : 0;
    pbits[j] = (W)(lowmask<W>(nj) << (a + o.b[j]));
  }
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int j = i - c;
    W add = (j == 0) ? pbits[0] : (j == 1) ? pbits[1] : (j == 2) ? pbits[2] : (j == 3) ? pbits[3] : (W)0;
    col[i] |= add;
  }
  return a;
}

// ---- one env step (game.py:82-92) --------------------------------------------
struct StepOut {
  float obs[8];
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
  uint32_t key_step;   // hash_key(seed, 4*step_idx + 0): draw inside step (game.py:87)
  uint32_t key_reset;  // hash_key(seed, 4*step_idx + 1): draw of the in-kernel reset (game.py:60)
  float direct_by[8];  // state.py:49-50
  int has_direct_by;
};

// `draw` = replay piece for the step draw (or -1: use the bag),
// `draw_reset` = replay piece for the reset draw (or -1).
template <typename W, int C>
TET_HD void env_step(W (&col)[C], uint64_t& meta, int action, const SetTable& tab, const StepCfg& cfg,
                     uint64_t env, int draw, int draw_reset, StepOut& out) {
  const int R = cfg.R;
  const uint64_t mask = meta_mask(meta);
  int piece = meta_piece(meta);
  uint32_t bag = meta_bag(meta);
  const int nv = popc(mask);
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
  // decode action -> (loop, column, orientation): game.py:69,83
  const int s = select_bit(mask, action);
  const int L = (s >= 2 * C) ? 1 : 0;
  const int q = s - L * 2 * C;
  const int c = q >> 1;
  const int oi = q & 1;
  const Orient o = unpack_orient(tab.orient[piece][L * 2 + oi]);

  int h[C];
  heights_of<W, C>(col, h);
  W pbits[4];
  const int a = stamp_dynamic<W, C>(col, h, c, o, pbits);  // tetromino.py get_after_states
  int eroded = 0;
  const int k = clear_lines<W, C>(col, pbits, &eroded);   // state.py:33
  heights_of<W, C>(col, h);
  bcts_features<W, C>(col, h, R, a, o.H, eroded, k, out.obs);  // game.py:91
  if (cfg.has_direct_by) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out.obs[i] *= cfg.direct_by[i];
  }
  // game.py:87 next piece, :88 is_game_over for THAT piece
  int np = draw >= 0 ? draw : bag_draw(bag, cfg.n_pieces, cfg.key_step, env);
  uint32_t d4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) d4[i] = tab.orient[np][i];
  uint64_t nmask = valid_mask<W, C>(col, h, d4, tab.fullmask[np], R);
  int nnv = popc(nmask);
  int done = nnv == 0;
  out.reward = k - 1 + (done ? -100 : 0);  // game.py:86,89-90 (rewards :34-35)
  out.done = done;
  out.lines = k;
  if (done && cfg.auto_reset) {  // game.py:53-63 on the caller's behalf; the bag survives
#pragma unroll
    for (int i = 0; i < C; ++i) col[i] = 0;
    np = draw_reset >= 0 ? draw_reset : bag_draw(bag, cfg.n_pieces, cfg.key_reset, env);
    nmask = tab.fullmask[np];
    nnv = popc(nmask);
  }
  out.n_valid = nnv;
  out.piece = np;
  meta = meta_pack(nmask, np, bag);
}

}  // namespace tet
