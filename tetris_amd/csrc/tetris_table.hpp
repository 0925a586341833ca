// tetris_table.hpp -- host-side piece catalogue and per-set table builder.
// Shared by the C-ABI (tetris_kernels.hip) and the CPU test harness
// (tests/harness); contains no device code.
#pragma once
#include <string.h>

#include "../../include/tetris_hip.h"
#include "tetris_core.hpp"

namespace tet {

// ---- piece catalogue (tetromino.py:33-576) -------------------
struct CatOrient { int w; int b[4]; int n[4]; };
struct CatPiece { int n_orient[2]; CatOrient o[2][2]; };

static const CatPiece kCatalogue[TETRIS_N_CATALOGUE] = {
    /* Straight  :44-57, :60-74 */
    {{1, 1}, {{{1, {0}, {4}}}, {{4, {0, 0, 0, 0}, {1, 1, 1, 1}}}}},
    /* Square    :90-103 */
    {{1, 0}, {{{2, {0, 0}, {2, 2}}}}},
    /* SnakeR    :120-135, :138-153 */
    {{1, 1}, {{{3, {0, 0, 1}, {1, 2, 1}}}, {{2, {1, 0}, {2, 2}}}}},
    /* ThreeLine :168-181, :184-198 */
    {{1, 1}, {{{1, {0}, {3}}}, {{3, {0, 0, 0}, {1, 1, 1}}}}},
    /* ThreeL    :214-247, :250-281 */
    {{2, 2}, {{{2, {0, 0}, {1, 2}}, {2, {0, 1}, {2, 1}}}, {{2, {1, 0}, {1, 2}}, {2, {0, 0}, {2, 1}}}}},
    /* SnakeL    :297-312, :315-330 */
    {{1, 1}, {{{3, {1, 0, 0}, {1, 2, 1}}}, {{2, {0, 1}, {2, 2}}}}},
    /* T         :346-378, :381-413 */
    {{2, 2}, {{{3, {0, 0, 0}, {1, 2, 1}}, {3, {1, 0, 1}, {1, 2, 1}}}, {{2, {1, 0}, {1, 3}}, {2, {0, 1}, {3, 1}}}}},
    /* RCorner   :429-460, :463-494 */
    {{2, 2}, {{{3, {0, 0, 0}, {1, 1, 2}}, {3, {0, 1, 1}, {2, 1, 1}}}, {{2, {2, 0}, {1, 3}}, {2, {0, 0}, {3, 1}}}}},
    /* LCorner   :509-540, :543-575 */
    {{2, 2}, {{{3, {0, 0, 0}, {2, 1, 1}}, {3, {1, 1, 0}, {1, 1, 2}}}, {{2, {0, 2}, {3, 1}}, {2, {0, 0}, {1, 3}}}}},
};

inline uint32_t pack_orient(const CatOrient& o) {
  int H = 0;
  uint32_t d = (uint32_t)o.w;
  for (int j = 0; j < o.w; ++j) {
    if (o.b[j] + o.n[j] > H) H = o.b[j] + o.n[j];
    d |= (uint32_t)o.b[j] << (6 + 5 * j);
    d |= (uint32_t)o.n[j] << (8 + 5 * j);
  }
  d |= (uint32_t)H << 3;
  d |= 1u << 31;
  return d;
}

inline int n_placements(int pid, int C) {
  const CatPiece& p = kCatalogue[pid];
  int total = 0;
  for (int l = 0; l < 2; ++l)
    if (p.n_orient[l] > 0 && C - p.o[l][0].w + 1 > 0) total += (C - p.o[l][0].w + 1) * p.n_orient[l];
  return total;
}

// thresholds: footprint column j needs slack R - h >= need_j = H - b_j.  valid_mask keeps the
// column sets B_l = {c : slack < l} in LS-bit fields of one 64-bit word (level l at bit LS l, LS =
// tet::level_stride(C): 10 up to ten columns, else 12); the shift LS (need_j - 1) + j lines column c + j
// of level need_j up with bit LS + c and of level need_j - 1 with bit c.
inline void pack_mask_fields(const CatOrient& o, OrientEntry* e, int C) {
  const int LS = level_stride(C);
  int H = 0;
  for (int j = 0; j < o.w; ++j)
    if (o.b[j] + o.n[j] > H) H = o.b[j] + o.n[j];
  for (int j = 0; j < o.w; ++j) e->sh[j] = (uint32_t)(LS * (H - o.b[j] - 1) + j);
  for (int j = o.w; j < 4; ++j) e->sh[j] = e->sh[0];  // absent columns repeat column 0's term (OR is idempotent)
  e->vert4 = (o.w == 1 && H == 4) ? ~0u : 0u;
  for (int t = 1; t < 3; ++t) {  // board row R-3+t holds piece row rho when the anchor is R+1-H
    const int rho = t - 4 + H;
    e->rj0[t - 1] = 31;  // no such row: the interval is empty
    e->rj1[t - 1] = 0;
    if (rho < 0 || rho > H - 2) continue;
    int j0 = -1, j1 = -1;
    for (int j = 0; j < o.w; ++j)
      if (o.b[j] <= rho && rho < o.b[j] + o.n[j]) {
        if (j0 < 0) j0 = j;
        j1 = j;
      }
    e->rj0[t - 1] = (uint32_t)j0;
    e->rj1[t - 1] = (uint32_t)j1;
  }
}

inline void build_table(const TetrisDesc* d, SetTable* t) {
  memset(t, 0, sizeof(*t));
  const int C = d->num_columns;
  for (int i = 0; i < d->n_pieces; ++i) {
    const CatPiece& p = kCatalogue[d->piece_ids[i]];
    uint64_t full = 0;
    for (int l = 0; l < 2; ++l)
      for (int oi = 0; oi < p.n_orient[l]; ++oi) {
        OrientEntry* e = &t->orient[i][l * 2 + oi];
        e->desc = pack_orient(p.o[l][oi]);
        pack_mask_fields(p.o[l][oi], e, C);
        for (int c = 0; c + p.o[l][oi].w <= C; ++c) full |= 1ull << mask_bit(2 * l + oi, c);
      }
    t->fullmask[i] = full;
  }
}

// num_columns values the kernels are instantiated for (one place for the
// library, the dispatch switch and the CPU test harness)
// (5 is the lower bound: with 4 columns a horizontal Straight IS a full row wherever it lands, even in the
// overflow rows, which the bit-parallel valid mask does not model; 12 the upper one: four 12-bit fields
// are the 48-bit valid mask of the control word, see valid_mask)
#ifndef TET_COLUMNS  // (experiment builds narrow this: -D'TET_COLUMNS(X)=X(10)')
#define TET_COLUMNS(X) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#endif

inline bool columns_supported(int C) {
  switch (C) {
#define TET_X(CC) case CC:
    TET_COLUMNS(TET_X)
#undef TET_X
    return true;
    default:
      return false;
  }
}

// text of the library's own (non-positive) return codes; NULL for anything else
inline const char* error_text(int code) {
  switch (code) {
    case TETRIS_OK: return "ok";
    case TETRIS_E_NULL: return "required pointer is NULL";
    case TETRIS_E_DESC: return "descriptor not initialised by tetris_hip_desc_init or inconsistent";
    case TETRIS_E_COLUMNS: return "num_columns not compiled into libtetris_hip";
    case TETRIS_E_ROWS: return "num_rows outside [4, 59]";
    case TETRIS_E_PIECES: return "bad piece list";
    case TETRIS_E_BATCH: return "batch size must be positive";
    case TETRIS_E_STREAM: return "replay stream needs cursor and stream_len > 0";
    case TETRIS_E_STRIDE: return "afterstate strides must be multiples of 4 floats and >= 8";
    default: return nullptr;
  }
}

inline int check_desc(const TetrisDesc* d) {
  if (!d) return TETRIS_E_NULL;
  if (d->abi_version != TETRIS_HIP_ABI_VERSION) return TETRIS_E_DESC;
  if (d->num_rows < 4 || d->num_rows > 59) return TETRIS_E_ROWS;
  if (d->word_bytes != ((d->num_rows + 4 <= 31) ? 4 : 8)) return TETRIS_E_DESC;
  if (d->n_pieces < 1 || d->n_pieces > TETRIS_MAX_PIECES) return TETRIS_E_PIECES;
  for (int i = 0; i < d->n_pieces; ++i)
    if (d->piece_ids[i] < 0 || d->piece_ids[i] >= TETRIS_N_CATALOGUE) return TETRIS_E_PIECES;
  return TETRIS_OK;
}


// game.py:21-51 constructor arguments -> descriptor
inline int desc_init(TetrisDesc* desc, int32_t num_columns, int32_t num_rows, const int32_t* piece_ids,
                     int32_t n_pieces, const float* direct_by) {
  if (!desc || !piece_ids) return TETRIS_E_NULL;
  memset(desc, 0, sizeof(*desc));
  if (!columns_supported(num_columns)) return TETRIS_E_COLUMNS;
  if (num_rows < 4 || num_rows > 59) return TETRIS_E_ROWS;
  if (n_pieces < 1 || n_pieces > TETRIS_MAX_PIECES) return TETRIS_E_PIECES;
  desc->abi_version = TETRIS_HIP_ABI_VERSION;
  desc->num_columns = num_columns;
  desc->num_rows = num_rows;
  desc->word_bytes = (num_rows + 4 <= 31) ? 4 : 8;
  desc->n_pieces = n_pieces;
  int amax = 0;
  for (int i = 0; i < n_pieces; ++i) {
    if (piece_ids[i] < 0 || piece_ids[i] >= TETRIS_N_CATALOGUE) return TETRIS_E_PIECES;
    desc->piece_ids[i] = piece_ids[i];
    int n = n_placements(piece_ids[i], num_columns);
    if (n > amax) amax = n;
  }
  desc->a_max = amax;
  if (direct_by) {
    desc->has_direct_by = 1;
    for (int i = 0; i < 8; ++i) desc->direct_by[i] = direct_by[i];
  }
  return TETRIS_OK;
}

}  // namespace tet
