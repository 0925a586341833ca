/*
 * tetris_oracle.c -- TEST INFRASTRUCTURE ONLY (see tetris_oracle.h).
 *
 * Plain-C restatement of the reference algorithm on the reference's own data
 * layout (a (R+4) x C cell array + a heights vector).  It deliberately does
 * NOT use bitboards: the HIP product computes everything on column bitboards,
 * so agreement between the two is agreement between two independent
 * formulations.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).
 *
 * Pinned by tests/test_oracle_golden.py against fixtures captured from the
 * live reference (tests/golden/make_golden.py).
 */
#include "tetris_oracle.h"

#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_version(void) { return 1; }

/* ---------------------------------------------------------------------------
 * Placement table.  One entry per orientation of every piece, grouped in the
 * "loops" of tetromino.py (orientations inside one loop are interleaved per
 * column).  For footprint column j: the piece occupies rows
 * anchor+b[j] .. anchor+b[j]+n[j]-1, and anchor = max_j(h[c+j] - b[j]).
 * n_changed / ppr / bonus are the changed_lines length,
 * pieces_per_changed_row and landing_height_bonus each State() call passes.
 * ------------------------------------------------------------------------- */
typedef struct {
  int w;
  int b[4];
  int n[4];
  int n_changed;
  int ppr[4];
  float bonus;
} Orient;

typedef struct {
  int n_loops;
  int n_orient[2];
  Orient o[2][2];
} PieceDef;

static const PieceDef PIECES[ORC_N_CATALOGUE] = {
    /* Straight: tetromino.py:44-57 (vertical), :60-74 (horizontal) */
    {2, {1, 1},
     {{{1, {0}, {4}, 4, {1, 1, 1, 1}, 1.5f}},
      {{4, {0, 0, 0, 0}, {1, 1, 1, 1}, 1, {4}, 0.0f}}}},
    /* Square: tetromino.py:90-103 */
    {1, {1, 0}, {{{2, {0, 0}, {2, 2}, 2, {2, 2}, 0.5f}}}},
    /* SnakeR: tetromino.py:120-135, :138-153 */
    {2, {1, 1},
     {{{3, {0, 0, 1}, {1, 2, 1}, 1, {2}, 0.5f}},
      {{2, {1, 0}, {2, 2}, 2, {1, 2}, 1.0f}}}},
    /* ThreeLine: tetromino.py:168-181, :184-198 */
    {2, {1, 1},
     {{{1, {0}, {3}, 3, {1, 1, 1}, 1.0f}},
      {{3, {0, 0, 0}, {1, 1, 1}, 1, {3}, 0.0f}}}},
    /* ThreeL: tetromino.py:214-247 (two per column), :250-281 (two per column) */
    {2, {2, 2},
     {{{2, {0, 0}, {1, 2}, 1, {2}, 0.5f}, {2, {0, 1}, {2, 1}, 2, {1, 2}, 0.5f}},
      {{2, {1, 0}, {1, 2}, 2, {1, 2}, 0.5f}, {2, {0, 0}, {2, 1}, 1, {2}, 0.5f}}}},
    /* SnakeL: tetromino.py:297-312, :315-330 */
    {2, {1, 1},
     {{{3, {1, 0, 0}, {1, 2, 1}, 1, {2}, 0.5f}},
      {{2, {0, 1}, {2, 2}, 2, {1, 2}, 1.0f}}}},
    /* T: tetromino.py:346-378, :381-413 */
    {2, {2, 2},
     {{{3, {0, 0, 0}, {1, 2, 1}, 1, {3}, 0.5f}, {3, {1, 0, 1}, {1, 2, 1}, 2, {1, 3}, 0.5f}},
      {{2, {1, 0}, {1, 3}, 2, {1, 2}, 1.0f}, {2, {0, 1}, {3, 1}, 2, {1, 2}, 1.0f}}}},
    /* RCorner: tetromino.py:429-460, :463-494 */
    {2, {2, 2},
     {{{3, {0, 0, 0}, {1, 1, 2}, 1, {3}, 0.5f}, {3, {0, 1, 1}, {2, 1, 1}, 2, {1, 3}, 0.5f}},
      {{2, {2, 0}, {1, 3}, 3, {1, 1, 2}, 1.0f}, {2, {0, 0}, {3, 1}, 1, {2}, 1.0f}}}},
    /* LCorner: tetromino.py:509-540, :543-575 */
    {2, {2, 2},
     {{{3, {0, 0, 0}, {2, 1, 1}, 1, {3}, 0.5f}, {3, {1, 1, 0}, {1, 1, 2}, 2, {1, 3}, 0.5f}},
      {{2, {0, 2}, {3, 1}, 3, {1, 1, 2}, 1.0f}, {2, {0, 0}, {1, 3}, 1, {2}, 1.0f}}}},
};

int orc_n_placements(int pid, int C) {
  const PieceDef* p = &PIECES[pid];
  int total = 0;
  for (int l = 0; l < p->n_loops; ++l) {
    int w = p->o[l][0].w;
    if (C - w + 1 > 0) total += (C - w + 1) * p->n_orient[l];
  }
  return total;
}

/* state.py:111-117 check_terminal: any cell in row index n_legal_rows */
static int check_terminal(const OrcState* s, int R, int C) {
  for (int c = 0; c < C; ++c)
    if (s->cells[R][c]) return 1;
  return 0;
}

/* state.py:121-143 clear_lines_jitted, including its height fix-up loop */
static void clear_lines(OrcState* s, int R, int C, int first_row, int n_changed) {
  int rows = R + 4;
  int is_full[4] = {0, 0, 0, 0};
  int n_cleared = 0;
  int last_cleared = -1;
  for (int k = 0; k < n_changed; ++k) { /* state.py:122-123 row sums */
    int sum = 0;
    for (int c = 0; c < C; ++c) sum += s->cells[first_row + k][c];
    is_full[k] = (sum == C);
    if (is_full[k]) {
      ++n_cleared;
      last_cleared = first_row + k; /* lines_to_clear[-1] */
    }
  }
  for (int k = 0; k < 4; ++k) s->cleared_rel[k] = (k < n_changed) ? is_full[k] : 0;
  s->n_cleared = n_cleared;
  if (n_cleared == 0) return;
  /* state.py:127-131: keep the other rows in order, zero rows appended on top */
  int dst = 0;
  for (int r = 0; r < rows; ++r) {
    int k = r - first_row;
    if (k >= 0 && k < n_changed && is_full[k]) continue;
    if (dst != r) memcpy(s->cells[dst], s->cells[r], sizeof(s->cells[0]));
    ++dst;
  }
  for (; dst < rows; ++dst) memset(s->cells[dst], 0, sizeof(s->cells[0]));
  /* state.py:132-142 per-column height repair */
  for (int c = 0; c < C; ++c) {
    int old = s->heights[c];
    if (old > last_cleared + 1) {
      s->heights[c] = old - n_cleared;
    } else {
      int lowest = 0;
      for (int r = old - n_cleared - 1; r >= 0; --r) {
        if (s->cells[r][c] == 1) {
          lowest = r + 1;
          break;
        }
      }
      s->heights[c] = lowest;
    }
  }
}

int orc_enumerate(const OrcDesc* d, const OrcState* cur, int pid, OrcState* out) {
  const int C = d->num_columns, R = d->num_rows;
  const PieceDef* p = &PIECES[pid];
  int count = 0;
  for (int l = 0; l < p->n_loops; ++l) {
    const int w = p->o[l][0].w;
    for (int c = 0; c + w <= C; ++c) {
      for (int oi = 0; oi < p->n_orient[l]; ++oi) {
        const Orient* o = &p->o[l][oi];
        OrcState* s = &out[count++];
        /* landing row, e.g. tetromino.py:122,140,217,234,253 */
        int a = -1000;
        for (int j = 0; j < w; ++j) {
          int v = cur->heights[c + j] - o->b[j];
          if (v > a) a = v;
        }
        /* representation.copy() + stamp + heights.copy() + set, e.g. :123-128 */
        memcpy(s->cells, cur->cells, sizeof(s->cells));
        memcpy(s->heights, cur->heights, sizeof(s->heights));
        for (int j = 0; j < w; ++j) {
          for (int k = 0; k < o->n[j]; ++k) s->cells[a + o->b[j] + k][c + j] = 1;
          s->heights[c + j] = a + o->b[j] + o->n[j];
        }
        /* State.__init__ : state.py:14-37 */
        s->anchor_col = c;
        s->anchor_row = a; /* changed_lines[0] */
        s->n_changed = o->n_changed;
        for (int k = 0; k < 4; ++k) s->pieces_per_changed_row[k] = (k < o->n_changed) ? o->ppr[k] : 0;
        s->bonus = o->bonus;
        clear_lines(s, R, C, a, o->n_changed); /* state.py:33 */
        s->terminal = check_terminal(s, R, C); /* state.py:36 */
      }
    }
  }
  return count;
}

/* state.py:175-280 get_feature_values_jitted; out = [rows_with_holes,
 * column_transitions, holes, cumulative_wells, row_transitions, hole_depth] */
static void board_features(const OrcState* s, int R, int C, int out[6]) {
  const int rows = R + 4;
  /* walls of ones over every stored row (state.py:177-178); heights of the
   * walls are num_rows = R (state.py:179) */
  int8_t rep[ORC_MAX_ROWS][ORC_MAX_COLS + 2];
  int hexp[ORC_MAX_COLS + 2];
  for (int r = 0; r < rows; ++r) {
    rep[r][0] = 1;
    for (int c = 0; c < C; ++c) rep[r][c + 1] = s->cells[r][c];
    rep[r][C + 1] = 1;
  }
  hexp[0] = R;
  for (int c = 0; c < C; ++c) hexp[c + 1] = s->heights[c];
  hexp[C + 1] = R;

  uint8_t row_has_hole[ORC_MAX_ROWS];
  memset(row_has_hole, 0, sizeof(row_has_hole));
  int column_transitions = 0, holes = 0, cumulative_wells = 0, row_transitions = 0, hole_depth = 0;

  { /* state.py:190 right-hand wall */
    int sum = 0;
    for (int r = 0; r < rows; ++r) sum += rep[r][C];
    row_transitions += R - sum;
  }

  for (int ci = 1; ci <= C; ++ci) { /* state.py:192 */
    const int lfr = s->heights[ci - 1];
    column_transitions += 1; /* :194 */
    int streak = 0;
    if (lfr > 0) { /* :197 */
      int full_above = 0;
      for (int r = 0; r < lfr; ++r) full_above += rep[r][ci]; /* :200 */
      if (hexp[ci - 1] > hexp[ci]) row_transitions += hexp[ci - 1] - hexp[ci]; /* :203-204 */
      int cell_below = 1; /* :206 */
      for (int r = 0; r < lfr; ++r) { /* :208 */
        int cell = rep[r][ci];
        if (cell == 0) {
          holes += 1;          /* :213 */
          row_has_hole[r] = 1; /* :215 */
          if (rep[r + 1][ci] == 1) hole_depth += full_above; /* :216 */
          if (cell_below) column_transitions += 1;           /* :219-220 */
          int left = rep[r][ci - 1], right = rep[r][ci + 1]; /* :223-224 */
          if (left) {
            row_transitions += 1;
            if (right) {
              streak += 1;
              cumulative_wells += streak;
            } else {
              streak = 0;
            }
          } else {
            streak = 0;
          }
        } else {
          streak = 0;      /* :236 */
          full_above -= 1; /* :239 */
          if (!cell_below) column_transitions += 1; /* :242-243 */
          if (!rep[r][ci - 1]) row_transitions += 1; /* :246-248 */
        }
        cell_below = cell; /* :251 */
      }
    } else { /* :253-254 */
      for (int r = 0; r < hexp[ci - 1]; ++r) row_transitions += rep[r][ci - 1];
    }
    /* :258-272 open wells above the column */
    int hl = hexp[ci - 1], hr = hexp[ci + 1];
    int top = hl < hr ? hl : hr;
    if (top > lfr) {
      for (int r = lfr; r < top; ++r) {
        int left = rep[r][ci - 1], right = rep[r][ci + 1];
        if (left) {
          if (right) {
            streak += 1;
            cumulative_wells += streak;
          } else {
            streak = 0;
          }
        } else {
          streak = 0;
        }
      }
    }
  }
  int rows_with_holes = 0; /* :274-275 */
  for (int r = 0; r < rows; ++r) rows_with_holes += row_has_hole[r];
  out[0] = rows_with_holes;
  out[1] = column_transitions;
  out[2] = holes;
  out[3] = cumulative_wells;
  out[4] = row_transitions;
  out[5] = hole_depth;
}

/* state.py:97-107 calc_bcts_features */
void orc_features(const OrcDesc* d, const OrcState* s, float f[8]) {
  int v[6];
  int eroded = 0, ncl = 0;
  for (int k = 0; k < s->n_changed; ++k) { /* :99-100 */
    eroded += s->cleared_rel[k] * s->pieces_per_changed_row[k];
    ncl += s->cleared_rel[k];
  }
  board_features(s, d->num_rows, d->num_columns, v);
  f[6] = (float)(eroded * ncl);                       /* :101 */
  f[3] = (float)s->anchor_row + s->bonus + 1.0f;      /* :102 */
  f[0] = (float)v[0];                                 /* :103 scatter [0,1,2,4,5,7] */
  f[1] = (float)v[1];
  f[2] = (float)v[2];
  f[4] = (float)v[3];
  f[5] = (float)v[4];
  f[7] = (float)v[5];
}

/* ----- flat front ends --------------------------------------------------- */

/* state.py:162-172 calc_lowest_free_rows */
static void heights_from_cells(OrcState* s, int R, int C) {
  for (int c = 0; c < C; ++c) {
    int lowest = 0;
    for (int r = R + 3; r >= 0; --r)
      if (s->cells[r][c] == 1) {
        lowest = r + 1;
        break;
      }
    s->heights[c] = lowest;
  }
}

static void state_from_flat(const OrcDesc* d, const int8_t* cells, OrcState* s) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  memset(s, 0, sizeof(*s));
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) s->cells[r][c] = cells[r * C + c];
  heights_from_cells(s, d->num_rows, C);
  /* reset-State defaults: changed_lines = arange(1), ppr = [0], bonus 0 (state.py:7-9) */
  s->n_changed = 1;
  s->anchor_row = 0;
}

static void state_to_flat(const OrcDesc* d, const OrcState* s, int8_t* cells) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) cells[r * C + c] = s->cells[r][c];
}

int orc_placements_flat(const OrcDesc* d, const int8_t* cells, int pid, int8_t* cells_out,
                        int32_t* heights_out, int32_t* n_cleared, int32_t* terminal,
                        int32_t* anchor_row, int32_t* anchor_col, float* feats) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  OrcState cur;
  static _Thread_local OrcState after[ORC_MAX_PLACEMENTS];
  state_from_flat(d, cells, &cur);
  int n = orc_enumerate(d, &cur, pid, after);
  for (int i = 0; i < n; ++i) {
    state_to_flat(d, &after[i], cells_out + (int64_t)i * rows * C);
    for (int c = 0; c < C; ++c) heights_out[i * C + c] = after[i].heights[c];
    n_cleared[i] = after[i].n_cleared;
    terminal[i] = after[i].terminal;
    anchor_row[i] = after[i].anchor_row;
    anchor_col[i] = after[i].anchor_col;
    orc_features(d, &after[i], feats + i * 8);
  }
  return n;
}

void orc_board_features_flat(const OrcDesc* d, const int8_t* cells, float f[8]) {
  OrcState s;
  state_from_flat(d, cells, &s);
  orc_features(d, &s, f);
}

/* ----- sampler ------------------------------------------------------------ */

static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

uint32_t orc_hash32(uint64_t seed, uint64_t env, uint64_t counter) {
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9U);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ (uint32_t)counter);
  k = mix32(k ^ (uint32_t)(counter >> 32));
  uint32_t h = (k ^ (uint32_t)env) * 0x9E3779B1U; /* global env index mod 2^32 */
  h ^= h >> 16;
  h *= 0x85EBCA6BU;
  h ^= h >> 13;
  return h;
}

/* Bag without replacement: same distribution as popping a fresh
 * np.random.permutation front to back (tetromino.py:17-22).  r16 = 16 random bits. */
int orc_bag_draw(uint16_t* bag, int n_pieces, uint32_t r16) {
  if (*bag == 0) *bag = (uint16_t)((1u << n_pieces) - 1u);
  int m = __builtin_popcount(*bag);
  int k = (int)((r16 * (uint32_t)m) >> 16);
  int p = -1;
  for (int i = 0; i < n_pieces; ++i) {
    if ((*bag >> i) & 1) {
      if (k == 0) {
        p = i;
        break;
      }
      --k;
    }
  }
  *bag = (uint16_t)(*bag & ~(1u << p));
  return p;
}

/* phase 0: draw inside step (high 16 bits of the step hash); phase 1: draw of the in-step
 * reset (low 16 bits of the same hash); phase 2: explicit reset (its own counter) */
static int next_piece(const OrcDesc* d, uint16_t* bag, const uint8_t* stream, int32_t* cursor,
                      int64_t stream_len, int64_t i, int64_t B, uint64_t seed, uint64_t env,
                      uint64_t step_idx, int phase) {
  if (stream) {
    int64_t row = cursor[i];
    if (row >= stream_len) row = stream_len - 1;
    cursor[i] += 1;
    return stream[row * B + i];
  }
  if (phase == 2) return orc_bag_draw(&bag[i], d->n_pieces, orc_hash32(seed, env, step_idx * 4u + 2u) >> 16);
  uint32_t h = orc_hash32(seed, env, step_idx * 4u);
  return orc_bag_draw(&bag[i], d->n_pieces, phase == 0 ? h >> 16 : h & 0xFFFFu);
}

/* ----- batched env -------------------------------------------------------- */

static int count_valid(const OrcDesc* d, const OrcState* st, int pid, OrcState* scratch) {
  int n = orc_enumerate(d, st, pid, scratch);
  int v = 0;
  for (int i = 0; i < n; ++i) v += !scratch[i].terminal;
  return v;
}

void orc_reset_batch(const OrcDesc* d, int8_t* cells, int32_t* piece, uint16_t* bag,
                     const uint8_t* stream, int32_t* cursor, int64_t stream_len,
                     uint8_t* n_valid, int init_bag, uint64_t seed, uint64_t step_idx,
                     int64_t env_offset, int64_t B) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  static _Thread_local OrcState scratch[ORC_MAX_PLACEMENTS];
  for (int64_t i = 0; i < B; ++i) {
    memset(cells + i * rows * C, 0, (size_t)rows * C);
    if (!stream && init_bag) bag[i] = 0;
    piece[i] = next_piece(d, bag, stream, cursor, stream_len, i, B, seed,
                          (uint64_t)(env_offset + i), step_idx, 2);
    OrcState s;
    state_from_flat(d, cells + i * rows * C, &s);
    n_valid[i] = (uint8_t)count_valid(d, &s, d->piece_ids[piece[i]], scratch);
  }
}

int64_t orc_step_batch(const OrcDesc* d, int8_t* cells, int32_t* piece, uint16_t* bag,
                       const int32_t* action, int32_t* action_out, const uint8_t* stream, int32_t* cursor,
                       int64_t stream_len, float* obs, int32_t* reward, uint8_t* done,
                       uint8_t* lines, uint8_t* n_valid_next, uint8_t* invalid, int auto_reset,
                       uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B,
                       int nthreads) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  int64_t n_invalid = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(+ : n_invalid) if (nthreads != 1)
#endif
  for (int64_t i = 0; i < B; ++i) {
    OrcState after[ORC_MAX_PLACEMENTS];
    OrcState scratch[ORC_MAX_PLACEMENTS];
    OrcState cur;
    int8_t* my = cells + i * rows * C;
    state_from_flat(d, my, &cur);
    /* game.py:68-69: enumerate, keep the non-terminal ones in order */
    int n = orc_enumerate(d, &cur, d->piece_ids[piece[i]], after);
    int n_cur = 0;
    for (int k = 0; k < n; ++k) n_cur += !after[k].terminal;
    /* action == NULL: the build's uniform random policy (16 hash bits scaled to n_valid) */
    int act = action ? action[i]
                     : (int)(((orc_hash32(seed, (uint64_t)(env_offset + i), step_idx * 4u + 3u) >> 16) *
                              (uint32_t)n_cur) >> 16);
    if (action_out) action_out[i] = act;
    int chosen = -1, seen = 0;
    for (int k = 0; k < n; ++k) {
      if (after[k].terminal) continue;
      if (seen == act) chosen = k;
      ++seen;
    }
    if (act < 0 || chosen < 0) { /* game.py:83 would raise IndexError: env untouched, outputs zeroed */
      invalid[i] = 1;
      ++n_invalid;
      for (int q = 0; q < 8; ++q) obs[i * 8 + q] = 0.0f;
      reward[i] = 0;
      done[i] = (uint8_t)(n_cur == 0);
      lines[i] = 0;
      n_valid_next[i] = (uint8_t)n_cur;
      continue;
    }
    invalid[i] = 0;
    const OrcState* ns = &after[chosen];     /* game.py:83 */
    int ncl = ns->n_cleared;                 /* :85 */
    int rew = ncl + (-1);                    /* :86, timestep_reward game.py:35 */
    int np_ = next_piece(d, bag, stream, cursor, stream_len, i, B, seed,
                         (uint64_t)(env_offset + i), step_idx, 0); /* :87 */
    int nv = count_valid(d, ns, d->piece_ids[np_], scratch);       /* :88, 94-100 */
    int dn = (nv == 0);
    if (dn) rew += -100;                     /* :89-90, loss_reward game.py:34 */
    orc_features(d, ns, obs + i * 8);        /* :91 */
    reward[i] = rew;
    done[i] = (uint8_t)dn;
    lines[i] = (uint8_t)ncl;
    state_to_flat(d, ns, my);
    piece[i] = np_;
    n_valid_next[i] = (uint8_t)nv;
    if (dn && auto_reset) { /* game.py:53-63 on the caller's behalf */
      memset(my, 0, (size_t)rows * C);
      piece[i] = next_piece(d, bag, stream, cursor, stream_len, i, B, seed,
                            (uint64_t)(env_offset + i), step_idx, 1);
      OrcState e;
      state_from_flat(d, my, &e);
      n_valid_next[i] = (uint8_t)count_valid(d, &e, d->piece_ids[piece[i]], scratch);
    }
  }
  return n_invalid;
}

void orc_afterstates_batch(const OrcDesc* d, const int8_t* cells, const int32_t* piece, int a_max,
                           float* feats, uint8_t* n_valid, float* feats_all, uint8_t* n_all,
                           int64_t B, int nthreads) {
  const int C = d->num_columns, rows = d->num_rows + 4;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) if (nthreads != 1)
#endif
  for (int64_t i = 0; i < B; ++i) {
    OrcState after[ORC_MAX_PLACEMENTS];
    OrcState cur;
    state_from_flat(d, cells + i * rows * C, &cur);
    int n = orc_enumerate(d, &cur, d->piece_ids[piece[i]], after);
    float* fv = feats + i * a_max * 8;
    memset(fv, 0, sizeof(float) * (size_t)a_max * 8);
    if (feats_all) memset(feats_all + i * a_max * 8, 0, sizeof(float) * (size_t)a_max * 8);
    int v = 0;
    for (int k = 0; k < n; ++k) {
      float f[8];
      orc_features(d, &after[k], f);
      if (feats_all && k < a_max) memcpy(feats_all + (i * a_max + k) * 8, f, sizeof(f));
      if (!after[k].terminal && v < a_max) { /* game.py:69-72 */
        memcpy(fv + v * 8, f, sizeof(f));
        ++v;
      }
    }
    n_valid[i] = (uint8_t)v;
    if (n_all) n_all[i] = (uint8_t)n;
  }
}

/* ----- NumPy legacy MT19937 (np.random.seed / np.random.permutation) -------
 * Published algorithm: Matsumoto & Nishimura MT19937; NumPy's legacy seeding
 * init_genrand(s) and shuffle = Fisher-Yates from the top with
 * rk_interval-style masked rejection on 32-bit outputs.
 * state[0..623] = key, state[624] = pos. */
void orc_mt_seed(uint32_t* st, uint32_t seed) {
  st[0] = seed;
  for (int i = 1; i < 624; ++i) st[i] = 1812433253U * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
  st[624] = 624;
}

uint32_t orc_mt_next(uint32_t* st) {
  if (st[624] >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (st[i] & 0x80000000U) | (st[(i + 1) % 624] & 0x7fffffffU);
      st[i] = st[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    st[624] = 0;
  }
  uint32_t y = st[st[624]++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= y >> 18;
  return y;
}

void orc_np_permutation(uint32_t* st, int n, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    uint32_t mask = (uint32_t)i;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
      v = orc_mt_next(st) & mask;
    } while (v > (uint32_t)i);
    int32_t t = out[i];
    out[i] = out[v];
    out[v] = t;
  }
}

/* ----- rollouts (game.py:129-160), with the build's per-rollout bag fork ------------------
 * returns[i][a] = mean over n of single_rollout(a): -1 when the env is over already or dies at
 * any step, else the sum of the rewards of steps 2..length.  policy 0 = the build's uniform
 * random policy, 1 = greedy on the float32 fitness of game.py:107-120 (first maximum among the
 * non-terminal placements). */
static uint32_t hash_key32(uint64_t seed, uint64_t counter) {
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9U);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ (uint32_t)counter);
  k = mix32(k ^ (uint32_t)(counter >> 32));
  return k;
}

static uint32_t hash_env32(uint32_t key, uint32_t env) {
  uint32_t h = (key ^ env) * 0x9E3779B1U;
  h ^= h >> 16;
  h *= 0x85EBCA6BU;
  h ^= h >> 13;
  return h;
}

static float fitness32(const float f[8], const float w[8]) {
  volatile float acc = f[0] * w[0]; /* every product and partial sum rounded to float32 */
  for (int q = 1; q < 8; ++q) {
    volatile float prod = f[q] * w[q];
    acc = acc + prod;
  }
  return acc;
}

void orc_rollouts_batch(const OrcDesc* d, const int8_t* cells, const int32_t* piece, const uint16_t* bag,
                        double* returns, int a_max, int length, int n, int policy, const float* weights,
                        uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B, int nthreads) {
  const int C = d->num_columns, rows = d->num_rows + 4;
  const uint32_t key = hash_key32(seed ^ 0x526F6C6C6F757473ULL, step_idx);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) if (nthreads != 1)
#endif
  for (int64_t i = 0; i < B; ++i) {
    OrcState after[ORC_MAX_PLACEMENTS];
    OrcState start;
    state_from_flat(d, cells + i * rows * C, &start);
    int n0 = orc_enumerate(d, &start, d->piece_ids[piece[i]], after);
    int nv0 = 0;
    for (int k = 0; k < n0; ++k) nv0 += !after[k].terminal;
    for (int a0 = 0; a0 < a_max; ++a0) {
      if (a0 >= nv0) {
        returns[i * a_max + a0] = __builtin_nan("");
        continue;
      }
      int sum = 0;
      for (int r = 0; r < n; ++r) {
        const uint64_t uid = ((uint64_t)(env_offset + i) * (uint64_t)a_max + (uint64_t)a0) * (uint64_t)n + (uint64_t)r;
        const uint32_t key0 = mix32(key ^ ((uint32_t)(uid >> 32) * 0x9E3779B1U));
        OrcState cur = start;
        int cur_piece = piece[i];
        uint16_t cur_bag = bag[i];
        int ret = 0;
        for (int t = 0; t < length; ++t) {
          const uint32_t key_step = mix32(key0 + 2u * (uint32_t)t), key_pol = mix32(key0 + 2u * (uint32_t)t + 1u);
          int m = orc_enumerate(d, &cur, d->piece_ids[cur_piece], after);
          int nv = 0;
          for (int k = 0; k < m; ++k) nv += !after[k].terminal;
          if (nv == 0) { /* cannot happen after a non-done step; covers the already-over start */
            ret = -1;
            break;
          }
          int act = a0;
          if (t > 0) {
            if (policy == 0) {
              act = (int)(((hash_env32(key_pol, (uint32_t)uid) >> 16) * (uint32_t)nv) >> 16);
            } else {
              float best = 0.f;
              int best_row = -1, row = 0;
              for (int k = 0; k < m; ++k) {
                if (after[k].terminal) continue;
                float f[8];
                orc_features(d, &after[k], f);
                float v = fitness32(f, weights);
                if (best_row < 0 || v > best) {
                  best = v;
                  best_row = row;
                }
                ++row;
              }
              act = best_row;
            }
          }
          int chosen = -1, seen = 0;
          for (int k = 0; k < m; ++k) {
            if (after[k].terminal) continue;
            if (seen == act) chosen = k;
            ++seen;
          }
          OrcState nxt = after[chosen];
          int np_ = orc_bag_draw(&cur_bag, d->n_pieces, hash_env32(key_step, (uint32_t)uid) >> 16);
          int nvn = 0;
          {
            OrcState tmp[ORC_MAX_PLACEMENTS];
            int mm = orc_enumerate(d, &nxt, d->piece_ids[np_], tmp);
            for (int k = 0; k < mm; ++k) nvn += !tmp[k].terminal;
          }
          int done = (nvn == 0);
          int rew = nxt.n_cleared - 1 + (done ? -100 : 0);
          if (done) {
            ret = -1;
            break;
          }
          if (t > 0) ret += rew;
          cur = nxt;
          cur_piece = np_;
        }
        sum += ret;
      }
      returns[i * a_max + a0] = (double)sum / (double)n;
    }
  }
}

