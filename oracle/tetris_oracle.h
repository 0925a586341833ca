/*
 * tetris_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, cell arrays, one env at a time) of the reference
 * s0phia-/tetris placement-level Tetris: game.py / state.py / tetromino.py.
 * It is the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under tetris_amd/ may include, link
 * or call it.
 *
 * Parity pin: the tests/golden .npz fixtures were produced by importing the reference
 * itself (tests/golden/make_golden.py) and tests/test_oracle_golden.py checks
 * every function below against them.
 *
 * Layout mirrors the reference: board = (R+4) x C cells, row 0 = bottom
 * (game.py:56, state.py:27-30); heights = lowest_free_rows (state.py:21-24).
 */
#ifndef TETRIS_ORACLE_H
#define TETRIS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_COLS 16
#define ORC_MAX_ROWS 64 /* stored rows = R + 4 */
#define ORC_MAX_PLACEMENTS 64
#define ORC_N_CATALOGUE 9

/* catalogue ids = class order in tetromino.py:33-576 */
enum {
  ORC_STRAIGHT = 0, ORC_SQUARE = 1, ORC_SNAKE_R = 2, ORC_THREE_LINE = 3,
  ORC_THREE_L = 4, ORC_SNAKE_L = 5, ORC_T = 6, ORC_R_CORNER = 7, ORC_L_CORNER = 8
};

typedef struct {
  int32_t num_columns;  /* C */
  int32_t num_rows;     /* R (legal rows); stored rows = R + 4 */
  int32_t n_pieces;     /* length of Tetris.tetrominos (game.py:38-39) */
  int32_t piece_ids[16];/* catalogue id of every list entry, list order */
} OrcDesc;

/* one State (state.py:5-38) */
typedef struct {
  int8_t cells[ORC_MAX_ROWS][ORC_MAX_COLS];
  int32_t heights[ORC_MAX_COLS];
  int32_t n_cleared;
  int32_t terminal;
  int32_t anchor_row;
  int32_t anchor_col;
  int32_t n_changed;
  int32_t pieces_per_changed_row[4];
  int32_t cleared_rel[4];
  float bonus;
} OrcState;

int orc_version(void);

/* number of raw placements of catalogue piece `pid` on C columns */
int orc_n_placements(int pid, int C);

/* tetromino.py <Piece>.get_after_states: all placements in reference order.
 * `out` must hold ORC_MAX_PLACEMENTS states.  Returns the count. */
int orc_enumerate(const OrcDesc* d, const OrcState* cur, int pid, OrcState* out);

/* state.py:97-107 + 175-280 */
void orc_features(const OrcDesc* d, const OrcState* s, float f[8]);

/* flat-array front ends used from Python (ctypes + numpy) ---------------- */

/* cells: int8 [(R+4)*C] row-major (row 0 = bottom).  Outputs are arrays of
 * length n (= return value): cells_out [n][(R+4)*C], heights_out [n][C],
 * n_cleared[n], terminal[n], anchor_row[n], anchor_col[n], feats[n][8]. */
int orc_placements_flat(const OrcDesc* d, const int8_t* cells, int pid,
                        int8_t* cells_out, int32_t* heights_out,
                        int32_t* n_cleared, int32_t* terminal,
                        int32_t* anchor_row, int32_t* anchor_col, float* feats);

/* features of a bare board as the reset State sees it (state.py:7-9 defaults) */
void orc_board_features_flat(const OrcDesc* d, const int8_t* cells, float f[8]);

/* ---------------------------------------------------------------------------
 * Batched env step with the build's sampler modes.
 *
 *  piece       : [B] list index of the current piece
 *  bag         : [B] bitmask of list indices still in the bag (device-bag mode)
 *  stream/cursor: replay mode when stream != NULL: stream is [stream_len][B]
 *                 (u8 list indices), cursor[B] = next unread row per env.
 *  action      : [B] index into the NON-TERMINAL placements (game.py:69,83); NULL = the
 *                build's uniform random policy; action_out (optional) receives it
 *  invalid     : [B] set to 1 where action is out of range (env untouched)
 * Mirrors game.py:82-92 (step), 94-100 (is_game_over), 53-63 (reset when
 * auto_reset and done).  Returns the number of invalid actions.
 * ------------------------------------------------------------------------- */
int64_t orc_step_batch(const OrcDesc* d, int8_t* cells, int32_t* piece,
                       uint16_t* bag, const int32_t* action, int32_t* action_out,
                       const uint8_t* stream, int32_t* cursor, int64_t stream_len,
                       float* obs, int32_t* reward, uint8_t* done, uint8_t* lines,
                       uint8_t* n_valid_next, uint8_t* invalid,
                       int auto_reset, uint64_t seed, uint64_t step_idx,
                       int64_t env_offset, int64_t B, int nthreads);

/* reset every env: empty board, draw first piece (game.py:53-63).  In device-
 * bag mode `bag` is (re)initialised only when init_bag != 0 (the reference bag
 * survives reset: game.py:50 vs 53-63). */
void orc_reset_batch(const OrcDesc* d, int8_t* cells, int32_t* piece,
                     uint16_t* bag, const uint8_t* stream, int32_t* cursor,
                     int64_t stream_len, uint8_t* n_valid, int init_bag,
                     uint64_t seed, uint64_t step_idx, int64_t env_offset,
                     int64_t B);

/* game.py:67-80 batched: feats [B][a_max][8] (rows >= n_valid zeroed),
 * n_valid[B]; feats_all [B][a_max][8] + n_all[B] when feats_all != NULL. */
void orc_afterstates_batch(const OrcDesc* d, const int8_t* cells,
                           const int32_t* piece, int a_max, float* feats,
                           uint8_t* n_valid, float* feats_all, uint8_t* n_all,
                           int64_t B, int nthreads);

/* game.py:129-160 perform_rollouts as a fan-out over (env, first action), with the build's
 * hash-driven policies / per-rollout bag fork.  returns [B][a_max] (NaN beyond n_valid). */
void orc_rollouts_batch(const OrcDesc* d, const int8_t* cells, const int32_t* piece, const uint16_t* bag,
                        double* returns, int a_max, int length, int n, int policy, const float* weights,
                        uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B, int nthreads);

/* counter-based bag draw shared with the HIP kernel (build design, not from
 * the reference; same distribution as tetromino.py:12-22). */
uint32_t orc_hash32(uint64_t seed, uint64_t env, uint64_t counter);
int orc_bag_draw(uint16_t* bag, int n_pieces, uint32_t r16);

/* NumPy legacy RandomState (MT19937) restatement, tetromino.py:15,19 call
 * np.random.permutation(n) on the global stream.  state = 625 uint32. */
void orc_mt_seed(uint32_t* state, uint32_t seed);
uint32_t orc_mt_next(uint32_t* state);
void orc_np_permutation(uint32_t* state, int n, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
