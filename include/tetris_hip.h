/*
 * tetris_hip.h -- C-ABI of libtetris_hip.so: the MI355X (gfx950) hot path of a
 * vectorised placement-level Tetris environment.
 *
 * The reference (s0phia-/tetris) is pure Python and has NO FFI / plugin
 * interface; the surface this library replaces is the Python one of
 * game.Tetris (+ state.State / tetromino.*).  Each entry point cites the
 * reference lines it stands in for.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add to game.py to bind them.
 *
 * Conventions
 *  - every pointer is DEVICE memory owned by the caller (PyTorch-ROCm tensors
 *    in the shipped host code); the library allocates nothing persistent,
 *    keeps no global state and never frees caller memory;
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream); no implicit synchronisation;
 *  - every function returns 0 on success, a negative TETRIS_E_* argument error
 *    (nothing was launched), or a positive hipError_t;
 *  - functions are re-entrant.
 *
 * Data layout (all [..] are element counts; B = batch)
 *  cols  : column bitboards: bit r of a column = cell (row r, column c), row 0 =
 *          bottom, rows 0..R+3 stored (game.py:56, state.py:27-30).  Word = uint32 if
 *          R+4 <= 31, uint64 if R+4 <= 63 (TetrisDesc.word_bytes).  Stored tile-major,
 *          word[ceil(B/64)][tetris_hip_n_planes(desc)][64]: envs in tiles of 64 (one
 *          wavefront), the planes of a tile back to back; one plane per column, or -- when
 *          the stored rows fit three quarters of the word -- the columns bit-packed four to
 *          three words (see tetris_hip_n_planes / tetris_hip_board_words below).
 *  meta  : uint64[B] per-env control word:
 *            bits  0-47 valid mask: four 12-bit fields, field 2L + o (loop L,
 *                       orientation o of tetromino.py's enumeration) at bit 12 (2L + o),
 *                       bit c of a field = left column c; action k (game.py:69,83) counts
 *                       the set bits of fields 0,1 interleaved by column, then of fields 2,3
 *            bits 48-51 current piece (index into the piece list, game.py:38-39)
 *            bits 52-63 bag: list indices still to be drawn (tetromino.py:12-22)
 */
#ifndef TETRIS_HIP_H
#define TETRIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TETRIS_HIP_ABI_VERSION 6

#define TETRIS_MAX_PIECES 12
#define TETRIS_MAX_COLUMNS 12
#define TETRIS_N_CATALOGUE 9

/* catalogue ids: class order of tetromino.py:33-576 */
enum {
  TETRIS_STRAIGHT = 0, TETRIS_SQUARE = 1, TETRIS_SNAKE_R = 2, TETRIS_THREE_LINE = 3,
  TETRIS_THREE_L = 4, TETRIS_SNAKE_L = 5, TETRIS_T = 6, TETRIS_R_CORNER = 7, TETRIS_L_CORNER = 8
};

enum {
  TETRIS_OK = 0,
  TETRIS_E_NULL = -1,        /* required pointer is NULL */
  TETRIS_E_DESC = -2,        /* descriptor not initialised / inconsistent */
  TETRIS_E_COLUMNS = -3,     /* num_columns not built into this library */
  TETRIS_E_ROWS = -4,        /* num_rows outside [4, 59] */
  TETRIS_E_PIECES = -5,      /* bad piece list */
  TETRIS_E_BATCH = -6,       /* B <= 0 (tetris_hip_step: or B beyond ~53 M envs per call: 32-bit byte offsets) */
  TETRIS_E_STREAM = -7,      /* replay stream given without cursor / length */
  TETRIS_E_STRIDE = -8       /* afterstate strides not multiples of 4 floats / too small / matrix beyond 2^32 float4 */
};

/* Constructor arguments of game.Tetris (game.py:21-23) that shape the kernels. */
typedef struct TetrisDesc {
  int32_t abi_version;       /* TETRIS_HIP_ABI_VERSION (set by tetris_hip_desc_init) */
  int32_t num_columns;       /* C */
  int32_t num_rows;          /* R = Tetris.num_rows (legal rows); stored rows = R+4 */
  int32_t word_bytes;        /* 4 or 8, derived from R */
  int32_t n_pieces;          /* len(Tetris.tetrominos) */
  int32_t piece_ids[TETRIS_MAX_PIECES]; /* catalogue id per list entry */
  int32_t a_max;             /* max raw placements of any piece in the set */
  int32_t has_direct_by;     /* feature_directions given (game.py:26, state.py:49-50) */
  float direct_by[8];
} TetrisDesc;

/* Counters the step kernel adds to.  To stay free of atomics every wavefront
 * owns one slot of 4 uint32: status is uint32[tetris_hip_status_words(B)] =
 * [n_waves][4], zeroed by the caller; the totals are the column sums. */
enum { TETRIS_STATUS_INVALID = 0, TETRIS_STATUS_EPISODES = 1, TETRIS_STATUS_LINES = 2, TETRIS_STATUS_STEPS = 3 };
int64_t tetris_hip_status_words(int64_t B);

/* Board storage.  `cols` is tile-major: word_bytes-wide words words[tile][p][lane] with tile =
 * env / 64, lane = env % 64, p < n_planes -- the state of one wavefront's 64 envs is one contiguous
 * record of n_planes * 64 words; the last tile is padded to 64 lanes.
 * When the stored rows num_rows + 4 fit three quarters of the word (24 bits of 4-byte, 48 bits of
 * 8-byte words: e.g. 10x20 and 10x40) the columns are 24- / 48-bit fields of one bit string, four
 * columns per three words, and ten columns take eight planes; otherwise plane c = column c.  The
 * caller allocates word_bytes * tetris_hip_board_words(desc, B) bytes (16-byte aligned) and treats
 * them as opaque: tetris_hip_decode / tetris_hip_encode convert to and from the reference layout.
 * A shard of whole tiles (a multiple of 64 envs) is a contiguous slice of the batch's storage. */
int tetris_hip_n_planes(const TetrisDesc* desc);
int64_t tetris_hip_board_words(const TetrisDesc* desc, int64_t B);

int tetris_hip_version(void);
const char* tetris_hip_error_string(int code);

/* number of columns values compiled into the library; fills out[] (<= 16) */
int tetris_hip_supported_columns(int32_t* out, int cap);

/* game.py:21-51 constructor arguments -> descriptor.  direct_by may be NULL. */
int tetris_hip_desc_init(TetrisDesc* desc, int32_t num_columns, int32_t num_rows,
                         const int32_t* piece_ids, int32_t n_pieces, const float* direct_by);

/* raw placement count of one catalogue piece (tetromino.py loops) */
int tetris_hip_n_placements(int32_t catalogue_id, int32_t num_columns);

/*
 * Tetris.reset (game.py:53-63): empty board, draw the first piece.
 *  reset_mask : uint8[B] or NULL (NULL = every env)
 *  piece_out / n_valid_out : uint8[B] or NULL, written for the envs that reset
 *  stream/cursor/stream_len : replay mode (piece list indices, [stream_len][B]
 *      plane-major, cursor int32[B] = next unread row); NULL = device bag
 *  status : per-wave counters (as tetris_hip_step) or NULL.  Replay mode: a reset consumes one stream
 *      row; an env whose cursor is at or past the end of the stream is left untouched and counted in
 *      status[TETRIS_STATUS_INVALID] (like a step past the end: the recorded game is never continued
 *      on a repeated last row)
 *  init_bag : also empty the bag (sampler construction, tetromino.py:13-15);
 *      0 keeps it, as the reference does across resets (game.py:50 vs 53-63)
 */
int tetris_hip_reset(const TetrisDesc* desc, void* cols, uint64_t* meta, const uint8_t* reset_mask,
                     uint8_t* piece_out, uint8_t* n_valid_out, const uint8_t* stream,
                     int32_t* cursor, int64_t stream_len, uint32_t* status, int32_t init_bag, uint64_t seed,
                     uint64_t step_idx, int64_t env_offset, int64_t B, void* hip_stream);

/*
 * Tetris.step (game.py:82-92) for every env in lockstep, fused with the
 * placement enumeration it depends on (game.py:67-69, tetromino.py
 * get_after_states), the line clear (state.py:121-143), is_game_over for the
 * next piece (game.py:94-100), the BCTS observation (state.py:97-107,175-280)
 * and, when auto_reset != 0, reset() of finished envs (game.py:53-63).
 *
 *  action       : int32[B] index into the non-terminal placements, or NULL: every env
 *                 plays a uniform random valid action (the same one
 *                 tetris_hip_policy_random would return for this seed / step_idx)
 *  action_out   : int32[B] or NULL: the action each env played
 *  obs          : float32[B][8]   observation of the chosen afterstate, or NULL to skip the
 *                 feature computation (a caller that ran tetris_hip_afterstates already holds it:
 *                 it is row `action` of that matrix)
 *  reward       : int32[B]        lines - 1 (- 100 when done)   game.py:86-90
 *  done         : uint8[B]
 *  lines        : uint8[B]
 *  n_valid_next : uint8[B]        non-terminal placements of the new piece
 *                                 (after auto-reset: of the fresh episode)
 *  piece_next   : uint8[B] or NULL  list index of the new current piece
 *  status       : uint32[tetris_hip_status_words(B)] or NULL: per-wave counters (TETRIS_STATUS_*)
 * An out-of-range action (game.py:83 raises IndexError) leaves that env
 * untouched, writes obs = 0, reward = 0, lines = 0 and counts it in
 * status[TETRIS_STATUS_INVALID].  Negative actions are out of range here (upstream's
 * `self.afterstates[action]` is NumPy indexing and accepts -n..-1; the single-env facade
 * tetris_amd.Tetris emulates that on the host).
 * Replay mode: a step consumes one stream row, two when it ends the episode under
 * auto_reset.  An env whose cursor cannot cover that (cursor + 1, or + 2 with auto_reset,
 * > stream_len) is treated exactly like an out-of-range action: untouched and counted as
 * invalid -- the recorded game is never continued with made-up pieces (tetris_hip_reset past the
 * end of the stream reports the env the same way).
 */
int tetris_hip_step(const TetrisDesc* desc, void* cols, uint64_t* meta, const int32_t* action,
                    int32_t* action_out, const uint8_t* stream, int32_t* cursor, int64_t stream_len, float* obs,
                    int32_t* reward, uint8_t* done, uint8_t* lines, uint8_t* n_valid_next,
                    uint8_t* piece_next, uint32_t* status, int32_t auto_reset, uint64_t seed,
                    uint64_t step_idx, int64_t env_offset, int64_t B, void* hip_stream);

/*
 * The same step as a BOUND CALL, for loops that step one batch many times (game.py:82-92 called
 * per step of an episode): tetris_hip_step_call_init validates the arguments and prepares, once,
 * everything that does not change between steps (pointers, geometry, the per-set placement table)
 * inside `call` -- caller-owned HOST memory of tetris_hip_step_call_size() bytes, 16-byte aligned;
 * the library still allocates nothing -- and tetris_hip_step_call_run only fills in the action
 * pointer and the step's hash keys and enqueues the kernel (about a third of the host time of
 * tetris_hip_step).  Semantics are exactly tetris_hip_step's with the bound arguments; action_out
 * is written only when action == NULL.  A call object is bound to its buffers: re-init after
 * re-allocating any of them.  One call object must not be run from two threads at once.
 */
int64_t tetris_hip_step_call_size(void);
int tetris_hip_step_call_init(void* call, const TetrisDesc* desc, void* cols, uint64_t* meta,
                              int32_t* action_out, const uint8_t* stream, int32_t* cursor,
                              int64_t stream_len, float* obs, int32_t* reward, uint8_t* done,
                              uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next,
                              uint32_t* status, int32_t auto_reset, uint64_t seed,
                              int64_t env_offset, int64_t B);
int tetris_hip_step_call_run(void* call, const int32_t* action, uint64_t step_idx, void* hip_stream);

/*
 * The bound step that also writes the payload of the done/reset gather between GPUs (the
 * path's only exchange) from its own epilogue, so that a gather costs the stepping stream no launch:
 *  done_bits       : uint64[ceil(B / 64)] or NULL: bit l of word w = done flag of env 64 w + l (the bytes
 *                    tetris_hip_pack_done_bits would produce from `done`)
 *  status_snapshot : uint32[tetris_hip_status_words(B)] or NULL: the per-wave counter slots as of THIS
 *                    step -- a consistent copy in memory no later step touches (the live `status` slots
 *                    keep moving)
 * Both buffers belong to the caller; hand each gather its own pair (double-buffer) and run the
 * collective on another stream behind tetris_hip_stream_link.
 */
int tetris_hip_step_call_run_gather(void* call, const int32_t* action, uint64_t step_idx, uint64_t* done_bits,
                                    uint32_t* status_snapshot, void* hip_stream);

/* `to_stream` waits for everything enqueued on `from_stream` so far, through an event that releases to
 * DEVICE scope (producer and consumer are kernels on this GPU; no system-scope cache write-back). */
int tetris_hip_stream_link(void* from_stream, void* to_stream);

/* hash of the kernel sources this library was built from (tetris_amd/build.py passes it to the
 * compiler): measurements are only quoted for the library that produced them */
const char* tetris_hip_source_hash(void);

/*
 * The bound step with its step index in DEVICE memory, for HIP graphs: the arguments of a captured
 * kernel launch are frozen at capture time, but every step needs fresh hash keys (piece draw,
 * built-in policy).  `step_counter` points to one uint64 on the device; this launch plays step
 * *step_counter + step_rel.  A graph of K steps captures K of these with step_rel = 0 .. K-1
 * followed by tetris_hip_counter_add(step_counter, K), so that every replay continues where the
 * last one stopped -- bit-identical to tetris_hip_step with the same indices.
 */
int tetris_hip_step_call_run_counted(void* call, const int32_t* action, const uint64_t* step_counter,
                                     uint32_t step_rel, void* hip_stream);
int tetris_hip_counter_add(uint64_t* counter, uint64_t n, void* hip_stream);

/*
 * K consecutive Tetris.step calls of every env in ONE launch, for policies that live in the
 * kernel (policy 0: uniform random valid action; 1: greedy on `weights`, HOST pointer to 8
 * floats).  Boards and meta stay in registers between the steps; every step's outputs go to
 * trajectory buffers indexed [k][i] (k-major): action_out int32[K][B] (or NULL), obs
 * float32[K][B][8] (or NULL), reward int32[K][B], done / lines / n_valid_next / piece_next
 * uint8[K][B] (piece_next may be NULL).  Bit-identical to K calls of tetris_hip_step with
 * action = NULL (policy 0) and step_idx = step_idx0 .. step_idx0 + K - 1.  K * B < 2^26.
 */
int tetris_hip_step_many(const TetrisDesc* desc, void* cols, uint64_t* meta, int32_t n_steps,
                         int32_t policy, const float* weights, int32_t* action_out, float* obs,
                         int32_t* reward, uint8_t* done, uint8_t* lines, uint8_t* n_valid_next,
                         uint8_t* piece_next, uint32_t* status, int32_t auto_reset, uint64_t seed,
                         uint64_t step_idx0, int64_t env_offset, int64_t B, void* hip_stream);

/*
 * Tetris.get_after_states (game.py:67-80): BCTS features of every placement
 * of the current piece.
 *  feats     : float32, row k of env i at feats + i*env_stride + k*row_stride (strides in
 *              floats, multiples of 4): row k = k-th NON-terminal placement, rows >= n_valid
 *              are zero.  env-major [B][a_max][8] = (a_max*8, 8) (fastest: a wave's
 *              rows stay within one 72 KiB span); action-major [a_max][B][8] = (8, B*8).
 *  n_valid   : uint8[B]
 *  feats_all : same layout or NULL: every placement in raw order
 *              (include_terminal=True, game.py:74-78)
 *  n_all     : uint8[B] or NULL
 */
int tetris_hip_afterstates(const TetrisDesc* desc, const void* cols, const uint64_t* meta,
                           float* feats, uint8_t* n_valid, float* feats_all, uint8_t* n_all,
                           int64_t env_stride, int64_t row_stride, int64_t B, void* hip_stream);

/*
 * Tetris.get_best_policy / fitness (game.py:102-120) batched: linear evaluation
 * sum_k features[k] * weights[k] (float32, left to right) of every placement.
 *  weights     : HOST pointer to 8 floats (game.py:111-118 uses -24.04 -19.77 -13.08 -12.63
 *                -10.49 -9.22 6.6 -1.61)
 *  best_action : int32[B]  first non-terminal action of maximal fitness (-1: none)
 *  best_value  : float32[B] or NULL
 *  fitness_all : float32[B][a_max] or NULL: every placement in raw order, terminal
 *                included (the domain of get_best_policy, game.py:103)
 */
int tetris_hip_policy_greedy(const TetrisDesc* desc, const void* cols, const uint64_t* meta,
                             const float* weights, int32_t* best_action, float* best_value,
                             float* fitness_all, int64_t B, void* hip_stream);

/*
 * Tetris.perform_rollouts / single_rollout (game.py:129-160) as a batch fan-out: for every env
 * and every valid first action a, `n` rollouts of `length` steps with the board kept in
 * registers.  returns[i][a] (double[B][a_max]) = mean over the n rollouts of the rollout return
 * (game.py:133-146: -1 if the env is already over or dies at any step, else the sum of the
 * rewards of steps 2..length); NaN for a >= n_valid.
 *  policy  : 0 uniform random valid action, 1 greedy on `weights` (HOST pointer to 8 floats)
 *  pieces  : NULL, or uint8[B][a_max][n][length]: the list index every step of every rollout draws
 *            (game.py:87 inside single_rollout) -- a recorded run of the reference's sampler, for
 *            exact replays of perform_rollouts on multi-piece sets
 * cols / meta are not modified (the reference restores the env after every rollout but lets
 * its global bag advance; with pieces == NULL every rollout draws from its own fork of the env's bag).
 */
int tetris_hip_rollouts(const TetrisDesc* desc, const void* cols, const uint64_t* meta,
                        double* returns, int32_t length, int32_t n, int32_t policy,
                        const float* weights, const uint8_t* pieces, uint64_t seed, uint64_t step_idx,
                        int64_t env_offset, int64_t B, void* hip_stream);

/*
 * The reference's TetrominoSampler on NumPy's legacy global stream, on the device (tetromino.py:12-22
 * over np.random.seed / np.random.permutation): stream[t][i] (uint8 [L][B], the layout of the replay
 * `stream` of tetris_hip_reset / tetris_hip_step) = the list index the sampler of a reference game
 * constructed right after np.random.seed(seeds[i]) returns at its t-th call (call 0 = the draw of
 * Tetris.__init__'s reset, then one per step and one per reset).  seeds: uint32[B] on the device.
 * With it, B envs replay B independently seeded reference games piece for piece; large batches
 * use the counter-based bag instead (stream = NULL).
 */
int tetris_hip_numpy_bag_stream(const uint32_t* seeds, int32_t n_pieces, int64_t L, uint8_t* stream,
                                int64_t B, void* hip_stream);

/* done flags (uint8[B], as written by tetris_hip_step) -> bitmask, bit i % 8 of byte i / 8: the payload of
 * the done/reset gather between GPUs (one all-gather of B / 8 bytes per rank).  `bits` must hold
 * 8 * ceil(B / 64) bytes, 8-byte aligned (whole 64-env words are written). */
int tetris_hip_pack_done_bits(const uint8_t* done, uint8_t* bits, int64_t B, void* hip_stream);

/* uniform random valid action per env: floor(u * n_valid), u from the
 * counter-based hash (the probe policy of SURVEY section 6 / example_play) */
int tetris_hip_policy_random(const uint8_t* n_valid, int32_t* action, uint64_t seed,
                             uint64_t step_idx, int64_t env_offset, int64_t B, void* hip_stream);

/* bitboards -> reference layout: cells int8[B][R+4][C] (State.representation,
 * state.py:14) and heights int32[B][C] (lowest_free_rows, state.py:21-24);
 * either output may be NULL. */
int tetris_hip_decode(const TetrisDesc* desc, const void* cols, int8_t* cells, int32_t* heights,
                      int64_t B, void* hip_stream);

/* reference layout -> bitboards (inverse of tetris_hip_decode).  Boards handed to
 * tetris_hip_refresh / tetris_hip_step / tetris_hip_afterstates must be states a game can be in:
 * no cell at row >= R.  (A board with such a cell is a terminal State, state.py:33-36, which the
 * reference never steps from, game.py:69; the placement mask and the stepping kernels' feature
 * tables assume there is none.  The Python host refuses such boards in set_boards.) */
int tetris_hip_encode(const TetrisDesc* desc, const int8_t* cells, void* cols, int64_t B,
                      void* hip_stream);

/* recompute meta's valid mask for the piece stored in meta (used after the
 * caller edits cols / meta directly, e.g. to restore a snapshot or set a board) */
int tetris_hip_refresh(const TetrisDesc* desc, const void* cols, uint64_t* meta,
                       uint8_t* n_valid_out, int64_t B, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
