// core_host.cpp -- TEST HARNESS ONLY: compiles the per-lane device logic of
// tetris_amd/csrc/tetris_core.hpp for the host (g++) so that the CPU test
// suite (and -fsanitize builds) can check it against the oracle without a GPU.
// It is not part of the product: nothing under tetris_amd/ loads it, and the
// product path (libtetris_hip.so) has no CPU fallback.
//
// Each function mirrors the per-env wrapper of the kernel of the same name in
// tetris_kernels.hip, looping over envs instead of lanes.
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "../../include/tetris_hip.h"
#include "../../tetris_amd/csrc/tetris_core.hpp"
#include "../../tetris_amd/csrc/tetris_table.hpp"

namespace {

struct alignas(16) FeatureLut {
  uint8_t bytes[tet::kFeatureLutBytes];
};
const FeatureLut kFeatureLutHost = {{
#include "../../tetris_amd/csrc/tetris_feature_lut.inc"
}};
const uint8_t* const kHoleLut = reinterpret_cast<const uint8_t*>(&kFeatureLutHost);
// tables of the kernels that walk the afterstates (tet::AfterLut)
struct alignas(16) AfterLutData {
  uint8_t bytes[tet::kAfterLutBytes];
};
const AfterLutData kAfterLutHost = {{
#include "../../tetris_amd/csrc/tetris_after_lut.inc"
}};
const uint8_t* const kAfterLut = reinterpret_cast<const uint8_t*>(&kAfterLutHost);

// boards in memory: the library's own packed / unpacked plane format (tet::board_packed)
template <typename W, int C>
void host_load(const W* planes, int64_t B, int64_t i, int R, W (&col)[C]) {
  if (tet::board_packed((int)sizeof(W), R))
    tet::load_board<W, C, true>(planes, B, i, col);
  else
    tet::load_board<W, C, false>(planes, B, i, col);
}
template <typename W, int C>
void host_store(W* planes, int64_t B, int64_t i, int R, const W (&col)[C]) {
  if (tet::board_packed((int)sizeof(W), R))
    tet::store_board<W, C, true>(planes, B, i, col);
  else
    tet::store_board<W, C, false>(planes, B, i, col);
}
// the 10-row-chunk set the stepping kernels use on u32 boards of up to 20 rows (LaunchStep)
struct alignas(16) FeatureLut10 {
  uint8_t bytes[tet::kFeatureLut10Bytes];
};
const FeatureLut10 kFeatureLut10Host = {{
#include "../../tetris_amd/csrc/tetris_feature_lut10.inc"
}};
const uint8_t* const kHoleLut10 = reinterpret_cast<const uint8_t*>(&kFeatureLut10Host);

template <typename W, int C>
void step_impl(const TetrisDesc* desc, void* cols_, uint64_t* meta, const int32_t* action, int32_t* action_out,
               const uint8_t* stream,
               int32_t* cursor, int64_t stream_len, float* obs, int32_t* reward, uint8_t* done, uint8_t* lines,
               uint8_t* n_valid, uint8_t* piece_next, uint32_t* status, int auto_reset, uint64_t seed,
               uint64_t step_idx, int64_t env_offset, int64_t B) {
  tet::SetTable tab;
  tet::build_table(desc, &tab);
  tet::StepCfg cfg;
  cfg.R = desc->num_rows;
  cfg.n_pieces = desc->n_pieces;
  cfg.auto_reset = auto_reset;
  cfg.key_step = tet::hash_key(seed, step_idx * 4u + 0u);
  cfg.key_policy = tet::hash_key(seed, step_idx * 4u + 3u);
  cfg.compute_obs = obs != nullptr;
  cfg.has_direct_by = desc->has_direct_by;
  for (int i = 0; i < 8; ++i) cfg.direct_by[i] = desc->direct_by[i];
  W* cols = static_cast<W*>(cols_);
  for (int64_t i = 0; i < B; ++i) {
    W col[C];
    host_load<W, C>(cols, B, i, desc->num_rows, col);
    uint64_t m = meta[i];
    int draw = -1, draw_reset = -1, cur = 0;
    bool exhausted = false;
    if (stream) {
      cur = cursor[i];
      exhausted = (int64_t)cur + (auto_reset ? 2 : 1) > stream_len || cur < 0;
      int64_t r0 = cur < stream_len ? cur : stream_len - 1;
      int64_t r1 = cur + 1 < stream_len ? cur + 1 : stream_len - 1;
      draw = stream[r0 * B + i];
      draw_reset = stream[r1 * B + i];
    }
    tet::StepOut out;
    W scratch[C];
    if (sizeof(W) == 4 && cfg.R <= 20)  // same dispatch as the library's LaunchStep
      tet::env_step<W, C, 2, 10>(col, m, (action && !exhausted) ? action[i] : -1, action == nullptr && !exhausted, tab, kHoleLut10, scratch, 1,
                                 cfg, (uint32_t)(env_offset + i), draw, draw_reset, out);
    else if (sizeof(W) == 8 && cfg.R <= 40)
      tet::env_step<W, C, 4, 10>(col, m, (action && !exhausted) ? action[i] : -1, action == nullptr && !exhausted, tab, kHoleLut10, scratch, 1,
                                 cfg, (uint32_t)(env_offset + i), draw, draw_reset, out);
    else
      tet::env_step<W, C>(col, m, (action && !exhausted) ? action[i] : -1, action == nullptr && !exhausted, tab, kHoleLut, scratch, 1, cfg,
                          (uint32_t)(env_offset + i), draw, draw_reset, out);
    if (action_out) action_out[i] = out.action;
    if (!out.invalid) {
      host_store<W, C>(cols, B, i, desc->num_rows, col);
      meta[i] = m;
      if (stream) cursor[i] = cur + 1 + ((out.done && auto_reset) ? 1 : 0);
    }
    if (obs)
      for (int k = 0; k < 8; ++k) obs[i * 8 + k] = out.obs[k];
    reward[i] = out.reward;
    done[i] = (uint8_t)out.done;
    lines[i] = (uint8_t)out.lines;
    n_valid[i] = (uint8_t)out.n_valid;
    if (piece_next) piece_next[i] = (uint8_t)out.piece;
    if (status) {  // one slot of 4 counters per 64 envs (= per wavefront on the GPU)
      uint32_t* slot = status + (i >> 6) * 4;
      slot[TETRIS_STATUS_INVALID] += out.invalid;
      if (!out.invalid) {
        slot[TETRIS_STATUS_EPISODES] += out.done;
        slot[TETRIS_STATUS_LINES] += out.lines;
        slot[TETRIS_STATUS_STEPS] += 1;
      }
    }
  }
}

template <typename W, int C>
void reset_impl(const TetrisDesc* desc, void* cols_, uint64_t* meta, const uint8_t* reset_mask, uint8_t* piece_out,
                uint8_t* n_valid_out, const uint8_t* stream, int32_t* cursor, int64_t stream_len, uint32_t* status,
                int init_bag, uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B) {
  tet::SetTable tab;
  tet::build_table(desc, &tab);
  const uint32_t key = tet::hash_key(seed, step_idx * 4u + 2u);
  W* cols = static_cast<W*>(cols_);
  for (int64_t i = 0; i < B; ++i) {
    if (reset_mask && !reset_mask[i]) continue;
    if (stream && (cursor[i] < 0 || cursor[i] >= stream_len)) {  // exhausted replay stream: untouched, counted
      if (status) status[(i >> 6) * 4 + TETRIS_STATUS_INVALID] += 1;
      continue;
    }
    {
      W zero[C] = {};
      host_store<W, C>(cols, B, i, desc->num_rows, zero);
    }
    uint32_t bag = init_bag ? 0u : tet::meta_bag(meta[i]);
    int piece;
    if (stream) {
      int cur = cursor[i];
      piece = stream[(int64_t)cur * B + i];
      cursor[i] = cur + 1;
    } else {
      piece = tet::bag_draw(bag, desc->n_pieces, tet::hash_env(key, (uint32_t)(env_offset + i)) >> 16);
    }
    const uint64_t mask = tab.fullmask[piece];
    meta[i] = tet::meta_pack(mask, piece, bag);
    if (piece_out) piece_out[i] = (uint8_t)piece;
    if (n_valid_out) n_valid_out[i] = (uint8_t)tet::popc(mask);
  }
}

template <typename W, int C>
void refresh_impl(const TetrisDesc* desc, const void* cols_, uint64_t* meta, uint8_t* n_valid_out, int64_t B) {
  tet::SetTable tab;
  tet::build_table(desc, &tab);
  const W* cols = static_cast<const W*>(cols_);
  for (int64_t i = 0; i < B; ++i) {
    W col[C];
    int h[C];
    host_load<W, C>(cols, B, i, desc->num_rows, col);
    tet::heights_of<W, C>(col, h);
    const int piece = tet::meta_piece(meta[i]);
    const uint64_t mask = tet::valid_mask<W, C>(col, h, tab.orient[piece], tab.fullmask[piece], desc->num_rows);
    meta[i] = tet::meta_pack(mask, piece, tet::meta_bag(meta[i]));
    if (n_valid_out) n_valid_out[i] = (uint8_t)tet::popc(mask);
  }
}

template <typename W, int C>
void after_impl(const TetrisDesc* desc, const void* cols_, const uint64_t* meta, float* feats, uint8_t* n_valid,
                float* feats_all, uint8_t* n_all, int64_t env_stride, int64_t rs, int64_t B) {
  tet::SetTable tab;
  tet::build_table(desc, &tab);
  const W* cols = static_cast<const W*>(cols_);
  const int R = desc->num_rows, a_max = desc->a_max;
  for (int64_t i = 0; i < B; ++i) {
    W col[C];
    int h[C];
    host_load<W, C>(cols, B, i, desc->num_rows, col);
    tet::heights_of<W, C>(col, h);
    const int piece = tet::meta_piece(meta[i]);
    const uint64_t full = tab.fullmask[piece];
    const uint64_t valid = tet::meta_mask(meta[i]) & full;
    float* out_valid = feats + i * env_stride;
    float* out_all = feats_all ? feats_all + i * env_stride : nullptr;
    for (int k = 0; k < a_max; ++k) {
      memset(out_valid + k * rs, 0, sizeof(float) * 8);
      if (out_all) memset(out_all + k * rs, 0, sizeof(float) * 8);
    }
    int nv = tet::popc(valid), na = tet::popc(full);
    bool consistent = true;
    tet::afterstates_env<W, C, 0>(col, meta[i], tab, kAfterLut, R, [&](bool has, int sk, int sc, float (&f)[8], int row_all, int row_valid, bool is_valid) {
          if (!has) return;
      const int s = tet::mask_bit(sk, sc);
      // (0) the running row indices of the walk against the popcount form
      consistent = consistent && row_all == tet::row_of_slot<C>(full, sk, sc) && is_valid == (bool)((valid >> s) & 1) &&
                   (!is_valid || row_valid == tet::row_of_slot<C>(valid, sk, sc));
      // cross-checks (test harness only): (1) the incremental features against the full
      // evaluation of the same placement, (2) the cached mask against the direct terminal test
      const tet::Orient o = tet::unpack_orient(tab.orient[piece][sk].desc);
      W nb[C];
      W pbits[4];
      int nh[C];
      const int a = tet::stamp_static<W, C>(col, h, sc, o, nb, pbits);
      int eroded = 0;
      const int k = tet::clear_lines<W, C>(nb, pbits, a, &eroded);
      tet::heights_of<W, C>(nb, nh);
      float g[8];
      tet::bcts_features<W, C>(nb, nh, R, kHoleLut, a, o.H, eroded, k, g);
      for (int q = 0; q < 8; ++q) consistent = consistent && (g[q] == f[q]);
      const bool terminal = (a + o.H - k) > R;  // state.py:36 after :33
      consistent = consistent && (terminal != (bool)((valid >> s) & 1));
      if (desc->has_direct_by)
        for (int q = 0; q < 8; ++q) f[q] *= desc->direct_by[q];
      if (out_all) memcpy(out_all + tet::row_of_slot<C>(full, sk, sc) * rs, f, sizeof(float) * 8);
      if ((valid >> s) & 1) memcpy(out_valid + tet::row_of_slot<C>(valid, sk, sc) * rs, f, sizeof(float) * 8);
    });
    if (!consistent) nv = 255;  // fail loudly in the tests
    n_valid[i] = (uint8_t)nv;
    if (n_all) n_all[i] = (uint8_t)na;
  }
}

template <typename F>
int dispatch(const TetrisDesc* desc, F&& f) {
  int rc = tet::check_desc(desc);
  if (rc) return rc;
  switch (desc->num_columns) {
#define TET_X(CC)                                                      \
  case CC:                                                             \
    if (desc->word_bytes == 4) f(uint32_t{}, std::integral_constant<int, CC>{}); \
    else f(uint64_t{}, std::integral_constant<int, CC>{});             \
    return 0;
    TET_COLUMNS(TET_X)
#undef TET_X
    default:
      return TETRIS_E_COLUMNS;
  }
}

}  // namespace

extern "C" {

int tetris_host_desc_init(TetrisDesc* desc, int32_t num_columns, int32_t num_rows, const int32_t* piece_ids,
                          int32_t n_pieces, const float* direct_by) {
  return tet::desc_init(desc, num_columns, num_rows, piece_ids, n_pieces, direct_by);
}

int tetris_host_step(const TetrisDesc* desc, void* cols, uint64_t* meta, const int32_t* action,
                     int32_t* action_out, const uint8_t* stream, int32_t* cursor, int64_t stream_len, float* obs, int32_t* reward,
                     uint8_t* done, uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next, uint32_t* status,
                     int32_t auto_reset, uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B,
                     void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto w, auto c) {
    step_impl<decltype(w), decltype(c)::value>(desc, cols, meta, action, action_out, stream, cursor, stream_len, obs, reward, done,
                                               lines, n_valid_next, piece_next, status, auto_reset, seed, step_idx,
                                               env_offset, B);
  });
}

// bound step call (mirrors tetris_hip_step_call_*): the arguments are kept and replayed
struct HostStepCall {
  TetrisDesc desc;
  void* cols; uint64_t* meta; int32_t* action_out; const uint8_t* stream; int32_t* cursor; int64_t stream_len;
  float* obs; int32_t* reward; uint8_t* done; uint8_t* lines; uint8_t* n_valid; uint8_t* piece; uint32_t* status;
  int32_t auto_reset; uint64_t seed; int64_t env_offset; int64_t B;
};
int64_t tetris_host_board_words(const TetrisDesc* desc, int64_t B) {
  const int rc = tet::check_desc(desc);
  if (rc) return rc;
  return tet::board_words(B, tet::n_planes(desc->num_columns, tet::board_packed(desc->word_bytes, desc->num_rows)));
}
int64_t tetris_host_step_call_size(void) { return (int64_t)sizeof(HostStepCall); }
int tetris_host_step_call_init(void* call_, const TetrisDesc* desc, void* cols, uint64_t* meta, int32_t* action_out,
                               const uint8_t* stream, int32_t* cursor, int64_t stream_len, float* obs, int32_t* reward,
                               uint8_t* done, uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next,
                               uint32_t* status, int32_t auto_reset, uint64_t seed, int64_t env_offset, int64_t B) {
  int rc = tet::check_desc(desc);
  if (rc) return rc;
  if (!call_ || !cols || !meta || !reward || !done || !lines || !n_valid_next) return TETRIS_E_NULL;
  HostStepCall c = {*desc, cols, meta, action_out, stream, cursor, stream_len, obs, reward, done, lines, n_valid_next,
                    piece_next, status, auto_reset, seed, env_offset, B};
  memcpy(call_, &c, sizeof(c));
  return 0;
}
int tetris_host_step_call_run(void* call_, const int32_t* action, uint64_t step_idx, void* unused) {
  HostStepCall c;
  memcpy(&c, call_, sizeof(c));
  return tetris_host_step(&c.desc, c.cols, c.meta, action, action ? nullptr : c.action_out, c.stream, c.cursor,
                          c.stream_len, c.obs, c.reward, c.done, c.lines, c.n_valid, c.piece, c.status, c.auto_reset,
                          c.seed, step_idx, c.env_offset, c.B, unused);
}

// the step + the payload of the done gather (done bitmask word per 64 envs, counter slots as of this step)
int tetris_host_step_call_run_gather(void* call_, const int32_t* action, uint64_t step_idx, uint64_t* done_bits,
                                     uint32_t* status_snapshot, void* unused) {
  HostStepCall c;
  memcpy(&c, call_, sizeof(c));
  const int rc = tetris_host_step_call_run(call_, action, step_idx, unused);
  if (rc) return rc;
  const int64_t n_words = (c.B + 63) / 64;
  if (done_bits) {
    for (int64_t wv = 0; wv < n_words; ++wv) done_bits[wv] = 0;
    for (int64_t i = 0; i < c.B; ++i)
      if (c.done[i]) done_bits[i >> 6] |= 1ull << (i & 63);
  }
  if (status_snapshot && c.status) memcpy(status_snapshot, c.status, (size_t)n_words * 16);
  return 0;
}
int tetris_host_stream_link(void* a, void* b) {
  (void)a;
  (void)b;
  return 0;
}
const char* tetris_host_source_hash(void) { return "harness"; }
const char* tetris_host_error_string(int code) {
  const char* own = tet::error_text(code);
  return own ? own : "unknown error";
}

int tetris_host_step_call_run_counted(void* call_, const int32_t* action, const uint64_t* step_counter, uint32_t step_rel,
                                      void* unused) {
  return tetris_host_step_call_run(call_, action, *step_counter + step_rel, unused);
}
int tetris_host_pack_done_bits(const uint8_t* done, uint8_t* bits, int64_t B, void* unused) {
  (void)unused;
  memset(bits, 0, (size_t)((B + 63) / 64) * 8);
  for (int64_t i = 0; i < B; ++i)
    if (done[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
  return 0;
}
int tetris_host_counter_add(uint64_t* counter, uint64_t n, void* unused) {
  (void)unused;
  *counter += n;
  return 0;
}

int tetris_host_reset(const TetrisDesc* desc, void* cols, uint64_t* meta, const uint8_t* reset_mask,
                      uint8_t* piece_out, uint8_t* n_valid_out, const uint8_t* stream, int32_t* cursor,
                      int64_t stream_len, uint32_t* status, int32_t init_bag, uint64_t seed, uint64_t step_idx,
                      int64_t env_offset, int64_t B, void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto w, auto c) {
    reset_impl<decltype(w), decltype(c)::value>(desc, cols, meta, reset_mask, piece_out, n_valid_out, stream, cursor,
                                                stream_len, status, init_bag, seed, step_idx, env_offset, B);
  });
}

int tetris_host_refresh(const TetrisDesc* desc, const void* cols, uint64_t* meta, uint8_t* n_valid_out, int64_t B,
                        void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto w, auto c) {
    refresh_impl<decltype(w), decltype(c)::value>(desc, cols, meta, n_valid_out, B);
  });
}

int tetris_host_afterstates(const TetrisDesc* desc, const void* cols, const uint64_t* meta, float* feats,
                            uint8_t* n_valid, float* feats_all, uint8_t* n_all, int64_t env_stride,
                            int64_t row_stride, int64_t B, void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto w, auto c) {
    after_impl<decltype(w), decltype(c)::value>(desc, cols, meta, feats, n_valid, feats_all, n_all, env_stride,
                                                row_stride, B);
  });
}

int tetris_host_policy_random(const uint8_t* n_valid, int32_t* action, uint64_t seed, uint64_t step_idx,
                              int64_t env_offset, int64_t B, void* unused) {
  (void)unused;
  const uint32_t key = tet::hash_key(seed, step_idx * 4u + 3u);
  for (int64_t i = 0; i < B; ++i) {
    action[i] = tet::policy_random(key, (uint32_t)(env_offset + i), n_valid[i]);
  }
  return 0;
}

int tetris_host_policy_greedy(const TetrisDesc* desc, const void* cols_, const uint64_t* meta, const float* weights,
                              int32_t* best_action, float* best_value, float* fitness_all, int64_t B, void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto wt, auto ct) {
    using W = decltype(wt);
    constexpr int C = decltype(ct)::value;
    tet::SetTable tab;
    tet::build_table(desc, &tab);
    float w[8];
    for (int q = 0; q < 8; ++q) w[q] = weights[q];
    const W* cols = static_cast<const W*>(cols_);
    for (int64_t i = 0; i < B; ++i) {
      W col[C];
      host_load<W, C>(cols, B, i, desc->num_rows, col);
      const int piece = tet::meta_piece(meta[i]);
      const uint64_t full = tab.fullmask[piece];
      const uint64_t valid = tet::meta_mask(meta[i]) & full;
      float* fall = fitness_all ? fitness_all + i * desc->a_max : nullptr;
      if (fall)
        for (int k = 0; k < desc->a_max; ++k) fall[k] = 0.f;
      float best = 0.f;
      int best_row = -1;
      tet::afterstates_env<W, C, 0>(col, meta[i], tab, kAfterLut, desc->num_rows, [&](bool has, int sk, int sc, float (&f)[8], int, int, bool) {
          if (!has) return;
        const float v = tet::fitness_of(f, w);
        if (fall) fall[tet::row_of_slot<C>(full, sk, sc)] = v;
        if ((valid >> tet::mask_bit(sk, sc)) & 1) {
          const int row = tet::row_of_slot<C>(valid, sk, sc);
          if (best_row < 0 || v > best || (v == best && row < best_row)) {
            best = v;
            best_row = row;
          }
        }
      });
      best_action[i] = best_row;
      if (best_value) best_value[i] = best;
    }
  });
}

int tetris_host_rollouts(const TetrisDesc* desc, const void* cols_, const uint64_t* meta, double* returns,
                         int32_t length, int32_t n, int32_t policy, const float* weights, const uint8_t* pieces,
                         uint64_t seed, uint64_t step_idx, int64_t env_offset, int64_t B, void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto wt, auto ct) {
    using W = decltype(wt);
    constexpr int C = decltype(ct)::value;
    tet::SetTable tab;
    tet::build_table(desc, &tab);
    float w[8];
    for (int q = 0; q < 8; ++q) w[q] = weights ? weights[q] : 0.f;
    const uint32_t key = tet::hash_key(seed ^ 0x526F6C6C6F757473ull, step_idx);
    const W* cols = static_cast<const W*>(cols_);
    const int a_max = desc->a_max;
    for (int64_t i = 0; i < B; ++i) {
      W col[C];
      host_load<W, C>(cols, B, i, desc->num_rows, col);
      const int nv = tet::popc(tet::meta_mask(meta[i]));
      for (int a0 = 0; a0 < a_max; ++a0) {
        double mean = __builtin_nan("");
        if (a0 < nv) {
          int sum = 0;
          for (int r = 0; r < n; ++r) {
            const uint64_t uid = ((uint64_t)(env_offset + i) * (uint64_t)a_max + (uint64_t)a0) * (uint64_t)n + r;
            const uint32_t key0 = tet::mix32(key ^ ((uint32_t)(uid >> 32) * 0x9E3779B1u));
            W scratch[C];
            const uint64_t fed = ((uint64_t)(i * a_max + a0) * (uint64_t)n + (uint64_t)r) * (uint64_t)length;
            sum += tet::rollout_env<W, C>(col, meta[i], a0, length, policy, w, tab, kAfterLut, scratch, 1,
                                          desc->num_rows, desc->n_pieces, key0, (uint32_t)uid,
                                          pieces ? pieces + fed : nullptr);
          }
          mean = (double)sum / (double)n;
        }
        returns[i * a_max + a0] = mean;
      }
    }
  });
}

int tetris_host_step_many(const TetrisDesc* desc, void* cols_, uint64_t* meta, int32_t n_steps, int32_t policy,
                          const float* weights, int32_t* action_out, float* obs, int32_t* reward, uint8_t* done,
                          uint8_t* lines, uint8_t* n_valid_next, uint8_t* piece_next, uint32_t* status,
                          int32_t auto_reset, uint64_t seed, uint64_t step_idx0, int64_t env_offset, int64_t B,
                          void* unused) {
  (void)unused;
  return dispatch(desc, [&](auto wt, auto ct) {
    using W = decltype(wt);
    constexpr int C = decltype(ct)::value;
    tet::SetTable tab;
    tet::build_table(desc, &tab);
    float w[8];
    for (int q = 0; q < 8; ++q) w[q] = weights ? weights[q] : 0.f;
    tet::StepCfg cfg;
    cfg.R = desc->num_rows;
    cfg.n_pieces = desc->n_pieces;
    cfg.auto_reset = auto_reset;
    cfg.has_direct_by = desc->has_direct_by;
    cfg.compute_obs = obs != nullptr;
    for (int q = 0; q < 8; ++q) cfg.direct_by[q] = desc->direct_by[q];
    W* cols = static_cast<W*>(cols_);
    for (int64_t i = 0; i < B; ++i) {
      W col[C];
      host_load<W, C>(cols, B, i, desc->num_rows, col);
      uint64_t m = meta[i];
      for (int k = 0; k < n_steps; ++k) {
        cfg.key_step = tet::hash_key(seed, (step_idx0 + k) * 4u + 0u);
        cfg.key_policy = tet::hash_key(seed, (step_idx0 + k) * 4u + 3u);
        int action = -1;
        bool use_policy = true;
        if (policy == 1) {
          const uint64_t valid = tet::meta_mask(m);
          float best = 0.f;
          int best_row = -1;
          tet::afterstates_env<W, C, 0>(col, m, tab, kAfterLut, cfg.R, [&](bool has, int sk, int sc, float (&f)[8], int, int, bool) {
          if (!has) return;
            if ((valid >> tet::mask_bit(sk, sc)) & 1) {
              const float v = tet::fitness_of(f, w);
              const int row = tet::row_of_slot<C>(valid, sk, sc);
              if (best_row < 0 || v > best || (v == best && row < best_row)) {
                best = v;
                best_row = row;
              }
            }
          });
          action = best_row;
          use_policy = false;
        }
        W scratch[C];
        tet::StepOut out;
        if (policy == 1)  // as the kernel: the greedy variant steps on the afterstate tables
          tet::env_step<W, C, 0, 12, true>(col, m, action, use_policy, tab, kAfterLut, scratch, 1, cfg,
                                           (uint32_t)(env_offset + i), -1, -1, out);
        else
          tet::env_step<W, C>(col, m, action, use_policy, tab, kHoleLut, scratch, 1, cfg, (uint32_t)(env_offset + i),
                              -1, -1, out);
        const int64_t e = (int64_t)k * B + i;
        if (obs)
          for (int q = 0; q < 8; ++q) obs[e * 8 + q] = out.obs[q];
        reward[e] = out.reward;
        done[e] = (uint8_t)out.done;
        lines[e] = (uint8_t)out.lines;
        n_valid_next[e] = (uint8_t)out.n_valid;
        if (piece_next) piece_next[e] = (uint8_t)out.piece;
        if (action_out) action_out[e] = out.action;
        if (status) {
          uint32_t* slot = status + (i >> 6) * 4;
          slot[TETRIS_STATUS_INVALID] += out.invalid;
          if (!out.invalid) {
            slot[TETRIS_STATUS_EPISODES] += out.done;
            slot[TETRIS_STATUS_LINES] += out.lines;
            slot[TETRIS_STATUS_STEPS] += 1;
          }
        }
      }
      host_store<W, C>(cols, B, i, desc->num_rows, col);
      meta[i] = m;
    }
  });
}

int tetris_host_version(void) { return TETRIS_HIP_ABI_VERSION; }

int tetris_host_n_planes(const TetrisDesc* desc) {
  const int rc = tet::check_desc(desc);
  if (rc) return rc;
  return tet::n_planes(desc->num_columns, tet::board_packed(desc->word_bytes, desc->num_rows));
}

int64_t tetris_host_status_words(int64_t B) { return B <= 0 ? 0 : 4 * (((B + 1023) / 1024) * 16); }

int tetris_host_n_placements(int32_t catalogue_id, int32_t num_columns) {
  if (catalogue_id < 0 || catalogue_id >= TETRIS_N_CATALOGUE) return TETRIS_E_PIECES;
  return tet::n_placements(catalogue_id, num_columns);
}

int tetris_host_numpy_bag_stream(const uint32_t* seeds, int32_t n_pieces, int64_t L, uint8_t* stream, int64_t B,
                                 void* unused) {
  (void)unused;
  if (!seeds || !stream) return TETRIS_E_NULL;
  for (int64_t i = 0; i < B; ++i) {  // same algorithm as numpy_bag_stream_kernel (tetris_kernels.hip)
    uint32_t mt[624];
    mt[0] = seeds[i];
    for (int k = 1; k < 624; ++k) mt[k] = 1812433253U * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
    int pos = 624;
    auto next_u32 = [&]() -> uint32_t {
      if (pos >= 624) {
        for (int k = 0; k < 624; ++k) {
          const uint32_t y = (mt[k] & 0x80000000U) | (mt[(k + 1) % 624] & 0x7fffffffU);
          mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
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
      if (left == 0) {
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
      stream[t * B + i] = bag[n_pieces - left];
      --left;
    }
  }
  return 0;
}

int tetris_host_decode(const TetrisDesc* desc, const void* cols, int8_t* cells, int32_t* heights, int64_t B,
                       void* unused) {
  (void)unused;
  const int C = desc->num_columns, rows = desc->num_rows + 4;
  const bool packed = tet::board_packed(desc->word_bytes, desc->num_rows);
  for (int64_t i = 0; i < B; ++i)
    for (int c = 0; c < C; ++c) {
      uint64_t x = desc->word_bytes == 4
                       ? tet::load_column_rt<uint32_t>(static_cast<const uint32_t*>(cols), B, i, c, C, packed)
                       : tet::load_column_rt<uint64_t>(static_cast<const uint64_t*>(cols), B, i, c, C, packed);
      if (heights) heights[i * C + c] = tet::bitlen(x);
      if (cells)
        for (int r = 0; r < rows; ++r) cells[(i * rows + r) * C + c] = (int8_t)((x >> r) & 1);
    }
  return 0;
}

int tetris_host_encode(const TetrisDesc* desc, const int8_t* cells, void* cols, int64_t B, void* unused) {
  (void)unused;
  const int C = desc->num_columns, rows = desc->num_rows + 4;
  const bool packed = tet::board_packed(desc->word_bytes, desc->num_rows);
  for (int64_t i = 0; i < B; ++i) {
    uint64_t col[tet::kMaxCols];
    uint32_t col32[tet::kMaxCols];
    for (int c = 0; c < C; ++c) {
      uint64_t x = 0;
      for (int r = 0; r < rows; ++r) x |= (uint64_t)(cells[(i * rows + r) * C + c] != 0) << r;
      col[c] = x;
      col32[c] = (uint32_t)x;
    }
    if (desc->word_bytes == 4)
      tet::store_columns_rt<uint32_t>(static_cast<uint32_t*>(cols), B, i, col32, C, packed);
    else
      tet::store_columns_rt<uint64_t>(static_cast<uint64_t*>(cols), B, i, col, C, packed);
  }
  return 0;
}

}  // extern "C"
