// MemorySanitizer driver for the per-lane logic (CPU only; not part of the test suite):
//   /opt/rocm/lib/llvm/bin/clang++ -fsanitize=memory -fsanitize-memory-track-origins -O1 -g -std=c++17 msan_main.cpp -o msan_main
// Runs every entry of the afterstate family on boards of several widths; any use of an
// uninitialised value in a branch, address or output is reported.
#include "core_host.cpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

static int run(int C, int R, const std::vector<int32_t>& ids, int64_t B) {
  TetrisDesc d;
  memset(&d, 0, sizeof d);
  if (tetris_host_desc_init(&d, C, R, ids.data(), (int32_t)ids.size(), nullptr) != 0) return 1;
  const int64_t words = tetris_host_board_words(&d, B);
  std::vector<uint64_t> cols(words + 8, 0), meta(B, 0);
  std::vector<uint8_t> piece(B, 0), nv(B, 0), done(B * 40, 0), lines(B * 40, 0), nvn(B * 40, 0), pn(B * 40, 0);
  std::vector<uint32_t> status(tetris_host_status_words(B) + 16, 0);
  std::vector<int32_t> act(B * 40, 0), rew(B * 40, 0);
  std::vector<float> obs(B * 40 * 8, 0.f);
  const float w[8] = {-12.63f, 6.60f, -9.22f, -19.77f, -13.08f, -10.49f, -1.61f, -24.04f};
  int rc = tetris_host_reset(&d, cols.data(), meta.data(), nullptr, piece.data(), nv.data(), nullptr, nullptr, 0, status.data(), 1,
                             7, 0, 0, B, nullptr);
  rc |= tetris_host_step_many(&d, cols.data(), meta.data(), 30, 0, w, act.data(), obs.data(), rew.data(), done.data(), lines.data(),
                              nvn.data(), pn.data(), status.data(), 1, 7, 0, 0, B, nullptr);
  const int A = d.a_max;
  std::vector<float> feats(B * A * 8, 0.f), feats_all(B * A * 8, 0.f), fit(B * A, 0.f), bv(B, 0.f);
  std::vector<uint8_t> n1(B, 0), n2(B, 0);
  std::vector<int32_t> ba(B, 0);
  std::vector<double> ret(B * A, 0.0);
  rc |= tetris_host_afterstates(&d, cols.data(), meta.data(), feats.data(), n1.data(), feats_all.data(), n2.data(), A * 8, 8, B, nullptr);
  rc |= tetris_host_policy_greedy(&d, cols.data(), meta.data(), w, ba.data(), bv.data(), fit.data(), B, nullptr);
  rc |= tetris_host_rollouts(&d, cols.data(), meta.data(), ret.data(), 3, 2, 1, w, nullptr, 7, 30, 0, B, nullptr);
  rc |= tetris_host_step_many(&d, cols.data(), meta.data(), 6, 1, w, act.data(), obs.data(), rew.data(), done.data(), lines.data(),
                              nvn.data(), pn.data(), status.data(), 1, 7, 30, 0, B, nullptr);
  double s = 0;
  for (float x : fit) s += x;
  for (double x : ret) s += (x == x) ? x : 0;
  for (float x : feats_all) s += x;
  for (int32_t x : ba) s += x;
  printf("%2d x %2d, %zu pieces: rc %d checksum %.3f\n", C, R, ids.size(), rc, s);
  return rc;
}

int main() {
  const std::vector<int32_t> dflt = {4, 3}, std7 = {0, 7, 8, 1, 2, 5, 6};
  int rc = 0;
  for (int C : {10, 12, 11, 9, 5})
    for (int R : {20, 40}) {
      rc |= run(C, R, dflt, 256);
      rc |= run(C, R, std7, 256);
    }
  return rc;
}
