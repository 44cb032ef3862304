// Test shim around the host logic of popscle_amd/csrc/score_exact.hpp (settle_with): which cells get their singlet sums
// recomputed, and when the set closes.  The "exact sums" come from the caller's arrays; nothing here touches a device.
#include "score_exact.hpp"

extern "C" int probe_settle(int64_t C, double* l0, double* l2, const double* x0, const double* x2, int64_t* n_exact,
                            int32_t* calls, uint8_t* was_exact) {
  std::string err;
  *calls = 0;
  return score_exact::settle_with(C, l0, l2, n_exact, &err, [&](const std::vector<int32_t>& cells) -> int {
    ++*calls;
    for (size_t k = 0; k < cells.size(); ++k) {
      if (k && cells[k] <= cells[k - 1]) return 1;  // (ascending, no cell twice)
      if (was_exact[cells[k]]) return 1;            // (no cell is asked for twice)
      was_exact[cells[k]] = 1;
      l0[cells[k]] = x0[cells[k]];
      l2[cells[k]] = x2[cells[k]];
    }
    return 0;
  });
}
