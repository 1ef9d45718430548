// reduce.hpp — deterministic scalar reductions: per-wave partials + fixed-order finish (k_dense.hip).
#pragma once
#include "common.hpp"
#include "dev_math.hpp"

namespace covgpu {
// the calling wave's partial (valid in every lane after wave_sum) goes to part[slot][index]
COV_DEV void part_put(const DevProblem& P, int slot, int index, double wave_total) {
  if ((threadIdx.x & 63) == 0) P.part[(size_t)slot * P.part_n + index] = wave_total;
}
}  // namespace covgpu
