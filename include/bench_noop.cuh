// bench_noop.cuh — the reference's header name (cuda/bench_noop.cuh), forwarding to the
// benchmark classes of this repo (same class names, constructor signatures and defaults).
#pragma once
#include "gab/benchmarks.hpp"
