// bench_utils.cuh — the reference's header name (cuda/bench_utils.cuh), forwarding to this repo's
// HIP implementation of the same interface so that code written against the reference includes
// compiles unchanged with hipcc.
#pragma once
#include "gab/bench_utils.hpp"
