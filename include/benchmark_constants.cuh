// benchmark_constants.cuh — the reference's header name (cuda/benchmark_constants.cuh), forwarding to this repo's
// HIP implementation of the same interface so that code written against the reference includes
// compiles unchanged with hipcc.
#pragma once
#include "gab/benchmark_constants.hpp"
