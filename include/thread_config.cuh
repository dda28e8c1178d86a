// thread_config.cuh — the reference's header name (cuda/thread_config.cuh), forwarding to this repo's
// HIP implementation of the same interface so that code written against the reference includes
// compiles unchanged with hipcc.
#pragma once
#include "gab/thread_config.hpp"
