// globals.cuh — the reference's header name (cuda/globals.cuh), forwarding to this repo's
// implementation of the same globals and result writers.  The reference header also pulls in the
// standard headers below and three using-declarations (cuda/globals.cuh:3-13), which code written
// against it relies on; numElements is its leftover constant (:32).
#pragma once
#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <stdio.h>
#include <thread>
#include <vector>

using std::cout;
using std::endl;
using std::vector;

#include "gab/globals.hpp"

constexpr int numElements = 50000;
