// thread_config.hpp — launch-shape presets.
//
// The gfx950 kernels pick their own shapes (64-lane wavefronts, 256-thread workgroups, float4 rows
// for the stencil); what is kept here is the reference's vocabulary (cuda/thread_config.cuh:4-35),
// value for value, for code written against it — as a table, plus the gfx950 facts the kernels
// themselves go by.
//
//   X(name, value)
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#define GAB_THREAD_CONFIG_TABLE(X)                                                    \
    /* 1-D presets */                                                                 \
    X(DEFAULT_BLOCK_SIZE_1D, 256) X(SMALL_BLOCK_SIZE_1D, 128)                         \
    X(LARGE_BLOCK_SIZE_1D, 512)   X(MAX_BLOCK_SIZE_1D, 1024)                          \
    /* 3-D presets: cubic, thin (stencil slabs), small */                             \
    X(BLOCK_SIZE_3D_X, 8)        X(BLOCK_SIZE_3D_Y, 8)        X(BLOCK_SIZE_3D_Z, 8)   \
    X(BLOCK_SIZE_3D_THIN_X, 16)  X(BLOCK_SIZE_3D_THIN_Y, 16)  X(BLOCK_SIZE_3D_THIN_Z, 2) \
    X(BLOCK_SIZE_3D_SMALL_X, 4)  X(BLOCK_SIZE_3D_SMALL_Y, 4)  X(BLOCK_SIZE_3D_SMALL_Z, 4) \
    /* gfx950 (additive) */                                                           \
    X(WAVEFRONT_SIZE, 64) X(STENCIL_TILE_X, 64) X(STENCIL_TILE_Y, 4)

namespace ThreadConfig {

#define GAB_DEFINE_PRESET(name, value) constexpr int name = value;
GAB_THREAD_CONFIG_TABLE(GAB_DEFINE_PRESET)
#undef GAB_DEFINE_PRESET

// ceil(work / block) in each dimension
inline int calculateGridSize1D(size_t totalThreads, int blockSize = DEFAULT_BLOCK_SIZE_1D) {
    const size_t b = static_cast<size_t>(blockSize);
    return static_cast<int>((totalThreads + b - 1) / b);
}

inline dim3 calculateGridSize3D(int nx, int ny, int nz, int blockX = BLOCK_SIZE_3D_X,
                                int blockY = BLOCK_SIZE_3D_Y, int blockZ = BLOCK_SIZE_3D_Z) {
    auto blocks = [](int n, int b) { return static_cast<unsigned>((n + b - 1) / b); };
    return dim3(blocks(nx, blockX), blocks(ny, blockY), blocks(nz, blockZ));
}

}  // namespace ThreadConfig
