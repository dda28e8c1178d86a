// thread_config.hpp — launch presets, name-compatible with the reference's
// cuda/thread_config.cuh:4-35.  The gfx950 kernels choose their own shapes
// (64-lane wavefronts, 256-thread workgroups, x-fastest 64x4 tiles for the
// stencil); these constants remain for code written against the reference.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace ThreadConfig {

constexpr int DEFAULT_BLOCK_SIZE_1D = 256;
constexpr int SMALL_BLOCK_SIZE_1D = 128;
constexpr int LARGE_BLOCK_SIZE_1D = 512;
constexpr int MAX_BLOCK_SIZE_1D = 1024;

constexpr int BLOCK_SIZE_3D_X = 8;
constexpr int BLOCK_SIZE_3D_Y = 8;
constexpr int BLOCK_SIZE_3D_Z = 8;
constexpr int BLOCK_SIZE_3D_THIN_X = 16;
constexpr int BLOCK_SIZE_3D_THIN_Y = 16;
constexpr int BLOCK_SIZE_3D_THIN_Z = 2;
constexpr int BLOCK_SIZE_3D_SMALL_X = 4;
constexpr int BLOCK_SIZE_3D_SMALL_Y = 4;
constexpr int BLOCK_SIZE_3D_SMALL_Z = 4;

// gfx950 additions
constexpr int WAVEFRONT_SIZE = 64;
constexpr int STENCIL_TILE_X = 64;
constexpr int STENCIL_TILE_Y = 4;

inline int calculateGridSize1D(size_t totalThreads, int blockSize = DEFAULT_BLOCK_SIZE_1D) {
    return static_cast<int>((totalThreads + blockSize - 1) / blockSize);
}

inline dim3 calculateGridSize3D(int nx, int ny, int nz, int blockX = BLOCK_SIZE_3D_X,
                                int blockY = BLOCK_SIZE_3D_Y, int blockZ = BLOCK_SIZE_3D_Z) {
    return dim3((nx + blockX - 1) / blockX, (ny + blockY - 1) / blockY, (nz + blockZ - 1) / blockZ);
}

}  // namespace ThreadConfig
