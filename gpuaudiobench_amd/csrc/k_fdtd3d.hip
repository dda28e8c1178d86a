// k_fdtd3d.hip — 3-D staggered-grid acoustic FDTD for gfx950.
//
// Replaces the four kernels of cuda/bench_fdtd3d.cu (velocity :14-57, pressure
// :60-98, inject :101-120, extract :123-139) and the launch sequence of
// runFDTD3DTimeStep (:384-438: 8 launches and one device sync per audio sample).
//
// One fused kernel per leapfrog step.  A cell's thread recomputes the three
// "high" faces its divergence needs (they belong to the +x/+y/+z neighbours)
// from the OLD fields, exactly as their owners do, so every array is read once
// and written once per step: 2*4*(n^3 + 3(n+1)n^2) bytes, the algorithmic
// minimum (the two-kernel form reads p and v twice: 1.5x the traffic).  All four
// grids are ping-ponged so that recomputation can never see a half-updated
// field.  Source injection (summed over tracks in track order — the reference's
// atomicAdd order is unspecified) is precomputed per buffer and folded into the
// step before each sample; the receiver is sampled into a B-long strip and
// broadcast to the T identical output tracks once per buffer.  Nothing
// synchronises with the host inside a buffer.
//
// Arithmetic is the reference kernels' with the single-rounding a -= c*d nvcc
// emits (explicit fmaf here, in the harness golden and in the oracle), so all
// three are bit-identical.
//
// Layouts are the reference's (cuda/bench_fdtd3d.cuh:189-206), x fastest:
//   p [nz][ny][nx], vx [nz][ny][nx+1], vy [nz][ny+1][nx], vz [nz+1][ny][nx],
// except that vx rows are stored with a pitch of nx+4 floats so that, like every
// other row, they start 16-byte aligned (the grids are plan-internal; the
// pressure grid, which is what a caller can read back, is unchanged).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <utility>
#include <vector>

#include "gab_common.hpp"

namespace gab {
namespace {

struct Grid { int nx, ny, nz, px; };   // px = pitch of a vx row (nx + 4)

struct Fields { float *p, *vx, *vy, *vz; };

// inj[s] = ((0 + 0.1 in[0,s]) + 0.1 in[1,s]) + ...     (FDTD3D_SOURCE_SCALE)
__global__ void fdtd_source_sums_kernel(const float* __restrict__ in, float* __restrict__ inj, int T,
                                        int B, int first, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = first + i;
    float acc = 0.0f;
    int t = 0;
    for (; t + 16 <= T; t += 16) {          // sixteen rows in flight; the adds stay in track order
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = in[(size_t)(t + k) * B + s];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __fadd_rn(acc, __fmul_rn(v[k], 0.1f));
    }
    for (; t < T; ++t) acc = __fadd_rn(acc, __fmul_rn(in[(size_t)t * B + s], 0.1f));
    inj[s] = acc;
}

__global__ void fdtd_add_source_kernel(float* __restrict__ p, size_t src, const float* __restrict__ inj,
                                       int s) {
    if (blockIdx.x == 0 && threadIdx.x == 0) p[src] = __fadd_rn(p[src], inj[s]);
}

// out[t*B + s] = strip[s] for every track (the receiver value is track-independent)
__global__ void fdtd_broadcast_kernel(const float* __restrict__ strip, float* __restrict__ out, int T,
                                      int B, int first, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (i >= count) return;
    out[(size_t)t * B + first + i] = strip[first + i];
}

// One leapfrog step, old fields -> new fields.
//   add_next  : inj value to add to the NEW source cell (the next step opens a sample), else null
//   strip_out : where to store 0.1f * p_new[rcv] (this step closes a sample), else null
__global__ __launch_bounds__(256) void fdtd_step_kernel(Fields o, Fields n, Grid g, float c1, float c2,
                                                       float damp, size_t src, size_t rcv,
                                                       const float* __restrict__ add_next,
                                                       float* __restrict__ strip_out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const size_t sxy = (size_t)nx * ny;
    const size_t pi = z * sxy + (size_t)y * nx + x;
    const size_t ix = ((size_t)z * ny + y) * g.px + x;
    const size_t iy = ((size_t)z * (ny + 1) + y) * nx + x;
    const size_t iz = pi;

    const float pc = o.p[pi];
    // own (low) faces: updated on interior faces, carried over otherwise
    float fx = o.vx[ix], fy = o.vy[iy], fz = o.vz[iz];
    if (x > 0) fx = __builtin_fmaf(-c1, __fsub_rn(pc, o.p[pi - 1]), fx);
    if (y > 0) fy = __builtin_fmaf(-c1, __fsub_rn(pc, o.p[pi - nx]), fy);
    if (z > 0) fz = __builtin_fmaf(-c1, __fsub_rn(pc, o.p[pi - sxy]), fz);
    n.vx[ix] = fx;
    n.vy[iy] = fy;
    n.vz[iz] = fz;
    // the outermost high faces have no owning cell: carry them over
    if (x == nx - 1) n.vx[ix + 1] = o.vx[ix + 1];
    if (y == ny - 1) n.vy[iy + nx] = o.vy[iy + nx];
    if (z == nz - 1) n.vz[iz + sxy] = o.vz[iz + sxy];

    float pv;
    const bool interior = x > 0 && x < nx - 1 && y > 0 && y < ny - 1 && z > 0 && z < nz - 1;
    if (interior) {
        // high faces, recomputed from the old fields exactly as their owners do
        const float hx = __builtin_fmaf(-c1, __fsub_rn(o.p[pi + 1], pc), o.vx[ix + 1]);
        const float hy = __builtin_fmaf(-c1, __fsub_rn(o.p[pi + nx], pc), o.vy[iy + nx]);
        const float hz = __builtin_fmaf(-c1, __fsub_rn(o.p[pi + sxy], pc), o.vz[iz + sxy]);
        const float div = __fadd_rn(__fadd_rn(__fsub_rn(hx, fx), __fsub_rn(hy, fy)), __fsub_rn(hz, fz));
        pv = __builtin_fmaf(-c2, div, pc);
    } else {
        pv = __fmul_rn(pc, damp);
    }
    if (strip_out != nullptr && pi == rcv) *strip_out = __fmul_rn(pv, 0.1f);   // FDTD3D_OUTPUT_SCALE
    if (add_next != nullptr && pi == src) pv = __fadd_rn(pv, *add_next);
    n.p[pi] = pv;
}

// Same step, four consecutive x-cells per thread (nx % 4 == 0): every row access is
// a 16-byte load or store.  Cell j of the thread is x0 + j.
__global__ __launch_bounds__(256) void fdtd_step_vec4_kernel(Fields o, Fields n, Grid g, float c1,
                                                            float c2, float damp, size_t src, size_t rcv,
                                                            const float* __restrict__ add_next,
                                                            float* __restrict__ strip_out) {
    const int tx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const int x0 = 4 * tx;
    if (x0 >= nx || y >= ny) return;
    const size_t sxy = (size_t)nx * ny;
    const size_t pi = z * sxy + (size_t)y * nx + x0;
    const size_t ix = ((size_t)z * ny + y) * g.px + x0;
    const size_t iy = ((size_t)z * (ny + 1) + y) * nx + x0;
    const size_t iz = pi;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    auto st4 = [](float* p, float a, float b, float c, float d) {
        *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    };

    const float4 pc4 = ld4(o.p + pi);
    const float pc[4] = {pc4.x, pc4.y, pc4.z, pc4.w};
    const float4 vx4 = ld4(o.vx + ix), vy4 = ld4(o.vy + iy), vz4 = ld4(o.vz + iz);
    float fx[5] = {vx4.x, vx4.y, vx4.z, vx4.w, o.vx[ix + 4]};      // faces x0 .. x0+4
    float fy[4] = {vy4.x, vy4.y, vy4.z, vy4.w};
    float fz[4] = {vz4.x, vz4.y, vz4.z, vz4.w};

    // low faces (owned): x
    if (x0 > 0) fx[0] = __builtin_fmaf(-c1, __fsub_rn(pc[0], o.p[pi - 1]), fx[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) fx[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pc[j - 1]), fx[j]);
    // face x0+4 belongs to the next thread (or is the carried-over outer face)
    const bool has_right = x0 + 4 < nx;
    const float pr = has_right ? o.p[pi + 4] : 0.0f;
    const float fx4_new = has_right ? __builtin_fmaf(-c1, __fsub_rn(pr, pc[3]), fx[4]) : fx[4];
    if (y > 0) {
        const float4 q = ld4(o.p + pi - nx);
        const float pm[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) fy[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pm[j]), fy[j]);
    }
    if (z > 0) {
        const float4 q = ld4(o.p + pi - sxy);
        const float pm[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) fz[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pm[j]), fz[j]);
    }
    st4(n.vx + ix, fx[0], fx[1], fx[2], fx[3]);
    st4(n.vy + iy, fy[0], fy[1], fy[2], fy[3]);
    st4(n.vz + iz, fz[0], fz[1], fz[2], fz[3]);
    if (!has_right) n.vx[ix + 4] = fx[4];                             // outer face x = nx
    if (y == ny - 1) *reinterpret_cast<float4*>(n.vy + iy + nx) = ld4(o.vy + iy + nx);
    if (z == nz - 1) *reinterpret_cast<float4*>(n.vz + iz + sxy) = ld4(o.vz + iz + sxy);

    float pv[4];
    const bool row_interior = y > 0 && y < ny - 1 && z > 0 && z < nz - 1;
    if (row_interior) {
        const float4 pyp = ld4(o.p + pi + nx), pzp = ld4(o.p + pi + sxy);
        const float4 vyp = ld4(o.vy + iy + nx), vzp = ld4(o.vz + iz + sxy);
        const float py[4] = {pyp.x, pyp.y, pyp.z, pyp.w}, pz[4] = {pzp.x, pzp.y, pzp.z, pzp.w};
        const float hyo[4] = {vyp.x, vyp.y, vyp.z, vyp.w}, hzo[4] = {vzp.x, vzp.y, vzp.z, vzp.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            if (x > 0 && x < nx - 1) {
                const float hx = (j < 3) ? fx[j + 1] : fx4_new;       // already the updated face
                const float hy = __builtin_fmaf(-c1, __fsub_rn(py[j], pc[j]), hyo[j]);
                const float hz = __builtin_fmaf(-c1, __fsub_rn(pz[j], pc[j]), hzo[j]);
                const float div = __fadd_rn(__fadd_rn(__fsub_rn(hx, fx[j]), __fsub_rn(hy, fy[j])),
                                            __fsub_rn(hz, fz[j]));
                pv[j] = __builtin_fmaf(-c2, div, pc[j]);
            } else {
                pv[j] = __fmul_rn(pc[j], damp);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = __fmul_rn(pc[j], damp);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (strip_out != nullptr && pi + j == rcv) *strip_out = __fmul_rn(pv[j], 0.1f);
        if (add_next != nullptr && pi + j == src) pv[j] = __fadd_rn(pv[j], *add_next);
    }
    st4(n.p + pi, pv[0], pv[1], pv[2], pv[3]);
}

// The same step with the in-plane neighbours staged through LDS: a workgroup covers ROWS rows
// of one z-plane over the whole x extent (nx <= 4 * LX); every thread parks its own pressure and
// vy row segment in LDS, threads of the first / last row add the halo rows above and below, and
// p(x +/- 1), p(y +/- 1), vy(y + 1) then come from LDS instead of five more global loads.  Same
// arithmetic, same bits.
template <int LX, int ROWS>
__global__ __launch_bounds__(LX * ROWS) void fdtd_step_lds_kernel(Fields o, Fields n, Grid g, float c1,
                                                                 float c2, float damp, size_t src, size_t rcv,
                                                                 const float* __restrict__ add_next,
                                                                 float* __restrict__ strip_out) {
    constexpr int W = 4 * LX + 8;                      // row pitch in LDS: 4 floats of margin each side
    __shared__ float sp[(ROWS + 2) * W];               // pressure rows y0-1 .. y0+ROWS
    __shared__ float svy[(ROWS + 1) * W];              // vy rows y0 .. y0+ROWS
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int y0 = blockIdx.y * ROWS, y = y0 + ty, z = blockIdx.z;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const int x0 = 4 * tx;
    const bool live = x0 < nx && y < ny;
    const size_t sxy = (size_t)nx * ny;
    const size_t pi = z * sxy + (size_t)y * nx + x0;
    const size_t ix = ((size_t)z * ny + y) * g.px + x0;
    const size_t iy = ((size_t)z * (ny + 1) + y) * nx + x0;
    const size_t iz = pi;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    auto st4 = [](float* p, float a, float b, float c, float d) {
        *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    };
    auto lds4 = [](float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; };
    float* const prow = sp + (ty + 1) * W + 4 + x0;
    float* const vrow = svy + ty * W + 4 + x0;

    float4 pc4 = make_float4(0.f, 0.f, 0.f, 0.f), vy4 = pc4, vx4 = pc4, vz4 = pc4;
    if (live) {
        pc4 = ld4(o.p + pi);
        vy4 = ld4(o.vy + iy);
        vx4 = ld4(o.vx + ix);
        vz4 = ld4(o.vz + iz);
        lds4(prow, pc4);
        lds4(vrow, vy4);
        if (ty == 0 && y > 0) lds4(prow - W, ld4(o.p + pi - nx));                          // row y0-1
        const bool last_row = ty == ROWS - 1 || y == ny - 1;
        if (last_row) {
            lds4(vrow + W, ld4(o.vy + iy + nx));                                           // vy face y+1 (exists up to ny)
            if (y < ny - 1) lds4(prow + W, ld4(o.p + pi + nx));                            // row y+1
        }
    }
    __syncthreads();
    if (!live) return;

    const float pc[4] = {pc4.x, pc4.y, pc4.z, pc4.w};
    float fx[5] = {vx4.x, vx4.y, vx4.z, vx4.w, o.vx[ix + 4]};
    float fy[4] = {vy4.x, vy4.y, vy4.z, vy4.w};
    float fz[4] = {vz4.x, vz4.y, vz4.z, vz4.w};
    if (x0 > 0) fx[0] = __builtin_fmaf(-c1, __fsub_rn(pc[0], prow[-1]), fx[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) fx[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pc[j - 1]), fx[j]);
    const bool has_right = x0 + 4 < nx;
    const float pr = has_right ? prow[4] : 0.0f;
    const float fx4_new = has_right ? __builtin_fmaf(-c1, __fsub_rn(pr, pc[3]), fx[4]) : fx[4];
    if (y > 0) {
        const float4 q = *reinterpret_cast<const float4*>(prow - W);
        const float pm[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) fy[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pm[j]), fy[j]);
    }
    if (z > 0) {
        const float4 q = ld4(o.p + pi - sxy);
        const float pm[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) fz[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pm[j]), fz[j]);
    }
    st4(n.vx + ix, fx[0], fx[1], fx[2], fx[3]);
    st4(n.vy + iy, fy[0], fy[1], fy[2], fy[3]);
    st4(n.vz + iz, fz[0], fz[1], fz[2], fz[3]);
    if (!has_right) n.vx[ix + 4] = fx[4];
    if (y == ny - 1) *reinterpret_cast<float4*>(n.vy + iy + nx) = *reinterpret_cast<const float4*>(vrow + W);
    if (z == nz - 1) *reinterpret_cast<float4*>(n.vz + iz + sxy) = ld4(o.vz + iz + sxy);

    float pv[4];
    const bool row_interior = y > 0 && y < ny - 1 && z > 0 && z < nz - 1;
    if (row_interior) {
        const float4 pyp = *reinterpret_cast<const float4*>(prow + W), pzp = ld4(o.p + pi + sxy);
        const float4 vyp = *reinterpret_cast<const float4*>(vrow + W), vzp = ld4(o.vz + iz + sxy);
        const float py[4] = {pyp.x, pyp.y, pyp.z, pyp.w}, pz[4] = {pzp.x, pzp.y, pzp.z, pzp.w};
        const float hyo[4] = {vyp.x, vyp.y, vyp.z, vyp.w}, hzo[4] = {vzp.x, vzp.y, vzp.z, vzp.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            if (x > 0 && x < nx - 1) {
                const float hx = (j < 3) ? fx[j + 1] : fx4_new;
                const float hy = __builtin_fmaf(-c1, __fsub_rn(py[j], pc[j]), hyo[j]);
                const float hz = __builtin_fmaf(-c1, __fsub_rn(pz[j], pc[j]), hzo[j]);
                const float div = __fadd_rn(__fadd_rn(__fsub_rn(hx, fx[j]), __fsub_rn(hy, fy[j])),
                                            __fsub_rn(hz, fz[j]));
                pv[j] = __builtin_fmaf(-c2, div, pc[j]);
            } else {
                pv[j] = __fmul_rn(pc[j], damp);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = __fmul_rn(pc[j], damp);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (strip_out != nullptr && pi + j == rcv) *strip_out = __fmul_rn(pv[j], 0.1f);
        if (add_next != nullptr && pi + j == src) pv[j] = __fadd_rn(pv[j], *add_next);
    }
    st4(n.p + pi, pv[0], pv[1], pv[2], pv[3]);
}

}  // namespace
}  // namespace gab

// A buffer is hundreds to thousands of tiny dependent launches (3 per sample); at the
// reference's 52^3 grid the kernel is shorter than the host's launch cost, so the
// chain for one (input, output, range, ping-pong phase) signature is captured into a
// hipGraph once and replayed.
struct FdtdGraphKey {
    const float* in; float* out; int tracks, bufsize, first, count; const float* cur_p;
    bool operator==(const FdtdGraphKey& o) const {
        return in == o.in && out == o.out && tracks == o.tracks && bufsize == o.bufsize &&
               first == o.first && count == o.count && cur_p == o.cur_p;
    }
};

struct gab_fdtd_plan {
    gab_fdtd_params P;
    gab::Fields cur{}, nxt{};          // ping-pong
    float* inj = nullptr;              // per-sample source sums of the current buffer
    float* strip = nullptr;            // per-sample receiver values
    int strip_cap = 0;
    size_t np = 0, nvx = 0, nvy = 0, nvz = 0;
    bool use_graphs = true;
    bool lds_tiles = true;    // rows wide enough to fill a 32-lane row of the LDS-halo kernel (GAB_FDTD_LDS=0: off)
    hipStream_t capture_stream = nullptr;   // capture target (the caller's stream may be the null stream)
    std::vector<std::pair<FdtdGraphKey, hipGraphExec_t>> graphs;   // small LRU, newest last
};

namespace {

void free_fields(gab::Fields& f) {
    if (f.p) (void)hipFree(f.p);
    if (f.vx) (void)hipFree(f.vx);
    if (f.vy) (void)hipFree(f.vy);
    if (f.vz) (void)hipFree(f.vz);
    f = gab::Fields{};
}

void alloc_fields(gab::Fields& f, const gab_fdtd_plan& pl) {
    GAB_HIP_CHECK(hipMalloc(&f.p, pl.np * sizeof(float)));
    GAB_HIP_CHECK(hipMalloc(&f.vx, pl.nvx * sizeof(float)));
    GAB_HIP_CHECK(hipMalloc(&f.vy, pl.nvy * sizeof(float)));
    GAB_HIP_CHECK(hipMalloc(&f.vz, pl.nvz * sizeof(float)));
}

void zero_fields(gab::Fields& f, const gab_fdtd_plan& pl, hipStream_t s) {
    GAB_HIP_CHECK(hipMemsetAsync(f.p, 0, pl.np * sizeof(float), s));
    GAB_HIP_CHECK(hipMemsetAsync(f.vx, 0, pl.nvx * sizeof(float), s));
    GAB_HIP_CHECK(hipMemsetAsync(f.vy, 0, pl.nvy * sizeof(float), s));
    GAB_HIP_CHECK(hipMemsetAsync(f.vz, 0, pl.nvz * sizeof(float), s));
}

}  // namespace

extern "C" {

int gab_fdtd_default_params(int nx, int ny, int nz, gab_fdtd_params* out) {
    if (!out) return gab::bad_arg("gab_fdtd_default_params: null pointer");
    if (nx < 4 || ny < 4 || nz < 4) return gab::bad_arg("gab_fdtd_default_params: grid must be at least 4^3");
    // cuda/bench_fdtd3d.cuh:17-41 — all single precision
    const float c = 343.0f, dx = 0.01f, rho = 1.225f;
    const float dt = 0.5f * dx / (c * 1.732050808f);
    out->nx = nx; out->ny = ny; out->nz = nz;
    const int rx = nx - 2, ry = ny - 2, rz = nz - 2;
    out->source_x = rx / 2;       out->source_y = ry / 2;        out->source_z = rz / 10;
    out->receiver_x = rx * 4 / 5; out->receiver_y = ry * 3 / 10; out->receiver_z = rz / 2;
    out->steps_per_sample = 3;
    out->dt_over_rho_dx = dt / (rho * dx);
    out->rho_c2_dt_over_dx = rho * c * c * dt / dx;
    out->absorption_coeff = 0.2f;
    return GAB_OK;
}

int gab_fdtd_create(gab_fdtd_plan** out, const gab_fdtd_params* params) {
    return gab::guarded([&]() -> int {
        if (!out || !params) return gab::bad_arg("gab_fdtd_create: null pointer");
        const gab_fdtd_params& P = *params;
        if (P.nx < 3 || P.ny < 3 || P.nz < 3) return gab::bad_arg("gab_fdtd_create: grid too small");
        auto inside = [&](int x, int y, int z) {
            return x >= 0 && x < P.nx && y >= 0 && y < P.ny && z >= 0 && z < P.nz;
        };
        if (!inside(P.source_x, P.source_y, P.source_z) || !inside(P.receiver_x, P.receiver_y, P.receiver_z))
            return gab::bad_arg("gab_fdtd_create: source/receiver outside the grid");
        if (P.steps_per_sample < 1) return gab::bad_arg("gab_fdtd_create: steps_per_sample must be >= 1");
        auto* f = new gab_fdtd_plan;
        f->P = P;
        if (const char* v = getenv("GAB_FDTD_GRAPH")) f->use_graphs = atoi(v) != 0;
        if (const char* v = getenv("GAB_FDTD_LDS")) f->lds_tiles = atoi(v) != 0;
        f->np = (size_t)P.nx * P.ny * P.nz;
        f->nvx = (size_t)(P.nx + 4) * P.ny * P.nz + 4;   // padded pitch, see file header
        f->nvy = (size_t)P.nx * (P.ny + 1) * P.nz;
        f->nvz = (size_t)P.nx * P.ny * (P.nz + 1);
        try {
            alloc_fields(f->cur, *f);
            alloc_fields(f->nxt, *f);
        } catch (...) {
            gab_fdtd_destroy(f);
            throw;
        }
        *out = f;
        int rc = gab_fdtd_reset(f, nullptr);
        if (rc) return rc;
        GAB_HIP_CHECK(hipStreamSynchronize(nullptr));
        return GAB_OK;
    });
}

int gab_fdtd_destroy(gab_fdtd_plan* f) {
    if (!f) return GAB_OK;
    (void)hipDeviceSynchronize();
    for (auto& g : f->graphs) (void)hipGraphExecDestroy(g.second);
    if (f->capture_stream) (void)hipStreamDestroy(f->capture_stream);
    free_fields(f->cur);
    free_fields(f->nxt);
    if (f->inj) (void)hipFree(f->inj);
    if (f->strip) (void)hipFree(f->strip);
    delete f;
    return GAB_OK;
}

int gab_fdtd_reset(gab_fdtd_plan* f, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_reset: null plan");
        hipStream_t s = gab::as_stream(stream);
        zero_fields(f->cur, *f, s);
        zero_fields(f->nxt, *f, s);
        return GAB_OK;
    });
}

int gab_fdtd_process(gab_fdtd_plan* f, const float* d_in, float* d_out, int tracks, int bufsize,
                     int first_sample, int n_samples, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_in || !d_out) return gab::bad_arg("gab_fdtd_process: null pointer");
        if (tracks <= 0 || bufsize <= 0 || first_sample < 0 || n_samples < 0 ||
            first_sample + n_samples > bufsize)
            return gab::bad_arg("gab_fdtd_process: sample range outside the buffer");
        if (n_samples == 0) return GAB_OK;
        const gab_fdtd_params& P = f->P;
        hipStream_t s = gab::as_stream(stream);
        if (f->strip_cap < bufsize) {
            GAB_HIP_CHECK(hipStreamSynchronize(s));
            for (auto& c : f->graphs) (void)hipGraphExecDestroy(c.second);   // they point at the old strips
            f->graphs.clear();
            if (f->inj) (void)hipFree(f->inj);
            if (f->strip) (void)hipFree(f->strip);
            f->inj = f->strip = nullptr;
            GAB_HIP_CHECK(hipMalloc(&f->inj, sizeof(float) * bufsize));
            GAB_HIP_CHECK(hipMalloc(&f->strip, sizeof(float) * bufsize));
            f->strip_cap = bufsize;
        }
        gab::Grid g{P.nx, P.ny, P.nz, P.nx + 4};
        const size_t sxy = (size_t)P.nx * P.ny;
        const size_t src = P.source_z * sxy + (size_t)P.source_y * P.nx + P.source_x;
        const size_t rcv = P.receiver_z * sxy + (size_t)P.receiver_y * P.nx + P.receiver_x;
        const float damp = 1.0f - P.absorption_coeff;
        const int last = first_sample + n_samples;

        // enqueue the whole chain on `q`, walking local copies of the ping-pong pair
        auto enqueue = [&](hipStream_t q, gab::Fields cur, gab::Fields nxt) {
            gab::fdtd_source_sums_kernel<<<(n_samples + 127) / 128, 128, 0, q>>>(d_in, f->inj, tracks, bufsize,
                                                                               first_sample, n_samples);
            // the first sample's source goes straight into the current pressure grid; later
            // ones are folded into the step that precedes them
            gab::fdtd_add_source_kernel<<<1, 64, 0, q>>>(cur.p, src, f->inj, first_sample);
            const bool vec4 = (P.nx % 4) == 0;
            const int tx = vec4 ? P.nx / 4 : P.nx;                    // threads along x
            const int bx = tx >= 64 ? 64 : (tx >= 32 ? 32 : 16);
            dim3 block(bx, 256 / bx, 1);
            dim3 grid((tx + bx - 1) / bx, (P.ny + block.y - 1) / block.y, P.nz);
            for (int smp = first_sample; smp < last; ++smp) {
                for (int step = 0; step < P.steps_per_sample; ++step) {
                    const bool closes = step == P.steps_per_sample - 1;
                    const float* add_next = (closes && smp + 1 < last) ? f->inj + smp + 1 : nullptr;
                    float* strip_out = closes ? f->strip + smp : nullptr;
                    if (vec4 && f->lds_tiles && tx > 16 && tx <= 64) {
                        // whole x extent in one workgroup: LX x ROWS = 256 threads, (ROWS + 2) pressure
                        // rows in LDS; LX is the smallest of 16 / 32 / 64 that covers nx / 4
#define GAB_FDTD_LDS_LAUNCH(LX, ROWS)                                                                 \
    gab::fdtd_step_lds_kernel<LX, ROWS><<<dim3(1, (P.ny + ROWS - 1) / ROWS, P.nz), dim3(LX, ROWS, 1), 0, q>>>( \
        cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, damp, src, rcv, add_next, strip_out)
                        // 512-thread tiles (half as many halo rows) once they still make >= 4 workgroups
                        // per CU; 256-thread tiles below that (measured: 13.6 vs 14.3 us/step at 128^3,
                        // 47.5 vs 49.6 at 200^3, but 9.5 vs 8.8 at 96^3)
                        const int lx = tx <= 32 ? 32 : 64;
                        const bool big = (long)((P.ny + 512 / lx - 1) / (512 / lx)) * P.nz >= 1024;
                        if (lx == 32 && big) GAB_FDTD_LDS_LAUNCH(32, 16);
                        else if (lx == 32) GAB_FDTD_LDS_LAUNCH(32, 8);
                        else if (big) GAB_FDTD_LDS_LAUNCH(64, 8);
                        else GAB_FDTD_LDS_LAUNCH(64, 4);
#undef GAB_FDTD_LDS_LAUNCH
                    } else if (vec4)
                        gab::fdtd_step_vec4_kernel<<<grid, block, 0, q>>>(cur, nxt, g, P.dt_over_rho_dx,
                                                                         P.rho_c2_dt_over_dx, damp, src, rcv,
                                                                         add_next, strip_out);
                    else
                        gab::fdtd_step_kernel<<<grid, block, 0, q>>>(cur, nxt, g, P.dt_over_rho_dx,
                                                                    P.rho_c2_dt_over_dx, damp, src, rcv,
                                                                    add_next, strip_out);
                    std::swap(cur, nxt);
                }
            }
            dim3 bgrid((n_samples + 127) / 128, tracks);
            gab::fdtd_broadcast_kernel<<<bgrid, 128, 0, q>>>(f->strip, d_out, tracks, bufsize, first_sample,
                                                            n_samples);
        };
        const long launches = 3L + (long)n_samples * P.steps_per_sample;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        bool replayed = false;
        if (f->use_graphs && launches >= 24 && cap == hipStreamCaptureStatusNone) {
            const FdtdGraphKey key{d_in, d_out, tracks, bufsize, first_sample, n_samples, f->cur.p};
            hipGraphExec_t exec = nullptr;
            for (size_t i = 0; i < f->graphs.size(); ++i)
                if (f->graphs[i].first == key) {
                    auto hit = f->graphs[i];
                    f->graphs.erase(f->graphs.begin() + i);
                    f->graphs.push_back(hit);
                    exec = hit.second;
                    break;
                }
            if (!exec) {
                hipGraph_t graph = nullptr;
                if (!f->capture_stream)
                    GAB_HIP_CHECK(hipStreamCreateWithFlags(&f->capture_stream, hipStreamNonBlocking));
                GAB_HIP_CHECK(hipStreamBeginCapture(f->capture_stream, hipStreamCaptureModeThreadLocal));
                enqueue(f->capture_stream, f->cur, f->nxt);
                GAB_HIP_CHECK(hipStreamEndCapture(f->capture_stream, &graph));
                GAB_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                if (f->graphs.size() >= 4) {
                    (void)hipGraphExecDestroy(f->graphs.front().second);
                    f->graphs.erase(f->graphs.begin());
                }
                f->graphs.emplace_back(key, exec);
            }
            GAB_HIP_CHECK(hipGraphLaunch(exec, s));
            replayed = true;
        }
        if (!replayed) enqueue(s, f->cur, f->nxt);
        if (((long)n_samples * P.steps_per_sample) & 1) std::swap(f->cur, f->nxt);
        return gab::launch_status("fdtd kernels");
    });
}

int gab_fdtd_copy_pressure(gab_fdtd_plan* f, float* d_dst, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_dst) return gab::bad_arg("gab_fdtd_copy_pressure: null pointer");
        GAB_HIP_CHECK(hipMemcpyAsync(d_dst, f->cur.p, f->np * sizeof(float), hipMemcpyDeviceToDevice,
                                     gab::as_stream(stream)));
        return GAB_OK;
    });
}

}  // extern "C"
