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

#include <cstdint>
#include <cstdlib>
#include <string>
#include <mutex>
#include <utility>
#include <algorithm>
#include <vector>

#include "gab_common.hpp"

namespace gab {
namespace {

// px = pitch of a vx row (nx + 4); z0 = global z of the first plane a launch covers (a z-slab of
// a decomposed grid launches only its own planes; nz stays the GLOBAL depth for the boundary tests)
struct Grid { int nx, ny, nz, px, z0; };

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

// out[t*B + s] = strip[s] for every track (the receiver value is track-independent).  `gave_up`, when given, is the
// resident kernel's timeout word: a launch that stopped waiting for a neighbour leaves undefined fields and an
// undefined strip, so THAT call's output is NaN from the first sample on — unmistakable — instead of plausible.
__global__ void fdtd_broadcast_kernel(const float* __restrict__ strip, float* __restrict__ out, int T,
                                      int B, int first, int count, const unsigned* __restrict__ gave_up = nullptr,
                                      unsigned* __restrict__ gave_up_host = nullptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    // the pinned copy of the word travels with this kernel (no copy command: a 4-byte engine copy per call costs more
    // than the store, and copy commands are where the runtime's one-off multi-millisecond stalls were found)
    if (gave_up_host != nullptr && i == 0 && t == 0 && *gave_up != 0)
        __hip_atomic_store(gave_up_host, *gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i >= count) return;
    const bool bad = gave_up != nullptr && *gave_up != 0;
    out[(size_t)t * B + first + i] = bad ? __uint_as_float(0x7fc00000u) : strip[first + i];
}

// Track-dependent source / receiver cells (gab_fdtd_set_track_positions): between two samples,
// ONE workgroup first reads every track's receiver cell for the sample that just closed, then —
// after a barrier, a receiver may also be somebody's source — adds the next sample of every
// track into its source cell.  Tracks that share a source cell form a group handled by one
// thread in track order, so the sum is the oracle's whatever the cells are.
__global__ __launch_bounds__(256) void fdtd_track_io_kernel(float* __restrict__ p, const float* __restrict__ in,
                                                            float* __restrict__ out, int T, int B, int extract_sample,
                                                            int inject_sample, const long long* __restrict__ rcv,
                                                            int n_groups, const long long* __restrict__ group_cell,
                                                            const int* __restrict__ group_start,
                                                            const int* __restrict__ group_tracks) {
    if (extract_sample >= 0)
        for (int t = threadIdx.x; t < T; t += blockDim.x)
            out[(size_t)t * B + extract_sample] = __fmul_rn(p[rcv[t]], 0.1f);      // FDTD3D_OUTPUT_SCALE
    __syncthreads();
    if (inject_sample >= 0)
        for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
            float v = p[group_cell[g]];
            for (int k = group_start[g]; k < group_start[g + 1]; ++k)
                v = __fadd_rn(v, __fmul_rn(in[(size_t)group_tracks[k] * B + inject_sample], 0.1f));
            p[group_cell[g]] = v;
        }
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
    const int z = blockIdx.z + g.z0;
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
    const int z = blockIdx.z + g.z0;
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
template <int LX, int ROWS, bool HOIST>
__global__ __launch_bounds__(LX * ROWS) void fdtd_step_lds_kernel(Fields o, Fields n, Grid g, float c1,
                                                                 float c2, float damp, size_t src, size_t rcv,
                                                                 const float* __restrict__ add_next,
                                                                 float* __restrict__ strip_out) {
    constexpr int W = 4 * LX + 8;                      // row pitch in LDS: 4 floats of margin each side
    __shared__ float sp[(ROWS + 2) * W];               // pressure rows y0-1 .. y0+ROWS
    __shared__ float svy[(ROWS + 1) * W];              // vy rows y0 .. y0+ROWS
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int y0 = blockIdx.y * ROWS, y = y0 + ty, z = blockIdx.z + g.z0;
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
    // HOIST: the z-neighbour planes and the vx face beyond the quad are requested with everything
    // else, so that the step pays ONE memory round trip (they are needed after the barrier only)
    float4 h_pzm = pc4, h_pzp = pc4, h_vzp = pc4;
    float h_vx4 = 0.0f;
    // The halo rows of the tile (its first row also parks row y0-1, its last row the rows above) are
    // requested in the same burst: as "load, wait, park" inside their branches each of them was a
    // further dependent round trip for the edge waves, and a workgroup is as late as its last wave.
    float4 halo_pm, halo_vy, halo_pp;        // set and used under the same predicates only (no default: a merge copy
                                             // behind a conditional load costs an s_waitcnt vmcnt(0) in the middle of the burst)
    const bool need_pm = live && ty == 0 && y > 0;                              // row y0-1
    const bool need_vy = live && (ty == ROWS - 1 || y == ny - 1);               // vy face y+1 (exists up to ny)
    const bool need_pp = need_vy && y < ny - 1;                                 // row y+1
    if (live) {
        pc4 = ld4(o.p + pi);
        vy4 = ld4(o.vy + iy);
        vx4 = ld4(o.vx + ix);
        vz4 = ld4(o.vz + iz);
        if (need_pm) halo_pm = ld4(o.p + pi - nx);
        if (need_vy) halo_vy = ld4(o.vy + iy + nx);
        if (need_pp) halo_pp = ld4(o.p + pi + nx);
        if constexpr (HOIST) {
            // unconditional, with the offset clamped on the first / last plane (the value is not used there):
            // under `if (z > 0)` the compiler merges the loaded registers with their zero defaults by a copy
            // placed right behind the load — an s_waitcnt vmcnt(0) in the middle of the burst
            h_vx4 = o.vx[ix + 4];
            h_pzm = ld4(o.p + pi - (z > 0 ? sxy : 0));
            h_pzp = ld4(o.p + pi + (z < nz - 1 ? sxy : 0));
            h_vzp = ld4(o.vz + iz + sxy);
        }
    }
    __builtin_amdgcn_sched_barrier(0);           // every request is out before the first one is waited for
    if (live) {
        lds4(prow, pc4);
        lds4(vrow, vy4);
        if (need_pm) lds4(prow - W, halo_pm);
        if (need_vy) lds4(vrow + W, halo_vy);
        if (need_pp) lds4(prow + W, halo_pp);
    }
    __syncthreads();
    if (!live) return;

    const float pc[4] = {pc4.x, pc4.y, pc4.z, pc4.w};
    float fx[5] = {vx4.x, vx4.y, vx4.z, vx4.w, HOIST ? h_vx4 : o.vx[ix + 4]};
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
        const float4 q = HOIST ? h_pzm : ld4(o.p + pi - sxy);
        const float pm[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) fz[j] = __builtin_fmaf(-c1, __fsub_rn(pc[j], pm[j]), fz[j]);
    }
    st4(n.vx + ix, fx[0], fx[1], fx[2], fx[3]);
    st4(n.vy + iy, fy[0], fy[1], fy[2], fy[3]);
    st4(n.vz + iz, fz[0], fz[1], fz[2], fz[3]);
    if (!has_right) n.vx[ix + 4] = fx[4];
    if (y == ny - 1) *reinterpret_cast<float4*>(n.vy + iy + nx) = *reinterpret_cast<const float4*>(vrow + W);
    if (z == nz - 1) *reinterpret_cast<float4*>(n.vz + iz + sxy) = HOIST ? h_vzp : ld4(o.vz + iz + sxy);

    float pv[4];
    const bool row_interior = y > 0 && y < ny - 1 && z > 0 && z < nz - 1;
    if (row_interior) {
        const float4 pyp = *reinterpret_cast<const float4*>(prow + W), pzp = HOIST ? h_pzp : ld4(o.p + pi + sxy);
        const float4 vyp = *reinterpret_cast<const float4*>(vrow + W), vzp = HOIST ? h_vzp : ld4(o.vz + iz + sxy);
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

// ---- small grids: one launch per audio SAMPLE (S leapfrog steps), temporal blocking in a tile ----
// The reference's own TODO (cuda/bench_fdtd3d.cu:12-13): tiles with halos and several steps per
// launch.  At its default 52^3 grid a step is shorter than a kernel boundary, so a chain of
// one-step launches is bound by the boundaries (4.2 us per step).  Here a workgroup owns a
// nx x TY x TZ tile, loads the tile plus a halo of S cells in y and z (clipped to the room),
// advances that box S steps on its own — the rim goes stale one cell per step, the owned cells are
// S cells inside it — and stores its tile.  Neighbouring tiles recompute each other's halo cells
// from the same old fields with the same operations, so the result is bit-identical to S
// one-step launches (and to the oracle).
// A thread holds one (x, y) column of the box in registers: p and the three low faces of its BZ
// cells.  Per step, phase 1 updates the faces (needs p of the x-1 and y-1 columns: LDS; z-1: own
// registers), phase 2 the pressure (needs the NEW faces of the x+1 and y+1 columns: LDS; z+1: own
// registers); the high faces are the neighbours' new low faces — the same fmaf the one-step kernels
// recompute.  Three LDS arrays (p, fx, fy), seven LDS accesses per cell and step.
// Tiles span the WHOLE x extent (rooms of this kernel are at most 56 wide) and are cut in y and z
// only: a box plane is then one contiguous run of every array, so the loads and stores of a wave
// are whole cache lines (a first version with 13 x 13 x 4 tiles moved six times as many lines as
// bytes it used and took 9 us per launch).
template <int TY, int TZ, int S, int NT>
__global__ __launch_bounds__(NT) void fdtd_sample_tile_kernel(
    Fields o, Fields n, Grid g, float c1, float c2, float damp, size_t src, size_t rcv,
    const float* __restrict__ add_next, float* __restrict__ strip_out) {
    constexpr int BZ = TZ + 2 * S;
    constexpr int LZ = NT;                             // LDS plane pitch: every thread has a slot
    constexpr int PAD = 64;                            // front / back pad: the y-1 / y+1 reads of the edge rows stay inside
    __shared__ float sp[PAD + BZ * LZ + PAD];
    __shared__ float sfx[PAD + BZ * LZ + PAD];
    __shared__ float sfy[PAD + BZ * LZ + PAD];
    const int nx = g.nx, ny = g.ny, nz = g.nz;        // nx <= 56 <= PAD
    // owned tile and its box, clipped to the room
    const int y0 = blockIdx.x * TY, y1 = min(ny, y0 + TY);
    const int z0 = blockIdx.y * TZ, z1 = min(nz, z0 + TZ);
    const int by0 = max(0, y0 - S), bz0 = max(0, z0 - S);
    const int ez = min(nz, z1 + S) - bz0;                              // planes of the box inside the room
    const int x = threadIdx.x % nx, ly = threadIdx.x / nx;
    const int y = by0 + ly;
    // The arithmetic below is straight-line for every thread and every register cell: cells outside
    // the room or the box only ever feed the rim that goes stale anyway (shell cells read no
    // neighbour), so they need no branches — only clamped load addresses and guarded stores.
    const int yc = min(y, ny - 1);
    // 32-bit element offsets: one address register per access
    const int sxy = nx * ny;
    const int l0 = PAD + threadIdx.x;                                  // LDS slot of the column at lz = 0 (= PAD + ly*nx + x)
    const int sx = ny * g.px, sy = (ny + 1) * nx;
    const int pi0 = bz0 * sxy + yc * nx + x;                           // p and vz, plane bz0
    const int ix0 = (bz0 * ny + yc) * g.px + x;
    const int iy0 = (bz0 * (ny + 1) + yc) * nx + x;

    float p[BZ], fx[BZ], fy[BZ], fz[BZ];
#pragma unroll
    for (int k = 0; k < BZ; ++k) {
        const int kc = min(k, ez - 1);
        p[k] = o.p[pi0 + kc * sxy];
        fx[k] = o.vx[ix0 + kc * sx];
        fy[k] = o.vy[iy0 + kc * sy];
        fz[k] = o.vz[pi0 + kc * sxy];
    }
    // the outermost high faces have no owning cell and are carried over: requested here, with
    // everything else, so that their latency is not paid after the last step
    const bool own_y = y >= y0 && y < y1;
    float cx[BZ], cy[BZ], cz = 0.0f;
#pragma unroll
    for (int k = 0; k < BZ; ++k) {
        const int kc = min(k, ez - 1);
        cx[k] = (own_y && x == nx - 1) ? o.vx[ix0 + kc * sx + 1] : 0.0f;
        cy[k] = (own_y && y == ny - 1) ? o.vy[iy0 + kc * sy + nx] : 0.0f;
    }
    if (own_y && z1 == nz) cz = o.vz[pi0 + (nz - bz0) * sxy];
#pragma unroll
    for (int k = 0; k < BZ; ++k) sp[l0 + k * LZ] = p[k];
    __syncthreads();
    const bool shell_xy = x == 0 || x == nx - 1 || y == 0 || y >= ny - 1;
    const bool has_xm = x > 0, has_ym = ly > 0;
#pragma unroll 1
    for (int step = 0; step < S; ++step) {
#pragma unroll
        for (int k = 0; k < BZ; ++k) {
            const float ux = __builtin_fmaf(-c1, __fsub_rn(p[k], sp[l0 + k * LZ - 1]), fx[k]);
            const float uy = __builtin_fmaf(-c1, __fsub_rn(p[k], sp[l0 + k * LZ - nx]), fy[k]);
            fx[k] = has_xm ? ux : fx[k];
            fy[k] = has_ym ? uy : fy[k];
        }
        // z faces from the thread's own registers (p is not touched in this phase)
#pragma unroll
        for (int k = 1; k < BZ; ++k) fz[k] = __builtin_fmaf(-c1, __fsub_rn(p[k], p[k - 1]), fz[k]);
#pragma unroll
        for (int k = 0; k < BZ; ++k) {
            sfx[l0 + k * LZ] = fx[k];
            sfy[l0 + k * LZ] = fy[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BZ; ++k) {
            const int z = bz0 + k;
            const bool shell = shell_xy || z == 0 || z >= nz - 1;
            const float hx = sfx[l0 + k * LZ + 1], hy = sfy[l0 + k * LZ + nx];
            const float hz = fz[k + 1 < BZ ? k + 1 : k];
            const float div = __fadd_rn(__fadd_rn(__fsub_rn(hx, fx[k]), __fsub_rn(hy, fy[k])), __fsub_rn(hz, fz[k]));
            const float pin = __builtin_fmaf(-c2, div, p[k]);
            p[k] = shell ? __fmul_rn(p[k], damp) : pin;
            sp[l0 + k * LZ] = p[k];
        }
        __syncthreads();
    }
    // ---- store the owned cells; the last step of a sample records the receiver and folds in the
    // next sample's source, like the one-step kernels
    if (y < y0 || y >= y1) return;
#pragma unroll
    for (int k = 0; k < BZ; ++k) {
        const int z = bz0 + k;
        if (z >= z0 && z < z1) {
            const int pi = pi0 + k * sxy, ix = ix0 + k * sx, iy = iy0 + k * sy;
            float pv = p[k];
            if (strip_out != nullptr && (size_t)pi == rcv) *strip_out = __fmul_rn(pv, 0.1f);     // FDTD3D_OUTPUT_SCALE
            if (add_next != nullptr && (size_t)pi == src) pv = __fadd_rn(pv, *add_next);
            n.p[pi] = pv;
            n.vx[ix] = fx[k];
            n.vy[iy] = fy[k];
            n.vz[pi] = fz[k];
            if (x == nx - 1) n.vx[ix + 1] = cx[k];
            if (y == ny - 1) n.vy[iy + nx] = cy[k];
            if (z == nz - 1) n.vz[pi + sxy] = cz;
        }
    }
}

// ---- the whole room resident in LDS: ONE launch per buffer ------------------------------------------------
// 128^3 cells with their three face arrays are 33.75 MB; the chip has 256 x 160 KB = 40 MB of LDS.  The room is
// cut into blocks of nx x BY x BZ cells, one 1024-thread workgroup (= one CU) each.  A thread owns the same one
// or two rows x four cells for the whole launch: p, vx, vy, vz in registers; p, vy, vz also in LDS, where the
// rows above and below inside the block read them; vx neighbours by a lane shift.  No field touches HBM between
// the first step of a buffer and its last; only the blocks' boundary PRESSURES cross between workgroups, once per
// step, through memory (DESIGN.md section 4a has the measurements and the history of the form).
//
// A step is the reference's two phases (velocity kernel, pressure kernel: cuda/bench_fdtd3d.cu:14-98) done in
// place — the same values as the fused step kernels above, so bit-identical to them and to the oracle:
//   V  every low face from the old pressures.  The faces on the block's first row / plane need the neighbour
//      block's pressures; the block also keeps the neighbour's low faces above its LAST row / plane and advances
//      them itself (same operation, same operands as the neighbour), so velocities never cross workgroups;
//   P  pressure from the new faces; the receiver tap and the next sample's source add ride on the step that closes
//      a sample, as in the fused kernels;
//   X  the hand-off: every boundary pressure is an 8-byte granule {value, tag = global step number} written by one
//      sc1 (write-through) store; the consumer is the very thread that needs the quad and sc1-loads it until its
//      four tags are the step's (MI355X_MICROARCH.md, form R2: no flag, no fence).  Two buffers by step parity: a
//      neighbour is never more than one step apart.  Tags go on across launches.
// Order inside a step (see the loop): ask | V from own pressures | barrier | P of the interior rows | what arrived
// -> the face rows' ghost-dependent faces -> their P, stored at once | barrier.  Face rows are every thread's
// first row, interior rows its second.
// Every workgroup must be on a CU at once: the grid is at most one workgroup per CU, and every poll is bounded —
// on a timeout the kernel sets a word the host turns into an error and a fall-back to the step kernels.
struct ResidentGeom {
    int by, bz;            // block extent in y and z (cells); x is the whole row
    int gy, gz;            // blocks along y and z
    int nq;                // float4 quads per row (nx / 4)
    int rows;              // rows per block = by * bz
    int fr;                // rows reserved per exchanged face = max(by, bz)
};

constexpr int kResThreads = 1024;
constexpr int kResRowSlots = kResThreads / 32;      // a row of up to 128 cells = 32 lanes x float4
constexpr int kResRowDwords = 256;                  // one exchanged row: 128 granules {pressure, tag}
#ifdef GAB_ABLATE
constexpr unsigned kResSpinLimit = 1u << 13;        // diagnostic builds: give up after milliseconds (tools/fdtd_timeout_check.py)
__device__ unsigned long long g_res_rounds[4];      // [0] extra poll rounds, [1] polls (bit 4; one lane per wave of one workgroup counts)
__device__ unsigned long long g_res_phase[16 * 8];  // [wave][mark]: clocks from a step's start to each mark, summed over steps (bit 16, one workgroup)
__device__ int g_res_ablate = 0;                    // 1 = no exchange between workgroups (wrong results); 2 = workgroup 0 never publishes
#else
constexpr unsigned kResSpinLimit = 1u << 20;        // ~ a second of polling before giving up
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// the value of the lane before / after this one (whole-wave DPP shift: one VALU instruction, no LDS crossbar)
__device__ __forceinline__ float lane_before(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ float lane_after(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
__device__ __forceinline__ f4 fnma4(float c, f4 a, f4 b, f4 acc) {      // acc - c * (a - b): a rounded difference, then one fma
    const f4 d = a - b;                                                  // (vector forms: two floats per v_pk_* instruction)
    return __builtin_elementwise_fma((f4){-c, -c, -c, -c}, d, acc);
}

// One exchanged quad = four granules {pressure bits, tag}, each 8 bytes written by one store; the two 16-byte
// stores of a row land in the row's two 512-byte halves, so every store instruction writes whole lines.
__device__ __forceinline__ void publish_quad(unsigned* row, f4 p, unsigned tag) {
    const u4 a = {__float_as_uint(p.x), tag, __float_as_uint(p.y), tag};
    const u4 b = {__float_as_uint(p.z), tag, __float_as_uint(p.w), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:512 sc1"
                 ::"v"(row), "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ unsigned peek_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int RPT>      // rows per thread slot: ceil(rows / 32)
__global__ __launch_bounds__(kResThreads, 1) void fdtd_resident_kernel(
    Fields f, Grid g, ResidentGeom rg, float c1, float c2, float damp, size_t src, size_t rcv,
    const float* __restrict__ inj, float* __restrict__ strip, int first_sample, int n_samples, int steps_per_sample,
    unsigned* __restrict__ xbuf, unsigned tag_base, unsigned* __restrict__ timeout_word) {
    extern __shared__ float lds[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, px = g.px;
    const int by = rg.by, bz = rg.bz;
    const int wg = blockIdx.x;
    const int bj = wg % rg.gy, bk = wg / rg.gy;
    const int y0 = bj * by, z0 = bk * bz;
    const bool has_ym = bj > 0, has_yp = bj + 1 < rg.gy, has_zm = bk > 0, has_zp = bk + 1 < rg.gz;
    // LDS images, [row = lz * by + ly][nx] each: what a thread's neighbours inside the block read
    const int lp = 4 * rg.nq;                                          // LDS row pitch: the row padded to whole quads
    float* const sp = lds;
    float* const svy = sp + rg.rows * lp;
    float* const svz = svy + rg.rows * lp;

    const int tid = threadIdx.x;
    const int xq = tid & 31;                                           // quad within the row
    const int slot = tid >> 5;
    const int x0 = 4 * xq;
    const size_t sxy = (size_t)nx * ny;
    // exchange rows (dword offsets into xbuf, one parity): workgroup w, face c (0: y-, 1: y+, 2: z-, 3: z+), row i
    const int wg_dwords = 4 * rg.fr * kResRowDwords;
    const int parity_dwords = (int)gridDim.x * wg_dwords;
    auto xrow = [&](int w, int face, int i) { return (w * 4 + face) * rg.fr * kResRowDwords + i * kResRowDwords + 4 * xq; };

    // ---- the thread's cells: row slot + 32 k, k < RPT.  Per row: the four field quads, one ghost pressure
    // quad per direction (a row is the block's first OR last in a direction: blocks are at least 2 x 2) and,
    // for a last row, the neighbour's low face above it, which this thread advances itself.  What kind of row
    // it is sits in one word of bits; the exchange rows are dword offsets into xbuf (-1: none).
    enum : unsigned {
        kOn = 1u, kHasYm = 2u, kHasZm = 4u, kHasYp = 8u, kHasZp = 16u,       // the row exists; it has a neighbour row INSIDE the block
        kFirstY = 32u, kFirstZ = 64u, kLastY = 128u, kLastZ = 256u,           // a neighbour block supplies / wants this row
        kM0 = 512u, kM1 = 1024u, kM2 = 2048u, kM3 = 4096u,                    // which cells of the quad are interior cells of the room
        kRcvShift = 13, kSrcShift = 16                                        // 1 + j of the receiver / source cell (0: not here)
    };
    // Face rows are every thread's FIRST row (the host only takes geometries whose face rows fit the row slots):
    // the ghost state exists once per thread, not once per row.
    f4 p4[RPT], vx4[RPT], vy4[RPT], vz4[RPT];
    f4 gpy, gpz, gfy, gfz;
    gpy = gpz = gfy = gfz = (f4){0.f, 0.f, 0.f, 0.f};
    int o_own[RPT], o_ym[RPT], o_zm[RPT], o_yp[RPT], o_zp[RPT], pub_y = -1, pub_z = -1;     // LDS float offsets (own row where there is no neighbour row)
    unsigned kind[RPT];
    const int up_y = 3 * rg.fr * kResRowDwords;                        // from my y+ row to the next block's y- row (and back: minus)
    const int up_z = (4 * rg.gy - 1) * rg.fr * kResRowDwords;          // from my z+ row to the next plane of blocks' z- row
    // Rows are dealt to the slots FACE ROWS FIRST (the block's two z faces, then its two y faces, then the
    // interior): the rows whose pressures the neighbours wait for are made and stored at the head of the
    // pressure phase, so that their way through memory overlaps the interior rows' work.
    const int n_face = rg.rows - (by - 2) * (bz - 2);
    auto row_of = [&](int q, int& ly, int& lz) {
        if (q < by) { ly = q; lz = 0; }
        else if (q < 2 * by) { ly = q - by; lz = bz - 1; }
        else if (q < 2 * by + (bz - 2)) { ly = 0; lz = 1 + q - 2 * by; }
        else if (q < n_face) { ly = by - 1; lz = 1 + q - 2 * by - (bz - 2); }
        else { const int i = q - n_face; ly = 1 + i % (by - 2); lz = 1 + i / (by - 2); }
    };
    // a thread's quad of a global row: one 16-byte access when rows are multiples of four cells (every quad is then
    // whole and aligned), cell by cell otherwise (entry and exit of the launch only; inside it rows are padded quads)
    const bool whole_quads = (nx & 3) == 0;
    const int my_cells = nx - x0 < 4 ? nx - x0 : 4;
    auto ld = [&](const float* a) -> f4 {
        if (whole_quads) return *reinterpret_cast<const f4*>(a);
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (my_cells > 0) v.x = a[0];
        if (my_cells > 1) v.y = a[1];
        if (my_cells > 2) v.z = a[2];
        if (my_cells > 3) v.w = a[3];
        return v;
    };
    auto st = [&](float* a, f4 v) {
        if (whole_quads) { *reinterpret_cast<f4*>(a) = v; return; }
        if (my_cells > 0) a[0] = v.x;
        if (my_cells > 1) a[1] = v.y;
        if (my_cells > 2) a[2] = v.z;
        if (my_cells > 3) a[3] = v.w;
    };
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int q = slot + kResRowSlots * k;
        int ly = 0, lz = 0;
        row_of(q < rg.rows ? q : 0, ly, lz);
        const int r = lz * by + ly;
        const int y = y0 + ly, z = z0 + lz;
        const bool on = xq < rg.nq && q < rg.rows && y < ny && z < nz;
        const size_t pi = (size_t)z * sxy + (size_t)y * nx + x0;
        o_own[k] = r * lp + x0;
        o_ym[k] = o_own[k] - (ly > 0 ? lp : 0);
        o_zm[k] = o_own[k] - (lz > 0 ? by * lp : 0);
        o_yp[k] = o_own[k] + (ly + 1 < by ? lp : 0);
        o_zp[k] = o_own[k] + (lz + 1 < bz ? by * lp : 0);
        const bool face = k == 0;                                      // (rows of the later slots are interior rows)
        const bool first_y = face && on && ly == 0 && has_ym, last_y = face && on && ly == by - 1 && has_yp;
        const bool first_z = face && on && lz == 0 && has_zm, last_z = face && on && lz == bz - 1 && has_zp;
        if (face) {
            pub_y = first_y ? xrow(wg, 0, lz) : last_y ? xrow(wg, 1, lz) : -1;
            pub_z = first_z ? xrow(wg, 2, ly) : last_z ? xrow(wg, 3, ly) : -1;
        }
        const bool row_interior = y > 0 && y < ny - 1 && z > 0 && z < nz - 1;
        unsigned kd = (on ? kOn : 0u) | (ly > 0 ? kHasYm : 0u) | (lz > 0 ? kHasZm : 0u) | (ly + 1 < by ? kHasYp : 0u) |
                      (lz + 1 < bz ? kHasZp : 0u) | (first_y ? kFirstY : 0u) | (first_z ? kFirstZ : 0u) |
                      (last_y ? kLastY : 0u) | (last_z ? kLastZ : 0u) | (row_interior && x0 > 0 && x0 < nx - 1 ? kM0 : 0u) |
                      (row_interior && x0 + 1 < nx - 1 ? kM1 : 0u) | (row_interior && x0 + 2 < nx - 1 ? kM2 : 0u) |
                      (row_interior && x0 + 3 < nx - 1 ? kM3 : 0u);
        p4[k] = vx4[k] = vy4[k] = vz4[k] = (f4){0.f, 0.f, 0.f, 0.f};
        if (on) {
            const size_t cells = (size_t)(nx - x0 < 4 ? nx - x0 : 4);  // a row that is no multiple of four ends in a part quad
            if (rcv >= pi && rcv < pi + cells) kd |= (unsigned)(rcv - pi + 1) << kRcvShift;
            if (src >= pi && src < pi + cells) kd |= (unsigned)(src - pi + 1) << kSrcShift;
            p4[k] = ld(f.p + pi);
            vx4[k] = ld(f.vx + ((size_t)z * ny + y) * px + x0);
            vy4[k] = ld(f.vy + ((size_t)z * (ny + 1) + y) * nx + x0);
            vz4[k] = ld(f.vz + pi);
            if (first_y) gpy = ld(f.p + pi - nx);
            if (last_y) {
                gpy = ld(f.p + pi + nx);
                gfy = ld(f.vy + ((size_t)z * (ny + 1) + y + 1) * nx + x0);
            }
            if (first_z) gpz = ld(f.p + pi - sxy);
            if (last_z) {
                gpz = ld(f.p + pi + sxy);
                gfz = ld(f.vz + pi + sxy);
            }
            *reinterpret_cast<f4*>(sp + o_own[k]) = p4[k];
        }
        kind[k] = kd;
    }
    __syncthreads();

    // where this thread's ghost quads come from (byte offsets into one parity of xbuf); a thread that needs none
    // asks beyond the descriptor's range: the hardware drops that load and returns zeros
    const bool need_y = pub_y >= 0, need_z = pub_z >= 0;
    const unsigned kNowhere = 0xfffff000u;
    const unsigned get_y = need_y ? 4u * (unsigned)(pub_y + ((kind[0] & kFirstY) ? -up_y : up_y)) : kNowhere;
    const unsigned get_z = need_z ? 4u * (unsigned)(pub_z + ((kind[0] & kFirstZ) ? -up_z : up_z)) : kNowhere;
    const auto xsrd0 = __builtin_amdgcn_make_buffer_rsrc(xbuf, 0, 4 * parity_dwords, 0x00020000);
    const auto xsrd1 = __builtin_amdgcn_make_buffer_rsrc(xbuf + parity_dwords, 0, 4 * parity_dwords, 0x00020000);
    auto ask = [&](bool odd, unsigned off) -> u4 {                     // one 16-byte sc1 load the compiler keeps track of
        return odd ? __builtin_amdgcn_raw_buffer_load_b128(xsrd1, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(xsrd0, off, 0, 16);
    };

#ifdef GAB_ABLATE
    unsigned phase_acc[6] = {0u, 0u, 0u, 0u, 0u, 0u};            // clocks from a step's start to each mark, summed in registers
#endif
    unsigned tag = tag_base;
    bool dead = false;                                                  // a neighbour never arrived: stop waiting for good
    const int last = first_sample + n_samples;
    unsigned step = 0;
    for (int smp = first_sample; smp < last; ++smp) {
        for (int st = 0; st < steps_per_sample; ++st, ++step) {
            const bool closes = st == steps_per_sample - 1;
            ++tag;
            // The order inside a step hides the hand-off's round trip (about 1.5 us from request to data) behind
            // everything that does not need it — only the face rows' own pressures do:
            //   ask | low faces of every row from the block's own pressures | barrier | pressures of the INTERIOR rows
            //   | the quads that arrived -> the face rows' ghost-dependent faces -> their pressures, stored at once
            //   | barrier
#ifdef GAB_ABLATE
            const bool stamping = (g_res_ablate & 16) && (tid & 63) == 0 && wg == (int)gridDim.x / 2 + 1;
            const unsigned long long t_step = __builtin_amdgcn_s_memtime();
#define GAB_RSTAMP(i) do { if (stamping) phase_acc[i] += (unsigned)(__builtin_amdgcn_s_memtime() - t_step); } while (0)
#else
#define GAB_RSTAMP(i) do {} while (0)
#endif
            // ---- X, first half: ask for the neighbours' boundary quads of the previous step
            const bool odd = ((step - 1) & 1) != 0;
            const bool polls = step > 0 && !dead;
            const unsigned ay = polls ? get_y : kNowhere, az = polls ? get_z : kNowhere;
            u4 a0 = ask(odd, ay), a1 = ask(odd, ay + 512), b0 = ask(odd, az), b1 = ask(odd, az + 512);
            __builtin_amdgcn_sched_barrier(0);
            // ---- V: the low faces from the old pressures.  A row with no neighbour in a direction (the room's
            // first row or plane, the row's first cell) reads its OWN pressure there: the difference is +0,
            // -c1 * +0 is -0 for c1 > 0 (checked by the host), and f + -0 = f for every f — the face keeps its
            // bits without a branch.  The rows that wait for a neighbour BLOCK's pressures do the same here and
            // get their real update below, once those have arrived (no other row reads those faces).
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                unsigned kd = kind[k];
                asm volatile("" : "+v"(kd));                        // re-derive the row's predicates here, not in 80 hoisted SGPRs
                const f4 pc = p4[k];
                float pl = lane_before(pc.w);                            // the cell before the quad: the previous lane's last
                pl = x0 > 0 ? pl : pc.x;
                if (!(kd & kOn)) continue;
                const f4 pym = *reinterpret_cast<const f4*>(sp + o_ym[k]);
                const f4 pzm = *reinterpret_cast<const f4*>(sp + o_zm[k]);
                vx4[k] = fnma4(c1, pc, (f4){pl, pc.x, pc.y, pc.z}, vx4[k]);
                vy4[k] = fnma4(c1, pc, pym, vy4[k]);
                vz4[k] = fnma4(c1, pc, pzm, vz4[k]);
                *reinterpret_cast<f4*>(svy + o_own[k]) = vy4[k];
                *reinterpret_cast<f4*>(svz + o_own[k]) = vz4[k];
            }
            GAB_RSTAMP(0);
            __syncthreads();
            GAB_RSTAMP(1);
            // ---- P: pressure from the new faces
            unsigned* const xb = xbuf + (step & 1) * parity_dwords;
            auto pressure = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                unsigned kd = kind[k];
                asm volatile("" : "+v"(kd));
                const float fxn = lane_after(vx4[k].x);                 // the face after the quad: the next lane's first
                if (!(kd & kOn)) return;
                f4 hy = *reinterpret_cast<const f4*>(svy + o_yp[k]);
                f4 hz = *reinterpret_cast<const f4*>(svz + o_zp[k]);
                if (k == 0) {
                    if (kd & kLastY) hy = gfy;
                    if (kd & kLastZ) hz = gfz;
                }
                const f4 pc = p4[k], fx = vx4[k];
                const f4 div = (((f4){fx.y, fx.z, fx.w, fxn} - fx) + (hy - vy4[k])) + (hz - vz4[k]);
                const f4 pin = __builtin_elementwise_fma((f4){-c2, -c2, -c2, -c2}, div, pc);
                const f4 pd = pc * damp;
                float pv[4] = {(kd & kM0) ? pin.x : pd.x, (kd & kM1) ? pin.y : pd.y, (kd & kM2) ? pin.z : pd.z,
                               (kd & kM3) ? pin.w : pd.w};
                if (closes && (kd >> kRcvShift) != 0) {                 // the receiver's or the source's quad (two lanes of the room)
                    const int jr = (int)((kd >> kRcvShift) & 7u) - 1, js = (int)((kd >> kSrcShift) & 7u) - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j == jr) strip[smp] = __fmul_rn(pv[j], 0.1f);                       // FDTD3D_OUTPUT_SCALE
                        if (j == js && smp + 1 < last) pv[j] = __fadd_rn(pv[j], inj[smp + 1]);
                    }
                }
                p4[k] = (f4){pv[0], pv[1], pv[2], pv[3]};
                *reinterpret_cast<f4*>(sp + o_own[k]) = p4[k];
            };
            // the interior rows first: they need nothing from outside the block
            if constexpr (RPT > 1) pressure(std::integral_constant<int, 1>());
            __builtin_amdgcn_sched_barrier(0);
            GAB_RSTAMP(2);
            // ---- X, second half: the granules' tags say whether they are that step's (no flag, no fence: guide
            // R2); a thread whose quads were not there yet asks again until they are
            if (polls) {
                const unsigned want = tag - 1;
#ifdef GAB_ABLATE
                unsigned rounds = 0;
#endif
                bool ok = !(need_y || need_z);
#ifdef GAB_ABLATE
                if (g_res_ablate & 1) ok = true;
#endif
                unsigned spins = 0;
                for (;;) {
                    if (!ok) {
                        const bool oky = !need_y || (a0.y == want && a0.w == want && a1.y == want && a1.w == want);
                        const bool okz = !need_z || (b0.y == want && b0.w == want && b1.y == want && b1.w == want);
                        if (oky && okz) {
                            ok = true;
                            if (need_y) gpy = (f4){__uint_as_float(a0.x), __uint_as_float(a0.z), __uint_as_float(a1.x), __uint_as_float(a1.z)};
                            if (need_z) gpz = (f4){__uint_as_float(b0.x), __uint_as_float(b0.z), __uint_as_float(b1.x), __uint_as_float(b1.z)};
                        } else if ((++spins & 1023u) == 0) {           // a long wait: has anyone given up? is it time to?
                            if (spins > kResSpinLimit)
                                __hip_atomic_store(timeout_word, 1u + step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (spins > kResSpinLimit || peek_sc1(timeout_word) != 0) ok = dead = true;
                        }
                    }
                    if (__all(ok)) break;
                    const unsigned ry = ok ? kNowhere : get_y, rz = ok ? kNowhere : get_z;
                    a0 = ask(odd, ry), a1 = ask(odd, ry + 512), b0 = ask(odd, rz), b1 = ask(odd, rz + 512);
#ifdef GAB_ABLATE
                    ++rounds;
#endif
                }
#ifdef GAB_ABLATE
                if ((g_res_ablate & 4) && (tid & 63) == 0 && wg == (int)gridDim.x / 2 + 1) {
                    atomicAdd(&g_res_rounds[0], (unsigned long long)rounds);
                    atomicAdd(&g_res_rounds[1], 1ull);
                }
#endif
            }
            GAB_RSTAMP(3);
            // ---- V, the rows at the block's faces: the low faces that need the neighbour block's pressures (read
            // by this row's own pressure only), and the neighbour's low faces above the block's last row / plane
            // (same operation, same operands as there); then their pressures, stored at once
            {
                unsigned kd = kind[0];
                asm volatile("" : "+v"(kd));
                if (kd & (kFirstY | kFirstZ | kLastY | kLastZ)) {
                    const f4 pc = p4[0];
                    if (kd & kFirstY) vy4[0] = fnma4(c1, pc, gpy, vy4[0]);
                    if (kd & kFirstZ) vz4[0] = fnma4(c1, pc, gpz, vz4[0]);
                    if (kd & kLastY) gfy = fnma4(c1, gpy, pc, gfy);
                    if (kd & kLastZ) gfz = fnma4(c1, gpz, pc, gfz);
                }
            }
            pressure(std::integral_constant<int, 0>());
#ifdef GAB_ABLATE
            if (!((g_res_ablate & 2) && wg == 0))
#endif
            {
                if (pub_y >= 0) publish_quad(xb + pub_y, p4[0], tag);
                if (pub_z >= 0) publish_quad(xb + pub_z, p4[0], tag);
            }
            GAB_RSTAMP(4);
            __syncthreads();
            GAB_RSTAMP(5);
        }
    }
#ifdef GAB_ABLATE
    if ((g_res_ablate & 16) && (tid & 63) == 0 && wg == (int)gridDim.x / 2 + 1)
        for (int i = 0; i < 6; ++i) g_res_phase[(tid >> 6) * 8 + i] += phase_acc[i];
#endif
    // ---- the block's fields go back to memory (the room's last faces never moved: they are still there)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (!(kind[k] & kOn)) continue;
        int ly = 0, lz = 0;
        row_of(slot + kResRowSlots * k, ly, lz);
        const int y = y0 + ly, z = z0 + lz;
        const size_t pi = (size_t)z * sxy + (size_t)y * nx + x0;
        st(f.p + pi, p4[k]);
        st(f.vx + ((size_t)z * ny + y) * px + x0, vx4[k]);
        st(f.vy + ((size_t)z * (ny + 1) + y) * nx + x0, vy4[k]);
        st(f.vz + pi, vz4[k]);
    }
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
    size_t np = 0, nvx = 0, nvy = 0, nvz = 0;   // floats ALLOCATED per field (slab + ghosts)
    // z-slab of a decomposed grid: planes [z_begin, z_end) are owned; the pressure array carries one
    // ghost plane below and above, the vz array one ghost face plane above.  cur / nxt hold VIRTUAL
    // base pointers (real base minus the slab's offset), so the kernels index with global z.
    int z_begin = 0, z_end = 0;
    gab::Fields cur_real{}, nxt_real{};
    // track-dependent source / receiver cells (0 tracks: the shared cells of P)
    int pos_tracks = 0, pos_groups = 0;
    long long *d_pos_rcv = nullptr, *d_pos_group_cell = nullptr;
    int *d_pos_group_start = nullptr, *d_pos_group_tracks = nullptr;
    bool use_graphs = true;
    bool lds_tiles = true;    // rows wide enough to fill a 32-lane row of the LDS-halo kernel (GAB_FDTD_LDS=0: off)
    bool sample_tiles = true; // small rooms: one launch per sample, S steps in a tile (GAB_FDTD_TILE=0: off)
    // the room resident in LDS for a whole buffer (fdtd_resident_kernel): geometry, exchange buffers, flags
    bool resident = true;
    gab::ResidentGeom rgeom{};
    int res_rpt = 0;                    // rows per thread slot (0: the grid does not take the resident kernel)
    size_t res_lds_bytes = 0;
    unsigned* res_xbuf = nullptr;       // [2 parities][workgroups][4 faces][fr rows][128 granules {pressure, tag}] + the timeout word
    size_t res_xbuf_dwords = 0;
    unsigned res_tag = 0;               // steps the resident kernel has run on this plan: the exchange tags go on from here
    unsigned* res_timeout_host = nullptr;   // pinned copy of the timeout word (copied behind every resident launch)
    bool form_step = false;             // gab_fdtd_set_form(GAB_FDTD_FORM_STEP): never the resident kernel
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

void free_track_positions(gab_fdtd_plan* f) {
    if (f->d_pos_rcv) (void)hipFree(f->d_pos_rcv);
    if (f->d_pos_group_cell) (void)hipFree(f->d_pos_group_cell);
    if (f->d_pos_group_start) (void)hipFree(f->d_pos_group_start);
    if (f->d_pos_group_tracks) (void)hipFree(f->d_pos_group_tracks);
    f->d_pos_rcv = f->d_pos_group_cell = nullptr;
    f->d_pos_group_start = f->d_pos_group_tracks = nullptr;
    f->pos_tracks = f->pos_groups = 0;
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

// real base -> the base the kernels index with global z (integer arithmetic: the result may lie
// before the allocation; only planes z_begin-1 .. z_end are ever dereferenced through it)
gab::Fields virtual_base(const gab::Fields& real, const gab_fdtd_plan& pl) {
    const gab_fdtd_params& P = pl.P;
    const long long sxy = (long long)P.nx * P.ny;
    auto shift = [](float* p, long long floats) {
        return reinterpret_cast<float*>(reinterpret_cast<intptr_t>(p) - (intptr_t)(floats * (long long)sizeof(float)));
    };
    gab::Fields v;
    v.p = shift(real.p, ((long long)pl.z_begin - 1) * sxy);
    v.vz = shift(real.vz, (long long)pl.z_begin * sxy);
    v.vx = shift(real.vx, (long long)pl.z_begin * P.ny * (P.nx + 4));
    v.vy = shift(real.vy, (long long)pl.z_begin * (P.ny + 1) * P.nx);
    return v;
}

// Blocks of nx x by x bz cells, one workgroup (one CU) each: the most workgroups the device can hold at once
// whose LDS image fits, at most 64 rows per block (two per thread slot).
// Two resident launches must not share the device: each needs every one of its workgroups on a CU at once, and
// two half-placed grids would wait for each other until the polls time out.  Within this process they are chained:
// a resident launch waits (on its own stream) for the previous one on the same device, whatever stream that was on.
struct ResidentChain {
    std::mutex mu;
    hipEvent_t last[64] = {};
    template <class Launch>
    void run(hipStream_t q, Launch&& launch) {        // wait for the previous one, launch, leave the mark — one at a time
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        std::lock_guard<std::mutex> lock(mu);
        if (known) {
            if (last[dev]) (void)hipStreamWaitEvent(q, last[dev], 0);
            else if (hipEventCreateWithFlags(&last[dev], hipEventDisableTiming) != hipSuccess) last[dev] = nullptr;
        }
        launch();
        if (known && last[dev]) (void)hipEventRecord(last[dev], q);
    }
};
ResidentChain g_resident_chain;

void choose_resident_geometry(gab_fdtd_plan* f) {
    const gab_fdtd_params& P = f->P;
    f->res_rpt = 0;
    if (f->z_begin != 0 || f->z_end != P.nz || P.nx > 128) return;
    if (!(P.dt_over_rho_dx > 0.0f)) return;                      // the kernel's branch-free boundary faces need -c1 * 0 = -0
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return;
    // the kernel is written for gfx950 (sc1 hand-off, 160 KB of LDS per workgroup): any other device, or one that
    // cannot give a workgroup the image, takes the step kernels
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return;
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds_max <= 0) return;
    lds_max = std::min(lds_max, 160 * 1024);
    long best_w = 0, best_surface = 0;
    for (int bz = 2; bz <= 32; ++bz)
        for (int by = 2; by <= 32; ++by) {
            const int rows = by * bz;
            if (rows > 2 * gab::kResRowSlots) continue;
            if (rows > gab::kResRowSlots && 2 * (by + bz) - 4 > gab::kResRowSlots) continue;   // face rows: every thread's first row
            const long gy = (P.ny + by - 1) / by, gz = (P.nz + bz - 1) / bz, w = gy * gz;
            if (w > cus || w < 2) continue;
            const size_t floats = (size_t)3 * rows * (size_t)((P.nx + 3) / 4 * 4);   // p, vy, vz images, rows padded to whole quads
            if (floats * sizeof(float) + 64 > (size_t)lds_max) continue;
            const long surface = by + bz;                       // exchanged rows per block ~ 2 (by + bz)
            if (w > best_w || (w == best_w && surface < best_surface)) {
                best_w = w;
                best_surface = surface;
                f->rgeom = gab::ResidentGeom{by, bz, (int)gy, (int)gz, (P.nx + 3) / 4, rows, by > bz ? by : bz};
                f->res_rpt = (rows + gab::kResRowSlots - 1) / gab::kResRowSlots;
                f->res_lds_bytes = floats * sizeof(float);
            }
        }
}

// The resident form's one-time set-up: the exchange buffer, the kernel's LDS allowance and the proof that every one
// of its workgroups fits the device at once (the kernel's neighbours wait for each other: a grid the device can only
// hold in two rounds would wait for good).  Anything that fails leaves the plan on the step kernels.
bool prepare_resident(gab_fdtd_plan* f, hipStream_t s) {
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (f->res_xbuf) { (void)hipFree(f->res_xbuf); f->res_xbuf = nullptr; }
        f->res_rpt = 0;
        return false;
    };
    const void* fn = f->res_rpt == 1 ? reinterpret_cast<const void*>(gab::fdtd_resident_kernel<1>)
                                     : reinterpret_cast<const void*>(gab::fdtd_resident_kernel<2>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->res_lds_bytes) != hipSuccess) return give_up();
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, gab::kResThreads, f->res_lds_bytes) != hipSuccess)
        return give_up();
    const size_t wgs = (size_t)f->rgeom.gy * f->rgeom.gz;
    if (per_cu < 1 || wgs > (size_t)per_cu * (size_t)cus) return give_up();
    f->res_xbuf_dwords = (size_t)2 * wgs * 4 * f->rgeom.fr * gab::kResRowDwords;
    if (hipMalloc(&f->res_xbuf, sizeof(unsigned) * (f->res_xbuf_dwords + 4)) != hipSuccess) return give_up();
    if (hipMemsetAsync(f->res_xbuf, 0, sizeof(unsigned) * (f->res_xbuf_dwords + 4), s) != hipSuccess) return give_up();   // tag 0 = never written
    if (!f->res_timeout_host) {
        if (hipHostMalloc(&f->res_timeout_host, sizeof(unsigned), hipHostMallocDefault) != hipSuccess) return give_up();
        *f->res_timeout_host = 0;
    }
    return true;
}

// A resident launch that gave up waiting left its word in the pinned copy (it travels behind every resident launch on
// the launch's stream).  Consumes it: the plan takes the step kernels from now on.  0, or 1 + the step it stopped at.
unsigned take_resident_timeout(gab_fdtd_plan* f) {
    if (!f->res_timeout_host || *f->res_timeout_host == 0) return 0;
    const unsigned at = *f->res_timeout_host;
    *f->res_timeout_host = 0;
    f->resident = false;
    return at;
}

int create_slab(gab_fdtd_plan** out, const gab_fdtd_params* params, int z_begin, int z_end, const char* who) {
    if (!out || !params) return gab::bad_arg((std::string(who) + ": null pointer").c_str());
    if (int rc = gab::refuse_unsupported_runtime_mode(who)) return rc;
    const gab_fdtd_params& P = *params;
    if (P.nx < 3 || P.ny < 3 || P.nz < 3) return gab::bad_arg((std::string(who) + ": grid too small").c_str());
    auto inside = [&](int x, int y, int z) {
        return x >= 0 && x < P.nx && y >= 0 && y < P.ny && z >= 0 && z < P.nz;
    };
    if (!inside(P.source_x, P.source_y, P.source_z) || !inside(P.receiver_x, P.receiver_y, P.receiver_z))
        return gab::bad_arg((std::string(who) + ": source/receiver outside the grid").c_str());
    if (P.steps_per_sample < 1) return gab::bad_arg((std::string(who) + ": steps_per_sample must be >= 1").c_str());
    if (z_begin < 0 || z_end > P.nz || z_end - z_begin < 1)
        return gab::bad_arg((std::string(who) + ": slab must be a non-empty range inside [0, nz)").c_str());
    auto* f = new gab_fdtd_plan;
    f->P = P;
    f->z_begin = z_begin;
    f->z_end = z_end;
#ifdef GAB_ABLATE       // diagnostic builds: pick a kernel form by hand (every form is bit-identical)
    if (const char* v = getenv("GAB_FDTD_RESIDENT")) f->resident = atoi(v) != 0;
    if (const char* v = getenv("GAB_FDTD_RES_ABLATE")) {
        const int a = atoi(v);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gab::g_res_ablate), &a, sizeof(int));
    }
    if (const char* v = getenv("GAB_FDTD_GRAPH")) f->use_graphs = atoi(v) != 0;
    if (const char* v = getenv("GAB_FDTD_LDS")) f->lds_tiles = atoi(v) != 0;
    if (const char* v = getenv("GAB_FDTD_TILE")) f->sample_tiles = atoi(v) != 0;
#endif
    const size_t nzl = (size_t)(z_end - z_begin);
    f->np = (size_t)P.nx * P.ny * (nzl + 2);
    f->nvx = (size_t)(P.nx + 4) * P.ny * nzl + 4;    // padded pitch, see file header
    f->nvy = (size_t)P.nx * (P.ny + 1) * nzl;
    f->nvz = (size_t)P.nx * P.ny * (nzl + 1);
    try {
        alloc_fields(f->cur_real, *f);
        alloc_fields(f->nxt_real, *f);
    } catch (...) {
        gab_fdtd_destroy(f);
        throw;
    }
    f->cur = virtual_base(f->cur_real, *f);
    f->nxt = virtual_base(f->nxt_real, *f);
    choose_resident_geometry(f);
    *out = f;
    int rc = gab_fdtd_reset(f, nullptr);
    if (rc) return rc;
    GAB_HIP_CHECK(hipStreamSynchronize(nullptr));
    return GAB_OK;
}

// LDS-halo kernel: the z-neighbour planes are requested before the barrier, with the cell's own
// loads — one memory round trip per step instead of two: 12.9 -> 11.7 us per step at 128^3 (the
// 2 us stagger of odd planes that round 2 used to overlap loads and stores then only costs time)
constexpr bool kFdtdHoist = true;

// One leapfrog step old -> new over the plan's own planes, with the kernel form that suits the row
// width (see the kernels above).
void launch_step(const gab_fdtd_plan* f, hipStream_t q, const gab::Fields& cur, const gab::Fields& nxt,
                 const float* add_next, float* strip_out) {
    const gab_fdtd_params& P = f->P;
    const int nzl = f->z_end - f->z_begin;
    gab::Grid g{P.nx, P.ny, P.nz, P.nx + 4, f->z_begin};
#ifdef GAB_ABLATE
    static const bool hoist = getenv("GAB_FDTD_HOIST") ? atoi(getenv("GAB_FDTD_HOIST")) != 0 : kFdtdHoist;
    static const int tile_env = getenv("GAB_FDTD_TILE_THREADS") ? atoi(getenv("GAB_FDTD_TILE_THREADS")) : 0;
#else
    constexpr bool hoist = kFdtdHoist;
    constexpr int tile_env = 0;
#endif
    const size_t sxy = (size_t)P.nx * P.ny;
    const size_t src = P.source_z * sxy + (size_t)P.source_y * P.nx + P.source_x;
    const size_t rcv = P.receiver_z * sxy + (size_t)P.receiver_y * P.nx + P.receiver_x;
    const float damp = 1.0f - P.absorption_coeff;
    const bool vec4 = (P.nx % 4) == 0;
    const int tx = vec4 ? P.nx / 4 : P.nx;                    // threads along x
    const int bx = tx >= 64 ? 64 : (tx >= 32 ? 32 : 16);
    dim3 block(bx, 256 / bx, 1);
    dim3 grid((tx + bx - 1) / bx, (P.ny + block.y - 1) / block.y, nzl);
    if (vec4 && f->lds_tiles && tx > 16 && tx <= 64) {
        // whole x extent in one workgroup: LX x ROWS threads, (ROWS + 2) pressure rows in LDS
#define GAB_FDTD_LDS_LAUNCH(LX, ROWS)                                                                 \
    do {                                                                                              \
        if (hoist)                                                                                    \
            gab::fdtd_step_lds_kernel<LX, ROWS, true><<<dim3(1, (P.ny + ROWS - 1) / ROWS, nzl), dim3(LX, ROWS, 1), 0, q>>>( \
                cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, damp, src, rcv, add_next, strip_out);  \
        else                                                                                          \
            gab::fdtd_step_lds_kernel<LX, ROWS, false><<<dim3(1, (P.ny + ROWS - 1) / ROWS, nzl), dim3(LX, ROWS, 1), 0, q>>>( \
                cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, damp, src, rcv, add_next, strip_out); \
    } while (0)
        // 512-thread tiles (half as many halo rows) once they still make >= 4 workgroups
        // per CU; 256-thread tiles below that (measured: 13.6 vs 14.3 us/step at 128^3,
        // 47.5 vs 49.6 at 200^3, but 9.5 vs 8.8 at 96^3)
        const int lx = tx <= 32 ? 32 : 64;
        bool big = (long)((P.ny + 512 / lx - 1) / (512 / lx)) * nzl >= 1024;
        if (tile_env) big = tile_env == 512;
        if (lx == 32 && big) GAB_FDTD_LDS_LAUNCH(32, 16);
        else if (lx == 32) GAB_FDTD_LDS_LAUNCH(32, 8);
        else if (big) GAB_FDTD_LDS_LAUNCH(64, 8);
        else GAB_FDTD_LDS_LAUNCH(64, 4);
#undef GAB_FDTD_LDS_LAUNCH
    } else if (vec4) {
        gab::fdtd_step_vec4_kernel<<<grid, block, 0, q>>>(cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, damp,
                                                         src, rcv, add_next, strip_out);
    } else {
        gab::fdtd_step_kernel<<<grid, block, 0, q>>>(cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, damp, src,
                                                    rcv, add_next, strip_out);
    }
}

void ensure_strips(gab_fdtd_plan* f, int bufsize, hipStream_t s) {
    if (f->strip_cap >= bufsize) return;
    GAB_HIP_CHECK(hipStreamSynchronize(s));
    for (auto& c : f->graphs) (void)hipGraphExecDestroy(c.second);   // they point at the old strips
    f->graphs.clear();
    if (f->inj) (void)hipFree(f->inj);
    if (f->strip) (void)hipFree(f->strip);
    f->inj = f->strip = nullptr;
    GAB_HIP_CHECK(hipMalloc(&f->inj, sizeof(float) * bufsize));
    GAB_HIP_CHECK(hipMalloc(&f->strip, sizeof(float) * bufsize));
    GAB_HIP_CHECK(hipMemsetAsync(f->strip, 0, sizeof(float) * bufsize, s));
    f->strip_cap = bufsize;
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
        return create_slab(out, params, 0, params ? params->nz : 0, "gab_fdtd_create");
    });
}

int gab_fdtd_create_slab(gab_fdtd_plan** out, const gab_fdtd_params* params, int z_begin, int z_end) {
    return gab::guarded([&]() -> int { return create_slab(out, params, z_begin, z_end, "gab_fdtd_create_slab"); });
}

int gab_fdtd_destroy(gab_fdtd_plan* f) {
    if (!f) return GAB_OK;
    (void)hipDeviceSynchronize();
    const unsigned gave_up_at = take_resident_timeout(f);     // a last call nobody asked about
    for (auto& g : f->graphs) (void)hipGraphExecDestroy(g.second);
    if (f->capture_stream) (void)hipStreamDestroy(f->capture_stream);
    free_fields(f->cur_real);
    free_fields(f->nxt_real);
    if (f->inj) (void)hipFree(f->inj);
    if (f->strip) (void)hipFree(f->strip);
    if (f->res_xbuf) (void)hipFree(f->res_xbuf);
    if (f->res_timeout_host) (void)hipHostFree(f->res_timeout_host);
    free_track_positions(f);
    delete f;
    if (gave_up_at) {
        gab::set_last_error("gab_fdtd_destroy: the plan's last LDS-resident launch had timed out at step " + std::to_string(gave_up_at - 1) +
                            " waiting for a neighbour workgroup; that call's output was NaN");
        return GAB_ERR_RUNTIME;
    }
    return GAB_OK;
}

int gab_fdtd_reset(gab_fdtd_plan* f, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_reset: null plan");
        hipStream_t s = gab::as_stream(stream);
        zero_fields(f->cur_real, *f, s);
        zero_fields(f->nxt_real, *f, s);
        if (const unsigned at = take_resident_timeout(f)) {       // the fields are defined again; the caller still hears of it
            gab::set_last_error("gab_fdtd_reset: an earlier LDS-resident launch had timed out at step " + std::to_string(at - 1) +
                                " waiting for a neighbour workgroup (its output was NaN); the plan is reset and takes the step kernels from now on");
            return GAB_ERR_RUNTIME;
        }
        return GAB_OK;
    });
}

#ifdef GAB_ABLATE
extern "C" int gab_debug_fdtd_rounds(unsigned long long* h_out) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gab::g_res_rounds), sizeof(unsigned long long) * 4);
}
extern "C" int gab_debug_fdtd_phases(unsigned long long* h_out) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gab::g_res_phase), sizeof(unsigned long long) * 128);
}
#endif

int gab_fdtd_set_form(gab_fdtd_plan* f, int form) {
    if (!f) return gab::bad_arg("gab_fdtd_set_form: null plan");
    if (form != GAB_FDTD_FORM_AUTO && form != GAB_FDTD_FORM_STEP) return gab::bad_arg("gab_fdtd_set_form: unknown form");
    f->form_step = form == GAB_FDTD_FORM_STEP;
    return GAB_OK;
}

int gab_fdtd_status(gab_fdtd_plan* f, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_status: null plan");
        GAB_HIP_CHECK(hipStreamSynchronize(gab::as_stream(stream)));
        if (const unsigned at = take_resident_timeout(f)) {
            gab::set_last_error("gab_fdtd_status: the LDS-resident launch of the last gab_fdtd_process timed out at step " +
                                std::to_string(at - 1) + " waiting for a neighbour workgroup (device shared with other work?); its "
                                "output is NaN and the plan's fields are undefined: reset it (it takes the step kernels from now on)");
            return GAB_ERR_RUNTIME;
        }
        return GAB_OK;
    });
}

int gab_fdtd_resident(const gab_fdtd_plan* f, int* resident, int* workgroups) {
    if (!f) return gab::bad_arg("gab_fdtd_resident: null plan");
    const bool r = f->resident && !f->form_step && f->res_rpt > 0 && !f->pos_tracks;
    if (resident) *resident = r ? 1 : 0;
    if (workgroups) *workgroups = r ? f->rgeom.gy * f->rgeom.gz : 0;
    return GAB_OK;
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
        if (f->z_begin != 0 || f->z_end != P.nz)
            return gab::bad_arg("gab_fdtd_process: a z-slab is stepped with gab_fdtd_step (halo exchange between steps)");
        hipStream_t s = gab::as_stream(stream);
        ensure_strips(f, bufsize, s);
        const size_t sxy = (size_t)P.nx * P.ny;
        const size_t src = P.source_z * sxy + (size_t)P.source_y * P.nx + P.source_x;
        const int last = first_sample + n_samples;

        // rooms up to 56 cells wide (where one step is shorter than a kernel boundary) take one
        // launch per SAMPLE: three steps inside a tile (fdtd_sample_tile_kernel).  Measured per step:
        // 20^3 1.98 vs 3.8 us, 32^3 2.11 vs 3.7, 52^3 3.49 vs 4.2, 56^3 3.59; at 64^3 the one-step chain wins
        // a previous resident launch that gave up waiting for a neighbour workgroup left its word here
        if (const unsigned at = take_resident_timeout(f)) {
            gab::set_last_error("gab_fdtd_process: the LDS-resident kernel of the PREVIOUS call timed out at step " +
                                std::to_string(at - 1) + " waiting for a neighbour workgroup (device shared with other "
                                "work?); the plan's fields are undefined, reset it");
            return GAB_ERR_RUNTIME;
        }
        // (not inside a caller's stream capture: the exchange tags are a launch argument that a replay would freeze)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        bool resident = f->resident && !f->form_step && f->res_rpt > 0 && !f->pos_tracks && cap == hipStreamCaptureStatusNone;
        if (resident && !f->res_xbuf) resident = prepare_resident(f, s);     // false: this device cannot hold the grid at once
        const bool by_sample = f->sample_tiles && !f->pos_tracks && P.steps_per_sample == 3 && f->z_begin == 0 &&
                               f->z_end == P.nz && P.nx <= 56 && P.ny <= 56 && P.nz <= 56;
        // enqueue the whole chain on `q`, walking local copies of the ping-pong pair
        if (f->pos_tracks && f->pos_tracks != tracks)
            return gab::bad_arg("gab_fdtd_process: the plan has track positions for a different track count");
        auto enqueue = [&](hipStream_t q, gab::Fields cur, gab::Fields nxt) {
            if (f->pos_tracks) {
                auto io = [&](float* p, int extract, int inject) {
                    gab::fdtd_track_io_kernel<<<1, 256, 0, q>>>(p, d_in, d_out, tracks, bufsize, extract, inject,
                                                               f->d_pos_rcv, f->pos_groups, f->d_pos_group_cell,
                                                               f->d_pos_group_start, f->d_pos_group_tracks);
                };
                io(cur.p, -1, first_sample);
                for (int smp = first_sample; smp < last; ++smp) {
                    for (int step = 0; step < P.steps_per_sample; ++step) {
                        launch_step(f, q, cur, nxt, nullptr, nullptr);
                        std::swap(cur, nxt);
                    }
                    io(cur.p, smp, smp + 1 < last ? smp + 1 : -1);
                }
                return;
            }
            gab::fdtd_source_sums_kernel<<<(n_samples + 127) / 128, 128, 0, q>>>(d_in, f->inj, tracks, bufsize,
                                                                               first_sample, n_samples);
            // the first sample's source goes straight into the current pressure grid; later
            // ones are folded into the step that precedes them
            gab::fdtd_add_source_kernel<<<1, 64, 0, q>>>(cur.p, src, f->inj, first_sample);
            if (resident) {
                // the whole buffer in ONE launch, the room resident in LDS (fields updated in place in `cur`)
                const gab::Grid g{P.nx, P.ny, P.nz, P.nx + 4, 0};
                const size_t rcv = P.receiver_z * sxy + (size_t)P.receiver_y * P.nx + P.receiver_x;
                const unsigned wgs = (unsigned)(f->rgeom.gy * f->rgeom.gz);
                unsigned* const tmo = f->res_xbuf + f->res_xbuf_dwords;
#define GAB_RESIDENT_LAUNCH(RPT)                                                                                   \
    gab::fdtd_resident_kernel<RPT><<<dim3(wgs), dim3(gab::kResThreads), f->res_lds_bytes, q>>>(                    \
        cur, g, f->rgeom, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, 1.0f - P.absorption_coeff, src, rcv, f->inj, f->strip, \
        first_sample, n_samples, P.steps_per_sample, f->res_xbuf, f->res_tag, tmo)
                g_resident_chain.run(q, [&]() { if (f->res_rpt == 1) GAB_RESIDENT_LAUNCH(1); else GAB_RESIDENT_LAUNCH(2); });
#undef GAB_RESIDENT_LAUNCH
                f->res_tag += (unsigned)n_samples * (unsigned)P.steps_per_sample;
                dim3 bgrid((n_samples + 127) / 128, tracks);
                gab::fdtd_broadcast_kernel<<<bgrid, 128, 0, q>>>(f->strip, d_out, tracks, bufsize, first_sample,
                                                                n_samples, tmo, f->res_timeout_host);
                return;
            }
            if (by_sample) {
                constexpr int TY = 4, TZ = 4, NT = 576;             // 56 x (4 + 6) columns fit 576 threads
                const gab::Grid g{P.nx, P.ny, P.nz, P.nx + 4, 0};
                const size_t rcv = P.receiver_z * sxy + (size_t)P.receiver_y * P.nx + P.receiver_x;
                const dim3 grid((P.ny + TY - 1) / TY, (P.nz + TZ - 1) / TZ);
                const int cols = P.nx * (TY + 6);
                const dim3 block((cols + 63) / 64 * 64);
                for (int smp = first_sample; smp < last; ++smp) {
                    gab::fdtd_sample_tile_kernel<TY, TZ, 3, NT><<<grid, block, 0, q>>>(
                        cur, nxt, g, P.dt_over_rho_dx, P.rho_c2_dt_over_dx, 1.0f - P.absorption_coeff, src, rcv,
                        smp + 1 < last ? f->inj + smp + 1 : nullptr, f->strip + smp);
                    std::swap(cur, nxt);
                }
                dim3 bgrid((n_samples + 127) / 128, tracks);
                gab::fdtd_broadcast_kernel<<<bgrid, 128, 0, q>>>(f->strip, d_out, tracks, bufsize, first_sample,
                                                                n_samples);
                return;
            }
            for (int smp = first_sample; smp < last; ++smp) {
                for (int step = 0; step < P.steps_per_sample; ++step) {
                    const bool closes = step == P.steps_per_sample - 1;
                    const float* add_next = (closes && smp + 1 < last) ? f->inj + smp + 1 : nullptr;
                    float* strip_out = closes ? f->strip + smp : nullptr;
                    launch_step(f, q, cur, nxt, add_next, strip_out);
                    std::swap(cur, nxt);
                }
            }
            dim3 bgrid((n_samples + 127) / 128, tracks);
            gab::fdtd_broadcast_kernel<<<bgrid, 128, 0, q>>>(f->strip, d_out, tracks, bufsize, first_sample,
                                                            n_samples);
        };
        const long swaps = resident ? 0 : (long)n_samples * (by_sample ? 1 : P.steps_per_sample);
        const long launches = 3L + swaps;
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
        if (swaps & 1) {
            std::swap(f->cur, f->nxt);
            std::swap(f->cur_real, f->nxt_real);
        }
        return gab::launch_status("fdtd kernels");
    });
}

int gab_fdtd_set_track_positions(gab_fdtd_plan* f, const int* src_xyz, const int* rcv_xyz, int tracks) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_set_track_positions: null plan");
        const gab_fdtd_params& P = f->P;
        if (f->z_begin != 0 || f->z_end != P.nz)
            return gab::bad_arg("gab_fdtd_set_track_positions: not available on a z-slab");
        if (tracks < 0 || (tracks > 0 && (!src_xyz || !rcv_xyz)))
            return gab::bad_arg("gab_fdtd_set_track_positions: null positions");
        const size_t sxy = (size_t)P.nx * P.ny;
        auto cell = [&](const int* q, long long* out) {
            if (q[0] < 0 || q[0] >= P.nx || q[1] < 0 || q[1] >= P.ny || q[2] < 0 || q[2] >= P.nz) return false;
            *out = (long long)(q[2] * sxy + (size_t)q[1] * P.nx + q[0]);
            return true;
        };
        std::vector<long long> rcv(tracks), src(tracks);
        for (int t = 0; t < tracks; ++t)
            if (!cell(src_xyz + 3 * t, &src[t]) || !cell(rcv_xyz + 3 * t, &rcv[t]))
                return gab::bad_arg("gab_fdtd_set_track_positions: a position lies outside the grid");
        GAB_HIP_CHECK(hipDeviceSynchronize());
        for (auto& c : f->graphs) (void)hipGraphExecDestroy(c.second);     // captured for the old cells
        f->graphs.clear();
        free_track_positions(f);
        if (tracks == 0) return GAB_OK;
        // tracks that share a source cell, in track order (stable sort by cell)
        std::vector<int> order(tracks);
        for (int t = 0; t < tracks; ++t) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return src[a] < src[b]; });
        std::vector<long long> gcell;
        std::vector<int> gstart;
        for (int k = 0; k < tracks; ++k)
            if (k == 0 || src[order[k]] != src[order[k - 1]]) {
                gcell.push_back(src[order[k]]);
                gstart.push_back(k);
            }
        gstart.push_back(tracks);
        auto upload = [](auto** d, const auto& h) {
            GAB_HIP_CHECK(hipMalloc(d, h.size() * sizeof(h[0])));
            GAB_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice));
        };
        upload(&f->d_pos_rcv, rcv);
        upload(&f->d_pos_group_cell, gcell);
        upload(&f->d_pos_group_start, gstart);
        upload(&f->d_pos_group_tracks, order);
        f->pos_tracks = tracks;
        f->pos_groups = (int)gcell.size();
        return GAB_OK;
    });
}

// ---- z-slab stepping (domain decomposition) -------------------------------------
int gab_fdtd_owns(const gab_fdtd_plan* f, int* owns_source, int* owns_receiver) {
    if (!f) return gab::bad_arg("gab_fdtd_owns: null plan");
    if (owns_source) *owns_source = f->P.source_z >= f->z_begin && f->P.source_z < f->z_end;
    if (owns_receiver) *owns_receiver = f->P.receiver_z >= f->z_begin && f->P.receiver_z < f->z_end;
    return GAB_OK;
}

int gab_fdtd_source_sums(gab_fdtd_plan* f, const float* d_in, int tracks, int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_in) return gab::bad_arg("gab_fdtd_source_sums: null pointer");
        if (tracks <= 0 || bufsize <= 0) return gab::bad_arg("gab_fdtd_source_sums: sizes must be > 0");
        hipStream_t s = gab::as_stream(stream);
        ensure_strips(f, bufsize, s);
        gab::fdtd_source_sums_kernel<<<(bufsize + 127) / 128, 128, 0, s>>>(d_in, f->inj, tracks, bufsize, 0, bufsize);
        return gab::launch_status("fdtd_source_sums_kernel");
    });
}

int gab_fdtd_inject(gab_fdtd_plan* f, int sample, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_inject: null plan");
        if (sample < 0 || sample >= f->strip_cap) return gab::bad_arg("gab_fdtd_inject: call gab_fdtd_source_sums first");
        const gab_fdtd_params& P = f->P;
        if (P.source_z < f->z_begin || P.source_z >= f->z_end) return GAB_OK;       // another slab's cell
        const size_t sxy = (size_t)P.nx * P.ny;
        const size_t src = P.source_z * sxy + (size_t)P.source_y * P.nx + P.source_x;
        gab::fdtd_add_source_kernel<<<1, 64, 0, gab::as_stream(stream)>>>(f->cur.p, src, f->inj, sample);
        return gab::launch_status("fdtd_add_source_kernel");
    });
}

int gab_fdtd_step(gab_fdtd_plan* f, int strip_sample, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_step: null plan");
        if (strip_sample >= f->strip_cap) return gab::bad_arg("gab_fdtd_step: strip_sample outside the buffer");
        float* strip_out = strip_sample >= 0 ? f->strip + strip_sample : nullptr;
        launch_step(f, gab::as_stream(stream), f->cur, f->nxt, nullptr, strip_out);
        std::swap(f->cur, f->nxt);
        std::swap(f->cur_real, f->nxt_real);
        return gab::launch_status("fdtd step");
    });
}

int gab_fdtd_halo(gab_fdtd_plan* f, int which, float** d_ptr, size_t* n_floats) {
    if (!f || !d_ptr || !n_floats) return gab::bad_arg("gab_fdtd_halo: null pointer");
    const size_t sxy = (size_t)f->P.nx * f->P.ny;
    const size_t nzl = (size_t)(f->z_end - f->z_begin);
    *n_floats = sxy;
    switch (which) {
        case GAB_FDTD_SEND_DOWN_P:  *d_ptr = f->cur_real.p + sxy; break;                 // plane z_begin
        case GAB_FDTD_SEND_DOWN_VZ: *d_ptr = f->cur_real.vz; break;                       // face z_begin
        case GAB_FDTD_SEND_UP_P:    *d_ptr = f->cur_real.p + sxy * nzl; break;            // plane z_end-1
        case GAB_FDTD_RECV_DOWN_P:  *d_ptr = f->cur_real.p; break;                        // ghost z_begin-1
        case GAB_FDTD_RECV_UP_P:    *d_ptr = f->cur_real.p + sxy * (nzl + 1); break;      // ghost z_end
        case GAB_FDTD_RECV_UP_VZ:   *d_ptr = f->cur_real.vz + sxy * nzl; break;           // ghost face z_end
        default: return gab::bad_arg("gab_fdtd_halo: unknown plane selector");
    }
    return GAB_OK;
}

int gab_fdtd_emit(gab_fdtd_plan* f, float* d_out, int tracks, int bufsize, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_out) return gab::bad_arg("gab_fdtd_emit: null pointer");
        if (tracks <= 0 || bufsize <= 0 || bufsize > f->strip_cap)
            return gab::bad_arg("gab_fdtd_emit: sizes must be > 0 and within the recorded strip");
        gab::fdtd_broadcast_kernel<<<dim3((bufsize + 127) / 128, tracks), 128, 0, gab::as_stream(stream)>>>(
            f->strip, d_out, tracks, bufsize, 0, bufsize);
        return gab::launch_status("fdtd_broadcast_kernel");
    });
}

int gab_fdtd_strip(gab_fdtd_plan* f, float** d_strip, int* capacity) {
    if (!f || !d_strip) return gab::bad_arg("gab_fdtd_strip: null pointer");
    *d_strip = f->strip;
    if (capacity) *capacity = f->strip_cap;
    return GAB_OK;
}

int gab_fdtd_copy_pressure(gab_fdtd_plan* f, float* d_dst, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_dst) return gab::bad_arg("gab_fdtd_copy_pressure: null pointer");
        if (f->res_timeout_host && *f->res_timeout_host) {           // left in place: the next process / status / reset consumes it
            gab::set_last_error("gab_fdtd_copy_pressure: an LDS-resident launch timed out waiting for a neighbour workgroup; "
                                "the plan's fields are undefined, reset it");
            return GAB_ERR_RUNTIME;
        }
        const size_t sxy = (size_t)f->P.nx * f->P.ny;
        GAB_HIP_CHECK(hipMemcpyAsync(d_dst, f->cur_real.p + sxy, sxy * (size_t)(f->z_end - f->z_begin) * sizeof(float),
                                     hipMemcpyDeviceToDevice, gab::as_stream(stream)));
        return GAB_OK;
    });
}

}  // extern "C"
