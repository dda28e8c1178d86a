// k_fdtd3d.hip — 3-D staggered-grid acoustic FDTD for gfx950.
//
// Replaces the four kernels of cuda/bench_fdtd3d.cu (velocity :14-57, pressure
// :60-98, inject :101-120, extract :123-139) and the launch sequence of
// runFDTD3DTimeStep (:384-438: 8 launches and one device sync per audio
// sample).  Cell updates use the single-rounding a -= c*d form nvcc emits for
// those kernels (explicit fmaf here and in the oracle), the source injection is
// summed in track order instead of by float atomics, and nothing synchronises
// with the host inside a buffer.
//
// Layouts are the reference's (cuda/bench_fdtd3d.cuh:189-206), x fastest:
//   p [nz][ny][nx], vx [nz][ny][nx+1], vy [nz][ny+1][nx], vz [nz+1][ny][nx].
#include <hip/hip_runtime.h>

#include "gab_common.hpp"

namespace gab {
namespace {

constexpr int kBlock = 256;

struct Grid { int nx, ny, nz; };

// p[src] += 0.1f * in[t*B + s] for t = 0..T-1, in that order (FDTD3D_SOURCE_SCALE).
__global__ void fdtd_inject_kernel(float* __restrict__ p, const float* __restrict__ in, size_t src,
                                   int T, int B, int s) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float v = p[src];
    for (int t = 0; t < T; ++t) v = __fadd_rn(v, __fmul_rn(in[(size_t)t * B + s], 0.1f));
    p[src] = v;
}

// Interior faces only: vx for 0<x<nx, vy for 0<y<ny, vz for 0<z<nz.
__global__ __launch_bounds__(kBlock) void fdtd_velocity_kernel(const float* __restrict__ p,
                                                              float* __restrict__ vx,
                                                              float* __restrict__ vy,
                                                              float* __restrict__ vz, Grid g,
                                                              float c1) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    const size_t sxy = (size_t)g.nx * g.ny;
    const size_t pi = z * sxy + (size_t)y * g.nx + x;
    const float pc = p[pi];
    if (x > 0) {
        size_t i = ((size_t)z * g.ny + y) * (g.nx + 1) + x;
        vx[i] = __builtin_fmaf(-c1, __fsub_rn(pc, p[pi - 1]), vx[i]);
    }
    if (y > 0) {
        size_t i = ((size_t)z * (g.ny + 1) + y) * g.nx + x;
        vy[i] = __builtin_fmaf(-c1, __fsub_rn(pc, p[pi - g.nx]), vy[i]);
    }
    if (z > 0) {
        size_t i = pi;   // vz has the same x/y strides
        vz[i] = __builtin_fmaf(-c1, __fsub_rn(pc, p[pi - sxy]), vz[i]);
    }
}

// Interior: p -= c2 * div v; boundary shell: p *= (1 - absorption).
// When `out` is set (last sub-step of a sample) the receiver cell's thread
// writes out[t*B + s] = 0.1f * p[rcv] for every track (FDTD3D_OUTPUT_SCALE).
__global__ __launch_bounds__(kBlock) void fdtd_pressure_kernel(float* __restrict__ p,
                                                              const float* __restrict__ vx,
                                                              const float* __restrict__ vy,
                                                              const float* __restrict__ vz, Grid g,
                                                              float c2, float damp, size_t rcv,
                                                              float* __restrict__ out, int T, int B,
                                                              int s) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    const size_t sxy = (size_t)g.nx * g.ny;
    const size_t pi = z * sxy + (size_t)y * g.nx + x;
    float pv = p[pi];
    const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && z > 0 && z < g.nz - 1;
    if (interior) {
        size_t ix = ((size_t)z * g.ny + y) * (g.nx + 1) + x;
        size_t iy = ((size_t)z * (g.ny + 1) + y) * g.nx + x;
        float dx = __fsub_rn(vx[ix + 1], vx[ix]);
        float dy = __fsub_rn(vy[iy + g.nx], vy[iy]);
        float dz = __fsub_rn(vz[pi + sxy], vz[pi]);
        float div = __fadd_rn(__fadd_rn(dx, dy), dz);
        pv = __builtin_fmaf(-c2, div, pv);
    } else {
        pv = __fmul_rn(pv, damp);
    }
    p[pi] = pv;
    if (out != nullptr && pi == rcv) {
        float o = __fmul_rn(pv, 0.1f);
        for (int t = 0; t < T; ++t) out[(size_t)t * B + s] = o;
    }
}

}  // namespace
}  // namespace gab

struct gab_fdtd_plan {
    gab_fdtd_params P;
    float *p = nullptr, *vx = nullptr, *vy = nullptr, *vz = nullptr;
    size_t np = 0, nvx = 0, nvy = 0, nvz = 0;
};

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
        f->np = (size_t)P.nx * P.ny * P.nz;
        f->nvx = (size_t)(P.nx + 1) * P.ny * P.nz;
        f->nvy = (size_t)P.nx * (P.ny + 1) * P.nz;
        f->nvz = (size_t)P.nx * P.ny * (P.nz + 1);
        try {
            GAB_HIP_CHECK(hipMalloc(&f->p, f->np * sizeof(float)));
            GAB_HIP_CHECK(hipMalloc(&f->vx, f->nvx * sizeof(float)));
            GAB_HIP_CHECK(hipMalloc(&f->vy, f->nvy * sizeof(float)));
            GAB_HIP_CHECK(hipMalloc(&f->vz, f->nvz * sizeof(float)));
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
    if (f->p) (void)hipFree(f->p);
    if (f->vx) (void)hipFree(f->vx);
    if (f->vy) (void)hipFree(f->vy);
    if (f->vz) (void)hipFree(f->vz);
    delete f;
    return GAB_OK;
}

int gab_fdtd_reset(gab_fdtd_plan* f, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f) return gab::bad_arg("gab_fdtd_reset: null plan");
        hipStream_t s = gab::as_stream(stream);
        GAB_HIP_CHECK(hipMemsetAsync(f->p, 0, f->np * sizeof(float), s));
        GAB_HIP_CHECK(hipMemsetAsync(f->vx, 0, f->nvx * sizeof(float), s));
        GAB_HIP_CHECK(hipMemsetAsync(f->vy, 0, f->nvy * sizeof(float), s));
        GAB_HIP_CHECK(hipMemsetAsync(f->vz, 0, f->nvz * sizeof(float), s));
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
        const gab_fdtd_params& P = f->P;
        hipStream_t s = gab::as_stream(stream);
        gab::Grid g{P.nx, P.ny, P.nz};
        const size_t sxy = (size_t)P.nx * P.ny;
        const size_t src = P.source_z * sxy + (size_t)P.source_y * P.nx + P.source_x;
        const size_t rcv = P.receiver_z * sxy + (size_t)P.receiver_y * P.nx + P.receiver_x;
        const float damp = 1.0f - P.absorption_coeff;
        dim3 block(64, 4, 1);
        dim3 grid((P.nx + 63) / 64, (P.ny + 3) / 4, P.nz);
        for (int smp = first_sample; smp < first_sample + n_samples; ++smp) {
            gab::fdtd_inject_kernel<<<1, 64, 0, s>>>(f->p, d_in, src, tracks, bufsize, smp);
            for (int step = 0; step < P.steps_per_sample; ++step) {
                gab::fdtd_velocity_kernel<<<grid, block, 0, s>>>(f->p, f->vx, f->vy, f->vz, g,
                                                                 P.dt_over_rho_dx);
                const bool last = step == P.steps_per_sample - 1;
                gab::fdtd_pressure_kernel<<<grid, block, 0, s>>>(
                    f->p, f->vx, f->vy, f->vz, g, P.rho_c2_dt_over_dx, damp, rcv,
                    last ? d_out : nullptr, tracks, bufsize, smp);
            }
        }
        return gab::launch_status("fdtd kernels");
    });
}

int gab_fdtd_copy_pressure(gab_fdtd_plan* f, float* d_dst, gab_stream_t stream) {
    return gab::guarded([&]() -> int {
        if (!f || !d_dst) return gab::bad_arg("gab_fdtd_copy_pressure: null pointer");
        GAB_HIP_CHECK(hipMemcpyAsync(d_dst, f->p, f->np * sizeof(float), hipMemcpyDeviceToDevice,
                                     gab::as_stream(stream)));
        return GAB_OK;
    });
}

}  // extern "C"
