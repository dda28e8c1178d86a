// h_golden.cpp — see h_golden.hpp.  Plain scalar loops, one rounding per
// operation (the library is built with -ffp-contract=off), in the operation
// order of the reference's CPU golden functions.
#include "h_golden.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "gab/benchmarks.hpp"

namespace gab {
namespace golden {

void gain(const float* in, float* out, size_t n, float g) {
    for (size_t i = 0; i < n; ++i) out[i] = g * in[i];
}

void gainstats(const float* in, float* out, float* stats, size_t T, size_t B) {
    for (size_t i = 0; i < T * B; ++i) out[i] = BenchmarkConstants::GAINSTATS_GAIN * in[i];
    for (size_t t = 0; t < T; ++t) {
        float mean = 0.0f, mx = -1e9f;
        const float* x = in + t * B;
        for (size_t s = 0; s < B; ++s) {
            mean += x[s];
            if (x[s] > mx) mx = x[s];
        }
        mean /= B;
        stats[t * GainStatsBenchmark::NSTATS + 0] = mean;
        stats[t * GainStatsBenchmark::NSTATS + 1] = mx;
    }
}

void datatransfer(const float* in, float* out, int in_size, int out_size) {
    using namespace BenchmarkConstants;
    for (int i = 0; i < out_size; ++i)
        out[i] = (i < in_size) ? in[i]
                               : DATATRANSFER_SIGNAL_OFFSET +
                                     DATATRANSFER_SIGNAL_AMPLITUDE * sinf(static_cast<float>(i) * DATATRANSFER_SIGNAL_FREQ);
}

void dft1024(const float* in, float* re, float* im, size_t tracks) {
    const float PI = 3.14159265358979323846f;
    const int size = FFTBenchmark::FFT_SIZE, bins = size / 2 + 1;
    for (size_t t = 0; t < tracks; ++t) {
        const float* x = in + t * size;
        for (int k = 0; k < bins; ++k) {
            float sr = 0.0f, si = 0.0f;
            for (int n = 0; n < size; ++n) {
                float angle = -2.0f * PI * k * n / size;
                sr += x[n] * cosf(angle);
                si += x[n] * sinf(angle);
            }
            re[t * bins + k] = sr;
            im[t * bins + k] = si;
        }
    }
}

void dft1024_f64(const float* in, double* re, double* im, size_t tracks) {
    const int size = FFTBenchmark::FFT_SIZE, bins = size / 2 + 1;
    std::vector<double> c(size), s(size);
    for (int i = 0; i < size; ++i) {
        c[i] = std::cos(-2.0 * M_PI * i / size);
        s[i] = std::sin(-2.0 * M_PI * i / size);
    }
    for (size_t t = 0; t < tracks; ++t) {
        const float* x = in + t * size;
        for (int k = 0; k < bins; ++k) {
            double sr = 0.0, si = 0.0;
            for (int n = 0; n < size; ++n) {
                int m = (k * n) & (size - 1);
                sr += x[n] * c[m];
                si += x[n] * s[m];
            }
            re[t * bins + k] = sr;
            im[t * bins + k] = si;
        }
    }
}

void iir(const float* in, float* out, const IIRCoefficients* c, float* state, int T, int B) {
    for (int t = 0; t < T; ++t) {
        float z1 = state[t * 2], z2 = state[t * 2 + 1];
        const size_t base = static_cast<size_t>(t) * B;
        for (int i = 0; i < B; ++i) {
            float x = in[base + i];
            float w = x - c->a1 * z1 - c->a2 * z2;
            float y = c->b0 * w + c->b1 * z1 + c->b2 * z2;
            z2 = z1;
            z1 = w;
            out[base + i] = y;
        }
        state[t * 2] = z1;
        state[t * 2 + 1] = z2;
    }
}

void conv1d_rows(const float* in, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T) {
    const long total = static_cast<long>(T) * B;
    for (int t = t_lo; t < t_hi; ++t)
        for (int i = 0; i < B; ++i) {
            float samp = 0.0f;
            for (int j = 0; j < L; ++j) {
                long idx = static_cast<long>(t) * B + i - j;
                if (idx >= 0 && idx < total) samp += ir[static_cast<size_t>(t) * L + j] * in[idx];
            }
            out[static_cast<size_t>(t) * B + i] = samp;
        }
}

// The same loop for tracks [t_lo, t_hi) of a SHARD of T tracks whose input buffer starts `halo` tracks before its first
// own track (the flat index runs over the halo rows too; before them lies either nothing — the job's start — or
// samples no tap of these tracks reaches).  ir and out hold the shard's own tracks only.
void conv1d_shard_rows(const float* in_with_halo, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T,
                       int halo) {
    const long total = static_cast<long>(halo + T) * B;
    for (int t = t_lo; t < t_hi; ++t)
        for (int i = 0; i < B; ++i) {
            float samp = 0.0f;
            for (int j = 0; j < L; ++j) {
                long idx = static_cast<long>(halo + t) * B + i - j;
                if (idx >= 0 && idx < total) samp += ir[static_cast<size_t>(t) * L + j] * in_with_halo[idx];
            }
            out[static_cast<size_t>(t) * B + i] = samp;
        }
}

void conv1d(const float* in, const float* ir, float* out, int L, int B, int T) {
    std::memset(out, 0, sizeof(float) * static_cast<size_t>(T) * B);
    conv1d_rows(in, ir, out, L, B, 0, T, T);
}

void conv_accel_rows(const float* in, const float* ir, float* out, int L, int B, int t_lo, int t_hi, int T) {
    for (int t = t_lo; t < t_hi; ++t)
        for (int s = 0; s < B; ++s) {
            float acc = 0.0f;
            for (int k = 0; k < L; ++k) {
                int idx = s - k;
                if (idx >= 0 && idx < B)
                    acc += in[static_cast<size_t>(t) * B + idx] * ir[static_cast<size_t>(t) * L + k];
            }
            out[static_cast<size_t>(T) * s + t] = acc;
        }
}

void conv_accel(const float* in, const float* ir, float* out, int L, int B, int T) {
    conv_accel_rows(in, ir, out, L, B, 0, T, T);
}

void modal(const float* params, float* out, int n_modes, int B, int out_tracks) {
    std::memset(out, 0, sizeof(float) * static_cast<size_t>(B) * out_tracks);
    const int m = std::min(n_modes, out_tracks);
    const float cexp_real = expf(BenchmarkConstants::MODAL_STATE_INIT_REAL) *
                            cosf(BenchmarkConstants::MODAL_STATE_INIT_IMAG);
    for (int i = 0; i < m; ++i) {
        const float v = params[static_cast<size_t>(i) * ModalBenchmark::NUM_MODE_PARAMS +
                               ModalBenchmark::AMPLITUDE] * cexp_real;
        for (int s = 0; s < B; ++s) out[static_cast<size_t>(i) * B + s] = v;
    }
}

// Mode m: phasor (re, im) turned by 2*pi*freq per sample, out[(m % tracks)*B + i] += amp*re,
// fp32, modes in index order.  (cos, sin) = (float)cos/sin((double)angle): reproducible on
// host and device (the reference's cosf/sinf differ in the last bit between libraries).
void modal_bank(const float* params, float* out, int n_modes, int B, int out_tracks) {
    std::memset(out, 0, sizeof(float) * static_cast<size_t>(B) * out_tracks);
    for (int m = 0; m < n_modes; ++m) {
        const float* p = params + static_cast<size_t>(m) * ModalBenchmark::NUM_MODE_PARAMS;
        const float amp = p[ModalBenchmark::AMPLITUDE];
        const float ang = 2.0f * 3.14159265358979323846f * p[ModalBenchmark::FREQUENCY];
        const float c = static_cast<float>(std::cos(static_cast<double>(ang)));
        const float s = static_cast<float>(std::sin(static_cast<double>(ang)));
        float re = p[ModalBenchmark::STATE_REAL], im = p[ModalBenchmark::STATE_IMAG];
        float* o = out + static_cast<size_t>(m % out_tracks) * B;
        for (int i = 0; i < B; ++i) {
            const float a = re * c, b = im * s, d = re * s, e = im * c;
            re = a - b;
            im = d + e;
            const float contrib = amp * re;
            o[i] = o[i] + contrib;
        }
    }
}

void dwg(const WaveguideState* wgs, float* fwd, float* bwd, const float* in, float* out,
         const DWGParams* p) {
    std::memset(out, 0, sizeof(float) * p->bufferSize);
    for (int g = 0; g < p->numWaveguides; ++g) {
        const WaveguideState& wg = wgs[g];
        const size_t base = static_cast<size_t>(g) * p->maxLength;
        for (int s = 0; s < p->bufferSize; ++s) {
            float x = in[s] * wg.gain;
            int cur = (wg.writePos + s) % wg.length;
            int bp = (cur + wg.length / 2) % wg.length;
            float f = fwd[base + cur] * wg.damping;
            float b = bwd[base + bp] * wg.damping;
            if (cur == wg.inputTapPos) { f += x; b += x; }
            fwd[base + cur] = b * wg.reflection;
            bwd[base + bp] = f * wg.reflection;
            if (cur == wg.outputTapPos && g < p->outputTracks)
                out[s] += (f + b) * BenchmarkConstants::WAVEGUIDE_MIX_FACTOR;
        }
    }
}

void fdtd_placeholder(const float* in, float* out, size_t T, size_t B) {
    for (size_t t = 0; t < T; ++t)
        for (size_t s = 0; s < B; ++s) {
            size_t i = t * B + s;
            out[i] = in[i] * BenchmarkConstants::FDTD3D_OUTPUT_SCALE *
                     cosf(static_cast<float>(s) * BenchmarkConstants::FDTD3D_CPU_REF_FREQ);
        }
}

void fdtd3d(const gab_fdtd_params& P, float* p, float* vx, float* vy, float* vz, const float* in,
            float* out, int T, int B, int first_sample, int n_samples) {
    const int nx = P.nx, ny = P.ny, nz = P.nz;
    const size_t sxy = static_cast<size_t>(nx) * ny;
    const float c1 = P.dt_over_rho_dx, c2 = P.rho_c2_dt_over_dx, damp = 1.0f - P.absorption_coeff;
    const size_t src = P.source_z * sxy + static_cast<size_t>(P.source_y) * nx + P.source_x;
    const size_t rcv = P.receiver_z * sxy + static_cast<size_t>(P.receiver_y) * nx + P.receiver_x;
    for (int s = first_sample; s < first_sample + n_samples; ++s)
        for (int step = 0; step < P.steps_per_sample; ++step) {
            if (step == 0) {
                // atomicAdd order is unspecified in the reference; fixed as: sum in track order, add once
                float acc = 0.0f;
                for (int t = 0; t < T; ++t)
                    acc += in[static_cast<size_t>(t) * B + s] * BenchmarkConstants::FDTD3D_SOURCE_SCALE;
                p[src] += acc;
            }
            for (int z = 0; z < nz; ++z)
                for (int y = 0; y < ny; ++y) {
                    const float* pr = p + z * sxy + static_cast<size_t>(y) * nx;
                    float* ax = vx + (static_cast<size_t>(z) * ny + y) * (nx + 1);
                    float* ay = vy + (static_cast<size_t>(z) * (ny + 1) + y) * nx;
                    float* az = vz + z * sxy + static_cast<size_t>(y) * nx;
                    for (int x = 0; x < nx; ++x) {
                        // single rounding, like the FFMA nvcc emits for v -= c*(p1-p0)
                        if (x > 0) ax[x] = fmaf(-c1, pr[x] - pr[x - 1], ax[x]);
                        if (y > 0) ay[x] = fmaf(-c1, pr[x] - pr[x - nx], ay[x]);
                        if (z > 0) az[x] = fmaf(-c1, pr[x] - pr[x - static_cast<long>(sxy)], az[x]);
                    }
                }
            for (int z = 0; z < nz; ++z)
                for (int y = 0; y < ny; ++y) {
                    float* pr = p + z * sxy + static_cast<size_t>(y) * nx;
                    const float* ax = vx + (static_cast<size_t>(z) * ny + y) * (nx + 1);
                    const float* ay = vy + (static_cast<size_t>(z) * (ny + 1) + y) * nx;
                    const float* az = vz + z * sxy + static_cast<size_t>(y) * nx;
                    const bool edge = z == 0 || z == nz - 1 || y == 0 || y == ny - 1;
                    for (int x = 0; x < nx; ++x) {
                        if (edge || x == 0 || x == nx - 1) {
                            pr[x] *= damp;
                        } else {
                            float div = (ax[x + 1] - ax[x]) + (ay[x + nx] - ay[x]) + (az[x + sxy] - az[x]);
                            pr[x] = fmaf(-c2, div, pr[x]);
                        }
                    }
                }
            if (step == P.steps_per_sample - 1) {
                const float o = p[rcv] * BenchmarkConstants::FDTD3D_OUTPUT_SCALE;
                for (int t = 0; t < T; ++t) out[static_cast<size_t>(t) * B + s] = o;
            }
        }
}

void rndmem(const float* pool, const int* playheads, float* out, int B, int T) {
    for (int t = 0; t < T; ++t) {
        const int ph = playheads[t];
        for (int i = 0; i < B; ++i) out[static_cast<size_t>(T) * i + t] = pool[static_cast<size_t>(ph) + i];
    }
}

}  // namespace golden
}  // namespace gab
