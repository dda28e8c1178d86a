/*
 * gab_oracle.c — CPU oracle (test infrastructure only; see gab_oracle.h).
 *
 * Plain-C restatement of the reference's CPU golden functions.  Citations are
 * file-local line numbers under the reference's cuda/ directory.
 */
#include "gab_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
static uint64_t fnv1a64_from(uint64_t h, const void* data, size_t nbytes) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

uint64_t orc_fnv1a64(const void* data, size_t nbytes) {
    return fnv1a64_from(0xcbf29ce484222325ULL, data, nbytes);
}

/* The hashes recorded in SURVEY.md §8c were taken with the offset basis
 * 1469598103934665603 (the standard 14695981039346656037 minus its last
 * digit) and the standard prime; found by matching the noise(42) pin, whose
 * values are pinned independently.  Kept so the pins can be compared as
 * recorded.                                                                 */
uint64_t orc_fnv1a64_survey(const void* data, size_t nbytes) {
    return fnv1a64_from(1469598103934665603ULL, data, nbytes);
}

/* ======================================================================== */
/* Random sources                                                           */
/* ======================================================================== */

/* bench_utils.cu:238-245 — std::mt19937 + uniform_real_distribution<float>.
 * libstdc++'s generate_canonical<float,24> takes ONE 32-bit draw per value:
 * u = float(r) / 2^32 (clamped below 1), then u*(b-a)+a with a=-1,b=1.      */
void orc_noise_mt19937(float* buf, size_t n, uint32_t seed) {
    uint32_t mt[624];
    int idx = 624;
    mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (size_t k = 0; k < n; ++k) {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        float u = (float)y / 4294967296.0f;
        if (u >= 1.0f) u = nextafterf(1.0f, 0.0f);
        buf[k] = u * 2.0f + (-1.0f);
    }
}

/* glibc random_r.c TYPE_3: x^31 + x^3 + 1, seeded by the 16807 LCG, first
 * 310 outputs discarded, result = state >> 1.                              */
void orc_srand(orc_rand_t* st, unsigned seed) {
    int32_t word = (int32_t)(seed ? seed : 1u);
    st->r[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        st->r[i] = (uint32_t)word;
    }
    st->f = 3; st->b = 0;
    for (int i = 0; i < 310; ++i) (void)orc_rand(st);
}

int orc_rand(orc_rand_t* st) {
    st->r[st->f] += st->r[st->b];
    uint32_t out = st->r[st->f] >> 1;
    if (++st->f >= 31) st->f = 0;
    if (++st->b >= 31) st->b = 0;
    return (int)out;
}

void orc_rand_unit(orc_rand_t* st, float* buf, size_t n) {
    for (size_t i = 0; i < n; ++i) buf[i] = (float)orc_rand(st) / (float)ORC_RAND_MAX;
}

void orc_rand_bipolar(orc_rand_t* st, float* buf, size_t n) {
    for (size_t i = 0; i < n; ++i)
        buf[i] = ((float)orc_rand(st) / (float)ORC_RAND_MAX) * 2.0f - 1.0f;
}

/* ======================================================================== */
/* gain / gainstats / noop / datatransfer                                   */
/* ======================================================================== */

/* bench_gain.cu:85-95 */
void orc_gain(const float* in, float* out, size_t n, float gain) {
    for (size_t i = 0; i < n; ++i) out[i] = gain * in[i];
}

/* bench_gainstats.cu:120-144 (GAINSTATS_GAIN = 0.5f, benchmark_constants.cuh:7) */
void orc_gainstats(const float* in, float* out, float* stats,
                   size_t tracks, size_t bufsize) {
    for (size_t i = 0; i < tracks * bufsize; ++i) out[i] = 0.5f * in[i];
    for (size_t t = 0; t < tracks; ++t) {
        float mean = 0.0f, maxv = -1e9f;
        for (size_t s = 0; s < bufsize; ++s) {
            float x = in[t * bufsize + s];
            mean += x;
            if (x > maxv) maxv = x;
        }
        mean /= (float)bufsize;
        stats[2 * t + 0] = mean;
        stats[2 * t + 1] = maxv;
    }
}

/* bench_noop.cu:86-93 */
void orc_noop(const float* in, float* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = in[i];
}

/* bench_datatransfer.cu:27-33, bench_datatransfer.cuh:21 */
int orc_datatransfer_size(float ratio) {
    const int base = 10 * 1024 * 1024 / (int)sizeof(float);
    return (int)((float)base * ratio);
}

/* bench_datatransfer.cu:139-147 */
void orc_datatransfer(const float* in, float* out, int in_size, int out_size) {
    for (int i = 0; i < out_size; ++i)
        out[i] = (i < in_size) ? in[i] : 0.5f + 0.5f * sinf((float)i * 0.001f);
}

/* ======================================================================== */
/* FFT                                                                      */
/* ======================================================================== */

/* bench_fft.cu:27-54 */
void orc_fft_input(orc_rand_t* st, float* in, size_t tracks, size_t bufsize) {
    const size_t N = 1024, m = bufsize < N ? bufsize : N;
    for (size_t t = 0; t < tracks; ++t) {
        orc_rand_bipolar(st, in + t * N, m);
        for (size_t i = m; i < N; ++i) in[t * N + i] = 0.0f;
    }
}

/* bench_fft.cu:149-168 (per track), :134-147 (track loop) */
void orc_fft_golden(const float* in, float* re, float* im, size_t tracks) {
    const float PI = 3.14159265358979323846f;
    const int size = 1024, bins = size / 2 + 1;
    for (size_t t = 0; t < tracks; ++t) {
        const float* x = in + t * size;
        for (int k = 0; k < bins; ++k) {
            float sr = 0.0f, si = 0.0f;
            for (int n = 0; n < size; ++n) {
                float angle = -2.0f * PI * (float)k * (float)n / (float)size;
                float c = cosf(angle), s = sinf(angle);
                sr += x[n] * c;
                si += x[n] * s;
            }
            re[t * bins + k] = sr;
            im[t * bins + k] = si;
        }
    }
}

void orc_fft_truth(const float* in, double* re, double* im, size_t tracks) {
    const int size = 1024, bins = size / 2 + 1;
    static double ct[1024], stb[1024];
    for (int i = 0; i < size; ++i) {
        ct[i] = cos(-2.0 * M_PI * i / size);
        stb[i] = sin(-2.0 * M_PI * i / size);
    }
    for (size_t t = 0; t < tracks; ++t) {
        const float* x = in + t * size;
        for (int k = 0; k < bins; ++k) {
            double sr = 0.0, si = 0.0;
            for (int n = 0; n < size; ++n) {
                int m = (int)(((long)k * n) & (size - 1));
                sr += (double)x[n] * ct[m];
                si += (double)x[n] * stb[m];
            }
            re[t * bins + k] = sr;
            im[t * bins + k] = si;
        }
    }
}

/* ======================================================================== */
/* IIR                                                                      */
/* ======================================================================== */

/* bench_iir.cu:199-226 */
orc_iir_coeffs orc_iir_butterworth(float nf) {
    const float PI = 3.14159265358979323846f;
    float omega = 2.0f * PI * nf;
    float cos_omega = cosf(omega);
    float sin_omega = sinf(omega);
    float alpha = sin_omega / (2.0f * 0.707f);
    float b0 = (1.0f - cos_omega) / 2.0f;
    float b1 = 1.0f - cos_omega;
    float b2 = (1.0f - cos_omega) / 2.0f;
    float a0 = 1.0f + alpha;
    float a1 = -2.0f * cos_omega;
    float a2 = 1.0f - alpha;
    orc_iir_coeffs c;
    c.b0 = b0 / a0; c.b1 = b1 / a0; c.b2 = b2 / a0; c.a1 = a1 / a0; c.a2 = a2 / a0;
    return c;
}

/* bench_iir.cu:170-197 */
void orc_iir(const float* in, float* out, const orc_iir_coeffs* c,
             float* state, int tracks, int bufsize) {
    for (int t = 0; t < tracks; ++t) {
        float z1 = state[2 * t], z2 = state[2 * t + 1];
        const size_t base = (size_t)t * bufsize;
        for (int i = 0; i < bufsize; ++i) {
            float x = in[base + i];
            float w = x - c->a1 * z1 - c->a2 * z2;
            float y = c->b0 * w + c->b1 * z1 + c->b2 * z2;
            z2 = z1; z1 = w;
            out[base + i] = y;
        }
        state[2 * t] = z1; state[2 * t + 1] = z2;
    }
}

/* ======================================================================== */
/* conv1d (time domain)                                                     */
/* ======================================================================== */

/* bench_conv1d.cu:159-181 — all-float arithmetic with a float PI           */
void orc_conv1d_ir(float* ir, int L, size_t tracks) {
    const float PI = 3.14159265358979323846f;
    for (size_t t = 0; t < tracks; ++t) {
        for (int i = 0; i < L; ++i) {
            float freq = 0.1f + 0.05f * (float)t / (float)tracks;
            float tt = (float)i - (float)L / 2.0f;
            float window = 0.54f - 0.46f * cosf(2.0f * PI * (float)i / (float)(L - 1));
            float sinc = (tt == 0.0f) ? 1.0f
                       : sinf(2.0f * PI * freq * tt) / (2.0f * PI * freq * tt);
            ir[t * (size_t)L + i] = window * sinc / (float)L;
        }
    }
}

/* bench_conv1d.cu:188-208 — history bleeds in from the previous track's
 * samples of the FLAT buffer; only the very start sees zeros.               */
void orc_conv1d(const float* in, const float* ir, float* out,
                int L, int B, int T) {
    const long total = (long)T * B;
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < B; ++i) {
            float samp = 0.0f;
            for (int j = 0; j < L; ++j) {
                long idx = (long)t * B + i - j;
                if (idx >= 0 && idx < total)
                    samp += ir[(size_t)t * L + j] * in[idx];
            }
            out[(size_t)t * B + i] = samp;
        }
    }
}

/* ======================================================================== */
/* conv1d_accel                                                             */
/* ======================================================================== */

/* bench_conv1d_accel.cu:152-173 — same formula but M_PI is a double, so the
 * trig arguments are formed in double and rounded to float at the call, and
 * the sinc quotient is a double rounded on assignment.                      */
void orc_conv_accel_ir(float* ir, int L, size_t track_offset,
                       size_t n_tracks, size_t total_tracks) {
    for (size_t lt = 0; lt < n_tracks; ++lt) {
        size_t t = track_offset + lt;
        for (int i = 0; i < L; ++i) {
            float freq = 0.1f + 0.05f * (float)t / (float)total_tracks;
            float tt = (float)i - (float)L / 2.0f;
            float window = 0.54f - 0.46f *
                cosf((float)(2.0f * M_PI * (float)i / (float)(L - 1)));
            float sinc = (tt == 0.0f) ? 1.0f
                : (float)(sinf((float)(2.0f * M_PI * freq * tt)) /
                          (2.0f * M_PI * freq * tt));
            ir[lt * (size_t)L + i] = window * sinc / (float)L;
        }
    }
}

/* bench_conv1d_accel.cu:234-252 */
void orc_conv_accel(const float* in, const float* ir, float* out,
                    int L, int B, int T) {
    for (int t = 0; t < T; ++t) {
        for (int s = 0; s < B; ++s) {
            float acc = 0.0f;
            for (int k = 0; k < L; ++k) {
                int ii = s - k;
                if (ii >= 0 && ii < B)
                    acc += in[(size_t)t * B + ii] * ir[(size_t)t * L + k];
            }
            out[(size_t)T * s + t] = acc;
        }
    }
}

/* Streaming extension: x[s-k] for s-k < 0 comes from the carried history.  */
void orc_conv_accel_stream(const float* in, const float* ir, float* out,
                           float* hist, int L, int B, int T) {
    float* line = (float*)malloc(sizeof(float) * (size_t)(L + B));
    for (int t = 0; t < T; ++t) {
        float* h = hist + (size_t)t * L;
        memcpy(line, h, sizeof(float) * L);
        memcpy(line + L, in + (size_t)t * B, sizeof(float) * B);
        const float* taps = ir + (size_t)t * L;
        for (int s = 0; s < B; ++s) {
            float acc = 0.0f;
            const float* x = line + L + s;
            for (int k = 0; k < L; ++k) acc += x[-k] * taps[k];
            out[(size_t)T * s + t] = acc;
        }
        memcpy(h, line + B, sizeof(float) * L);
    }
    free(line);
}

void orc_conv_accel_stream_f64(const float* in, const float* ir, double* out,
                               float* hist, int L, int B, int T) {
    float* line = (float*)malloc(sizeof(float) * (size_t)(L + B));
    for (int t = 0; t < T; ++t) {
        float* h = hist + (size_t)t * L;
        memcpy(line, h, sizeof(float) * L);
        memcpy(line + L, in + (size_t)t * B, sizeof(float) * B);
        const float* taps = ir + (size_t)t * L;
        for (int s = 0; s < B; ++s) {
            double acc = 0.0;
            const float* x = line + L + s;
            for (int k = 0; k < L; ++k) acc += (double)x[-k] * (double)taps[k];
            out[(size_t)T * s + t] = acc;
        }
        memcpy(h, line + B, sizeof(float) * L);
    }
    free(line);
}

/* ======================================================================== */
/* modal (placeholder)                                                      */
/* ======================================================================== */

/* bench_modal.cu:130-145 */
void orc_modal_params(float* params, int n_modes) {
    orc_rand_t st;
    orc_srand(&st, 42);
    for (int i = 0; i < n_modes; ++i) {
        float* p = params + (size_t)i * 8;
        for (int k = 0; k < 7; ++k) p[k] = (float)orc_rand(&st) / (float)ORC_RAND_MAX;
        p[7] = 0.0f;
    }
}

/* metal-swift .../ModalFilterBankBenchmark.swift:73-101 */
static void modal_bank_rotation(float freq, float* c, float* s) {
    const float ang = 2.0f * 3.14159265358979323846f * freq;   /* 2.0 * Float.pi * freq, fp32 */
    *c = (float)cos((double)ang);
    *s = (float)sin((double)ang);
}

void orc_modal_bank(const float* params, float* out, int n_modes, int bufsize, int out_tracks) {
    memset(out, 0, sizeof(float) * (size_t)bufsize * out_tracks);
    for (int m = 0; m < n_modes; ++m) {
        const float* p = params + (size_t)m * 8;
        const float amp = p[0];
        float re = p[3], im = p[4], c, s;
        modal_bank_rotation(p[1], &c, &s);
        float* o = out + (size_t)(m % out_tracks) * bufsize;
        for (int i = 0; i < bufsize; ++i) {
            const float a = re * c, b = im * s, d = re * s, e = im * c;
            re = a - b;
            im = d + e;
            const float contrib = amp * re;
            o[i] = o[i] + contrib;
        }
    }
}

void orc_modal_bank_f64acc(const float* params, double* out, int n_modes, int bufsize, int out_tracks) {
    memset(out, 0, sizeof(double) * (size_t)bufsize * out_tracks);
    for (int m = 0; m < n_modes; ++m) {
        const float* p = params + (size_t)m * 8;
        const float amp = p[0];
        float re = p[3], im = p[4], c, s;
        modal_bank_rotation(p[1], &c, &s);
        double* o = out + (size_t)(m % out_tracks) * bufsize;
        for (int i = 0; i < bufsize; ++i) {
            const float a = re * c, b = im * s, d = re * s, e = im * c;
            re = a - b;
            im = d + e;
            const float contrib = amp * re;
            o[i] += (double)contrib;
        }
    }
}

/* bench_modal.cu:152-179 */
void orc_modal(const float* params, float* out, int n_modes, int bufsize,
               int out_tracks) {
    memset(out, 0, sizeof(float) * (size_t)bufsize * out_tracks);
    int m = n_modes < out_tracks ? n_modes : out_tracks;
    const float cexp_real = expf(0.5f) * cosf(0.5f);
    for (int i = 0; i < m; ++i) {
        float v = params[(size_t)i * 8 + 0] * cexp_real;
        for (int s = 0; s < bufsize; ++s) out[(size_t)i * bufsize + s] = v;
    }
}

/* ======================================================================== */
/* digital waveguide                                                        */
/* ======================================================================== */

/* bench_dwg.cu:325-348 then :177-180 */
void orc_dwg_init(orc_wg_state* wg, float* input, int n_wg, int bufsize) {
    orc_rand_t st;
    orc_srand(&st, 42);
    for (int i = 0; i < n_wg; ++i) {
        wg[i].length = 100 + (orc_rand(&st) % (2000 - 100));
        wg[i].inputTapPos = wg[i].length / 4;
        wg[i].outputTapPos = 3 * wg[i].length / 4;
        wg[i].writePos = 0;
        wg[i].gain = 0.1f + 0.9f * ((float)orc_rand(&st) / (float)ORC_RAND_MAX);
        wg[i].reflection = 0.99f + 0.01f * ((float)orc_rand(&st) / (float)ORC_RAND_MAX - 0.5f);
        wg[i].damping = 0.9999f + 0.0001f * ((float)orc_rand(&st) / (float)ORC_RAND_MAX - 0.5f);
        wg[i].padding = 0.0f;
    }
    orc_rand_bipolar(&st, input, (size_t)bufsize);
}

/* bench_dwg.cu:356-399 (WAVEGUIDE_MIX_FACTOR = 0.5f) */
void orc_dwg(const orc_wg_state* wgs, float* fwd, float* bwd,
             const float* input, float* out, int n_wg, int bufsize,
             int max_len, int out_tracks) {
    memset(out, 0, sizeof(float) * (size_t)bufsize);
    for (int g = 0; g < n_wg; ++g) {
        const orc_wg_state wg = wgs[g];
        const size_t base = (size_t)g * max_len;
        for (int s = 0; s < bufsize; ++s) {
            float x = input[s] * wg.gain;
            int cur = (wg.writePos + s) % wg.length;
            int fp = cur;
            int bp = (cur + wg.length / 2) % wg.length;
            float f = fwd[base + fp];
            float b = bwd[base + bp];
            f *= wg.damping;
            b *= wg.damping;
            if (cur == wg.inputTapPos) { f += x; b += x; }
            float nf = b * wg.reflection;
            float nb = f * wg.reflection;
            fwd[base + fp] = nf;
            bwd[base + bp] = nb;
            if (cur == wg.outputTapPos && g < out_tracks)
                out[s] += (f + b) * 0.5f;
        }
    }
}

/* ======================================================================== */
/* FDTD3D                                                                   */
/* ======================================================================== */

/* bench_fdtd3d.cuh:12-41, bench_fdtd3d.cu:340-369.  Source/receiver scale
 * with the room (nx-2) so that the 52^3 reference grid gives the reference's
 * (25,25,5) / (40,15,25).                                                   */
orc_fdtd_params orc_fdtd_default_params(int nx, int ny, int nz) {
    orc_fdtd_params P;
    const float c = 343.0f, dx = 0.01f, rho = 1.225f;
    const float dt = 0.5f * dx / (c * 1.732050808f);
    P.nx = nx; P.ny = ny; P.nz = nz;
    int rx = nx - 2, ry = ny - 2, rz = nz - 2;
    P.src_x = rx / 2;      P.src_y = ry / 2;      P.src_z = rz / 10;
    P.rcv_x = rx * 4 / 5;  P.rcv_y = ry * 3 / 10; P.rcv_z = rz / 2;
    P.steps_per_sample = 3;
    P.dt_over_rho_dx = dt / (rho * dx);
    P.rho_c2_dt_over_dx = rho * c * c * dt / dx;
    P.absorption = 0.2f;
    return P;
}

/* bench_fdtd3d.cu:445-459 (FDTD3D_OUTPUT_SCALE 0.1f, FDTD3D_CPU_REF_FREQ 0.01f) */
void orc_fdtd_placeholder(const float* in, float* out, int tracks, int bufsize) {
    for (int t = 0; t < tracks; ++t)
        for (int s = 0; s < bufsize; ++s) {
            size_t i = (size_t)t * bufsize + s;
            out[i] = in[i] * 0.1f * cosf((float)s * 0.01f);
        }
}

/* Restates the four kernels (bench_fdtd3d.cu:14-139) in the order of
 * runFDTD3DTimeStep (:384-438).  The atomicAdd injection is fixed to "sum the
 * tracks in order, add the sum once".  `fused` selects the single-rounding a-=c*d nvcc emits for the
 * kernels (-fmad=true); fused==0 is the two-rounding host form.             */
static void fdtd_step(const orc_fdtd_params* P, float* p, float* vx, float* vy, float* vz, int fused) {
    const int nx = P->nx, ny = P->ny, nz = P->nz;
    const size_t sxy = (size_t)nx * ny;
    const float c1 = P->dt_over_rho_dx, c2 = P->rho_c2_dt_over_dx;
    const float damp = 1.0f - P->absorption;
    /* velocity (interior faces) */
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            const float* pr = p + z * sxy + (size_t)y * nx;
            float* vr = vx + ((size_t)z * ny + y) * (nx + 1);
            for (int x = 1; x < nx; ++x) {
                float d = pr[x] - pr[x - 1];
                vr[x] = fused ? fmaf(-c1, d, vr[x]) : vr[x] - c1 * d;
            }
        }
    for (int z = 0; z < nz; ++z)
        for (int y = 1; y < ny; ++y) {
            const float* pr = p + z * sxy + (size_t)y * nx;
            float* vr = vy + ((size_t)z * (ny + 1) + y) * nx;
            for (int x = 0; x < nx; ++x) {
                float d = pr[x] - pr[x - nx];
                vr[x] = fused ? fmaf(-c1, d, vr[x]) : vr[x] - c1 * d;
            }
        }
    for (int z = 1; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            const float* pr = p + z * sxy + (size_t)y * nx;
            float* vr = vz + z * sxy + (size_t)y * nx;
            for (int x = 0; x < nx; ++x) {
                float d = pr[x] - pr[x - (long)sxy];
                vr[x] = fused ? fmaf(-c1, d, vr[x]) : vr[x] - c1 * d;
            }
        }

    /* pressure */
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            float* pr = p + z * sxy + (size_t)y * nx;
            const int edge_zy = (z == 0 || z == nz - 1 || y == 0 || y == ny - 1);
            if (edge_zy) {
                for (int x = 0; x < nx; ++x) pr[x] *= damp;
                continue;
            }
            const float* ax = vx + ((size_t)z * ny + y) * (nx + 1);
            const float* ay = vy + ((size_t)z * (ny + 1) + y) * nx;
            const float* az = vz + z * sxy + (size_t)y * nx;
            pr[0] *= damp;
            for (int x = 1; x < nx - 1; ++x) {
                float div = (ax[x + 1] - ax[x]) + (ay[x + nx] - ay[x]) +
                            (az[x + sxy] - az[x]);
                pr[x] = fused ? fmaf(-c2, div, pr[x]) : pr[x] - c2 * div;
            }
            pr[nx - 1] *= damp;
        }
}

void orc_fdtd(const orc_fdtd_params* P, float* p, float* vx, float* vy,
              float* vz, const float* in, float* out, int tracks, int bufsize,
              int first_sample, int n_samples, int fused) {
    const size_t sxy = (size_t)P->nx * P->ny;
    const size_t src = (size_t)P->src_z * sxy + (size_t)P->src_y * P->nx + P->src_x;
    const size_t rcv = (size_t)P->rcv_z * sxy + (size_t)P->rcv_y * P->nx + P->rcv_x;

    for (int s = first_sample; s < first_sample + n_samples; ++s) {
        /* the reference's atomicAdd order is unspecified; fixed here as: scaled
         * samples summed in track order, the sum added to the cell once */
        float acc = 0.0f;
        for (int t = 0; t < tracks; ++t) acc += in[(size_t)t * bufsize + s] * 0.1f;
        p[src] += acc;
        for (int step = 0; step < P->steps_per_sample; ++step) fdtd_step(P, p, vx, vy, vz, fused);
        float o = p[rcv] * 0.1f;
        for (int t = 0; t < tracks; ++t) out[(size_t)t * bufsize + s] = o;
    }
}

/* The source cell as ONE ORDER OF THE REFERENCE'S atomicAdd leaves it (cuda/bench_fdtd3d.cu:101-120: every track adds
 * 0.1f*in[t,s] to p[src], order unspecified): the scaled samples accumulate INTO THE CELL one by one, tracks in `order`
 * (NULL: ascending) — p = (((p + a_0) + a_1) + ...), where orc_fdtd adds the tracks' own sum once, p + ((a_0 + a_1) + ...),
 * a grouping no atomic order produces (it can differ in the cell's last bit).  The bit-exact tests pin orc_fdtd's order
 * (the kernels' definition); this form bounds what that choice is worth against the reference.          */
void orc_fdtd_trackwise(const orc_fdtd_params* P, float* p, float* vx, float* vy,
                        float* vz, const float* in, float* out, int tracks, int bufsize,
                        int first_sample, int n_samples, int fused, const int* order) {
    const size_t sxy = (size_t)P->nx * P->ny;
    const size_t src = (size_t)P->src_z * sxy + (size_t)P->src_y * P->nx + P->src_x;
    const size_t rcv = (size_t)P->rcv_z * sxy + (size_t)P->rcv_y * P->nx + P->rcv_x;
    for (int s = first_sample; s < first_sample + n_samples; ++s) {
        for (int k = 0; k < tracks; ++k) {
            const int t = order ? order[k] : k;
            p[src] += in[(size_t)t * bufsize + s] * 0.1f;
        }
        for (int step = 0; step < P->steps_per_sample; ++step) fdtd_step(P, p, vx, vy, vz, fused);
        float o = p[rcv] * 0.1f;
        for (int t = 0; t < tracks; ++t) out[(size_t)t * bufsize + s] = o;
    }
}

/* Track-dependent source and receiver cells — what the Metal port's inject/extract kernels
 * announce ("can be made track-dependent later", kernels_fdtd3d.metal:184,217) and never do:
 * track t adds 0.1f*in[t,s] into ITS source cell (tracks in order, so cells shared by several
 * tracks receive their samples in track order) and reads 0.1f*p at ITS receiver cell.
 * src_xyz / rcv_xyz: tracks x (x, y, z).                                                      */
void orc_fdtd_tracks(const orc_fdtd_params* P, float* p, float* vx, float* vy, float* vz,
                     const float* in, float* out, int tracks, int bufsize, int first_sample,
                     int n_samples, int fused, const int* src_xyz, const int* rcv_xyz) {
    const size_t sxy = (size_t)P->nx * P->ny;
    for (int s = first_sample; s < first_sample + n_samples; ++s) {
        for (int t = 0; t < tracks; ++t) {
            const size_t c = (size_t)src_xyz[3 * t + 2] * sxy + (size_t)src_xyz[3 * t + 1] * P->nx + src_xyz[3 * t];
            p[c] += in[(size_t)t * bufsize + s] * 0.1f;
        }
        for (int step = 0; step < P->steps_per_sample; ++step) fdtd_step(P, p, vx, vy, vz, fused);
        for (int t = 0; t < tracks; ++t) {
            const size_t c = (size_t)rcv_xyz[3 * t + 2] * sxy + (size_t)rcv_xyz[3 * t + 1] * P->nx + rcv_xyz[3 * t];
            out[(size_t)t * bufsize + s] = p[c] * 0.1f;
        }
    }
}

/* ======================================================================== */
/* rndmem                                                                   */
/* ======================================================================== */

/* bench_rndmem.cu:140-149 */
void orc_rndmem_pool(float* pool, size_t n) {
    orc_rand_t st;
    orc_srand(&st, 42);
    for (size_t i = 0; i < n; ++i) pool[i] = (float)orc_rand(&st) / (float)ORC_RAND_MAX;
}

/* bench_rndmem.cu:151-174 — starts/ends are FLOATS in the reference        */
void orc_rndmem_playheads(int* playheads, float* starts, float* ends,
                          int tracks, int bufsize, size_t pool_elems,
                          int min_loop, int max_loop) {
    orc_rand_t st;
    orc_srand(&st, 42);
    const int end_ = (int)pool_elems - bufsize;
    for (int i = 0; i < tracks; ++i) {
        starts[i] = (float)(orc_rand(&st) % end_);
        int loop_len = min_loop + (orc_rand(&st) % (max_loop - min_loop));
        ends[i] = starts[i] + (float)loop_len;
        if (ends[i] >= (float)end_) ends[i] = (float)(end_ - 1);
        playheads[i] = (int)starts[i];
    }
}

/* bench_rndmem.cu:176-186 */
void orc_rndmem_advance(int* playheads, const float* starts, const float* ends,
                        int tracks, int bufsize) {
    for (int i = 0; i < tracks; ++i) {
        playheads[i] += bufsize;
        if (playheads[i] >= (int)ends[i]) playheads[i] = (int)starts[i];
    }
}

/* bench_rndmem.cu:194-205 */
void orc_rndmem(const float* pool, const int* playheads, float* out,
                int bufsize, int tracks) {
    for (int t = 0; t < tracks; ++t) {
        int ph = playheads[t];
        for (int i = 0; i < bufsize; ++i)
            out[(size_t)tracks * i + t] = pool[(size_t)ph + i];
    }
}

/* ======================================================================== */
/* harness statistics                                                       */
/* ======================================================================== */

static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* bench_utils.cu:358-414 */
orc_stats orc_statistics(const float* lat, size_t n) {
    orc_stats s;
    memset(&s, 0, sizeof s);
    if (n == 0) return s;
    s.count = n;
    float* sorted = (float*)malloc(sizeof(float) * n);
    memcpy(sorted, lat, sizeof(float) * n);
    qsort(sorted, n, sizeof(float), cmp_float);
    s.min_val = sorted[0];
    s.max_val = sorted[n - 1];
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum += lat[i];
    s.mean = sum / (float)n;
    size_t mid = n / 2;
    s.median = (n % 2 == 0) ? (sorted[mid - 1] + sorted[mid]) / 2.0f : sorted[mid];
    float var = 0.0f;
    for (size_t i = 0; i < n; ++i) { float d = lat[i] - s.mean; var += d * d; }
    var /= (float)(n - 1);
    s.std_dev = sqrtf(var);
    const float ps[2] = {95.0f, 99.0f};
    float res[2];
    for (int q = 0; q < 2; ++q) {
        float index = ps[q] / 100.0f * (float)(n - 1);
        size_t lo = (size_t)floorf(index), hi = (size_t)ceilf(index);
        if (lo == hi) res[q] = sorted[lo];
        else {
            float w = index - (float)lo;
            res[q] = sorted[lo] * (1.0f - w) + sorted[hi] * w;
        }
    }
    s.p95 = res[0]; s.p99 = res[1];
    free(sorted);
    return s;
}
