// gab_fft.hpp — workgroup-level complex FFT for gfx950, registers + LDS.
//
// Replaces the cuFFT calls of the reference (cuda/bench_fft.cu:63,105;
// cuda/bench_conv1d_accel.cu:137,144,211,276,289) with a Stockham autosort
// transform that lives inside the calling kernel, so padding, spectral
// multiply, normalisation and extraction fuse with it instead of costing a
// pass over HBM each.
//
// Shape: N = R * NT complex points, NT threads, every thread owns exactly one
// radix-R butterfly per pass (R values in registers).  Thread `tid` enters with
// v[r] = x[tid + r*NT] and leaves with v[r] = X[tid + r*NT] — the same layout,
// so a forward transform, a bin-wise product and an inverse transform chain
// without any re-shuffle.  Between passes the values cross LDS once
// (ds_write_b64 / ds_read_b64); indices are padded so the strided pass-0
// stores stay bank-conflict free (32 write banks, 16-lane groups).
//
// Numerics: twiddles come from a table computed in float64 and rounded once;
// complex products are 2 mul + 2 fma.  Contraction is otherwise off for the
// whole library (-ffp-contract=off), so results do not depend on the optimiser.
#pragma once

#include <hip/hip_runtime.h>

namespace gab {
namespace fft {

// A complex value is a 64-bit register pair, so that gfx950's packed fp32 ops
// (v_pk_add/mul/fma_f32, two lanes of arithmetic per instruction) apply directly.
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf conj(cf a) { return mk(a.x, -a.y); }

// Complex products as TWO packed instructions.  op_sel / op_sel_hi pick which half
// of each 64-bit operand feeds the low / high result, neg_lo / neg_hi negate it —
// so the (-b.y, b.x) operand needs no v_xor + v_mov, which is what hipcc emits
// for the plain C expression (4 instructions per product, and this kernel is
// VALU-bound).  The intermediate is early-clobber: it is written while a and b
// are still live.
// (a.x + i a.y)(b.x + i b.y)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   // (-a.y*b.y, a.y*b.x)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"                             // + (a.x*b.x, a.x*b.y)
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"                 // (a.y*b.y, a.y*b.x)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]"              // + (a.x*b.x, -a.x*b.y)
        : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
// acc + a*b
__device__ __forceinline__ cf cfma(cf a, cf b, cf acc) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
        : "=&v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
// acc + conj(a)*b
__device__ __forceinline__ cf cfma_cj(cf a, cf b, cf acc) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]\n\t"   // (+a.y*b.y, -a.y*b.x)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"                                      // (+a.x*b.x, +a.x*b.y)
        : "=&v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
// acc + conj(a*b)
__device__ __forceinline__ cf cfma_cjcj(cf a, cf b, cf acc) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"   // (-a.y*b.y, -a.y*b.x)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]"                                      // (+a.x*b.x, -a.x*b.y)
        : "=&v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
// a * a
__device__ __forceinline__ cf csqr(cf a) { return cmul(a, a); }

// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ cf rot90(cf a) {
    return INV ? mk(-a.y, a.x) : mk(a.y, -a.x);
}

// a -/+ i*b in ONE packed add: op_sel swaps b's halves, neg_* flips one of them.
// Written as rot90() + cadd() the compiler emits v_mov + v_xor + v_pk_add (a sixth
// of this library's VALU instructions before these existed).
// a - i*b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf addmi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + i*b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf addpi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + rot90<INV>(b)  and  a - rot90<INV>(b)
template <bool INV>
__device__ __forceinline__ cf add_rot(cf a, cf b) { return INV ? addpi(a, b) : addmi(a, b); }
template <bool INV>
__device__ __forceinline__ cf sub_rot(cf a, cf b) { return INV ? addmi(a, b) : addpi(a, b); }

template <bool INV>
__device__ __forceinline__ void bfly2(cf& a, cf& b) {
    cf t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// 4-point DFT, natural order in and out.
template <bool INV>
__device__ __forceinline__ void bfly4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = cadd(a0, a2), t1 = csub(a0, a2);
    cf t2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(t0, t2);
    a1 = add_rot<INV>(t1, d);
    a2 = csub(t0, t2);
    a3 = sub_rot<INV>(t1, d);
}
// The same with input 2 (ROT3: and input 3) still owed a factor -i (forward) / +i
// (inverse): the rotation rides on the first layer's adds.
template <bool INV, bool ROT3>
__device__ __forceinline__ void bfly4_rot(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = add_rot<INV>(a0, a2), t1 = sub_rot<INV>(a0, a2);
    cf t2 = ROT3 ? add_rot<INV>(a1, a3) : cadd(a1, a3);
    cf d = ROT3 ? sub_rot<INV>(a1, a3) : csub(a1, a3);
    a0 = cadd(t0, t2);
    a1 = add_rot<INV>(t1, d);
    a2 = csub(t0, t2);
    a3 = sub_rot<INV>(t1, d);
}

// multiply by W16^m (forward: exp(-2*pi*i*m/16); inverse: conjugate)
template <bool INV, int M>
__device__ __forceinline__ cf tw16(cf a) {
    constexpr float C1 = 0.92387953251128674f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f;   // sin(pi/8)
    constexpr float H = 0.70710678118654752f;    // sqrt(1/2)
    if constexpr (M == 0) return a;
    else if constexpr (M == 4) return rot90<INV>(a);
    else if constexpr (M == 2) {
        // (x+iy)(H -/+ iH): two packed instructions
        return INV ? cmulc(a, mk(H, -H)) : cmul(a, mk(H, -H));
    } else if constexpr (M == 6) {
        // (x+iy)(-H -/+ iH)
        return INV ? mk(-(a.x + a.y) * H, (a.x - a.y) * H) : mk((a.y - a.x) * H, -(a.x + a.y) * H);
    } else {
        constexpr float wr = (M == 1) ? C1 : (M == 3) ? S1 : /* M == 9 */ -C1;
        constexpr float wi = (M == 1) ? -S1 : (M == 3) ? -C1 : /* M == 9 */ S1;
        return INV ? cmulc(a, mk(wr, wi)) : cmul(a, mk(wr, wi));
    }
}

// In-register radix-R DFT.  Output k of the transform is left in v[out_slot(k)].
template <int R, bool INV>
struct Butterfly;

template <bool INV>
struct Butterfly<2, INV> {
    __device__ static __forceinline__ void run(cf (&v)[2]) { bfly2<INV>(v[0], v[1]); }
    __host__ __device__ static constexpr int out_slot(int k) { return k; }
};

template <bool INV>
struct Butterfly<4, INV> {
    __device__ static __forceinline__ void run(cf (&v)[4]) { bfly4<INV>(v[0], v[1], v[2], v[3]); }
    __host__ __device__ static constexpr int out_slot(int k) { return k; }
};

template <bool INV>
struct Butterfly<8, INV> {
    // n = 2*n1 + n2 (n1 < 4, n2 < 2); k = k1 + 4*k2
    __device__ static __forceinline__ void run(cf (&v)[8]) {
        bfly4<INV>(v[0], v[2], v[4], v[6]);
        bfly4<INV>(v[1], v[3], v[5], v[7]);
        // y[k1][n2] sits in v[2*k1 + n2]; twiddle W8^(n2*k1) = W16^(2*n2*k1)
        v[3] = tw16<INV, 2>(v[3]);
        v[5] = tw16<INV, 4>(v[5]);
        v[7] = tw16<INV, 6>(v[7]);
        bfly2<INV>(v[0], v[1]);
        bfly2<INV>(v[2], v[3]);
        bfly2<INV>(v[4], v[5]);
        bfly2<INV>(v[6], v[7]);
    }
    // X[k1 + 4*k2] is in v[2*k1 + k2]
    __host__ __device__ static constexpr int out_slot(int k) { return 2 * (k & 3) + (k >> 2); }
};

template <bool INV>
struct Butterfly<16, INV> {
    // n = 4*n1 + n2; k = k1 + 4*k2
    __device__ static __forceinline__ void run(cf (&v)[16]) {
        bfly4<INV>(v[0], v[4], v[8], v[12]);
        bfly4<INV>(v[1], v[5], v[9], v[13]);
        bfly4<INV>(v[2], v[6], v[10], v[14]);
        bfly4<INV>(v[3], v[7], v[11], v[15]);
        // y[k1][n2] sits in v[4*k1 + n2]; twiddle W16^(n2*k1)
        v[5] = tw16<INV, 1>(v[5]);
        v[6] = tw16<INV, 2>(v[6]);
        v[7] = tw16<INV, 3>(v[7]);
        // W^4 = -/+i and W^6 = -/+i * W^2: the quarter turns are folded into the next layer
        v[9] = tw16<INV, 2>(v[9]);
        v[11] = tw16<INV, 2>(v[11]);
        v[13] = tw16<INV, 3>(v[13]);
        v[14] = tw16<INV, 2>(v[14]);
        v[15] = tw16<INV, 9>(v[15]);
        bfly4<INV>(v[0], v[1], v[2], v[3]);
        bfly4<INV>(v[4], v[5], v[6], v[7]);
        bfly4_rot<INV, true>(v[8], v[9], v[10], v[11]);
        bfly4_rot<INV, false>(v[12], v[13], v[14], v[15]);
    }
    // X[k1 + 4*k2] is in v[4*k1 + k2]
    __host__ __device__ static constexpr int out_slot(int k) { return 4 * (k & 3) + (k >> 2); }

    // Only outputs 14 and 15 (k1 = 2, 3 with k2 = 3): what overlap-save keeps of the
    // last pass of a 4096-point inverse.  ~40 instructions instead of ~100.
    __device__ static __forceinline__ void run_last2(const cf (&v)[16], cf& x14, cf& x15) {
        cf y2[4], y3[4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            cf t0 = cadd(v[n2], v[8 + n2]), t1 = csub(v[n2], v[8 + n2]);
            cf t2 = cadd(v[4 + n2], v[12 + n2]), d = csub(v[4 + n2], v[12 + n2]);
            y2[n2] = csub(t0, t2);
            y3[n2] = sub_rot<INV>(t1, d);
        }
        // twiddles W^(2 n2) and W^(3 n2); the quarter turns of W^4, W^6 ride on the adds below
        y2[1] = tw16<INV, 2>(y2[1]); y2[3] = tw16<INV, 2>(y2[3]);            // y2[2], y2[3] owe a rot90
        y3[1] = tw16<INV, 3>(y3[1]); y3[2] = tw16<INV, 2>(y3[2]); y3[3] = tw16<INV, 9>(y3[3]);   // y3[2] owes one
        x14 = sub_rot<INV>(sub_rot<INV>(y2[0], y2[2]), sub_rot<INV>(y2[1], y2[3]));
        x15 = sub_rot<INV>(sub_rot<INV>(y3[0], y3[2]), csub(y3[1], y3[3]));
    }

    // Outputs 12..15 (every k1 with k2 = 3): the last QUARTER of the transform's output, for a
    // partition that keeps 1024 samples of a 4096-point inverse.  ~70 instructions.
    __device__ static __forceinline__ void run_last4(const cf (&v)[16], cf& x12, cf& x13, cf& x14, cf& x15) {
        cf y0[4], y1[4], y2[4], y3[4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            cf t0 = cadd(v[n2], v[8 + n2]), t1 = csub(v[n2], v[8 + n2]);
            cf t2 = cadd(v[4 + n2], v[12 + n2]), d = csub(v[4 + n2], v[12 + n2]);
            y0[n2] = cadd(t0, t2);
            y1[n2] = add_rot<INV>(t1, d);
            y2[n2] = csub(t0, t2);
            y3[n2] = sub_rot<INV>(t1, d);
        }
        y1[1] = tw16<INV, 1>(y1[1]); y1[2] = tw16<INV, 2>(y1[2]); y1[3] = tw16<INV, 3>(y1[3]);
        y2[1] = tw16<INV, 2>(y2[1]); y2[3] = tw16<INV, 2>(y2[3]);            // y2[2], y2[3] owe a rot90
        y3[1] = tw16<INV, 3>(y3[1]); y3[2] = tw16<INV, 2>(y3[2]); y3[3] = tw16<INV, 9>(y3[3]);   // y3[2] owes one
        x12 = sub_rot<INV>(csub(y0[0], y0[2]), csub(y0[1], y0[3]));
        x13 = sub_rot<INV>(csub(y1[0], y1[2]), csub(y1[1], y1[3]));
        x14 = sub_rot<INV>(sub_rot<INV>(y2[0], y2[2]), sub_rot<INV>(y2[1], y2[3]));
        x15 = sub_rot<INV>(sub_rot<INV>(y3[0], y3[2]), csub(y3[1], y3[3]));
    }
};

constexpr int kTwiddleN = 4096;   // table holds exp(-2*pi*i*m/4096), m < 4096

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
__host__ __device__ constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }
__host__ __device__ constexpr int passes_of(int n, int r) { return n <= 1 ? 0 : 1 + passes_of(n / r, r); }

// LDS image: logical index i lives at i + (i >> PADSH); one pad slot per
// 2^PADSH entries breaks the power-of-two stride of the pass-0 stores.
template <int R>
struct Pad {
    static constexpr int SH = ilog2(R);
    __host__ __device__ static constexpr int at(int i) { return i + (i >> SH); }
    __host__ __device__ static constexpr int size(int n) { return n + (n >> SH); }
};


// Twiddles of one transform shape: for every pass p >= 1 the R-1 factors
// W_{Ns*R}^(r*k), r = 1..R-1, k = tid mod Ns.  They depend on tid only, so a
// kernel forms them once — one table read per pass (w^1) plus in-register
// powers (products at most four deep, ~3e-7 relative) — ideally while it waits
// for its first data, and reuses them for the forward AND the inverse transform
// (the inverse conjugates).  A per-lane gather of all R-1 powers from a table
// costs R-1 scattered L1 transactions per pass and was this kernel's first
// bottleneck.
template <int R, int PASSES>
struct TwiddleSet {
    cf w[PASSES > 1 ? PASSES - 1 : 1][R - 1];
};

// Cheaper in registers: only the base w^1 of each pass; powers are re-formed
// inside every pass (R-2 complex products of VALU work per pass).
template <int PASSES>
struct TwiddleBases {
    cf w[PASSES > 1 ? PASSES - 1 : 1];
};

// In between: all powers of the LAST pass, only the base of the earlier ones (which are
// re-formed inside each pass).  Saves (PASSES-2)*(R-2) complex registers for R-2
// products per earlier pass and direction.
template <int R, int PASSES>
struct TwiddleMixed {
    cf base[PASSES > 2 ? PASSES - 2 : 1];
    cf last[R - 1];
};

template <int R>
__device__ __forceinline__ void powers_of(cf w, cf (&out)[R - 1]) {
    out[0] = w;
    if constexpr (R >= 4) {
        out[1] = csqr(w);
        out[2] = cmul(out[1], w);
    }
    if constexpr (R >= 8) {
        out[3] = csqr(out[1]);
        out[4] = cmul(out[3], w);
        out[5] = cmul(out[3], out[1]);
        out[6] = cmul(out[3], out[2]);
    }
    if constexpr (R >= 16) {
        out[7] = csqr(out[3]);
#pragma unroll
        for (int i = 0; i < 7; ++i) out[8 + i] = cmul(out[7], out[i]);
    }
}

// One workgroup-wide transform.  `ldsA`/`ldsB` each hold Pad<R>::size(N) cf.
// All NT threads must call it (it contains barriers).  On return the last pass
// has read from `ldsB`; a caller chains stages so that each stage first writes
// the buffer whose last readers are already behind a barrier.
template <int N, int R, bool INV>
struct BlockFFT {
    static constexpr int NT = N / R;
    static constexpr int PASSES = passes_of(N, R);
    static_assert(ipow(R, PASSES) == N, "N must be a power of R");
    static_assert(kTwiddleN % N == 0, "twiddle table too small");
    using P = Pad<R>;
    using Twiddles = TwiddleSet<R, PASSES>;
    using Bases = TwiddleBases<PASSES>;

    __device__ static __forceinline__ void load_twiddles(Bases& t, const cf* __restrict__ tw, int tid) {
#pragma unroll
        for (int p = 1; p < PASSES; ++p) {
            const int Ns = ipow(R, p);
            t.w[p - 1] = tw[(tid & (Ns - 1)) * (kTwiddleN / (Ns * R))];
        }
    }

    __device__ static __forceinline__ void pass_twiddles(const Twiddles& t, int p, cf (&w)[R - 1]) {
#pragma unroll
        for (int r = 0; r < R - 1; ++r) w[r] = t.w[p - 1][r];
    }
    __device__ static __forceinline__ void pass_twiddles(const Bases& t, int p, cf (&w)[R - 1]) {
        cf b = t.w[p - 1];
        // pin the power computation to this pass: hoisted to kernel entry it would
        // hold (PASSES-1)*(R-1) complex registers live instead of PASSES-1
        asm volatile("" : "+v"(b.x), "+v"(b.y));
        powers_of<R>(b, w);
    }

    using Mixed = TwiddleMixed<R, PASSES>;
    __device__ static __forceinline__ void pass_twiddles(const Mixed& t, int p, cf (&w)[R - 1]) {
        if (p == PASSES - 1) {
#pragma unroll
            for (int r = 0; r < R - 1; ++r) w[r] = t.last[r];
        } else {
            cf b = t.base[p - 1];
            asm volatile("" : "+v"(b.x), "+v"(b.y));
            powers_of<R>(b, w);
        }
    }
    __device__ static __forceinline__ void load_twiddles(Mixed& t, const cf* __restrict__ tw, int tid) {
        cf lastb;
#pragma unroll
        for (int p = 1; p < PASSES; ++p) {
            const int Ns = ipow(R, p);
            cf b = tw[(tid & (Ns - 1)) * (kTwiddleN / (Ns * R))];
            if (p == PASSES - 1) lastb = b; else t.base[p - 1] = b;
        }
        powers_of<R>(lastb, t.last);
    }

    __device__ static __forceinline__ void load_twiddles(Twiddles& t, const cf* __restrict__ tw, int tid) {
        cf base[PASSES];
#pragma unroll
        for (int p = 1; p < PASSES; ++p) {
            const int Ns = ipow(R, p);
            base[p] = tw[(tid & (Ns - 1)) * (kTwiddleN / (Ns * R))];
        }
#pragma unroll
        for (int p = 1; p < PASSES; ++p) powers_of<R>(base[p], t.w[p - 1]);
    }

    // bases -> all powers, for callers that fetch the bases early but cannot afford the
    // (PASSES-1)*(R-1) registers until later
    __device__ static __forceinline__ void expand_twiddles(const Bases& b, Twiddles& t) {
#pragma unroll
        for (int p = 1; p < PASSES; ++p) powers_of<R>(b.w[p - 1], t.w[p - 1]);
    }

    // `active` lets a workgroup wider than NT threads run the transform on its first
    // NT threads: the others skip the arithmetic but still meet every barrier.
    // KEEP = 2 / 4 (radix 16 only): the caller needs just the last 2 / 4 of the 16 outputs,
    // X[tid + r*NT] for r >= 14 / r >= 12; they are returned in v[r], every other v[] is then
    // unspecified.  KEEP = 0: everything.
    // `hook(p)` runs right after the barrier that closes pass p (p < PASSES-1): a place
    // for the caller to slip independent work (e.g. a couple of global loads) into the
    // transform's instruction stream.
    struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

    template <class TW, int KEEP = 0, class Hook = NoHook>
    __device__ static __forceinline__ void run(cf (&v)[R], cf* __restrict__ ldsA,
                                               cf* __restrict__ ldsB, const TW& tws, int tid,
                                               bool active = true, Hook hook = Hook()) {
        // Opaque copy: stops the compiler from sharing LDS address arithmetic between
        // separate transforms of one kernel, which it otherwise keeps live (and spills)
        // across everything in between.
        unsigned t = (unsigned)tid;
        asm volatile("" : "+v"(t));
        constexpr int SH = P::SH;
        static_assert(NT % R == 0, "pad arithmetic assumes R | NT");
        // Every LDS access below is ONE base register plus a compile-time offset:
        // Pad(i) = i + (i >> SH) distributes over the strides used here because they
        // are multiples of R = 2^SH (or, in pass 0, the R values of a thread are
        // contiguous).  Recomputing Pad() per element costs ~3 integer VALU ops per
        // access, ~100 per radix-16 pass, in a kernel that is VALU-bound.
#ifdef GAB_FFT_NOPAD_READS                                      // EXPERIMENT builds only (wrong results): what the linear reads' two-way bank
        const unsigned rd_base = t;                             // conflicts cost — the same reads without the pad term are conflict-free
#else
        const unsigned rd_base = t + (t >> SH);                 // Pad(t + r*NT) = rd_base + r*RD
#endif
        constexpr unsigned RD = NT + (NT >> SH);
        cf* buf = ldsA;
        cf* other = ldsB;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int Ns = ipow(R, p);
            if (p > 0) {
                if (active) {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = buf[rd_base + r * RD];
                    cf w[R - 1];
                    pass_twiddles(tws, p, w);
#pragma unroll
                    for (int r = 1; r < R; ++r) v[r] = INV ? cmulc(v[r], w[r - 1]) : cmul(v[r], w[r - 1]);
                }
                cf* tmp = buf; buf = other; other = tmp;
            }
            if constexpr (KEEP != 0 && R == 16) {
                static_assert(KEEP == 2 || KEEP == 4, "KEEP is 0, 2 or 4");
                if (p == PASSES - 1) {
                    if (active) {
                        cf x12, x13, x14, x15;
                        if constexpr (KEEP == 2) {
                            Butterfly<R, INV>::run_last2(v, x14, x15);
                        } else {
                            Butterfly<R, INV>::run_last4(v, x12, x13, x14, x15);
                            v[12] = x12;
                            v[13] = x13;
                        }
                        v[14] = x14;
                        v[15] = x15;
                    }
                    break;
                }
            }
            if (active) Butterfly<R, INV>::run(v);
            if (p < PASSES - 1) {
                if (active) {
                    if (Ns >= R) {
                        // base = (t / Ns) * Ns * R + (t mod Ns); Pad(base + r*Ns) = wb + r*WS
                        const unsigned base = (t / (unsigned)Ns) * (unsigned)(Ns * R) + (t & (unsigned)(Ns - 1));
                        const unsigned wb = base + (base >> SH);
                        const unsigned WS = (unsigned)Ns + ((unsigned)Ns >> SH);
#pragma unroll
                        for (int r = 0; r < R; ++r) buf[wb + r * WS] = v[Butterfly<R, INV>::out_slot(r)];
                    } else {
                        // Ns == 1: indices t*R + r, r < R  ->  Pad = t*(R+1) + r
                        const unsigned wb = t * (unsigned)(R + 1);
#pragma unroll
                        for (int r = 0; r < R; ++r) buf[wb + r] = v[Butterfly<R, INV>::out_slot(r)];
                    }
                }
                __syncthreads();
                hook(p);
            } else if (active) {
                // leave X[tid + r*NT] in v[r]
                cf o[R];
#pragma unroll
                for (int r = 0; r < R; ++r) o[r] = v[Butterfly<R, INV>::out_slot(r)];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = o[r];
            }
        }
    }
};

// A 1024-point transform held by ONE wave: 64 lanes x 16 values, lane j enters with
// v[r] = x[j + 64 r] and leaves with v[r] = X[j + 64 r].  Passes: four radix-4
// butterflies per lane (Ns = 1), radix-16 (Ns = 4), radix-16 (Ns = 64).  The two
// exchanges go through a wave-private LDS image of Pad<16>::size(1024) entries and
// need no workgroup barrier: a wave's LDS instructions execute in order.
struct Wave1024Twiddles {
    cf w1;         // W64^(j&3): its powers are re-formed in the pass (registers)
    cf w2[15];     // W1024^(j i), i = 1..15
};
// Two registers instead of sixteen: both passes re-form their powers from the base (the same
// powers_of, so the same values) — for kernels whose waves carry other state across transforms.
struct Wave1024TwiddlesLean {
    cf w1;         // W64^(j&3)
    cf w2b;        // W1024^j
};

template <bool INV>
struct WaveFFT1024 {
    static constexpr int N = 1024;
    using P = Pad<16>;
    using Twiddles = Wave1024Twiddles;

    __device__ static __forceinline__ void load_twiddles(Twiddles& t, const cf* __restrict__ tw, int lane) {
        t.w1 = tw[(lane & 3) * (kTwiddleN / 64)];
        powers_of<16>(tw[lane * (kTwiddleN / 1024)], t.w2);
    }
    // The same in two steps, for callers that want the two table reads at the head of their
    // request stream and the power expansion later: raw leaves W1024^lane in w2[0].
    __device__ static __forceinline__ void load_twiddles_raw(Twiddles& t, const cf* __restrict__ tw, int lane) {
        t.w1 = tw[(lane & 3) * (kTwiddleN / 64)];
        t.w2[0] = tw[lane * (kTwiddleN / 1024)];
    }
    __device__ static __forceinline__ void expand_twiddles(Twiddles& t) {
        const cf b = t.w2[0];
        powers_of<16>(b, t.w2);
    }
    using Lean = Wave1024TwiddlesLean;
    __device__ static __forceinline__ void load_twiddles(Lean& t, const cf* __restrict__ tw, int lane) {
        t.w1 = tw[(lane & 3) * (kTwiddleN / 64)];
        t.w2b = tw[lane * (kTwiddleN / 1024)];
    }
    __device__ static __forceinline__ void last_pass_twiddles(const Twiddles& t, cf (&w)[15]) {
#pragma unroll
        for (int r = 0; r < 15; ++r) w[r] = t.w2[r];
    }
    __device__ static __forceinline__ void last_pass_twiddles(const Lean& t, cf (&w)[15]) {
        cf b = t.w2b;
        asm volatile("" : "+v"(b.x), "+v"(b.y));          // pinned to the pass, like w1's
        powers_of<16>(b, w);
    }

    struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

    // `hook(i)` runs at the transform's two exchange points (i = 0, 1), after the wave's LDS writes and
    // before its reads: a caller whose workgroup shares ONE hardware barrier between roles arrives at
    // that barrier there, so that a transform spans three barrier intervals instead of one.
    template <class Hook = NoHook, class TW = Twiddles>
    __device__ static __forceinline__ void run(cf (&v)[16], cf* __restrict__ img, const TW& t,
                                               int lane_in, Hook hook = Hook()) {
        unsigned lane = (unsigned)lane_in;
        asm volatile("" : "+v"(lane));
        // pass 0: butterflies b = lane + 64 m on v[m + 4 i]; y[4 b + i]
#pragma unroll
        for (int m = 0; m < 4; ++m) bfly4<INV>(v[m], v[m + 4], v[m + 8], v[m + 12]);
        {
            // Pad(4 lane + 256 m + i) = 4 lane + (lane >> 2) + 272 m + i
            const unsigned wb = 4u * lane + (lane >> 2);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) img[wb + 272 * m + i] = v[m + 4 * i];
        }
        __builtin_amdgcn_wave_barrier();
        hook(0);
#ifdef GAB_FFT_NOPAD_READS
        const unsigned rb = lane;
#else
        const unsigned rb = lane + (lane >> 4);         // Pad(lane + 64 r) = rb + 68 r
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = img[rb + 68 * r];
        {
            cf b = t.w1, w1[15];
            asm volatile("" : "+v"(b.x), "+v"(b.y));
            powers_of<16>(b, w1);
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = INV ? cmulc(v[r], w1[r - 1]) : cmul(v[r], w1[r - 1]);
        }
        Butterfly<16, INV>::run(v);
        __builtin_amdgcn_wave_barrier();
        {
            // y[(lane/4) 64 + (lane&3) + 4 k]; Pad = (lane>>2) 68 + (lane&3) + 4 k + (k >> 2)
            const unsigned wb = (lane >> 2) * 68u + (lane & 3u);
#pragma unroll
            for (int k = 0; k < 16; ++k) img[wb + 4 * k + (k >> 2)] = v[Butterfly<16, INV>::out_slot(k)];
        }
        __builtin_amdgcn_wave_barrier();
        hook(1);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = img[rb + 68 * r];
        {
            cf w2[15];
            last_pass_twiddles(t, w2);
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = INV ? cmulc(v[r], w2[r - 1]) : cmul(v[r], w2[r - 1]);
        }
        Butterfly<16, INV>::run(v);
        __builtin_amdgcn_wave_barrier();
        cf o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = v[Butterfly<16, INV>::out_slot(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = o[k];
    }
};

// Host: fill the float64-accurate twiddle table (kTwiddleN complex floats).
void build_twiddles(float* table_xy);

}  // namespace fft
}  // namespace gab
