// k_conv_accel_diag.hpp — forms of the split cut that were measured against the product's and not kept: DIAGNOSTIC builds only
// (-DGAB_ABLATE -> libgab_hip_ablate.so, chosen with GAB_LIB_PATH), included by k_conv_accel.hip inside namespace gab { namespace {.
// The product library carries none of it.
//   * round 5's eight-wave workgroup (two waves per SIMD, 244-249 registers): conv_split_batch_kernel (GAB_BATCH_WAVES=8) and the
//     doorbell-fed conv_split_engine_kernel (GAB_ENGINE_WAVES=8) — the baselines of tools/batch_waves_ab.sh, tools/engine_waves_ab.sh;
//   * round 6's six-wave forms (one pair per workgroup; counter barriers): GAB_BATCH_WAVES=6 / 64 / 26 — profiles/r06_batch_forms.txt.
// Same bits as the product's launches, all of them (the engine and batch tests run under GAB_*_WAVES in the diagnostic build).
#pragma once

constexpr int kBatchThreads = 2 * kThreads;                                            // round 5's workgroup: eight waves
constexpr int kBatchLds = 6 * kWaveImg + 2 * kLdsHalf + 2 * kCarrySlots * kB;      // cf entries (151 KB)

// ---- n buffers per launch (gab_conv_process_batch; bench.py's `value`) --------------------------------------------------
// Round 5: the batch launch has this function to itself again.  Round 4 ran batch launch and engine from one template;
// when the engine's period loop was wrapped in a loop over bursts (one buffer in flight, below), the SAME period code
// compiled to a different schedule for the batch instantiation too and ran 1.0-1.4 % slower (5.10 -> 5.17 us per buffer,
// same box, three alternations: profiles/r05_batch_ab.txt).  The text below is round 4's with the engine's branches taken
// out: the compiler's output for conv_split_batch_kernel is instruction for instruction what it was.
__device__ __forceinline__ void conv_split_batch_resident(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head0, int n_buffers,
    cf* __restrict__ lds) {
    constexpr bool ENGINE = false;                                    // (for GAB_BSTAMP's period choice in diagnostic builds)
    (void)ENGINE;
    cf* const far_x = lds + 6 * kWaveImg;
    cf* const far_y = far_x + kLdsHalf;
    cf* const carry = far_y + kLdsHalf;                               // [pair of the duo][slot][512]
    const int tid = threadIdx.x;
    const int d = xcd_contiguous(blockIdx.x, gridDim.x);
    const size_t step = (size_t)T * kB;
    cf* const carry_g = sp.carry + (size_t)(2 * d) * kCarrySlots * kB;   // the duo's two rings are contiguous
    for (int i = tid; i < 2 * kCarrySlots * kB; i += kBatchThreads) carry[i] = carry_g[i];
    __syncthreads();
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GAB_FFT_NOPAD_READS                                         // EXPERIMENT builds only (wrong results): gab_fft.hpp
    const unsigned rb = (unsigned)lane;
#else
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
#endif
    auto gate = [&](int) -> int { return n_buffers; };    // (the engine's doorbell gate stands here in conv_split_engine_resident)
    // buffer nb of the launch; callers walk the buffers with next_slot()
    auto in_slot = [&](int slot) -> const float* { return in + (size_t)slot * step; };
    auto next_slot = [&](int slot) -> int { return slot + 1; };
    auto ld = [](const float* p) -> float { return *p; };

    if (w >= 4) {
        // ---- far waves: F of the pair whose turn it is (window = blocks k-7 .. k; the two newest straight from
        // the input buffers, the next buffer's operands requested under the inverse transform); the share is
        // parked in LDS
        const int ft = tid - kThreads;
        using FB = fft::BlockFFT<kNB, 16, false>;
        using FBi = fft::BlockFFT<kNB, 16, true>;
        typename FB::Twiddles twb;                                  // every pass's powers stay in registers
        FB::load_twiddles(twb, tw, ft);
        // (Measured, not kept: the next buffer's 36 requests spread over three barrier intervals instead of
        // one burst after the spectral product — the burst's 0.6 us on the chain only moves: 5.75 vs 5.31 us.)
        // Blocks that lie inside the launch come from the input buffers (block k-j = buffer nb-j), older ones
        // from the history ring as the previous launch left it: the ring is neither read nor written in the
        // steady state of a launch (the forward waves refresh it over the launch's last eight buffers).
        // FIRST (a compile-time tag): the launch's first window, whose block k-1 is still the history ring's.  (As a
        // run-time test on nb the engine's k-1 loads became conditional loads: a register merge behind them, i.e. a
        // wait for the whole request burst in the middle of the far chain — 2.3 instead of 1.4 us for that interval.)
        auto load_window = [&](auto first_tag, int nb, int slot, int slot_before, cf (&z)[16], float4 (&c)[16]) {
            [[maybe_unused]] constexpr bool FIRST = decltype(first_tag)::value;
            const int head = (head0 + nb) & (kSlots - 1);
            const int q = 2 * d + (head & 1);
            const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
            const size_t ca = (size_t)(2 * q) * kB, cb_ = ca + kB;
            const float* const cur = in_slot(slot);
            z[14] = mk(ld(cur + ca + ft), ld(cur + cb_ + ft));
            z[15] = mk(ld(cur + ca + ft + kThreads), ld(cur + cb_ + ft + kThreads));
            if (nb >= kSlots - 1) {                                   // the whole window lies inside the launch
#pragma unroll
                for (int bl = 0; bl < 7; ++bl) {
                    const int sb = slot - (7 - bl);
                    const float* const src = in_slot(sb);
                    z[2 * bl] = mk(src[ca + ft], src[cb_ + ft]);
                    z[2 * bl + 1] = mk(src[ca + ft + kThreads], src[cb_ + ft + kThreads]);
                }
            } else {
#pragma unroll
                for (int bl = 0; bl < 7; ++bl) {                      // block k-7+bl = buffer nb-7+bl (uniform branch)
                    if (nb - 7 + bl >= 0) {
                        const float* const src = in_slot(slot - (7 - bl));       // nb < 7: no wrap yet
                        z[2 * bl] = mk(src[ca + ft], src[cb_ + ft]);
                        z[2 * bl + 1] = mk(src[ca + ft + kThreads], src[cb_ + ft + kThreads]);
                    } else {
                        const int s = ((head + 1 + bl) & (kSlots - 1)) * kB;
                        z[2 * bl] = hp[s + ft];
                        z[2 * bl + 1] = hp[s + kThreads + ft];
                    }
                }
            }
            load_spectra<kNB, 16>(c, sp.pmF + (size_t)q * kBinsB, ft);
        };
        cf zb[16], zn[16];
        float4 cb[16];
        int avail = gate(0);
        if (avail > 0) load_window(std::true_type{}, 0, 0, 0, zb, cb);
        int slot = 0;                                                 // of buffer nb
        for (int nb = 0;; ++nb, slot = next_slot(slot)) {
            if (nb > 0) avail = gate(nb);
            if (nb >= avail) break;
#ifdef GAB_ABLATE
            if (GAB_SDBG(4)) {                                        // diagnostic builds: far role idle
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                continue;
            }
#endif
            const int head = (head0 + nb) & (kSlots - 1);
            cf* const cp = carry + (head & 1) * kCarrySlots * kB;
            GAB_BSTAMP(6);
#ifdef GAB_ABLATE
            FB::run(zb, far_x, far_y, twb, ft, true, [&](int p) { GAB_BSTAMP(p); });
#else
            FB::run(zb, far_x, far_y, twb, ft);                       // barriers 1, 2
#endif
            partner_exchange<kNB, 16, true>(zb, zn, far_x, ft);       // barrier 3
            GAB_BSTAMP(2);
            spectral_product<kNB, 16>(zb, zn, cb, ft);
            __builtin_amdgcn_sched_barrier(0);
#ifdef GAB_ABLATE
            if (!GAB_SDBG(512)) { keep_alive(zb[0]); keep_alive(zb[15]); GAB_BSTAMP(7); }      // slot 7: the product is done
#endif
            if (nb + 1 < avail) load_window(std::false_type{}, nb + 1, next_slot(slot), slot, zn, cb);   // flies under the inverse transform
            __builtin_amdgcn_sched_barrier(0);
#ifdef GAB_ABLATE
            if (GAB_SDBG(512)) GAB_BSTAMP(7);                                                   // or: the request burst has been issued
#endif
#ifdef GAB_ABLATE
            FBi::template run<typename FB::Twiddles, 4>(zb, far_y, far_x, twb, ft, true, [&](int p) { GAB_BSTAMP(3 + p); });
#else
            FBi::template run<typename FB::Twiddles, 4>(zb, far_y, far_x, twb, ft);   // barriers 4, 5; only [12..15]
#endif
            cf* const c1 = cp + ((head + 1) & (kCarrySlots - 1)) * kB;              // block k+1
            cf* const c2 = cp + ((head + 2) & (kCarrySlots - 1)) * kB;              // block k+2
            c1[ft] = zb[12];
            c1[ft + kThreads] = zb[13];
            c2[ft] = zb[14];
            c2[ft + kThreads] = zb[15];
            __syncthreads();                                          // barrier 6 closes the period
            GAB_BSTAMP(5);
#pragma unroll
            for (int r = 0; r < 16; ++r) zb[r] = zn[r];
        }
        for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();    // the pipeline's last period
    } else if (w < 2) {
        // ---- forward waves: wave w holds pair w of the duo
        const int q = 2 * d + w;
        cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const img = lds + w * kWaveImg;                           // the transform's exchanges, then its spectrum
        cf* const hand = lds + (2 + w) * kWaveImg;                    // output spectrum for the inverse wave
        using WF = fft::WaveFFT1024<false>;
        WF::Lean t;
        WF::load_twiddles(t, tw, lane);
        const float4* const pa = pmA + (size_t)q * kBinsA;
        const float4* const pa2 = sp.pmA2 + (size_t)q * kBinsA;
        const size_t xoff = (size_t)(2 * q) * kB;                     // channel a of a buffer; channel b is kB further
        cf z[16], prev[8], nxt[8];
        float4 c[16];
        int avail = gate(0);
        {   // prologue: the spectrum of the ring's blocks [k-2 | k-1] into the image
            const int s1 = ((head0 + kSlots - 1) & (kSlots - 1)) * kB, s2 = ((head0 + kSlots - 2) & (kSlots - 1)) * kB;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = hp[s1 + lane + 64 * j];
            if (avail > 0) {
                const float* const x0 = in_slot(0) + xoff;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(x0 + lane + 64 * j), ld(x0 + kB + lane + 64 * j));
            }
            load_spectra<kNA, 16>(c, pa2, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[8 + j] = prev[j];
            WF::run(z, img, t, lane, WF::NoHook());
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
            __builtin_amdgcn_wave_barrier();
        }
        int slot = 0;                                                 // of buffer nb
        for (int nb = 0;; ++nb, slot = next_slot(slot)) {
            if (nb > 0) avail = gate(nb);
            if (nb >= avail) break;
#ifdef GAB_ABLATE
            if (GAB_SDBG(1)) {                                        // diagnostic builds: near role idle
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                continue;
            }
#endif
            // One piece of work per barrier interval (the far role's transform has six):
            //   A2 share | window + pass 0 | pass 1 | pass 2 | spectrum + A product | hand-over + requests
            const int head = (head0 + nb) & (kSlots - 1);
            cf share[16];                                             // taps [512,1024): last period's spectrum x pmA2
            {
                cf vp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) share[r] = img[rb + 68 * r];
#pragma unroll
                for (int r = 0; r < 16; ++r) vp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                spectral_product<kNA, 16>(share, vp, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // an opaque copy of the lane index: sixteen loop-invariant 64-bit addresses would otherwise be
                // hoisted out of the loop, spilled, and reloaded one by one between the loads they feed
                int lo = lane;
                asm volatile("" : "+v"(lo));
                load_spectra<kNA, 16>(c, pa, lo);                     // for this buffer's A product (interval 5)
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_BSTAMP(0);
            __syncthreads();                                          // barrier 1
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[j] = prev[j]; z[8 + j] = nxt[j]; }
            if (nb + kSlots >= n_buffers) {                              // the ring only has to hold the launch's LAST eight blocks
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = nxt[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = nxt[j];
            if (nb + 1 < avail) {                                     // the next buffer's block: needed a period from now
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(next_slot(slot)) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(xa + 64 * j), ld(xa + kB + 64 * j));
            }
#ifdef GAB_ABLATE
            WF::run(z, img, t, lane, [&](int i) { GAB_BSTAMP(1 + i); __syncthreads(); });
#else
            WF::run(z, img, t, lane, ArriveAtBarrier());              // barriers 2, 3 from inside
#endif
            GAB_BSTAMP(3);
            __syncthreads();                                          // barrier 4
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];     // the spectrum stays here for the next period
            __builtin_amdgcn_wave_barrier();
            {
                cf zp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                spectral_product<kNA, 16>(z, zp, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                load_spectra<kNA, 16>(c, pa2, lo);                    // for the next period's A2 share
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_BSTAMP(4);
            __syncthreads();                                          // barrier 5
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = fft::cadd(z[r], share[r]);   // A product + A2 share
            GAB_BSTAMP(5);
            __syncthreads();                                          // barrier 6 closes the period
            GAB_BSTAMP(6);
        }
        for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();    // the pipeline's last period
    } else {
        // ---- inverse waves: wave 2 + p turns the output spectrum of pair p into samples, one period later
        const int pr = w - 2;
        const cf* const hand = lds + (2 + pr) * kWaveImg;
        cf* const img = lds + (4 + pr) * kWaveImg;                    // the transform's exchanges, then the output swap
        const cf* const other = lds + (4 + (1 - pr)) * kWaveImg;
        const cf* const cring = carry + pr * kCarrySlots * kB;
        using WFi = fft::WaveFFT1024<true>;
        WFi::Lean t;
        WFi::load_twiddles(t, tw, lane);
        for (int nb = 0;; ++nb) {
            const int avail = gate(nb);
            const bool more = nb < avail;                             // the other roles work on buffer nb in this period
            if (nb == 0) {                                            // first period: nothing to turn yet
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                if (!more) break;
                continue;
            }
#ifdef GAB_ABLATE
            if (GAB_SDBG(1)) {                                        // diagnostic builds: near role idle
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                if (!more) break;
                continue;
            }
#endif
            // One piece per barrier interval: hand-over read | pass 0 | pass 1 | pass 2 + far share | swap | stores
            const int b = nb - 1;                                     // the buffer whose spectrum was handed over last period
            const int head = (head0 + b) & (kSlots - 1);
            float* const outb = out + (size_t)b * step;
            cf z[16], y[8], park[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = hand[rb + 68 * r];    // the forward wave writes the next one in interval 6
#pragma unroll
            for (int j = 0; j < 8; ++j) park[j] = cring[(head & (kCarrySlots - 1)) * kB + lane + 64 * j];
            GAB_BSTAMP(0);
            __syncthreads();                                          // barrier 1
#ifdef GAB_ABLATE
            WFi::run(z, img, t, lane, [&](int i) { GAB_BSTAMP(1 + i); __syncthreads(); });
#else
            WFi::run(z, img, t, lane, ArriveAtBarrier());             // barriers 2, 3 from inside
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
            GAB_BSTAMP(3);
            __syncthreads();                                          // barrier 4
            // the two pairs of a duo are four neighbouring channels: the waves swap halves through LDS
            // so that each stores float4 pieces (pair 0 keeps samples lane + 64 j, j < 4, pair 1 j >= 4)
            if (pr == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[4 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[j];
            }
            GAB_BSTAMP(4);
            __syncthreads();                                          // barrier 5: the swapped halves are in LDS
            {
                float* const o0 = outb + 4 * (size_t)d;
                auto put = [&](float* dst, float a, float b2, float c2, float d2) {
                    *reinterpret_cast<float4*>(dst) = make_float4(a, b2, c2, d2);
                };
                if (pr == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const cf theirs = other[lane + 64 * j];
                        put(o0 + (size_t)T * (lane + 64 * j), y[j].x, y[j].y, theirs.x, theirs.y);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const cf theirs = other[lane + 64 * j];
                        put(o0 + (size_t)T * (lane + 64 * (4 + j)), theirs.x, theirs.y, y[4 + j].x, y[4 + j].y);
                    }
                }
            }
            GAB_BSTAMP(5);
            __syncthreads();                                          // barrier 6 closes the period
            GAB_BSTAMP(6);
            if (!more) break;
        }
    }
    // every wave is past the last closing barrier: the duo's carry ring goes back to memory
    for (int i = tid; i < 2 * kCarrySlots * kB; i += kBatchThreads) carry_g[i] = carry[i];
}


__device__ __forceinline__ void conv_split_engine_resident(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head0,
    const ConvEngine& eng, cf* __restrict__ lds, unsigned* __restrict__ s_door) {
    constexpr bool ENGINE = true;                                     // (GAB_BSTAMP's period choice; the GAB_ENGV experiments' conditions)
    constexpr int n_buffers = 0;                                      // (GAB_ENGV bit 8 only: a batch launch's history rule)
    (void)ENGINE; (void)n_buffers;
    cf* const far_x = lds + 6 * kWaveImg;
    cf* const far_y = far_x + kLdsHalf;
    cf* const carry = far_y + kLdsHalf;                               // [pair of the duo][slot][512]
    const int tid = threadIdx.x;
    const int d = xcd_contiguous(blockIdx.x, gridDim.x);
    const size_t step = (size_t)T * kB;
    cf* const carry_g = sp.carry + (size_t)(2 * d) * kCarrySlots * kB;   // the duo's two rings are contiguous
    for (int i = tid; i < 2 * kCarrySlots * kB; i += kBatchThreads) carry[i] = carry_g[i];
    constexpr int kPoller = 2 * 64;                                   // lane 0 of the first inverse wave
    // the doorbell as this workgroup may read it: workgroup 0 asks the host and passes the answer on, the others ask the relay
    auto read_door = [&]() -> unsigned {
        if (blockIdx.x == 0) {
            const unsigned v = __hip_atomic_load(eng.doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(eng.relay, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        }
        return __hip_atomic_load(eng.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (tid == 0) {
        s_door[2] = 0;
        // has the launch become resident?  The first and the last workgroup to begin say so in host words: a wait that
        // runs out can then tell "never started" (something ahead of it on its hardware queue) and "some workgroups are
        // kept out" (waves of another launch hold registers or LDS on their compute units) from a silent producer
        const unsigned before = __hip_atomic_fetch_add(eng.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0) __hip_atomic_store(&eng.resident[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (before + 1 == gridDim.x) __hip_atomic_store(&eng.resident[1], gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
    // The aggregator — the first inverse wave of workgroup 1 (workgroup 0 where there is only one) — takes every inverse wave's count
    // of finished buffers (8 per lane, sc1 loads) and writes the minimum into `completed` (pinned host word), from the period loop
    // and from the idle loop below.  Not workgroup 0: that one's idle loop reads the doorbell over the link, two microseconds a look,
    // and a count that waits behind such a look reaches the host that much later.
    const bool aggregator = blockIdx.x == (gridDim.x > 1 ? 1u : 0u);
    unsigned reported = 0;                                            // (meaningful in that wave only)
    auto aggregate_request = [&](u4& a, u4& b) {
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(eng.progress, 0, (int)(8u * gridDim.x), 0x00020000);
        a = __builtin_amdgcn_raw_buffer_load_b128(srd, 32u * (unsigned)lane, 0, 16);          // sc1; beyond the end: zeros dropped below
        b = __builtin_amdgcn_raw_buffer_load_b128(srd, 32u * (unsigned)lane + 16u, 0, 16);
    };
    auto aggregate_report = [&](const u4& a, const u4& b) {
        const unsigned words = 2u * gridDim.x;                        // lanes beyond the array read zeros: mask them out
        auto pick = [&](unsigned v, unsigned idx) { return idx < words ? v : 0xffffffffu; };
        unsigned m = min(min(min(pick(a[0], 8u * lane), pick(a[1], 8u * lane + 1)), min(pick(a[2], 8u * lane + 2), pick(a[3], 8u * lane + 3))),
                         min(min(pick(b[0], 8u * lane + 4), pick(b[1], 8u * lane + 5)), min(pick(b[2], 8u * lane + 6), pick(b[3], 8u * lane + 7))));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
        if (m != reported) {
            reported = m;
            if (lane == 0) __hip_atomic_store(eng.completed, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // How many buffers may be touched, asked by EVERY wave — at the top of period nb of a burst (same answer in all of them:
    // it is read from LDS, written before the previous period's closing barrier), and with idle = true between bursts.
    // Batch launches: n_buffers.  The engine's doorbell word: bits 0-29 buffers published so far, bit 31 STOP (no more will
    // come), bit 30 FLUSH (finish what is published without waiting for more).  A period runs buffer nb when buffer nb + 1
    // is there too (its operands are requested one period ahead) — or, on STOP or FLUSH, when nb is the last one published:
    // that period requests nothing, the burst ends behind it with a drain period, and the workgroup idles here until the
    // doorbell moves (the next burst starts cold: a real-time caller with ONE buffer in flight rings FLUSH with every
    // buffer).  Returns the count published (> nb), or -1: the stop rung with nothing pending — the launch ends.
    auto gate = [&](int nb, bool idle) -> int {
        {
            if (idle && __builtin_amdgcn_readfirstlane(s_door[2]) != 0) return -1;   // the doorbell ran out of time in this burst: no further wait
            bool look = !idle;                                        // an idle gate asks first: the word in LDS is the one the last burst ended on
            for (;;) {
                if (look) {
                    // (the same word in every lane: said so, or every test on it becomes an exec-masked region — the far
                    // role's request burst under a divergent branch took 2.4 instead of 1.4 us of its barrier interval)
                    const unsigned D = __builtin_amdgcn_readfirstlane(s_door[nb & 1]);
                    const int pub = (int)(D & 0x3fffffffu);
                    const bool stop = (D >> 31) != 0, flush = ((D >> 30) & 1u) != 0;
                    if (pub >= nb + 2 || ((stop || flush) && pub >= nb + 1)) return pub;
                    if (stop) return -1;                              // nothing more will come
                }
                look = true;
                __syncthreads();                                      // every wave has read the word
                if (w == 2) {                                         // the first inverse wave polls (lane 0 asks; workgroup 0's also aggregates)
                    unsigned v = 0;
                    int tries = 0;
                    const unsigned long long t_poll = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        u4 pa, pb;
                        if (aggregator && !GAB_EABL(1)) aggregate_request(pa, pb);
                        unsigned mine = 0;
                        if (lane == 0) mine = read_door();
                        v = __builtin_amdgcn_readfirstlane(mine);
                        if (aggregator && !GAB_EABL(1)) aggregate_report(pa, pb);
                        const int p2 = (int)(v & 0x3fffffffu);
                        if (p2 >= nb + 2 || (v >> 31) || (((v >> 30) & 1u) && p2 >= nb + 1)) break;
                        if ((++tries & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t_poll > eng.idle_ticks) {   // the producer is gone: stop here, say so
                            if (lane == 0) {
                                __hip_atomic_store(eng.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                s_door[2] = 1;
                            }
                            v = 0x80000000u | (unsigned)(p2 < nb ? p2 : nb);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(6);                      // (~0.2 us between looks: with 20, a quarter of a microsecond more from doorbell to count)
                    }
                    if (lane == 0) s_door[nb & 1] = v;
                }
                __syncthreads();
            }
        }
    };
    // buffer nb lives in slot nb % ring of the engine's rings (a batch launch: buffer nb itself); callers walk the slots
    // with next_slot() instead of dividing
    auto in_slot = [&](int slot) -> const float* { return in + (size_t)slot * step; };
    auto next_slot = [&](int slot) -> int { return (ENGINE && slot + 1 == eng.ring) ? 0 : slot + 1; };
    // The engine's input ring is rewritten while the launch runs (by copy engines): its loads are system-scope loads,
    // answered by memory and never by a line an L1 or an L2 kept.  The rings are ORDINARY device memory (round 4, measured
    // at 1024 channels: fine-grained rings read by non-temporal loads 6.85 us per buffer, ordinary rings read by
    // system- or agent-scope loads 6.09-6.13, by plain loads — which may be stale — 6.2-6.3).
    auto ld = [](const float* p) -> float {
        if constexpr (ENGINE)
            return GAB_EABL(4)     ? *p
                   : GAB_EABL(256) ? __builtin_nontemporal_load(p)
                                   : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sc0 sc1
        else return *p;
    };

    if (w >= 4) {
        // ---- far waves: F of the pair whose turn it is (window = blocks k-7 .. k; the two newest straight from
        // the input buffers, the next buffer's operands requested under the inverse transform); the share is
        // parked in LDS
        const int ft = tid - kThreads;
        using FB = fft::BlockFFT<kNB, 16, false>;
        using FBi = fft::BlockFFT<kNB, 16, true>;
        typename FB::Twiddles twb;                                  // every pass's powers stay in registers
        FB::load_twiddles(twb, tw, ft);
        // (Measured, not kept: the next buffer's 36 requests spread over three barrier intervals instead of
        // one burst after the spectral product — the burst's 0.6 us on the chain only moves: 5.75 vs 5.31 us.)
        // Blocks that lie inside the launch come from the input buffers (block k-j = buffer nb-j), older ones
        // from the history ring as the previous launch left it: the ring is neither read nor written in the
        // steady state of a launch (the forward waves refresh it over the launch's last eight buffers).
        // FIRST (a compile-time tag): the launch's first window, whose block k-1 is still the history ring's.  (As a
        // run-time test on nb the engine's k-1 loads became conditional loads: a register merge behind them, i.e. a
        // wait for the whole request burst in the middle of the far chain — 2.3 instead of 1.4 us for that interval.)
        auto load_window = [&](auto first_tag, int nb, int slot, int slot_before, cf (&z)[16], float4 (&c)[16], int ft) {   // (ft: the caller's copy, see the bursts)
            constexpr bool FIRST = decltype(first_tag)::value;
            const int head = (head0 + nb) & (kSlots - 1);
            const int q = 2 * d + (head & 1);
            const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
            const size_t ca = (size_t)(2 * q) * kB, cb_ = ca + kB;
            const float* const cur = in_slot(slot);
            z[14] = mk(ld(cur + ca + ft), ld(cur + cb_ + ft));
            z[15] = mk(ld(cur + ca + ft + kThreads), ld(cur + cb_ + ft + kThreads));
            if constexpr (ENGINE && !GAB_EABL(8)) {                   // k-1 from the input ring, the rest from the history ring
                if constexpr (!FIRST) {
                    const float* const prv = in_slot(slot_before);
                    z[12] = mk(ld(prv + ca + ft), ld(prv + cb_ + ft));
                    z[13] = mk(ld(prv + ca + ft + kThreads), ld(prv + cb_ + ft + kThreads));
                } else {
                    const int s = ((head + kSlots - 1) & (kSlots - 1)) * kB;
                    z[12] = hp[s + ft];
                    z[13] = hp[s + kThreads + ft];
                }
#pragma unroll
                for (int r = 0; r < 12; ++r)
                    z[r] = hp[((head + 1 + (r >> 1)) & (kSlots - 1)) * kB + (r & 1) * kThreads + ft];
            } else if (nb >= kSlots - 1) {                            // the whole window lies inside the launch
#pragma unroll
                for (int bl = 0; bl < 7; ++bl) {
                    int sb = slot - (7 - bl);                         // (engine experiments: the ring wraps)
                    if (ENGINE && sb < 0) sb += eng.ring;
                    const float* const src = in_slot(sb);
                    z[2 * bl] = mk(src[ca + ft], src[cb_ + ft]);
                    z[2 * bl + 1] = mk(src[ca + ft + kThreads], src[cb_ + ft + kThreads]);
                }
            } else {
#pragma unroll
                for (int bl = 0; bl < 7; ++bl) {                      // block k-7+bl = buffer nb-7+bl (uniform branch)
                    if (nb - 7 + bl >= 0) {
                        const float* const src = in_slot(slot - (7 - bl));       // nb < 7: no wrap yet
                        z[2 * bl] = mk(src[ca + ft], src[cb_ + ft]);
                        z[2 * bl + 1] = mk(src[ca + ft + kThreads], src[cb_ + ft + kThreads]);
                    } else {
                        const int s = ((head + 1 + bl) & (kSlots - 1)) * kB;
                        z[2 * bl] = hp[s + ft];
                        z[2 * bl + 1] = hp[s + kThreads + ft];
                    }
                }
            }
            load_spectra<kNB, 16>(c, sp.pmF + (size_t)q * kBinsB, ft);
        };
        cf zb[16], zn[16];
        float4 cb[16];
        // one period of the far role: the transform of window nb, the next window's requests under its inverse
        auto far_period = [&](int nb, int slot, int avail) {
#ifdef GAB_ABLATE
            if (GAB_SDBG(4)) {                                        // diagnostic builds: far role idle
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                return;
            }
#endif
            const int head = (head0 + nb) & (kSlots - 1);
            cf* const cp = carry + (head & 1) * kCarrySlots * kB;
            GAB_BSTAMP(6);
#ifdef GAB_ABLATE
            FB::run(zb, far_x, far_y, twb, ft, true, [&](int p) { GAB_BSTAMP(p); });
#else
            FB::run(zb, far_x, far_y, twb, ft);                       // barriers 1, 2
#endif
            partner_exchange<kNB, 16, true>(zb, zn, far_x, ft);       // barrier 3
            GAB_BSTAMP(2);
            spectral_product<kNB, 16>(zb, zn, cb, ft);
            __builtin_amdgcn_sched_barrier(0);
#ifdef GAB_ABLATE
            if (!GAB_SDBG(512)) { keep_alive(zb[0]); keep_alive(zb[15]); GAB_BSTAMP(7); }      // slot 7: the product is done
#endif
            if (nb + 1 < avail) load_window(std::false_type{}, nb + 1, next_slot(slot), slot, zn, cb, ft);   // flies under the inverse transform
            __builtin_amdgcn_sched_barrier(0);
#ifdef GAB_ABLATE
            if (GAB_SDBG(512)) GAB_BSTAMP(7);                                                   // or: the request burst has been issued
#endif
#ifdef GAB_ABLATE
            FBi::template run<typename FB::Twiddles, 4>(zb, far_y, far_x, twb, ft, true, [&](int p) { GAB_BSTAMP(3 + p); });
#else
            FBi::template run<typename FB::Twiddles, 4>(zb, far_y, far_x, twb, ft);   // barriers 4, 5; only [12..15]
#endif
            cf* const c1 = cp + ((head + 1) & (kCarrySlots - 1)) * kB;              // block k+1
            cf* const c2 = cp + ((head + 2) & (kCarrySlots - 1)) * kB;              // block k+2
            c1[ft] = zb[12];
            c1[ft + kThreads] = zb[13];
            c2[ft] = zb[14];
            c2[ft + kThreads] = zb[15];
            __syncthreads();                                          // barrier 6 closes the period
            GAB_BSTAMP(5);
#pragma unroll
            for (int r = 0; r < 16; ++r) zb[r] = zn[r];
        };
        int nb = 0, slot = 0;                                         // the next buffer and its ring slot
        for (;;) {                                                    // bursts
            int avail = gate(nb, true);
            if (nb >= avail) break;
            {
                // (an opaque copy of the thread index: inside the loop over bursts the window's burst-invariant
                // 64-bit addresses would otherwise be hoisted out of that loop and spilled)
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_window(std::true_type{}, nb, slot, 0, zb, cb, fo);   // a burst starts cold: block k-1 is the history ring's
            }
            for (;;) {
                far_period(nb, slot, avail);
                ++nb;
                slot = next_slot(slot);
                if (nb >= avail) break;                               // nothing was requested for buffer nb: the burst ends here
                avail = gate(nb, false);
                if (nb >= avail) break;                               // (the doorbell ran out of time)
            }
            for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();    // the burst's drain period: the inverse waves' alone
        }
    } else if (w < 2) {
        // ---- forward waves: wave w holds pair w of the duo
        const int q = 2 * d + w;
        cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const img = lds + w * kWaveImg;                           // the transform's exchanges, then its spectrum
        cf* const hand = lds + (2 + w) * kWaveImg;                    // output spectrum for the inverse wave
        using WF = fft::WaveFFT1024<false>;
        WF::Lean t;
        WF::load_twiddles(t, tw, lane);
        const float4* const pa = pmA + (size_t)q * kBinsA;
        const float4* const pa2 = sp.pmA2 + (size_t)q * kBinsA;
        const size_t xoff = (size_t)(2 * q) * kB;                     // channel a of a buffer; channel b is kB further
        cf z[16], prev[8], nxt[8];
        float4 c[16];
        // one period of the forward wave
        auto fwd_period = [&](int nb, int slot, int avail) {
#ifdef GAB_ABLATE
            if (GAB_SDBG(1)) {                                        // diagnostic builds: near role idle
                for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();
                return;
            }
#endif
            // One piece of work per barrier interval (the far role's transform has six):
            //   A2 share | window + pass 0 | pass 1 | pass 2 | spectrum + A product | hand-over + requests
            const int head = (head0 + nb) & (kSlots - 1);
            cf share[16];                                     // taps [512,1024): last period's spectrum x pmA2
            {
                cf vp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) share[r] = img[rb + 68 * r];
#pragma unroll
                for (int r = 0; r < 16; ++r) vp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                spectral_product<kNA, 16>(share, vp, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // an opaque copy of the lane index: sixteen loop-invariant 64-bit addresses would otherwise be
                // hoisted out of the loop, spilled, and reloaded one by one between the loads they feed
                int lo = lane;
                asm volatile("" : "+v"(lo));
                load_spectra<kNA, 16>(c, pa, lo);             // for this buffer's A product (interval 5)
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_BSTAMP(0);
            __syncthreads();                                  // barrier 1
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[j] = prev[j]; z[8 + j] = nxt[j]; }
            if ((ENGINE && !GAB_EABL(8)) || nb + kSlots >= n_buffers) {   // the ring only has to hold the launch's LAST eight blocks
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = nxt[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = nxt[j];
            if (nb + 1 < avail) {                             // the next buffer's block: needed a period from now
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(next_slot(slot)) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(xa + 64 * j), ld(xa + kB + 64 * j));
            }
#ifdef GAB_ABLATE
            WF::run(z, img, t, lane, [&](int i) { GAB_BSTAMP(1 + i); __syncthreads(); });
#else
            WF::run(z, img, t, lane, ArriveAtBarrier());      // barriers 2, 3 from inside
#endif
            GAB_BSTAMP(3);
            __syncthreads();                                  // barrier 4
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];     // the spectrum stays here for the next period
            __builtin_amdgcn_wave_barrier();
            {
                cf zp[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[r] = img[PadA16::at((kNA - (lane + 64 * r)) & (kNA - 1))];
                spectral_product<kNA, 16>(z, zp, c, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                load_spectra<kNA, 16>(c, pa2, lo);            // for the next period's A2 share
            }
            __builtin_amdgcn_sched_barrier(0);
            GAB_BSTAMP(4);
            __syncthreads();                                  // barrier 5
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = fft::cadd(z[r], share[r]);   // A product + A2 share
            GAB_BSTAMP(5);
            __syncthreads();                                  // barrier 6 closes the period
            GAB_BSTAMP(6);
        };
        int nb = 0, slot = 0;                                         // the next buffer and its ring slot
        bool primed = false;                                          // the image holds the spectrum of blocks [k-2 | k-1]
        for (;;) {                                                    // bursts
            int avail = gate(nb, true);
            if (nb >= avail) break;
            if (!primed) {
                // the launch's first burst: the spectrum of the ring's blocks [k-2 | k-1] into the image.  (Later bursts
                // find it there — the last period left the spectrum of [k-1 | k], which is the next buffer's [k-2 | k-1] —
                // with block k-1 in `prev` and the A2 spectra in `c`: only the new block is loaded.)
                const int s1 = ((head0 + kSlots - 1) & (kSlots - 1)) * kB, s2 = ((head0 + kSlots - 2) & (kSlots - 1)) * kB;
                int lo = lane;                                    // (inside the loop: see the far role)
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lo + 64 * j];
#pragma unroll
                for (int j = 0; j < 8; ++j) prev[j] = hp[s1 + lo + 64 * j];
                // (the burst's first block is asked for below, as in every burst: registers)
                load_spectra<kNA, 16>(c, pa2, lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) z[8 + j] = prev[j];
                WF::run(z, img, t, lane, WF::NoHook());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
                __builtin_amdgcn_wave_barrier();
                primed = true;
            }
            {                                                     // the burst's first block
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(slot) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(ld(xa + 64 * j), ld(xa + kB + 64 * j));
            }
            for (;;) {
                fwd_period(nb, slot, avail);
                ++nb;
                slot = next_slot(slot);
                if (nb >= avail) break;                               // nothing was requested for buffer nb: the burst ends here
                avail = gate(nb, false);
                if (nb >= avail) break;
            }
            for (int i = 0; i < kBatchBarriers; ++i) __syncthreads();    // the burst's drain period
        }
    } else {
        // ---- inverse waves: wave 2 + p turns the output spectrum of pair p into samples, one period later
        const int pr = w - 2;
        const cf* const hand = lds + (2 + pr) * kWaveImg;
        cf* const img = lds + (4 + pr) * kWaveImg;                    // the transform's exchanges, then the output swap
        const cf* const other = lds + (4 + (1 - pr)) * kWaveImg;
        const cf* const cring = carry + pr * kCarrySlots * kB;
        using WFi = fft::WaveFFT1024<true>;
        WFi::Lean t;
        WFi::load_twiddles(t, tw, lane);
        int oslot = 0;                                                // of the next buffer to leave
        int nb = 0, avail = 0, base = 0;                              // the period, buffers that may be touched, the burst's first buffer
        bool boundary = true;                                         // between bursts (the launch's start is a boundary)
        for (;;) {                                                    // periods base .. end of every burst; in a burst's last one only this role works (the drain)
            if (boundary) {
                avail = gate(nb, true);
                if (nb >= avail) break;
                base = nb;                                            // nothing to turn in a burst's first period
                boundary = false;
            } else if (nb < avail) {
                avail = gate(nb, false);                              // (nb == avail: the drain period — the other roles ask nothing either)
            }
            const bool more = nb < avail;                         // the other roles work on buffer nb in this period
            unsigned door_next = s_door[nb & 1];                  // (no per-period poll: the word as last seen)
            u4 prog_a = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, prog_b = prog_a;
            if constexpr (ENGINE) {
                if (nb >= base + 2 && !GAB_EABL(1)) {
                    // this wave's rows of buffer nb - 2 were stored a period ago: drained by now, so the wait is free,
                    // and the count of finished buffers can go out (write-through, nobody waits for it)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&eng.progress[2 * blockIdx.x + pr], (unsigned)(nb - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == kPoller && eng.poll_every_period)      // asked now, needed at the period's end
                    door_next = blockIdx.x == 0 ? __hip_atomic_load(eng.doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                : __hip_atomic_load(eng.relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (aggregator && pr == 0 && !GAB_EABL(1)) aggregate_request(prog_a, prog_b);
            }
            auto close_period = [&]() {                           // before the closing barrier
                if constexpr (ENGINE) {
                    if (aggregator && pr == 0 && !GAB_EABL(1)) aggregate_report(prog_a, prog_b);
                    if (tid == kPoller) {
                        s_door[(nb + 1) & 1] = door_next;
                        if (blockIdx.x == 0 && eng.poll_every_period)
                            __hip_atomic_store(eng.relay, door_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            };
            bool idle_period = nb == base;                        // a burst's first period: nothing to turn yet
#ifdef GAB_ABLATE
            if (GAB_SDBG(1)) idle_period = true;                  // diagnostic builds: near role idle
#endif
            if (idle_period) {
                for (int i = 0; i < kBatchBarriers - 1; ++i) __syncthreads();
                close_period();
                __syncthreads();
            } else {
                // One piece per barrier interval: hand-over read | pass 0 | pass 1 | pass 2 + far share | swap | stores
                const int b = nb - 1;                             // the buffer whose spectrum was handed over last period
                const int head = (head0 + b) & (kSlots - 1);
                float* const outb = out + (size_t)(ENGINE ? oslot : b) * step;
                oslot = next_slot(oslot);
                cf z[16], y[8], park[8];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = hand[rb + 68 * r];    // the forward wave writes the next one in interval 6
#pragma unroll
                for (int j = 0; j < 8; ++j) park[j] = cring[(head & (kCarrySlots - 1)) * kB + lane + 64 * j];
                GAB_BSTAMP(0);
                __syncthreads();                                  // barrier 1
#ifdef GAB_ABLATE
                WFi::run(z, img, t, lane, [&](int i) { GAB_BSTAMP(1 + i); __syncthreads(); });
#else
                WFi::run(z, img, t, lane, ArriveAtBarrier());     // barriers 2, 3 from inside
#endif
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
                GAB_BSTAMP(3);
                __syncthreads();                                  // barrier 4
                // the two pairs of a duo are four neighbouring channels: the waves swap halves through LDS
                // so that each stores float4 pieces (pair 0 keeps samples lane + 64 j, j < 4, pair 1 j >= 4)
                if (pr == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[4 + j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) img[lane + 64 * j] = y[j];
                }
                GAB_BSTAMP(4);
                __syncthreads();                                  // barrier 5: the swapped halves are in LDS
                {
                    float* const o0 = outb + 4 * (size_t)d;
                    auto put = [&](float* dst, float a, float b2, float c2, float d2) {
                        if (ENGINE && !GAB_EABL(2)) {             // write-through: in memory before `completed` says so
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            const f4v val = {a, b2, c2, d2};
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
                        } else {
                            *reinterpret_cast<float4*>(dst) = make_float4(a, b2, c2, d2);
                        }
                    };
                    if (pr == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const cf theirs = other[lane + 64 * j];
                            put(o0 + (size_t)T * (lane + 64 * j), y[j].x, y[j].y, theirs.x, theirs.y);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const cf theirs = other[lane + 64 * j];
                            put(o0 + (size_t)T * (lane + 64 * (4 + j)), theirs.x, theirs.y, y[4 + j].x, y[4 + j].y);
                        }
                    }
                }
                GAB_BSTAMP(5);
                close_period();
                __syncthreads();                                  // barrier 6 closes the period
                GAB_BSTAMP(6);
            }
            if (!more) {
                // the burst is through: this wave's last rows must be in memory before its count says so (the one wait for
                // stores on this path: a workgroup that goes idle has nothing to hide it behind)
                if (!GAB_EABL(1)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&eng.progress[2 * blockIdx.x + pr], (unsigned)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                boundary = true;
                continue;
            }
            ++nb;
        }
    }
    // every wave is past the last closing barrier: the duo's carry ring goes back to memory
    for (int i = tid; i < 2 * kCarrySlots * kB; i += kBatchThreads) carry_g[i] = carry[i];
}

// ---- the same on SIX waves per workgroup, one pair each, two workgroups per compute unit (round 6) --------------------------------
// Barrier timeline of the twelve-wave launch (tools/stamp_batch12.py, profiles/r06_batch12_stamps.txt): at EVERY barrier the last
// wave to arrive is a far wave, the near waves wait half of the time, and each far group spends a third of its time waiting
// for the OTHER group (an interval lasts as long as the slower of the two groups' steps, and the steps do not match).  The
// two groups share nothing but the hardware barrier.  So each pair gets a workgroup of its own — a forward wave, an inverse
// wave, four far waves; 79 KB of LDS, two workgroups per compute unit, still three waves per SIMD — and its own barrier.
// What that gives up: the duo's output as 16-byte pieces (the inverse waves of two pairs swapping halves through LDS); a
// pair stores its two channels as 8-byte pieces.  Same arithmetic: bit-identical.
constexpr int kB6Threads = 6 * kWave;
constexpr int kB6Lds = 3 * kWaveImg + kLdsHalf + 2 * kBinsA * 2 + kB12Tw1Cf;       // cf entries (79 232 bytes)

#ifdef GAB_ABLATE
#define GAB_PAIRBAR(period, i)                                                                                        \
    do {                                                                                                              \
        if (GAB_SDBG(64) && ((period) == 32 || (period) == 33) && lane == 0)                                          \
            g_split_stamps[(((size_t)blockIdx.x * 12 + wave_id) * 2 + ((period) - 32)) * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        sync();                                                                                                       \
    } while (0)
#else
#define GAB_PAIRBAR(period, i) sync()
#endif
// One pair's six waves (forward, inverse, four far).  `tid` = the thread's index among the pair's 384, `q` = the pair, `sync` = the
// barrier the six waves meet at (the workgroup's hardware barrier where the workgroup IS the pair; a counter in LDS where two
// pairs share a workgroup), `wave_id` = the wave's index in the workgroup (for the diagnostic stamps only).
template <bool LEAN_TW, class Sync>
__device__ __forceinline__ void conv_split_pair_resident(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, const ConvSplit& sp, const cf* __restrict__ tw, int T, int head0, int n_buffers,
    cf* __restrict__ lds, const int tid, const int q, Sync& sync, const int wave_id) {
    cf* const far_img = lds + 3 * kWaveImg;                           // [kLdsHalf]
    float4* const spec = reinterpret_cast<float4*>(far_img + kLdsHalf);   // [A | A2][513]
    cf* const tw1 = reinterpret_cast<cf*>(spec + 2 * kBinsA);          // [16][15]: W256^(r k), k = thread & 15 (the middle pass)
    const size_t step = (size_t)T * kB;
    {
        const float4* const gA = pmA + (size_t)q * kBinsA;
        const float4* const gA2 = sp.pmA2 + (size_t)q * kBinsA;
        for (int i = tid; i < kBinsA; i += kB6Threads) { spec[i] = gA[i]; spec[kBinsA + i] = gA2[i]; }
        if (tid < 16) {
            cf pw[15];
            fft::powers_of<16>(tw[tid * (fft::kTwiddleN / 256)], pw);
#pragma unroll
            for (int r = 0; r < 15; ++r) tw1[tid * 15 + r] = pw[r];
        }
    }
    sync();
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned rb = (unsigned)lane + ((unsigned)lane >> 4);      // Pad(lane + 64 r) = rb + 68 r
    auto in_slot = [&](int slot) -> const float* { return in + (size_t)slot * step; };
    auto idle_period = [&]() { for (int i = 0; i < kBatchBarriers; ++i) sync(); };

    if (w >= 2) {
        // ---- far waves: the pair's transform, one per two periods (its turn comes every other buffer)
        const int g = q & 1;                                          // the pair's turn: buffers whose ring slot has this parity
        const int ft = tid - 2 * kWave;
        cf* const img = far_img;
        const cf* const hp = reinterpret_cast<const cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const cg = sp.carry + (size_t)q * kCarrySlots * kB;
        const size_t ca = (size_t)(2 * q) * kB, cb_ = ca + kB;
        const float4* const pf = sp.pmF + (size_t)q * kBinsB;
        using B16 = fft::Butterfly<16, false>;
        using B16i = fft::Butterfly<16, true>;
        // the last pass's powers, W4096^(r t): kept (30 registers), or — LEAN_TW, the 128-register build — re-formed from their
        // base where they are used (the same powers_of: the same values)
        cf tw2_kept[LEAN_TW ? 1 : 15];
        const cf tw2_base = tw[ft];
        if constexpr (!LEAN_TW) fft::powers_of<16>(tw2_base, tw2_kept);
        auto tw2_of = [&](cf (&w)[15]) {
            if constexpr (LEAN_TW) {
                cf b = tw2_base;
                asm volatile("" : "+v"(b.x), "+v"(b.y));             // pinned to the pass
                fft::powers_of<16>(b, w);
            } else {
#pragma unroll
                for (int r = 0; r < 15; ++r) w[r] = tw2_kept[r];
            }
        };
        // LDS positions are formed where they are used, from an opaque copy of the thread index: as loop invariants they
        // would be kept (and spilled) across both halves of a transform
        auto opaque_t = [&]() -> unsigned { unsigned v = (unsigned)ft; asm volatile("" : "+v"(v)); return v; };
        auto w0_of = [](unsigned t) -> unsigned { return t * 17u; };                       // pass-0 writes: Pad(16 t + r) = 17 t + r
        auto rd_of = [](unsigned t) -> unsigned { return t + (t >> 4); };                  // linear reads: Pad(t + 256 r) = rd + 272 r
        auto w1_of = [](unsigned t) -> unsigned { const unsigned b1 = (t >> 4) * 256u + (t & 15u); return b1 + (b1 >> 4); };   // pass-1 writes: + 17 r
        auto tw1row_of = [&](unsigned t) -> const cf* { return tw1 + (t & 15u) * 15u; };
        // PART 0: the seven older blocks' first halves + the newest block (16 requests); PART 1: the rest (16 requests) — the
        // requests of one window go out in two intervals: thirty-two in one made that interval the period's longest
        auto load_window = [&](auto part_tag, int nb, cf (&z)[16]) {
            constexpr int PART = decltype(part_tag)::value;
            const int head = (head0 + nb) & (kSlots - 1);
            int fo = ft;
            asm volatile("" : "+v"(fo));                              // (addresses formed here, not hoisted and spilled)
            if constexpr (PART == 0) {
                const float* const cur = in_slot(nb);
                z[14] = mk(cur[ca + fo], cur[cb_ + fo]);
                z[15] = mk(cur[ca + fo + kThreads], cur[cb_ + fo + kThreads]);
            }
            constexpr int BL0 = PART == 0 ? 4 : 0, BL1 = PART == 0 ? 7 : 4;       // blocks k-7+bl
            if (nb >= kSlots - 1) {
#pragma unroll
                for (int bl = BL0; bl < BL1; ++bl) {
                    const float* const src = in_slot(nb - (7 - bl));
                    z[2 * bl] = mk(src[ca + fo], src[cb_ + fo]);
                    z[2 * bl + 1] = mk(src[ca + fo + kThreads], src[cb_ + fo + kThreads]);
                }
            } else {
#pragma unroll
                for (int bl = BL0; bl < BL1; ++bl) {                  // block k-7+bl = buffer nb-7+bl (uniform branch)
                    if (nb - 7 + bl >= 0) {
                        const float* const src = in_slot(nb - (7 - bl));
                        z[2 * bl] = mk(src[ca + fo], src[cb_ + fo]);
                        z[2 * bl + 1] = mk(src[ca + fo + kThreads], src[cb_ + fo + kThreads]);
                    } else {
                        const int s = ((head + 1 + bl) & (kSlots - 1)) * kB;
                        z[2 * bl] = hp[s + fo];
                        z[2 * bl + 1] = hp[s + kThreads + fo];
                    }
                }
            }
        };
        cf z[16], zn[16];
        // The requests ride on the lighter steps (every instruction of a step costs the wave about a dozen clocks beside two
        // others on its SIMD: thirty-two requests in one interval made it the period's longest), the wait for the carry's stores
        // stands in the idle interval.  (Measured and not kept, profiles/r06_batch12_stamps.txt: steps cut as read + twiddle |
        // butterfly + write, so that no step is a bare write — 5.08 against 5.01 us per buffer.)
        // first half of a transform (period nb):   pass 0, write | read, twiddle, pass 1 | write (+ spectra 0-7) |
        //                                          read, twiddle (+ spectra 8-15), pass 2 | partner write | partner read, product, inverse pass 0
        auto first_half = [&](int nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = zn[r];
            B16::run(z);
            {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16::out_slot(r)];
            }
            GAB_PAIRBAR(nb, 0);                                          // 1
            {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw1row[r - 1]);
            }
            B16::run(z);
            GAB_PAIRBAR(nb, 1);                                          // 2: every wave has read
            {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16::out_slot(r)];
            }
            float4 clo[8], chi[8];                                    // the far spectra (LEAN_TW: asked for later, registers)
            if constexpr (!LEAN_TW) {
                __builtin_amdgcn_sched_barrier(0);
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 0, 8>(clo, pf, fo);
                __builtin_amdgcn_sched_barrier(0);
            }
            GAB_PAIRBAR(nb, 2);                                          // 3
            {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
            }
            {
                cf tw2[15];
                tw2_of(tw2);
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmul(z[r], tw2[r - 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                int fo = ft;
                asm volatile("" : "+v"(fo));
                if constexpr (LEAN_TW) load_spectra_part<kNB, 16, 0, 8>(clo, pf, fo);      // (behind the twiddle multiply: its powers are dead)
                else load_spectra_part<kNB, 16, 8, 16>(chi, pf, fo);
            }
            __builtin_amdgcn_sched_barrier(0);
            B16::run(z);
            GAB_PAIRBAR(nb, 3);                                          // 4
            {
                const unsigned t = opaque_t();
#pragma unroll
                for (int r = 0; r < 16; ++r) img[t + 256u * r] = z[B16::out_slot(r)];   // Z[t + 256 r]: the partner exchange, raw
            }
            if constexpr (LEAN_TW) {
                __builtin_amdgcn_sched_barrier(0);
                int fo = ft;
                asm volatile("" : "+v"(fo));
                load_spectra_part<kNB, 16, 8, 16>(chi, pf, fo);
                __builtin_amdgcn_sched_barrier(0);
            }
            GAB_PAIRBAR(nb, 4);                                          // 5
            {
                // the thread's own bins back in order, and Z[(N - k) mod N], k = t + 256 r: one base and constant offsets for
                // r >= 1 (N - k > 0 there); bin k = t alone wraps
                cf o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = z[B16::out_slot(r)];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = o[r];            // z[r] = Z[t + 256 r]
                const unsigned t = opaque_t();
                const cf* const pb = img + (kNB - 256 * 15) - t;      // pb[256 (15 - r)] = img[N - t - 256 r]
                cf zp[8];
                zp[0] = img[(kNB - t) & (kNB - 1)];
#pragma unroll
                for (int r = 1; r < 8; ++r) zp[r] = pb[256 * (15 - r)];
                spectral_product_part<kNB, 16, 0, 8>(z, zp, clo, ft);
#pragma unroll
                for (int r = 0; r < 8; ++r) zp[r] = pb[256 * (7 - r)];
                spectral_product_part<kNB, 16, 8, 16>(z, zp, chi, ft);
            }
            B16i::run(z);
            GAB_PAIRBAR(nb, 5);                                          // 6
        };
        // second half (period nb + 1):   (half of the next window's requests) | write | read, twiddle, pass 1 |
        //                                write (+ the other half) | read, twiddle, last pass (4 of 16), carry out | (the carry's stores leave)
        auto second_half = [&](int period, int nb_done, int nb_next) {
            if (nb_next >= 0) {
                load_window(std::integral_constant<int, 0>{}, nb_next, zn);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zn[r] = mk(0.0f, 0.0f);  // (no value survives from the last window: registers)
            }
            GAB_PAIRBAR(period, 0);                                          // 1
            if (nb_done >= 0) {
                const unsigned w0 = w0_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w0 + r] = z[B16i::out_slot(r)];
            }
            GAB_PAIRBAR(period, 1);                                          // 2
            if (nb_done >= 0) {
                const unsigned t = opaque_t(), rd = rd_of(t);
                const cf* const tw1row = tw1row_of(t);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
#pragma unroll
                for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw1row[r - 1]);
                B16i::run(z);
            }
            GAB_PAIRBAR(period, 2);                                          // 3
            if (nb_done >= 0) {
                const unsigned w1 = w1_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) img[w1 + 17 * r] = z[B16i::out_slot(r)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nb_next >= 0) load_window(std::integral_constant<int, 1>{}, nb_next, zn);      // the window's other half, beside the light write
            __builtin_amdgcn_sched_barrier(0);
            GAB_PAIRBAR(period, 3);                                          // 4
            if (nb_done >= 0) {
                const unsigned rd = rd_of(opaque_t());
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = img[rd + 272 * r];
                {
                    cf tw2[15];
                    tw2_of(tw2);
#pragma unroll
                    for (int r = 1; r < 16; ++r) z[r] = fft::cmulc(z[r], tw2[r - 1]);
                }
                cf x12, x13, x14, x15;
                B16i::run_last4(z, x12, x13, x14, x15);
                const int head = (head0 + nb_done) & (kSlots - 1);
                cf* const c1 = cg + ((head + 1) & (kCarrySlots - 1)) * kB;    // block k+1
                cf* const c2 = cg + ((head + 2) & (kCarrySlots - 1)) * kB;    // block k+2
                c1[ft] = x12;
                c1[ft + kThreads] = x13;
                c2[ft] = x14;
                c2[ft + kThreads] = x15;
            }
            GAB_PAIRBAR(period, 4);                                          // 5
            // the inverse wave of this pair asks for block k+1's share behind the period's closing barrier: the stores must
            // have left this wave by then (waited for HERE, in the group's idle interval, not on its last pass)
            if (nb_done >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GAB_PAIRBAR(period, 5);                                          // 6
        };
        const int first = (g - head0) & 1;                             // this group's first window
        int nb = 0;
        if (first == 1) {
            second_half(0, -1, 1 < n_buffers ? 1 : -1);                   // period 0: nothing to finish, window 1 asked for
            nb = 1;
        } else if (n_buffers > 0) {
            load_window(std::integral_constant<int, 0>{}, 0, zn);
            load_window(std::integral_constant<int, 1>{}, 0, zn);
        }
        for (;;) {                                                     // nb: a period in which a transform of this group starts
            if (nb > n_buffers) break;
            if (nb < n_buffers) first_half(nb); else idle_period();
            ++nb;
            if (nb > n_buffers) break;
            second_half(nb, nb - 1 < n_buffers ? nb - 1 : -1, nb + 1 < n_buffers ? nb + 1 : -1);
            ++nb;
        }
    } else if (w == 0) {
        // ---- the forward wave
        cf* const hp = reinterpret_cast<cf*>(hist) + (size_t)q * kSlots * kB;
        cf* const img = lds;                                          // the transform's exchanges, then its spectrum
        cf* const hand = lds + kWaveImg;                              // A2 share, then the output spectrum for the inverse wave
        const float4* const sa = spec;                                // taps [0,512)
        const float4* const sa2 = spec + kBinsA;                      // taps [512,1024)
        using WF = fft::WaveFFT1024<false>;
        WF::Lean t;
        WF::load_twiddles(t, tw, lane);
        const size_t xoff = (size_t)(2 * q) * kB;                     // channel a of a buffer; channel b is kB further
        cf z[16], prev[8], nxt[8];
        // W = Z x spectra (from LDS) in halves of eight bins: partner values and spectra of a half live in registers at a time
        auto product_from_image = [&](cf (&v)[16], const float4* sp_lds) {
            auto quarter = [&](auto r0_tag) {
                constexpr int R0 = decltype(r0_tag)::value;
                cf zp[4];
                float4 ch[4];
                int lo = lane;
                asm volatile("" : "+v"(lo));                          // (LDS positions formed here: as loop invariants they were kept and spilled)
#pragma unroll
                for (int r = 0; r < 4; ++r) zp[r] = img[PadA16::at((kNA - (lo + 64 * (R0 + r))) & (kNA - 1))];
                load_spectra_part<kNA, 16, R0, R0 + 4>(ch, sp_lds, lo);
                spectral_product_part<kNA, 16, R0, R0 + 4>(v, zp, ch, lane);
                __builtin_amdgcn_sched_barrier(0);
            };
            quarter(std::integral_constant<int, 0>{});
            quarter(std::integral_constant<int, 4>{});
            quarter(std::integral_constant<int, 8>{});
            quarter(std::integral_constant<int, 12>{});
        };
        {   // prologue: the spectrum of the ring's blocks [k-2 | k-1] into the image
            const int s1 = ((head0 + kSlots - 1) & (kSlots - 1)) * kB, s2 = ((head0 + kSlots - 2) & (kSlots - 1)) * kB;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = hp[s2 + lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = hp[s1 + lane + 64 * j];
            if (n_buffers > 0) {
                const float* const x0 = in_slot(0) + xoff;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(x0[lane + 64 * j], x0[kB + lane + 64 * j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) z[8 + j] = prev[j];
            WF::run(z, img, t, lane, WF::NoHook());
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];
            __builtin_amdgcn_wave_barrier();
        }
        for (int nb = 0; nb < n_buffers; ++nb) {
            // A2 share | hand-over + window + pass 0 | pass 1 | pass 2 | spectrum + A product | sum into the hand-over
            const int head = (head0 + nb) & (kSlots - 1);
            {
                cf share[16];                                         // taps [512,1024): last period's spectrum x its spectra
#pragma unroll
                for (int r = 0; r < 16; ++r) share[r] = img[rb + 68 * r];
                product_from_image(share, sa2);
                GAB_PAIRBAR(nb, 0);                                   // barrier 1: the inverse wave has read the hand-over image
#pragma unroll
                for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = share[r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { z[j] = prev[j]; z[8 + j] = nxt[j]; }
            if (nb + kSlots >= n_buffers) {                           // the ring only has to hold the launch's LAST eight blocks
#pragma unroll
                for (int j = 0; j < 8; ++j) hp[head * kB + lane + 64 * j] = nxt[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = nxt[j];
            if (nb + 1 < n_buffers) {                                 // the next buffer's block: needed a period from now
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const float* const xa = in_slot(nb + 1) + xoff + lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) nxt[j] = mk(xa[64 * j], xa[kB + 64 * j]);
            }
            WF::run(z, img, t, lane, [&](int i) { GAB_PAIRBAR(nb, 1 + i); });   // barriers 2, 3 from inside
            GAB_PAIRBAR(nb, 3);                                       // barrier 4
#pragma unroll
            for (int r = 0; r < 16; ++r) img[rb + 68 * r] = z[r];     // the spectrum stays here for the next period
            __builtin_amdgcn_wave_barrier();
            product_from_image(z, sa);
            GAB_PAIRBAR(nb, 4);                                       // barrier 5
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[rb + 68 * r] = fft::cadd(z[r], hand[rb + 68 * r]);   // A product + A2 share
            GAB_PAIRBAR(nb, 5);                                       // barrier 6 closes the period
        }
        idle_period();                                                // the pipeline's last period
    } else {
        // ---- the inverse wave: turns the output spectrum into samples, one period later
        const cf* const hand = lds + kWaveImg;
        cf* const img = lds + 2 * kWaveImg;                           // the transform's exchanges
        const cf* const cg = sp.carry + (size_t)q * kCarrySlots * kB;
        using WFi = fft::WaveFFT1024<true>;
        WFi::Lean t;
        WFi::load_twiddles(t, tw, lane);
        idle_period();                                                // first period: nothing to turn yet
        for (int nb = 1; nb <= n_buffers; ++nb) {
            // hand-over read + far share asked for | pass 0 | pass 1 | pass 2 + far share | - | stores
            const int b = nb - 1;                                     // the buffer whose spectrum was handed over last period
            const int head = (head0 + b) & (kSlots - 1);
            float* const outb = out + (size_t)b * step;
            cf z[16], y[8], park[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = hand[rb + 68 * r];
            {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                // written by this workgroup's far waves one to three periods ago, behind barriers; agent-scope loads (answered
                // by the L2, never by a line this compute unit's L1 kept from the slot's previous use four buffers ago)
                const unsigned long long* const cy = reinterpret_cast<const unsigned long long*>(cg + (head & (kCarrySlots - 1)) * kB + lo);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned long long v = __hip_atomic_load(cy + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    park[j] = mk(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
                }
            }
            GAB_PAIRBAR(nb, 0);                                       // barrier 1
            WFi::run(z, img, t, lane, [&](int i) { GAB_PAIRBAR(nb, 1 + i); });  // barriers 2, 3 from inside
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = fft::cadd(z[8 + j], park[j]);
            GAB_PAIRBAR(nb, 3);                                       // barrier 4
            {
                float* const o0 = outb + 2 * (size_t)q;               // the pair's two channels: eight bytes per sample
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<float2*>(o0 + (size_t)T * (lane + 64 * j)) = make_float2(y[j].x, y[j].y);
            }
            GAB_PAIRBAR(nb, 4);                                       // barrier 5
            GAB_PAIRBAR(nb, 5);                                       // barrier 6 closes the period
        }
    }
}

struct HardwareBarrier { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
// Six waves of a twelve-wave workgroup meet at a counter in LDS: lane 0 of each adds one (a release: the wave's LDS writes are
// behind it in the LDS queue), every wave then reads the counter until all six of this round are in (an acquire).  Bounded:
// a wave that is never joined goes on after about a second (the results are then wrong, the launch still ends).
struct CounterBarrier {
    unsigned* count;
    unsigned target;
    bool dead;                       // a wait has run out: no further waits (the launch ends at once, its results are wrong)
    __device__ __forceinline__ void operator()() {
        target += 6u;
        if ((threadIdx.x & 63u) == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (dead) return;
        for (int tries = 0;; ++tries) {
            const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
            if ((int)(v - target) >= 0) break;
            if (tries > (1 << 22)) { dead = true; break; }
            __builtin_amdgcn_s_sleep(0);
        }
    }
};

__global__ __launch_bounds__(kB6Threads, 3) void conv_split_batch6_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, int n_buffers) {
    __shared__ __attribute__((aligned(16))) cf lds[kB6Lds];
    HardwareBarrier sync;
    // the pair: neighbours in one XCD, their output lines meet in its L2
    conv_split_pair_resident<false>(in, out, hist, pmA, sp, tw, T, head0, n_buffers, lds, (int)threadIdx.x,
                                    xcd_contiguous(blockIdx.x, gridDim.x), sync, (int)(threadIdx.x >> 6));
}

// The same at FOUR waves per SIMD (128 registers: the last pass's twiddle powers re-formed where they are used), so that two of
// these workgroups — 2, 2, 1, 1 waves on the SIMDs each — fit one compute unit.
// (Its LDS is DYNAMIC: with the 79 KB as a static array the compiler knows that two workgroups = three waves per SIMD on average
// fit a compute unit and hands out 168 registers whatever the launch bounds say; the SIMDs that get FOUR of the twelve waves need 128.)
__global__ __launch_bounds__(kB6Threads, 4) void conv_split_batch6r_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, int n_buffers) {
    extern __shared__ __attribute__((aligned(16))) cf lds_dyn[];
    HardwareBarrier sync;
    conv_split_pair_resident<true>(in, out, hist, pmA, sp, tw, T, head0, n_buffers, lds_dyn, (int)threadIdx.x,
                                   xcd_contiguous(blockIdx.x, gridDim.x), sync, (int)(threadIdx.x >> 6));
}

// ---- two pairs per workgroup again, each with a barrier of its own (round 6) -------------------------------------------------------
// conv_split_batch6_kernel runs a period in 3.43 us where ONE of its workgroups has a compute unit to itself (512 channels) —
// but two never share one (a workgroup's six waves go to the SIMDs as 2, 2, 1, 1, and twice that is four waves of 166
// registers on a SIMD: 1024 channels take two rounds, 6.9 us: profiles/r06_batch_forms.txt).  Here the two pairs of a duo are
// again ONE workgroup of twelve waves — three per SIMD, 158 KB of LDS — but each pair's six waves meet at a counter in LDS
// instead of the workgroup's hardware barrier, which all twelve would have to reach: nothing couples the two pairs.
constexpr int kB26Threads = 2 * kB6Threads;
__global__ __launch_bounds__(kB26Threads) void conv_split_batch2x6_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, int n_buffers) {
    __shared__ __attribute__((aligned(16))) cf lds[2 * kB6Lds];
    __shared__ unsigned counters[2 * 32];                             // one per pair, a line apart
    if (threadIdx.x < 2) counters[32 * threadIdx.x] = 0;
    __syncthreads();
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) >= 6 ? 1 : 0);
    CounterBarrier sync{counters + 32 * h, 0u, false};
    conv_split_pair_resident<false>(in, out, hist, pmA, sp, tw, T, head0, n_buffers, lds + h * kB6Lds, (int)threadIdx.x - h * kB6Threads,
                                    2 * xcd_contiguous(blockIdx.x, gridDim.x) + h, sync, (int)(threadIdx.x >> 6));
}


__global__ __launch_bounds__(kBatchThreads, 2) void conv_split_batch_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, int n_buffers) {
    __shared__ cf lds[kBatchLds];
    conv_split_batch_resident(in, out, hist, pmA, sp, tw, T, head0, n_buffers, lds);
}

__global__ __launch_bounds__(kBatchThreads, 2) void conv_split_engine_kernel(
    const float* __restrict__ in, float* __restrict__ out, float* __restrict__ hist,
    const float4* __restrict__ pmA, ConvSplit sp, const cf* __restrict__ tw, int T, int head0, ConvEngine eng) {
    __shared__ cf lds[kBatchLds];
    __shared__ unsigned door[4];          // [0], [1] the doorbell as last seen, by period parity; [2] the engine has given up
    conv_split_engine_resident(in, out, hist, pmA, sp, tw, T, head0, eng, lds, door);
}
