/*
 * gab_c_api.h — C ABI of libgab_hip.so, the MI355X (gfx950) implementation of
 * the gpuaudiobench hot path.
 *
 * The reference (tskare/gpuaudiobench, cuda/) has no FFI of its own: its
 * boundary is the C++ class GPUABenchmark (cuda/bench_base.cuh:18-139) plus one
 * __global__ kernel (or cuFFT pipeline) per benchmark.  This header is what a
 * binding for that path would import:
 *   - section K: one entry point per reference kernel / vendor-library call
 *     site, taking DEVICE pointers, sizes and a HIP stream;
 *   - section P: plan objects for the two stateful pipelines (FFT convolution,
 *     FDTD3D);
 *   - section H: the harness itself (create/setup/run/validate by registry
 *     name), mirroring main.cu's runSelectedBenchmark (cuda/main.cu:117-164).
 * C++ users include include/gab/ *.hpp instead and get the reference's class
 * surface directly.
 *
 * Conventions: plain pointers and sizes only; every function returns GAB_OK (0),
 * a positive hipError_t value, or a negative GAB_ERR_*; no exception crosses
 * this boundary (gab_last_error() holds the text, thread-local).  All kernels
 * are asynchronous on `stream` (a hipStream_t cast to void*; NULL = default
 * stream) unless stated otherwise.  Layouts follow the reference:
 * "track-major" = [t*B + s], "sample-major" = [T*s + t].
 */
#ifndef GAB_C_API_H
#define GAB_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAB_OK 0
#define GAB_ERR_INVALID_ARG (-1)
#define GAB_ERR_RUNTIME     (-2)
#define GAB_ERR_UNSUPPORTED (-3)

typedef void* gab_stream_t;

int         gab_version(void);            /* 10000*major + 100*minor + patch   */
const char* gab_last_error(void);
int         gab_device_count(int* count); /* cuda/main.cu:307-314              */

/* ===================================================================== */
/* K. kernels                                                            */
/* ===================================================================== */

/* NoOpKernel (cuda/bench_noop.cu:9-16): out[i] = in[i], i < n.  (The
 * reference launch covers only ceil(T/256)*256 elements; this copies all n.) */
int gab_noop(const float* d_in, float* d_out, size_t n, gab_stream_t stream);

/* GainKernel (cuda/bench_gain.cu:6-24): out[i] = gain * in[i].  Bit-exact.   */
int gab_gain(const float* d_in, float* d_out, size_t n, float gain,
             gab_stream_t stream);

/* GainStatsKernel (cuda/bench_gainstats.cu:7-31): out = gain*in (track-major
 * T x B); stats[2t] = mean of the INPUT track, stats[2t+1] = max (from -1e9). */
int gab_gainstats(const float* d_in, float* d_out, float* d_stats, int tracks,
                  int bufsize, float gain, gab_stream_t stream);

/* DataTransferKernel (cuda/bench_datatransfer.cu:15-25):
 * out[i] = i < in_size ? in[i] : 0.5f + 0.5f*sinf(0.001f*i), i < out_size.    */
int gab_datatransfer(const float* d_in, float* d_out, int in_size, int out_size,
                     gab_stream_t stream);

/* The same operation with both link directions busy at once (replaces the reference's H2D -> kernel -> D2H
 * sequence around DataTransferKernel, cuda/bench_datatransfer.cu:62-75, in one call).  h_in [in_size] is any
 * host memory hipMemcpyAsync accepts (pinned for the overlap; pageable memory is uploaded completely before the
 * launch); h_out [out_size] MUST be pinned (hipHostMalloc) or
 * device memory: the kernel writes it itself while ONE engine copy of h_in lands in the plan's staging buffer.
 * Returns when h_out is complete AND the whole input has been uploaded; bit-identical to gab_datatransfer on the
 * uploaded input.  in_size <= the plan's max_in_size.  The call blocks (it is the benchmark's timed unit); `stream`
 * carries the kernel.  One call at a time per plan.  GAB_ERR_RUNTIME: a wait inside the launch ran out (about a
 * second) — h_out is then invalid and the plan has been re-armed for the next call.                              */
typedef struct gab_link_plan gab_link_plan;
int gab_link_plan_create(int max_in_size, gab_link_plan** out);
void gab_link_plan_destroy(gab_link_plan* plan);
int gab_datatransfer_round_trip(gab_link_plan* plan, const float* h_in, float* h_out, int in_size, int out_size,
                                gab_stream_t stream);
/* The kernel takes input words while the upload is still running: that rests on engine writes landing whole and once (an
 * observation).  Since round 6 a second, small launch behind every call — ordered behind the upload's completion event —
 * compares what the kernel took with what the COMPLETED upload left and puts the staging buffer back; it costs the call
 * nothing (the call returns on the main launch's end).  Its verdict is read by the plan's NEXT call (which returns
 * GAB_ERR_RUNTIME: the PREVIOUS call's output was wrong) or by this function (waits a few microseconds for the check).   */
int gab_datatransfer_round_trip_check(gab_link_plan* plan);

/* ---- keep-warm (additive; no counterpart in the reference, whose iterations run back to back) ---------------------
 * A device left idle for a DAW slot (512 / 48000 s = 10.667 ms) answers the next call later than one that has just been
 * busy: at C3 gab_conv_round_trip's p50 is 76-81 us one call per slot against 67-70 us back to back on the same box — and
 * 67-68 us per slot with EIGHT idle waves resident beside (profiles/r05_paced_keep_warm.txt: one wave buys nothing — the
 * workgroups of a launch go round the eight XCDs, eight wake them all; 64 waves and more cost, their looks cross the link
 * the round trip is using).  A gab_keep_warm is that launch: `workgroups` single-wave workgroups that sleep and look at a
 * pinned word every ~64 us (no LDS, no memory traffic but the look), on a highest-priority stream of their own.
 * gab_keep_warm_kick starts the launch if none is there and pushes its end out otherwise: the launch ends by itself
 * `idle_seconds` after the last kick (and at destroy, at once).  Kick once per slot.
 * Opt-in, because it costs what resident waves cost (power, one wave slot on `workgroups` compute units) and because,
 * like the engine below, the launch is THERE: hipDeviceSynchronize, hipFree and hipHostFree — which wait for every launch
 * on the device — return only once it has ended (up to idle_seconds after the last kick), and work queued on another
 * stream that the runtime maps to the same hardware queue stands behind it for as long (streams of the default priority
 * never share a queue with it: profiles/r05_incident_engine_queue_sharing.txt).  Keep idle_seconds short: a few slots.
 * gab_conv_round_trip_keep_warm(plan, 1) makes every gab_conv_round_trip of that plan end with a kick (0: no more kicks,
 * the launch ends idle_seconds later; the plan owns the object: 8 workgroups, idle limit = eight buffer periods at
 * 44.1 kHz, at least 0.05 s).
 * NOT beside a resident engine, and the library enforces it (it makes every keep-warm launch and every engine, so it knows):
 * gab_conv_engine_start needs every compute unit whole (its workgroup fills the register files) — with keep-warm waves on eight of
 * them its first buffer waited out their idle limit (484 ms on record).  gab_conv_engine_start drops the plan's OWN keep-warm and
 * returns GAB_ERR_INVALID_ARG while any other keep-warm launch is running on the device (destroy the object, or let it run out);
 * gab_keep_warm_kick returns GAB_ERR_INVALID_ARG when it would have to START a launch while an engine is resident on the device.
 * The engine keeps the device awake itself.
 * Workgroup 0 of the launch alone decides that it has been idle long enough; the other waves leave when it says so.
 * One thread at a time per object (like a plan).  gab_keep_warm_running: is the launch on the device right now?
 * gab_keep_warm_placement: where the waves of the current (or last) launch landed — for each wave that has started, the raw
 * HW_ID register (gfx9 layout: bits 3:0 wave slot, 5:4 SIMD, 11:8 compute unit, 12 shader array, 15:13 shader engine) and the
 * XCC_ID register (bits 3:0: the XCD); `*started` = how many have (of `workgroups`); at most `capacity` pairs are written.
 * Eight waves are meant to sit on eight XCDs: a paced measurement that prints this beside its p50 classifies itself.          */
typedef struct gab_keep_warm gab_keep_warm;
int gab_keep_warm_create(gab_keep_warm** out, int workgroups, double idle_seconds);
int gab_keep_warm_kick(gab_keep_warm* warm);
int gab_keep_warm_running(gab_keep_warm* warm, int* running);
int gab_keep_warm_placement(gab_keep_warm* warm, unsigned* hw_id, unsigned* xcc_id, int capacity, int* started);
int gab_keep_warm_destroy(gab_keep_warm* warm);

/* IIRFilterKernel (cuda/bench_iir.cu:10-44): DF-II biquad per track,
 * coeffs = {b0,b1,b2,a1,a2} (HOST pointer, 5 floats), d_state = T x {z1,z2}
 * read and written back.
 * gab_iir: one wavefront per track, wave-level scan of the state recurrence
 * (bufsize in {64,128,256,512,1024}, 16-byte aligned buffers; other shapes take
 * the sequential kernel).  Re-associates the recurrence: within ~1e-7 of the
 * golden, not bit-identical.
 * gab_iir_sequential: one lane per track in the golden's exact operation order,
 * bit-identical to it.                                                        */
int gab_iir(const float* d_in, float* d_out, const float* coeffs,
            float* d_state, int tracks, int bufsize, gab_stream_t stream);
int gab_iir_sequential(const float* d_in, float* d_out, const float* coeffs,
                       float* d_state, int tracks, int bufsize, gab_stream_t stream);

/* Conv1DTextureMemoryImplKernel (cuda/bench_conv1d.cu:7-27) with the CPU
 * golden's semantics (:188-208): y[t*B+i] = sum_j h[t*L+j] * x_flat[t*B+i-j]
 * over the FLAT input (history = previous track), j ascending.  d_ir is
 * T x L track-major (replaces the 2-D texture).                              */
int gab_conv1d(const float* d_in, float* d_out, const float* d_ir, int ir_len,
               int tracks, int bufsize, gab_stream_t stream);

/* The same for a channel SHARD (additive): d_in starts halo_tracks tracks BEFORE the first of the
 * `tracks` computed — the preceding tracks' rows, whose last ir_len-1 samples are the first track's history in
 * the golden's flat indexing (halo_tracks = min(first global track, ceil((ir_len-1)/bufsize))); d_ir and d_out
 * hold the computed tracks only.  halo_tracks = 0 is gab_conv1d.                                        */
int gab_conv1d_shard(const float* d_in, float* d_out, const float* d_ir, int ir_len,
                     int tracks, int bufsize, int halo_tracks, gab_stream_t stream);

/* RndMemKernel (cuda/bench_rndmem.cu:7-20): out[T*i+t] = pool[playhead[t]+i].
 * pool_elems is used to range-check nothing on device; it is the caller's
 * promise that playhead[t]+bufsize <= pool_elems.  Bit-exact copy.           */
int gab_rndmem(const float* d_pool, const int* d_playheads, float* d_out,
               int tracks, int bufsize, gab_stream_t stream);

/* ModalSynthesisKernel (cuda/bench_modal.cu:15-36), placeholder semantics:
 * out[i*B+s] = params[8i+0] * Re(exp(0.5+0.5i)) for i < out_tracks.           */
int gab_modal(const float* d_params, float* d_out, int n_modes, int bufsize,
              int out_tracks, gab_stream_t stream);

/* The real bank — the reference's Metal kernel BenchmarkModalFilterBank
 * (metal-swift/MetalSwiftBench/Metal/kernels_benchmark_staging.metal:121-162; golden
 * Benchmarks/ModalFilterBankBenchmark.swift:73-101) on the same 8-float parameter
 * records: out[(m % out_tracks)*B + i] = sum over modes of amp * Re(state * e^{i 2 pi f (i+1)}),
 * fp32 phasor recurrence, fixed summation order (no atomics).  out_tracks in 1..64;
 * d_params 16-byte aligned; d_workspace holds gab_modal_bank_workspace_bytes().  */
size_t gab_modal_bank_workspace_bytes(int n_modes, int out_tracks, int bufsize);
int gab_modal_bank(const float* d_params, float* d_out, int n_modes, int bufsize, int out_tracks,
                   float* d_workspace, gab_stream_t stream);

/* WaveguideState (cuda/bench_dwg.cuh:19-28), 32 bytes.                       */
typedef struct {
    int   length, inputTapPos, outputTapPos, writePos;
    float gain, reflection, damping, padding;
} gab_waveguide_state;

#define GAB_DWG_NAIVE 0   /* DWG1DNaiveKernel (cuda/bench_dwg.cu:10-59)       */
#define GAB_DWG_ACCEL 1   /* DWG1DAccelKernel (cuda/bench_dwg.cu:61-141)      */
/* Workspace the ordered (atomic-free) output reduction needs, in bytes: the tap contributions [n][bufsize], then
 * (since round 3) where every line reaches its tap, the per-sample hit counters and the hit lists.  gab_dwg takes no
 * size argument and writes ALL of these on every call: d_workspace MUST hold at least what this function returns
 * for the same (n_waveguides, bufsize) — a buffer sized n*bufsize floats by an older reading of this header is too
 * small and would be written past its end.                                                                */
size_t gab_dwg_workspace_bytes(int n_waveguides, int bufsize);
/* Updates the two delay-line banks (n_wg x max_len) in place and writes the
 * mono mix d_out[bufsize], summed over waveguides in index order.
 * GAB_DWG_ACCEL with 2048 or more mixed waveguides and bufsize <= 2048 keeps its per-sample hit counters in one of 16
 * slots of the library (not in d_workspace: they must be zero before the call's first append), picked round robin:
 * at most 16 such calls may be in flight at once per device (calls on one stream never are).                     */
int gab_dwg(const gab_waveguide_state* d_wg, float* d_fwd, float* d_bwd,
            const float* d_in, float* d_out, void* d_workspace, int n_waveguides,
            int bufsize, int max_len, int out_tracks, int variant,
            gab_stream_t stream);

/* cufftExecR2C, N=1024, batch = tracks (cuda/bench_fft.cu:63,105):
 * d_in tracks x 1024 real, d_out tracks x 513 interleaved complex.           */
int gab_fft_r2c_1024(const float* d_in, float* d_out, int tracks,
                     gab_stream_t stream);

/* ===================================================================== */
/* P. plans                                                              */
/* ===================================================================== */

/* ---- FFT convolution (Conv1DAccelBenchmark, cuda/bench_conv1d_accel.cu) -- */
typedef struct gab_conv_plan gab_conv_plan;

#define GAB_CONV_STATELESS 0  /* reference semantics: zero history each call  */
#define GAB_CONV_STREAMING 1  /* overlap-save with carried history            */
/* How a streaming plan cuts the taps.  Default SPLIT where the shape allows (512-sample buffers,
 * 1025..4096 taps, channel count divisible by 4), else CLASSIC; gab_conv_set_scheme changes it on a
 * fresh plan (before the first buffer or right after a reset).  A plan keeps its cut until then:
 * gab_conv_process (device or pinned host buffers) and gab_conv_process_batch all launch that cut.
 *   CLASSIC  taps [0,512) + [512,4096), both transforms in one workgroup per channel pair;
 *   SPLIT    taps [0,512) + [512,1024) + [1024,4096): the far partition runs for a pair every
 *            other buffer, one buffer ahead.  Same convolution, different rounding: results agree
 *            to ~1e-7 of the peak, not bit for bit.
 * Other power-of-two buffer sizes (32..2048) and responses up to 16384 taps run the fused
 * uniform-partition kernel (one cut, no choice); anything else the direct-form last resort.
 * gab_conv_set_ir on a plan that is mid-stream takes effect with the next buffer; on the SPLIT cut
 * the far shares already parked for the next two buffers were made with the previous taps.       */
#define GAB_CONV_SCHEME_CLASSIC 0
#define GAB_CONV_SCHEME_SPLIT 1
int gab_conv_set_scheme(gab_conv_plan* plan, int scheme);
int gab_conv_get_scheme(const gab_conv_plan* plan, int* scheme);
#define GAB_CONV_STREAMING_HOST_IO 2  /* the same, d_in / d_out in pinned host memory: identical kernel
                                       * under its own name, so that link-speed launches do not
                                       * mix into per-kernel profiles of the HBM-resident ones    */

/* allocateAccelBuffers + setupFFTPlans (:88-150).  Allocates the spectra bank
 * and the history ring on the current device.                                */
int gab_conv_create(gab_conv_plan** plan, int tracks, int bufsize, int ir_len);
int gab_conv_destroy(gab_conv_plan* plan);
/* precomputeImpulseResponseFFTs (:175-228): d_ir is tracks x ir_len floats on
 * the device.  Synchronous with respect to `stream`.                         */
int gab_conv_set_ir(gab_conv_plan* plan, const float* d_ir, gab_stream_t stream);
/* Forget all history (the state a freshly created plan has).                 */
int gab_conv_reset(gab_conv_plan* plan, gab_stream_t stream);
/* One buffer: d_in track-major T x B, d_out sample-major [T*s+t]
 * (performBenchmarkIteration :258-304 without the host copies).  d_in / d_out
 * must be device-ACCESSIBLE: device memory, or pinned host memory
 * (hipHostMalloc), in which case the kernel moves the buffer over PCIe itself
 * (zero-copy round trip: 94 us against 117 us with copy commands at C3).      */
int gab_conv_process(gab_conv_plan* plan, const float* d_in, float* d_out,
                     int mode, gab_stream_t stream);
/* n_buffers consecutive buffers in ONE launch (streaming mode): d_in = [n][T*B]
 * track-major buffers back to back, d_out = [n][B*T].  Same results, bit for bit, as n calls of
 * gab_conv_process; for callers whose input is resident ahead of time (offline rendering, and
 * bench.py's throughput figure): no kernel boundary between buffers.  On the split cut a 512-thread
 * workgroup owns a duo of channel pairs for the whole launch, near role on four waves, far role on
 * the other four (conv_split_batch_kernel).  A call of more than 256 buffers goes out as launches of at most 256 on
 * `stream` (a launch boundary keeps the workgroups in step: over many hundred buffers they drift apart and the output
 * lines leave the L2s in pieces — 5.63 against 5.10 us per buffer at 2048; profiles/r05_batch_buffers_per_launch.txt).
 * Additive: the reference processes one buffer per iteration.                                                      */
int gab_conv_process_batch(gab_conv_plan* plan, const float* d_in, float* d_out,
                           int n_buffers, gab_stream_t stream);
/* One buffer from pinned host memory to pinned host memory, returning when h_out holds the result — the
 * reference's whole iteration (transferToDevice, the pipeline, transferToHost: cuda/bench_base.cu:30-42 +
 * bench_conv1d_accel.cu:258-304) with both link directions busy at once.  Streaming mode, same state and same
 * bits as gab_conv_process on device buffers.  On a CLASSIC-cut plan (512-sample buffers, 513..4096 taps,
 * channel count divisible by 4) the upload is one engine copy into a staging buffer that the kernel — launched
 * at once, its history-only partition first — consumes as it lands, and the outputs go back in channel groups
 * while later groups are still arriving (conv_round_trip_kernel).  Other plans: the kernel moves both buffers
 * over the link itself (as GAB_CONV_STREAMING_HOST_IO; h_in must then be pinned as well) and the call waits
 * for the stream.  h_out must be pinned (hipHostMalloc) — the kernel writes it.  h_in: pinned for the overlap; pageable
 * memory is accepted and uploaded completely before the launch (the kernel consumes an upload as it lands only when every
 * word is written exactly once and in one piece, which engine copies from pinned memory do when no engine packet ends
 * inside a word: the upload goes out in pieces of 4 MiB - 256 bytes; and a staging word counts as landed when its top
 * byte is no longer the sentinel's 0xff, so input words that are negative NaNs, -inf or below -1.7e38 wait for the
 * upload's completion instead: slower, same bits).  h_in is read from the moment of the
 * call on the plan's own upload stream: it must be complete by then (the upload is NOT ordered behind work queued on
 * `stream`).  Blocking; one call at a time per plan.  The call returns when the LAUNCH HAS ENDED on `stream`
 * (the launch's own stop event has completed): from then on h_out is the host's and the staging buffer the next call's — the completion
 * rule of cuda/bench_base.cu:30-42,177-179 (copy back after a device synchronisation), not a word the kernel writes.
 * GAB_ERR_RUNTIME if the input never arrived: the output of that call is then invalid AND so is the
 * plan's carried history (the kernel took placeholders for samples) — gab_conv_reset before the stream goes on;
 * the staging buffer has been re-armed, the next call works.  A launch that could not be made leaves the plan as it
 * was (history, epoch, staging buffer).                                                                      */
int gab_conv_round_trip(gab_conv_plan* plan, const float* h_in, float* h_out, gab_stream_t stream);
/* The overlapped round trip's kernel takes input words while the upload is still running: that rests on engine writes landing
 * whole and once (an observation; a violation was silent wrong audio once: profiles/r05_incident_torn_word.txt).  Since round 6
 * a second, small launch behind every call — on the same stream, ordered behind the UPLOAD'S COMPLETION EVENT — compares the
 * words the kernel consumed (the plan's newest history block) with what the completed upload left in the staging buffer and
 * only then re-arms the buffer.  It costs the call nothing: the call returns on the main launch's end, as before.
 *   set_check(plan, 1)  (default) the verdict is read by a FOLLOWING gab_conv_round_trip on the plan — the next one when the
 *                       check launch is through by then (one call per audio slot), the one after it for back-to-back calls
 *                       (the plan has two staging buffers, taken in turn, so that no call waits for the previous call's check) —
 *                       which then returns GAB_ERR_RUNTIME: an EARLIER buffer's output was wrong, gab_conv_reset before the
 *                       stream goes on; or by gab_conv_round_trip_check (after the last buffer of a stream);
 *   set_check(plan, 2)  the call itself waits for the verdict (15-20 us more per call: the upload's completion event goes
 *                       through the command processor) and fails AT the call;
 *   set_check(plan, 0)  the verdict is ignored (the check launch still re-arms the buffer).
 * (The check cannot run INSIDE the kernel for free: the earliest "the upload is complete" that reaches a running kernel
 * arrives 14-20 us after the last byte — measured both ways, profiles/r06_roundtrip_check.txt.)                              */
int gab_conv_round_trip_check(gab_conv_plan* plan);
int gab_conv_round_trip_set_check(gab_conv_plan* plan, int mode);
/* Every later gab_conv_round_trip of this plan ends with a gab_keep_warm_kick (see keep-warm above); on = 0 stops kicking. */
int gab_conv_round_trip_keep_warm(gab_conv_plan* plan, int on);
/* gab_keep_warm_placement of the plan's own keep-warm launch (started = 0 if the plan has none). */
int gab_conv_round_trip_keep_warm_placement(gab_conv_plan* plan, unsigned* hw_id, unsigned* xcc_id, int capacity, int* started);
/* The block the plan consumed LAST, as its kernels keep it (the newest slot of the history ring of a 512-sample
 * plan), written to d_out in the input's layout [tracks][512].  An inspection call (additive): after
 * gab_conv_round_trip it must equal that call's h_in word for word — the check of the upload hand-off that tests
 * and tools/roundtrip_stress.py make.                                                                         */
int gab_conv_newest_block(gab_conv_plan* plan, float* d_out, gab_stream_t stream);
/* ---- a resident engine fed through a doorbell (additive; split-cut plans) -----------------------------------------
 * For a caller whose buffers ARRIVE one at a time but who can keep a couple in flight: ONE launch (the batch launch's
 * kernel) stays on the device and convolves buffer k as soon as the host — or anything that can write the slot — has
 * published it, so no kernel boundary separates buffers (conv_split_kernel: 8.9 us per buffer; the engine: the batch
 * rate).  Same state, same bits as gab_conv_process.
 *   rings     allocates (once per ring size) input / output rings of ring_buffers slots ([slot][T*B] track-major in,
 *             [slot][B*T] sample-major out; ordinary device memory that the launch reads with system-scope loads and
 *             writes with write-through stores: COPY ENGINES may write and read them while the launch runs — kernels
 *             cannot: at 1024 channels the engine holds every compute unit until it stops);
 *   start     the same rings, and launches BEHIND what `stream` holds at the call, on a stream of the plan's own at the
 *             highest priority (the runtime maps streams onto a few hardware queues per priority; work that shared a queue with
 *             the resident launch would stand behind it until stop — at normal priority a copy on another stream did);
 *   publish   after buffer k has been written to slot k % ring_buffers: the doorbell count goes up by n_more;
 *   submit    publish with a rung: flush != 0 says "finish what is published, do not wait for more" — the real-time form,
 *             ONE buffer in flight: a period then runs buffer k although k + 1 is not there (it requests nothing for it),
 *             a drain period delivers it and its count is reported at once; the engine idles until the doorbell moves and
 *             takes the next buffer cold (cuda/bench_conv1d_accel.cu:258-304: one buffer per iteration).  With two or more
 *             buffers pending the engine pipelines as before, whatever the rung says;
 *   completed buffers whose output is complete in its slot.  Pipelined (no flush): the engine asks for a buffer one period
 *             before it uses it, delivers one period after, counts a period later and passes the doorbell on inside the
 *             device, so buffer k is reported once k + 5 is published (or the flush / stop rung): keep at least six in
 *             flight, and a ring of at least seven slots.  A producer reuses slot k % ring only when completed > k - ring;
 *   wait      spins until completed >= count (GAB_ERR_RUNTIME after timeout_seconds, or if the engine gave up);
 *   feed_one_in_flight   the real-time loop for resident rings: n_buffers times { submit(1, flush); wait for that buffer },
 *             the host-clock time of each into latency_us[i] (may be null);
 *   feed      a host loop for resident rings: rings the doorbell n_buffers times, one buffer each, never more than
 *             `ahead` (6 <= ahead < ring_buffers) in front of `completed`;
 *   stop      rings the stop bit, waits for the launch to end (every published buffer is finished), carries the
 *             history on for the next gab_conv_process / batch / engine;
 *   round_trip   the reference's iteration through the engine (cuda/bench_base.cu:30-42 around bench_conv1d_accel.cu:258-304),
 *             ONE buffer in flight: engine copy of h_in (pinned host, [T*B]) into the next ring slot, submit(1, flush), wait for
 *             that buffer, engine copy of its slot into h_out (pinned host, [B*T]).  Same bits as gab_conv_process.  The two link
 *             legs do not overlap with the transform — gab_conv_round_trip's do, and it is the faster round trip; this entry makes
 *             the per-buffer engine a complete replacement of that iteration.  Nothing else may be in flight;
 *   set_idle_limit   how long a stalled engine waits for the doorbell to move before it ends by itself (default 4 s; 0.5 .. 3600;
 *             taken at the next start).  A real-time caller that may pause for longer than that between buffers raises it.
 * The device is the engine's while it runs (256 workgroups at 1024 channels): other kernels queue behind it — and with
 * FEWER channels a kernel on another stream may still wait until stop: the runtime maps streams onto a few hardware
 * queues, and a kernel (a device-to-device copy is one) that lands on the engine's queue stands behind the resident
 * launch.  Only copy ENGINES (pinned host <-> device copies) are sure to move the rings while the launch runs.  Every
 * workgroup of the engine must be resident at once: start refuses a plan with more channels than 4 x the workgroups the
 * device holds (1024 channels on MI355X; more channels: one engine per device over channel shards).  If the
 * doorbell does not move for the idle limit (4 s unless set) the launch ends by itself and stop / feed / wait return
 * GAB_ERR_RUNTIME; stop then carries the plan's history on from what the engine CONSUMED (its message says how many of the
 * published buffers that is): a pipelined burst without the flush rung leaves its last buffer unconsumed — publish it again
 * after the next start.  start returns GAB_ERR_INVALID_ARG while a keep-warm launch other than the plan's own is resident on
 * the device (see keep-warm above).  A wait that runs out says in gab_last_error whether the launch ever became resident
 * (never started / only some workgroups / all of them), so a hardware-queue collision or a crowded device names itself.       */
int gab_conv_engine_rings(gab_conv_plan* plan, int ring_buffers, float** d_in_ring, float** d_out_ring);
int gab_conv_engine_start(gab_conv_plan* plan, int ring_buffers, float** d_in_ring, float** d_out_ring, gab_stream_t stream);
int gab_conv_engine_publish(gab_conv_plan* plan, int n_more);
int gab_conv_engine_submit(gab_conv_plan* plan, int n_more, int flush);
int gab_conv_engine_wait(gab_conv_plan* plan, int count, double timeout_seconds);
/* 1 while the resident launch is still on the device (it ends by itself if the doorbell stops moving; stop still has to be called) */
int gab_conv_engine_running(gab_conv_plan* plan, int* running);
int gab_conv_engine_completed(gab_conv_plan* plan, int* completed);
int gab_conv_engine_feed(gab_conv_plan* plan, int n_buffers, int ahead);
int gab_conv_engine_feed_one_in_flight(gab_conv_plan* plan, int n_buffers, float* latency_us);
int gab_conv_engine_stop(gab_conv_plan* plan);
int gab_conv_engine_round_trip(gab_conv_plan* plan, const float* h_in, float* h_out);
int gab_conv_engine_set_idle_limit(gab_conv_plan* plan, double seconds);
/* Bytes of device state the plan holds: spectra, history.                    */
int gab_conv_state_bytes(const gab_conv_plan* plan, size_t* spectra_bytes,
                         size_t* history_bytes);

/* ---- FDTD3D (FDTD3DBenchmark, cuda/bench_fdtd3d.cu) ---------------------- */
typedef struct gab_fdtd_plan gab_fdtd_plan;

/* FDTD3DParams (cuda/bench_fdtd3d.cuh:68-86), the fields the kernels read.   */
typedef struct {
    int   nx, ny, nz;
    int   source_x, source_y, source_z;
    int   receiver_x, receiver_y, receiver_z;
    int   steps_per_sample;
    float dt_over_rho_dx, rho_c2_dt_over_dx, absorption_coeff;
} gab_fdtd_params;

/* Reference constants (bench_fdtd3d.cuh:12-41) for an nx*ny*nz grid; source
 * and receiver scale with the room so 52^3 gives (25,25,5)/(40,15,25).       */
int gab_fdtd_default_params(int nx, int ny, int nz, gab_fdtd_params* out);
int gab_fdtd_create(gab_fdtd_plan** plan, const gab_fdtd_params* params);
int gab_fdtd_destroy(gab_fdtd_plan* plan);
int gab_fdtd_reset(gab_fdtd_plan* plan, gab_stream_t stream);     /* zero grids */
/* runFDTD3DTimeStep (:384-438) for samples [first_sample, first_sample+n):
 * inject -> steps_per_sample x {velocity, pressure} -> extract.
 * d_in/d_out are track-major T x B.                                          */
int gab_fdtd_process(gab_fdtd_plan* plan, const float* d_in, float* d_out,
                     int tracks, int bufsize, int first_sample, int n_samples,
                     gab_stream_t stream);
/* Which form gab_fdtd_process takes.  A room whose four fields fit the chip's LDS (nx <= 128;
 * up to 8192 cells per compute unit: 128^3 on 256 CUs) runs a whole buffer in ONE launch with the fields
 * resident in LDS and registers, one block of rows per workgroup, the blocks' boundary pressures handed to the
 * neighbours through memory every step (same bits as the step kernels, 4x their speed at 128^3).  It needs
 * every workgroup on the device at once (resident launches of one process are chained per device, whatever
 * their streams; other processes' kernels are not known; plan creation checks that the device can hold the whole grid at
 * once): a workgroup that waits about a second for a neighbour gives up, that call's output is NaN, gab_fdtd_status (or
 * the next call on the plan) returns GAB_ERR_RUNTIME and the plan uses the step kernels from then on.  Larger rooms, z-slabs,
 * per-track positions and calls inside a stream capture use the step kernels.
 * *resident = 1 when the next call (outside a capture) takes the resident form, *workgroups = its grid. */
int gab_fdtd_resident(const gab_fdtd_plan* plan, int* resident, int* workgroups);
/* A host that shares the device with other work can decline the resident form: STEP = one launch per step (per
 * sample for rooms up to 56^3), no workgroup ever waits for another; AUTO (default) = resident where it fits.  Same
 * bits either way.  Takes effect with the next gab_fdtd_process.                                          */
#define GAB_FDTD_FORM_AUTO 0
#define GAB_FDTD_FORM_STEP 1
int gab_fdtd_set_form(gab_fdtd_plan* plan, int form);
/* Errors at the call that failed (the reference throws from the failing iteration: synchronizeAndCheck,
 * cuda/bench_base.cu:177-179).  Synchronises `stream` and returns GAB_ERR_RUNTIME if the resident launch of the
 * last gab_fdtd_process on it gave up waiting for a neighbour workgroup.  That call's output is NaN in every
 * sample (never plausible audio); the plan then takes the step kernels and wants a gab_fdtd_reset.  Without this
 * call the same error is returned by the NEXT gab_fdtd_process, gab_fdtd_reset or gab_fdtd_destroy.        */
int gab_fdtd_status(gab_fdtd_plan* plan, gab_stream_t stream);
/* Track-dependent source and receiver cells — announced and never done by the Metal port
 * ("can be made track-dependent later", kernels_fdtd3d.metal:184,217).  src_xyz / rcv_xyz: HOST
 * arrays, tracks x (x, y, z).  From then on gab_fdtd_process (with that many tracks) adds
 * 0.1*in[t,s] into track t's source cell — tracks in order, so a cell shared by several tracks
 * receives their samples in track order — and writes out[t,s] = 0.1*p[receiver cell of t].
 * tracks = 0 returns to the shared cells of the params.                                          */
int gab_fdtd_set_track_positions(gab_fdtd_plan* plan, const int* src_xyz, const int* rcv_xyz, int tracks);
/* ---- z-slab domain decomposition (Metal comments kernels_fdtd3d.metal:184,217) ----
 * A plan that owns planes [z_begin, z_end) of the global grid, with one ghost pressure plane below
 * and above and one ghost vz face plane above.  One leapfrog step needs, from the slab below, its
 * top pressure plane, and from the slab above, its bottom pressure plane and bottom vz faces:
 *   per step:  gab_fdtd_step on every slab  ->  exchange the planes gab_fdtd_halo names
 *   per sample: gab_fdtd_inject (owner of the source acts, others return at once), steps_per_sample
 *   steps, the last one with strip_sample = the sample index (owner of the receiver stores
 *   0.1 * p[receiver] into its strip).
 * gab_fdtd_process needs the whole grid in one plan.                                              */
int gab_fdtd_create_slab(gab_fdtd_plan** plan, const gab_fdtd_params* params, int z_begin, int z_end);
int gab_fdtd_owns(const gab_fdtd_plan* plan, int* owns_source, int* owns_receiver);
/* inj[s] = sum over tracks of 0.1 * in[t, s], in track order, for the whole buffer */
int gab_fdtd_source_sums(gab_fdtd_plan* plan, const float* d_in, int tracks, int bufsize, gab_stream_t stream);
int gab_fdtd_inject(gab_fdtd_plan* plan, int sample, gab_stream_t stream);
int gab_fdtd_step(gab_fdtd_plan* plan, int strip_sample /* -1: none */, gab_stream_t stream);
#define GAB_FDTD_SEND_DOWN_P  0   /* my plane z_begin      -> lower slab's GAB_FDTD_RECV_UP_P   */
#define GAB_FDTD_SEND_DOWN_VZ 1   /* my faces z_begin      -> lower slab's GAB_FDTD_RECV_UP_VZ  */
#define GAB_FDTD_SEND_UP_P    2   /* my plane z_end-1      -> upper slab's GAB_FDTD_RECV_DOWN_P */
#define GAB_FDTD_RECV_DOWN_P  3
#define GAB_FDTD_RECV_UP_P    4
#define GAB_FDTD_RECV_UP_VZ   5
/* device pointer and length (nx*ny floats) of a halo plane of the CURRENT fields */
int gab_fdtd_halo(gab_fdtd_plan* plan, int which, float** d_ptr, size_t* n_floats);
/* out[t*bufsize + s] = strip[s] for every track: the receiver's owner writes the buffer's output */
int gab_fdtd_emit(gab_fdtd_plan* plan, float* d_out, int tracks, int bufsize, gab_stream_t stream);
/* the per-sample receiver values this slab recorded (meaningful on the receiver's owner) */
int gab_fdtd_strip(gab_fdtd_plan* plan, float** d_strip, int* capacity);
/* Copy the plan's own pressure planes (nx*ny*(z_end-z_begin) floats, x fastest) to d_dst. */
int gab_fdtd_copy_pressure(gab_fdtd_plan* plan, float* d_dst, gab_stream_t stream);

/* ===================================================================== */
/* G. host-side data generators of the harness                           */
/* ===================================================================== */

/* generateRandomAudioData (cuda/bench_utils.cu:238-245): mt19937(seed), U(-1,1). */
int gab_generate_noise(float* h_buf, size_t n, unsigned seed);
/* Conv1DBenchmark::generateImpulseResponses (cuda/bench_conv1d.cu:159-181) and
 * Conv1DAccelBenchmark::generateImpulseResponses (cuda/bench_conv1d_accel.cu:
 * 152-173).  The formulas use the GLOBAL track index and count: a shard asks
 * for tracks [track_offset, track_offset+n_tracks) of a bank of total_tracks.  */
int gab_generate_conv1d_ir(float* h_ir, int ir_len, size_t track_offset,
                           size_t n_tracks, size_t total_tracks);
int gab_generate_conv_accel_ir(float* h_ir, int ir_len, size_t track_offset,
                               size_t n_tracks, size_t total_tracks);

/* The stream glibc's srand(seed) + rand() gives (random_r.c TYPE_3), from its `skip`-th value on, as the harness draws it
 * for FFT1D's input and RndMemRead's pool and playheads (private generators: no other rand() user or thread moves
 * them, and a channel shard enters them at its first track).  Host arithmetic; additive.                          */
int gab_glibc_rand(unsigned seed, unsigned long long skip, int* out, size_t n);

/* Channel shards of a multi-GPU job (additive; BASELINE configs[4]): the contiguous range
 * [*lo, *hi) of `rank` out of `world`, the remainder going to the low ranks.  Host arithmetic.  */
int gab_shard_range(int rank, int world, size_t total_tracks, size_t* lo, size_t* hi);
/* The same cut at multiples of `granule` tracks, and the granule a benchmark (registry name) needs: FFT1D packs two
 * tracks into one complex transform (2), Conv1D_accel four channels into one workgroup's (4) — their bits depend on
 * which tracks share a transform, so their shards keep those groups whole; every other benchmark 1.            */
int gab_shard_range_aligned(int rank, int world, size_t total_tracks, size_t granule, size_t* lo, size_t* hi);
size_t gab_shard_granule(const char* benchmark);

/* calculateStatistics (cuda/bench_utils.cu:358-414): mean, median, sample
 * std-dev, min, max, linearly interpolated p95/p99.                          */
typedef struct {
    float  mean, median, std_dev, min_val, max_val, p95, p99;
    size_t count;
} gab_statistics;
int gab_calculate_statistics(const float* latencies, size_t n, gab_statistics* out);
/* The CLI-backed process globals the legacy writers read (cuda/globals.cu:4-7). */
int gab_set_globals(int fs, int buffer_size, int n_tracks, int n_runs);
/* generateJSONResults (cuda/globals.cu:137-182) into buf (NUL-terminated);
 * returns the length the full text needs, excluding the NUL.                 */
size_t gab_format_json_results(const float* latencies, size_t n, const char* name,
                               char* buf, size_t capacity);
/* writeCSVResults (cuda/globals.cu:69-122): appends one row, header if new.  */
int gab_write_csv_results(const float* latencies, size_t n, const char* name,
                          const char* filename);

/* ===================================================================== */
/* H. harness (GPUABenchmark by registry name)                           */
/* ===================================================================== */

typedef struct gab_bench gab_bench;

typedef struct {
    int    fs, buffer_size, n_tracks, n_runs;       /* cuda/globals.cu:4-7     */
    int    ir_length;       /* <=0: the benchmark's DEFAULT_IR_LEN             */
    int    fdtd_grid;       /* <=0: 52 (bench_fdtd3d.cuh:36-38)                */
    int    conv_mode;       /* GAB_CONV_STATELESS | GAB_CONV_STREAMING | 2: streaming with every
                               iteration ONE gab_conv_round_trip call (--convMode roundtrip)   */
    int    quiet;           /* suppress the reference's progress printf        */
    int    modal_mode;      /* ModalFilterBank: 0 the CUDA port's placeholder
                               (bench_modal.cu:15-36), 1 the real bank (Metal port) */
    int    conv_batch;      /* Conv1D_accel: <=1 one buffer per iteration with its copies (the
                               reference); n: n HBM-resident buffers per iteration in ONE
                               gab_conv_process_batch launch (throughput mode)           */
    int    fdtd_form;       /* FDTD3D: GAB_FDTD_FORM_AUTO (0) | GAB_FDTD_FORM_STEP (1)              */
    int    datacopy_mode;   /* datacopy*: 0 upload and download at once (gab_datatransfer_round_trip)
                               | 1 H2D -> kernel -> D2H one after the other (the reference's schedule) */
} gab_bench_config;

typedef struct {
    int    iterations;
    float  mean_ms, median_ms, std_dev_ms, min_ms, max_ms, p95_ms, p99_ms;
    float  gpu_median_ms;                    /* 0 when the benchmark records none */
    double throughput_gbps, samples_per_sec; /* cuda/bench_base.cu:110-115     */
    size_t bytes_processed;
} gab_bench_result;

typedef struct {
    int   status;            /* 0 SUCCESS, 1 FAILURE, -1 FATAL (bench_base.cuh:36-40) */
    float max_error, mean_error;
} gab_bench_validation;

void gab_bench_default_config(gab_bench_config* cfg);
int  gab_bench_count(void);
const char* gab_bench_name(int index);                 /* cuda/main.cu:84-100 */
int  gab_bench_create(gab_bench** b, const char* name, const gab_bench_config* cfg);
int  gab_bench_destroy(gab_bench* b);
/* Channel shard (additive; BASELINE configs[4]): the benchmark — created with n_tracks = the shard's
 * own track count — computes tracks [first_track, first_track + n_tracks) of a job of total_tracks: its inputs,
 * impulse responses, playheads and goldens are the global job's rows, so that the shards' results side by side
 * are the unsharded results bit for bit.  Before gab_bench_setup.  GAB_ERR_INVALID_ARG for benchmarks whose
 * tracks are not independent (DWG, modal, FDTD3D: replicas only).                                        */
int  gab_bench_set_shard(gab_bench* b, size_t first_track, size_t total_tracks);
/* What the last iteration left on the host: result arrays by index (0 .. count-1).  layout 0: track-major rows of
 * per_track floats (shards concatenate by rows), 1: sample-major [per_track][tracks] (shards concatenate by
 * columns).  The pointers stay valid until the next iteration / destroy.                                */
int  gab_bench_result_count(gab_bench* b);
int  gab_bench_result_array(gab_bench* b, int index, const char** name, const float** data, size_t* count,
                            int* layout, size_t* per_track);
int  gab_bench_setup(gab_bench* b);
int  gab_bench_run(gab_bench* b, int iterations, int warmup, gab_bench_result* out);
int  gab_bench_validate(gab_bench* b, gab_bench_validation* out);
/* text of the validation messages of the last gab_bench_validate, '\n'-joined */
const char* gab_bench_validation_text(gab_bench* b);
/* roofline numerator of one iteration (GPUABenchmark::algorithmicBytes)      */
int  gab_bench_algorithmic_bytes(gab_bench* b, size_t* bytes);
/* latencies of the last run (ms); returns how many were copied               */
int  gab_bench_latencies(gab_bench* b, float* out, int capacity);

/* ---- DAW-style pacing --------------------------------------------------------
 * The CUDA reference declares the knobs only (cuda/globals.cuh:27-30,
 * cuda/bench_utils.cuh:57-58,95 "--dawsim"); the scheduler is its Metal port's
 * DAWSimulator (metal-swift/MetalSwiftBench/Core/BenchmarkUtilities.swift:140-178,
 * used by Core/GPUABenchmark.swift:358-392): iteration k+1 starts no earlier
 * than t0 + (k+1)*buffer_seconds (+/- jitter).  mode: 0 spin, 1 sleep.          */
typedef struct gab_dawsim gab_dawsim;
int gab_dawsim_create(gab_dawsim** s, double buffer_seconds, int mode, double jitter_seconds);
int gab_dawsim_wait(gab_dawsim* s);
int gab_dawsim_stats(const gab_dawsim* s, unsigned long long* waits, unsigned long long* missed_slots);
int gab_dawsim_destroy(gab_dawsim* s);
/* pace every warm-up and timed iteration of gab_bench_run (enable = 0 turns it off) */
int gab_bench_set_dawsim(gab_bench* b, int enable, double buffer_seconds, int mode, double jitter_seconds);
/* leave a gab_keep_warm launch (8 workgroups; idle limit four pacing slots, at least 0.05 s) on the device for the length of every gab_bench_run and kick it after
 * every iteration: with pacing on, the device does not go idle while the loop waits for the next slot (enable = 0: off) */
int gab_bench_set_keep_warm(gab_bench* b, int enable);
/* pacing counters of the last gab_bench_run */
int gab_bench_dawsim_stats(gab_bench* b, unsigned long long* waits, unsigned long long* missed_slots);

#ifdef __cplusplus
}
#endif
#endif /* GAB_C_API_H */
