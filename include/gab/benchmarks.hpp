// benchmarks.hpp — the benchmark classes behind the 17 registry names.
//
// One GPUABenchmark subclass per reference benchmark, with the reference's
// class names, constructor signatures, public constants and device-struct
// layouts (cuda/bench_<name>.cuh), so main.cu-style factories compile against
// them.  Constructor parameters that did not exist in the reference are
// trailing and defaulted.  Each class computes its CPU golden in
// setupBenchmark() exactly where the reference does and compares against it in
// validate(); the golden functions live in gab::golden (validation only — no
// device result is ever produced by them).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "bench_base.hpp"
#include "benchmark_constants.hpp"
#include "thread_config.hpp"
#include "../gab_c_api.h"

// ---------------------------------------------------------------------------
// NoOp — launch-overhead copy (cuda/bench_noop.cuh)
// ---------------------------------------------------------------------------
class NoOpBenchmark : public GPUABenchmark {
public:
    NoOpBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~NoOpBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    const float* cpuReference() const { return cpu_reference; }

private:
    float* cpu_reference = nullptr;
};

// ---------------------------------------------------------------------------
// Gain (cuda/bench_gain.cuh)
// ---------------------------------------------------------------------------
class GainBenchmark : public GPUABenchmark {
public:
    explicit GainBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS,
                           bool enable_validation = true);
    ~GainBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    bool shardable() const override { return true; }
    const float* cpuReference() const { return cpu_reference; }

private:
    float* cpu_reference = nullptr;
    bool enable_validation_;
};

// ---------------------------------------------------------------------------
// GainStats (cuda/bench_gainstats.cuh)
// ---------------------------------------------------------------------------
class GainStatsBenchmark : public GPUABenchmark {
public:
    static const int NSTATS = 2;   // mean, max
    GainStatsBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~GainStatsBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    size_t algorithmicBytes() const override;
    bool shardable() const override { return true; }
    std::vector<ResultArray> resultArrays() const override;
    const float* hostStats() const { return h_stats; }
    const float* cpuStatsReference() const { return cpu_stats_reference; }

private:
    float* h_stats = nullptr;
    float* d_stats = nullptr;
    float* cpu_reference = nullptr;
    float* cpu_stats_reference = nullptr;
    size_t stats_count;
    size_t stats_size_bytes;
};

// ---------------------------------------------------------------------------
// DataTransfer (cuda/bench_datatransfer.cuh)
// ---------------------------------------------------------------------------
class DataTransferBenchmark : public GPUABenchmark {
public:
    struct Config {
        float inputRatio;
        float outputRatio;
        const char* name;
    };
    static const Config CONFIGS[];
    static const int NUM_CONFIGS;
    static constexpr int BASE_BUFFER_SIZE = (10 * 1024 * 1024 / sizeof(float));

    DataTransferBenchmark(const Config& config);
    DataTransferBenchmark(float input_ratio, float output_ratio, const char* name = "DataTransfer");
    ~DataTransferBenchmark() override;
    static DataTransferBenchmark* createFromName(const std::string& name);

    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenWhole() override;
    size_t algorithmicBytes() const override;
    int inputSize() const { return input_size; }
    int outputSize() const { return output_size; }
    std::vector<ResultArray> resultArrays() const override;

private:
    Config config_;
    float* h_input_var = nullptr;
    float* h_output_var = nullptr;
    float* d_input_var = nullptr;
    float* d_output_var = nullptr;
    float* cpu_reference = nullptr;
    struct gab_link_plan* link_plan_ = nullptr;   // DATACOPY_SEQUENTIAL == 0: gab_datatransfer_round_trip
    int input_size;
    int output_size;
    size_t input_size_bytes;
    size_t output_size_bytes;
};

// ---------------------------------------------------------------------------
// FFT1D (cuda/bench_fft.cuh) — cufftComplex becomes float2
// ---------------------------------------------------------------------------
class FFTBenchmark : public GPUABenchmark {
public:
    static const int FFT_SIZE = 1024;
    FFTBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~FFTBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    size_t algorithmicBytes() const override;
    bool shardable() const override { return true; }
    std::vector<ResultArray> resultArrays() const override;
    // max over bins of |dre|+|dim| against a float64 DFT of the same input,
    // for the product output and for the reference-style fp32 golden
    double truthErrorOfOutput() const { return err_out_vs_truth_; }
    double truthErrorOfGolden() const { return err_golden_vs_truth_; }

private:
    float* h_input_fft = nullptr;
    float2* h_output_fft = nullptr;
    float* d_input_fft = nullptr;
    float2* d_output_fft = nullptr;
    float* cpu_reference_real = nullptr;
    float* cpu_reference_imag = nullptr;
    size_t input_fft_size;
    size_t output_fft_size;
    size_t input_fft_bytes;
    size_t output_fft_bytes;
    double err_out_vs_truth_ = 0.0, err_golden_vs_truth_ = 0.0;
};

// ---------------------------------------------------------------------------
// IIRFilter (cuda/bench_iir.cuh)
// ---------------------------------------------------------------------------
struct IIRCoefficients {
    float b0, b1, b2;   // numerator
    float a1, a2;       // denominator (a0 = 1)
};

class IIRBenchmark : public GPUABenchmark {
public:
    static const int STATES_PER_TRACK = 2;   // z1, z2
    IIRBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~IIRBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    void resetState() override;
    size_t algorithmicBytes() const override;
    bool shardable() const override { return true; }
    std::vector<ResultArray> resultArrays() const override;
    const IIRCoefficients& coefficients() const { return *h_coeffs; }

private:
    IIRCoefficients calculateButterworthCoefficients(float normalized_frequency);
    IIRCoefficients* h_coeffs = nullptr;
    float* h_state = nullptr;
    float* d_state = nullptr;
    float* cpu_reference = nullptr;
    float* cpu_state_reference = nullptr;
    size_t state_count;
    size_t state_size_bytes;
};

// ---------------------------------------------------------------------------
// Conv1D — time-domain FIR (cuda/bench_conv1d.cuh); the IR bank lives in a
// plain device buffer staged through LDS (no textures on CDNA)
// ---------------------------------------------------------------------------
class Conv1DBenchmark : public GPUABenchmark {
public:
    static const int DEFAULT_IR_LEN = 1024;
    Conv1DBenchmark(int ir_length = (IR_LENGTH > 0 ? IR_LENGTH : DEFAULT_IR_LEN),
                    size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS);
    ~Conv1DBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    size_t algorithmicBytes() const override;
    bool shardable() const override { return true; }
    int irLength() const { return ir_length_; }
    size_t haloTracks() const { return halo_tracks_; }      // a shard's input rows in front of its own (0: whole job)

private:
    int ir_length_;
    float* h_ir_buf = nullptr;
    float* d_ir_buf = nullptr;
    float* cpu_reference = nullptr;
    float* h_halo_in_ = nullptr;      // shards: [halo tracks | own tracks] of the flat input
    float* d_halo_in_ = nullptr;
    size_t halo_tracks_ = 0;
    size_t ir_buffer_size;
    size_t ir_buffer_bytes;
};

// ---------------------------------------------------------------------------
// Conv1D_accel — FFT convolution (cuda/bench_conv1d_accel.cuh)
// ---------------------------------------------------------------------------
class Conv1DAccelBenchmark : public GPUABenchmark {
public:
    static const int DEFAULT_IR_LEN = 512;
    static const int DEFAULT_FFT_SIZE = 1024;
    enum class Mode { STATELESS = GAB_CONV_STATELESS, STREAMING = GAB_CONV_STREAMING };

    Conv1DAccelBenchmark(int ir_length = (IR_LENGTH > 0 ? IR_LENGTH : DEFAULT_IR_LEN),
                         size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS,
                         Mode mode = (CONV_STREAMING ? Mode::STREAMING : Mode::STATELESS),
                         size_t track_offset = 0, size_t total_tracks = 0);
    ~Conv1DAccelBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenSlice(size_t first_track, size_t count) override;
    void resetState() override;
    size_t algorithmicBytes() const override;
    bool shardable() const override { return true; }
    std::vector<ResultArray> resultArrays() const override;
    // Multi-GPU runs: the impulse responses of this shard already lie on this device (its rows of the
    // bank that was broadcast over RCCL); setupBenchmark() then transforms those instead of uploading
    // its own copy.  The golden still uses the host formula, so validate() cross-checks the bank.
    void shareImpulseResponses(const float* d_rows) { d_shared_ir_ = d_rows; }
    // Throughput mode (additive; bench.py's `value` through the harness): an iteration is ONE
    // gab_conv_process_batch launch over n consecutive buffers that are resident in HBM (generated once,
    // noise seeds 42, 43, ...; no per-iteration copies), history carried from one iteration to the next.
    // Streaming mode only; validate() checks the first buffer of a batch from reset against the golden.
    // Defaults to the CONV_BATCH global (--convBatch).
    void setBatch(int n_buffers) { batch_ = n_buffers > 1 ? n_buffers : 0; }
    int batch() const { return batch_; }
    int irLength() const { return ir_length_; }
    int fftSize() const { return fft_size_; }
    int overlapSize() const { return overlap_size_; }
    const float* cpuReference() const { return cpu_reference; }
    float peakNormalisedError() const { return peak_norm_error_; }

private:
    int ir_length_;
    int fft_size_;       // the reference's formula: next pow2 of L+B-1 (:52); informational
    int overlap_size_;   // L-1 (:53) — the history this build actually carries
    Mode mode_;
    size_t track_offset_, total_tracks_;
    bool round_trip_ = false;         // --convMode roundtrip: every iteration is one gab_conv_round_trip call
    float* h_ir_buf = nullptr;
    float* d_ir_buf = nullptr;
    float* cpu_reference = nullptr;
    gab_conv_plan* plan_ = nullptr;
    const float* d_shared_ir_ = nullptr;
    int batch_ = 0;
    float* d_batch_in_ = nullptr;
    float* d_batch_out_ = nullptr;
    size_t ir_buffer_size;
    size_t ir_buffer_bytes;
    float peak_norm_error_ = 0.0f;
};

// ---------------------------------------------------------------------------
// ModalFilterBank (cuda/bench_modal.cuh) — placeholder semantics of the CUDA port
// ---------------------------------------------------------------------------
class ModalBenchmark : public GPUABenchmark {
public:
    static const int NUM_MODES = 1024 * 1024;
    static const int NUM_MODE_PARAMS = 8;
    static const int MODAL_OUTPUT_TRACKS = 32;
    enum ModeParams { AMPLITUDE = 0, FREQUENCY = 1, PHASE = 2, STATE_REAL = 3, STATE_IMAG = 4,
                      RESERVED1 = 5, RESERVED2 = 6, RESERVED3 = 7 };
    // PLACEHOLDER: the CUDA port as it is (32 constants, 32 MiB parameter upload per iteration).
    // BANK: the bank the benchmark is named for — the reference's Metal kernel
    // (kernels_benchmark_staging.metal:121-162) on the same parameter records: min(1024*NTRACKS,
    // 2^20) modes onto min(NTRACKS, 32) tracks, parameters resident on the device.
    enum class Variant { PLACEHOLDER, BANK };
    explicit ModalBenchmark(Variant variant = MODAL_REAL ? Variant::BANK : Variant::PLACEHOLDER);
    ~ModalBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenWhole() override;
    size_t algorithmicBytes() const override;
    int modeCount() const { return num_modes_; }
    int outputTracks() const { return out_tracks_; }

private:
    Variant variant_;
    int num_modes_;
    int out_tracks_;
    float* h_mode_params = nullptr;
    float* d_mode_params = nullptr;
    float* h_modal_output = nullptr;
    float* d_modal_output = nullptr;
    float* d_workspace = nullptr;
    float* cpu_reference = nullptr;
    size_t mode_params_size;
    size_t mode_params_bytes;
    size_t modal_output_size;
    size_t modal_output_bytes;
};

// ---------------------------------------------------------------------------
// DWG1DNaive / DWG1DAccel (cuda/bench_dwg.cuh)
// ---------------------------------------------------------------------------
struct DWGParams {
    int numWaveguides;
    int bufferSize;
    int outputTracks;
    int minLength;
    int maxLength;
    float reflectionCoeff;
    float dampingCoeff;
};

struct WaveguideState {
    int length;
    int inputTapPos;
    int outputTapPos;
    int writePos;
    float gain;
    float reflection;
    float damping;
    float padding;   // 32 bytes
};

class DWGBenchmark : public GPUABenchmark {
public:
    enum class Variant { NAIVE, ACCELERATED };
    static const int DEFAULT_MIN_LENGTH = 100;
    static const int DEFAULT_MAX_LENGTH = 2000;
    static const float DEFAULT_REFLECTION_COEFF;
    static const float DEFAULT_DAMPING_COEFF;

    DWGBenchmark(Variant variant = Variant::NAIVE, size_t buffer_size = BUFSIZE,
                 size_t track_count = NTRACKS);
    ~DWGBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    void resetState() override;
    size_t algorithmicBytes() const override;
    // the output golden is identically zero; the delay-line state is
    // what carries information, so validate() compares it too
    float delayLineMaxError() const { return delay_max_error_; }

private:
    Variant variant_;
    WaveguideState* h_waveguide_params = nullptr;
    WaveguideState* d_waveguide_params = nullptr;
    DWGParams* h_dwg_params = nullptr;
    float* h_delay_forward = nullptr;
    float* d_delay_forward = nullptr;
    float* h_delay_backward = nullptr;
    float* d_delay_backward = nullptr;
    float* h_input_signal = nullptr;
    float* d_input_signal = nullptr;
    float* h_output_buffer = nullptr;
    float* d_output_buffer = nullptr;
    void* d_workspace = nullptr;
    float* cpu_reference = nullptr;
    float* cpu_delay_forward = nullptr;
    float* cpu_delay_backward = nullptr;
    size_t delay_line_size;
    size_t delay_line_bytes;
    size_t output_buffer_size;
    size_t output_buffer_bytes;
    float delay_max_error_ = 0.0f;
};

// ---------------------------------------------------------------------------
// FDTD3D (cuda/bench_fdtd3d.cuh)
// ---------------------------------------------------------------------------
constexpr int kFDTD3D_RoomX = 50;
constexpr int kFDTD3D_RoomY = 50;
constexpr int kFDTD3D_RoomZ = 50;
constexpr float kFDTD3D_SoundSpeed = 343.0f;
constexpr float kFDTD3D_SpatialStep = 0.01f;
constexpr float kFDTD3D_AirDensity = 1.225f;
constexpr float kFDTD3D_AbsorptionCoeff = 0.2f;
constexpr float kFDTD3D_CFLNumber = 0.5f;
constexpr int kFDTD3D_StepsPerSample = 3;
constexpr int kFDTD3D_SourceX = 25;
constexpr int kFDTD3D_SourceY = 25;
constexpr int kFDTD3D_SourceZ = 5;
constexpr int kFDTD3D_ReceiverX = 40;
constexpr int kFDTD3D_ReceiverY = 15;
constexpr int kFDTD3D_ReceiverZ = 25;
constexpr int kFDTD3D_GridNX = kFDTD3D_RoomX + 2;
constexpr int kFDTD3D_GridNY = kFDTD3D_RoomY + 2;
constexpr int kFDTD3D_GridNZ = kFDTD3D_RoomZ + 2;
constexpr float kFDTD3D_TimeStep =
    kFDTD3D_CFLNumber * kFDTD3D_SpatialStep / (kFDTD3D_SoundSpeed * 1.732050808f);

struct FDTD3DParams {
    int nx, ny, nz;
    float sound_speed;
    float spatial_step;
    float time_step;
    float air_density;
    float absorption_coeff;
    int source_x, source_y, source_z;
    int receiver_x, receiver_y, receiver_z;
    int buffer_size;
    int track_count;
    int steps_per_sample;
    float dt_over_rho_dx;
    float rho_c2_dt_over_dx;
};

class FDTD3DBenchmark : public GPUABenchmark {
public:
    FDTD3DBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS,
                    int grid = (FDTD_GRID > 0 ? FDTD_GRID : kFDTD3D_GridNX));
    ~FDTD3DBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    void resetState() override;
    size_t algorithmicBytes() const override;     // per buffer = per-step bytes * 3 * B
    bool workingSetOnChip() const override;
    const FDTD3DParams& params() const { return *h_fdtd3d_params; }
    // error against the real field evolution (gab::golden::fdtd3d), not the placeholder
    float fieldMaxError() const { return field_max_error_; }

private:
    int grid_;
    FDTD3DParams* h_fdtd3d_params = nullptr;
    gab_fdtd_plan* plan_ = nullptr;
    float* h_input_signal = nullptr;
    float* d_input_signal = nullptr;
    float* h_output_buffer = nullptr;
    float* d_output_buffer = nullptr;
    float* cpu_reference = nullptr;          // the reference's placeholder golden
    size_t input_signal_bytes;
    size_t output_buffer_bytes;
    float field_max_error_ = 0.0f;
};

// ---------------------------------------------------------------------------
// RndMemRead (cuda/bench_rndmem.cuh)
// ---------------------------------------------------------------------------
class RndMemBenchmark : public GPUABenchmark {
public:
    static constexpr int SAMPLE_MEM_NUM_ELEMS = 512 * 1024 * 1024 / sizeof(float);
    static const int DEFAULT_MIN_LOOP_LEN = 1000;
    static const int DEFAULT_MAX_LOOP_LEN = 48000;

    RndMemBenchmark(size_t buffer_size = BUFSIZE, size_t track_count = NTRACKS,
                    int min_loop_len = DEFAULT_MIN_LOOP_LEN, int max_loop_len = DEFAULT_MAX_LOOP_LEN);
    ~RndMemBenchmark() override;
    void setupBenchmark() override;
    void runKernel() override;
    void performBenchmarkIteration() override;
    void validate(ValidationData& validation_data) override;
    bool cpuGoldenWhole() override;
    void resetState() override;
    bool shardable() const override { return true; }
    std::vector<ResultArray> resultArrays() const override;

private:
    void initializePlayheads();
    void updatePlayheads();
    int min_loop_length_;
    int max_loop_length_;
    int sample_buffer_end_;
    float* h_sample_memory = nullptr;
    float* d_sample_memory = nullptr;
    int* h_playheads = nullptr;
    int* d_playheads = nullptr;
    float* playheads_start = nullptr;
    float* playheads_end = nullptr;
    float* h_output_buffer = nullptr;
    float* d_output_buffer = nullptr;
    float* cpu_reference = nullptr;
    size_t sample_memory_bytes;
    size_t playheads_bytes;
    size_t output_buffer_bytes;
};

// ---------------------------------------------------------------------------
// Registry (cuda/main.cu:75-115): the 17 names, in the reference's order.
// ---------------------------------------------------------------------------
namespace gab {
const std::vector<std::string>& benchmarkNames();
std::unique_ptr<GPUABenchmark> createBenchmark(const std::string& name);
// additive: the benchmarks whose tracks are independent, built for a channel shard of `tracks` tracks
// (setShard() then places it in the job); nullptr / false for those that reduce into shared outputs
std::unique_ptr<GPUABenchmark> createBenchmarkShard(const std::string& name, size_t tracks);
bool benchmarkShards(const std::string& name);
}  // namespace gab
