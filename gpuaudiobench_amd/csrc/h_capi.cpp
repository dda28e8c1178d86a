// h_capi.cpp — sections G and H of the C ABI: the harness's host-side data
// generators and GPUABenchmark driven by registry name
// (cuda/main.cu:117-164 runSelectedBenchmark, without the printing).
#include <cstring>
#include <mutex>

#include "gab/benchmarks.hpp"
#include "gab_common.hpp"

struct gab_bench {
    std::unique_ptr<GPUABenchmark> impl;
    GPUABenchmark::BenchmarkResult last;
    std::string validation_text;
    bool set_up = false;
};

namespace {
// The reference keeps its configuration in mutable process globals that the
// benchmark constructors read; creation is serialised around them.
std::mutex g_globals_mu;
}

extern "C" {

int gab_generate_noise(float* h_buf, size_t n, unsigned seed) {
    return gab::guarded([&]() -> int {
        if (!h_buf && n) return gab::bad_arg("gab_generate_noise: null pointer");
        BenchmarkUtils::generateRandomAudioData(h_buf, n, seed);
        return GAB_OK;
    });
}

int gab_generate_conv1d_ir(float* h_ir, int ir_len, size_t off, size_t n, size_t total) {
    return gab::guarded([&]() -> int {
        if (!h_ir || ir_len <= 0 || total == 0 || off + n > total)
            return gab::bad_arg("gab_generate_conv1d_ir: bad arguments");
        BenchmarkUtils::generateConv1DImpulseResponses(h_ir, ir_len, off, n, total);
        return GAB_OK;
    });
}

int gab_generate_conv_accel_ir(float* h_ir, int ir_len, size_t off, size_t n, size_t total) {
    return gab::guarded([&]() -> int {
        if (!h_ir || ir_len <= 0 || total == 0 || off + n > total)
            return gab::bad_arg("gab_generate_conv_accel_ir: bad arguments");
        BenchmarkUtils::generateConvAccelImpulseResponses(h_ir, ir_len, off, n, total);
        return GAB_OK;
    });
}

int gab_glibc_rand(unsigned seed, unsigned long long skip, int* out, size_t n) {
    return gab::guarded([&]() -> int {
        if (!out && n) return gab::bad_arg("gab_glibc_rand: null pointer");
        BenchmarkUtils::GlibcRand rng(seed);
        rng.discard(skip);
        for (size_t i = 0; i < n; ++i) out[i] = rng.next();
        return GAB_OK;
    });
}

int gab_calculate_statistics(const float* lat, size_t n, gab_statistics* out) {
    return gab::guarded([&]() -> int {
        if (!out || (!lat && n)) return gab::bad_arg("gab_calculate_statistics: null pointer");
        BenchmarkUtils::Statistics s = BenchmarkUtils::calculateStatistics(std::vector<float>(lat, lat + n));
        out->mean = s.mean; out->median = s.median; out->std_dev = s.std_dev;
        out->min_val = s.min_val; out->max_val = s.max_val; out->p95 = s.p95; out->p99 = s.p99;
        out->count = s.count;
        return GAB_OK;
    });
}

int gab_set_globals(int fs, int buffer_size, int n_tracks, int n_runs) {
    if (fs <= 0 || buffer_size <= 0 || n_tracks <= 0 || n_runs <= 0)
        return gab::bad_arg("gab_set_globals: all values must be > 0");
    std::lock_guard<std::mutex> lock(g_globals_mu);
    FS = fs; BUFSIZE = buffer_size; NTRACKS = n_tracks; NRUNS = n_runs;
    return GAB_OK;
}

size_t gab_format_json_results(const float* lat, size_t n, const char* name, char* buf, size_t cap) {
    std::string j;
    try {
        j = generateJSONResults(std::vector<float>(lat, lat + (lat ? n : 0)), name ? name : "");
    } catch (...) {
        return 0;
    }
    if (buf && cap) {
        size_t m = j.size() < cap - 1 ? j.size() : cap - 1;
        std::memcpy(buf, j.data(), m);
        buf[m] = '\0';
    }
    return j.size();
}

int gab_write_csv_results(const float* lat, size_t n, const char* name, const char* filename) {
    return gab::guarded([&]() -> int {
        if (!lat || !n || !name || !filename) return gab::bad_arg("gab_write_csv_results: bad arguments");
        const bool was_quiet = GAB_QUIET;
        writeCSVResults(std::vector<float>(lat, lat + n), name, filename);
        (void)was_quiet;
        return GAB_OK;
    });
}

void gab_bench_default_config(gab_bench_config* c) {
    if (!c) return;
    c->fs = 48000;
    c->buffer_size = 512;
    c->n_tracks = 128;
    c->n_runs = 100;
    c->ir_length = 0;
    c->fdtd_grid = 0;
    c->conv_mode = GAB_CONV_STREAMING;
    c->quiet = 1;
    c->modal_mode = 0;
    c->conv_batch = 0;
    c->fdtd_form = 0;
    c->datacopy_mode = 0;
}

int gab_bench_count(void) { return static_cast<int>(gab::benchmarkNames().size()); }

const char* gab_bench_name(int index) {
    const auto& names = gab::benchmarkNames();
    if (index < 0 || index >= static_cast<int>(names.size())) return nullptr;
    return names[index].c_str();
}

int gab_bench_create(gab_bench** out, const char* name, const gab_bench_config* cfg) {
    return gab::guarded([&]() -> int {
        if (!out || !name) return gab::bad_arg("gab_bench_create: null argument");
        gab_bench_config c;
        if (cfg) c = *cfg; else gab_bench_default_config(&c);
        if (c.buffer_size <= 0 || c.n_tracks <= 0 || c.fs <= 0 || c.n_runs <= 0)
            return gab::bad_arg("gab_bench_create: fs, buffer_size, n_tracks and n_runs must be > 0");
        std::lock_guard<std::mutex> lock(g_globals_mu);
        FS = c.fs; BUFSIZE = c.buffer_size; NTRACKS = c.n_tracks; NRUNS = c.n_runs;
        IR_LENGTH = c.ir_length; FDTD_GRID = c.fdtd_grid;
        CONV_STREAMING = (c.conv_mode == GAB_CONV_STREAMING) ? 1 : (c.conv_mode == GAB_CONV_STREAMING_HOST_IO ? 2 : 0);
        GAB_QUIET = c.quiet != 0;
        MODAL_REAL = c.modal_mode != 0;
        CONV_BATCH = c.conv_batch;
        FDTD_FORM = c.fdtd_form == 1 ? 1 : 0;
        DATACOPY_SEQUENTIAL = c.datacopy_mode == 1 ? 1 : 0;
        auto impl = gab::createBenchmark(name);
        if (!impl) return gab::bad_arg("gab_bench_create: unknown benchmark name");
        auto* b = new gab_bench;
        b->impl = std::move(impl);
        *out = b;
        return GAB_OK;
    });
}

int gab_bench_destroy(gab_bench* b) {
    delete b;
    return GAB_OK;
}

int gab_bench_set_shard(gab_bench* b, size_t first_track, size_t total_tracks) {
    return gab::guarded([&]() -> int {
        if (!b) return gab::bad_arg("gab_bench_set_shard: null benchmark");
        if (b->set_up) return gab::bad_arg("gab_bench_set_shard: call it before gab_bench_setup");
        b->impl->setShard(first_track, total_tracks);
        return GAB_OK;
    });
}

int gab_bench_result_count(gab_bench* b) { return (b && b->set_up) ? static_cast<int>(b->impl->resultArrays().size()) : 0; }

int gab_bench_result_array(gab_bench* b, int index, const char** name, const float** data, size_t* count, int* layout,
                           size_t* per_track) {
    return gab::guarded([&]() -> int {
        if (!b || !b->set_up) return gab::bad_arg("gab_bench_result_array: no benchmark that has been set up");
        const auto arrays = b->impl->resultArrays();
        if (index < 0 || index >= static_cast<int>(arrays.size())) return gab::bad_arg("gab_bench_result_array: index out of range");
        const auto& a = arrays[index];
        if (name) *name = a.name;
        if (data) *data = a.data;
        if (count) *count = a.count;
        if (layout) *layout = a.layout;
        if (per_track) *per_track = a.per_track;
        return GAB_OK;
    });
}

int gab_bench_setup(gab_bench* b) {
    return gab::guarded([&]() -> int {
        if (!b) return gab::bad_arg("gab_bench_setup: null benchmark");
        b->impl->setupBenchmark();
        b->set_up = true;
        return GAB_OK;
    });
}

int gab_bench_run(gab_bench* b, int iterations, int warmup, gab_bench_result* out) {
    return gab::guarded([&]() -> int {
        if (!b) return gab::bad_arg("gab_bench_run: null benchmark");
        if (!b->set_up) return gab::bad_arg("gab_bench_run: gab_bench_setup has not been called");
        if (iterations <= 0 || warmup < 0) return gab::bad_arg("gab_bench_run: bad iteration counts");
        b->last = b->impl->runBenchmark(iterations, warmup);
        if (out) {
            const auto& r = b->last;
            out->iterations = r.iterations;
            out->mean_ms = r.statistics.mean;
            out->median_ms = r.statistics.median;
            out->std_dev_ms = r.statistics.std_dev;
            out->min_ms = r.statistics.min_val;
            out->max_ms = r.statistics.max_val;
            out->p95_ms = r.statistics.p95;
            out->p99_ms = r.statistics.p99;
            out->gpu_median_ms = r.gpu_latencies.empty() ? 0.0f : r.gpu_statistics.median;
            out->throughput_gbps = r.throughput_gbps;
            out->samples_per_sec = r.samples_per_sec;
            out->bytes_processed = r.bytes_processed;
        }
        return GAB_OK;
    });
}

int gab_bench_validate(gab_bench* b, gab_bench_validation* out) {
    return gab::guarded([&]() -> int {
        if (!b) return gab::bad_arg("gab_bench_validate: null benchmark");
        if (!b->set_up) return gab::bad_arg("gab_bench_validate: gab_bench_setup has not been called");
        GPUABenchmark::ValidationData v;
        b->impl->validate(v);
        b->validation_text.clear();
        for (const auto& m : v.messages) {
            if (!b->validation_text.empty()) b->validation_text += "\n";
            b->validation_text += m;
        }
        if (out) {
            out->status = static_cast<int>(v.status);
            out->max_error = v.max_error;
            out->mean_error = v.mean_error;
        }
        return GAB_OK;
    });
}

const char* gab_bench_validation_text(gab_bench* b) { return b ? b->validation_text.c_str() : ""; }

int gab_bench_algorithmic_bytes(gab_bench* b, size_t* bytes) {
    if (!b || !bytes) return gab::bad_arg("gab_bench_algorithmic_bytes: null argument");
    *bytes = b->impl->algorithmicBytes();
    return GAB_OK;
}

int gab_bench_latencies(gab_bench* b, float* out, int capacity) {
    if (!b || (!out && capacity > 0)) return 0;
    int n = static_cast<int>(b->last.latencies.size());
    if (n > capacity) n = capacity;
    if (n > 0) std::memcpy(out, b->last.latencies.data(), sizeof(float) * n);
    return n;
}

struct gab_dawsim {
    BenchmarkUtils::DAWSimulator sim;
    BenchmarkUtils::DAWSimulationState state;
};

static int dawsim_fill(BenchmarkUtils::DAWSimulator& sim, double buffer_seconds, int mode, double jitter_seconds,
                       const char* who) {
    if (!(buffer_seconds > 0.0) || jitter_seconds < 0.0 || (mode != 0 && mode != 1)) {
        std::string m = std::string(who) + ": buffer_seconds must be > 0, jitter_seconds >= 0, mode 0 (spin) or 1 (sleep)";
        return gab::bad_arg(m.c_str());
    }
    sim.bufferDuration = buffer_seconds;
    sim.mode = mode == 1 ? BenchmarkUtils::DAWSimulationMode::SLEEP : BenchmarkUtils::DAWSimulationMode::SPIN;
    sim.jitterSeconds = jitter_seconds;
    return GAB_OK;
}

int gab_dawsim_create(gab_dawsim** s, double buffer_seconds, int mode, double jitter_seconds) {
    return gab::guarded([&]() -> int {
        if (!s) return gab::bad_arg("gab_dawsim_create: null handle pointer");
        BenchmarkUtils::DAWSimulator sim;
        if (int rc = dawsim_fill(sim, buffer_seconds, mode, jitter_seconds, "gab_dawsim_create")) return rc;
        *s = new gab_dawsim{sim, {}};
        return GAB_OK;
    });
}

int gab_dawsim_wait(gab_dawsim* s) {
    if (!s) return gab::bad_arg("gab_dawsim_wait: null handle");
    s->sim.wait(s->state);
    return GAB_OK;
}

int gab_dawsim_stats(const gab_dawsim* s, unsigned long long* waits, unsigned long long* missed_slots) {
    if (!s) return gab::bad_arg("gab_dawsim_stats: null handle");
    if (waits) *waits = s->state.waits;
    if (missed_slots) *missed_slots = s->state.late;
    return GAB_OK;
}

int gab_dawsim_destroy(gab_dawsim* s) {
    delete s;
    return GAB_OK;
}

int gab_bench_set_dawsim(gab_bench* b, int enable, double buffer_seconds, int mode, double jitter_seconds) {
    return gab::guarded([&]() -> int {
        if (!b) return gab::bad_arg("gab_bench_set_dawsim: null benchmark");
        if (!enable) { b->impl->clearDawSimulator(); return GAB_OK; }
        BenchmarkUtils::DAWSimulator sim;
        if (int rc = dawsim_fill(sim, buffer_seconds, mode, jitter_seconds, "gab_bench_set_dawsim")) return rc;
        b->impl->setDawSimulator(sim);
        return GAB_OK;
    });
}

int gab_bench_set_keep_warm(gab_bench* b, int enable) {
    if (!b) return gab::bad_arg("gab_bench_set_keep_warm: null benchmark");
    b->impl->setKeepWarm(enable != 0);
    return GAB_OK;
}

int gab_bench_dawsim_stats(gab_bench* b, unsigned long long* waits, unsigned long long* missed_slots) {
    if (!b) return gab::bad_arg("gab_bench_dawsim_stats: null benchmark");
    if (waits) *waits = b->last.daw_waits;
    if (missed_slots) *missed_slots = b->last.daw_missed_slots;
    return GAB_OK;
}

}  // extern "C"
