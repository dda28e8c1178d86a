"""ctypes binding of libgab_hip.so (include/gab_c_api.h).

There is no CPU fallback: if the shared object is missing or a symbol is
absent, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAB_LIB_PATH: a diagnostic build of the same library (see build.py, GAB_BUILD_TAG)
LIB_PATH = os.environ.get("GAB_LIB_PATH") or os.path.join(_HERE, "libgab_hip.so")

GAB_OK = 0
GAB_ERR_INVALID_ARG = -1
GAB_ERR_RUNTIME = -2
GAB_ERR_UNSUPPORTED = -3

CONV_STATELESS = 0
CONV_STREAMING = 1
CONV_STREAMING_HOST_IO = 2
CONV_SCHEME_CLASSIC = 0
CONV_SCHEME_SPLIT = 1
DWG_NAIVE = 0
DWG_ACCEL = 1


class GabError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("gab error %d: %s" % (code, text))
        self.code = code


class WaveguideState(C.Structure):
    _fields_ = [("length", C.c_int), ("inputTapPos", C.c_int), ("outputTapPos", C.c_int),
                ("writePos", C.c_int), ("gain", C.c_float), ("reflection", C.c_float),
                ("damping", C.c_float), ("padding", C.c_float)]


class FdtdParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "nx", "ny", "nz", "source_x", "source_y", "source_z",
        "receiver_x", "receiver_y", "receiver_z", "steps_per_sample")] + [
            (n, C.c_float) for n in ("dt_over_rho_dx", "rho_c2_dt_over_dx", "absorption_coeff")]


class BenchConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "fs", "buffer_size", "n_tracks", "n_runs", "ir_length", "fdtd_grid", "conv_mode", "quiet", "modal_mode", "conv_batch", "fdtd_form", "datacopy_mode")]


class BenchResult(C.Structure):
    _fields_ = [("iterations", C.c_int)] + [(n, C.c_float) for n in (
        "mean_ms", "median_ms", "std_dev_ms", "min_ms", "max_ms", "p95_ms", "p99_ms",
        "gpu_median_ms")] + [("throughput_gbps", C.c_double), ("samples_per_sec", C.c_double),
                             ("bytes_processed", C.c_size_t)]


class Statistics(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "mean", "median", "std_dev", "min_val", "max_val", "p95", "p99")] + [("count", C.c_size_t)]


class BenchValidation(C.Structure):
    _fields_ = [("status", C.c_int), ("max_error", C.c_float), ("mean_error", C.c_float)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes).  Every prototype of include/gab_c_api.h is here;
# tests/test_capi_symbols.py checks the two lists against each other.
PROTOTYPES = {
    "gab_version": (_I, []),
    "gab_last_error": (C.c_char_p, []),
    "gab_device_count": (_I, [C.POINTER(_I)]),
    "gab_noop": (_I, [_P, _P, _Z, _P]),
    "gab_gain": (_I, [_P, _P, _Z, _F, _P]),
    "gab_gainstats": (_I, [_P, _P, _P, _I, _I, _F, _P]),
    "gab_datatransfer": (_I, [_P, _P, _I, _I, _P]),
    "gab_link_plan_create": (_I, [_I, C.POINTER(_P)]),
    "gab_link_plan_destroy": (None, [_P]),
    "gab_datatransfer_round_trip": (_I, [_P, _P, _P, _I, _I, _P]),
    "gab_datatransfer_round_trip_check": (_I, [_P]),
    "gab_keep_warm_create": (_I, [C.POINTER(_P), _I, C.c_double]),
    "gab_keep_warm_kick": (_I, [_P]),
    "gab_keep_warm_running": (_I, [_P, C.POINTER(_I)]),
    "gab_keep_warm_placement": (_I, [_P, _P, _P, _I, C.POINTER(_I)]),
    "gab_keep_warm_destroy": (_I, [_P]),
    "gab_iir": (_I, [_P, _P, C.POINTER(_F), _P, _I, _I, _P]),
    "gab_iir_sequential": (_I, [_P, _P, C.POINTER(_F), _P, _I, _I, _P]),
    "gab_conv1d": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "gab_conv1d_shard": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "gab_rndmem": (_I, [_P, _P, _P, _I, _I, _P]),
    "gab_modal": (_I, [_P, _P, _I, _I, _I, _P]),
    "gab_modal_bank_workspace_bytes": (_Z, [_I, _I, _I]),
    "gab_modal_bank": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "gab_dwg_workspace_bytes": (_Z, [_I, _I]),
    "gab_dwg": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gab_fft_r2c_1024": (_I, [_P, _P, _I, _P]),
    "gab_conv_create": (_I, [C.POINTER(_P), _I, _I, _I]),
    "gab_conv_destroy": (_I, [_P]),
    "gab_conv_set_ir": (_I, [_P, _P, _P]),
    "gab_conv_reset": (_I, [_P, _P]),
    "gab_conv_process": (_I, [_P, _P, _P, _I, _P]),
    "gab_conv_set_scheme": (_I, [_P, _I]),
    "gab_conv_get_scheme": (_I, [_P, C.POINTER(_I)]),
    "gab_conv_process_batch": (_I, [_P, _P, _P, _I, _P]),
    "gab_conv_round_trip": (_I, [_P, _P, _P, _P]),
    "gab_conv_round_trip_check": (_I, [_P]),
    "gab_conv_round_trip_set_check": (_I, [_P, _I]),
    "gab_conv_newest_block": (_I, [_P, _P, _P]),
    "gab_conv_round_trip_keep_warm": (_I, [_P, _I]),
    "gab_conv_round_trip_keep_warm_placement": (_I, [_P, _P, _P, _I, C.POINTER(_I)]),
    "gab_conv_engine_rings": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P)]),
    "gab_conv_engine_start": (_I, [_P, _I, C.POINTER(_P), C.POINTER(_P), _P]),
    "gab_conv_engine_publish": (_I, [_P, _I]),
    "gab_conv_engine_submit": (_I, [_P, _I, _I]),
    "gab_conv_engine_wait": (_I, [_P, _I, C.c_double]),
    "gab_conv_engine_running": (_I, [_P, C.POINTER(_I)]),
    "gab_conv_engine_completed": (_I, [_P, C.POINTER(_I)]),
    "gab_conv_engine_feed": (_I, [_P, _I, _I]),
    "gab_conv_engine_feed_one_in_flight": (_I, [_P, _I, _P]),
    "gab_conv_engine_stop": (_I, [_P]),
    "gab_conv_engine_round_trip": (_I, [_P, _P, _P]),
    "gab_conv_engine_set_idle_limit": (_I, [_P, C.c_double]),
    "gab_conv_state_bytes": (_I, [_P, C.POINTER(_Z), C.POINTER(_Z)]),
    "gab_fdtd_default_params": (_I, [_I, _I, _I, C.POINTER(FdtdParams)]),
    "gab_fdtd_create": (_I, [C.POINTER(_P), C.POINTER(FdtdParams)]),
    "gab_fdtd_destroy": (_I, [_P]),
    "gab_fdtd_reset": (_I, [_P, _P]),
    "gab_fdtd_process": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "gab_fdtd_copy_pressure": (_I, [_P, _P, _P]),
    "gab_fdtd_resident": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "gab_fdtd_set_form": (_I, [_P, _I]),
    "gab_fdtd_status": (_I, [_P, _P]),
    "gab_fdtd_set_track_positions": (_I, [_P, C.POINTER(_I), C.POINTER(_I), _I]),
    "gab_fdtd_create_slab": (_I, [C.POINTER(_P), C.POINTER(FdtdParams), _I, _I]),
    "gab_fdtd_owns": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "gab_fdtd_source_sums": (_I, [_P, _P, _I, _I, _P]),
    "gab_fdtd_inject": (_I, [_P, _I, _P]),
    "gab_fdtd_step": (_I, [_P, _I, _P]),
    "gab_fdtd_halo": (_I, [_P, _I, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "gab_fdtd_emit": (_I, [_P, _P, _I, _I, _P]),
    "gab_fdtd_strip": (_I, [_P, C.POINTER(_P), C.POINTER(_I)]),
    "gab_generate_noise": (_I, [_P, _Z, C.c_uint]),
    "gab_glibc_rand": (_I, [C.c_uint, C.c_ulonglong, _P, _Z]),
    "gab_shard_range": (_I, [_I, _I, _Z, C.POINTER(_Z), C.POINTER(_Z)]),
    "gab_shard_range_aligned": (_I, [_I, _I, _Z, _Z, C.POINTER(_Z), C.POINTER(_Z)]),
    "gab_shard_granule": (_Z, [C.c_char_p]),
    "gab_generate_conv1d_ir": (_I, [_P, _I, _Z, _Z, _Z]),
    "gab_generate_conv_accel_ir": (_I, [_P, _I, _Z, _Z, _Z]),
    "gab_calculate_statistics": (_I, [_P, _Z, C.POINTER(Statistics)]),
    "gab_set_globals": (_I, [_I, _I, _I, _I]),
    "gab_format_json_results": (_Z, [_P, _Z, C.c_char_p, C.c_char_p, _Z]),
    "gab_write_csv_results": (_I, [_P, _Z, C.c_char_p, C.c_char_p]),
    "gab_bench_default_config": (None, [C.POINTER(BenchConfig)]),
    "gab_bench_count": (_I, []),
    "gab_bench_name": (C.c_char_p, [_I]),
    "gab_bench_create": (_I, [C.POINTER(_P), C.c_char_p, C.POINTER(BenchConfig)]),
    "gab_bench_destroy": (_I, [_P]),
    "gab_bench_set_shard": (_I, [_P, _Z, _Z]),
    "gab_bench_result_count": (_I, [_P]),
    "gab_bench_result_array": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_Z), C.POINTER(_I), C.POINTER(_Z)]),
    "gab_bench_setup": (_I, [_P]),
    "gab_bench_run": (_I, [_P, _I, _I, C.POINTER(BenchResult)]),
    "gab_bench_validate": (_I, [_P, C.POINTER(BenchValidation)]),
    "gab_bench_latencies": (_I, [_P, C.POINTER(_F), _I]),
    "gab_bench_validation_text": (C.c_char_p, [_P]),
    "gab_bench_algorithmic_bytes": (_I, [_P, C.POINTER(_Z)]),
    "gab_dawsim_create": (_I, [C.POINTER(_P), C.c_double, _I, C.c_double]),
    "gab_dawsim_wait": (_I, [_P]),
    "gab_dawsim_stats": (_I, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "gab_dawsim_destroy": (_I, [_P]),
    "gab_bench_set_dawsim": (_I, [_P, _I, C.c_double, _I, C.c_double]),
    "gab_bench_set_keep_warm": (_I, [_P, _I]),
    "gab_bench_dawsim_stats": (_I, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gpuaudiobench_amd: %s is missing — build it with `python gpuaudiobench_amd/build.py` "
            "(there is no CPU fallback)" % LIB_PATH)
    # torch first: it brings its own libamdhip64.so.7; loading ours afterwards
    # binds to that same runtime instead of a second copy from /opt/rocm.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != GAB_OK:
        raise GabError(rc, lib.gab_last_error().decode("utf-8", "replace"))
