"""The harness by registry name (section H of the C ABI) and the host-side
generators / statistics / legacy writers (section G)."""
import ctypes as C

import numpy as np

from ._capi import (lib, check, BenchConfig, BenchResult, BenchValidation, Statistics,
                    CONV_STREAMING, CONV_STATELESS)


def benchmark_names():
    return [lib.gab_bench_name(i).decode() for i in range(lib.gab_bench_count())]


def noise(n, seed=42):
    out = np.empty(n, np.float32)
    check(lib.gab_generate_noise(out.ctypes.data_as(C.c_void_p), n, seed))
    return out


def conv1d_ir(ir_len, tracks, track_offset=0, total_tracks=None):
    out = np.empty(tracks * ir_len, np.float32)
    check(lib.gab_generate_conv1d_ir(out.ctypes.data_as(C.c_void_p), ir_len, track_offset, tracks,
                                     tracks if total_tracks is None else total_tracks))
    return out


def conv_accel_ir(ir_len, tracks, track_offset=0, total_tracks=None):
    out = np.empty(tracks * ir_len, np.float32)
    check(lib.gab_generate_conv_accel_ir(out.ctypes.data_as(C.c_void_p), ir_len, track_offset, tracks,
                                         tracks if total_tracks is None else total_tracks))
    return out


def statistics(latencies):
    lat = np.ascontiguousarray(latencies, np.float32)
    s = Statistics()
    check(lib.gab_calculate_statistics(lat.ctypes.data_as(C.c_void_p), lat.size, C.byref(s)))
    return s


def set_globals(fs=48000, buffer_size=512, n_tracks=128, n_runs=100):
    check(lib.gab_set_globals(fs, buffer_size, n_tracks, n_runs))


def json_results(latencies, name):
    lat = np.ascontiguousarray(latencies, np.float32)
    need = lib.gab_format_json_results(lat.ctypes.data_as(C.c_void_p), lat.size, name.encode(), None, 0)
    buf = C.create_string_buffer(need + 1)
    lib.gab_format_json_results(lat.ctypes.data_as(C.c_void_p), lat.size, name.encode(), buf, need + 1)
    return buf.value.decode()


def write_csv_results(latencies, name, filename):
    lat = np.ascontiguousarray(latencies, np.float32)
    check(lib.gab_write_csv_results(lat.ctypes.data_as(C.c_void_p), lat.size, name.encode(),
                                    filename.encode()))


class DawSim:
    """Buffer-slot scheduler: wait() returns at t0 + k*buffer_seconds (+/- jitter), spinning or
    sleeping (metal-swift/MetalSwiftBench/Core/BenchmarkUtilities.swift:140-178)."""
    MODES = {"spin": 0, "sleep": 1}

    def __init__(self, buffer_seconds=512.0 / 48000.0, mode="spin", jitter_us=0.0):
        self._h = C.c_void_p()
        check(lib.gab_dawsim_create(C.byref(self._h), buffer_seconds, self.MODES[mode], jitter_us * 1e-6))

    def wait(self):
        check(lib.gab_dawsim_wait(self._h))

    def stats(self):
        w, m = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib.gab_dawsim_stats(self._h, C.byref(w), C.byref(m)))
        return w.value, m.value

    def close(self):
        if self._h:
            lib.gab_dawsim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Benchmark:
    """GPUABenchmark by registry name: setup() / run() / validate()."""

    def __init__(self, name, **cfg):
        c = BenchConfig()
        lib.gab_bench_default_config(C.byref(c))
        for k, v in cfg.items():
            if not hasattr(c, k):
                raise TypeError("unknown config field %r" % k)
            setattr(c, k, v)
        self.name = name
        self._h = C.c_void_p()
        check(lib.gab_bench_create(C.byref(self._h), name.encode(), C.byref(c)))

    def setup(self):
        check(lib.gab_bench_setup(self._h))

    def set_shard(self, first_track, total_tracks):
        """This benchmark (created with n_tracks = the shard's own count) computes tracks
        [first_track, first_track + n_tracks) of a job of total_tracks (gab_bench_set_shard)."""
        check(lib.gab_bench_set_shard(self._h, first_track, total_tracks))

    def results(self):
        """{name: (array copy, layout, per_track)} of what the last iteration left on the host."""
        out = {}
        for i in range(lib.gab_bench_result_count(self._h)):
            name, data, n = C.c_char_p(), C.c_void_p(), C.c_size_t(0)
            layout, per = C.c_int(0), C.c_size_t(0)
            check(lib.gab_bench_result_array(self._h, i, C.byref(name), C.byref(data), C.byref(n),
                                             C.byref(layout), C.byref(per)))
            arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_float)), shape=(n.value,)).copy()
            out[name.value.decode()] = (arr, layout.value, per.value)
        return out

    def run(self, iterations=10, warmup=3):
        r = BenchResult()
        check(lib.gab_bench_run(self._h, iterations, warmup, C.byref(r)))
        return r

    def validate(self):
        v = BenchValidation()
        check(lib.gab_bench_validate(self._h, C.byref(v)))
        return v, lib.gab_bench_validation_text(self._h).decode()

    def algorithmic_bytes(self):
        n = C.c_size_t(0)
        check(lib.gab_bench_algorithmic_bytes(self._h, C.byref(n)))
        return n.value

    def latencies(self, capacity=100000):
        buf = (C.c_float * capacity)()
        n = lib.gab_bench_latencies(self._h, buf, capacity)
        return np.array(buf[:n], np.float32)

    def set_dawsim(self, buffer_seconds=512.0 / 48000.0, mode="spin", jitter_us=0.0, enable=True):
        """Pace every iteration of run() to one buffer slot (DAWSimulator of the Metal port)."""
        check(lib.gab_bench_set_dawsim(self._h, 1 if enable else 0, buffer_seconds,
                                       DawSim.MODES[mode], jitter_us * 1e-6))

    def set_keep_warm(self, enable=True):
        """Leave eight idle waves on the device for the length of run(), kicked after every iteration (gab_keep_warm)."""
        check(lib.gab_bench_set_keep_warm(self._h, 1 if enable else 0))

    def dawsim_stats(self):
        w, m = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib.gab_bench_dawsim_stats(self._h, C.byref(w), C.byref(m)))
        return w.value, m.value

    def close(self):
        if self._h:
            lib.gab_bench_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
