"""gpuaudiobench_amd — MI355X (gfx950) implementation of the gpuaudiobench hot path.

The product is libgab_hip.so (hand-written HIP kernels behind the C ABI of
include/gab_c_api.h, plus the C++ GPUABenchmark harness and the `gpubench`
driver).  This package is the thin Python side: ctypes bindings that take torch
CUDA(=HIP) tensors for device memory and streams.  Nothing here computes on the
CPU; if the shared object is missing the import fails.
"""
from . import _capi
from ._capi import GabError, lib, check, CONV_STATELESS, CONV_STREAMING, CONV_STREAMING_HOST_IO, DWG_NAIVE, DWG_ACCEL
from .ops import (noop, gain, gainstats, datatransfer, iir, conv1d, rndmem, modal, modal_bank, dwg,
                  fft_r2c_1024, ConvPlan, FdtdPlan, LinkPlan, KeepWarm, fdtd_default_params, device_count)

from . import harness
from .harness import Benchmark, benchmark_names

__all__ = [
    "harness", "Benchmark", "benchmark_names",
    "GabError", "lib", "check", "CONV_STATELESS", "CONV_STREAMING", "CONV_STREAMING_HOST_IO", "DWG_NAIVE", "DWG_ACCEL",
    "noop", "gain", "gainstats", "datatransfer", "iir", "conv1d", "rndmem", "modal", "modal_bank", "dwg",
    "fft_r2c_1024", "ConvPlan", "FdtdPlan", "LinkPlan", "KeepWarm", "fdtd_default_params", "device_count",
]
