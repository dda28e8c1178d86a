"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so, GAB_RT_STREAM_PAGEABLE=1): does the runtime's upload of a
PAGEABLE buffer write a destination word more than once?  The round trip's kernel takes a word the moment it lands and
puts the sentinel back; a word written again afterwards is found non-sentinel once the call is over.  Pageable inputs at
several byte offsets from an allocation's start, many calls; prints how many staging words were left dirty.
    python tools/roundtrip_pageable_probe.py [channels] [calls]"""
import ctypes, os, sys
os.environ["GAB_RT_STREAM_PAGEABLE"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
B, L = 512, 4096
fn = gab.lib.gab_debug_rt_stage_dirty; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]; fn.restype = ctypes.c_int
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
b = gab.ConvPlan(T, B, L, scheme="classic"); b.set_ir(ir)
h_out = torch.empty(T * B).pin_memory()
big = np.zeros(T * B + 4096, np.float32)
events = 0
for i in range(N):
    off = [0, 1, 3, 16, 17, 64, 100, 1024, 1031][i % 9]            # floats from the allocation's start
    x = big[off:off + T * B]
    x[:] = gab.harness.noise(T * B, seed=2000 + i)
    b.round_trip(torch.from_numpy(x), h_out)
    first = ctypes.c_longlong(-1)
    dirty = fn(b._h, ctypes.byref(first))
    if dirty:
        events += 1
        print("call %d (offset %d floats): %d staging words left non-sentinel, first at word %d (channel %d, sample %d)"
              % (i, off, dirty, first.value, first.value // B, first.value % B), flush=True)
        b.reset()
        # put the sentinel back by a fresh plan state: recreate the plan (the staging buffer belongs to it)
        b.close(); b = gab.ConvPlan(T, B, L, scheme="classic"); b.set_ir(ir)
print("%d calls with pageable inputs consumed as they landed: %d left staging words dirty" % (N, events))
