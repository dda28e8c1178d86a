#!/bin/bash
# Diagnostic: per-stage timing of the streaming conv kernel (outputs are wrong in ablated runs).
set -e
GAB_ABLATE=1 python gpuaudiobench_amd/build.py --force > /dev/null
for a in 0 1 2 3 4 5; do echo -n "ablate=$a "; GAB_CONV_ABLATE=$a python tools/conv_loop.py 1000; done
python gpuaudiobench_amd/build.py --force > /dev/null
