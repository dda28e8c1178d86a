#!/bin/bash
# round 5: the round trip's completion rule priced, its parity tests and the self-classifying stress (one gpurun call)
set -e -o pipefail
O=gpurun_out/r05_rt
mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "round_trip or datatransfer or datacopy or engine" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
echo "== product: the call returns at the launch's end" > $O/roundtrip_conv.txt
timeout -k 10 200 python tools/roundtrip_conv.py 1024 520 >> $O/roundtrip_conv.txt 2>&1
echo "== diagnostic build, GAB_RT_RETURN_ON_HINT=1: round 4's rule (returns on the pinned word)" >> $O/roundtrip_conv.txt
GAB_LIB_PATH=gpuaudiobench_amd/libgab_hip_ablate.so GAB_RT_RETURN_ON_HINT=1 timeout -k 10 200 python tools/roundtrip_conv.py 1024 520 >> $O/roundtrip_conv.txt 2>&1
echo "== product again" >> $O/roundtrip_conv.txt
timeout -k 10 200 python tools/roundtrip_conv.py 1024 520 >> $O/roundtrip_conv.txt 2>&1
cat $O/roundtrip_conv.txt
timeout -k 10 300 python tools/roundtrip_stress.py 8192 90 > $O/stress_8192.txt 2>&1 || { cat $O/stress_8192.txt; exit 1; }
tail -2 $O/stress_8192.txt
timeout -k 10 300 python tools/roundtrip_stress.py 1024 1500 > $O/stress_1024.txt 2>&1 || { cat $O/stress_1024.txt; exit 1; }
tail -2 $O/stress_1024.txt
