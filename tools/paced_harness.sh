#!/bin/bash
# The harness under DAW pacing (one iteration per 512 / 48000 s slot), every case as it is and with --keepWarm:
# median wall latency of the iteration (what a reference user reads), 200 iterations each.
#   tools/paced_harness.sh [tag]        -> gpurun_out/<tag>/paced_harness.txt
set -e
cd "$(dirname "$0")/.."
TAG=${1:-paced_harness}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
G=gpuaudiobench_amd/gpubench
run() {   # label, then gpubench flags
    local label="$1"; shift
    for warm in "" "--keepWarm"; do
        local med
        med=$(timeout -k 10 120 $G "$@" --nRuns 200 --cpu-threads 0 --dawsim $warm --json 2>/dev/null | python3 -c "
import sys, json
t = sys.stdin.read(); i = t.index('{\n  \"benchmark\"'); d = json.loads(t[i:t.index('\n}\n', i) + 3])
s = d['statistics']
print('%.1f %.1f %s' % (1e3 * s['p50_ms'], 1e3 * s['p95_ms'], d['validation']['passed']))")
        printf "%-44s %-10s median %8s us  p95 %8s us  valid %s\n" "$label" "${warm:-idle}" $med
    done
}
{
run "Conv1D_accel C3 roundtrip"        --benchmark Conv1D_accel --nTracks 1024 --irLength 4096 --convMode roundtrip
run "Conv1D_accel C3 stream (copies)"  --benchmark Conv1D_accel --nTracks 1024 --irLength 4096
run "gain 1024 tracks"                 --benchmark gain --nTracks 1024
run "FFT1D 1024 tracks"                --benchmark FFT1D --nTracks 1024
run "IIRFilter 1024 tracks"            --benchmark IIRFilter --nTracks 1024
run "Conv1D C2 256 x 256"              --benchmark Conv1D --nTracks 256 --irLength 256
run "datacopy5050"                     --benchmark datacopy5050
run "DWG1DAccel 1024"                  --benchmark DWG1DAccel --nTracks 1024
} | tee "$OUT/paced_harness.txt"
