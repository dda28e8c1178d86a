cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_fdtd52
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fdtd52 -o tile -- python3 tools/fdtd_loop.py 52 512 128 > gpurun_out/prof_fdtd52/tile.txt 2>&1
GAB_FDTD_TILE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fdtd52 -o step -- python3 tools/fdtd_loop.py 52 512 128 > gpurun_out/prof_fdtd52/step.txt 2>&1
head -5 gpurun_out/prof_fdtd52/*kernel_stats.csv
