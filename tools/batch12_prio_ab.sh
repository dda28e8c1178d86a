export GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so NBUF=128 GAB_BATCH_WAVES=12
for rep in 1 2; do for D in 0 2048 4096; do export GAB_CONV_SPLIT_DEBUG=$D; echo "debug $D"; timeout -k 10 100 python3 tools/batch_conv.py 1024 2>&1 | grep -v amdgpu; done; done
