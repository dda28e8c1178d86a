# Role / store ablations of conv_split_kernel, queued by a C++ host (tools/ubench/two_chains, 2.8 us per launch).
# GAB_CONV_SPLIT_DEBUG bits: 1 near off, 4 far off, 8 no ring write, 16 no carry write, 32 no output write,
# 256 far workgroups first, 512 far at raised priority, 1024 near at raised priority, 2048 float2 output pieces
# (write-through sc1 stores of carry / ring / output were tried in round 2: 9.02-9.15 vs 9.10 us, no effect; removed)
for d in 0 1 4 8 16 32 56 57 60 256 512 1024 2048; do echo -n "debug=$d  "; GAB_CONV_SPLIT_DEBUG=$d tools/ubench/bin/two_chains --chains 1 --buffers 4000; done
for d in 0 2048; do echo -n "2 chains debug=$d  "; GAB_CONV_SPLIT_DEBUG=$d tools/ubench/bin/two_chains --chains 2 --buffers 4000; done
