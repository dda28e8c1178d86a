set -e
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "engine" > gpurun_out/eng_tests.log 2>&1
timeout -k 10 300 python tools/engine_conv.py 1024 63 > gpurun_out/engine_conv_new.txt 2>&1
timeout -k 10 800 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
