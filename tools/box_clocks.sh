#!/bin/bash
# What differs between a box where bench.py reads 0.93 and one where it reads 0.98?  The driver's launches in a long loop
# (20 000 steps, ~13 s) with rocm-smi's clocks / power / temperature sampled beside it, then the bench line's value.
#   bash tools/box_clocks.sh [tag] -> gpurun_out/box_clocks_<tag>.txt
TAG=${1:-r06}
OUT=$PWD/gpurun_out/box_clocks_$TAG.txt
: > $OUT
python3 bench.py --steps 20000 --warmup 20 --no-cpu-baseline --no-side-legs > $OUT.line 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3; do
  rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)|Performance" >> $OUT
  echo "--" >> $OUT
  sleep 2
done
wait $BP
python3 -c "
import json; d=json.loads(open('$OUT.line').read().strip().splitlines()[-1])
print('value %.1f buffers/s  frac %.4f  launch %.1f us' % (d['value'], d['roofline']['frac'], d['roofline']['launch_us']))" >> $OUT
rm -f $OUT.line
cat $OUT
