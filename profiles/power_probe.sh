#!/bin/bash
# Runs ON THE GPU BOX: samples power / clocks with rocm-smi while the benchmark's forward loop runs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30
echo "---- idle above, loaded below"
timeout -k 10 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | head -8
  echo "--"
  sleep 1
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-160
