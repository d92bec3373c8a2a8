#!/bin/bash
# Stream-part sweep of the image engine (bench.py --streams / --split), interleaved rounds.  usage: bash profiles/ab_streams_r02.sh "1" "2" "3" "144,112"
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    case "$v" in *,*) arg="--split $v";; *) arg="--streams $v";; esac
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-dtype --steps 40 --profile-every 0 $arg 2> gpurun_out/abs_$round.err \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts $v round $round:', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')" || { tail -5 gpurun_out/abs_$round.err; exit 1; }
  done
done
