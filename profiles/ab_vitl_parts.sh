#!/bin/bash
# ViT-L/14@336 B=128 fp16: how the batch is cut into stream parts (bench.py --streams / --split), interleaved.  Round fill of the 384 x 256 tiles on 256 CUs: 64 + 64 images
# = 97 row blocks each (out-proj / c_proj 1.52 rounds, qkv 4.55, c_fc 6.06); 85 + 43 = 128 + 65 row blocks (2.0 / 6.0 / 8.0 and 1.02 / 3.05 / 4.06); one part = 193 (3.02 / 9.05 / 12.06)
for r in 1 2; do for s in "--streams 1" "--streams 2" "--split 85,43" "--split 86,42" "--split 43,85"; do
python bench.py --arch ViT-L/14@336px --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-second-dtype --no-companions $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$s round $r:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms  e2e', round(d['end_to_end_mfma_frac'],4))"
done; done
