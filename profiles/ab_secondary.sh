for round in 1 2; do
for fam in "" "--gemm-family 256"; do
python bench.py --arch ViT-L/14@336px --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-second-dtype $fam 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vitl [$fam]', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms gemm frac', round(d['roofline']['frac'],3), 'e2e', round(d['end_to_end_mfma_frac'],3), {k:round(v['avg_us'],1) for k,v in d['gemm_shapes'].items()})"
python bench.py --mode tune --dtype bf16 --steps 10 --warmup 3 $fam 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune [$fam]', round(d['value'],1), d['unit'], round(d['ms_per_step'],2))"
python bench.py --mode tune --tune-model DenseCLIP --dtype fp16 --steps 10 --warmup 3 $fam 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dense [$fam]', round(d['value'],1), d['unit'], round(d['ms_per_step'],2))"
done; done
