python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "patch_embed or cfg5" 2>&1 | tail -3
for i in 1 2; do python bench.py --arch ViT-L/14@336px --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-second-dtype 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vitl', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'e2e', round(d['end_to_end_mfma_frac'],3), 'patch', round(d['kernels']['patch_embed']['avg_us'],1))"; done
