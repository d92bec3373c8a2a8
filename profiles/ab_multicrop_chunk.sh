for c in 256 512 1024; do python bench.py --mode multicrop --batch $c 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c:', round(d['value'],2), 'img/s', round(d['crops_per_s']), 'crops/s', round(d['ms_per_step'],1), 'ms')"; done
