B="python bench.py --no-cpu-baseline --no-extras --select-dtype none"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"; }
run
run --fine-max 0
run --fine-max 256
run --fine-max 600
run --chunk 146 --batch 1168
run --chunk 146 --batch 1168 --streams 4
run --chunk 219 --batch 1314 --streams 3
run
run --dtype f16x2
run --dtype f16x2 --chunk 146 --batch 1168 --streams 4
