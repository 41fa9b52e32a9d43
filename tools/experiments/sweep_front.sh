B="python bench.py --no-cpu-baseline --no-extras --select-dtype none"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"; }
for i in 1 2; do run --dtype f16x2 --streams 4; run --dtype f16x2 --streams 2; done
for i in 1 2; do run --dtype f16 --weights normalized --streams 4; run --dtype f16 --weights normalized --streams 2; done
for i in 1 2; do run --model r50 --streams 4; run --model r50 --streams 2; done
run --model r50 --streams 2 --batch 1024 --chunk 256
run --model r50 --streams 2 --batch 1168 --chunk 584
