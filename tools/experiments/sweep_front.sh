B="python bench.py --no-cpu-baseline --no-extras --select-dtype none"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f' % d['value'])"; }
R50="--model r50 --batch 256 --chunk 256 --streams 1"
for i in 1 2 3; do run $R50; run $R50 --no-c56; done
for i in 1 2 3; do run; run --no-c56; done
