for s in 1 2 3 4; do
  python bench.py --model r50 --batch 256 --chunk 256 --streams 1 --shards $s --no-extras --no-cpu-baseline --select-dtype none --steps 200 --warmup 20 2>/dev/null > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('shards $s', round(d['value']), round(d['frac_mfma_peak_end_to_end'],4))"
done
