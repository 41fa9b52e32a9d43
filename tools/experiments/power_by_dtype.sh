#!/bin/bash
# socket power / shader clock (rocm-smi, 4 samples/s) while bench.py runs ~12 s of each precision: usage power_by_dtype.sh OUTDIR
out=$1; mkdir -p $out
for dt in bf16 f16x2; do
  bash tools/power_trace.sh $out/smi_$dt.jsonl python3 bench.py --dtype $dt --steps $([ $dt = bf16 ] && echo 450 || echo 160) --warmup 5 \
      --no-cpu-baseline --no-extras --select-dtype none > $out/bench_$dt.json 2>/dev/null
  sleep 2
done
python3 - $out <<'PY'
import json, sys, os, re
out = sys.argv[1]
print("dtype,embeddings_per_s,samples,power_W_median,power_W_max,sclk_MHz_median")
for dt in ("bf16", "f16x2"):
    rate = json.loads(open(os.path.join(out, "bench_%s.json" % dt)).read().strip().splitlines()[-1])["value"]
    pw, ck = [], []
    for ln in open(os.path.join(out, "smi_%s.jsonl" % dt)):
        try:
            d = json.loads(ln)
        except Exception:
            continue
        c = d.get("card0", {})
        for k, v in c.items():
            if "Power" in k and "W" in k:
                try: pw.append(float(v))
                except Exception: pass
            if k.startswith("sclk clock speed"):
                m = re.search(r"(\d+)", str(v))
                if m: ck.append(int(m.group(1)))
    busy = [p for p in pw if p > 0.6 * max(pw)] if pw else []
    bck = ck[len(ck) // 4: -max(1, len(ck) // 8)] if len(ck) > 8 else ck
    med = lambda a: sorted(a)[len(a) // 2] if a else float("nan")
    print("%s,%.0f,%d,%.0f,%.0f,%s" % (dt, rate, len(pw), med(busy), max(pw) if pw else float("nan"), med(bck)))
PY
