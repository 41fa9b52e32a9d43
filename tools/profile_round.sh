# usage (on the GPU box): bash tools/profile_round.sh <prefix, e.g. r03a>
# Every bench line / rocprofv3 summary the round's documents cite; writes gpurun_out/<prefix>_*.
set -x
P=${1:-r05a}
PART=${2:-all}      # benches | traces | all
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O/prof
cd $R
if [ $PART != traces ]; then
python bench.py --steps 100 --warmup 10 > $O/${P}_bench_r100.json 2> $O/${P}_bench_r100.err      # the driver's command: every leg
X="--no-config3 --no-config4 --no-configs1"        # the secondary lines: headline + roofline + exact leg only
python bench.py $X --dtype f16x2 --no-cpu-baseline > $O/${P}_bench_r100_f16x2.json 2>/dev/null
python bench.py $X --dtype f16x2 --weights normalized --no-cpu-baseline > $O/${P}_bench_r100_f16x2_normalized.json 2>/dev/null
python bench.py $X --dtype f16 --weights normalized --no-cpu-baseline --select-dtype none > $O/${P}_bench_r100_f16_normalized.json 2>/dev/null
python bench.py $X --dtype f32 --batch 256 --chunk 128 --steps 5 --warmup 2 --no-cpu-baseline --select-dtype none > $O/${P}_bench_r100_f32.json 2>/dev/null
python bench.py $X --model r50 --no-cpu-baseline > $O/${P}_bench_r50.json 2>/dev/null
python bench.py $X --model r50 --batch 256 --chunk 256 --streams 1 --no-cpu-baseline > $O/${P}_bench_r50_b256.json 2>/dev/null
python tools/layer_profile.py --batch 292 > $O/${P}_layers_r100_b292.txt 2>&1
python tools/layer_profile.py --batch 292 --dtype f16x2 > $O/${P}_layers_r100_b292_f16x2.txt 2>&1
python tools/layer_profile.py --model r50 --batch 256 > $O/${P}_layers_r50_b256.txt 2>&1
fi
if [ $PART = benches ]; then ls -la $O | tail -25; exit 0; fi
cd /tmp
for DT in bf16 f16x2 f32; do
  if [ $DT = f32 ]; then B="--batch 128 --chunk 128 --steps 3 --warmup 1"; else B="--batch 292 --chunk 292 --steps 10 --warmup 3"; fi
  ARGS="$R/bench.py --dtype $DT --streams 1 $B --shards 1 --no-cpu-baseline --no-extras --select-dtype none"
  rocprofv3 --kernel-trace -d $O/prof/kt_$DT -- python3 $ARGS > $O/prof/kt_$DT.log 2>&1
  DB=$(find $O/prof/kt_$DT -name "*.db" | head -1); echo DB=$DB
  python3 $R/tools/rocprof_db_stats.py $DB > $O/${P}_bench_r100_${DT}_streams1_kernel_stats.csv
  rm -rf $O/prof/kt_$DT
done
for DT in bf16 f16x2; do
  ARGS="$R/bench.py --dtype $DT --streams 1 --batch 292 --chunk 292 --shards 1 --no-cpu-baseline --no-extras --select-dtype none --steps 10 --warmup 3"
  for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    rocprofv3 --pmc $C --output-format csv -d $O/prof/pmc_$C -- python3 $ARGS > $O/prof/pmc_$C.log 2>&1
  done
  F=$(find $O/prof/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find $O/prof/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  M=$(find $O/prof/pmc_MfmaUtil -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $F $W 13 > $O/${P}_pmc_hbm_traffic_r100_${DT}_b292.csv
  python3 - "$M" > $O/${P}_pmc_mfma_util_r100_${DT}_b292.csv <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "MfmaUtil":
            continue
        k = (r["Kernel_Name"][:96], int(r["Grid_Size"]))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
print("kernel,grid_threads,launches,MfmaUtil_mean_percent")
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('"%s",%d,%d,%.1f' % (k[0].replace('"', "'"), k[1], n, s / n))
PY
  rm -rf $O/prof/pmc_FETCH_SIZE $O/prof/pmc_WRITE_SIZE $O/prof/pmc_MfmaUtil
done
ls -la $O | tail -25
