set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O/prof
cd $R
python bench.py > $O/r02d_bench_r100.json 2> $O/r02d_bench_r100.err
python bench.py --model r50 --no-cpu-baseline > $O/r02d_bench_r50.json 2>/dev/null
python bench.py --model r50 --batch 256 --chunk 256 --streams 1 --no-cpu-baseline > $O/r02d_bench_r50_b256.json 2>/dev/null
python tools/layer_profile.py --batch 292 > $O/r02d_layers_r100_b292.txt 2>&1
python tools/layer_profile.py --model r50 --batch 256 > $O/r02d_layers_r50_b256.txt 2>&1
cd /tmp
ARGS="$R/bench.py --streams 1 --batch 292 --chunk 292 --shards 1 --no-cpu-baseline --no-extras --steps 10 --warmup 3"
rocprofv3 --kernel-trace -d $O/prof/kt -- python3 $ARGS > $O/prof/kt.log 2>&1
DB=$(find $O/prof/kt -name "*.db" | head -1); echo DB=$DB
python3 $R/tools/rocprof_db_stats.py $DB > $O/r02d_bench_r100_streams1_b292_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --pmc $C --output-format csv -d $O/prof/pmc_$C -- python3 $ARGS > $O/prof/pmc_$C.log 2>&1
done
F=$(find $O/prof/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $O/prof/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
M=$(find $O/prof/pmc_MfmaUtil -name "*counter_collection.csv" | head -1)
echo $F $W $M
python3 $R/tools/pmc_summary.py $F $W 13 > $O/r02d_pmc_hbm_traffic_r100_b292.csv
python3 - "$M" > $O/r02d_pmc_mfma_util_r100_b292.csv <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "MfmaUtil":
            continue
        k = (r["Kernel_Name"][:80], int(r["Grid_Size"]))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
print("kernel,grid_threads,launches,MfmaUtil_mean_percent")
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('"%s",%d,%d,%.1f' % (k[0].replace('"', "'"), k[1], n, s / n))
PY
rm -rf $O/prof/kt $O/prof/pmc_FETCH_SIZE $O/prof/pmc_WRITE_SIZE $O/prof/pmc_MfmaUtil
ls -la $O | tail -15
