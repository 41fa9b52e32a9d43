# usage (on the GPU box): bash tools/trace.sh <name> <script.py> [args...]
# rocprofv3 --kernel-trace of `python3 <script.py> args` -> gpurun_out/<name>_kernel_stats.csv (+ <name>_timeline.txt: 40
# consecutive launches from the middle of the run: start offset us, duration us, kernel)
N=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O/prof
S=$R/$1; shift
cd /tmp
rocprofv3 --kernel-trace -d $O/prof/kt_$N -- python3 $S "$@" > $O/prof/kt_$N.log 2>&1
DB=$(find $O/prof/kt_$N -name "*.db" | head -1)
python3 $R/tools/rocprof_db_stats.py $DB > $O/${N}_kernel_stats.csv
python3 - $DB > $O/${N}_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[len(rows) // 2:len(rows) // 2 + 40]
t0 = rows[0][1]
for n, s, e in rows:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
PY
rm -rf $O/prof/kt_$N
tail -3 $O/prof/kt_$N.log
cut -c1-180 $O/${N}_kernel_stats.csv | head -${LINES_STATS:-25}
cat $O/${N}_timeline.txt
