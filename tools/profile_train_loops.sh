# usage (on the GPU box): bash tools/profile_train_loops.sh <prefix>
# rocprofv3 kernel traces of the two training loops (SmallRes train_on_batch, customTrainModel) -> gpurun_out/<prefix>_*
P=${1:-r06a}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O/prof
cd /tmp
python3 $R/tools/smallres_step_profile.py 200 > $O/${P}_smallres_step.json 2>/dev/null
rocprofv3 --kernel-trace -d $O/prof/kt_sr -- python3 $R/tools/smallres_step_profile.py 200 > $O/prof/kt_sr.log 2>&1
DB=$(find $O/prof/kt_sr -name "*.db" | head -1)
python3 $R/tools/rocprof_db_stats.py $DB > $O/${P}_smallres_step_kernel_stats.csv
rm -rf $O/prof/kt_sr
python3 $R/tools/custom_train_time.py 4000 > $O/${P}_custom_train_time.json 2>/dev/null
rocprofv3 --kernel-trace -d $O/prof/kt_ctm -- python3 $R/tools/custom_train_time.py 1000 > $O/prof/kt_ctm.log 2>&1
DB=$(find $O/prof/kt_ctm -name "*.db" | head -1)
python3 $R/tools/rocprof_db_stats.py $DB > $O/${P}_custom_train_kernel_stats.csv
rm -rf $O/prof/kt_ctm
cat $O/${P}_smallres_step.json $O/${P}_custom_train_time.json
head -40 $O/${P}_smallres_step_kernel_stats.csv
head -20 $O/${P}_custom_train_kernel_stats.csv
# the few-pixel attack in lock-step (8 pairs, 16-bit search): which kernels its time is in
cd $R && bash tools/trace.sh ${P}_pixel_attack_lockstep tools/pixel_attack_time.py 8 screen 8 > $O/prof/kt_pa.log 2>&1
head -12 $O/${P}_pixel_attack_lockstep_kernel_stats.csv
