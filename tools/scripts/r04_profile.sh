# Round-4 working profile (one MI355X box): rocprofv3 kernel stats of the bench command, two-stream default and single-stream,
# with steady-state summaries.  Usage: bash tools/scripts/r04_profile.sh <tag>
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_prof_$1
mkdir -p $OUT
( while true; do date >> $OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/prof_off.json 2> $GRAFT_REPO_ROOT/$OUT/prof_off.err || exit 1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/off_bench_kernel_stats.csv
python profiles/summarize_trace.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 60 > $OUT/off_bench_steady_state_summary.txt
python profiles/underfill.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_underfill.txt 2>&1
python profiles/aten_per_step.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_aten_per_step.txt 2>&1
head -70 $OUT/off_bench_steady_state_summary.txt
