# Round-3 evidence (one MI355X box, one call): default bench line, rocprofv3 kernel stats of the same command
# (two-stream default and single-stream), steady-state summaries, underfill, the two PMC passes (FETCH_SIZE, WRITE_SIZE)
# behind profiles/r03_pmc_bench.json (and through it bench.py's roofline.traffic), the north-star forward's kernel trace and
# its same-process A/B.
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_final
mkdir -p $OUT
( while true; do date >> $OUT/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
python bench.py > $OUT/bench_pre.json 2> $OUT/bench_pre.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 > $GRAFT_REPO_ROOT/$OUT/prof_default.json 2> $GRAFT_REPO_ROOT/$OUT/prof_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/prof_off.json 2> $GRAFT_REPO_ROOT/$OUT/prof_off.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ns -o ns -- python3 $GRAFT_REPO_ROOT/tools/ns_forward_trace.py > $GRAFT_REPO_ROOT/$OUT/ns_fwd.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $OUT/default_bench_kernel_stats.csv
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $OUT/off_bench_kernel_stats.csv
python profiles/summarize_trace.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 45 > $OUT/off_bench_steady_state_summary.txt
python profiles/summarize_trace.py $(find /tmp/prof_a -name "*kernel_trace.csv" | head -1) 4 45 > $OUT/default_bench_steady_state_summary.txt
python profiles/underfill.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_underfill.txt 2>&1
python profiles/aten_per_step.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_aten_per_step.txt 2>&1
python3 profiles/summarize_ns_forward.py $(find /tmp/prof_ns -name "*kernel_trace.csv" | head -1) --timeline > $OUT/ns_forward_summary.txt 2>&1
cp $(find /tmp/prof_ns -name "*kernel_trace.csv" | head -1) $OUT/ns_forward_kernel_trace.csv
timeout -k 10 400 python tools/ns_ab.py 3 > $OUT/ns_ab.txt 2>&1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/pmc_f.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_f.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/pmc_w.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_w.err || exit 1
cd $GRAFT_REPO_ROOT
python profiles/pmc_summarize.py /tmp/pmc_f /tmp/pmc_w $OUT/pmc_bench.json > $OUT/pmc_sum.txt 2>&1
cp $OUT/pmc_bench.json profiles/r03_pmc_bench.json
# the bench line WITH this tree's measured traffic (profiles/r03_pmc_bench.json carries the same source stamp)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -4 $OUT/off_bench_steady_state_summary.txt
head -c 600 $OUT/bench.json
