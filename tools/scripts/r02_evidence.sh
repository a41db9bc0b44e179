# Round-2 evidence (one MI355X box, one call): default bench line, rocprofv3 kernel stats of the same command
# (two-stream default and single-stream), steady-state summaries, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) that
# profiles/r02_pmc_bench.json - and through it bench.py's roofline.traffic - come from.
set -x
cd /root/repo
( while true; do date >> gpurun_out/r02_heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
python bench.py > gpurun_out/r02_bench_pre.json 2> gpurun_out/r02_bench_pre.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 > /root/repo/gpurun_out/r02_prof_default.json 2> /root/repo/gpurun_out/r02_prof_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/r02_prof_off.json 2> /root/repo/gpurun_out/r02_prof_off.err || exit 1
cd /root/repo
cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) gpurun_out/r02_default_bench_kernel_stats.csv
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) gpurun_out/r02_off_bench_kernel_stats.csv
python profiles/summarize_trace.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/r02_off_bench_steady_state_summary.txt
python profiles/summarize_trace.py $(find /tmp/prof_a -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/r02_default_bench_steady_state_summary.txt
python profiles/underfill.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/r02_off_underfill.txt 2>&1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 /root/repo/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/r02_pmc_f.json 2> /root/repo/gpurun_out/r02_pmc_f.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 /root/repo/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/r02_pmc_w.json 2> /root/repo/gpurun_out/r02_pmc_w.err || exit 1
cd /root/repo
python profiles/pmc_summarize.py /tmp/pmc_f /tmp/pmc_w gpurun_out/r02_pmc_bench.json > gpurun_out/r02_pmc_sum.txt 2>&1
cp gpurun_out/r02_pmc_bench.json profiles/r02_pmc_bench.json
# the bench line WITH this tree's measured traffic (profiles/r02_pmc_bench.json carries the same source stamp)
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
head -4 gpurun_out/r02_off_bench_steady_state_summary.txt
head -c 500 gpurun_out/r02_bench.json
