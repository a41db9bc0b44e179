# End-of-session evidence: default bench line, rocprofv3 kernel stats (two-stream default and single-stream), summaries.
set -x
cd /root/repo
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o a -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 1 > /root/repo/gpurun_out/final_prof_default.json 2> /root/repo/gpurun_out/final_prof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/final_prof_off.json 2> /root/repo/gpurun_out/final_prof_off.err
cd /root/repo
cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) gpurun_out/final_default_kernel_stats.csv
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) gpurun_out/final_off_kernel_stats.csv
python profiles/summarize_trace.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/final_off_summary.txt
python profiles/summarize_trace.py $(find /tmp/prof_a -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/final_default_summary.txt
python profiles/underfill.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > gpurun_out/final_off_underfill.txt 2>&1
head -4 gpurun_out/final_off_summary.txt
head -c 400 gpurun_out/bench_final.json
