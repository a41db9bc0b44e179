# Round-4 evidence (one MI355X box, one call): default bench line, rocprofv3 kernel stats of the same command (two-stream
# default and single-stream) with steady-state summaries, underfill, ATen launches, the two PMC passes (FETCH_SIZE,
# WRITE_SIZE) behind profiles/r04_pmc_bench.json (and through it bench.py's roofline.traffic), the MFMA / LDS counters of
# the P3 conv, the north-star forward's kernel trace, its same-process A/B, the kernel probes and the secondary configs.
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final
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
python profiles/summarize_trace.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 60 > $OUT/off_bench_steady_state_summary.txt
python profiles/summarize_trace.py $(find /tmp/prof_a -name "*kernel_trace.csv" | head -1) 4 60 > $OUT/default_bench_steady_state_summary.txt
python profiles/underfill.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_underfill.txt 2>&1
python profiles/aten_per_step.py $(find /tmp/prof_b -name "*kernel_trace.csv" | head -1) 4 > $OUT/off_aten_per_step.txt 2>&1
python3 profiles/summarize_ns_forward.py $(find /tmp/prof_ns -name "*kernel_trace.csv" | head -1) --timeline > $OUT/ns_forward_summary.txt 2>&1
cp $(find /tmp/prof_ns -name "*kernel_trace.csv" | head -1) $OUT/ns_forward_kernel_trace.csv
timeout -k 10 400 python tools/ns_ab.py 3 > $OUT/ns_ab.txt 2>&1
timeout -k 10 300 python tools/halo_probe.py > $OUT/halo_probe.txt 2>&1
timeout -k 10 300 python tools/pyr_probe.py > $OUT/pyr_probe.txt 2>&1
timeout -k 10 200 python tools/ns_host_probe.py > $OUT/ns_host_vs_graph.txt 2>&1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/pmc_f.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_f.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/pmc_w.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_w.err || exit 1
cd $GRAFT_REPO_ROOT
python profiles/pmc_summarize.py /tmp/pmc_f /tmp/pmc_w $OUT/pmc_bench.json > $OUT/pmc_sum.txt 2>&1
cp $OUT/pmc_bench.json profiles/r04_pmc_bench.json
# MFMA / LDS counters of the P3 conv (forward: conv_halo; weight gradient: conv_wgrad_dma), one pass per counter pair
cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcm_$tag -o c -- python3 $GRAFT_REPO_ROOT/profiles/pmc_conv.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_mfma_$tag.err || echo "FAILED $grp"
done
cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/pmc_mfma_l3.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmcm_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'conv_halo' if 'conv_halo' in n else 'conv_igemm_dma' if 'conv_igemm_dma' in n else 'conv_wgrad_dma' if 'conv_wgrad_dma' in n else None
        if k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    print(k, {c: round(v, 1) for c, v in m.items()})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'SQ_BUSY_CU_CYCLES' in m:
        print(f"   matrix pipe busy / CU busy: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / m['SQ_BUSY_CU_CYCLES']:.3f} (MFMA busy cycles summed over the 4 SIMDs of a CU / 4)")
PY
# weight gradients: slab bytes of a step, K-loop stamps of the panel kernel (diagnostic build, if present), host phases
timeout -k 10 200 python tools/wgrad_slab_probe.py > $OUT/wgrad_slab_probe.txt 2>&1
if [ -f tools/micro/tuning_build/libsihl_hip_wstamps.so ]; then
  SIHL_HIP_LIB=$PWD/tools/micro/tuning_build/libsihl_hip_wstamps.so timeout -k 10 120 python tools/wgrad_stamps.py > $OUT/wgrad_stamps.txt 2>&1
fi
timeout -k 10 200 python tools/host_phases.py > $OUT/host_phases.txt 2>&1
(echo '== LDS-DMA panel kernel (launches that aim at the whole chip)'; timeout -k 10 200 python tools/wgrad_thin_probe.py; echo '== register-staged 128 x 128 kernel'; SIHL_WGRAD_NO_THIN_DMA=1 timeout -k 10 200 python tools/wgrad_thin_probe.py) > $OUT/wgrad_thin_probe.txt 2>&1
# secondary configurations on the final tree
timeout -k 10 600 python tools/config_probe.py > $OUT/config_probe.txt 2>&1
# the bench line WITH this tree's measured traffic (profiles/r04_pmc_bench.json carries the same source stamp), CPU baseline at bs 32
python bench.py --cpu-sample 32 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
head -4 $OUT/off_bench_steady_state_summary.txt
head -c 700 $OUT/bench.json
