# MFMA / LDS counters of the two matrix-core kernels on the flagship L3 shape (profiles/pmc_conv.py), one rocprofv3 pass
# per counter group (--kernel-trace only beside --pmc).
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag -o c -- python3 /root/repo/profiles/pmc_conv.py > /dev/null 2> /root/repo/gpurun_out/pmc_mfma_$tag.err || echo "FAILED $grp"
done
cd /root/repo
python - <<'PY' > gpurun_out/pmc_mfma_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'conv_halo' if 'conv_halo' in n else 'conv_igemm_dma' if 'conv_igemm_dma' in n else 'conv_wgrad_dma' if 'conv_wgrad_dma' in n else None
        if k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())})
PY
cat gpurun_out/pmc_mfma_summary.txt
