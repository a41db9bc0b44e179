set -x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 /root/repo/bench.py --steps 2 --warmup 3 --no-cpu-baseline --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/pmc_f.json 2> /root/repo/gpurun_out/pmc_f.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 /root/repo/bench.py --steps 2 --warmup 3 --no-cpu-baseline --profile-steps 1 --wgrad-stream off > /root/repo/gpurun_out/pmc_w.json 2> /root/repo/gpurun_out/pmc_w.err || exit 1
cd /root/repo
python profiles/pmc_summarize.py /tmp/pmc_f /tmp/pmc_w gpurun_out/r02_pmc_bench.json > gpurun_out/pmc_sum.txt 2>&1
tail -3 gpurun_out/pmc_sum.txt
python -c "
import json;d=json.load(open('gpurun_out/r02_pmc_bench.json'));print(d['steps_analysed']);print(d['kernels']['conv_igemm_dma_kernel'])"
