# A/B of the residual tails' ReLU mask (SIHL_TAIL_MASK_BITS=0: the backward re-reads y; 1: mask bytes from the forward):
# rocprofv3 kernel stats of the single-stream bench + default-mode step times.
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tail_bits
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for o in 0 1; do
  SIHL_TAIL_MASK_BITS=$o rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tb_$o -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/prof_$o.json 2> $GRAFT_REPO_ROOT/$OUT/prof_$o.err || exit 1
  cp $(find /tmp/tb_$o -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$OUT/stats_$o.csv
done
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for o in 0 1; do
    SIHL_TAIL_MASK_BITS=$o python bench.py --no-cpu-baseline --lean > $OUT/bench_${o}_$r.json 2> $OUT/bench_${o}_$r.err
  done
done
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_*.json
