# A/B of the per-channel passes' sweep order (SIHL_EW_ORDER: 0 = all ascending, 3 = affine_act + norm_bwd_reduce from the end):
# rocprofv3 kernel stats of the single-stream bench, twice each, plus default-mode step times.
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ew_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do
  for o in 0 3; do
    SIHL_EW_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ew_${o}_$r -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/prof_${o}_$r.json 2> $GRAFT_REPO_ROOT/$OUT/prof_${o}_$r.err || exit 1
    cp $(find /tmp/ew_${o}_$r -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$OUT/stats_${o}_$r.csv
  done
done
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for o in 0 3; do
    SIHL_EW_ORDER=$o python bench.py --no-cpu-baseline --lean > $OUT/bench_${o}_$r.json 2> $OUT/bench_${o}_$r.err
  done
done
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_*.json
