# A/B of the LDS-DMA weight-gradient kernel's wave count (SIHL_WGRAD_WAVES=8: 2 x 4 waves of 128 x 64; default 16 waves of 64 x 64)
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wgrad_waves
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for o in 8 16; do
  SIHL_WGRAD_WAVES=$o rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ww_$o -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --lean --profile-steps 1 --wgrad-stream off > $GRAFT_REPO_ROOT/$OUT/prof_$o.json 2> $GRAFT_REPO_ROOT/$OUT/prof_$o.err || exit 1
  cp $(find /tmp/ww_$o -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$OUT/stats_$o.csv
done
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for o in 8 16; do
    SIHL_WGRAD_WAVES=$o python bench.py --no-cpu-baseline --lean > $OUT/bench_${o}_$r.json 2> $OUT/bench_${o}_$r.err
  done
done
grep wgrad_dma $OUT/stats_8.csv $OUT/stats_16.csv | cut -c1-200
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_*.json
