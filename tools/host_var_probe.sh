# host-issue time of the default bench, repeated, with and without Python's cyclic garbage collector
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hostvar
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --lean > gpurun_out/hostvar/gc_on_$i.json 2> /dev/null
  python -c "import gc, sys, runpy; gc.disable(); sys.argv=['bench.py','--no-cpu-baseline','--lean']; runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/hostvar/gc_off_$i.json 2> /dev/null
done
grep -o '"host_issue_ms_per_step": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/hostvar/*.json
