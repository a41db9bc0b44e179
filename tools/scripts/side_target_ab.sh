mkdir -p gpurun_out/r04i
for r in 1 2; do for t in 128 96 112 144; do echo "== side target $t"; SIHL_SIDE_WGRAD_TARGET=$t timeout -k 10 200 python bench.py --no-cpu-baseline --lean 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['value'])"; done; done
