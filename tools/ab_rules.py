"""Developer A/B (GPU box): the bench under different dispatch-rule masks, same process order repeated."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); from sihl_amd import _C; _C.lib().sihl_conv2d_rules_off(%d); _C.lib().sihl_conv2d_splitk_enable(%d); import bench; sys.argv=['bench.py','--no-cpu-baseline']; bench.main()"
for rep in range(2):
    for mask, sk in ((0, 1), (1, 1), (2, 1), (3, 1), (3, 0)):
        r = subprocess.run([sys.executable, "-c", code % (ROOT, mask, sk)], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {}
        print(f"rules_off={mask} splitk={sk}: {d.get('value', 0):.1f} img/s {d.get('ms_per_step', 0):.2f} ms", flush=True)
