"""Developer probe (GPU box): A/B of two builds of the library on the same box, alternating processes (A B A B ...).
Each child is `bench.py --no-cpu-baseline` under SIHL_HIP_LIB; compared: the training step (ms_per_step), the north-star
forward (north_star_forward.ms) and the matrix-core seconds per step of the roofline object.
Usage: python tools/lib_ab.py <libA.so> <libB.so> [...] [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib):
    env = dict(os.environ, SIHL_HIP_LIB=os.path.abspath(lib))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "3"],
                       env=env, capture_output=True, text=True, timeout=900)
    for line in reversed(p.stdout.splitlines()):
        if line.startswith("{"):
            j = json.loads(line)
            return {"step_ms": j["ms_per_step"], "ns_ms": j["north_star_forward"]["ms"],
                    "conv_ms": j["roofline"]["kernel_ms_per_step"], "wgrad_ms": j["roofline"]["wgrad"]["kernel_ms_per_step"],
                    "l3_us": j["measured_peaks"]["l3_conv_us"]["sihl"]}
    raise RuntimeError(p.stdout[-2000:] + p.stderr[-2000:])


def main():
    libs = [a for a in sys.argv[1:] if not a.isdigit()]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    res = {lib: [] for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            r = run(lib)
            res[lib].append(r)
            print(os.path.basename(lib), r, flush=True)
    for lib in libs:
        for k in res[lib][0]:
            v = sorted(x[k] for x in res[lib])
            print(f"{os.path.basename(lib):24s} {k:9s} median {v[len(v) // 2]:8.3f}   all {' '.join(f'{x:.3f}' for x in v)}")


if __name__ == "__main__":
    main()
