"""Developer probe (GPU box): per-shape conv forward times (training epilogue: ReLU + BatchNorm statistics) under several
builds of the library, one child process per build and round, alternating.
Usage: python tools/lib_conv_ab.py [--rounds N] lib1.so lib2.so ...      (child: --child)"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("L3 3x3", 32, 64, 64, 256, 256, 3), ("lat3 1x1", 32, 64, 64, 512, 256, 1), ("r2 1x1 128>512", 32, 64, 64, 128, 512, 1),
          ("r1 1x1 64>256", 32, 128, 128, 64, 256, 1), ("L4 3x3", 32, 32, 32, 256, 256, 3), ("mlp 1x1", 1, 1, 174592, 256, 256, 1),
          ("r2 3x3 128", 32, 64, 64, 128, 128, 3), ("r3 1x1 256>1024", 32, 32, 32, 256, 1024, 1)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    from sihl_amd import _C, ops
    lib = _C.lib()
    dt = torch.bfloat16
    w0 = torch.randn(8192, 8192, device="cuda", dtype=dt)
    for _ in range(300):
        w0 @ w0
    torch.cuda.synchronize()
    out = {}
    for name, N, H, W, Cin, Cout, K in SHAPES:
        xs = [torch.randn(N, H, W, Cin, device="cuda", dtype=dt) for _ in range(8)]
        w = torch.randn(Cout, K, K, Cin, device="cuda", dtype=dt) * 0.05
        fn = lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", stats_mode=2)  # noqa: E731
        for i in range(24):
            fn(i % 8)
        torch.cuda.synchronize()
        lib.sihl_profile_enable(1)
        for i in range(64):
            fn(i % 8)
        torch.cuda.synchronize()
        lib.sihl_profile_enable(0)
        cnt = lib.sihl_profile_records(0, _C.BF16, None, 0)
        buf = (ctypes.c_double * (3 * cnt))()
        lib.sihl_profile_records(0, _C.BF16, buf, cnt)
        ts = sorted(buf[3 * i] for i in range(cnt))
        out[name] = ts[len(ts) // 2] * 1e3  # us
    print("RESULT " + json.dumps(out))


def main():
    args = sys.argv[1:]
    rounds = 2
    if args[0] == "--rounds":
        rounds = int(args[1])
        args = args[2:]
    res = {lib: [] for lib in args}
    for _ in range(rounds):
        for lib in args:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, SIHL_HIP_LIB=os.path.abspath(lib)),
                               capture_output=True, text=True, timeout=300)
            line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
            if not line:
                raise RuntimeError(p.stdout[-1500:] + p.stderr[-1500:])
            res[lib].append(json.loads(line[0][7:]))
    print(f"{'':22s}" + "".join(f"{n[:14]:>15s}" for n, *_ in SHAPES))
    for lib in args:
        row = ""
        for n, *_ in SHAPES:
            v = sorted(r[n] for r in res[lib])
            row += f"{v[len(v) // 2]:15.1f}"
        print(f"{os.path.basename(lib):22s}" + row, flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
