#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward+backward (+ gradient all-reduce + clip + AdamW step),
ResNet50 + BiFPN(3-7, 256 ch, 3 layers) + ObjectDetection(80 classes), bs=32 per GPU, 3x512x512, bf16,
synthetic data (BASELINE.json configs[2] / SURVEY §8(d)).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One process per GPU; each rank owns its own 32-image shard (weak scaling) and gradients are averaged with
bucketed RCCL all-reduces overlapped with backward.  Rank 0 prints ONE JSON line.  Besides the contract's
fields it carries
  roofline     - the dominant kernel (bf16 implicit-GEMM conv on the matrix cores): algorithmic flops per launch
                 over its average launch duration, timed with HIP events on the launch stream DURING the timed
                 steps, against the dense bf16 MFMA peak;
  cpu_baseline - the CPU oracle (fp32 PyTorch restatement of the reference) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# HIP-graph replays through ROCm's "packet capture" path are not safe beside other device allocations (sihl_amd/__init__.py): the
# switch is set here, before anything can have initialised the HIP runtime
if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None:  # before the HIP runtime reads its flags: sihl_amd/__init__.py
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    _t = sys.modules.get("torch")
    os.environ["SIHL_GRAPH_ENV_EARLY"] = "0" if (_t is not None and _t.cuda.is_initialized()) else "1"

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec fwd+bwd, ResNet50+BiFPN+det-head bs=32 512², 1/2/4/8 MI355X"
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def synthetic_batch(batch, size, device, seed):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(batch, 3, size, size, generator=g)
    gt = torch.Generator().manual_seed(seed + 1)
    classes, boxes = [], []
    for b in range(batch):
        n = int(torch.randint(0, 9, (1,), generator=gt))
        if b == 1:
            n = 0  # always keep an image without objects (tests/heads/test_object_detection.py:42)
        xy = torch.rand(n, 2, generator=gt) * (size * 0.75)
        wh = 16 + torch.rand(n, 2, generator=gt) * (size * 0.25 - 16)
        boxes.append(torch.cat([xy, xy + wh], dim=1).to(device))
        classes.append(torch.randint(0, 80, (n,), generator=gt).to(device))
    images = images.to(device).contiguous(memory_format=torch.channels_last)
    return images, [{"classes": classes, "boxes": boxes}]


def build_model(ns, device, native_backbone=None):
    torch.manual_seed(0)
    kw = {} if native_backbone is None else {"native": native_backbone}
    backbone = ns.ResNetBackbone("resnet50", top_level=5, **kw)
    neck = ns.BiFPN(backbone.out_channels, 256, 3, 7, num_layers=3)
    head = ns.ObjectDetection(neck.out_channels, num_classes=80, bottom_level=3, top_level=7, num_channels=256)
    model = ns.SihlModel(backbone, neck, [head])
    return model.to(device).to(memory_format=torch.channels_last)


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_model():
    """The host CPU's model name (SURVEY 8d asks for it next to the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_batch, size, iters=5, warm=2):
    """The oracle (CPU port of the reference path) on a bounded sample: same model, same step, fp32; median of `iters`
    steps after `warm` warm-ups.  The sample is a smaller BATCH of the same 512x512 workload (images are
    independent, so images/s does not depend on it beyond cache effects): bs 32 costs ~8 s per step on 16 cores, 7 steps
    of it would double the bench's run time."""
    import types

    import oracle
    from sihl_amd.train import Trainer

    ns = types.SimpleNamespace(ResNetBackbone=oracle.ResNetBackbone, BiFPN=oracle.BiFPN,
                               ObjectDetection=oracle.ObjectDetection, SihlModel=oracle.SihlModel)
    cores = usable_cores()
    torch.set_num_threads(cores)
    if sample_batch >= 16:  # the full batch costs ~8 s per step: one warm-up and three timed steps keep the leg near 30 s
        iters, warm = min(iters, 3), min(warm, 1)
    model = build_model(ns, "cpu")
    trainer = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1)
    images, targets = synthetic_batch(sample_batch, size, "cpu", seed=0)
    for _ in range(warm):
        trainer.step(images, targets)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        trainer.step(images, targets)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": sample_batch / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle (CPU fp32 restatement) fwd+bwd+step, bs={sample_batch} of the same {size}x{size} "
                      f"workload, median of {iters} steps after {warm} warm-ups ({dt:.2f} s/step)"}


def north_star_cpu_baseline(sample_batch, size, iters=5, warm=2):
    """The north-star sub-metric on the host cores: the oracle's BiFPN(3-7) + ObjectDetection.forward (the CPU restatement of
    reference layers/bifpn.py:89-97 + heads/object_detection.py:99-122), eval, fp32, on the same seeded level list at a
    bounded batch (images are independent); median of `iters` forwards after `warm` warm-ups."""
    import oracle

    cores = usable_cores()
    torch.set_num_threads(cores)
    chans = [3, 64, 256, 512, 1024, 2048]
    torch.manual_seed(0)
    neck = oracle.BiFPN(chans, 256, 3, 7).eval()
    head = oracle.ObjectDetection(neck.out_channels, 80, 3, 7).eval()
    g = torch.Generator().manual_seed(1)
    levels = [torch.zeros(sample_batch, 3, size, size)] + [
        torch.randn(sample_batch, c, size // 2 ** l, size // 2 ** l, generator=g).contiguous(memory_format=torch.channels_last)
        for l, c in enumerate(chans) if l > 0]
    times = []
    with torch.no_grad():
        for i in range(warm + iters):
            t0 = time.perf_counter()
            head(neck(levels))
            if i >= warm:
                times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": sample_batch / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle BiFPN + ObjectDetection.forward (CPU fp32), bs={sample_batch} of the same {size}x{size} level "
                      f"list, median of {iters} forwards after {warm} warm-ups ({dt:.2f} s/forward)"}


def source_stamp():
    """sha256 over the kernel sources: profiles/*pmc*.json carry it, so a traffic figure measured on other code is
    recognisably stale."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "sihl_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "sihl_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_peaks(device):
    """On-box yardsticks: streaming copy (read + write bytes per second) and a vendor bf16 GEMM
    (torch.matmul = hipBLASLt / rocBLAS), both on random data, HIP events, best of 3 rounds."""
    n = 1 << 28  # 512 MiB of bf16 per buffer: past the 256 MiB Infinity Cache
    src = torch.randn(n, device=device, dtype=torch.bfloat16)
    dst = torch.empty_like(src)
    a = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
    bmat = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def best(fn, reps):
        fn()
        t = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = min(t, e0.elapsed_time(e1) / reps)
        return t * 1e-3

    t_copy = best(lambda: dst.copy_(src), 10)
    t_gemm = best(lambda: torch.matmul(a, bmat), 10)
    out = {"copy_GBps": 2 * n * 2 / t_copy / 1e9, "gemm_bf16_TFLOPs": 2 * 8192 ** 3 / t_gemm / 1e12,
           "what": "torch copy_ of 512 MiB (bytes read + written), torch.matmul 8192^3 bf16 (vendor GEMM), random data"}
    del src, dst, a, bmat
    # the vendor CONV on the flagship pyramid shape (L3 of the BiFPN: bs 32, 64x64, 256 -> 256, 3x3), channels_last bf16
    # through torch / MIOpen, next to the sihl kernel on the same tensors: what "a vendor conv reaches on the L3 shape"
    try:
        import torch.nn.functional as F
        from sihl_amd import ops
        x = torch.randn(32, 256, 64, 64, device=device, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(256, 256, 3, 3, device=device) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        flops = 2.0 * 32 * 64 * 64 * 256 * 256 * 9
        t_vendor = best(lambda: F.conv2d(x, w, padding=1), 10)
        xn, wn = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1)  # the same memory, NHWC / [O][KH][KW][I] views
        assert xn.is_contiguous() and wn.is_contiguous()
        t_sihl = best(lambda: ops.conv2d_raw(xn, wn, None, 1, 1, 1), 10)
        out["l3_conv_vendor_TFLOPs"] = flops / t_vendor / 1e12
        out["l3_conv_sihl_TFLOPs"] = flops / t_sihl / 1e12
        out["l3_conv_us"] = {"vendor": t_vendor * 1e6, "sihl": t_sihl * 1e6}
        out["what"] += "; l3_conv_*: 3x3 256->256 on 32x64x64 (154.6 GFLOP), torch F.conv2d (MIOpen) vs sihl_conv2d_fwd, same tensors"
    except Exception as e:  # a yardstick must not take the benchmark down
        out["l3_conv_error"] = repr(e)[:200]
    return out


def north_star_forward(model, device, dtype, batch, size, iters=20):
    """The north-star sub-metric, measured live: BiFPN + ObjectDetection.forward (eval, inference kernels: BatchNorm
    folded into the conv epilogues) on a seeded ResNet50-shaped level list, HIP events around `iters` forwards.
    Algorithmic work per image: 45.64 + 3.69 GFLOP, 85.8 + 8.5 MB (bf16) at 512^2."""
    if size != 512:
        return None
    chans = [3, 64, 256, 512, 1024, 2048]
    g = torch.Generator(device=device).manual_seed(1)
    levels = [torch.zeros(batch, 3, size, size, device=device)] + [
        torch.randn(batch, c, size // 2 ** l, size // 2 ** l, device=device, generator=g).to(dtype)
        .contiguous(memory_format=torch.channels_last) for l, c in enumerate(chans) if l > 0]
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for _ in range(3):
                model.heads[0](model.neck(levels))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                model.heads[0](model.neck(levels))
            e1.record()
            torch.cuda.synchronize()
    finally:
        model.train(was_training)
    ms = e0.elapsed_time(e1) / iters
    es = 2 if dtype == torch.bfloat16 else 4
    gflop, mbytes = batch * (45.64 + 3.69), batch * (42.9 + 4.26) * es
    peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else 157.3
    return {"what": f"BiFPN(3-7) + ObjectDetection.forward, eval, bs {batch}, {size}x{size}, seeded level list",
            "ms": ms, "images_per_s": batch / ms * 1e3, "achieved_tflops": gflop / ms, "peak_tflops": peak,
            "frac_of_mfma_roofline": gflop / ms / peak, "algorithmic_hbm_gbps": mbytes / ms,
            "frac_of_hbm_roofline": mbytes / ms / 8000.0, "bound": "mfma (AI 520 flop/B against a ridge of ~310)"}


def self_launch(n):
    """Start `n` ranks of this script under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) as a child
    process and wait for it.  The caller must not have initialised the GPU: nothing is re-executed in place."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, usable_cores() // n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4, help="batch of the CPU-baseline sample (see cpu_baseline)")
    ap.add_argument("--backbone", default="native", choices=["torch", "native"],
                    help="native (default) = ResNet50 residual stages on the sihl HIP kernels; torch = trunk on "
                         "PyTorch-ROCm/MIOpen")
    ap.add_argument("--graph", action="store_true",
                    help="N=1 only: capture the whole step into one HIP graph and replay it (single-stream: a "
                         "multi-branch graph replays slower than eager two-stream launches on this runtime)")
    ap.add_argument("--no-graph", action="store_true", help="(default since the wgrad side stream; kept for old commands)")
    ap.add_argument("--wgrad-stream", default="all", choices=["off", "small", "all"],
                    help="weight-gradient kernels on a second HIP stream beside the dgrad chain (Trainer.wgrad_stream)")
    ap.add_argument("--wgrad-target", type=int, default=0,
                    help="tuning: workgroups the LDS-DMA wgrad kernel's K-split aims for (0 = library default)")
    ap.add_argument("--profile-steps", type=int, default=3,
                    help="eager steps after the timed region over which the per-kernel HIP-event timings of the "
                         "roofline object are taken")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo on a one-GPU box)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="N=1 only: run the multi-GPU code path (process group of one rank, gradient buckets, hooks, "
                         "all-reduce calls on the collective's stream) on a single GPU")
    ap.add_argument("--graph-warmup-stream", default="off", choices=["off", "small", "all"],
                    help="debugging only, with --graph: stream mode of the eager warm-up steps before the capture (the "
                         "round-1 fault needed 'all'; a graph Trainer otherwise never uses a second stream)")
    ap.add_argument("--krot", type=int, default=-1,
                    help="A/B, tuning builds only: sihl_conv2d_krot value (stage stride between workgroups' K-loop starts; + 1000 x minimum stages)")
    ap.add_argument("--main-priority", type=int, default=0,
                    help="A/B: run the step on a user stream of this HIP priority (-1 = high) instead of the default stream")
    ap.add_argument("--no-fused-loss", action="store_true",
                    help="A/B: the detection loss as PyTorch device ops instead of the fused sihl_od_loss kernel")
    ap.add_argument("--lean", action="store_true",
                    help="profiling passes: skip the north-star probe, the sub-metrics and the yardsticks")
    ap.add_argument("--extra-stream", action="store_true",
                    help="debugging: create a second HIP stream and run one trivial kernel on it before the warm-up")
    ap.add_argument("--sync-warmup", action="store_true", help="debugging: synchronize after every warm-up step")
    ap.add_argument("--sync-steps", action="store_true", help="debugging: synchronize after every timed step")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (no GPU work): every rank joins the process group, rank 0 prints the number "
                         "of ranks seen - the CPU test of the self-launcher")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process has not touched the GPU (no torch.cuda call so far) and never
        # will - it starts N fresh ranks, lets rank 0's JSON line through on the inherited stdout and exits with
        # the launcher's return code (non-zero as soon as any rank fails)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        seen = torch.ones(1)
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(seen.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    # stdout carries the ONE JSON line and nothing else: libraries that print to file descriptor 1 (RCCL's version
    # banner at communicator creation, MIOpen notes) are pointed at stderr for the rest of the run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dev_index = 0 if args.same_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or args.rehearse_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import sihl_amd
    from sihl_amd import _C
    from sihl_amd.train import Trainer

    import types

    if args.no_fused_loss:
        from sihl_amd.heads import object_detection as _od
        _od.FUSED_LOSS = False
    hip_ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                                   ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
    model = build_model(hip_ns, device, native_backbone=args.backbone == "native")
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    use_graph = world == 1 and args.graph and not args.no_graph and not args.rehearse_dp
    trainer = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1,
                      autocast_dtype=amp, graph=use_graph, wgrad_stream=args.wgrad_stream,
                      force_buckets=args.rehearse_dp,
                      **({"_graph_warmup_stream": args.graph_warmup_stream} if use_graph else {}))
    images, targets = synthetic_batch(args.batch, args.size, device, seed=rank)
    # graph mode needs its eager warm-up steps + the capture before the timed region
    n_warm = max(args.warmup, Trainer.GRAPH_WARMUP + 1) if use_graph else args.warmup

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = (lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)) if os.environ.get("SIHL_BENCH_TRACE") \
        else (lambda m: None)

    lib = _C.lib()
    if args.wgrad_target:
        from sihl_amd import ops
        ops.SIDE_WGRAD_TARGET = args.wgrad_target
    if os.environ.get("SIHL_NO_STRIDED_CLASSES"):  # A/B switch: zero-dilated read for the 3x3 stride-2 dgrads
        lib.sihl_conv2d_strided_classes_enable(0)
    if args.extra_stream:
        extra = torch.cuda.Stream(device=device)
        with torch.cuda.stream(extra):
            _dummy = torch.zeros(1024, device=device).add_(1)
        torch.cuda.synchronize()
    if args.krot >= 0 and _C.lib().sihl_conv2d_krot(args.krot) != 0:
        raise SystemExit("--krot needs a `make TUNING=1` library (SIHL_HIP_LIB=...): the shipped one has no tuning state")
    if args.main_priority:
        torch.cuda.synchronize()
        _main = torch.cuda.Stream(device=device, priority=args.main_priority)
        torch.cuda.set_stream(_main)
    for i in range(n_warm):
        trainer.step(images, targets)
        if args.sync_warmup:
            torch.cuda.synchronize()
        trace(f"warm-up step {i} issued")
    sync()
    trace("warm-up done")
    allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.step(images, targets)
        if args.sync_steps:
            torch.cuda.synchronize()
    t_enqueued = time.perf_counter() - t0  # host time to ISSUE the steps (the device is still working): if this is
    sync()                                 # close to the measured time, the step is host-bound
    dt = time.perf_counter() - t0
    trace("timed steps done")
    device_allocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs0  # hipMalloc calls inside the timed
    final_loss = float(loss)                                                              # region (each one stalls the host)
    # Per-kernel timing for the roofline object: HIP events around every matrix-core launch, over extra eager steps of
    # the same workload right after the timed region.  Timing events are queue barriers (~3 us of GPU time each, 700
    # per step = 2 ms, 5 % of a step) and a graph replay cannot carry them, so the timed steps themselves run without.
    profiled_steps = max(1, args.profile_steps)
    trainer.wgrad_stream = "off"  # per-kernel durations are taken with every launch alone on the device
    lib.sihl_profile_enable(1)
    for _ in range(profiled_steps):
        trainer._eager_step(images, targets)
    sync()
    trace("profiled eager steps done")
    lib.sihl_profile_enable(0)

    fwd = None
    sub = None
    peaks = None
    if world == 1 and rank == 0 and not args.rehearse_dp and not args.lean:
        fwd = north_star_forward(model, device, amp or torch.float32, args.batch, args.size)
        trace("north-star forward probe done")
        # sub-metrics of the same step: forward only, and forward + backward without clip / optimizer
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        model.train()
        trainer.wgrad_stream = args.wgrad_stream
        n_sub = 5

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n_sub):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n_sub

        def fwd_only():
            with torch.no_grad():
                trainer.forward_loss(images, targets)

        def fwd_bwd():
            trainer.optimizer.zero_grad(set_to_none=True)
            loss_, _ = trainer.forward_loss(images, targets)
            trainer._backward(loss_)

        sub = {"fwd_only_ms": timed(fwd_only), "fwd_bwd_no_opt_ms": timed(fwd_bwd),
               "what": "training-mode forward + loss under no_grad; forward + backward without clip / AdamW / weight "
                       f"preparation; eager, mean of {n_sub} after 1 warm-up"}
        peaks = measured_peaks(device)
        trace("sub-metrics and yardsticks done")

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    seen = torch.ones(1, device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    dt = float(t.item())
    ranks_seen = int(seen.item())
    # per-rank host issue time and the time the step's stream stood still for the all-reduces: a poor 1 -> N curve is then
    # attributable (host-bound ranks / exposed collectives) from this one line
    dp_stats = None
    if world > 1:
        wait_ms = trainer.averager.wait_ms() if trainer.averager.active else 0.0
        v = torch.tensor([t_enqueued / args.steps * 1e3, wait_ms], device=device, dtype=torch.float64)
        vmax, vsum = v.clone(), v.clone()
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vsum)
        dp_stats = {"host_issue_ms_per_step_max": float(vmax[0]), "host_issue_ms_per_step_mean": float(vsum[0]) / world,
                    "allreduce_wait_ms_per_step_max": float(vmax[1]), "allreduce_wait_ms_per_step_mean": float(vsum[1]) / world,
                    "gradient_buckets": len(trainer.averager.buckets),
                    "what": "host time to enqueue a step; time the step's stream waited in GradientAverager.finish() for "
                            "the bucketed all-reduces (the part backward did not hide), mean of the timed steps per rank"}

    # roofline of the dominant kernel: matrix-core conv (forward / dgrad / linear launches)
    dt_code = _C.BF16 if args.dtype == "bf16" else _C.F32
    n, ms, fl, by = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.sihl_profile_collect(0, dt_code, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
    nw, msw, flw, byw = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.sihl_profile_collect(1, dt_code, ctypes.byref(nw), ctypes.byref(msw), ctypes.byref(flw), ctypes.byref(byw))
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3

    def per_launch_roofline(slot):
        """sum over launches of max(flops / MFMA peak, algorithmic bytes / HBM peak) divided by the measured time: the
        thin ResNet layers are HBM-bound even at the matrix-core kernel (AI 50-200 flop/B against a ridge of ~310)."""
        cnt = lib.sihl_profile_records(slot, dt_code, None, 0)
        if cnt <= 0:
            return None
        buf = (ctypes.c_double * (3 * cnt))()
        lib.sihl_profile_records(slot, dt_code, buf, cnt)
        ideal = sum(max(buf[3 * i + 1] / (peak * 1e12), buf[3 * i + 2] / 8e12) for i in range(cnt))
        spent = sum(buf[3 * i] for i in range(cnt)) * 1e-3
        return ideal / spent if spent > 0 else None

    roofline = None
    # HBM traffic of the conv kernel per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE, WRITE_SIZE;
    # profiles/pmc_summarize.py applies the gfx950 correction), so the figure comes from the committed summary of
    # those passes over this same command, not from this run
    pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_bench.json") and f[:1] == "r")
    pmc_name = pmc_files[-1] if pmc_files else "r03_pmc_bench.json"  # the newest round's summary
    traffic, traffic_note = None, f"no PMC summary for this source tree (profiles/{pmc_name})"
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
        # every kernel family the profiler's conv slot times: the tile kernel, the small-level kernel and the split-K
        # finishing launch (one profiled "launch" = one conv call); per-launch traffic = their HBM bytes per step / calls per step
        fam = [pmc_all["kernels"][k] for k in ("conv_igemm_dma_kernel", "conv_igemm_kernel", "conv_small_kernel", "conv_pyr_kernel",
                                                "conv_halo_kernel", "conv_splitk_epilogue_kernel") if k in pmc_all["kernels"]]
        pmc_total_mb = sum(f["traffic_MB_per_launch"] * f["launches_per_step"] for f in fam)
        if pmc_all.get("source_stamp") != source_stamp():
            traffic_note = (f"profiles/{pmc_name} was measured on other kernel sources (stamp "
                            f"{pmc_all.get('source_stamp')} != {source_stamp()}): stale, not reported")
        elif args.dtype == "bf16" and args.batch == 32 and args.size == 512:
            traffic = pmc_total_mb * 1e6 / (n.value / profiled_steps) if n.value else None
            traffic_note = ("HBM bytes per conv call (FETCH_SIZE x2 + WRITE_SIZE of the tile, halo, pyramid-top and split-K finishing "
                            "kernels per step / conv calls per step; separate rocprofv3 --pmc passes over this "
                            f"command, profiles/{pmc_name}, same source stamp); algorithmic bytes per launch: "
                            "avg_algorithmic_mb_per_launch")
    except (OSError, KeyError, ValueError):
        pass
    if n.value:
        achieved = fl.value / (ms.value * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "NHWC implicit-GEMM conv on the matrix cores: fwd / dgrad / linear (conv_igemm_dma_kernel tiles, conv_halo_kernel on 64- / 32-wide maps, conv_pyr_kernel on the pyramid's top levels)",
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                    "traffic_note": traffic_note,
                    "avg_algorithmic_mb_per_launch": by.value / n.value / 1e6,
                    "frac_of_per_launch_roofline": per_launch_roofline(0),  # each launch against min(MFMA, HBM) bound
                    "launches_per_step": n.value / profiled_steps, "avg_launch_us": ms.value * 1e3 / n.value,
                    "avg_gflop_per_launch": fl.value / n.value / 1e9,
                    "kernel_ms_per_step": ms.value / profiled_steps,
                    "measured_over": f"{profiled_steps} eager steps of the same workload right after the timed region "
                                     "(timing events cost ~5 % of a step and cannot ride in a graph replay)",
                    "wgrad": {"achieved": (flw.value / (msw.value * 1e-3) / 1e12) if nw.value else None,
                              "frac_of_per_launch_roofline": per_launch_roofline(1),
                              "launches_per_step": nw.value / profiled_steps,
                              "kernel_ms_per_step": msw.value / profiled_steps}}

    if rank == 0:
        out = {
            "metric": METRIC, "value": args.batch * world * args.steps / dt, "unit": "images/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "host_issue_ms_per_step": t_enqueued / args.steps * 1e3,
            "device_allocs_in_timed_region": device_allocs,
            "steps": args.steps, "warmup": n_warm, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "ResNet50 + BiFPN(3-7,256ch,3 layers) + ObjectDetection(80 cls) training step: "
                                   "fwd + bwd + grad all-reduce + clip(0.1) + AdamW, random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "image": f"3x{args.size}x{args.size}", "parallelism": f"dp{world}",
                       "backbone": "resnet50 trunk on " + ("sihl HIP kernels (bf16: incl. the 7x7 stem conv, csrc/stem.hip)"
                                                            if args.backbone == "native" else "PyTorch-ROCm (MIOpen/CK)"),
                       "wgrad_stream": "off" if use_graph else args.wgrad_stream,
                       "execution": "one HIP graph replay per step" if use_graph else
                       "eager launches, weight gradients on a second HIP stream" if args.wgrad_stream != "off" else "eager launches",
                       "final_loss": final_loss},
            "roofline": roofline,
        }
        if dp_stats is not None:
            out["data_parallel"] = dp_stats
        if fwd is not None:
            out["north_star_forward"] = fwd
        if sub is not None:
            out["sub_metrics"] = sub
        if peaks is not None:
            out["measured_peaks"] = peaks
            if roofline is not None:
                roofline["frac_of_measured_gemm"] = roofline["achieved"] / peaks["gemm_bf16_TFLOPs"]
            if fwd is not None:
                fwd["frac_of_measured_gemm"] = fwd["achieved_tflops"] / peaks["gemm_bf16_TFLOPs"]
                fwd["frac_of_measured_copy"] = fwd["algorithmic_hbm_gbps"] / peaks["copy_GBps"]
        out["source_stamp"] = source_stamp()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.size)
            if "north_star_forward" in out and args.size == 512:
                out["north_star_forward"]["cpu_baseline"] = north_star_cpu_baseline(args.cpu_sample, args.size)
                out["north_star_forward"]["vs_cpu_baseline"] = out["north_star_forward"]["images_per_s"] / \
                    out["north_star_forward"]["cpu_baseline"]["value"]
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
