"""Device ops of the sihl hot path: thin wrappers over the C-ABI + ``torch.autograd.Function`` glue.

PyTorch is plumbing here (device memory from the caching allocator, the current HIP stream,
autograd bookkeeping); every forward/backward computation below is a hand-written HIP kernel from
``libsihl_hip.so``.  Activations are NHWC in memory (``torch.channels_last`` for the NCHW-logical
tensors the reference API exchanges), fp32 or bf16; statistics, weights' gradients and norm
parameters are fp32.
"""
import ctypes
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from sihl_amd import _C
from sihl_amd._C import ACT, BF16, F32, check

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _dt(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"sihl_amd ops support float32 and bfloat16 activations, got {t.dtype}") from None


def _require_gpu(t: Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("sihl_amd ops run on a HIP device only (no CPU fallback); got a CPU tensor")


def _p(t: Optional[Tensor]):
    # a plain int: the prototypes in _C declare c_void_p, ctypes converts (and None is NULL) - building a c_void_p object
    # per pointer was ~1 ms of host time per training step (2 200 pointers)
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current HIP stream (the fast path avoids building a Stream object per launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_WS = {}
_WS_RETIRED = []
_SIZES = {}


def _sized(name: str, *args):
    """Result of one of the library's pure size helpers (`*_ws_bytes`, `*_stat_rows`: functions of the shape alone),
    memoised: a training step asked the same ~150 questions through ctypes every step (~0.7 ms of host time)."""
    key = (name,) + args
    v = _SIZES.get(key)
    if v is None:
        v = _SIZES[key] = getattr(_C.lib(), name)(*args)
    return v

FUSION_GACC_FLOATS = 4 * 4096  # SIHL_FUSION_GACC_FLOATS of include/sihl_hip.h: scratch of the fusion nodes' backward
_SPLIT_TAIL_BWD = bool(os.environ.get("SIHL_SPLIT_TAIL_BWD"))
_ATEN_STEM_WGRAD = bool(os.environ.get("SIHL_ATEN_STEM_WGRAD"))
TAIL_MASK_BITS = os.environ.get("SIHL_TAIL_MASK_BITS", "1") != "0"  # A/B / test switch: 0 = the tail's backward re-reads y

# ---- BatchNorm step counters: one multi-tensor add per training step instead of one tiny kernel per layer
_DEFERRED_COUNTERS = None


def bump_counter(counter: Tensor) -> None:
    """``num_batches_tracked += 1`` now, or collected for ONE fused add when inside deferred_bn_counters()."""
    if _DEFERRED_COUNTERS is not None:
        _DEFERRED_COUNTERS.append(counter)
    else:
        counter += 1


class deferred_bn_counters:
    """Context manager: the ~100 per-layer counter increments of a training forward become one _foreach_add_."""

    def __enter__(self):
        global _DEFERRED_COUNTERS
        self._outer = _DEFERRED_COUNTERS
        _DEFERRED_COUNTERS = []
        return self

    def __exit__(self, *exc):
        global _DEFERRED_COUNTERS
        counters, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, self._outer
        if counters and exc[0] is None:
            torch._foreach_add_(counters, 1)
        return False


def workspace(nbytes: int, device, stream=None) -> Tensor:
    """Grow-only scratch buffer per device and stream (kernels are stream-ordered, so one buffer is shared).  stream: raw
    handle of the stream the kernel will be launched on when that is not torch's current stream (side-stream launches by
    handle: they must NOT share the current stream's buffer)."""
    key = (device.index, _stream() if stream is None else stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and stream is not None:
            # outgrown while kernels of ANOTHER stream may still use it: the block belongs to the current stream's pool, so
            # it is kept (a handful of buffers during the first steps) rather than handed back under a running kernel
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ---- weight gradients on a side stream
# In backward the critical chain is dgrad -> norm backward -> dgrad ...; a layer's weight gradient hangs off that chain
# (nothing reads it before the optimizer / the gradient all-reduce).  At the small pyramid levels and in the heads both
# kernels leave most of the 256 CUs idle, so the wgrad launches go to a second HIP stream: fork after the tensor they
# read is produced, one join before the gradients are consumed.  The operands (x, dz) are main-stream allocations the
# side stream reads, so they are held alive until the join (the caching allocator would otherwise hand their blocks
# to later main-stream kernels while the wgrad is still queued).
class _Side:
    __slots__ = ("stream", "handle", "holds", "mode", "dirty", "seen")

    def __init__(self, stream, mode):
        self.stream, self.holds, self.mode, self.dirty, self.seen = stream, [], mode, False, set()
        self.handle = stream.cuda_stream  # kernels are launched on it BY HANDLE: entering torch.cuda.stream(...) per launch
        #                                   cost ~10 us of host time x 140 launches per step


_SIDE: Optional[_Side] = None
_SIDE_STREAMS = {}
WGRAD_SIDE_MAX_PIXELS = {"off": 0, "small": 32 * 32 * 32, "all": 1 << 62}
# K-split aim of weight gradients launched beside the dgrad chain (the `target` argument of sihl_conv2d_wgrad: LDS-DMA
# kernel + 10000 x register-staged kernels): they should not claim the whole chip - 128 workgroups instead of one per
# CU write half the fp32 partial slabs and leave CUs to the main stream (flagship step, same box: 35.4-35.8 ->
# 34.4-34.7 ms; 64-96 workgroups about the same, 32: 40.2 ms)
SIDE_WGRAD_TARGET = 128 + 10000 * 128
if os.environ.get("SIHL_SIDE_WGRAD_TARGET"):  # developer A/B: "<LDS-DMA aim>[,<register-staged aim>]"
    _t = [int(v) for v in os.environ["SIHL_SIDE_WGRAD_TARGET"].split(",")]
    SIDE_WGRAD_TARGET = _t[0] + 10000 * (_t[1] if len(_t) > 1 else 128)
WGRAD_TARGET = 0  # K-split aim of weight gradients on the main stream (0 = library default, one workgroup per CU)


def side_stream(device, priority: int = 0):
    """The per-device side stream (one per priority), created on first use."""
    dev = torch.device(device).index if not isinstance(device, int) else device
    if dev is None:
        dev = torch.cuda.current_device()
    key = (dev, priority)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return _SIDE_STREAMS[key]


class wgrad_side_stream:
    """Context manager around ``loss.backward()``.  mode: "off", "small" (layers with at most 32 768 output pixels: the
    launches that cannot fill the chip) or "all".

    priority: HIP stream priority of the side stream.  Normal-priority streams share a small pool of hardware queues
    (GPU_MAX_HW_QUEUES, default 4) that the runtime hands out in creation order; once a process group exists (RCCL
    and torch's collective stream take queues first) the side stream can land on the MAIN stream's hardware queue,
    where the fork/join edges cost more than they gain (measured on one MI355X, multi-GPU code path on one rank:
    40.1 ms/step against 37.9 single-stream; with 2, 3 or 8 hardware queues 36.0-36.6).  A high-priority stream
    (-1) gets a queue of its own whatever else exists (36.2 ms); without a process group the normal-priority stream
    is the better one (35.1-35.4 against 36.0), so the Trainer picks by ``dist.is_initialized()``."""

    def __init__(self, mode: str = "all", device=None, priority: int = 0):
        self.mode, self.device, self.priority = mode, device, priority

    def __enter__(self):
        global _SIDE
        self._outer = _SIDE
        if self.mode != "off" and torch.cuda.is_available():
            dev = torch.cuda.current_device() if self.device is None else torch.device(self.device).index
            _SIDE = _Side(side_stream(dev, self.priority), self.mode)
        else:
            _SIDE = None
        return self

    def __exit__(self, *exc):
        global _SIDE
        join_side_stream()
        _SIDE = self._outer
        return False


def _guard_shared_parameters(keys) -> None:
    """A parameter used MORE THAN ONCE in one backward (a module called twice per step): autograd then adds its gradients -
    a kernel on the main stream - and must not start before the side stream has written them.  `keys`: storage addresses
    of the parameters whose gradients this backward node has just queued on the side stream; from a parameter's second
    appearance on, the main stream waits for the side stream before the node returns.  (A parameter used once is handed to
    AccumulateGrad as it is, no kernel: no wait - the common case costs a set lookup.)"""
    side = _SIDE
    if side is None:
        return
    # (a clean side stream has nothing to wait for, but the keys are remembered all the same: in "small" mode a shared
    # module's FIRST weight gradient can run on the main stream with nothing yet queued on the side stream)
    if side.dirty and any(k in side.seen for k in keys):
        torch.cuda.current_stream().wait_stream(side.stream)
    side.seen.update(keys)


def side_stream_history() -> bool:
    """True once any side stream has been created in this process (eager two-stream steps have run)."""
    return bool(_SIDE_STREAMS)


def side_stream_in_use():
    """The wgrad side stream if weight gradients have been queued on it since the last join, else None."""
    return _SIDE.stream if _SIDE is not None and _SIDE.dirty else None


def join_side_stream() -> None:
    """The current stream waits for every weight gradient queued on the side stream so far, then the operands the side
    stream read (x, dout, dw) are let go.

    Releasing them at the ENQUEUE of the wait is safe: they are main-stream allocations, the caching allocator never hands
    a block of one stream's pool to another stream (short of hipFree, which synchronises the device), and every
    main-stream kernel that could reuse such a block is queued behind this wait - hence behind the side stream's last
    reader.  (For most of round 2 they were kept until an event behind the side stream's last kernel had COMPLETED.  The
    host runs one to two steps ahead of the device, so a whole step's activations and gradients - 9 GiB at the benchmark
    size - stayed alive into the next forward: reserved memory grew 10 -> 20 -> 30 GiB in bursts of 150-250 hipMalloc
    calls whenever the host got further ahead than before (tools/alloc_probe.py), 15-30 per step inside the timed region,
    and on a box whose hipMalloc was slow the step went from 31 to 42 ms.)"""
    if _SIDE is not None and _SIDE.dirty:
        torch.cuda.current_stream().wait_stream(_SIDE.stream)
        _SIDE.holds = []
        _SIDE.dirty = False
        _SIDE.seen = set()


def nhwc(x: Tensor) -> Tensor:
    """(N,C,H,W)-logical tensor -> contiguous (N,H,W,C) view (copies only if not channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def nchw_view(x_nhwc: Tensor) -> Tensor:
    return x_nhwc.permute(0, 3, 1, 2)


def weight_khwc(w: Tensor, dtype: torch.dtype) -> Tensor:
    """(O,I,KH,KW) parameter -> contiguous [O][KH][KW][I] in the compute dtype."""
    prep = prepared(w, dtype)
    if prep is not None:
        return prep.w
    if not torch.is_grad_enabled() and w.is_cuda:
        # inference outside a Trainer (model.eval()(x) in bf16 with no PreparedWeights): the operand copy is made once per
        # weight version, not once per forward (~60 cast launches per BiFPN + head forward otherwise).  Same key as
        # PreparedWeights: storage address + torch's version counter (writes through .data do not bump it).
        key = (dtype, w.data_ptr(), w._version, _PARAM_GEN)
        hit = getattr(w, "_sihl_cast", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        out = w.detach().permute(0, 2, 3, 1).to(dtype).contiguous()
        if not torch.cuda.is_current_stream_capturing():
            w._sihl_cast = (key, out)
        return out
    return w.detach().permute(0, 2, 3, 1).to(dtype).contiguous()


# ---- per-step operand copies of all weights (one kernel per step instead of a cast + a flip/transpose per layer)
class _Prepared:
    __slots__ = ("w", "wt", "version", "dtype")


def prepared(weight: Tensor, dtype: torch.dtype):
    """The up-to-date prepared operands of ``weight`` (see PreparedWeights), or None."""
    prep = getattr(weight, "_sihl_prepared", None)
    if prep is not None and prep.dtype == dtype and prep.version == weight._version:
        return prep
    return None


class _WeightDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("w", ctypes.c_void_p), ("wt", ctypes.c_void_p),
                ("so", ctypes.c_int64), ("si", ctypes.c_int64), ("sky", ctypes.c_int64), ("skx", ctypes.c_int64),
                ("O", ctypes.c_int32), ("Op", ctypes.c_int32), ("KH", ctypes.c_int32), ("KW", ctypes.c_int32),
                ("I", ctypes.c_int32), ("flip", ctypes.c_int32), ("first_block", ctypes.c_int64),
                ("first_tblock", ctypes.c_int64)]


class PreparedWeights:
    """bf16 operand copies of every Conv2d / Linear weight of a model, refreshed by ONE kernel launch.

    For each fp32 master weight the conv kernels need w [O][KH][KW][I] (forward, wgrad) and the flipped /
    transposed wt [I][KH][KW][O] (dgrad) in the compute dtype.  ``refresh()`` rewrites all of them (call it after
    every optimizer step; Trainer does) and stamps each parameter's version; conv_block / linear pick the copies up
    while the stamp matches ``weight._version`` and fall back to their own per-layer cast otherwise (a parameter
    changed behind the trainer's back is never read stale - as long as the change bumps the tensor's version
    counter: in-place ops, ``copy_``, ``load_state_dict`` do; writes through ``.data`` do not, so call ``refresh()``
    after those)."""

    def __init__(self, model: torch.nn.Module, dtype: torch.dtype = torch.bfloat16):
        if dtype != torch.bfloat16:
            raise ValueError("prepared operands are bf16 (fp32 kernels read the master weights directly)")
        self.dtype = dtype
        weights = []
        for m in model.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)) and m.weight.is_cuda and m.weight.dtype == torch.float32:
                if all(m.weight is not w for w in weights):
                    weights.append(m.weight)
        self.weights = weights
        self._table = None
        if not weights:
            return
        geo = []
        for p in weights:
            O, I = p.shape[0], p.shape[1]
            KH, KW = (p.shape[2], p.shape[3]) if p.dim() == 4 else (1, 1)
            Op = (O + 7) // 8 * 8
            geo.append((O, Op, KH, KW, I))
        total = sum(Op * KH * KW * I for _, Op, KH, KW, I in geo)
        dev = weights[0].device
        self._flat = torch.zeros(2 * total + 16, dtype=dtype, device=dev)
        descs = (_WeightDesc * len(weights))()
        off, block, tblock = 0, 0, 0
        for k, (p, (O, Op, KH, KW, I)) in enumerate(zip(weights, geo)):
            n = Op * KH * KW * I
            w = self._flat[off: off + n].view(Op, KH, KW, I)
            wt = self._flat[total + off: total + off + n].view(I, KH, KW, Op)
            off += n
            prep = _Prepared()
            prep.w, prep.wt, prep.version, prep.dtype = w, wt, -1, dtype
            p._sihl_prepared = prep
            st = p.stride()
            d = descs[k]
            d.src, d.w, d.wt = p.data_ptr(), w.data_ptr(), wt.data_ptr()
            d.so, d.si = st[0], st[1]
            d.sky, d.skx = (st[2], st[3]) if p.dim() == 4 else (0, 0)
            d.O, d.Op, d.KH, d.KW, d.I = O, Op, KH, KW, I
            d.flip = 1 if p.dim() == 4 else 0
            d.first_block, d.first_tblock = block, tblock
            block += (n + 2047) // 2048
            tblock += KH * KW * ((Op + 63) // 64) * ((I + 63) // 64)
        self._blocks, self._tblocks = block, tblock
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        self._table = raw.to(dev)
        self._ptrs = [p.data_ptr() for p in weights]
        self.refresh()

    def refresh(self) -> None:
        if self._table is None:
            return
        if [p.data_ptr() for p in self.weights] != self._ptrs:
            raise RuntimeError("a prepared weight was re-allocated; build a new PreparedWeights")
        rc = _C.lib().sihl_weight_prepare(_p(self._table), len(self.weights), self._blocks, self._tblocks, _stream())
        check(rc, "sihl_weight_prepare")
        for p in self.weights:
            p._sihl_prepared.version = p._version


# ----------------------------------------------------------------------------- raw kernels
def conv2d_raw(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, stride: int = 1, pad: int = 0, dil: int = 1,
               act: Optional[str] = None, pre: Optional[Tuple[Tensor, Tensor]] = None,
               post: Optional[Tuple[Tensor, Tensor]] = None, stats_mode: int = 0,
               out: Optional[Tensor] = None, out_image_stride: int = 0):
    """x (N,H,W,Cin) contiguous, w (Cout,KH,KW,Cin) contiguous, same dtype.  Returns (y, stats) with
    stats = fp32 [rows][2][Cout] partial sums when stats_mode != 0."""
    _require_gpu(x)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    if (stats_mode and out is None and W <= 8 and H == W and KH == 3 and KW == 3 and stride == 1 and pad == 1 and dil == 1
            and pyr_conv_supported(N, W, Cin, Cout, 0, x.dtype) and _C.lib().sihl_conv2d_small_mode() == 1):
        # 8x8 / 4x4 maps with BatchNorm statistics: conv_pyr.hip writes one partial row per TILE (an 8x8 image, four 4x4
        # images) instead of the generic one per 128 pixels, so it is reached through its own entry; sihl_bn_finalize takes
        # any number of rows
        return pyr_conv_raw(w, x=x, bias=bias, act=act, pre=pre, post=post, stats_mode=stats_mode)[:2]
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    stats, stats_bytes = None, 0
    if stats_mode:
        rows = _sized("sihl_conv2d_stat_rows", N * Ho * Wo)
        stats = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
        stats_bytes = stats.numel() * 4
    lib = _C.lib()
    ws_bytes = _sized("sihl_conv2d_ws_bytes", N, H, W, Cin, Cout, KH, KW, stride, pad, dil)  # > 0: tiny level, split-K
    ws = workspace(ws_bytes, x.device) if ws_bytes else None
    rc = lib.sihl_conv2d_fwd_ws(
        _p(x), _p(w), _p(bias), _p(out), N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt(x), ACT[act],
        _p(pre[0]) if pre else None, _p(pre[1]) if pre else None,
        _p(post[0]) if post else None, _p(post[1]) if post else None,
        stats_mode, _p(stats), stats_bytes, out_image_stride, _p(ws), ws.numel() if ws is not None else 0, _stream())
    check(rc, "sihl_conv2d_fwd_ws")
    return out, stats


def pyr_conv_supported(N: int, W: int, Cin: int, Cout: int, mode: int, dtype) -> bool:
    return dtype == torch.bfloat16 and bool(_sized("sihl_pyr_conv_supported", N, W, Cin, Cout, mode))


def pyr_conv_raw(w: Tensor, x: Optional[Tensor] = None, fuse=None, bias: Optional[Tensor] = None, act: Optional[str] = None,
                 pre: Optional[Tuple[Tensor, Tensor]] = None, post: Optional[Tuple[Tensor, Tensor]] = None,
                 stats_mode: int = 0, want_merged: bool = False, emit=None, write_y: bool = True):
    """3x3 / pad 1 conv of a small square map (16, 8 or 4 wide, bf16) with its fusion node folded into the loader
    (sihl_pyr_conv_fwd, csrc/conv_pyr.hip).  Input: ``x`` (N, W, W, Cin), or ``fuse`` =
    ("up2", a, b, wraw)                  -> softmax(wraw)_0 * bilinear_x2(a) + softmax(wraw)_1 * b,  or
    ("blur", a, b, c, wraw, a_affine)    -> softmax(wraw)_0 * (blur_s2(a) * scale + shift) + .._1 * b + .._2 * c
    (a_affine = (scale, shift) or None).  Returns (y, stats, merged): stats fp32 [rows][2][Cout] when stats_mode != 0,
    merged = the conv's input when want_merged (training: the weight gradient reads it).

    emit (inference): the fusion node that consumes y, computed in the same launch - ("up2", b, wraw) ->
    softmax(wraw)_0 * bilinear_x2(y) + .._1 * b with b (N, 2W, 2W, Cout); ("blur", b, c, wraw) -> softmax(wraw)_0 *
    blur_s2(y) + .._1 * b + .._2 * c with b, c (N, W/2, W/2, Cout).  The return value then has a 4th entry, the node;
    write_y=False skips storing y itself (returned as None)."""
    Cout, KH, KW, Cin = w.shape
    assert KH == 3 and KW == 3
    a = b = c = wraw = a_scale = a_shift = None
    if fuse is None:
        mode, ref = 0, x
    elif fuse[0] == "up2":
        mode, (a, b, wraw), ref = 1, fuse[1:4], fuse[2]
    else:
        mode, (a, b, c, wraw), ref = 2, fuse[1:5], fuse[2]
        if len(fuse) > 5 and fuse[5] is not None:
            a_scale, a_shift = fuse[5]
    _require_gpu(ref)
    N, W = ref.shape[0], ref.shape[2]
    if ref.shape[1] != W or ref.shape[3] != Cin or not pyr_conv_supported(N, W, Cin, Cout, mode, ref.dtype):
        raise ValueError(f"pyr_conv_raw: unsupported problem {tuple(ref.shape)} -> {Cout}, mode {mode}, {ref.dtype}")
    out = torch.empty((N, W, W, Cout), dtype=ref.dtype, device=ref.device) if (write_y or emit is None) else None
    merged = torch.empty_like(ref) if (want_merged and mode != 0) else None
    e_mode, e_b, e_c, e_fw, e_out = 0, None, None, None, None
    if emit is not None:
        e_mode = 1 if emit[0] == "up2" else 2
        e_b, e_fw = emit[1], emit[-1]
        e_c = emit[2] if e_mode == 2 else None
        Wo = 2 * W if e_mode == 1 else W // 2
        want = (N, Wo, Wo, Cout)
        if tuple(e_b.shape) != want or e_b.dtype != ref.dtype or (e_c is not None and (tuple(e_c.shape) != want or e_c.dtype != ref.dtype)):
            raise ValueError(f"pyr_conv_raw: emit inputs must be {want} {ref.dtype}")
        e_out = torch.empty(want, dtype=ref.dtype, device=ref.device)
    stats, stats_bytes = None, 0
    if stats_mode:
        stats = torch.empty((_sized("sihl_pyr_conv_stat_rows", N, W), 2, Cout), dtype=torch.float32, device=ref.device)
        stats_bytes = stats.numel() * 4
    rc = _C.lib().sihl_pyr_conv_fwd(
        _p(x), _p(w), _p(bias), _p(out), N, W, Cin, Cout, ACT[act],
        _p(pre[0]) if pre else None, _p(pre[1]) if pre else None, _p(post[0]) if post else None, _p(post[1]) if post else None,
        stats_mode, _p(stats), stats_bytes, mode, _p(a), _p(b), _p(c), _p(wraw), _p(a_scale), _p(a_shift), _p(merged),
        e_mode, _p(e_b), _p(e_c), _p(e_fw), _p(e_out), _stream())
    check(rc, "sihl_pyr_conv_fwd")
    if emit is not None:
        return out, stats, (x if mode == 0 else merged), e_out
    return out, stats, (x if mode == 0 else merged)


def conv2d_wgrad_raw(x: Tensor, dout: Tensor, KH: int, KW: int, stride: int, pad: int, dil: int, side_ok: bool = True) -> Tensor:
    """Returns fp32 dW [Cout][KH][KW][Cin].  side_ok False: never on the side stream (the weight is not a leaf - a padded or
    otherwise derived tensor - so autograd will run a kernel on this gradient on the main stream right away)."""
    N, H, W, Cin = x.shape
    Cout = dout.shape[-1]
    lib = _C.lib()
    # dw is a main-stream allocation either way: it is consumed on the main stream (optimizer, clipping, all-reduce) after
    # the join, and its block goes back to the main stream's pool
    dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.float32, device=x.device)
    side = _SIDE if side_ok else None
    if side is not None and dout.numel() // Cout <= WGRAD_SIDE_MAX_PIXELS[side.mode]:
        nbytes = _sized("sihl_conv2d_wgrad_ws_bytes", N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt(x), SIDE_WGRAD_TARGET)
        side.stream.wait_stream(torch.cuda.current_stream())  # x and dout are complete on the main stream
        ws = workspace(nbytes, x.device, side.handle)  # the side stream's own scratch buffer (keyed by stream)
        rc = lib.sihl_conv2d_wgrad(_p(x), _p(dout), _p(dw), N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt(x),
                                   0, SIDE_WGRAD_TARGET, _p(ws), ws.numel(), side.handle)
        side.holds.append((x, dout, dw))
        side.dirty = True
    else:
        nbytes = _sized("sihl_conv2d_wgrad_ws_bytes", N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt(x), WGRAD_TARGET)
        ws = workspace(nbytes, x.device)
        rc = lib.sihl_conv2d_wgrad(_p(x), _p(dout), _p(dw), N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt(x), 0,
                                   WGRAD_TARGET, _p(ws), ws.numel(), _stream())
    check(rc, "sihl_conv2d_wgrad")
    return dw


def weight_for_dgrad(w: Tensor, flip: bool) -> Tensor:
    """[Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout] (spatially flipped for conv dgrad), same dtype."""
    Cout, KH, KW, Cin = w.shape
    o = torch.empty((Cin, KH, KW, Cout), dtype=w.dtype, device=w.device)
    rc = _C.lib().sihl_weight_flip_transpose(_p(w), _p(o), Cout, KH, KW, Cin, int(flip), _dt(w), _dt(w), _stream())
    check(rc, "sihl_weight_flip_transpose")
    return o


def bn_finalize(stats: Tensor, count: int, gamma, beta, eps, momentum, running_mean, running_var):
    C = stats.shape[-1]
    mean, rstd, scale, shift = torch.empty((4, C), dtype=torch.float32, device=stats.device).unbind(0)
    rc = _C.lib().sihl_bn_finalize(_p(stats), stats.shape[0], C, count, _p(gamma), _p(beta), eps, momentum,
                                   _p(running_mean), _p(running_var), _p(mean), _p(rstd), _p(scale), _p(shift),
                                   _stream())
    check(rc, "sihl_bn_finalize")
    if running_mean is not None:
        global _BN_STATS_GEN
        _BN_STATS_GEN += 1
    return mean, rstd, scale, shift


_BN_STATS_GEN = 0  # bumped whenever a sihl kernel rewrites running statistics (raw pointers: no version-counter bump)
_PARAM_GEN = 0  # bumped by Trainer.step after EVERY optimisation step, eager or HIP-graph replay


def bump_param_generation() -> None:
    """Tell the inference-side caches (eval BatchNorm affines, fp32 -> bf16 operand casts, MLP pointer plans) that parameters
    and running statistics may have changed behind torch's version counters: a HIP-graph replay updates weights, gamma /
    beta and running statistics by raw pointer, so neither ``tensor._version`` nor ``_BN_STATS_GEN`` (bumped at capture
    only) moves.  Every cache key below holds this count."""
    global _PARAM_GEN
    _PARAM_GEN += 1


def bn_eval_affine(gamma, beta, running_mean, running_var, eps):
    """Eval-mode BatchNorm as a per-channel (scale, shift) pair, CACHED on the running_mean buffer: the pair is a function
    of four tensors that do not change between inference forwards, and recomputing it was one 5 us launch of constants in
    front of every conv block (46 per BiFPN + detection-head forward, 0.21 of 3.4 ms, profiles/r03_ns_forward_*).  The key
    holds the storage addresses and torch's version counters of all four tensors (in-place ops, ``copy_`` and
    ``load_state_dict`` bump those; writes through ``.data`` do not - same contract as PreparedWeights) plus a generation
    count that every training-mode statistics update of this library bumps."""
    key = (_BN_STATS_GEN, _PARAM_GEN, eps, running_mean.data_ptr(), running_mean._version, running_var.data_ptr(), running_var._version,
           None if gamma is None else (gamma.data_ptr(), gamma._version),
           None if beta is None else (beta.data_ptr(), beta._version))
    hit = getattr(running_mean, "_sihl_eval_affine", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    scale, shift = _bn_eval_affine_launch(gamma, beta, running_mean, running_var, eps)
    if not torch.cuda.is_current_stream_capturing():  # a pair made inside a graph capture lives in the graph's pool
        running_mean._sihl_eval_affine = (key, scale, shift)
    return scale, shift


def _bn_eval_affine_launch(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    scale = torch.empty(C, dtype=torch.float32, device=running_mean.device)
    shift = torch.empty_like(scale)
    rc = _C.lib().sihl_bn_eval_affine(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C, _p(scale),
                                      _p(shift), _stream())
    check(rc, "sihl_bn_eval_affine")
    return scale, shift


def affine_act(x: Tensor, scale, shift, act) -> Tensor:
    C = x.shape[-1]
    y = torch.empty_like(x)
    rc = _C.lib().sihl_affine_act(_p(x), _p(y), x.numel() // C, C, _p(scale), _p(shift), ACT[act], _dt(x), _stream())
    check(rc, "sihl_affine_act")
    return y


def affine_add_act(x: Tensor, res: Tensor, scale, shift, act, want_mask: bool = False):
    """y = act(x * scale + shift + res).  want_mask: also return the (y > 0) bits, one byte per 16-byte vector of y (the
    residual tail's backward reads them instead of y)."""
    C = x.shape[-1]
    y = torch.empty_like(x)
    mask = torch.empty(x.numel() * x.element_size() // 16, dtype=torch.uint8, device=x.device) if want_mask else None
    rc = _C.lib().sihl_affine_add_act(_p(x), _p(res), _p(y), _p(mask), x.numel() // C, C, _p(scale), _p(shift), ACT[act],
                                      _dt(x), _stream())
    check(rc, "sihl_affine_add_act")
    return (y, mask) if want_mask else y


def affine_act_bwd(x: Tensor, dy: Tensor, scale, shift, act) -> Tensor:
    C = x.shape[-1]
    dx = torch.empty_like(x)
    rc = _C.lib().sihl_affine_act_bwd(_p(x), _p(dy), _p(dx), x.numel() // C, C, _p(scale), _p(shift), ACT[act],
                                      _dt(x), _stream())
    check(rc, "sihl_affine_act_bwd")
    return dx


def norm_act_bwd(s: Tensor, dy: Tensor, mean, rstd, gamma, beta, mode: int, act, batch_stats: bool):
    C = s.shape[-1]
    rows = s.numel() // C
    lib = _C.lib()
    ws = workspace(_sized("sihl_norm_act_bwd_ws_bytes", rows, C, _dt(s)), s.device)
    dz = torch.empty_like(s)
    dgamma = torch.empty(C, dtype=torch.float32, device=s.device)
    dbeta = torch.empty_like(dgamma)
    rc = lib.sihl_norm_act_bwd(_p(s), _p(dy), _p(dz), rows, C, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dgamma),
                               _p(dbeta), mode, ACT[act], int(batch_stats), _dt(s), _p(ws), ws.numel(), _stream())
    check(rc, "sihl_norm_act_bwd")
    return dz, dgamma, dbeta


def norm_add_relu_bwd(s: Tensor, dy: Tensor, y, mean, rstd, gamma, beta, batch_stats: bool, mask=None):
    """Backward of y = relu(BN(s) + identity): (dres, dz, dgamma, dbeta) with the ReLU mask applied inside the reduction
    pass (sihl_norm_add_relu_bwd) instead of a pass of its own.  The mask is read from ``mask`` (affine_add_act's bytes)
    when given, else from ``y``."""
    C = s.shape[-1]
    rows = s.numel() // C
    lib = _C.lib()
    ws = workspace(_sized("sihl_norm_act_bwd_ws_bytes", rows, C, _dt(s)), s.device)
    dres, dz = torch.empty_like(s), torch.empty_like(s)
    dgamma = torch.empty(C, dtype=torch.float32, device=s.device)
    dbeta = torch.empty_like(dgamma)
    rc = lib.sihl_norm_add_relu_bwd(_p(s), _p(dy), _p(y), _p(mask), _p(dres), _p(dz), rows, C, _p(mean), _p(rstd), _p(gamma),
                                    _p(beta), _p(dgamma), _p(dbeta), int(batch_stats), _dt(s), _p(ws), ws.numel(), _stream())
    check(rc, "sihl_norm_add_relu_bwd")
    return dres, dz, dgamma, dbeta


def colsum(x: Tensor, off_chain: bool = False) -> Tensor:
    """Column sums of a (rows, C) tensor.  off_chain: the result is a parameter gradient (a bias): nothing on the backward
    chain reads it, so inside ``wgrad_side_stream`` it is computed on the side stream like the weight gradients (the side
    stream ends 0.45 ms before the main stream reaches the join, tools/join_probe.py; 20 such sums per detection step)."""
    C = x.shape[-1]
    rows = x.numel() // C
    lib = _C.lib()
    out = torch.empty(C, dtype=torch.float32, device=x.device)  # a main-stream allocation either way (see conv2d_wgrad_raw)
    side = _SIDE if off_chain else None
    if side is not None and rows <= WGRAD_SIDE_MAX_PIXELS[side.mode]:
        side.stream.wait_stream(torch.cuda.current_stream())  # x is complete on the main stream
        ws = workspace(_sized("sihl_colsum_ws_bytes", rows, C), x.device, side.handle)
        rc = lib.sihl_colsum(_p(x), rows, C, _p(out), _dt(x), _p(ws), ws.numel(), side.handle)
        side.holds.append((x, x, out))
        side.dirty = True
        check(rc, "sihl_colsum")
        # handed on as a VIEW (a tensor object of its own): AccumulateGrad takes a gradient as it is only when nothing else
        # references that tensor object - `holds` does - and otherwise clones it, on the main stream, which may run before
        # the side stream has written it (same contract as the weight gradients, which leave as permuted views)
        return out.view(C)
    else:
        ws = workspace(_sized("sihl_colsum_ws_bytes", rows, C), x.device)
        rc = lib.sihl_colsum(_p(x), rows, C, _p(out), _dt(x), _p(ws), ws.numel(), _stream())
    check(rc, "sihl_colsum")
    return out


# ----------------------------------------------------------------------------- BatchNorm + act on a foreign conv's output
class BNActFn(torch.autograd.Function):
    """y = act(BatchNorm(s)) in training mode for an NHWC tensor s that another library's conv produced (the ResNet stem:
    MIOpen's 7x7 / stride 2 conv over 3 input channels): batch statistics by ``sihl_bn_stats``, then the same
    finalize / normalise / backward kernels as the fused conv block (ConvBlockFn, order "norm_act")."""

    @staticmethod
    def forward(ctx, s, gamma, beta, running_mean, running_var, eps, momentum, act):
        sd = s.detach().contiguous()
        C = sd.shape[-1]
        rows = sd.numel() // C
        lib = _C.lib()
        n = lib.sihl_bn_stats_rows(rows, C, _dt(sd))
        stats = torch.empty((n, 2, C), dtype=torch.float32, device=sd.device)
        check(lib.sihl_bn_stats(_p(sd), rows, C, _p(stats), n, _dt(sd), _stream()), "sihl_bn_stats")
        mean, rstd, scale, shift = bn_finalize(stats, rows, gamma, beta, eps, momentum, running_mean, running_var)
        y = affine_act(sd, scale, shift, act)
        ctx.save_for_backward(sd, mean, rstd, gamma.detach(), beta.detach())
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        s, mean, rstd, gamma, beta = ctx.saved_tensors
        dz, dgamma, dbeta = norm_act_bwd(s, dy.contiguous(), mean, rstd, gamma, beta, 1, ctx.act, True)
        return (dz if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None, None, None, None


class StemFn(torch.autograd.Function):
    """The ResNet stem conv1 -> bn1 -> relu (torchvision resnet.py, behind src/sihl/torchvision_backbone.py:42-49) in bf16:
    the 7x7 / stride 2 conv over 3 channels on ``sihl_stem_conv_fwd`` (csrc/stem.hip: the image packed once into a padded
    NHWC bf16 copy, kernel rows as 32-element K segments), its batch statistics from the conv's own epilogue, then the
    finalize / normalise / backward kernels of the fused conv block.  NCHW image in, NHWC activation out.  The weight
    gradient is ``sihl_stem_conv_wgrad`` over the same packed image (SIHL_ATEN_STEM_WGRAD=1: ATen's, for the A/B)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, eps, momentum, act, training):
        xd, wd = x.detach(), weight.detach()
        N, _, H, W = xd.shape
        lib = _C.lib()
        xp = torch.empty(_sized("sihl_stem_xp_bytes", N, H, W) // 2, dtype=torch.bfloat16, device=xd.device)
        wp = torch.empty(64 * 7 * 32, dtype=torch.bfloat16, device=xd.device)
        s = torch.empty((N, H // 2, W // 2, 64), dtype=torch.bfloat16, device=xd.device)
        stats = torch.empty((_sized("sihl_stem_stats_rows", N, H), 2, 64), dtype=torch.float32, device=xd.device) \
            if training else None
        rc = lib.sihl_stem_conv_fwd(_p(xd), _dt(xd), *xd.stride(), _p(wd), *wd.stride(), _p(xp), _p(wp), _p(s), _p(stats),
                                    N, H, W, _stream())
        check(rc, "sihl_stem_conv_fwd")
        if training:
            mean, rstd, scale, shift = bn_finalize(stats, s.numel() // 64, gamma, beta, eps, momentum, running_mean, running_var)
        else:
            scale, shift = bn_eval_affine(gamma, beta, running_mean, running_var, eps)
            mean, rstd = running_mean.detach().clone(), torch.rsqrt(running_var.detach() + eps)
        y = affine_act(s, scale, shift, act)
        ctx.save_for_backward(xp, s, mean, rstd, gamma.detach(), beta.detach(), wd)
        ctx.act, ctx.training, ctx.shape = act, training, (N, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, s, mean, rstd, gamma, beta, w = ctx.saved_tensors
        N, H, W = ctx.shape
        dz, dgamma, dbeta = norm_act_bwd(s, dy.contiguous(), mean, rstd, gamma, beta, 1, ctx.act, ctx.training)
        dw = None
        if ctx.needs_input_grad[1] and not _ATEN_STEM_WGRAD:
            dw = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=dz.device)
            ws = workspace(_sized("sihl_stem_wgrad_parts", N, H) * 64 * 7 * 32 * 4, dz.device)
            rc = _C.lib().sihl_stem_conv_wgrad(_p(xp), _p(dz), _p(dw), *dw.stride(), _p(ws), N, H, W, _stream())
            check(rc, "sihl_stem_conv_wgrad")
        elif ctx.needs_input_grad[1]:  # A/B switch: MIOpen's weight gradient over the packed image
            Wp = xp.numel() // (N * (H + 6) * 3)
            x_img = xp.view(N, H + 6, Wp, 3)[:, 3:3 + H, 4:4 + W, :].permute(0, 3, 1, 2)  # the image, bf16, NCHW view
            dw = torch.ops.aten.convolution_backward(nchw_view(dz), x_img, w.to(torch.bfloat16), None, [2, 2], [3, 3], [1, 1],
                                                     False, [0, 0], 1, [False, True, False])[1].float()
        return None, dw, dgamma, dbeta, None, None, None, None, None, None


def stem_supported(x: Tensor, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d) -> bool:
    """The native stem covers torchvision's conv1 (3 -> 64, 7x7, stride 2, pad 3, no bias) on an even-sized CUDA image
    that needs no gradient, computed in bf16 (autocast), with a default BatchNorm2d behind it."""
    return (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and x.dtype in (torch.float32, torch.bfloat16) and not x.requires_grad
            and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16))
            and tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.weight.dtype == torch.float32
            and bn.momentum is not None and bn.track_running_stats and bn.affine
            and not os.environ.get("SIHL_ATEN_STEM"))  # env: A/B switch


def stem_conv_bn_act(x: Tensor, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, act: Optional[str]) -> Tensor:
    """conv1 -> bn1 -> act of the ResNet stem (see StemFn); updates bn's running statistics in training mode."""
    if bn.training:
        bump_counter(bn.num_batches_tracked)
    return StemFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act,
                        bn.training)


def bn_act_train(s_nhwc: Tensor, bn: torch.nn.BatchNorm2d, act: Optional[str]) -> Tensor:
    """Training-mode BatchNorm2d (+ activation) of an NHWC tensor through the sihl kernels; updates bn's running
    statistics and step counter like nn.BatchNorm2d."""
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise NotImplementedError("bn_act_train: affine BatchNorm2d with a fixed momentum and running statistics only")
    bump_counter(bn.num_batches_tracked)
    return BNActFn.apply(s_nhwc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act)


# ----------------------------------------------------------------------------- conv (+norm +act) block
class ConvBlockFn(torch.autograd.Function):
    """conv -> act -> BN ("act_norm", ConvNormAct), conv -> BN -> act ("norm_act", torchvision's
    Conv2dNormActivation) or conv(+bias) -> act (no norm), NHWC in and out."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, cfg, residual=None, hand_over=None,
                take_over=None, dx_to=None, defer=None, precomputed=None):
        # precomputed = (s, stats): the conv pass has already run (fused_node_conv_block: the launch that computed the
        # fusion node `x` inside its loader); everything else - BatchNorm, the saved tensors, the backward - is this block's
        stride, pad, dil, act, order, has_norm, training, eps, momentum, need_grad = cfg
        xd = x.detach()
        ctx.has_res = residual is not None
        # GradCarrier protocol of an identity residual block: the merge (hand_over) parks the identity branch's
        # gradient instead of returning it, the block's first conv (take_over) adds it inside its dgrad epilogue
        ctx.hand_over, ctx.take_over = hand_over, take_over
        # projection (downsample) conv of a residual block: its input gradient is parked COMPACT in the carrier and the
        # block's first conv adds it at the pixels the projection reads (sihl_conv2d_dgrad_add)
        ctx.dx_to = dx_to
        prep = prepared(weight, xd.dtype)
        w = prep.w if prep is not None else weight_khwc(weight, xd.dtype)
        ctx.wt = prep.wt if prep is not None else None  # dgrad operand, valid until the next optimizer step
        KH, KW = w.shape[1], w.shape[2]
        ctx.cfg, ctx.has_norm, ctx.kshape = cfg, has_norm, (KH, KW)
        ctx.has_bias = bias is not None
        ctx.pkeys = (weight.data_ptr(),) + ((bias.data_ptr(),) if bias is not None else ())
        # A gradient queued on the side stream may only go where no main-stream kernel touches it before the join: to a LEAF
        # (AccumulateGrad takes it as it is).  A derived weight / bias - the zero-padded copies of odd channel counts
        # (layers/convblocks.py: the 21-class segmentation classifier) - sends its gradient through autograd's slice / copy
        # kernels on the main stream at once: a race with the side stream's write that round 4 found as garbage gradient norms
        # (clip coefficient ~ 0, the SemanticSegmentation configurations stopped learning in two-stream mode).
        ctx.side_ok_w = weight.is_leaf
        ctx.side_ok_b = bias is None or bias.is_leaf
        if not has_norm:
            fused_ok = act in (None, "none", "relu")
            if fused_ok or not need_grad:
                y, _ = conv2d_raw(xd, w, bias, stride, pad, dil, act=act)
                ctx.save_for_backward(xd, w, y if act == "relu" else None)
                ctx.kind = "fused_act"
            else:  # silu / sigmoid with grad: keep the pre-activation
                z, _ = conv2d_raw(xd, w, bias, stride, pad, dil)
                y = affine_act(z, None, None, act)
                ctx.save_for_backward(xd, w, z)
                ctx.kind = "split_act"
            return y
        mode = 0 if order == "act_norm" else 1
        ctx.mode = mode
        if training:
            # one conv pass producing the pre-norm tensor + per-channel partial sums
            if precomputed is not None:
                s, stats = precomputed
            else:
                s, stats = conv2d_raw(xd, w, bias, stride, pad, dil, act=act if mode == 0 else None,
                                      stats_mode=2 if mode == 0 else 1)
            count = s.numel() // s.shape[-1]
            mean, rstd, scale, shift = bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean,
                                                   running_var)
            if residual is not None:  # block tail: y = relu(BN(conv) + identity) in one pass; keep y for the mask
                # the backward's ReLU mask: bits written here (1/16 of y's bytes) unless the A/B switch asks for y itself
                if need_grad and not _SPLIT_TAIL_BWD and TAIL_MASK_BITS:
                    y, bits = affine_add_act(s, residual.detach().contiguous(), scale, shift, "relu", want_mask=True)
                else:
                    y, bits = affine_add_act(s, residual.detach().contiguous(), scale, shift, "relu"), None
                ctx.tail_bits = bits is not None
                ctx.save_for_backward(xd, w, s, mean, rstd, gamma.detach(), beta.detach(), bits if bits is not None else y)
                ctx.batch_stats, ctx.kind = True, "norm"
                return y
            if defer is not None and mode == 0:
                # the consumer (a blur-pool) applies y = s * scale + shift: hand s out in y's place (see DeferredAffine);
                # the gradient that comes back is y's, as for any other consumer
                defer.scale, defer.shift = scale, shift
                y = s
            else:
                y = affine_act(s, scale, shift, None if mode == 0 else act)
            ctx.batch_stats = True
        else:
            scale, shift = bn_eval_affine(gamma, beta, running_mean, running_var, eps)
            if not need_grad:
                if mode == 0:
                    y, _ = conv2d_raw(xd, w, bias, stride, pad, dil, act=act, post=(scale, shift))
                else:
                    y, _ = conv2d_raw(xd, w, bias, stride, pad, dil, act=act, pre=(scale, shift))
                return y
            s, _ = conv2d_raw(xd, w, bias, stride, pad, dil, act=act if mode == 0 else None)
            y = affine_act(s, scale, shift, None if mode == 0 else act)
            mean = running_mean.detach().clone()
            rstd = torch.rsqrt(running_var.detach() + eps)
            ctx.batch_stats = False
        ctx.save_for_backward(xd, w, s, mean, rstd, gamma.detach() if gamma is not None else None,
                              beta.detach() if beta is not None else None)
        ctx.kind = "norm"
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, dil, act, order, has_norm, training, eps, momentum, _ = ctx.cfg
        KH, KW = ctx.kshape
        dy = dy.contiguous()
        dgamma = dbeta = None
        dres = None
        if ctx.kind == "norm" and ctx.has_res:
            x, w, s, mean, rstd, gamma, beta, y = ctx.saved_tensors
            # dres = dy * (y > 0), the gradient of both merge inputs, and BatchNorm's backward of it: the mask rides in
            # the reduction pass (7 tensor passes and 3 launches instead of 8 and 4)
            if _SPLIT_TAIL_BWD:  # env A/B switch: the separate ReLU-backward pass of round 1
                dres = affine_act_bwd(y, dy, None, None, "relu")
                dz, dgamma, dbeta = norm_act_bwd(s, dres, mean, rstd, gamma, beta, 1, None, ctx.batch_stats)
            else:
                if ctx.tail_bits:
                    dres, dz, dgamma, dbeta = norm_add_relu_bwd(s, dy, None, mean, rstd, gamma, beta, ctx.batch_stats, mask=y)
                else:
                    dres, dz, dgamma, dbeta = norm_add_relu_bwd(s, dy, y, mean, rstd, gamma, beta, ctx.batch_stats)
            if ctx.hand_over is not None:
                ctx.hand_over.tensor, dres = dres, None
        elif ctx.kind == "norm":
            x, w, s, mean, rstd, gamma, beta = ctx.saved_tensors
            dz, dgamma, dbeta = norm_act_bwd(s, dy, mean, rstd, gamma, beta, ctx.mode, act, ctx.batch_stats)
        elif ctx.kind == "split_act":
            x, w, z = ctx.saved_tensors
            dz = affine_act_bwd(z, dy, None, None, act)
        else:
            x, w, y = ctx.saved_tensors
            dz = affine_act_bwd(y, dy, None, None, "relu") if y is not None else dy
        dbias = colsum(dz, off_chain=ctx.side_ok_b) if ctx.has_bias else None
        dw = dx = None
        if ctx.needs_input_grad[1]:
            dw = conv2d_wgrad_raw(x, dz, KH, KW, stride, pad, dil, side_ok=ctx.side_ok_w).permute(0, 3, 1, 2)  # (O,I,KH,KW) view
        if ctx.needs_input_grad[0] and ctx.dx_to is not None and ctx.dx_to.tensor is None:
            # 1x1, pad 0, stride s: dx is W^T dz at the pixels the conv reads and zero elsewhere - compute it at dz's
            # resolution (a dense stride-1 1x1 dgrad over the Ho x Wo grid) and hand it over
            wt = ctx.wt if ctx.wt is not None else weight_for_dgrad(w, flip=True)
            N, Ho, Wo, Cout = dz.shape
            Cin = x.shape[-1]
            dxs = torch.empty((N, Ho, Wo, Cin), dtype=x.dtype, device=x.device)
            lib = _C.lib()
            ws_bytes = _sized("sihl_conv2d_ws_bytes", N, Ho, Wo, Cout, Cin, 1, 1, 1, 0, 1)
            ws = workspace(ws_bytes, x.device) if ws_bytes else None
            rc = lib.sihl_conv2d_dgrad_ws(_p(dz), _p(wt), _p(dxs), None, N, Ho, Wo, Cin, Cout, 1, 1, 1, 0, 1, _dt(x),
                                          _p(ws), ws.numel() if ws is not None else 0, _stream())
            check(rc, "sihl_conv2d_dgrad (compact projection gradient)")
            ctx.dx_to.tensor, ctx.dx_to.stride = dxs, stride
        elif ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else weight_for_dgrad(w, flip=True)
            dx = torch.empty_like(x)
            N, H, W, Cin = x.shape
            lib = _C.lib()
            ws_bytes = _sized("sihl_conv2d_ws_bytes", N, H, W, w.shape[0], Cin, KH, KW, 1, dil * (KH - 1) - pad, dil) \
                if stride == 1 else 0
            ws = workspace(ws_bytes, x.device) if ws_bytes else None
            add, add_stride = None, 1
            if ctx.take_over is not None and ctx.take_over.expect and ctx.take_over.tensor is None:
                raise RuntimeError("GradCarrier: the shortcut gradient was not parked before the first conv's backward "
                                   "(the producer must be created after the consumer in forward)")
            if ctx.take_over is not None and ctx.take_over.tensor is not None:
                add, ctx.take_over.tensor = ctx.take_over.tensor.contiguous(), None
                add_stride, ctx.take_over.stride, ctx.take_over.expect = ctx.take_over.stride, 1, False
                want = (N, (H + add_stride - 1) // add_stride, (W + add_stride - 1) // add_stride, Cin)
                assert tuple(add.shape) == want and add.dtype == dx.dtype, (tuple(add.shape), want)
            rc = lib.sihl_conv2d_dgrad_add(_p(dz), _p(wt), _p(dx), _p(add), add_stride, N, H, W, Cin, w.shape[0], KH, KW,
                                           stride, pad, dil, _dt(x), _p(ws), ws.numel() if ws is not None else 0,
                                           _stream())
            check(rc, "sihl_conv2d_dgrad")
        _guard_shared_parameters(ctx.pkeys)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, dres, None, None, None, None, None


DEFER_BN_AFFINE = os.environ.get("SIHL_DEFER_BN_AFFINE", "1") != "0"  # A/B and test switch: False = every conv block applies its BatchNorm affine itself


class DeferredAffine:
    """Lets a training-mode conv -> act -> BatchNorm block leave its last step, y = s * scale + shift, to the ONE consumer
    of y when that consumer is a blur-pool (the antialiased downscalers of the necks): the block returns the pre-norm
    tensor s and fills ``scale`` / ``shift``; ``blur_fuse(.., a_affine=carrier)`` applies them to the blurred value.  One
    launch and one read + write of the full-resolution tensor less per downscaler (the affine is linear and the blur taps
    sum to 1, so the result is the same up to rounding).  Left empty (scale None) whenever the block applied the affine
    itself (inference: folded into the conv epilogue)."""

    __slots__ = ("scale", "shift")

    def __init__(self):
        self.scale = self.shift = None


class GradCarrier:
    """Hands the identity branch's gradient of a residual block from the block's merge to its first conv, which adds
    it in its dgrad epilogue (autograd would otherwise launch one add kernel per block over the block-input tensor:
    1.3 ms per ResNet50 step).  Valid only when both convs read the SAME tensor x: the merge with
    ``residual=x, hand_over=c``, the first conv with ``take_over=c``."""
    __slots__ = ("tensor", "armed", "stride", "expect")

    def __init__(self):
        self.tensor = None
        self.armed = False  # set once the first conv has agreed to take the gradient over
        self.stride = 1  # > 1: the parked tensor is the compact input gradient of a strided 1x1 projection (dx_to)
        self.expect = False  # a producer (hand_over / dx_to) has committed to park a gradient in this backward


def conv_block_into(out: Tensor, out_image_stride: int, x_nhwc, weight, bias, gamma, beta, running_mean, running_var, *,
                    stride=1, pad=0, dil=1, act=None, order="act_norm", eps=1e-5) -> None:
    """Inference-only conv block (BatchNorm folded into the conv epilogue) that writes image n's output rows at
    ``out + n * out_image_stride`` elements: per-level laterals land directly in a head's flat (B, P, C) position buffer,
    no concatenation pass.  No autograd (call it under no_grad with eval-mode statistics)."""
    xd = x_nhwc.detach()
    prep = prepared(weight, xd.dtype)
    w = prep.w if prep is not None else weight_khwc(weight, xd.dtype)
    if running_mean is None:
        conv2d_raw(xd, w, bias, stride, pad, dil, act=act, out=out, out_image_stride=out_image_stride)
        return
    scale, shift = bn_eval_affine(gamma, beta, running_mean, running_var, eps)
    if order == "act_norm":
        conv2d_raw(xd, w, bias, stride, pad, dil, act=act, post=(scale, shift), out=out, out_image_stride=out_image_stride)
    else:
        conv2d_raw(xd, w, bias, stride, pad, dil, act=act, pre=(scale, shift), out=out, out_image_stride=out_image_stride)


def conv_block(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, *, stride=1, pad=0, dil=1, act=None,
               order="act_norm", training=False, eps=1e-5, momentum=0.1, residual=None, hand_over=None,
               take_over=None, dx_to=None, defer=None):
    """residual (optional, order "norm_act" only): the block computes relu(BN(conv(x)) + residual).
    defer: a DeferredAffine - the caller promises that the ONLY consumer of the result is ``blur_fuse(.., a_affine=defer)``.
    hand_over / take_over: see GradCarrier (training only).  dx_to: carrier that receives this (1x1, pad 0) conv's
    input gradient in compact form - the projection shortcut of a residual block whose first conv was given the same
    carrier as take_over and reads the same x."""
    has_norm = running_mean is not None
    # autograd.Function.forward always runs with grad mode off, so decide here whether a backward can follow
    need_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x_nhwc, weight, bias, gamma, beta))
    cfg = (stride, pad, dil, act, order, has_norm, training, eps, momentum, need_grad)
    if residual is None:
        if take_over is not None and need_grad and x_nhwc.requires_grad and stride == 1:
            take_over.armed = True
            return ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg, None, None,
                                     take_over)
        if dx_to is not None and dx_to.armed and need_grad and x_nhwc.requires_grad and pad == 0 and dil == 1 \
                and tuple(weight.shape[2:]) == (1, 1) and x_nhwc.shape[-1] % (8 if x_nhwc.dtype == torch.bfloat16 else 4) == 0:
            dx_to.expect = True
            return ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg, None, None,
                                     None, dx_to)
        if defer is not None and DEFER_BN_AFFINE and has_norm and training and order == "act_norm":
            return ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg, None, None, None,
                                     None, defer)
        return ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg)
    if order != "norm_act" or not has_norm or act is not None:
        raise ValueError("residual merge is defined for conv -> BatchNorm (no activation) blocks")
    if training:
        need_grad = need_grad or (torch.is_grad_enabled() and residual.requires_grad)
        cfg = (stride, pad, dil, act, order, has_norm, training, eps, momentum, need_grad)
        if hand_over is not None and not hand_over.armed:
            hand_over = None  # nobody will pick the gradient up: let autograd route it
        if hand_over is not None:
            hand_over.expect = True
        return ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg, residual, hand_over)
    return add_relu(ConvBlockFn.apply(x_nhwc, weight, bias, gamma, beta, running_mean, running_var, cfg), residual)


# Opt-in (tests, A/B): the fusion node inside the conv's loader.  Off by default: every one of a map's 8 output-slice
# workgroups recomputes the whole node, and on the GPU the one-launch form only ties with [fusion kernel -> conv] on the
# top-down nodes (P6 9.3 -> 8.8 us, P7 8.1 -> 8.3) and loses on the bottom-up ones (P6 10.8 -> 11.3, P7 8.8 -> 10.7;
# profiles/r04_pyr_probe.txt) - it saves host launches, not GPU time.
FUSE_NODE_CONV = os.environ.get("SIHL_FUSE_NODE_CONV", "0") != "0"


def fused_node_conv_block(fuse, weight, bias, gamma, beta, running_mean, running_var, *, act, training, eps=1e-5,
                          momentum=0.1, defer=None):
    """A BiFPN fusion node and the conv -> act -> BatchNorm block behind it as ONE launch (conv_pyr.hip: the node is computed
    inside the conv's loader; reference layers/bifpn.py:41-52).  fuse = ("up2", a, b, wraw) or ("blur", a, b, c, wraw,
    a_affine) with a_affine a DeferredAffine or None.  Returns the block's output, or None when the problem is outside
    the kernel's shapes (the caller then runs the node and the block separately - same values: the fused launch is
    bit-identical to that sequence).

    Training: the launch also stores the node's value (the weight gradient reads it) and the autograd graph is the usual
    one - FuseUp2Fn / BlurFuseFn feeding ConvBlockFn, each told that its forward has already been computed."""
    kind, a, b = fuse[0], fuse[1], fuse[2]
    c = fuse[3] if kind == "blur" else None
    wraw = fuse[3] if kind == "up2" else fuse[4]
    a_affine = fuse[5] if kind == "blur" and len(fuse) > 5 else None
    mode = 1 if kind == "up2" else 2
    N, W, Cin = b.shape[0], b.shape[2], b.shape[3]
    Cout = weight.shape[0]
    if (not FUSE_NODE_CONV or not b.is_cuda or b.shape[1] != W or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != Cin
            or not pyr_conv_supported(N, W, Cin, Cout, mode, b.dtype) or a.dtype != b.dtype
            or tuple(a.shape) != ((N, W // 2, W // 2, Cin) if mode == 1 else (N, 2 * W, 2 * W, Cin))
            or (c is not None and (c.shape != b.shape or c.dtype != b.dtype))):
        return None
    tensors = (a, b, c, wraw, weight, bias, gamma, beta)
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
    if not training and need_grad:
        return None  # eval-mode statistics with autograd: the general path keeps what the backward needs
    prep = prepared(weight, b.dtype)
    w = prep.w if prep is not None else weight_khwc(weight, b.dtype)
    aff = None
    if a_affine is not None and a_affine.scale is not None:
        aff = (a_affine.scale, a_affine.shift)
    wr = wraw.detach().float().contiguous()
    ad, bd = a.detach().contiguous(), b.detach().contiguous()
    raw = ("up2", ad, bd, wr) if mode == 1 else ("blur", ad, bd, c.detach().contiguous(), wr, aff)
    if not training:
        scale, shift = bn_eval_affine(gamma, beta, running_mean, running_var, eps)
        return pyr_conv_raw(w, fuse=raw, bias=bias, act=act, post=(scale, shift))[0]
    s, stats, merged = pyr_conv_raw(w, fuse=raw, bias=bias, act=act, stats_mode=2, want_merged=True)
    if mode == 1:
        m = FuseUp2Fn.apply(a, b, wraw, merged)
    else:
        m = BlurFuseFn.apply(a, b, c, wraw, aff[0] if aff else None, aff[1] if aff else None, merged)
    cfg = (1, 1, 1, act, "act_norm", True, True, eps, momentum, need_grad)
    if defer is not None and not DEFER_BN_AFFINE:
        defer = None
    return ConvBlockFn.apply(m, weight, bias, gamma, beta, running_mean, running_var, cfg, None, None, None, None, defer,
                             (s, stats))


EMIT_NODES = os.environ.get("SIHL_EMIT_NODES", "1") != "0"  # A/B and test switch: False = every fusion node its own launch


def conv_block_emit(x: Tensor, weight, bias, gamma, beta, running_mean, running_var, *, act, eps, emit, write_y=True):
    """Inference: a 3x3 conv -> act -> BatchNorm block on a small square map AND the fusion node that consumes its output
    in one launch (sihl_pyr_conv_fwd with `emit`).  Returns (y or None, node), or None when outside the kernel's shapes."""
    N, H, W, Cin = x.shape
    Cout = weight.shape[0]
    kind = emit[0]
    if (not EMIT_NODES or not x.is_cuda or H != W or not pyr_conv_supported(N, W, Cin, Cout, 0, x.dtype)
            or (kind == "blur" and W % 2) or _C.lib().sihl_conv2d_small_mode() != 1):
        return None
    Wo = 2 * W if kind == "up2" else W // 2
    others = emit[1:-1]
    if any(tuple(t.shape) != (N, Wo, Wo, Cout) or t.dtype != x.dtype for t in others):
        return None
    prep = prepared(weight, x.dtype)
    w = prep.w if prep is not None else weight_khwc(weight, x.dtype)
    scale, shift = bn_eval_affine(gamma, beta, running_mean, running_var, eps)
    wr = emit[-1].detach().float().contiguous()
    raw = (kind,) + tuple(t.detach().contiguous() for t in others) + (wr,)
    y, _, _, node = pyr_conv_raw(w, x=x.detach().contiguous(), bias=bias, act=act, post=(scale, shift), emit=raw, write_y=write_y)
    return y, node


# ----------------------------------------------------------------------------- fusion nodes
class FuseUp2Fn(torch.autograd.Function):
    """out = softmax(w)[0] * bilinear_x2(a) + softmax(w)[1] * b   (NHWC)."""

    @staticmethod
    def forward(ctx, a, b, wraw, precomputed=None):
        # precomputed: the node's value, already produced by the conv launch that consumes it (fused_node_conv_block)
        a, b, wr = a.detach().contiguous(), b.detach().contiguous(), wraw.detach().float().contiguous()
        N, H, W, C = b.shape
        if precomputed is not None:
            out = precomputed.view_as(precomputed)
        else:
            out = torch.empty_like(b)
            rc = _C.lib().sihl_fuse_up2(_p(a), _p(b), _p(wr), _p(out), N, H, W, C, _dt(b), _stream())
            check(rc, "sihl_fuse_up2")
        ctx.save_for_backward(a, b, wr)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, wr = ctx.saved_tensors
        dout = dout.contiguous()
        N, H, W, C = b.shape
        need_a, need_b, need_w = ctx.needs_input_grad[:3]
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        dw = torch.empty(2, dtype=torch.float32, device=b.device) if need_w else None
        gacc = torch.empty(FUSION_GACC_FLOATS, dtype=torch.float32, device=b.device) if need_w else None
        rc = _C.lib().sihl_fuse_up2_bwd(_p(dout), _p(a), _p(b), _p(wr), _p(da), _p(db), _p(dw), _p(gacc), N, H, W, C,
                                        _dt(b), _stream())
        check(rc, "sihl_fuse_up2_bwd")
        return da, db, dw, None


class BlurFuseFn(torch.autograd.Function):
    """out = w0 * blurpool_s2(a) + w1 * b + w2 * c, or plain blurpool_s2(a) when b is None  (NHWC).
    a_scale / a_shift: ``a`` is the pre-norm output of a conv block that deferred its BatchNorm affine (DeferredAffine); the
    gradient returned for it is that of ``a * scale + shift``, which is what the block's backward expects."""

    @staticmethod
    def forward(ctx, a, b, c, wraw, a_scale=None, a_shift=None, precomputed=None):
        a = a.detach().contiguous()
        N, H, W, C = a.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        fused = b is not None
        if fused:
            b, c, wr = b.detach().contiguous(), c.detach().contiguous(), wraw.detach().float().contiguous()
        else:
            wr = None
        if precomputed is not None:  # the node's value, already produced by the conv launch that consumes it
            out = precomputed.view_as(precomputed)
        else:
            out = torch.empty((N, Ho, Wo, C), dtype=a.dtype, device=a.device)
            rc = _C.lib().sihl_blur_fuse(_p(a), _p(b), _p(c), _p(wr), _p(a_scale), _p(a_shift), _p(out), N, H, W, C, _dt(a),
                                         _stream())
            check(rc, "sihl_blur_fuse")
        ctx.fused = fused
        ctx.save_for_backward(a, b, c, wr, a_scale, a_shift)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, c, wr, a_scale, a_shift = ctx.saved_tensors
        dout = dout.contiguous()
        N, H, W, C = a.shape
        need_a, need_b, need_c, need_w = ctx.needs_input_grad[:4]
        dev = a.device
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(dout) if (ctx.fused and need_b) else None
        dc = torch.empty_like(dout) if (ctx.fused and need_c) else None
        dw = torch.empty(3, dtype=torch.float32, device=dev) if (ctx.fused and need_w) else None
        gacc = torch.empty(FUSION_GACC_FLOATS, dtype=torch.float32, device=dev) if dw is not None else None
        rc = _C.lib().sihl_blur_fuse_bwd(_p(dout), _p(a), _p(b), _p(c), _p(wr), _p(a_scale), _p(a_shift), _p(da), _p(db),
                                         _p(dc), _p(dw), _p(gacc), N, H, W, C, _dt(a), _stream())
        check(rc, "sihl_blur_fuse_bwd")
        return da, db, dc, dw, None, None, None


class GradClipPlan:
    """Static tables of sihl_grad_clip for one list of gradient sizes: (tensor, 64 Ki-element chunk) per workgroup, the
    element counts, the scratch row.  Built once per model; ``run`` is three launches and no tensor creation."""

    GROUP, CHUNK = int(os.environ.get("SIHL_CLIP_GROUP", "320")), 1 << 16  # (developer A/B: 32)

    def __init__(self, numels, device):
        import ctypes

        self.numels = tuple(int(n) for n in numels)
        pairs, groups = [], []
        for g0 in range(0, len(self.numels), self.GROUP):
            before = len(pairs)
            for k, n in enumerate(self.numels[g0:g0 + self.GROUP]):
                pairs.extend((k, c) for c in range(max(1, -(-n // self.CHUNK))))
            groups.append(len(pairs) - before)
        self.nblocks = len(pairs)
        self.map = torch.tensor(pairs, dtype=torch.int32).to(device)
        self.numel = torch.tensor(self.numels, dtype=torch.int64).to(device)
        self.scratch = torch.zeros(self.nblocks + 2, dtype=torch.float32, device=device)
        self.group_blocks = (ctypes.c_int * len(groups))(*groups)
        self.ptrs = (ctypes.c_void_p * len(self.numels))()

    def run(self, grads, max_norm: float) -> Tensor:
        """Clips ``grads`` (dense fp32 tensors of the plan's sizes) in place; returns the 2-float (coefficient, total norm)."""
        ptrs = self.ptrs
        for k, g in enumerate(grads):
            ptrs[k] = g.data_ptr()
        rc = _C.lib().sihl_grad_clip(ptrs, len(grads), _p(self.map), self.group_blocks, _p(self.numel), float(max_norm),
                                     _p(self.scratch), self.scratch.numel(), self.GROUP, _stream())
        check(rc, "sihl_grad_clip")
        return self.scratch[self.nblocks:]


def grad_clip_supported(grads) -> bool:
    """Dense fp32 device tensors (any memory format: norm and scale do not care about the order of elements)."""
    return all(g.is_cuda and g.dtype == torch.float32 and not g.is_sparse and g.numel() > 0
               and (g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last)
                    or g.untyped_storage().nbytes() == 4 * g.numel()) for g in grads)


class Up2Fn(torch.autograd.Function):
    """Plain bilinear x2 upsample (align_corners=False), NHWC."""

    @staticmethod
    def forward(ctx, a):
        a = a.detach().contiguous()
        N, h, w, C = a.shape
        out = torch.empty((N, 2 * h, 2 * w, C), dtype=a.dtype, device=a.device)
        rc = _C.lib().sihl_fuse_up2(_p(a), None, None, _p(out), N, 2 * h, 2 * w, C, _dt(a), _stream())
        check(rc, "sihl_fuse_up2")
        ctx.shape = (N, h, w, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        N, h, w, C = ctx.shape
        da = torch.empty((N, h, w, C), dtype=dout.dtype, device=dout.device)
        rc = _C.lib().sihl_fuse_up2_bwd(_p(dout), None, None, None, _p(da), None, None, None, N, 2 * h, 2 * w, C,
                                        _dt(dout), _stream())
        check(rc, "sihl_fuse_up2_bwd")
        return da


class FuseSumFn(torch.autograd.Function):
    """sum_i softmax(w)_i * x_i for 2 or 3 same-shaped tensors (stand-alone FastNormalizedFusion)."""

    @staticmethod
    def forward(ctx, wraw, *xs):
        xs = [x.detach().contiguous() for x in xs]
        wr = wraw.detach().float().contiguous()
        n = len(xs)
        out = torch.empty_like(xs[0])
        rc = _C.lib().sihl_fuse_sum(_p(xs[0]), _p(xs[1]), _p(xs[2]) if n > 2 else None, _p(wr), _p(out),
                                    out.numel(), n, _dt(out), _stream())
        check(rc, "sihl_fuse_sum")
        ctx.save_for_backward(wr, *xs)
        return out

    @staticmethod
    def backward(ctx, dout):
        wr, *xs = ctx.saved_tensors
        n = len(xs)
        dout = dout.contiguous()
        need = ctx.needs_input_grad
        ds = [torch.empty_like(x) if need[1 + i] else None for i, x in enumerate(xs)]
        dw = torch.empty(n, dtype=torch.float32, device=dout.device) if need[0] else None
        gacc = torch.empty(FUSION_GACC_FLOATS, dtype=torch.float32, device=dout.device) if need[0] else None
        rc = _C.lib().sihl_fuse_sum_bwd(_p(dout), _p(xs[0]), _p(xs[1]), _p(xs[2]) if n > 2 else None, _p(wr),
                                        _p(ds[0]), _p(ds[1]), _p(ds[2]) if n > 2 else None, _p(dw), _p(gacc),
                                        dout.numel(), n, _dt(dout), _stream())
        check(rc, "sihl_fuse_sum_bwd")
        return (dw, *ds)


class NearestUp2AddFn(torch.autograd.Function):
    """out = nearest_x2(lo) + skip (FPN top-down merge), NHWC."""

    @staticmethod
    def forward(ctx, lo, skip):
        lo, skip = lo.detach().contiguous(), skip.detach().contiguous()
        N, H, W, C = skip.shape
        out = torch.empty_like(skip)
        rc = _C.lib().sihl_nearest_up2_add(_p(lo), _p(skip), _p(out), N, H, W, C, _dt(skip), _stream())
        check(rc, "sihl_nearest_up2_add")
        ctx.lo_shape = tuple(lo.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        N, H, W, C = dout.shape
        dlo = None
        if ctx.needs_input_grad[0]:
            dlo = torch.empty(ctx.lo_shape, dtype=dout.dtype, device=dout.device)
            rc = _C.lib().sihl_nearest_up2_add_bwd(_p(dout), _p(dlo), N, H, W, C, _dt(dout), _stream())
            check(rc, "sihl_nearest_up2_add_bwd")
        return dlo, dout if ctx.needs_input_grad[1] else None


class ResizeBilinearFn(torch.autograd.Function):
    """out = bilinear_resize(a, size) (+ add), align_corners=False, NHWC."""

    @staticmethod
    def forward(ctx, a, add, size):
        a = a.detach().contiguous()
        N, H, W, C = a.shape
        Ho, Wo = size
        addc = add.detach().contiguous() if add is not None else None
        out = torch.empty((N, Ho, Wo, C), dtype=a.dtype, device=a.device)
        rc = _C.lib().sihl_resize_bilinear(_p(a), _p(addc), _p(out), N, H, W, Ho, Wo, C, _dt(a), _stream())
        check(rc, "sihl_resize_bilinear")
        ctx.geom = (N, H, W, Ho, Wo, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        N, H, W, Ho, Wo, C = ctx.geom
        da = None
        if ctx.needs_input_grad[0]:
            da = torch.empty((N, H, W, C), dtype=dout.dtype, device=dout.device)
            rc = _C.lib().sihl_resize_bilinear_bwd(_p(dout), _p(da), N, H, W, Ho, Wo, C, _dt(dout), _stream())
            check(rc, "sihl_resize_bilinear_bwd")
        return da, dout if ctx.needs_input_grad[1] else None, None


def nearest_up2_add(lo, skip):
    return NearestUp2AddFn.apply(lo, skip)


def resize_bilinear(a, size, add=None):
    return ResizeBilinearFn.apply(a, add, tuple(size))


def fuse_up2(a, b, wraw):
    return FuseUp2Fn.apply(a, b, wraw)


def up2(a):
    return Up2Fn.apply(a)


def fuse_sum(wraw, xs):
    return FuseSumFn.apply(wraw, *xs)


def blur_fuse(a, b=None, c=None, wraw=None, a_affine=None):
    """a_affine: a DeferredAffine filled by the conv block that produced ``a`` (or None)."""
    if a_affine is not None and a_affine.scale is not None:
        return BlurFuseFn.apply(a, b, c, wraw, a_affine.scale, a_affine.shift)
    return BlurFuseFn.apply(a, b, c, wraw)


# ----------------------------------------------------------------------------- MLP pieces (rows x C)
def _pad_rows(t: Tensor, mult: int) -> Tensor:
    """Zero-pad dim 0 up to a multiple of ``mult`` (weights / biases of tiny output layers)."""
    n = t.shape[0]
    m = (n + mult - 1) // mult * mult
    if m == n:
        return t
    out = t.new_zeros((m,) + tuple(t.shape[1:]))
    out[:n] = t
    return out


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b over rows, as a 1x1 convolution on the matrix cores.  Output channels are
    padded to the 16-byte vector width inside (Cout = 1 / 4 heads) and sliced back."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xd = x.detach().contiguous()
        rows, Cin = xd.shape
        vec = 8 if xd.dtype == torch.bfloat16 else 4
        Cout = weight.shape[0]
        ctx.side_ok_w, ctx.side_ok_b = weight.is_leaf, bias is None or bias.is_leaf  # (see ConvBlockFn.forward)
        prep = prepared(weight, xd.dtype)
        if prep is not None:
            w, ctx.wt = prep.w.view(prep.w.shape[0], -1), prep.wt  # [Cp][Cin], [Cin][1][1][Cp]
        else:
            w, ctx.wt = _pad_rows(weight.detach().to(xd.dtype), vec).contiguous(), None
        b = _pad_rows(bias.detach().float(), vec).contiguous() if bias is not None else None
        Cp = w.shape[0]
        y, _ = conv2d_raw(xd.view(1, 1, rows, Cin), w.view(Cp, 1, 1, Cin), b)
        ctx.save_for_backward(xd, w)
        ctx.Cout, ctx.has_bias = Cout, bias is not None
        ctx.pkeys = (weight.data_ptr(),) + ((bias.data_ptr(),) if bias is not None else ())
        y = y.view(rows, Cp)
        return y if Cp == Cout else y[:, :Cout]

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        rows, Cin = x.shape
        Cp, Cout = w.shape[0], ctx.Cout
        if Cp != Cout:
            dyp = dy.new_zeros((rows, Cp))
            dyp[:, :Cout] = dy
            dy = dyp
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1]:  # first: on the side stream it then runs beside the input gradient below
            dw = conv2d_wgrad_raw(x.view(1, 1, rows, Cin), dy.view(1, 1, rows, Cp), 1, 1, 1, 0, 1, side_ok=ctx.side_ok_w).view(Cp, Cin)[:Cout]
        if ctx.has_bias and ctx.needs_input_grad[2]:  # (before dx: the side stream then does not wait for that launch)
            db = colsum(dy, off_chain=ctx.side_ok_b)[:Cout]
        if ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else weight_for_dgrad(w.view(Cp, 1, 1, Cin), flip=False)  # [Cin][1][1][Cp]
            dx, _ = conv2d_raw(dy.view(1, 1, rows, Cp), wt)
            dx = dx.view(rows, Cin)
        _guard_shared_parameters(ctx.pkeys)
        return dx, dw, db


class LayerNormActFn(torch.autograd.Function):
    """y = act(LayerNorm(z) * gamma + beta) over rows."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, act):
        zd = z.detach().contiguous()
        rows, C = zd.shape
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y = torch.empty_like(zd)
        mean = torch.empty(rows, dtype=torch.float32, device=zd.device)
        rstd = torch.empty_like(mean)
        rc = _C.lib().sihl_layernorm_act(_p(zd), _p(y), rows, C, _p(g), _p(b), eps, ACT[act], _p(mean), _p(rstd),
                                         _dt(zd), _stream())
        check(rc, "sihl_layernorm_act")
        ctx.save_for_backward(zd, g, b, mean, rstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        z, g, b, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        rows, C = z.shape
        lib = _C.lib()
        ws = workspace(_sized("sihl_layernorm_act_bwd_ws_bytes", rows, C), z.device)
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
        dbeta = torch.empty_like(dgamma)
        rc = lib.sihl_layernorm_act_bwd(_p(z), _p(dy), _p(dz), rows, C, _p(g), _p(b), _p(mean), _p(rstd),
                                        ACT[ctx.act], _p(dgamma), _p(dbeta), _dt(z), _p(ws), ws.numel(), _stream())
        check(rc, "sihl_layernorm_act_bwd")
        return dz, dgamma, dbeta, None, None


MLP_KERNEL = os.environ.get("SIHL_MLP_KERNEL", "rows")  # A/B switch: "rows" = activations in registers (mlp_rows.hip),
# "tile" = the 128-row LDS tile of mlp_fused.hip


class _MLPPlan:
    """Pointer tables of one MLP for sihl_mlp_fwd, rebuilt when a parameter (or its prepared bf16 copy) moved or changed."""
    __slots__ = ("key", "w", "w_rows", "bias", "gamma", "beta", "keep", "dims")


def _mlp_plan(linears, norms, dtype):
    key = tuple((m.weight.data_ptr(), m.weight._version, getattr(getattr(m.weight, "_sihl_prepared", None), "version", None),
                 0 if m.bias is None else m.bias.data_ptr()) for m in linears) + \
        tuple((n.weight.data_ptr(), n.bias.data_ptr()) for n in norms) + (MLP_KERNEL, _PARAM_GEN)
    plan = getattr(linears[0], "_sihl_mlp_plan", None)
    if plan is not None and plan.key == key:
        return plan
    plan = _MLPPlan()
    keep, ws = [], []
    for m in linears:
        prep = prepared(m.weight, dtype)
        w = prep.w.view(prep.w.shape[0], -1) if prep is not None else m.weight.detach().to(dtype).contiguous()
        keep.append(w)
        ws.append(w.data_ptr())
    n = len(linears)
    plan.w = (ctypes.c_void_p * n)(*ws)
    plan.bias = (ctypes.c_void_p * n)(*[None if m.bias is None else m.bias.data_ptr() for m in linears])
    plan.gamma = (ctypes.c_void_p * max(1, len(norms)))(*[x.weight.data_ptr() for x in norms])
    plan.beta = (ctypes.c_void_p * max(1, len(norms)))(*[x.bias.data_ptr() for x in norms])
    plan.w_rows = None
    if MLP_KERNEL == "rows" and len(linears) > 1 and all(w.shape[1] % 16 == 0 for w in keep[1:]):
        # sihl_mlp_rows_fwd keeps a row's activations in the accumulator registers' channel order: the weights of the layers
        # behind the first are read with that K order (one permuted copy per weight version, made here)
        rows_w = [keep[0]]
        for w in keep[1:]:
            wp = torch.empty_like(w)
            check(_C.lib().sihl_mlp_permute_k(_p(w), _p(wp), w.shape[0], w.shape[1], _stream()), "sihl_mlp_permute_k")
            rows_w.append(wp)
        keep = keep + rows_w
        plan.w_rows = (ctypes.c_void_p * n)(*[w.data_ptr() for w in rows_w])
    plan.keep, plan.key = keep, key
    # not stored during a stream capture (same rule as weight_khwc / bn_eval_affine): the permuted copies would live in the
    # graph's private pool and sihl_mlp_permute_k would only have been RECORDED, not run, when a later eager call reads them
    if (prepared(linears[0].weight, dtype) is not None or not torch.is_grad_enabled()) \
            and not torch.cuda.is_current_stream_capturing():
        linears[0]._sihl_mlp_plan = plan  # (casts made on the fly are cached too: the key holds the master weight's version)
    return plan


def mlp_fused_supported(x: Tensor, linears, norms, act: Optional[str]) -> bool:
    """True when sihl_mlp_fwd covers this MLP on this input: bf16 rows on a HIP device, every hidden layer C wide with a
    LayerNorm, widths <= 256 and multiples of 8, fp32 contiguous norm / bias vectors."""
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1 or not linears:
        return False
    hidden = linears[:-1]
    if len(norms) != len(hidden):
        return False
    C = hidden[0].weight.shape[0] if hidden else x.shape[1]
    if any(tuple(m.weight.shape) != (C, x.shape[1] if i == 0 else C) for i, m in enumerate(hidden)):
        return False
    if linears[-1].weight.shape[1] != C or any(m.weight.dtype != torch.float32 for m in linears):
        return False
    if any(n.weight is None or n.weight.dtype != torch.float32 or tuple(n.normalized_shape) != (C,) for n in norms):
        return False
    if len({n.eps for n in norms}) > 1:
        return False
    return bool(_C.lib().sihl_mlp_fwd_supported(x.shape[0], x.shape[1], C, linears[-1].weight.shape[0], len(hidden),
                                                ACT[act], BF16)) and x.shape[0] * x.stride(0) * 2 < (1 << 31) \
        and x.stride(0) % 8 == 0


def mlp_fused(x: Tensor, linears, norms, act: Optional[str]) -> Tensor:
    """The whole MLP - [Linear -> LayerNorm -> act] * len(norms) -> Linear - over the rows of x in ONE launch (inference;
    sihl_mlp_fwd).  Returns (rows, Cout) as a view of a buffer padded to the 16-byte vector width, like ``linear``."""
    rows, Cin = x.shape
    Cout = linears[-1].weight.shape[0]
    C = linears[0].weight.shape[0] if norms else Cin
    plan = _mlp_plan(linears, norms, x.dtype)
    Cp = (Cout + 7) // 8 * 8
    out = torch.empty((rows, Cp), dtype=x.dtype, device=x.device)
    lib = _C.lib()
    if plan.w_rows is not None and lib.sihl_mlp_rows_supported(rows, Cin, C, Cout, len(norms), ACT[act], BF16):
        rc = lib.sihl_mlp_rows_fwd(_p(x), x.stride(0), rows, Cin, C, len(norms), plan.w_rows, plan.bias, plan.gamma,
                                   plan.beta, norms[0].eps, ACT[act], Cout, _p(out), Cp, BF16, _stream())
        check(rc, "sihl_mlp_rows_fwd")
        return out if Cp == Cout else out[:, :Cout]
    rc = _C.lib().sihl_mlp_fwd(_p(x), x.stride(0), rows, Cin, C, len(norms), plan.w, plan.bias, plan.gamma, plan.beta,
                               norms[0].eps if norms else 0.0, ACT[act], Cout, _p(out), Cp, BF16, _stream())
    check(rc, "sihl_mlp_fwd")
    return out if Cp == Cout else out[:, :Cout]


class _MlpCall(ctypes.Structure):  # sihl_mlp_call of include/sihl_hip.h
    _fields_ = [("x", ctypes.c_void_p), ("x_stride", ctypes.c_long), ("rows", ctypes.c_long), ("Cin", ctypes.c_int),
                ("C", ctypes.c_int), ("nhidden", ctypes.c_int), ("Cout", ctypes.c_int), ("out_stride", ctypes.c_int),
                ("eps", ctypes.c_float), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p),
                ("beta", ctypes.c_void_p), ("out", ctypes.c_void_p)]


def mlp_fused_multi(x: Tensor, mlps, act: Optional[str]):
    """Several MLPs over the SAME rows x in ONE launch (sihl_mlp_rows_fwd_multi: a detection head's class and box MLPs run
    over the same few thousand selected rows; each alone fills a tenth of the chip for the same time).  mlps: list of
    (linears, norms).  Returns the list of outputs, or None when a member is outside the register kernel's shapes (the
    caller then runs them one by one)."""
    if len(mlps) < 2 or len(mlps) > 4 or MLP_KERNEL != "rows":
        return None
    lib = _C.lib()
    rows, Cin = x.shape
    calls = (_MlpCall * len(mlps))()
    outs, keep = [], []
    for k, (linears, norms) in enumerate(mlps):
        if not norms or not mlp_fused_supported(x, linears, norms, act):
            return None
        Cout, C = linears[-1].weight.shape[0], linears[0].weight.shape[0]
        plan = _mlp_plan(linears, norms, x.dtype)
        if plan.w_rows is None or not lib.sihl_mlp_rows_supported(rows, Cin, C, Cout, len(norms), ACT[act], BF16):
            return None
        Cp = (Cout + 7) // 8 * 8
        out = torch.empty((rows, Cp), dtype=x.dtype, device=x.device)
        c = calls[k]
        c.x, c.x_stride, c.rows, c.Cin, c.C, c.nhidden, c.Cout, c.out_stride = _p(x), x.stride(0), rows, Cin, C, len(norms), Cout, Cp
        c.eps = norms[0].eps
        c.w, c.bias = ctypes.cast(plan.w_rows, ctypes.c_void_p), ctypes.cast(plan.bias, ctypes.c_void_p)
        c.gamma, c.beta = ctypes.cast(plan.gamma, ctypes.c_void_p), ctypes.cast(plan.beta, ctypes.c_void_p)
        c.out = _p(out)
        keep.append(plan)
        outs.append(out if Cp == Cout else out[:, :Cout])
    check(lib.sihl_mlp_rows_fwd_multi(ctypes.cast(calls, ctypes.c_void_p), len(mlps), ACT[act], BF16, _stream()),
          "sihl_mlp_rows_fwd_multi")
    return outs


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias)


def layernorm_act(z, gamma, beta, eps=1e-5, act="silu"):
    return LayerNormActFn.apply(z, gamma, beta, eps, act)


# ----------------------------------------------------------------------------- detection loss
class ODLossFn(torch.autograd.Function):
    """losses[5] = (total, location, box, class, iou) of ObjectDetection.training_step from the head's four outputs and
    the matching's target tensors, and - in the same launch - the gradients of the total with respect to those outputs
    (sihl_od_loss): ~240 small ATen launches of loss arithmetic and its autograd become two."""

    @staticmethod
    def forward(ctx, loc, iou, box, cls, t):
        loc, iou, box, cls = (x.detach().contiguous() for x in (loc, iou, box, cls))
        dev = loc.device
        n1, R, C = loc.numel(), box.shape[0], cls.shape[1]
        lib = _C.lib()
        ws = workspace(lib.sihl_od_loss_ws_bytes(n1, R), dev)
        d_loc, d_iou, d_box, d_cls = (torch.empty_like(x) for x in (loc, iou, box, cls))
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        f32 = lambda v: v.detach().float().contiguous()  # noqa: E731
        rc = lib.sihl_od_loss(_p(loc), _p(iou), _p(box), _p(cls), _p(f32(t.loc_target)), _p(f32(t.rel_iou)),
                              _p(f32(t.cand_offsets)), _p(f32(t.cand_scales)), _p(f32(t.tgt_box)), _p(f32(t.wts)),
                              _p(t.tgt_cls.contiguous()), _p(f32(t.loc_norm)), _p(f32(t.iou_norm)), _p(f32(t.wsum)),
                              _p(t.none_matched), n1, R, C, _p(d_loc), _p(d_iou), _p(d_box), _p(d_cls), _p(losses),
                              _dt(loc), _p(ws), ws.numel(), _stream())
        check(rc, "sihl_od_loss")
        ctx.save_for_backward(d_loc, d_iou, d_box, d_cls)
        return losses

    @staticmethod
    def backward(ctx, g):
        grads = list(ctx.saved_tensors)
        # only the total is differentiated (the four components are reported values): scale the stored gradients of the
        # total by its upstream gradient, all four tensors in one multi-tensor launch
        out = torch._foreach_mul(grads, g[0])  # out of place: the saved gradients stay valid for a second backward
        return out[0], out[1], out[2], out[3], None


def od_loss(loc, iou, box, cls, t):
    return ODLossFn.apply(loc, iou, box, cls, t)


# ----------------------------------------------------------------------------- decode
def topk_rows(x: Tensor, B: int, P: int, K: int, estride: int = 1):
    vals = torch.empty((B, K), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, K), dtype=torch.int32, device=x.device)
    rc = _C.lib().sihl_topk_rows(_p(x), B, P, K, estride, _p(vals), _p(idx), _dt(x), _stream())
    check(rc, "sihl_topk_rows")
    return vals, idx


def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    B, P, C = src.shape
    K = idx.shape[1]
    out = torch.empty((B, K, C), dtype=src.dtype, device=src.device)
    rc = _C.lib().sihl_gather_rows(_p(src), _p(idx), _p(out), B, P, K, C, _dt(src), _stream())
    check(rc, "sihl_gather_rows")
    return out


def _levels_arr(level_hw):
    flat = [int(v) for hw in level_hw for v in hw]
    return (ctypes.c_int * len(flat))(*flat)


def od_decode(top_vals, top_idx, cls_logits, box_raw, level_hw, full_wh):
    B, K = top_vals.shape
    ncls = cls_logits.shape[-1]
    dev = top_vals.device
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    classes = torch.empty((B, K), dtype=torch.int64, device=dev)
    boxes = torch.empty((B, K, 4), dtype=torch.float32, device=dev)
    num = torch.empty((B,), dtype=torch.int64, device=dev)
    # (B, K, n) views of vector-padded MLP outputs are read through their row stride: no compaction copy
    def rows(t):
        t2 = t.reshape(B * K, t.shape[-1]) if t.dim() != 2 else t
        return t2 if t2.stride(1) == 1 else t2.contiguous()
    cls_logits, box_raw = rows(cls_logits), rows(box_raw)
    rc = _C.lib().sihl_od_decode(_p(top_vals), _p(top_idx), _p(cls_logits), cls_logits.stride(0), _p(box_raw),
                                 box_raw.stride(0), _levels_arr(level_hw), len(level_hw), B, K, ncls, int(full_wh[0]),
                                 int(full_wh[1]), _p(scores), _p(classes), _p(boxes), _p(num), _dt(cls_logits), _stream())
    check(rc, "sihl_od_decode")
    return num, scores, classes, boxes


def iseg_mask_decode(mask_feats: Tensor, dyn: Tensor, top_idx: Tensor, level_hw, out_hw) -> Tensor:
    """CondInst masks (B, K, H, W): mask_feats (B, h, w, 8) NHWC, dyn (B*K, 169) parameter rows (any row stride),
    top_idx (B, K) int32 pyramid positions; dynamic 10->8->8->1 network + sigmoid + bilinear resize in one kernel."""
    _require_gpu(mask_feats)
    B, h, w, c = mask_feats.shape
    K = top_idx.shape[1]
    if c != 8 or dyn.shape[-1] != 169 or dyn.dtype != mask_feats.dtype or dyn.stride(-1) != 1:
        raise ValueError("iseg_mask_decode: 8 mask channels and contiguous 169-parameter rows of the same dtype")
    mask_feats = mask_feats.contiguous()
    dyn2 = dyn.reshape(B * K, 169) if dyn.dim() != 2 else dyn
    H, W = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((B, K, H, W), dtype=mask_feats.dtype, device=mask_feats.device)
    rc = _C.lib().sihl_iseg_mask_decode(_p(mask_feats), _p(dyn2), dyn2.stride(0), _p(top_idx.contiguous()),
                                        _levels_arr(level_hw), len(level_hw), B, K, h, w, H, W, _p(out),
                                        _dt(mask_feats), _stream())
    check(rc, "sihl_iseg_mask_decode")
    return out


def od_anchors(level_hw, device):
    P = sum(h * w for h, w in level_hw)
    offsets = torch.empty((P, 4), dtype=torch.float32, device=device)
    scales = torch.empty_like(offsets)
    rc = _C.lib().sihl_od_anchors(_levels_arr(level_hw), len(level_hw), _p(offsets), _p(scales), _stream())
    check(rc, "sihl_od_anchors")
    return offsets, scales


# ----------------------------------------------------------------------------- semantic-segmentation pieces
class UAFMFn(torch.autograd.Function):
    """out = x1*a + x2*(1-a), a = sigmoid(conv3x3_{4->1}(channel mean/max of x1 and x2) + b)  (NHWC)."""

    @staticmethod
    def forward(ctx, x1, x2, conv_w, conv_b):
        x1, x2 = x1.detach().contiguous(), x2.detach().contiguous()
        N, H, W, C = x1.shape
        w = conv_w.detach().float().contiguous()  # (1,4,3,3) OIHW
        b = conv_b.detach().float().contiguous() if conv_b is not None else None
        dev = x1.device
        out = torch.empty_like(x1)
        stats = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
        arg = torch.empty((N, H, W, 2), dtype=torch.int32, device=dev)
        alpha = torch.empty((N, H, W), dtype=torch.float32, device=dev)
        rc = _C.lib().sihl_uafm_fwd(_p(x1), _p(x2), _p(w), _p(b), _p(out), _p(stats), _p(arg), _p(alpha), N, H, W, C,
                                    _dt(x1), _stream())
        check(rc, "sihl_uafm_fwd")
        ctx.save_for_backward(x1, x2, w, stats, arg, alpha)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x1, x2, w, stats, arg, alpha = ctx.saved_tensors
        dout = dout.contiguous()
        N, H, W, C = x1.shape
        lib = _C.lib()
        ws = workspace(lib.sihl_uafm_bwd_ws_bytes(N, H, W), x1.device)
        dx1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        dx2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        dw = torch.empty(36, dtype=torch.float32, device=x1.device)
        db = torch.empty(1, dtype=torch.float32, device=x1.device)
        rc = lib.sihl_uafm_bwd(_p(dout), _p(x1), _p(x2), _p(w), _p(stats), _p(arg), _p(alpha), _p(dx1), _p(dx2),
                               _p(dw), _p(db), N, H, W, C, _dt(x1), _p(ws), ws.numel(), _stream())
        check(rc, "sihl_uafm_bwd")
        return dx1, dx2, dw.view(1, 4, 3, 3), db if ctx.has_bias else None


def uafm(x1, x2, conv_w, conv_b):
    return UAFMFn.apply(x1, x2, conv_w, conv_b)


def softmax_max_resize(logits: Tensor, size):
    """logits (N,h,w,C) -> (scores fp32 (N,H,W), classes int64 (N,H,W)): nearest resize + softmax + max."""
    logits = logits.detach().contiguous()
    _require_gpu(logits)
    N, h, w, C = logits.shape
    H, W = size
    scores = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    classes = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
    rc = _C.lib().sihl_softmax_max_resize(_p(logits), _p(scores), _p(classes), N, h, w, C, H, W, _dt(logits),
                                          _stream())
    check(rc, "sihl_softmax_max_resize")
    return scores, classes


class CEResizeFn(torch.autograd.Function):
    """mean cross-entropy of nearest-resized logits (N,h,w,C) against targets (N,H,W) with ignore_index."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        lg = logits.detach().contiguous()
        _require_gpu(lg)
        N, h, w, C = lg.shape
        tg = targets.detach().to(torch.int64).contiguous()
        H, W = tg.shape[1:]
        inv_count = 1.0 / (tg != ignore_index).sum().to(torch.float32).reshape(1)  # device scalar, no host sync
        dl = torch.empty_like(lg)
        acc = torch.empty(2 + 2 * 2048, dtype=torch.float32, device=lg.device)  # SIHL_CE_ACC_FLOATS
        rc = _C.lib().sihl_ce_resize(_p(lg), _p(tg), int(ignore_index), _p(inv_count), _p(dl), _p(acc), N, h, w, C, H,
                                     W, _dt(lg), _stream())
        check(rc, "sihl_ce_resize")
        ctx.save_for_backward(dl)
        return acc[0] * inv_count[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g.to(dl.dtype), None, None


def ce_resize(logits, targets, ignore_index):
    return CEResizeFn.apply(logits, targets, ignore_index)


# ----------------------------------------------------------------------------- ResNet residual merge
class AddReluFn(torch.autograd.Function):
    """out = relu(a + b); both inputs receive dout * (out > 0)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.detach().contiguous(), b.detach().contiguous()
        _require_gpu(a)
        out = torch.empty_like(a)
        rc = _C.lib().sihl_add_act(_p(a), _p(b), _p(out), a.numel(), ACT["relu"], _dt(a), _stream())
        check(rc, "sihl_add_act")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        g = affine_act_bwd(out, dout.contiguous(), None, None, "relu")
        return g, g


def add_relu(a, b):
    return AddReluFn.apply(a, b)


class MaxPool3x3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on an NHWC tensor (the ResNet stem's pool).  Forward stores the winning tap per element
    (one byte); backward gathers per input pixel - one pass, no atomics (ATen's NHWC backward: 406 us at bs 32, 256^2
    x 64; this pair: see profiles)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xd = x.detach().contiguous()
        N, H, W, C = xd.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, Ho, Wo, C), dtype=xd.dtype, device=xd.device)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=xd.device)
        rc = _C.lib().sihl_maxpool3x3s2_fwd(_p(xd), _p(y), _p(idx), N, H, W, C, _dt(xd), _stream())
        check(rc, "sihl_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.in_shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        rc = _C.lib().sihl_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), N, H, W, C, _dt(dy), _stream())
        check(rc, "sihl_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x_nhwc: Tensor) -> Tensor:
    return MaxPool3x3s2Fn.apply(x_nhwc)
