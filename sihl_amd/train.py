"""Minimal data-parallel training loop for the hot path (one process per GPU, RCCL over xGMI).

Counterpart of what the reference delegates to Lightning (src/sihl/lightning_module.py:68-120 training_step,
:179-245 configure_optimizers, examples/object_detection.py:288-296 Trainer flags):
  * loss = sum of the heads' training_step losses on ONE extract_features pass,
  * AdamW parameter groups: backbone lr x backbone_lr_factor, no weight decay on biases / norm parameters,
  * gradient-norm clipping (gradient_clip_val), bf16 autocast for the backbone with fp32 loss islands,
  * DP: minibatch sharded across ranks, per-replica BatchNorm statistics (Lightning-DDP default), gradients
    averaged with bucketed all-reduces launched from grad-ready hooks so they overlap the rest of backward.
"""
import os
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor, nn

NO_DECAY_TYPES = (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d, nn.Embedding)


def configure_optimizer(model: nn.Module, optimizer=torch.optim.AdamW, lr: float = 1e-4, weight_decay: float = 1e-4,
                        backbone_lr_factor: float = 0.1, **kw) -> torch.optim.Optimizer:
    """Parameter grouping of the reference's configure_optimizers (lightning_module.py:179-222)."""
    backbone = {id(p) for p in model.backbone.parameters()} if hasattr(model, "backbone") else set()
    no_decay = set()
    for m in model.modules():
        for pn, p in m.named_parameters(recurse=False):
            if pn.endswith("bias") or isinstance(m, NO_DECAY_TYPES):
                no_decay.add(id(p))
    groups: Dict[Any, List[Tensor]] = {}
    for p in model.parameters():
        if not p.requires_grad:
            continue
        key = (id(p) in backbone, id(p) in no_decay)
        groups.setdefault(key, []).append(p)
    param_groups = []
    for (is_bb, is_nd), params in groups.items():
        param_groups.append({"params": params, "lr": lr * (backbone_lr_factor if is_bb else 1.0),
                             "weight_decay": 0.0 if is_nd else weight_decay})
    if optimizer in (torch.optim.AdamW, torch.optim.Adam) and "fused" not in kw and "foreach" not in kw:
        # one multi-tensor kernel per group instead of a host loop over ~400 parameters
        kw["fused"] = all(p.is_cuda for g in param_groups for p in g["params"])
    if kw.get("capturable") and not all(p.is_cuda for g in param_groups for p in g["params"]):
        kw.pop("capturable")
    return optimizer(param_groups, lr=lr, weight_decay=weight_decay, **kw)


class GradientAverager:
    """Bucketed, overlapped gradient all-reduce (mean) across the data-parallel group.

    Parameters are packed into ~bucket_mb fp32 buckets in REVERSE registration order (roughly the order
    autograd produces their gradients).  A post-accumulate hook counts finished gradients; when a bucket is
    complete its gradients are packed with ONE multi-tensor copy and its all-reduce is issued asynchronously
    (RCCL runs it on its own stream over xGMI while backward continues).  ``finish()`` waits and scatters the
    averages back, again one multi-tensor copy per bucket."""

    def __init__(self, params: Sequence[Tensor], group=None, bucket_mb: float = 32.0, force: bool = False):
        """force: build the buckets and issue the all-reduces even in a one-rank group (single-GPU rehearsal of the
        multi-GPU step: same hooks, same streams, same collective calls)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[Dict[str, Any]] = []
        self.where: Dict[int, Any] = {}
        self._waits: List[Any] = []
        self.active = self.world > 1 or (force and dist.is_initialized())
        if not self.active:
            return
        cap = int(bucket_mb * (1 << 20) // 4)
        cur: List[Tensor] = []
        size = 0
        for p in reversed(self.params):
            if cur and size + p.numel() > cap:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self._close(cur)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        self.use_avg = dist.get_backend(group) == "nccl"

    def _close(self, ps: List[Tensor]) -> None:
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            # same shape AND strides as the parameter (conv weights are channels_last), so the multi-tensor copy
            # below is a straight memcpy per parameter
            views.append(flat[off: off + p.numel()].as_strided(p.shape, p.stride()) if p.is_contiguous(
                memory_format=torch.channels_last) and p.dim() == 4 else flat[off: off + p.numel()].view(p.shape))
            off += p.numel()
        b = {"flat": flat, "params": ps, "views": views, "pending": len(ps), "work": None}
        for p in ps:
            self.where[id(p)] = b
        self.buckets.append(b)

    def _launch(self, b) -> None:
        grads = [p.grad for p in b["params"]]
        side = None
        if any(g is not None and g.is_cuda for g in grads):
            from sihl_amd import ops

            side = ops.side_stream_in_use()
        if side is None:
            self._pack_and_reduce(b, grads)
            return
        # Weight gradients of this bucket may still be queued on the wgrad side stream.  Rather than stalling the
        # main stream (the dgrad chain) until they are final, the bucket is packed ON the side stream, behind them,
        # and the collective is ordered after that stream: the main stream never waits inside backward.
        side.wait_stream(torch.cuda.current_stream())  # gradients produced on the main stream (norm / bias grads)
        with torch.cuda.stream(side):
            self._pack_and_reduce(b, grads)

    def _pack_and_reduce(self, b, grads) -> None:
        if any(g is None for g in grads):  # a parameter without gradient this step contributes zeros
            b["flat"].zero_()
        pairs = [(v, g) for v, g in zip(b["views"], grads) if g is not None]
        torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])  # ONE multi-tensor kernel per bucket
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def _on_grad(self, p: Tensor) -> None:
        b = self.where[id(p)]
        b["pending"] -= 1
        if b["pending"] == 0:  # every gradient of the bucket is final: pack it and start its all-reduce
            self._launch(b)

    def wait_ms(self) -> float:
        """Mean time per step (over the last <= 64) that the step's stream waited in ``finish()`` for the all-reduces to
        complete - the part of the exchange that backward did not hide.  Synchronises the device."""
        if not self._waits:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._waits) / len(self._waits)

    def finish(self) -> None:
        if not self.active:
            return
        for b in self.buckets:
            if b["work"] is None:  # some parameter of this bucket got no gradient this step
                self._launch(b)
        timed = bool(self.buckets) and self.buckets[0]["flat"].is_cuda
        if timed:  # how long the step's stream stands still for the collectives (read back lazily: wait_ms())
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for b in self.buckets:
            b["work"].wait()
        if timed:
            ev1.record()
            self._waits.append((ev0, ev1))
            if len(self._waits) > 64:
                del self._waits[:-64]
        for b in self.buckets:
            if not self.use_avg:
                b["flat"].div_(self.world)
            pairs = [(p.grad, v) for p, v in zip(b["params"], b["views"]) if p.grad is not None]
            torch._foreach_copy_([g for g, _ in pairs], [v for _, v in pairs])
            b["pending"], b["work"] = len(b["params"]), None


def broadcast_parameters(model: nn.Module, src: int = 0, group=None) -> None:
    """Identical initial replicas: rank ``src``'s parameters and buffers are sent to every rank once."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def _tree_tensors(obj, out: List[Tensor]) -> List[Tensor]:
    """All tensors of a nest of dicts / lists / tuples, in a fixed traversal order."""
    if isinstance(obj, Tensor):
        out.append(obj)
    elif isinstance(obj, dict):
        for k in sorted(obj):
            _tree_tensors(obj[k], out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tree_tensors(v, out)
    return out


def _tree_clone(obj):
    if isinstance(obj, Tensor):
        return obj.clone()
    if isinstance(obj, dict):
        return {k: _tree_clone(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_clone(v) for v in obj)
    return obj


class Trainer:
    """One optimisation step = forward (backbone -> neck -> heads' training_step) + backward +
    gradient all-reduce + clip + optimizer step.

    ``graph=True`` (single-process runs on a HIP device): after ``GRAPH_WARMUP`` eager steps the whole step -
    forward, losses, backward, clipping, AdamW - is captured into ONE HIP graph per input signature (tensor shapes
    of the images and of every target) and later steps copy their batch into the graph's static buffers and replay
    it: ~2 000 kernel launches per step without host work in between.  The step contains no host synchronisation
    (see ObjectDetection.training_step), which is what makes it capturable.  Batches with a new signature run
    eagerly until their own graph exists.  Multi-process runs stay eager: their bucketed all-reduces are issued
    from autograd hooks and overlap backward."""

    GRAPH_WARMUP = 2

    def __init__(self, model: nn.Module, optimizer: Optional[torch.optim.Optimizer] = None, group=None,
                 grad_clip_norm: Optional[float] = 0.1, autocast_dtype: Optional[torch.dtype] = None,
                 bucket_mb: float = 32.0, graph: bool = False, scheduler=None,
                 scheduler_kwargs: Optional[Dict[str, Any]] = None, wgrad_stream: str = "all",
                 force_buckets: bool = False, **opt_kw):
        self.model = model
        # eager steps launch weight gradients on a second HIP stream beside the dgrad chain (ops.wgrad_side_stream):
        # "off" | "small" | "all".  Measured on one MI355X, flagship step: eager off 854, small 866, all 912 img/s; a
        # captured graph with those fork/join edges replays SLOWER (830 vs 854 single-stream), so captures stay
        # single-stream.
        self.wgrad_stream = wgrad_stream
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        on_gpu = all(p.is_cuda for p in model.parameters())
        self.use_graph = bool(graph) and world == 1 and on_gpu
        if self.use_graph:
            import sihl_amd

            if not sihl_amd.graph_replay_safe() and not os.environ.get("SIHL_ALLOW_GRAPH_PACKET_CAPTURE"):
                raise RuntimeError(
                    "Trainer(graph=True): HIP-graph replays of this process would go through ROCm's graph packet capture "
                    "(DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0, or the HIP runtime was already initialised when sihl_amd was "
                    "imported, so its default came too late).  On that path device memory the process allocates beside the "
                    "graph can overwrite the replay's kernel arguments: silently wrong results or a GPU memory fault.  Set "
                    "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment (or import sihl_amd before the first device call); "
                    "SIHL_ALLOW_GRAPH_PACKET_CAPTURE=1 overrides this check.")
            # a graph Trainer is single-stream THROUGHOUT, warm-up steps included (a capture that
            # followed two-stream eager steps faulted on a later replay; the captured step never uses a second stream)
            self.wgrad_stream = opt_kw.pop("_graph_warmup_stream", "off")  # private: tools/graph_phase_probe.py only
        else:
            opt_kw.pop("_graph_warmup_stream", None)
        if self.wgrad_stream != "off" and on_gpu:
            # weight gradients written by the side stream are handed to autograd while that stream may still be
            # writing them; that is safe only while AccumulateGrad takes the tensor as it is (no kernel): fp32 parameters,
            # 4-D weights stored channels_last (the kernels' [O][KH][KW][I] layout), one use per step.  Anything else
            # keeps the weight gradients on the main stream.
            for name, p in model.named_parameters():
                bad = p.dtype != torch.float32 or (p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and
                                                   not p.is_contiguous(memory_format=torch.channels_last))
                if getattr(p, "_backward_hooks", None):  # a tensor hook is a main-stream reader of the gradient, before the join
                    import warnings
                    warnings.warn(f"wgrad_stream={self.wgrad_stream!r}: {name} has a gradient hook (it would read the gradient "
                                  "while the side stream may still be writing it): weight gradients stay on the main stream")
                    self.wgrad_stream = "off"
                    break
                if bad:
                    import warnings
                    warnings.warn(f"wgrad_stream={self.wgrad_stream!r} needs fp32 parameters with channels_last conv "
                                  f"weights ({name} is not): weight gradients stay on the main stream")
                    self.wgrad_stream = "off"
                    break
        if self.use_graph and optimizer is None:
            opt_kw.setdefault("capturable", True)  # optimizer step counters live on the device
        self.optimizer = optimizer or configure_optimizer(model, **opt_kw)
        self.grad_clip_norm = grad_clip_norm
        self.autocast_dtype = autocast_dtype
        broadcast_parameters(model, group=group)
        self.averager = GradientAverager(list(model.parameters()), group=group, bucket_mb=bucket_mb,
                                         force=force_buckets)
        self._params = list(model.parameters())  # walked once, not per step (clip_grad_norm_)
        self._graphs: Dict[Any, Any] = {}
        self._seen: Dict[Any, int] = {}
        self.scheduler = self._make_scheduler(scheduler, dict(scheduler_kwargs or {}))
        # bf16 operand copies of all conv / linear weights, rewritten by one kernel after every optimizer step
        self.prepared = None
        if autocast_dtype == torch.bfloat16 and on_gpu:
            from sihl_amd import ops
            self.prepared = ops.PreparedWeights(model, torch.bfloat16)

    def _make_scheduler(self, scheduler, kw: Dict[str, Any]):
        """Per-step learning-rate schedule of the reference (lightning_module.py:226-241): ``scheduler(optimizer,
        **kwargs)``, preceded by a LinearLR warm-up (start factor 0.01) of ``kwargs["warmup"]`` steps when given.

        In graph mode the optimizer reads its learning rates from device tensors (a replay cannot see a changed
        Python float): the schedule then runs on a parameter-free shadow optimizer with the same groups and its
        values are written into those tensors before every replay."""
        self._lr_tensors = None
        if scheduler is None:
            return None
        target = self.optimizer
        if self.use_graph:
            self._lr_tensors = []
            shadow_groups = []
            for g in self.optimizer.param_groups:
                dev = g["params"][0].device
                t = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
                shadow_groups.append({"params": [torch.zeros((), requires_grad=True)], "lr": float(g["lr"])})
                g["lr"] = t
                self._lr_tensors.append(t)
            target = torch.optim.SGD(shadow_groups, lr=1.0)
        warmup = kw.pop("warmup", None)
        sched = scheduler(target, **kw)
        if warmup:
            lrs = torch.optim.lr_scheduler
            sched = lrs.SequentialLR(target, [lrs.LinearLR(target, start_factor=0.01, total_iters=warmup), sched],
                                     milestones=[warmup])
        self._shadow = target if self.use_graph else None
        self._push_lrs()
        return sched

    def _push_lrs(self) -> None:
        if self._lr_tensors is not None:
            for t, g in zip(self._lr_tensors, self._shadow.param_groups):
                t.fill_(float(g["lr"]))

    def _step_scheduler(self) -> None:
        if self.scheduler is not None:
            if self._shadow is not None:
                self._shadow.step()  # keeps torch's "optimizer.step() before lr_scheduler.step()" bookkeeping quiet
            self.scheduler.step()
            self._push_lrs()

    def forward_loss(self, images: Tensor, targets: List[Any]):
        from sihl_amd import ops  # the HIP library loads lazily: CPU oracle models never need it

        with ops.deferred_bn_counters():
            return self._forward_loss(images, targets)

    def _side_priority(self) -> int:
        # with a process group the side stream is a high-priority one (a hardware queue of its own, see ops)
        return -1 if dist.is_initialized() else 0

    def _prematch(self, images: Tensor, targets: List[Any]):
        """Heads whose loss has a part that depends on the targets only (``prematch``: anchor matching of the detection
        head, ~1 ms of small launches at bs 32) compute it on the side stream while the backbone and the neck run.
        Returns the stream to wait for before the heads' training_step, or None."""
        if (self.wgrad_stream == "off" or not images.is_cuda or torch.cuda.is_current_stream_capturing()
                or not any(hasattr(h, "prematch") for h in self.model.heads)):
            return None
        from sihl_amd import ops

        side = ops.side_stream(images.device, self._side_priority())
        main = torch.cuda.current_stream()
        side.wait_stream(main)  # the targets' uploads, and every earlier reader of the blocks the side stream reuses
        with torch.cuda.stream(side):
            for head, target in zip(self.model.heads, targets):
                if hasattr(head, "prematch") and isinstance(target, dict):
                    head.prematch(tuple(images.shape[2:]), device=images.device, **target)
                    for v in vars(head._prematched[-1]).values():
                        if isinstance(v, Tensor) and v.is_cuda:
                            v.record_stream(main)  # allocated on the side stream, read by the main stream's loss
        return side

    def _forward_loss(self, images: Tensor, targets: List[Any]):
        dev_type = images.device.type
        prematched_on = self._prematch(images, targets)
        if self.autocast_dtype is not None:
            # the backbone runs under autocast (ATen / MIOpen); the neck and heads take its bf16 level list
            # (no autocast weight-cast cache inside a graph capture: its entries would be tensors of the graph's private pool
            # outliving the capture - torch's own graph helpers capture with cache_enabled=False too)
            capturing = images.is_cuda and torch.cuda.is_current_stream_capturing()
            with torch.autocast(device_type=dev_type, dtype=self.autocast_dtype, cache_enabled=not capturing):
                levels = self.model.backbone(images)
            levels = [t if i == 0 else t.to(self.autocast_dtype) for i, t in enumerate(levels)]
            feats = self.model.neck(levels) if self.model.neck is not None else levels
        else:
            feats = self.model.extract_features(images)
        if prematched_on is not None:
            torch.cuda.current_stream().wait_stream(prematched_on)
        losses, metrics = [], {}
        for i, (head, target) in enumerate(zip(self.model.heads, targets)):
            loss, m = head.training_step(feats, **target) if isinstance(target, dict) else head.training_step(feats, target)
            losses.append(loss)
            metrics.update({f"head{i}/train/{k}": v for k, v in m.items()})
        return torch.stack(losses).sum(), metrics

    def _clip_gradients(self) -> None:
        """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (the reference's Lightning ``gradient_clip_val``): the norm of
        all gradients together, every gradient scaled by min(1, max_norm / (norm + 1e-6)).  Dense fp32 device gradients take
        sihl_grad_clip (three launches, no per-tensor results: 2.8 ms -> 0.3 ms of host time per step on 320 parameters);
        anything else the library's multi-tensor routines in the reference's order."""
        grads = [p.grad for p in self._params if p.grad is not None]
        if not grads:
            return
        g0 = grads[0]
        if g0.is_cuda:
            from sihl_amd import ops

            plan = self.__dict__.get("_clip_plan")
            sizes = tuple(g.numel() for g in grads)
            if plan is None or plan.numels != sizes or plan.map.device != g0.device:
                plan = ops.GradClipPlan(sizes, g0.device) if ops.grad_clip_supported(grads) else None
                self.__dict__["_clip_plan"] = plan
            if plan is not None and all(g.device == g0.device and g.dtype == torch.float32 for g in grads):
                plan.run(grads, self.grad_clip_norm)
                return
        if any(g.device != g0.device or g.dtype != g0.dtype or g.is_sparse for g in grads):
            torch.nn.utils.clip_grad_norm_([p for p in self._params if p.grad is not None], self.grad_clip_norm)
            return
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
        coef = torch.clamp(self.grad_clip_norm / (total + 1e-6), max=1.0)
        torch._foreach_mul_(grads, coef)

    def _backward(self, loss: Tensor) -> None:
        if self.wgrad_stream == "off" or not loss.is_cuda or torch.cuda.is_current_stream_capturing():
            loss.backward()
            return
        from sihl_amd import ops

        with ops.wgrad_side_stream(self.wgrad_stream, loss.device, self._side_priority()):  # joins on exit
            loss.backward()

    def _eager_step(self, images: Tensor, targets: List[Any]):
        self.optimizer.zero_grad(set_to_none=True)
        loss, metrics = self.forward_loss(images, targets)
        self._backward(loss)
        self.averager.finish()
        if self.grad_clip_norm is not None:
            self._clip_gradients()
        self.optimizer.step()
        if self.prepared is not None:
            self.prepared.refresh()
        self._step_scheduler()
        # detached: a caller holding on to graph-attached metrics would keep this step's autograd graph - and its
        # AccumulateGrad nodes, which remember the stream they were created on - alive into a later graph capture
        return loss.detach(), {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in metrics.items()}

    def _capture(self, images: Tensor, targets: List[Any]):
        static_images, static_targets = images.clone(), _tree_clone(targets)
        self.optimizer.zero_grad(set_to_none=True)  # gradients are (re)created inside the graph's memory pool
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                loss, metrics = self.forward_loss(static_images, static_targets)
                self._backward(loss)
                if self.grad_clip_norm is not None:
                    self._clip_gradients()
                self.optimizer.step()
                if self.prepared is not None:
                    self.prepared.refresh()
        except RuntimeError as e:
            if "captur" not in str(e):
                raise
            # (ObjectDetection and SemanticSegmentation train without a host synchronisation; the instance, keypoint,
            # quadrilateral and depth heads keep the reference's boolean-mask indexing / nonzero, whose sizes the host must read)
            raise RuntimeError("Trainer(graph=True): the step cannot be captured into a HIP graph - a head's training_step "
                               "synchronises with the device (boolean-mask indexing, nonzero, .item()); use the eager "
                               f"Trainer for this model.  [{str(e).splitlines()[0]}]") from e
        leaves = [static_images] + _tree_tensors(static_targets, [])
        return graph, leaves, loss.detach(), {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in metrics.items()}

    def step(self, images: Tensor, targets: List[Any]):
        if not self.model.training:  # (Module.train() walks every submodule: 2 ms of host time per step when unconditional)
            self.model.train()
        if images.is_cuda:
            from sihl_amd import ops
            # whatever path runs below rewrites parameters / running statistics, a graph replay by raw pointer: the eval-side
            # caches (BatchNorm affines, operand casts, MLP plans) must not survive it
            ops.bump_param_generation()
        if not self.use_graph:
            return self._eager_step(images, targets)
        leaves = [images] + _tree_tensors(targets, [])
        sig = tuple((tuple(t.shape), t.dtype) for t in leaves)
        entry = self._graphs.get(sig)
        if entry is None:
            seen = self._seen.get(sig, 0)
            self._seen[sig] = seen + 1
            if seen < self.GRAPH_WARMUP:
                return self._eager_step(images, targets)
            entry = self._graphs[sig] = self._capture(images, targets)  # capturing does not execute the step ...
        graph, static_leaves, loss, metrics = entry
        torch._foreach_copy_(static_leaves, leaves)
        graph.replay()  # ... the replay does
        if images.is_cuda:
            import sihl_amd
            from sihl_amd import ops
            if ops.side_stream_history() and not sihl_amd.graph_replay_safe():
                # (rounds 1-3 mitigation, kept for processes that replay through ROCm's graph packet capture: replays queued
                # back to back faulted after eager two-stream steps, one replay in flight at a time never did.  The cause was
                # found in round 4 - see sihl_amd/__init__.py - and with packet capture off, the default, nothing waits here.)
                torch.cuda.current_stream().synchronize()
        self._step_scheduler()
        return loss, metrics
