"""Developer probe (GPU box): capture the training step as the graph Trainer does, after eager warm-up steps in the
given stream mode, and dump the captured HIP graph's topology (DOT) - without replaying it.
usage: graph_dump.py <off|all> <out.dot> [nofused]
Also lists every memset / memcpy node with its addresses and says whether they lie inside a caching-allocator segment
(a host pointer or a stale device pointer captured into the graph would show up here without replaying anything)."""
import os
import re
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

mode, out = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3 and sys.argv[3] == "nofused":
    from sihl_amd.heads import object_detection as _od
    _od.FUSED_LOSS = False
dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1,
             autocast_dtype=torch.bfloat16, graph=True, _graph_warmup_stream=mode)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(2):
    tr._eager_step(images, targets)
# the Trainer's own capture, keeping the hipGraph_t so that its topology can be read through the HIP graph API
import ctypes

orig = torch.cuda.CUDAGraph
torch.cuda.CUDAGraph = lambda *a, **k: orig(keep_graph=True)
graph, leaves, loss, metrics = tr._capture(images, targets)
torch.cuda.CUDAGraph = orig
hip = ctypes.CDLL("libamdhip64.so")
g = ctypes.c_void_p(graph.raw_cuda_graph())
n = ctypes.c_size_t(0)
assert hip.hipGraphGetNodes(g, None, ctypes.byref(n)) == 0
nodes = (ctypes.c_void_p * n.value)()
assert hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) == 0
nr = ctypes.c_size_t(0)
assert hip.hipGraphGetRootNodes(g, None, ctypes.byref(nr)) == 0
ne = ctypes.c_size_t(0)
assert hip.hipGraphGetEdges(g, None, None, ctypes.byref(ne)) == 0
src, dst = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
assert hip.hipGraphGetEdges(g, src, dst, ctypes.byref(ne)) == 0
outdeg, indeg = {}, {}
for a_, b_ in zip(src, dst):
    outdeg[a_] = outdeg.get(a_, 0) + 1
    indeg[b_] = indeg.get(b_, 0) + 1
types_ = {}
for nd in nodes:
    t = ctypes.c_int(0)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
    types_[t.value] = types_.get(t.value, 0) + 1
leaves_n = sum(1 for nd in nodes if outdeg.get(nd, 0) == 0)
print(f"mode {mode}: {n.value} nodes, {ne.value} edges, {nr.value} roots, {leaves_n} leaves, "
      f"max fan-out {max(outdeg.values())}, max fan-in {max(indeg.values())}, node types {types_}", flush=True)
fan = [(k, v) for k, v in outdeg.items() if v > 1]
print(f"nodes with fan-out > 1: {len(fan)}; with fan-in > 1: {sum(1 for v in indeg.values() if v > 1)}")


# ---- memset / memcpy nodes: where do they point?
class _Pos(ctypes.Structure):
    _fields_ = [("x", ctypes.c_size_t), ("y", ctypes.c_size_t), ("z", ctypes.c_size_t)]


class _Pitched(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("pitch", ctypes.c_size_t), ("xsize", ctypes.c_size_t), ("ysize", ctypes.c_size_t)]


class _Memcpy3D(ctypes.Structure):
    _fields_ = [("srcArray", ctypes.c_void_p), ("srcPos", _Pos), ("srcPtr", _Pitched), ("dstArray", ctypes.c_void_p),
                ("dstPos", _Pos), ("dstPtr", _Pitched), ("extent", _Pos), ("kind", ctypes.c_int)]


class _Memset(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]


segs = [(sg["address"], sg["address"] + sg["total_size"], tuple(sg.get("segment_pool_id", (0, 0)))) for sg in torch.cuda.memory_snapshot()]


def where(p):
    if not p:
        return "NULL"
    for a, b, pool in segs:
        if a <= p < b:
            return "graph pool" if pool != (0, 0) else "allocator"
    return "OUTSIDE every allocator segment"


KINDS = {0: "H2H", 1: "H2D", 2: "D2H", 3: "D2D", 4: "default"}
tally = {}
for nd in nodes:
    t = ctypes.c_int(0)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
    if t.value == 1:  # hipGraphNodeTypeMemcpy
        prm = _Memcpy3D()
        rc = hip.hipGraphMemcpyNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(prm))
        key = ("memcpy", rc, KINDS.get(prm.kind, prm.kind), "src " + where(prm.srcPtr.ptr), "dst " + where(prm.dstPtr.ptr))
        tally.setdefault(key, []).append(prm.extent.x * max(prm.extent.y, 1) * max(prm.extent.z, 1))
    elif t.value == 2:  # hipGraphNodeTypeMemset
        prm = _Memset()
        rc = hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nd), ctypes.byref(prm))
        key = ("memset", rc, "", "", "dst " + where(prm.dst))
        tally.setdefault(key, []).append(prm.width * max(prm.height, 1) * prm.elementSize)
for key, sizes in sorted(tally.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(sizes):4d} x {key[0]} rc={key[1]} {key[2]} {key[3]} {key[4]}  bytes min {min(sizes)} max {max(sizes)}", flush=True)
