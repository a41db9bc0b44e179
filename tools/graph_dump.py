"""Developer probe (GPU box): capture the training step as the graph Trainer does, after eager warm-up steps in the
given stream mode, and dump the captured HIP graph's topology (DOT) - without replaying it.
usage: graph_dump.py <off|all> <out.dot>"""
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
