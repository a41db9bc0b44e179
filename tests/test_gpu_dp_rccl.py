"""GPU, world size 2: the data-parallel Trainer step on the HIP path - replicas stay bit-identical and the averaged
gradients equal the serial mean of the per-shard gradients (the reference gets this from Lightning's DDP,
examples/object_detection.py:288-296).

`nccl` (= RCCL over xGMI) needs two GPUs: that test is skipped on a one-GPU box and runs wherever two are visible.  The
SAME worker also runs with both ranks on one card over `gloo` (collectives of device tensors staged through the host), so
the code under test - broadcast, grad-ready hooks, bucket packing on the wgrad side stream, all-reduce, scatter - is
exercised on the one-GPU development box too."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
LR = 1e-5
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model(device):
    import sihl_amd
    torch.manual_seed(0)
    backbone = sihl_amd.ResNetBackbone("resnet18", top_level=5)
    neck = sihl_amd.layers.BiFPN(backbone.out_channels, 32, 3, 6, num_layers=1)
    head = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=5, bottom_level=3, top_level=6, num_channels=32)
    return sihl_amd.SihlModel(backbone, neck, [head]).to(device).to(memory_format=torch.channels_last)


def _batch(seed, n_boxes, device):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(len(n_boxes), 3, 128, 128, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    boxes, classes = [], []
    for n in n_boxes:
        xy = torch.rand(n, 2, generator=g) * 70
        wh = 20 + torch.rand(n, 2, generator=g) * 30
        boxes.append(torch.cat([xy, xy + wh], 1).to(device))
        classes.append(torch.randint(0, 5, (n,), generator=g).to(device))
    return images, boxes, classes


BOXES = (2, 0, 3, 1)  # global batch of 4 images: rank r takes images 2r, 2r + 1


def _shard(step, rank, device):
    images, boxes, classes = _batch(step, BOXES, device)
    sl = slice(2 * rank, 2 * rank + 2)
    return images[sl], [{"classes": classes[sl], "boxes": boxes[sl]}]


def _worker(rank, world, port, backend, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    from sihl_amd.train import Trainer

    model = _model(dev)
    if rank == 1:  # replicas start different; the Trainer's broadcast must make them identical
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    tr = Trainer(model, lr=LR, grad_clip_norm=None, wgrad_stream="all", bucket_mb=1.0)
    assert tr.averager.active and len(tr.averager.buckets) > 1 and tr.averager.world == 2
    for step in range(STEPS):
        images, targets = _shard(step, rank, dev)
        tr.step(images, targets)
        if step == 0:
            torch.save([None if p.grad is None else p.grad.detach().cpu().clone() for p in model.parameters()], f"{out}/grads{rank}.pt")
    torch.cuda.synchronize()
    torch.save([p.detach().cpu().clone() for p in model.parameters()], f"{out}/params{rank}.pt")
    torch.save({"wait_ms": tr.averager.wait_ms()}, f"{out}/stats{rank}.pt")
    dist.destroy_process_group()


def _check(tmp_path):
    g0, g1 = torch.load(tmp_path / "grads0.pt"), torch.load(tmp_path / "grads1.pt")
    for a, b in zip(g0, g1):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)  # both ranks hold the same averaged gradient
    p0, p1 = torch.load(tmp_path / "params0.pt"), torch.load(tmp_path / "params1.pt")
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)  # replicas bit-identical after the optimizer steps
    # serial reference on this process's GPU: per-shard gradients (per-replica BatchNorm statistics, per-replica loss
    # normalisers - what DDP gives the reference), averaged
    from sihl_amd.train import Trainer
    dev = torch.device("cuda", 0)
    shard = []
    for r in range(2):
        m = _model(dev)
        tr = Trainer(m, lr=LR, grad_clip_norm=None, wgrad_stream="off")
        images, targets = _shard(0, r, dev)
        tr.optimizer.zero_grad(set_to_none=True)
        loss, _ = tr.forward_loss(images, targets)
        tr._backward(loss)
        torch.cuda.synchronize()
        shard.append([None if p.grad is None else p.grad.detach().cpu() for p in m.parameters()])
    checked = 0
    for a, s0, s1 in zip(g0, *shard):
        if a is None:
            continue
        want = ((0 if s0 is None else s0) + (0 if s1 is None else s1)) / 2
        scale = float(want.abs().max()) + 1e-12
        assert float((a - want).abs().max()) <= 2e-3 * scale + 1e-7, float((a - want).abs().max()) / scale
        checked += 1
    assert checked > 40
    st = torch.load(tmp_path / "stats0.pt")
    assert st["wait_ms"] >= 0.0


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL over two GPUs needs two visible devices")
def test_dp_two_ranks_rccl(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), "nccl", str(tmp_path)), nprocs=2, join=True)
    _check(tmp_path)


@pytest.mark.timeout(600)
def test_dp_two_ranks_one_device_gloo(tmp_path):
    """Both ranks on cuda:0, collectives over gloo: the same worker as the RCCL test, runnable on a one-GPU box."""
    mp.spawn(_worker, args=(2, _free_port(), "gloo", str(tmp_path)), nprocs=2, join=True)
    _check(tmp_path)
