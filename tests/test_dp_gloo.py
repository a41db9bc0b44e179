"""CPU, world_size 2 over gloo: the bucketed gradient averager equals the mean of per-shard gradients, and
two data-parallel Trainer replicas stay bit-identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(7)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(4, 3))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sihl_amd.train import GradientAverager, broadcast_parameters

    torch.set_num_threads(1)
    model = _net()
    if rank == 1:  # replicas start different; the broadcast must make them identical
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_parameters(model)
    avg = GradientAverager(list(model.parameters()), bucket_mb=0.0005)  # several tiny buckets
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 8, 8, generator=g)
    y = torch.randint(0, 3, (8,), generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        nn.functional.cross_entropy(model(xs), ys).backward()
        avg.finish()
        if step == 0:
            torch.save([p.grad.clone() for p in model.parameters()], f"{out}/grads{rank}.pt")
        opt.step()
    torch.save([p.detach().clone() for p in model.parameters()], f"{out}/params{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_allreduce_matches_serial_mean(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "grads0.pt"), torch.load(tmp_path / "grads1.pt")
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)  # both ranks hold the same averaged gradient
    # serial reference: per-shard gradients (per-replica BatchNorm statistics, as Lightning-DDP) averaged
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 8, 8, generator=g)
    y = torch.randint(0, 3, (8,), generator=g)
    shard = []
    for r in range(2):
        m = _net()
        nn.functional.cross_entropy(m(x[r * 4:(r + 1) * 4]), y[r * 4:(r + 1) * 4]).backward()
        shard.append([p.grad for p in m.parameters()])
    for a, s0, s1 in zip(g0, *shard):
        torch.testing.assert_close(a, (s0 + s1) / 2, rtol=1e-6, atol=1e-7)
    p0, p1 = torch.load(tmp_path / "params0.pt"), torch.load(tmp_path / "params1.pt")
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)  # replicas stay in lock-step after two optimizer steps


@pytest.mark.timeout(300)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE) starts two fresh ranks itself and relays rank 0's
    JSON line; --launch-check stops after the rendezvous so the test needs no GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2
    # a rank that fails makes the parent exit non-zero
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--dtype", "nope"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode != 0
