import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import oracle, sihl_amd
name, size, train = "resnet50", int(sys.argv[1]) if len(sys.argv) > 1 else 128, False
torch.manual_seed(0)
ref = oracle.ResNetBackbone(name); hip = sihl_amd.ResNetBackbone(name, native=True)
hip.load_state_dict(ref.state_dict()); hip = hip.cuda(); ref.eval(); hip.eval()
g = torch.Generator().manual_seed(1)
x = torch.rand(2, 3, size, size, generator=g)
cots = None
def run(m, dev):
    global cots
    xi = x.to(dev).requires_grad_(True)
    outs = m(xi)
    if cots is None: cots = [torch.randn(o.shape, generator=g) for o in outs[1:]]
    loss = sum((o * c.to(dev)).sum() for o, c in zip(outs[1:], cots))
    params = [p for _, p in m.named_parameters()]
    return [o.detach().cpu() for o in outs], [t.detach().cpu() for t in torch.autograd.grad(loss, [xi] + params)]
ro, rg = run(ref, "cpu"); ho, hg = run(hip, "cuda")
for l, (a, b) in enumerate(zip(ho, ro)):
    print("level", l, float((a - b).abs().max() / b.abs().max().clamp(min=1e-9)))
names = ["input"] + [n for n, _ in ref.named_parameters()]
rows = []
for n, a, b in zip(names, hg, rg):
    e = (a - b).abs(); rows.append((float(e.pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12)), n, tuple(b.shape)))
for r in sorted(rows, reverse=True)[:25]: print("%.3e %s %s" % r)

print("---- three-way: torch-GPU (MIOpen, native=False) vs CPU oracle, and vs HIP native")
tg = sihl_amd.ResNetBackbone(name, native=False); tg.load_state_dict(ref.state_dict()); tg = tg.cuda().eval()
to, tgr = run(tg, "cuda")
def rms(a, b):
    e = (a - b).abs(); return float(e.pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12))
for n, a, b, c in list(zip(names, tgr, rg, hg))[:6] + list(zip(names, tgr, rg, hg))[-6:]:
    print(f"{n:40s} torchGPU-vs-CPU {rms(a, b):.2e}   HIP-vs-CPU {rms(c, b):.2e}   HIP-vs-torchGPU {rms(c, a):.2e}")
