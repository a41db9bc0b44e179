import time, torch, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd
def log(*a):
    print(*a, flush=True)
    open("gpurun_out/bb_probe.log", "a").write(" ".join(str(x) for x in a) + "\n")
dev = "cuda"
torch.backends.cudnn.benchmark = "--find" in sys.argv
bb = sihl_amd.ResNetBackbone("resnet50").to(dev).to(memory_format=torch.channels_last)
x = torch.rand(32, 3, 512, 512, device=dev).contiguous(memory_format=torch.channels_last)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = bb(x)
    torch.cuda.synchronize(); t1 = time.time()
    loss = sum(o.float().mean() for o in outs[1:])
    loss.backward()
    torch.cuda.synchronize(); t2 = time.time()
    log(f"benchmark={torch.backends.cudnn.benchmark} it={it} fwd={t1-t0:.4f}s bwd={t2-t1:.4f}s")
