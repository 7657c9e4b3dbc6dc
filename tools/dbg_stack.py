"""development helper: block stack, fused / unfused block-end backward vs a torch fp32 reference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrpe_amd
from hrpe_amd import plan as P
from hrpe_amd.runtime import SingleTensorModule
from hrpe_amd.lib.models.backbones import HRnet as Hn
DEV = "cuda:0"
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4

class Stack(SingleTensorModule):
    def __init__(self):
        super().__init__()
        self.blocks = torch.nn.ModuleList([Hn.BasicBlock(Cc, Cc) for _ in range(NB)])
    def emit(self, pb, x):
        for b in self.blocks:
            x = b.emit(pb, x)
        return x

class RefBlock(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(Cc, Cc, 3, padding=1, bias=False); self.bn1 = torch.nn.BatchNorm2d(Cc)
        self.conv2 = torch.nn.Conv2d(Cc, Cc, 3, padding=1, bias=False); self.bn2 = torch.nn.BatchNorm2d(Cc)
    def forward(self, x):
        return torch.relu(self.bn2(self.conv2(torch.relu(self.bn1(self.conv1(x))))) + x)

W = 2048 // Cc; N, H = 4, W
g = torch.Generator().manual_seed(Cc + 1)
x = torch.randn(N, Cc, H, W, generator=g); gy = torch.randn(N, Cc, H, W, generator=g)
ref = Stack()
with torch.no_grad():
    for n, prm in ref.named_parameters():
        if prm.dim() == 1:
            prm.copy_(torch.rand(prm.shape, generator=g) + 0.5 if n.endswith("weight") else torch.randn(prm.shape, generator=g) * 0.2)
sd = {k: v.clone() for k, v in ref.state_dict().items()}
rm = torch.nn.Module(); rm.blocks = torch.nn.ModuleList([RefBlock() for _ in range(NB)])
rm.load_state_dict(sd); rm.train()
xr = x.clone().requires_grad_(True); yr = xr
for b in rm.blocks: yr = b(yr)
(yr * gy).sum().backward()
R = dict(y=yr.detach(), dx=xr.grad, **{n: p.grad for n, p in rm.named_parameters()})
res = {}
for fused in (True, False):
    P.BLOCK_END_FUSE = fused
    m = Stack(); m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to(DEV).set_compute_dtype(torch.bfloat16).train()
    xd = x.to(DEV).requires_grad_(True); y = m(xd); (y * gy.to(DEV)).sum().backward(); torch.cuda.synchronize()
    res[fused] = dict(y=y.detach().float().cpu(), dx=xd.grad.float().cpu(), **{n: p.grad.float().cpu() for n, p in m.named_parameters()})
    print(fused, {k: v for k, v in next(iter(m._plans.values())).plan.counters.items() if "block" in k or "fused" in k})
e = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-12)).item()
for k in R:
    print(f"{k:28s} fused-vs-ref {e(res[True][k], R[k]):.4f}  unfused-vs-ref {e(res[False][k], R[k]):.4f}  fused-vs-unfused {e(res[True][k], res[False][k]):.4f}")
