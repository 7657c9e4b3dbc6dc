"""development helper: is a bf16 training step of the full network bit-reproducible?  Two models with identical weights, the
same batch, one eager step each: outputs and the flat gradient arena compared bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hrpe_amd.lib.core.function import compute_k_values
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808).items()}
K = d["K"]; kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
res = []
for rep in range(3):
    m = bench.build_model(0.0).to(dev).set_compute_dtype(torch.bfloat16).train()
    pred = m(d["x_reg"], d["x_root"], kv, K)
    sum(p.float().mean() for p in pred).backward()
    torch.cuda.synchronize()
    res.append(([p.detach().float().clone() for p in pred], m.flat_grads()[0].clone(), {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None}))
for ref, rep in ((0, 1), (0, 2), (1, 2)):
    same_out = all(torch.equal(a, b) for a, b in zip(res[ref][0], res[rep][0]))
    print("   outputs max |diff|:", [float((a - b).abs().max()) for a, b in zip(res[ref][0], res[rep][0])])
    g0, g1 = res[ref][1], res[rep][1]
    diff = (g0 - g1).abs()
    print(f"run {rep} vs {ref}: outputs identical {same_out}; gradients identical {torch.equal(g0, g1)}; differing elements {int((diff > 0).sum())} of {g0.numel()}, rel l2 {float(diff.norm() / g0.norm()):.3e}")
# per-parameter spread between runs 1 and 2, in module order (the backward reaches the LAST entries first)
ga, gb = res[1][2], res[2][2]
bad = [(n, float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30))) for n in ga]
print("parameters:", len(bad), " differing:", sum(e > 0 for _, e in bad), " > 1e-6:", sum(e > 1e-6 for _, e in bad))
for n, e in [(n, e) for n, e in bad if e > 1e-6][-25:]:
    print(f"   {e:9.2e}  {n}")
