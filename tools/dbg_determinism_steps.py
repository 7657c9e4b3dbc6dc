"""development helper: are K whole training steps (reference loss, clip + Adam, dropout) bit-reproducible?  Two runs in one
process with freshly built models; parameters and the loss history are compared bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hrpe_amd.lib.core.function import compute_k_values, full_loss
from hrpe_amd.lib.utils.geometries import rotmat_to_rot6d
from hrpe_amd.optim import FusedClipAdam
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K_STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
PDROP = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
d = {k: torch.tensor(v).to(dev) for k, v in bench.synthetic_batch(B, 808).items()}
K = d["K"]; kv = compute_k_values(K[:, 0, 0], K[:, 1, 1], d["bbox"])
runs = []
for rep in range(2):
    torch.manual_seed(808)
    m = bench.build_model(PDROP).to(dev).set_compute_dtype(torch.bfloat16).train()
    rot6 = rotmat_to_rot6d(d["R"])
    with torch.no_grad():
        kp3d, kp2d = m.robot.get_keypoints_and_projection(d["q"], rot6, d["t"], K, root=0)
        gt = dict(pose=d["q"], root_rot=m.robot.get_rotation_at_specific_root(d["q"], rot6, d["t"], root=3),
                  root_trans=kp3d[:, 3].clone(), root_uv=kp2d[:, 3].clone(), kp3d=kp3d, kp2d=kp2d, mask=torch.ones(B, 7, device=dev))
    params = [p for p in m.parameters() if p.requires_grad]
    losses, grads = [], []
    opt = None
    for step in range(K_STEPS):
        loss, _ = full_loss(m(d["x_reg"], d["x_root"], kv, K), gt, K)
        loss.backward()
        if opt is None:
            opt = FusedClipAdam(params, lr=1e-4, max_norm=5.0)
        grads.append(m.flat_grads()[0].clone())
        opt.step()
        torch.cuda.synchronize()
        losses.append(float(loss))
    runs.append((losses, grads, torch.cat([p.detach().reshape(-1) for p in params]).clone()))
print("losses run 0:", runs[0][0])
print("losses run 1:", runs[1][0])
for s in range(K_STEPS):
    a, b = runs[0][1][s], runs[1][1][s]
    print(f"step {s}: gradients identical {torch.equal(a, b)} (rel {float((a - b).norm() / a.norm()):.2e})")
print("parameters identical after the steps:", torch.equal(runs[0][2], runs[1][2]))
