"""GPU tests of the round-3 additions that are not the row-strip convolution (tests/test_gpu_rowconv.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_depth_l1_loss_matches_torch():
    """hrp_l1_loss (the DepthNet trainer's loss, scripts/train_depthnet.py:231-250) against torch.nn.functional.l1_loss."""
    import hrpe_amd  # noqa: F401
    from hrpe_amd.lib.core.function import depth_l1_loss
    g = torch.Generator().manual_seed(3)
    for n in (1, 7, 64, 300):
        pred = (torch.rand(n, 1, generator=g) * 2000 + 500)
        gt = torch.rand(n, 1, generator=g) * 2 + 0.5
        pr = pred.clone().requires_grad_(True)
        ref = torch.nn.functional.l1_loss(pr / 1000.0, gt)
        (ref * 3.0).backward()
        pd = pred.to(DEV).requires_grad_(True)
        out = depth_l1_loss(pd, gt.to(DEV))
        (out * 3.0).backward()
        assert abs(out.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-7
        assert torch.allclose(pd.grad.cpu(), pr.grad, rtol=1e-6, atol=1e-9)
