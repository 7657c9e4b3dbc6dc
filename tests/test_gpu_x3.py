"""GPU tests of the HRP_F32X3 convolution mode (fp32 tensors, every product as three bf16 MFMAs on split operands; include/hrp.h,
csrc/conv_tile.h): the precision between bf16 and fp32 that VERDICT r4 item 4 asks about ("the cheapest mode that meets 0.5 px
on every key-point").  Kernels against torch in float64; the whole network against the reference's fixtures.

Expected error: a product loses the lo x lo term, 2^-16 relative to |x||w|; a sum over K products of random sign averages that to
~2^-16 / sqrt(K) of the sum's scale - measured 1e-6 .. 4e-6 of the output's max, gated at 2e-5 (the bf16 kernels: 4e-2, the fp32
kernels: 2e-4 gate / 1e-6 measured)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # cin, cout, k, stride, H, W, bias
    (64, 32, 3, 1, 16, 16, False),
    (32, 64, 3, 2, 32, 32, False),
    (64, 256, 1, 1, 16, 16, False),
    (3, 64, 3, 2, 64, 64, False),       # the stem: one 8-channel chunk (the upper K half of every MFMA is zero)
    (24, 40, 3, 1, 12, 20, True),       # three chunks (an odd tail), ragged tile, bias
    (256, 128, 1, 1, 8, 8, False),
    (128, 128, 3, 1, 16, 16, False),
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{s[0]}x{s[1]}k{s[2]}s{s[3]}@{s[4]}" for s in SHAPES])
def test_x3_conv_forward_and_data_gradient(shape):
    from hrpe_amd.lib.models.backbones.HRnet import Conv2d
    cin, cout, k, stride, H, W, bias = shape
    g = torch.Generator().manual_seed(cin * 131 + cout)
    conv = Conv2d(cin, cout, k, stride=stride, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cin * k * k) ** 0.5)
    x = torch.randn(3, cin, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, conv.weight.detach().double(), conv.bias.detach().double() if bias else None, stride=stride, padding=k // 2)
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy.double()).sum().backward()
    conv = conv.to(DEV).set_compute_dtype("fp32x3")
    with torch.no_grad():
        y = conv.eval()(x.to(DEV))
    rel = lambda a, b: ((a.double().cpu() - b.detach()).abs().max() / b.detach().abs().max()).item()     # noqa: E731
    e_y = rel(y, yr)
    xd = x.to(DEV).requires_grad_(True)
    y2 = conv.train()(xd)
    (y2 * gy.to(DEV)).sum().backward()
    e_dx = rel(xd.grad, xr.grad)
    wd = conv.weight.double().detach().cpu().requires_grad_(True)
    (F.conv2d(x.double(), wd, None, stride=stride, padding=k // 2) * gy.double()).sum().backward()
    e_dw = rel(conv.weight.grad, wd.grad)
    print(f"\n{shape}: y {e_y:.2e}  dx {e_dx:.2e}  dw {e_dw:.2e}")
    assert e_y < 2e-5 and e_dx < 2e-5 and e_dw < 2e-5, (e_y, e_dx, e_dw)


def test_x3_mode_is_between_bf16_and_fp32_on_one_layer():
    """the same layer in the three modes: bf16 ~1e-3, fp32x3 ~1e-6, fp32 ~1e-7 of the output's scale"""
    from hrpe_amd.lib.models.backbones.HRnet import Conv2d
    g = torch.Generator().manual_seed(7)
    conv = Conv2d(128, 128, 3, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 34.0)
    x = torch.randn(2, 128, 16, 16, generator=g)
    yr = F.conv2d(x.double(), conv.weight.double(), padding=1)
    conv = conv.to(DEV).eval()
    errs = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32x3", "fp32x3"), ("fp32", torch.float32)):
        conv.set_compute_dtype(dt)
        with torch.no_grad():
            y = conv(x.to(DEV))
        errs[name] = ((y.double().cpu() - yr).abs().max() / yr.abs().max()).item()
    print("\n", errs)
    assert errs["fp32"] <= errs["fp32x3"] * 4 and errs["fp32x3"] < errs["bf16"] / 50, errs


def test_x3_full_network_meets_the_pixel_bar_on_every_keypoint():
    """The reference's eval fixture through the full network in fp32x3: every key-point within 0.1 px of the reference run in
    float64 (bf16: key-point 0 at 4.5 px; fp32: 0.0011 px) - the north star's 0.5 px bar with room, at bf16-class matrix rates."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.keypoint_px_error(torch.device(DEV), [("fp32", torch.float32), ("fp32x3", "fp32x3")])
    print("\n", {k: v for k, v in r.items() if "vs_fp64" in k})
    assert r["fp32x3_vs_fp64"] < 0.1, r
    assert r["fp32_vs_fp64"] < 0.01, r


def test_x3_training_step_matches_the_reference_at_fp32_tolerances():
    """The reference's training step at B = 8 (golden_full_train_b8.npz, written by the imported reference) in fp32x3, held to the
    bounds the fp32 path is held to (tests/test_gpu_parity.py::test_full_train_step_golden_b8): forward 8-tuple 5e-4, loss terms
    1e-3, loss 5e-4 - and the sampled gradients of every parameter tensor at 4e-2 in l2 (fp32: 1.5e-2; the bf16 path is gated at 0.3
    on uvd and at a cosine of 0.65 on the trunk's gradients)."""
    from hrpe_amd.lib.core.function import full_loss
    from test_gpu_model import NAMES8, build_full, load, summary_check
    from test_gpu_parity import _sampled_grad_errors, _train_step_inputs
    g = load("golden_full_train_b8.npz")
    m = build_full().set_compute_dtype("fp32x3").train()
    x_reg, x_root, kv, K, gt = _train_step_inputs(g, m, 8)
    pred = m(x_reg, x_root, kv, K)
    for n, p in zip(NAMES8, pred):
        ref = g["fwd:" + n]
        err = np.abs(p.detach().cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-12)
        assert err < 5e-4, f"train fwd {n}: rel err {err}"
    loss, terms = full_loss(pred, gt, K)
    for k, v in terms.items():
        np.testing.assert_allclose(v.item(), g["term:" + k], rtol=1e-3, atol=1e-8, err_msg=k)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=5e-4)
    loss.backward()
    params = dict(m.named_parameters())
    errs = _sampled_grad_errors(g, params)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])
    print("\nB=8 fp32x3 gradient l2 err, worst first:", {k: f"{v:.2e}" for k, v in worst[:6]}, "median", float(np.median(list(errs.values()))))
    # fp32 is gated at 1.5e-2 (measured 5e-3 .. 9e-3, 1.0e-2 .. 1.6e-2 on the two deepest tensors).  fp32x3 measured: median 2.7e-4,
    # six tensors between 1.6e-2 and 2.6e-2 (the stems' first convolutions - the end of a 330-layer backward chain -, one BatchNorm
    # weight of stage 2, three early trunk convolutions): gate 4e-2 on every tensor (the bf16 trunk: cosine >= 0.65)
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":val"):
            name = key.split(":")[1]
            tol = 4e-2
            summary_check(params[name].grad, g, f"grad:{name}:", tol, what="full B=8 fp32x3 ")
