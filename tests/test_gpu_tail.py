"""GPU: the tail of a train-mode Bottleneck without conv3's raw output in HBM (hrp_conv_desc.tail_mode, csrc/conv_pw.h,
PlanBuilder.bottleneck_tail; reference lib/models/backbones/HRnet.py:88-96): the four launch forms against float64, and whole
blocks / a layer1-like stack against the CPU oracle and against the same plan with the fusion off."""
import ctypes as C
import os
import sys

import pytest
import torch

from test_gpu_kernels import DEV, _load_into, _rand_sd, l2_err, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


@pytest.fixture
def small_pointwise():
    old = os.environ.get("HRP_PW_MIN_PIXELS")
    os.environ["HRP_PW_MIN_PIXELS"] = "64"
    yield
    if old is None:
        os.environ.pop("HRP_PW_MIN_PIXELS", None)
    else:
        os.environ["HRP_PW_MIN_PIXELS"] = old


@pytest.mark.parametrize("cin,cout,npix", [(64, 256, 8192), (32, 128, 4096 + 37), (64, 128, 96), (32, 256, 20000)])
def test_tail_modes_against_float64(cin, cout, npix, small_pointwise):
    """Mode 1 (statistics of the unrounded product), mode 2 (normalise + shortcut + ReLU + bits), mode 3 (sum g, sum g xhat from the
    recomputed product), mode 4 (gradient of the product + shortcut rider, written and accumulated) of one problem, each against
    float64 arithmetic on the same bf16 operands; ragged last tile; repeats of mode 2 / 4 are bit-identical."""
    from hrpe_amd import _native as nv
    import bench_kernels as bk
    g = torch.Generator(device="cpu").manual_seed(cin + cout + npix)
    h = torch.randn(npix, cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
    wp, _ = bk.pack(w, torch.bfloat16)
    xs = torch.randn(npix, cout, generator=g).to(DEV).to(torch.bfloat16)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.3).to(DEV)
    d = nv.ConvDesc()
    d.x, d.w, d.dtype = h.data_ptr(), wp.data_ptr(), nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = 1, 1, npix, cin, cin
    d.Ho, d.Wo, d.Cout, d.y_H, d.y_W, d.y_pitch, d.res_pitch = 1, npix, cout, 1, npix, cout, cout
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 1, 1, cout
    stats = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
    bsums = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
    mask = torch.zeros(npix * cout // 8, dtype=torch.uint8, device=DEV)
    out = torch.zeros(npix, cout, device=DEV, dtype=torch.bfloat16)
    dummy = torch.zeros(npix, cout, device=DEV, dtype=torch.bfloat16)
    d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps, d.tail_mask = gamma.data_ptr(), beta.data_ptr(), float(npix), 1e-5, mask.data_ptr()

    def run(mode, **kw):
        q = nv.ConvDesc()
        C.memmove(C.byref(q), C.byref(d), C.sizeof(nv.ConvDesc))
        q.tail_mode, q.y = mode, dummy.data_ptr()
        for k, v in kw.items():
            setattr(q, k, v)
        assert nv.lib().hrp_conv_pointwise(C.byref(q)) == 1
        nv.call("hrp_conv2d_fwd", C.byref(q), None)
        torch.cuda.synchronize()

    wq = w.view(cout, cin).to(torch.bfloat16).double()
    y = h.double() @ wq.t()
    run(1, stats=stats.data_ptr())
    st = stats.view(8, 2, cout).sum(0)
    assert rel_err(st[0], y.sum(0)) < 1e-5 * (1 + float(y.abs().sum(0).max() / (y.sum(0).abs().max() + 1e-9))) and rel_err(st[1], (y * y).sum(0)) < 1e-5
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    want = torch.relu((y - mean) * inv * gamma.double() + beta.double() + xs.double())
    run(2, y=out.data_ptr(), res=xs.data_ptr(), tail_stats=stats.data_ptr())
    assert rel_err(out, want) < 1e-2, rel_err(out, want)
    bits = mask.view(npix, cout // 8)
    got_pos = torch.stack([(bits >> i) & 1 for i in range(8)], -1).view(npix, cout).bool()
    assert torch.equal(got_pos, out > 0), "bit i of byte j = channel 8 j + i was positive"
    first = out.clone()
    run(2, y=out.data_ptr(), res=xs.data_ptr(), tail_stats=stats.data_ptr())
    assert torch.equal(out, first)
    # mode 5: the shortcut is a second product under its own BatchNorm (the projection of a stack's first block)
    h2 = torch.randn(npix, cin, generator=g).to(DEV).to(torch.bfloat16)
    w2 = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
    wp2, _ = bk.pack(w2, torch.bfloat16)
    gamma2, beta2 = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.3).to(DEV)
    stats2 = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
    run(1, x=h2.data_ptr(), w=wp2.data_ptr(), stats=stats2.data_ptr())
    y2 = h2.double() @ w2.view(cout, cin).to(torch.bfloat16).double().t()
    m2, i2 = y2.mean(0), (y2.var(0, unbiased=False) + 1e-5).rsqrt()
    want5 = torch.relu((y - mean) * inv * gamma.double() + beta.double() + (y2 - m2) * i2 * gamma2.double() + beta2.double())
    out5, mask5 = torch.zeros_like(out), torch.zeros_like(mask)
    run(5, y=out5.data_ptr(), tail_stats=stats.data_ptr(), tail_mask=mask5.data_ptr(), tail_x2=h2.data_ptr(), tail_w2=wp2.data_ptr(),
        tail_stats2=stats2.data_ptr(), tail_gamma2=gamma2.data_ptr(), tail_beta2=beta2.data_ptr())
    assert rel_err(out5, want5) < 1e-2, rel_err(out5, want5)
    pos5 = torch.stack([(mask5.view(npix, cout // 8) >> i) & 1 for i in range(8)], -1).view(npix, cout).bool()
    assert torch.equal(pos5, out5 > 0)
    # backward: the gradient of out
    dout = torch.randn(npix, cout, generator=g).to(DEV).to(torch.bfloat16)
    gm = dout.double() * got_pos
    xhat = (y - mean) * inv
    run(3, stats=bsums.data_ptr(), tail_stats=stats.data_ptr(), tail_g=dout.data_ptr())
    bs = bsums.view(8, 2, cout).sum(0)
    assert l2_err(bs[0], gm.sum(0)) < 1e-4 and l2_err(bs[1], (gm * xhat).sum(0)) < 2e-3, (l2_err(bs[0], gm.sum(0)), l2_err(bs[1], (gm * xhat).sum(0)))
    k0, k1 = gm.sum(0) / npix, (gm * xhat).sum(0) / npix
    want_dy = gamma.double() * inv * (gm - k0 - xhat * k1)
    dy = torch.zeros(npix, cout, device=DEV, dtype=torch.bfloat16)
    side = torch.full((npix, cout), 0.5, device=DEV, dtype=torch.bfloat16)
    run(4, y=dy.data_ptr(), tail_stats=stats.data_ptr(), tail_bsums=bsums.data_ptr(), tail_g=dout.data_ptr(), tail_side=side.data_ptr(), tail_side_acc=1)
    assert rel_err(dy, want_dy) < 1.5e-2, rel_err(dy, want_dy)
    assert rel_err(side, gm + 0.5) < 1e-2
    run(4, y=dy.data_ptr(), tail_stats=stats.data_ptr(), tail_bsums=bsums.data_ptr(), tail_g=dout.data_ptr(), tail_side=side.data_ptr(), tail_side_acc=0)
    assert torch.equal(side.double(), gm.to(torch.bfloat16).double()), "the rider is the masked gradient itself"
    d2 = dy.clone()
    run(4, y=dy.data_ptr(), tail_stats=stats.data_ptr(), tail_bsums=bsums.data_ptr(), tail_g=dout.data_ptr())
    assert torch.equal(dy, d2)


def test_tail_refusals(small_pointwise):
    """What the pointwise kernel declines (the plan then keeps the element-wise path) and what hrp_conv2d_fwd refuses loudly."""
    from hrpe_amd import _native as nv
    t = torch.zeros(4096 * 256, device=DEV, dtype=torch.bfloat16)
    d = nv.ConvDesc()
    d.x = d.w = d.y = t.data_ptr()
    d.dtype, d.N, d.H, d.W, d.Cin, d.x_pitch = nv.HRP_BF16, 1, 1, 4096, 64, 64
    d.Ho, d.Wo, d.Cout, d.y_H, d.y_W, d.y_pitch, d.res_pitch = 1, 4096, 256, 1, 4096, 256, 256
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 1, 1, 256
    d.tail_mode = 1
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0            # mode 1 without statistic slots
    d.stats = t.data_ptr()
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 1
    d.Cout = d.y_pitch = d.res_pitch = d.w_cout_pad = 96               # not a multiple of 64
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0
    with pytest.raises(nv.HrpError):
        nv.call("hrp_conv2d_fwd", C.byref(d), None)
    d.Cout = d.y_pitch = d.res_pitch = d.w_cout_pad = 256
    d.tail_mode = 2                                                    # mode 2 without its operands
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0
    d.tail_mode, d.Cin, d.x_pitch = 1, 128, 128                        # 128 input channels
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0


def _run(kind, fuse, x, sd, gy):
    from hrpe_amd import plan as P
    from hrpe_amd.lib.models.backbones import HRnet as H
    import torch.nn as nn
    P.BNECK_TAIL_FUSE = fuse
    try:
        if kind == "narrow":
            m = H.Bottleneck(128, 32)
        elif kind == "wide":
            m = H.Bottleneck(256, 64)
        elif kind == "proj":          # layer1's first block (HRnet.py:291, 448-465): 64 -> 256 with a 1x1 projection shortcut
            m = H.Bottleneck(64, 64, downsample=H._Downsample(64, 256, 1))
        elif kind == "proj_narrow":   # incre_modules[0] of the cls head (HRnet.py:343-362): 32 -> 128
            m = H.Bottleneck(32, 32, downsample=H._Downsample(32, 128, 1))
        else:
            class Stack(H.SingleTensorModule):          # layer1 of the trunk (HRnet.py:291): a projection block and three identity blocks
                def __init__(self):
                    super().__init__()
                    self.layer1 = H._block_stack(H.Bottleneck, 64, 64, 4)

                def emit(self, pb, t):
                    return H._emit_seq(pb, self.layer1, t)
            m = Stack()
        _load_into(m, sd(m))
        m = m.to(DEV).set_compute_dtype(torch.bfloat16).train()
        xd = x.to(DEV).requires_grad_(True)
        y = m(xd)
        (y.float() * gy.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        plan = next(iter(m._plans.values())).plan
        return m, y.detach().float().cpu(), xd.grad.cpu(), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}, plan.counters.get("bottleneck_tails", 0)
    finally:
        P.BNECK_TAIL_FUSE = True


@pytest.mark.parametrize("kind", ["narrow", "wide", "proj", "proj_narrow"])
def test_bottleneck_with_fused_tail_matches_oracle(kind):
    """One Bottleneck with an identity shortcut at [32, C, 64, 64] (131 072 pixels: the pointwise kernel's threshold) in train mode,
    bf16: forward, input gradient, every parameter gradient and the running statistics against the CPU oracle in fp32 (bf16
    tolerances of test_gpu_kernels: 4e-2 forward, L2 on gradients), with the fusion on (counter checked) and - same bounds - off;
    the two plans agree with each other more closely than either with fp32."""
    from oracle import hrnet as O
    cin, cout = {"narrow": (128, 128), "wide": (256, 256), "proj": (64, 256), "proj_narrow": (32, 128)}[kind]
    x = torch.randn(32, cin, 64, 64, generator=torch.Generator().manual_seed(5))
    cache = {}

    def sd(m):
        if "sd" not in cache:
            cache["sd"] = _rand_sd(m, 3)
        return cache["sd"]
    gy = torch.randn(32, cout, 64, 64, generator=torch.Generator().manual_seed(6))
    mf, yf, dxf, gf, nf = _run(kind, True, x, sd, gy)
    mu, yu, dxu, gu, nu = _run(kind, False, x, sd, gy)
    assert nf == 1 and nu == 0
    osd = {"b." + k: v.clone() for k, v in cache["sd"].items()}
    for k, v in osd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O._bottleneck(O._Ctx(osd, "", True), "b", xr)
    (yr * gy).sum().backward()
    errs = {}
    for name, (y, dx, gr, m) in {"fused": (yf, dxf, gf, mf), "plain": (yu, dxu, gu, mu)}.items():
        assert rel_err(y, yr) < 4e-2, (name, rel_err(y, yr))
        errs[name] = {"dx": l2_err(dx, xr.grad)}
        for k, v in osd.items():
            if v.grad is not None:
                errs[name][k] = l2_err(gr[k[2:]], v.grad)
        bufs = dict(m.named_buffers())
        for k, v in osd.items():
            if "running" in k:
                assert rel_err(bufs[k[2:]], v) < 4e-2, (name, k)
    # bf16 gradients in L2 against fp32 (test_gpu_kernels: 3 x the forward tolerance), and the fused plan is not further from fp32
    # than the plain one (it rounds in fewer places)
    for k, e in errs["fused"].items():
        assert e < 0.12, ("fused", k, e)
        assert errs["plain"][k] < 0.12, ("plain", k, errs["plain"][k])
        assert e <= 1.25 * errs["plain"][k] + 5e-3, (k, e, errs["plain"][k])
    assert l2_err(yf, yu) < 1e-2 and l2_err(dxf, dxu) < 3e-2, (l2_err(yf, yu), l2_err(dxf, dxu))
    # the fused tail normalises the fp32 product, the plain path its bf16 rounding: closer to fp32, not further
    assert l2_err(yf, yr) <= 1.1 * l2_err(yu, yr) + 1e-4, (l2_err(yf, yr), l2_err(yu, yr))


def test_layer1_stack_with_fused_tails():
    """layer1 of the trunk (a projection block + three identity blocks, HRnet.py:291) at B = 32: all four tails fuse (the first in
    the projection form).  Output, input gradient and every parameter gradient of both plans - fusion on / off - against the CPU oracle
    in fp32: the fused plan is held to the plain plan's distance from fp32 (two bf16 plans that round at different places differ from
    EACH OTHER by up to 9 % in L2 on the input gradient after four blocks; what matters is that neither is further from fp32)."""
    from oracle import hrnet as O
    x = torch.randn(32, 64, 64, 64, generator=torch.Generator().manual_seed(7))
    cache = {}

    def sd(m):
        if "sd" not in cache:
            cache["sd"] = _rand_sd(m, 4)
        return cache["sd"]
    gy = torch.randn(32, 256, 64, 64, generator=torch.Generator().manual_seed(8))
    _, yf, dxf, gf, nf = _run("stack", True, x, sd, gy)
    _, yu, dxu, gu, nu = _run("stack", False, x, sd, gy)
    assert nf == 4 and nu == 0
    osd = {k: v.clone() for k, v in cache["sd"].items()}
    for k, v in osd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ctx = O._Ctx(osd, "", True)
    t = xr
    for i in range(4):
        t = O._bottleneck(ctx, f"layer1.{i}", t)
    (t * gy).sum().backward()
    assert rel_err(yf, t) < 6e-2 and l2_err(yf, t) <= 1.1 * l2_err(yu, t) + 1e-4, (l2_err(yf, t), l2_err(yu, t))
    ef, eu = l2_err(dxf, xr.grad), l2_err(dxu, xr.grad)
    assert ef < 0.15 and ef <= 1.25 * eu + 5e-3, (ef, eu)
    for k, v in osd.items():
        if v.grad is not None and float(v.grad.abs().max()) > 0:
            ef, eu = l2_err(gf[k], v.grad), l2_err(gu[k], v.grad)
            assert ef < 0.2 and ef <= 1.3 * eu + 1e-2, (k, ef, eu)
