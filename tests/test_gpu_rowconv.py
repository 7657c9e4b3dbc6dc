"""GPU parity tests of the row-strip convolution kernel (csrc/conv_row.h) through the C ABI: the lean 3x3 kernel of the
HRNet branch BasicBlocks (reference HRnet.py:28-57) with its fused BatchNorm prologues / epilogues, against plain torch
fp32 on the CPU.

Tolerance: bf16 operands, fp32 accumulation - 2e-2 of the tensor's scale on outputs (one bf16 ulp is 2^-8), 2e-3 on fp32
statistics (which are compared with sums over the kernel's OWN stored output, so only the summation order differs)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-5
SLOTS = 8
TAPS3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]


def nvmod():
    import hrpe_amd  # noqa: F401
    from hrpe_amd import _native as nv
    return nv


def rup(a, b):
    return (a + b - 1) // b * b


def pack(nv, w):
    """fp32 [Cout, Cin, 3, 3] -> (forward packing, transposed packing) in bf16 (hrp_pack_weights)."""
    cout, cin = w.shape[0], w.shape[1]
    nf = -(-cin // 16) * 9 * rup(cout, 32) * 16
    nb = -(-cout // 16) * 9 * rup(cin, 32) * 16
    dst = torch.zeros(nf, dtype=torch.bfloat16, device=DEV)
    dst_t = torch.zeros(nb, dtype=torch.bfloat16, device=DEV)
    tab = (nv.PackEntry * 1)()
    wd = w.to(DEV).contiguous()
    tab[0].src, tab[0].dst, tab[0].dst_t = wd.data_ptr(), dst.data_ptr(), dst_t.data_ptr()
    tab[0].Cout, tab[0].Cin, tab[0].ntaps = cout, cin, 9
    tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
    nv.call("hrp_pack_weights", tdev.data_ptr(), 1, nv.HRP_BF16, max(nf, nb), None)
    torch.cuda.synchronize()
    return dst, dst_t


def bf(x):
    return x.to(torch.bfloat16).float()


def nhwc(x_nchw_bf):
    """fp32 NCHW values (already bf16-representable) -> dense NHWC bf16 device tensor."""
    return x_nchw_bf.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc(t, N, H, W, Cc):
    return t.float().cpu().view(N, H, W, Cc).permute(0, 3, 1, 2)


def desc(nv, x, wp, y, N, H, W, Cc, transposed=False):
    d = nv.ConvDesc()
    d.x, d.w, d.y, d.dtype = x.data_ptr(), wp.data_ptr(), y.data_ptr(), nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, H, W, Cc, Cc
    d.Ho, d.Wo, d.Cout = H, W, Cc
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = H, W, Cc, Cc
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 9, 9, Cc
    for i, (a, b) in enumerate(TAPS3):
        if transposed:      # data gradient: mirrored taps on the transposed packing
            d.dy[i], d.dx[i], d.wtap[i] = -a, -b, i
        else:
            d.dy[i], d.dx[i], d.wtap[i] = a, b, i
    return d


def slots_of(total, g):
    """Split per-channel totals [2C] over the 8 statistic slots at random (the kernel must add the slots)."""
    r = torch.rand(SLOTS, total.numel(), generator=g)
    r = r / r.sum(0, keepdim=True)
    return (r * total[None]).double().contiguous().to(DEV)       # statistic slots are fp64 (include/hrp.h)


def bn_consts(x, gamma, beta):
    """fp32 per-channel train-mode BatchNorm constants of NCHW x: mean, invstd, sc, sh."""
    cnt = x.numel() / x.shape[1]
    s1, s2 = x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))
    m = s1 / cnt
    var = (s2 / cnt - m * m).clamp_min(0)
    inv = torch.rsqrt(var + EPS)
    sc = gamma * inv
    return m, inv, sc, beta - m * sc, torch.cat([s1, s2]), cnt


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


SHAPES = [(32, 3, 64), (64, 2, 32), (32, 1, 16), (64, 5, 8), (128, 3, 16), (256, 3, 8), (128, 1, 8), (256, 2, 16),
          (32, 32, 64)]     # (C, N, H); W = 2048 / C.  The last one: 256 strips -> persistent workgroups, two strips each


@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_plain_and_eval_epilogue(shape):
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 131 + N)
    x = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    wp, _ = pack(nv, w)
    xd = nhwc(x)
    y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    d = desc(nv, xd, wp, y, N, H, W, Cc)
    d.stats = st.data_ptr()
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), w.double(), padding=1).float()
    got = from_nhwc(y, N, H, W, Cc)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    s = st.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    own = torch.cat([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert rel(s, own) < 2e-3, rel(s, own)
    # eval-mode epilogue: folded BatchNorm affine + residual + ReLU (HRnet.py:52-56 in an inference plan)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    r = bf(torch.randn(N, Cc, H, W, generator=g))
    rd = nhwc(r)
    scd, shd = sc.to(DEV), sh.to(DEV)
    d.stats, d.scale, d.shift, d.res, d.relu = None, scd.data_ptr(), shd.data_ptr(), rd.data_ptr(), 1
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    ref2 = torch.relu(ref * sc[None, :, None, None] + sh[None, :, None, None] + r)
    got2 = from_nhwc(y, N, H, W, Cc)
    assert rel(got2, ref2) < 2e-2, rel(got2, ref2)


@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_bn_relu_prologue(shape):
    """conv(relu(bn(x))) with the activation as a side output (pro_mode 1) == hrp_ew_fwd followed by the conv."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 17 + N)
    x = bf(torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3)
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(x, gamma, beta)
    a = bf(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]))
    ref = F.conv2d(a.double(), w.double(), padding=1).float()
    wp, _ = pack(nv, w)
    xd = nhwc(x)
    y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    st_in, st = slots_of(tot, g), torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    d = desc(nv, xd, wp, y, N, H, W, Cc)
    d.stats = st.data_ptr()
    d.pro_mode, d.pro_stats, d.pro_gamma, d.pro_beta = 1, st_in.data_ptr(), gd.data_ptr(), bd.data_ptr()
    d.pro_count, d.pro_eps, d.pro_side = float(cnt), EPS, side.data_ptr()
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    got, gside = from_nhwc(y, N, H, W, Cc), from_nhwc(side, N, H, W, Cc)
    assert rel(gside, a) < 1e-2, rel(gside, a)           # (one bf16 ulp where the fp32 affine rounds the other way)
    assert ((gside > 0) != (a > 0)).float().mean().item() < 1e-4
    assert rel(got, ref) < 2e-2, rel(got, ref)
    s = st.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    own = torch.cat([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert rel(s, own) < 2e-3, rel(s, own)


@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_block_end_prologue(shape):
    """pro_mode 3 (round 6): conv(relu(bn(y2) + res)) - the previous BasicBlock's block-end activation (HRnet.py:52-56) applied while
    the rows are staged, the activation as side output and its ReLU bits as a second one == hrp_ew_fwd followed by the conv."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 19 + N)
    y2 = bf(torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3)
    res = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(y2, gamma, beta)
    a = bf(torch.relu(y2 * sc[None, :, None, None] + sh[None, :, None, None] + res))
    ref = F.conv2d(a.double(), w.double(), padding=1).float()
    wp, _ = pack(nv, w)
    y2d, resd = nhwc(y2), nhwc(res)
    y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    mask = torch.full((N * H * W * Cc // 8,), 0xAA, dtype=torch.uint8, device=DEV)
    st_in, st = slots_of(tot, g), torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    d = desc(nv, y2d, wp, y, N, H, W, Cc)
    d.stats = st.data_ptr()
    d.pro_mode, d.pro_stats, d.pro_gamma, d.pro_beta = 3, st_in.data_ptr(), gd.data_ptr(), bd.data_ptr()
    d.pro_count, d.pro_eps, d.pro_x2 = float(cnt), EPS, resd.data_ptr()
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == 0          # the side output and the mask are required
    d.pro_side, d.pro_mask = side.data_ptr(), mask.data_ptr()
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    got, gside = from_nhwc(y, N, H, W, Cc), from_nhwc(side, N, H, W, Cc)
    assert rel(gside, a) < 1e-2, rel(gside, a)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    bits = mask.view(N, H, W, Cc // 8).cpu()
    pos = torch.stack([(bits >> i) & 1 for i in range(8)], -1).view(N, H, W, Cc).permute(0, 3, 1, 2).bool()
    assert torch.equal(pos, gside > 0), "bit i of byte j = channel 8 j + i of the stored activation is positive"
    s = st.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    own = torch.cat([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert rel(s, own) < 2e-3, rel(s, own)
    y_first = y.clone()
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    assert torch.equal(y, y_first)


@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_bn_backward_reduce_epilogue(shape):
    """Data gradient whose epilogue accumulates sum g, sum g * xhat of the stored gradient (mask recomputed from the
    BatchNorm input) == the conv followed by hrp_ew_bwd_reduce."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 29 + N)
    dy = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    x1 = bf(torch.randn(N, Cc, H, W, generator=g) * 2.0 - 0.4)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(x1, gamma, beta)
    _, wpt = pack(nv, w)
    dyd, x1d = nhwc(dy), nhwc(x1)
    y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    st_in, bs = slots_of(tot, g), torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    d = desc(nv, dyd, wpt, y, N, H, W, Cc, transposed=True)
    d.stats, d.bnb_x, d.bnb_x_pitch = bs.data_ptr(), x1d.data_ptr(), Cc
    d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = st_in.data_ptr(), gd.data_ptr(), bd.data_ptr(), float(cnt), EPS
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1).float()      # the data gradient of conv2d(., w, padding=1)
    got = from_nhwc(y, N, H, W, Cc)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    act = x1 * sc[None, :, None, None] + sh[None, :, None, None]
    sure = act.abs() > 1e-4                                   # (elements whose mask could fall either way are left out of both sides)
    gm = got * (act > 0) * sure
    xh = (x1 - m[None, :, None, None]) * inv[None, :, None, None]
    want = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    s = bs.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    unsure = (got * (~sure)).abs().sum((0, 2, 3))
    err = (s - want).abs()
    bound = 2e-3 * want.abs().max() + torch.cat([unsure, unsure * xh.abs().max()])
    assert (err <= bound).all(), (err / bound).max()


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_bn_backward_apply_prologue(shape, masked):
    """Data gradient of conv1 whose staged operand is the BatchNorm + ReLU backward of (gradient of the activation,
    BatchNorm input) (pro_mode 2), accumulated onto an existing gradient (res == y), with the BatchNorm input gradient
    as a side output == hrp_ew_bwd_apply followed by the conv."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 31 + N)
    ga = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    x1 = bf(torch.randn(N, Cc, H, W, generator=g) * 2.0 + 0.2)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(x1, gamma, beta)
    act = x1 * sc[None, :, None, None] + sh[None, :, None, None]
    ga = ga * (act.abs() > 1e-4)                              # no gradient where the mask could fall either way
    gm = ga * (act > 0)
    xh = (x1 - m[None, :, None, None]) * inv[None, :, None, None]
    bt = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    k0, k1 = bt[:Cc] / cnt, bt[Cc:] / cnt
    dx1 = sc[None, :, None, None] * (gm - k0[None, :, None, None] - xh * k1[None, :, None, None])
    dx1b = bf(dx1)
    prev = bf(torch.randn(N, Cc, H, W, generator=g))
    onr = torch.rand(N, Cc, H, W, generator=g) > 0.5             # masked: residual = another tensor under a bit mask (res_mask)
    ref = F.conv_transpose2d(dx1b.double(), w.double(), padding=1).float() + (prev * onr if masked else prev)
    _, wpt = pack(nv, w)
    gad, x1d, y = nhwc(ga), nhwc(x1), nhwc(prev)
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    st_in, bs_in = slots_of(tot, g), slots_of(bt, g)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    d = desc(nv, gad, wpt, y, N, H, W, Cc, transposed=True)
    d.res = y.data_ptr()
    if masked:
        resd, mkr = nhwc(prev), mask_bits(onr)
        y.fill_(3.0)
        d.res, d.res_mask = resd.data_ptr(), mkr.data_ptr()
    d.pro_mode, d.pro_x2, d.pro_stats, d.pro_bsums = 2, x1d.data_ptr(), st_in.data_ptr(), bs_in.data_ptr()
    d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, d.pro_side = gd.data_ptr(), bd.data_ptr(), float(cnt), EPS, side.data_ptr()
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    gside, got = from_nhwc(side, N, H, W, Cc), from_nhwc(y, N, H, W, Cc)
    assert rel(gside, dx1) < 1.5e-2, rel(gside, dx1)
    assert rel(got, ref) < 2e-2, rel(got, ref)


def mask_bits(on_nchw):
    """[N, C, H, W] bool -> the ReLU bit mask hrp_ew_fwd writes: one byte per 8 channels of a pixel, bit i = channel 8k + i."""
    N, Cc, H, W = on_nchw.shape
    v = on_nchw.permute(0, 2, 3, 1).reshape(N * H * W, Cc // 8, 8).to(torch.int32)
    byte = (v << torch.arange(8, dtype=torch.int32)[None, None, :]).sum(-1)
    return byte.to(torch.uint8).contiguous().view(-1).to(DEV)


@pytest.mark.parametrize("acc2", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_block_end_apply_prologue(shape, acc2):
    """conv2's data gradient of a fused BasicBlock: the staged operand is the block-end BatchNorm + ReLU backward with the
    ReLU given as hrp_ew_fwd's bit mask (pro_mode 2 + pro_mask); side output = gradient of the BatchNorm input, second side
    output = the masked gradient for the residual (written / accumulated); the epilogue reduces the INTERIOR BatchNorm's
    sums with its mask recomputed == hrp_ew_bwd_apply (two outputs), the conv, hrp_ew_bwd_reduce."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 37 + N + acc2)
    gout = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    y2 = bf(torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3)
    on = torch.rand(N, Cc, H, W, generator=g) > 0.45            # the ReLU decision involves the residual: NOT a function of y2
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(y2, gamma, beta)
    gm = gout * on
    xh = (y2 - m[None, :, None, None]) * inv[None, :, None, None]
    bt = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    k0, k1 = bt[:Cc] / cnt, bt[Cc:] / cnt
    dy2 = sc[None, :, None, None] * (gm - k0[None, :, None, None] - xh * k1[None, :, None, None])
    dy2b = bf(dy2)
    ref = F.conv_transpose2d(dy2b.double(), w.double(), padding=1).float()
    # interior BatchNorm (bn1 over y1): reduce of the produced gradient
    y1 = bf(torch.randn(N, Cc, H, W, generator=g) * 2.0 - 0.4)
    gamma1, beta1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m1, inv1, sc1, sh1, tot1, _ = bn_consts(y1, gamma1, beta1)
    prev2 = bf(torch.randn(N, Cc, H, W, generator=g))
    _, wpt = pack(nv, w)
    god, y2d, y1d = nhwc(gout), nhwc(y2), nhwc(y1)
    y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    side2 = nhwc(prev2) if acc2 else torch.full((N * H * W * Cc,), 5.0, dtype=torch.bfloat16, device=DEV)
    mk = mask_bits(on)
    st2, bs2, st1, bs1 = slots_of(tot, g), slots_of(bt, g), slots_of(tot1, g), torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    gd, bd, gd1, bd1 = gamma.to(DEV), beta.to(DEV), gamma1.to(DEV), beta1.to(DEV)
    d = desc(nv, god, wpt, y, N, H, W, Cc, transposed=True)
    d.pro_mode, d.pro_x2, d.pro_stats, d.pro_bsums = 2, y2d.data_ptr(), st2.data_ptr(), bs2.data_ptr()
    d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps = gd.data_ptr(), bd.data_ptr(), float(cnt), EPS
    d.pro_mask, d.pro_side, d.pro_side2, d.pro_side2_acc = mk.data_ptr(), side.data_ptr(), side2.data_ptr(), acc2
    d.stats, d.bnb_x, d.bnb_x_pitch = bs1.data_ptr(), y1d.data_ptr(), Cc
    d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = st1.data_ptr(), gd1.data_ptr(), bd1.data_ptr(), float(cnt), EPS
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    gside, gside2, got = from_nhwc(side, N, H, W, Cc), from_nhwc(side2, N, H, W, Cc), from_nhwc(y, N, H, W, Cc)
    assert rel(gside, dy2) < 1.5e-2, rel(gside, dy2)
    want2 = bf(gm + prev2) if acc2 else gm
    assert torch.equal(gside2, want2)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    act1 = y1 * sc1[None, :, None, None] + sh1[None, :, None, None]
    sure = act1.abs() > 1e-4
    g1m = got * (act1 > 0) * sure
    xh1 = (y1 - m1[None, :, None, None]) * inv1[None, :, None, None]
    want = torch.cat([g1m.sum((0, 2, 3)), (g1m * xh1).sum((0, 2, 3))])
    sres = bs1.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    unsure = (got * (~sure)).abs().sum((0, 2, 3))
    err = (sres - want).abs()
    bound = 2e-3 * want.abs().max() + torch.cat([unsure, unsure * xh1.abs().max()])
    assert (err <= bound).all(), (err / bound).max()


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_rowconv_block_end_reduce_epilogue(shape, masked):
    """conv1's data gradient of the NEXT block completing the gradient of a block output: interior BatchNorm + ReLU backward
    in the prologue (mask recomputed), accumulation onto the residual's gradient (res == y), and in the epilogue the
    block-end BatchNorm's sums of the COMPLETED gradient masked by hrp_ew_fwd's bit mask (bnb_mask) == hrp_ew_bwd_apply,
    the conv, hrp_ew_bwd_reduce of the previous block's activation."""
    nv = nvmod()
    Cc, N, H = shape
    W = 2048 // Cc
    g = torch.Generator().manual_seed(Cc * 41 + N)
    ga = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    x1 = bf(torch.randn(N, Cc, H, W, generator=g) * 2.0 + 0.2)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(x1, gamma, beta)
    act = x1 * sc[None, :, None, None] + sh[None, :, None, None]
    ga = ga * (act.abs() > 1e-4)
    gm = ga * (act > 0)
    xh = (x1 - m[None, :, None, None]) * inv[None, :, None, None]
    bt = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    k0, k1 = bt[:Cc] / cnt, bt[Cc:] / cnt
    dx1b = bf(sc[None, :, None, None] * (gm - k0[None, :, None, None] - xh * k1[None, :, None, None]))
    prev = bf(torch.randn(N, Cc, H, W, generator=g))
    # masked: the residual is ANOTHER tensor (the block output's gradient) under a ReLU bit mask (res_mask), y is written once
    onr = torch.rand(N, Cc, H, W, generator=g) > 0.5
    ref = F.conv_transpose2d(dx1b.double(), w.double(), padding=1).float() + (prev * onr if masked else prev)
    # the previous block's end: BatchNorm over yp, ReLU decisions as bits
    yp = bf(torch.randn(N, Cc, H, W, generator=g) * 1.7 - 0.2)
    onp = torch.rand(N, Cc, H, W, generator=g) > 0.4
    gammap, betap = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    mp, invp, _, _, totp, _ = bn_consts(yp, gammap, betap)
    _, wpt = pack(nv, w)
    gad, x1d, y, ypd = nhwc(ga), nhwc(x1), nhwc(prev), nhwc(yp)
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    mk = mask_bits(onp)
    st_in, bs_in, stp, bsp = slots_of(tot, g), slots_of(bt, g), slots_of(totp, g), torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
    gd, bd, gdp, bdp = gamma.to(DEV), beta.to(DEV), gammap.to(DEV), betap.to(DEV)
    d = desc(nv, gad, wpt, y, N, H, W, Cc, transposed=True)
    d.res = y.data_ptr()
    if masked:
        resd, mkr = nhwc(prev), mask_bits(onr)
        y.fill_(3.0)
        d.res, d.res_mask = resd.data_ptr(), mkr.data_ptr()
    d.pro_mode, d.pro_x2, d.pro_stats, d.pro_bsums = 2, x1d.data_ptr(), st_in.data_ptr(), bs_in.data_ptr()
    d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, d.pro_side = gd.data_ptr(), bd.data_ptr(), float(cnt), EPS, side.data_ptr()
    d.stats, d.bnb_x, d.bnb_x_pitch, d.bnb_mask, d.bnb_mask_pitch = bsp.data_ptr(), ypd.data_ptr(), Cc, mk.data_ptr(), Cc // 8
    d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = stp.data_ptr(), gdp.data_ptr(), bdp.data_ptr(), float(cnt), EPS
    assert nv.lib().hrp_conv_rowstrip_channels(C.byref(d)) == Cc
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    got = from_nhwc(y, N, H, W, Cc)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    gpm = got * onp
    xhp = (yp - mp[None, :, None, None]) * invp[None, :, None, None]
    want = torch.cat([gpm.sum((0, 2, 3)), (gpm * xhp).sum((0, 2, 3))])
    sres = bsp.view(SLOTS, 2 * Cc).sum(0).float().cpu()
    err = (sres - want).abs()
    assert (err <= 2e-3 * want.abs().max()).all(), (err / want.abs().max()).max()


def test_rowconv_in_a_batched_launch_equals_single_launches():
    """Row-strip problems (32 and 64 channels) and a general-tile problem (128 channels) in ONE HRP_BATCH_CONV launch:
    every output bit-identical to the problem's single launch."""
    nv = nvmod()
    g = torch.Generator().manual_seed(5)
    keep, descs, outs = [], [], []
    for Cc, N, H in [(32, 2, 64), (64, 2, 32), (128, 2, 16), (32, 3, 64), (256, 2, 8), (128, 2, 20)]:      # (the last one: general tile program)
        W = 2048 // Cc
        x = nhwc(bf(torch.randn(N, Cc, H, W, generator=g)))
        wp, _ = pack(nv, bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc)))
        y = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
        st = torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
        d = desc(nv, x, wp, y, N, H, W, Cc)
        d.stats = st.data_ptr()
        keep += [x, wp, st]
        descs.append(d)
        outs.append(y)
    singles = []
    for d, y in zip(descs, outs):
        nv.call("hrp_conv2d_fwd", C.byref(d), None)
        torch.cuda.synchronize()
        singles.append(y.clone())
        y.zero_()
    n = len(descs)
    arr = (nv.ConvDesc * n)(*descs)
    info = nv.BatchInfo()
    nb = int(nv.lib().hrp_batch_table_bytes(nv.BATCH_CONV, n))
    host = (C.c_char * nb)()
    nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_CONV, arr, n, host, C.byref(info)), "prepare")
    table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(nv.lib().hrp_batch_launch(table.data_ptr(), C.byref(info), None), "launch")
    torch.cuda.synchronize()
    for y, s in zip(outs, singles):
        assert s.float().abs().max() > 0.1
        assert torch.equal(y, s)


@pytest.mark.parametrize("Cc", [32, 64, 128, 256])
def test_fused_basic_block_equals_elementwise_path(Cc):
    """A train-mode bf16 BasicBlock (HRnet.py:28-57) with its interior BatchNorm + ReLU inside the row-strip convolutions
    (plan.conv_bn_relu_conv) against the same block with the element-wise passes (HRP_NO_ROWCONV_FUSE path): outputs,
    input gradient, weight and BatchNorm gradients, running statistics."""
    from hrpe_amd import plan as P
    from hrpe_amd.lib.models.backbones import HRnet as Hn
    W = 2048 // Cc
    N, H = 4, W
    g = torch.Generator().manual_seed(Cc)
    x = torch.randn(N, Cc, H, W, generator=g)
    gy = torch.randn(N, Cc, H, W, generator=g)
    ref = Hn.BasicBlock(Cc, Cc)
    with torch.no_grad():
        for prm in ref.parameters():
            if prm.dim() == 1:
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5 if prm is ref.bn1.weight or prm is ref.bn2.weight
                          else torch.randn(prm.shape, generator=g) * 0.2)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    res = {}
    for fused in (True, False):
        P.ROWCONV_FUSE = fused
        try:
            m = Hn.BasicBlock(Cc, Cc)
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
            m = m.to(DEV).set_compute_dtype(torch.bfloat16).train()
            xd = x.to(DEV).requires_grad_(True)
            y = m(xd)
            (y * gy.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            res[fused] = dict(y=y.detach().float().cpu(), dx=xd.grad.float().cpu(),
                              **{n: p.grad.float().cpu() for n, p in m.named_parameters()},
                              **{n: b.float().cpu() for n, b in m.named_buffers()})
        finally:
            P.ROWCONV_FUSE = True
    for k in res[True]:
        a, b = res[True][k].double(), res[False][k].double()
        err = ((a - b).norm() / (b.norm() + 1e-12)).item()
        assert err < 1e-2, (k, err)


class _RefBlock(torch.nn.Module):
    """torch restatement of the reference BasicBlock (HRnet.py:28-57), fp32."""

    def __init__(self, Cc):
        super().__init__()
        self.conv1, self.bn1 = torch.nn.Conv2d(Cc, Cc, 3, padding=1, bias=False), torch.nn.BatchNorm2d(Cc)
        self.conv2, self.bn2 = torch.nn.Conv2d(Cc, Cc, 3, padding=1, bias=False), torch.nn.BatchNorm2d(Cc)

    def forward(self, x):
        return torch.relu(self.bn2(self.conv2(torch.relu(self.bn1(self.conv1(x))))) + x)


@pytest.mark.parametrize("Cc", [32, 64, 128, 256])
def test_block_stack_with_fused_block_end_backward_equals_elementwise_backward(Cc):
    """Four train-mode bf16 BasicBlocks in a row (one branch of an HRNet stage, HRnet.py:28-57 / :146-163): the block-end
    activation's backward inside the row-strip data gradients (apply pass -> conv2's prologue; reduce pass -> the next block's
    conv1 epilogue, hrp_ew_bwd_reduce only for the last block) against HRP_NO_BLOCK_END_FUSE (hrp_ew_bwd_reduce +
    hrp_ew_bwd_apply per block) and against torch fp32: the plan really dropped the launches, and every gradient is as close
    to fp32 as the element-wise path's.

    Tolerances: the two bf16 paths round at different places (the fused path never stores the masked gradient of the
    shortcut, the reduce sums in another order), a bf16 rounding then flips and a ReLU decision with it: measured
    fused-vs-unfused 1 % of the norm - so the gate is the distance to fp32: fused <= 1.15 x unfused + 2e-3 per tensor, and
    4e-2 between the two."""
    from hrpe_amd import plan as P
    from hrpe_amd.runtime import SingleTensorModule
    from hrpe_amd.lib.models.backbones import HRnet as Hn
    NB = 4

    class Stack(SingleTensorModule):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([Hn.BasicBlock(Cc, Cc) for _ in range(NB)])

        def emit(self, pb, x):
            for b in self.blocks:
                x = b.emit(pb, x)
            return x

    W = 2048 // Cc
    N, H = 4, W
    g = torch.Generator().manual_seed(Cc + 1)
    x = torch.randn(N, Cc, H, W, generator=g)
    gy = torch.randn(N, Cc, H, W, generator=g)
    ref = Stack()
    with torch.no_grad():
        for n, prm in ref.named_parameters():
            if prm.dim() == 1:
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5 if n.endswith("weight") else torch.randn(prm.shape, generator=g) * 0.2)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    rm = torch.nn.Module()
    rm.blocks = torch.nn.ModuleList([_RefBlock(Cc) for _ in range(NB)])
    rm.load_state_dict(sd)
    rm.train()
    xr = x.clone().requires_grad_(True)
    yr = xr
    for b in rm.blocks:
        yr = b(yr)
    (yr * gy).sum().backward()
    want = dict(y=yr.detach(), dx=xr.grad, **{n: p.grad for n, p in rm.named_parameters()})
    res, counters = {}, {}
    for fused in (True, False):
        P.BLOCK_END_FUSE = fused
        try:
            m = Stack()
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
            m = m.to(DEV).set_compute_dtype(torch.bfloat16).train()
            xd = x.to(DEV).requires_grad_(True)
            y = m(xd)
            (y * gy.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            res[fused] = dict(y=y.detach().float().cpu(), dx=xd.grad.float().cpu(),
                              **{n: p.grad.float().cpu() for n, p in m.named_parameters()})
            counters[fused] = dict(next(iter(m._plans.values())).plan.counters)
        finally:
            P.BLOCK_END_FUSE = True
    assert counters[True].get("block_end_apply_fused") == NB and counters[True].get("block_end_reduce_fused") == NB - 1, counters[True]
    assert counters[True].get("block_end_masked_residual") == NB, counters[True]
    assert not counters[False].get("block_end_apply_fused")

    def err(a, b):
        return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-12)).item()

    for k in want:
        ef, eu, ab = err(res[True][k], want[k]), err(res[False][k], want[k]), err(res[True][k], res[False][k])
        assert ef <= 1.15 * eu + 2e-3, (k, ef, eu)
        assert ab < 4e-2, (k, ab)


@pytest.mark.parametrize("Cc", [32, 64, 128, 256])
def test_block_stack_with_block_end_forward_in_the_next_conv1(Cc):
    """Round 6: the block-end FORWARD pass of three of the four blocks of a branch stack runs inside the next
    block's conv1 (row-strip pro_mode 3; plan.BLOCK_END_FWD_FUSE) - the last block's output is read by something else (here: the plan
    output) and keeps its hrp_ew_fwd launch.  Same stack, same bounds as the test above: every output and gradient as close to torch
    fp32 as the plan without the fusion (<= 1.15 x + 2e-3), the two plans within 4e-2 of each other; the launches really went."""
    from hrpe_amd import _native as nv
    from hrpe_amd import plan as P
    from hrpe_amd.runtime import SingleTensorModule
    from hrpe_amd.lib.models.backbones import HRnet as Hn
    NB = 4

    class Stack(SingleTensorModule):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([Hn.BasicBlock(Cc, Cc) for _ in range(NB)])

        def emit(self, pb, x):
            for b in self.blocks:
                x = b.emit(pb, x)
            return x

    W = 2048 // Cc
    N, H = 4, W
    g = torch.Generator().manual_seed(Cc + 2)
    x = torch.randn(N, Cc, H, W, generator=g)
    gy = torch.randn(N, Cc, H, W, generator=g)
    ref = Stack()
    with torch.no_grad():
        for n, prm in ref.named_parameters():
            if prm.dim() == 1:
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5 if n.endswith("weight") else torch.randn(prm.shape, generator=g) * 0.2)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    rm = torch.nn.Module()
    rm.blocks = torch.nn.ModuleList([_RefBlock(Cc) for _ in range(NB)])
    rm.load_state_dict(sd)
    rm.train()
    xr = x.clone().requires_grad_(True)
    yr = xr
    for b in rm.blocks:
        yr = b(yr)
    (yr * gy).sum().backward()
    want = dict(y=yr.detach(), dx=xr.grad, **{n: p.grad for n, p in rm.named_parameters()},
                **{n: b for n, b in rm.named_buffers() if "running" in n})
    res, counters, ew_launches = {}, {}, {}
    saved = P.BLOCK_END_FWD_FUSE
    for fused in (True, False):
        P.BLOCK_END_FWD_FUSE = fused
        try:
            m = Stack()
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
            m = m.to(DEV).set_compute_dtype(torch.bfloat16).train()
            xd = x.to(DEV).requires_grad_(True)
            y = m(xd)
            (y * gy.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            res[fused] = dict(y=y.detach().float().cpu(), dx=xd.grad.float().cpu(),
                              **{n: p.grad.float().cpu() for n, p in m.named_parameters()},
                              **{n: b.float().cpu() for n, b in m.named_buffers() if "running" in n})
            plan = next(iter(m._plans.values())).plan
            counters[fused] = dict(plan.counters)
            ew_launches[fused] = sum(1 for e in plan.fwd if isinstance(e.op, P.Launch) and e.op.fam == "ew_fwd")
        finally:
            P.BLOCK_END_FWD_FUSE = saved
    assert counters[True].get("block_end_forward_fused") == NB - 1 and not counters[False].get("block_end_forward_fused")
    assert ew_launches[True] == 1 and ew_launches[False] == NB, ew_launches

    def err(a, b):
        return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-12)).item()

    for k in want:
        ef, eu, ab = err(res[True][k], want[k]), err(res[False][k], want[k]), err(res[True][k], res[False][k])
        assert ef <= 1.15 * eu + 2e-3, (k, ef, eu)
        assert ab < 4e-2, (k, ab)


@pytest.mark.parametrize("dtype", ["bf16", "f32", "f32x3"])
def test_compact_weight_packing_equals_the_rectangular_launch(dtype):
    """hrp_pack_weights_compact (a network's table, hrp_pack_blocks workgroups per entry) writes what hrp_pack_weights writes, byte
    for byte: 1x1 / 3x3 / 16-tap layers, channel counts that are no multiple of a chunk, entries without a transposed packing, padded tap
    slots (which stay what the caller put there)."""
    nv = nvmod()
    code = {"bf16": nv.HRP_BF16, "f32": nv.HRP_F32, "f32x3": nv.HRP_F32X3}[dtype]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    ck = 16 if dtype == "bf16" else 8
    shapes = [(32, 32, 9, True, 0), (64, 3, 9, False, 0), (256, 64, 1, True, 0), (24, 40, 9, True, 0), (64, 12, 16, False, 0),
              (128, 64, 9, True, 1), (7, 2048, 1, True, 0), (512, 256, 9, True, 0), (1024, 512, 9, True, 0)]
    g = torch.Generator(device="cpu").manual_seed(11)
    outs = []
    for compact in (False, True):
        tab = (nv.PackEntry * len(shapes))()
        keep, first, most = [], [0], 0
        for i, (cout, cin, nt, need_t, pad_t) in enumerate(shapes):
            w = torch.randn(cout, cin, nt, generator=torch.Generator(device="cpu").manual_seed(100 + i)).to(DEV)
            nf = -(-cin // ck) * nt * rup(cout, 32) * ck
            nb = -(-cout // ck) * (nt + pad_t) * rup(cin, 32) * ck
            dst = torch.full((nf,), 7.0, dtype=tdt, device=DEV)
            dst_t = torch.full((nb,), 7.0, dtype=tdt, device=DEV) if need_t else None
            tab[i].src, tab[i].dst, tab[i].dst_t = w.data_ptr(), dst.data_ptr(), dst_t.data_ptr() if need_t else None
            tab[i].Cout, tab[i].Cin, tab[i].ntaps, tab[i].pad_t = cout, cin, nt, pad_t
            keep += [w, dst, dst_t]
            first.append(first[-1] + nv.lib().hrp_pack_blocks(cout, cin, nt, code, 1, 1 if need_t else 0))
            most = max(most, nf, nb)
        tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
        if compact:
            fdev = torch.tensor(first, dtype=torch.int32, device=DEV)
            nv.call("hrp_pack_weights_compact", tdev.data_ptr(), fdev.data_ptr(), len(shapes), first[-1], code, None)
        else:
            nv.call("hrp_pack_weights", tdev.data_ptr(), len(shapes), code, most, None)
        torch.cuda.synchronize()
        outs.append(keep)
    assert first[-1] < 2000          # (the rectangular grid of this table: 9 x 256 workgroups of which most find no row)
    for a, b in zip(*outs):
        if a is not None:
            assert torch.equal(a, b)
    # the padded tap slots of the transposed packing were not touched
    cout, cin, nt, _, pad_t = shapes[5]
    dst_t = outs[1][3 * 5 + 2].view(-(-cout // ck), nt + pad_t, rup(cin, 32), ck)
    assert bool((dst_t[:, nt:] == 7.0).all()) and not bool((dst_t[:, :nt] == 7.0).all())
