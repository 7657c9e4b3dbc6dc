"""GPU parity tests of the pointwise convolution kernel (csrc/conv_pw.h) through the C ABI: the dense bf16 1x1 layers of
the Bottleneck blocks (reference HRnet.py:60-98) and their data gradients, with every epilogue option the plans use, against
plain torch fp32 on the CPU and against the general tile program on the same inputs.

Tolerance: bf16 operands, fp32 accumulation - 2e-2 of the tensor's scale on outputs, 2e-3 on fp32 statistics (compared with
sums over the kernel's own stored output, so only the summation order differs)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from test_gpu_rowconv import DEV, EPS, SLOTS, bf, bn_consts, from_nhwc, mask_bits, nhwc, nvmod, rel, rup, slots_of

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def small_problems_allowed():
    old = os.environ.get("HRP_PW_MIN_PIXELS")
    os.environ["HRP_PW_MIN_PIXELS"] = "1"
    yield
    if old is None:
        os.environ.pop("HRP_PW_MIN_PIXELS", None)
    else:
        os.environ["HRP_PW_MIN_PIXELS"] = old


def pack1(nv, w):
    """fp32 [Cout, Cin, 1, 1] -> (forward packing, transposed packing) in bf16."""
    cout, cin = w.shape[0], w.shape[1]
    nf = -(-cin // 16) * rup(cout, 32) * 16
    nb = -(-cout // 16) * rup(cin, 32) * 16
    dst = torch.zeros(nf, dtype=torch.bfloat16, device=DEV)
    dst_t = torch.zeros(nb, dtype=torch.bfloat16, device=DEV)
    tab = (nv.PackEntry * 1)()
    wd = w.to(DEV).contiguous()
    tab[0].src, tab[0].dst, tab[0].dst_t = wd.data_ptr(), dst.data_ptr(), dst_t.data_ptr()
    tab[0].Cout, tab[0].Cin, tab[0].ntaps = cout, cin, 1
    tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(DEV)
    nv.call("hrp_pack_weights", tdev.data_ptr(), 1, nv.HRP_BF16, max(nf, nb), None)
    torch.cuda.synchronize()
    return dst, dst_t


def desc1(nv, x, wp, y, N, H, W, cin, cout):
    d = nv.ConvDesc()
    d.x, d.w, d.y, d.dtype = x.data_ptr(), wp.data_ptr(), y.data_ptr(), nv.HRP_BF16
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, H, W, cin, cin
    d.Ho, d.Wo, d.Cout = H, W, cout
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = H, W, cout, cout
    d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 1, 1, rup(cout, 32)
    return d


def run(nv, d, expect_pw=True):
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == (1 if expect_pw else 0)
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()


# (Cin, Cout, N, H, W): every instantiation (k-steps 2 / 4 / 8 / 16, one / two channel blocks per wave), every wave layout
# (4 x 1, 2 x 2, 1 x 4 waves over channel blocks x pixel tiles), several channel groups, a ragged last tile (N H W % 32 != 0)
SHAPES = [(64, 256, 2, 64, 64), (256, 64, 2, 64, 64), (32, 128, 3, 16, 16), (128, 32, 2, 32, 32), (64, 64, 1, 24, 20),
          (32, 32, 5, 9, 7), (32, 64, 2, 16, 16), (128, 512, 2, 16, 16), (256, 256, 1, 16, 16), (64, 96, 2, 8, 8),
          (256, 32, 3, 11, 5), (32, 1024, 1, 8, 8)]


@pytest.mark.parametrize("shape", SHAPES)
def test_pwconv_plain_statistics_and_eval_epilogue(shape):
    """y = conv1x1(x, w) with the train-mode statistics, and the eval epilogue relu(conv * scale + shift + residual) with
    statistics of the stored values; the plain output bit-identical to the general tile program's."""
    nv = nvmod()
    cin, cout, N, H, W = shape
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = bf(torch.randn(N, cin, H, W, generator=g))
    w = bf(torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin))
    wp, _ = pack1(nv, w)
    xd = nhwc(x)
    y = torch.zeros(N * H * W * cout, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(SLOTS * 2 * cout, dtype=torch.float64, device=DEV)
    d = desc1(nv, xd, wp, y, N, H, W, cin, cout)
    d.stats = st.data_ptr()
    run(nv, d)
    ref = torch.nn.functional.conv2d(x.double(), w.double()).float()
    got = from_nhwc(y, N, H, W, cout)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    s = st.view(SLOTS, 2 * cout).sum(0).float().cpu()
    want = torch.cat([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert rel(s, want) < 2e-3, rel(s, want)
    # the tile program on the same problem
    y2 = torch.zeros_like(y)
    d.y = y2.data_ptr()
    keep = os.environ.get("HRP_PW_MIN_PIXELS")
    os.environ["HRP_PW_MIN_PIXELS"] = str(2 ** 31 - 1)       # beyond every problem: the general tile program
    try:
        run(nv, d, expect_pw=False)
    finally:
        if keep is None:
            del os.environ["HRP_PW_MIN_PIXELS"]
        else:
            os.environ["HRP_PW_MIN_PIXELS"] = keep
    assert torch.equal(y, y2)
    # eval epilogue
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    r = bf(torch.randn(N, cout, H, W, generator=g))
    rd, scd, shd = nhwc(r), sc.to(DEV), sh.to(DEV)
    st.zero_()
    d.y, d.scale, d.shift, d.res, d.relu = y.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(), 1
    run(nv, d)
    ref2 = torch.relu(ref * sc[None, :, None, None] + sh[None, :, None, None] + r)
    got = from_nhwc(y, N, H, W, cout)
    assert rel(got, ref2) < 2e-2, rel(got, ref2)
    s = st.view(SLOTS, 2 * cout).sum(0).float().cpu()
    want = torch.cat([got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))])
    assert rel(s, want) < 2e-3, rel(s, want)


@pytest.mark.parametrize("shape", SHAPES)
def test_pwconv_data_gradient_accumulates_onto_residual(shape):
    """The data gradient of a 1x1 layer (transposed packing) accumulated onto an existing gradient (res == y)."""
    nv = nvmod()
    cin, cout, N, H, W = shape                      # the launch computes cin <- cout ... as a conv with Cin = cout, Cout = cin
    g = torch.Generator().manual_seed(cin * 11 + cout)
    if cout not in (32, 64, 128, 256):
        pytest.skip("the transposed problem has an input width the kernel does not take")
    dy = bf(torch.randn(N, cout, H, W, generator=g))
    w = bf(torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cout))
    prev = bf(torch.randn(N, cin, H, W, generator=g))
    _, wpt = pack1(nv, w)
    dyd, y = nhwc(dy), nhwc(prev)
    d = desc1(nv, dyd, wpt, y, N, H, W, cout, cin)
    d.res = y.data_ptr()
    run(nv, d)
    ref = torch.nn.functional.conv_transpose2d(dy.double(), w.double()).float() + prev
    got = from_nhwc(y, N, H, W, cin)
    assert rel(got, ref) < 2e-2, rel(got, ref)


@pytest.mark.parametrize("form", ["bits+consts", "bits+stats", "recomputed"])
@pytest.mark.parametrize("shape", SHAPES[:8])
def test_pwconv_bn_backward_reduce_epilogue(shape, form):
    """Data gradient whose epilogue accumulates sum g, sum g * xhat of the stored gradient: the plans' form (ReLU mask as
    hrp_ew_fwd's bits, mean / invstd from bnb_consts), the bits with constants derived from the statistic slots, and the
    mask recomputed from the BatchNorm input == the conv followed by hrp_ew_bwd_reduce."""
    nv = nvmod()
    cin, cout, N, H, W = shape
    g = torch.Generator().manual_seed(cin * 13 + cout + len(form))
    dy = bf(torch.randn(N, cin, H, W, generator=g))
    w = bf(torch.randn(cin, cout, 1, 1, generator=g) / np.sqrt(cin))       # forward layer cout -> cin; its data gradient: cin -> cout
    x1 = bf(torch.randn(N, cout, H, W, generator=g) * 2.0 - 0.4)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = bn_consts(x1, gamma, beta)
    act = x1 * sc[None, :, None, None] + sh[None, :, None, None]
    on = (torch.rand(N, cout, H, W, generator=g) > 0.45) if form != "recomputed" else (act > 0)
    _, wpt = pack1(nv, w)
    dyd, x1d = nhwc(dy), nhwc(x1)
    y = torch.zeros(N * H * W * cout, dtype=torch.bfloat16, device=DEV)
    bs = torch.zeros(SLOTS * 2 * cout, dtype=torch.float64, device=DEV)
    keep = []
    d = desc1(nv, dyd, wpt, y, N, H, W, cin, cout)
    d.stats, d.bnb_x, d.bnb_x_pitch = bs.data_ptr(), x1d.data_ptr(), cout
    if form != "recomputed":
        mk = mask_bits(on)
        keep.append(mk)
        d.bnb_mask, d.bnb_mask_pitch = mk.data_ptr(), cout // 8
    if form == "bits+consts":
        cs = torch.cat([m, inv]).float().to(DEV)
        keep.append(cs)
        d.bnb_consts = cs.data_ptr()
    else:
        st_in, gd, bd = slots_of(tot, g), gamma.to(DEV), beta.to(DEV)
        keep += [st_in, gd, bd]
        d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps = st_in.data_ptr(), gd.data_ptr(), bd.data_ptr(), float(cnt), EPS
    if cin == 256:       # 16 k-steps with the epilogue reduce stay on the tile program (which knows the plans' form only)
        assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0
        if form != "bits+consts":
            return
        run(nv, d, expect_pw=False)
    else:
        run(nv, d)
    ref = torch.nn.functional.conv_transpose2d(dy.double(), w.double()).float()
    got = from_nhwc(y, N, H, W, cout)
    assert rel(got, ref) < 2e-2, rel(got, ref)
    sure = (act.abs() > 1e-4) if form == "recomputed" else torch.ones_like(on)
    gm = got * on * sure
    xh = (x1 - m[None, :, None, None]) * inv[None, :, None, None]
    want = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    s = bs.view(SLOTS, 2 * cout).sum(0).float().cpu()
    unsure = (got * (~sure)).abs().sum((0, 2, 3))
    err = (s - want).abs()
    bound = 2e-3 * want.abs().max() + torch.cat([unsure, unsure * xh.abs().max()])
    assert (err <= bound).all(), (err / bound).max()


def test_pwconv_declines_what_it_does_not_cover():
    """Bias, strides, taps, odd widths, fp32 and small problems (default threshold) stay on the tile program."""
    nv = nvmod()
    x = torch.zeros(64 * 64 * 64, dtype=torch.bfloat16, device=DEV)
    d = desc1(nv, x, x, x, 1, 64, 64, 64, 64)
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 1
    os.environ.pop("HRP_PW_MIN_PIXELS")
    assert nv.lib().hrp_conv_pointwise(C.byref(d)) == 0          # 4 096 pixels < 131 072
    os.environ["HRP_PW_MIN_PIXELS"] = "1"
    for field, val in (("bias", x.data_ptr()), ("in_stride", 2), ("out_stride", 2), ("ntaps", 2), ("Cin", 48), ("Cout", 40),
                       ("dtype", nv.HRP_F32), ("x_pitch", 72), ("y_pitch", 72), ("pro_mode", 1)):
        e = nv.ConvDesc.from_buffer_copy(bytes(d))
        setattr(e, field, val)
        assert nv.lib().hrp_conv_pointwise(C.byref(e)) == 0, field
