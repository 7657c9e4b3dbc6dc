"""GPU parity tests of the fused inference BasicBlock launch (csrc/conv_block.h, hrp_block_*) through the C ABI:
out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x) with folded BatchNorm (reference HRnet.py:41-57 in eval mode) against
(a) plain torch on the CPU in fp64 with the intermediate rounded to bf16 where the kernel rounds it, and (b) the two
hrp_conv2d_fwd launches it replaces.

Tolerance: bf16 operands and a bf16 intermediate, fp32 accumulation: 2e-2 of the output's scale against torch (one bf16 ulp
is 2^-8), 1e-2 against the unfused launches (same arithmetic, different summation order inside a pixel: an fp32 sum that
lands on the other side of a bf16 rounding boundary moves the intermediate by one ulp)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_rowconv import DEV, bf, desc, from_nhwc, nhwc, nvmod, pack, rel

pytestmark = pytest.mark.gpu


def make_block(nv, Cc, N, H, seed, keep):
    """-> (BlockDesc, torch reference NCHW fp32 (CPU), the two unfused conv descriptors, output tensor)"""
    W = 2048 // Cc
    g = torch.Generator().manual_seed(seed)
    x = bf(torch.randn(N, Cc, H, W, generator=g))
    w1 = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    w2 = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    sc1, sh1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    sc2, sh2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    w1p, _ = pack(nv, w1)
    w2p, _ = pack(nv, w2)
    xd = nhwc(x)
    y = torch.full((N * H * W * Cc,), float("nan"), dtype=torch.bfloat16, device=DEV)
    consts = [t.to(DEV).contiguous() for t in (sc1, sh1, sc2, sh2)]
    b = nv.BlockDesc()
    for dst, wp, sc, sh in ((b.conv1, w1p, consts[0], consts[1]), (b.conv2, w2p, consts[2], consts[3])):
        d = desc(nv, xd, wp, y, N, H, W, Cc)
        d.scale, d.shift, d.relu = sc.data_ptr(), sh.data_ptr(), 1
        C.memmove(C.byref(dst), C.byref(d), C.sizeof(nv.ConvDesc))
    b.conv1.y = None
    b.conv2.x = None
    b.conv2.res, b.conv2.res_pitch = xd.data_ptr(), Cc
    h = F.relu(F.conv2d(x.double(), w1.double(), padding=1) * sc1.double()[None, :, None, None] + sh1.double()[None, :, None, None])
    h = bf(h.float()).double()
    ref = F.relu(F.conv2d(h, w2.double(), padding=1) * sc2.double()[None, :, None, None] + sh2.double()[None, :, None, None] + x.double()).float()
    # the unfused pair
    hbuf = torch.zeros_like(y)
    y2 = torch.zeros_like(y)
    d1 = desc(nv, xd, w1p, hbuf, N, H, W, Cc)
    d1.scale, d1.shift, d1.relu = consts[0].data_ptr(), consts[1].data_ptr(), 1
    d2 = desc(nv, hbuf, w2p, y2, N, H, W, Cc)
    d2.scale, d2.shift, d2.relu, d2.res, d2.res_pitch = consts[2].data_ptr(), consts[3].data_ptr(), 1, xd.data_ptr(), Cc
    keep += [xd, w1p, w2p, y, consts, hbuf, y2]
    return b, ref, (d1, d2, y2), y, (N, H, W, Cc)


def launch(nv, blocks):
    n = len(blocks)
    arr = (nv.BlockDesc * n)(*blocks)
    info = nv.BlockInfo()
    table = (C.c_char * int(nv.lib().hrp_block_table_bytes()))()
    nv.check(nv.lib().hrp_block_prepare(arr, n, table, C.byref(info)), "hrp_block_prepare")
    nv.check(nv.lib().hrp_block_launch(table, C.byref(info), None), "hrp_block_launch")
    torch.cuda.synchronize()
    return info


SHAPES = [(32, 2, 64), (64, 3, 32), (32, 1, 8), (64, 1, 16), (32, 5, 16), (64, 2, 64), (32, 64, 64), (32, 3, 24)]   # (C, N, H)


@pytest.mark.parametrize("shape", SHAPES)
def test_block_single_problem(shape):
    nv = nvmod()
    Cc, N, H = shape
    keep = []
    b, ref, (d1, d2, y2), y, (N, H, W, Cc) = make_block(nv, Cc, N, H, Cc * 7 + N * 3 + H, keep)
    assert nv.lib().hrp_block_channels(C.byref(b)) == Cc
    info = launch(nv, [b])
    assert info.grid == N * info.bands[0]
    out = from_nhwc(y, N, H, W, Cc)
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 2e-2, rel(out, ref)
    nv.call("hrp_conv2d_fwd", C.byref(d1), None)
    nv.call("hrp_conv2d_fwd", C.byref(d2), None)
    torch.cuda.synchronize()
    assert rel(out, from_nhwc(y2, N, H, W, Cc)) < 1e-2


@pytest.mark.parametrize("N", [2, 9])
def test_block_two_problems_one_launch(N):
    nv = nvmod()
    keep = []
    b32, ref32, _, y32, g32 = make_block(nv, 32, N, 64, 11 + N, keep)
    b64, ref64, _, y64, g64 = make_block(nv, 64, N, 32, 23 + N, keep)
    info = launch(nv, [b32, b64])
    assert info.n == 2 and info.first_wg[1] == N * info.bands[0]
    assert rel(from_nhwc(y32, *g32), ref32) < 2e-2
    assert rel(from_nhwc(y64, *g64), ref64) < 2e-2


def test_block_is_deterministic_and_rejects_other_shapes():
    nv = nvmod()
    keep = []
    b, _, _, y, _ = make_block(nv, 32, 4, 64, 5, keep)
    launch(nv, [b])
    first = y.clone()
    for _ in range(3):
        y.fill_(0)
        launch(nv, [b])
        assert torch.equal(y, first)
    bad = nv.BlockDesc()
    C.memmove(C.byref(bad), C.byref(b), C.sizeof(nv.BlockDesc))
    bad.conv2.res = bad.conv2.y                      # the residual must be the block input
    assert nv.lib().hrp_block_channels(C.byref(bad)) == 0
    C.memmove(C.byref(bad), C.byref(b), C.sizeof(nv.BlockDesc))
    bad.conv1.relu = 0
    assert nv.lib().hrp_block_channels(C.byref(bad)) == 0
    info = nv.BlockInfo()
    arr = (nv.BlockDesc * 1)(bad)
    assert nv.lib().hrp_block_prepare(arr, 1, None, C.byref(info)) != 0
    b64, _, _, _, _ = make_block(nv, 64, 2, 32, 6, keep)
    arr2 = (nv.BlockDesc * 2)(b64, b)                 # order: the 32-channel block first
    assert nv.lib().hrp_block_prepare(arr2, 2, None, C.byref(info)) != 0


def test_inference_plan_uses_the_fused_block_and_agrees_with_the_unfused_plan():
    """A bf16 inference plan of the DepthNet (HRNet-W32, reference depth_net.py:92-137) emits one fused launch per BasicBlock of the
    two high-resolution branches (2 branches x 4 blocks x 8 modules = 64, paired 32 + 64 channels: 32 launches) and its output
    agrees with the plan built without them (two convolutions with folded epilogues per block) within bf16 rounding."""
    from hrpe_amd import plan as P
    from hrpe_amd.lib.models.backbones import HRnet
    from hrpe_amd.lib.models.depth_net import get_rootnet
    from synth import synth_inputs, synth_state_dict

    def run(fuse):
        saved = (P.BLOCK_FUSE, HRnet.EVAL_LANES)
        P.BLOCK_FUSE, HRnet.EVAL_LANES = fuse, ("flat22" if fuse else "")
        try:
            m = get_rootnet("hrnet32")
            m.load_state_dict(synth_state_dict(m.state_dict()))
            m = m.to(DEV).set_compute_dtype(torch.bfloat16).eval()
            x, _, kv, _ = synth_inputs(4)
            with torch.no_grad():
                out = m(x.to(DEV), kv.to(DEV)).float().cpu()
            plans = [r.plan for mod in m.modules() for r in getattr(mod, "_plans", {}).values()]
            return out, sum(p.counters.get("block_fused", 0) for p in plans)
        finally:
            P.BLOCK_FUSE, HRnet.EVAL_LANES = saved

    fused, nf = run(True)
    plain, n0 = run(False)
    assert n0 == 0 and nf == 64, (n0, nf)
    assert rel(fused, plain) < 2e-2, rel(fused, plain)
