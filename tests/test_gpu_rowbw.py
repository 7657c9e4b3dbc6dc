"""GPU parity tests of the fused row-strip backward kernel (csrc/conv_rowbw.hip, hrp_rowbw_*) through the C ABI: data gradient
+ weight gradient of a BasicBlock convolution (reference HRnet.py:41-57, autograd of conv1 / conv2) from one staging of the
output gradient.

Checked against (a) the kernels it replaces - hrp_conv2d_fwd with the same descriptor (+ the BatchNorm-input gradient as a side
output) followed by hrp_conv2d_bwd_weight on that side output: the data gradient must be bit-identical (same MFMA order), the
statistics and the weight gradient agree to summation order - and (b) plain torch fp32 / fp64 on the CPU (2e-2 of the tensor's
scale on bf16 outputs, 2e-3 on the fp32 weight gradient, whose operands are exactly the bf16 values both kernels multiply)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_rowconv as R  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-5
SLOTS = 8


def fold(nv, folds):
    L = nv.lib()
    n = len(folds)
    farr = (nv.WgradFoldDesc * n)(*folds)
    finfo = nv.BatchInfo()
    fhost = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, n)))()
    nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, farr, n, fhost, C.byref(finfo)), "fold prepare")
    ftab = torch.frombuffer(bytearray(bytes(fhost)), dtype=torch.uint8).to(DEV)
    nv.check(L.hrp_batch_launch(ftab.data_ptr(), C.byref(finfo), None), "fold launch")
    torch.cuda.synchronize()


def run_rowbw(nv, qs, max_wgs=0):
    """qs: list of RowBwDesc (workspace unset) -> launches them as ONE fused launch + the fold.  Returns info."""
    L = nv.lib()
    n = len(qs)
    arr = (nv.RowBwDesc * n)(*qs)
    info = nv.RowBwInfo()
    nv.check(L.hrp_rowbw_prepare(arr, n, max_wgs, None, C.byref(info)), "rowbw size query")
    keep = []
    for i in range(n):
        ws = torch.full((int(info.ws_bytes[i]) // 4 + 4,), float("nan"), device=DEV)      # every slab element must be written
        arr[i].workspace, arr[i].workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
        keep.append(ws)
    table = (C.c_char * int(L.hrp_rowbw_table_bytes()))()
    nv.check(L.hrp_rowbw_prepare(arr, n, max_wgs, table, C.byref(info)), "rowbw prepare")
    nv.check(L.hrp_rowbw_launch(table, C.byref(info), None), "rowbw launch")
    torch.cuda.synchronize()
    folds = (nv.WgradFoldDesc * n)()
    nv.check(L.hrp_rowbw_fold_descs(arr, C.byref(info), folds), "rowbw fold descs")
    fold(nv, list(folds))
    return info, keep


def wgrad_separate(nv, x_dev, dy_dev, N, H, W, Cc):
    """The kernel the fused one replaces: hrp_conv2d_bwd_weight (single launch, immediate fold)."""
    g = nv.WgradDesc()
    dw = torch.zeros(Cc * Cc * 9, device=DEV)
    g.x, g.dy, g.dw, g.dtype = x_dev.data_ptr(), dy_dev.data_ptr(), dw.data_ptr(), nv.HRP_BF16
    g.N, g.H, g.W, g.Cin, g.x_pitch = N, H, W, Cc, Cc
    g.Ho, g.Wo, g.Cout, g.dy_pitch = H, W, Cc, Cc
    g.in_stride, g.ntaps = 1, 9
    for i, (a, b) in enumerate(R.TAPS3):
        g.dy_t[i], g.dx_t[i] = a, b
    g.dw_cin, g.accumulate = Cc, 0
    need = int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g)))
    ws = torch.zeros(need // 4 + 4, device=DEV)
    g.workspace, g.workspace_bytes = ws.data_ptr(), need
    nv.call("hrp_conv2d_bwd_weight", C.byref(g), None)
    torch.cuda.synchronize()
    return dw.view(Cc, Cc, 3, 3).cpu()


def wgrad_ref(x_nchw, dy_nchw):
    """dW[co][ci][ky][kx] = sum dy[p][co] x[p + (ky-1, kx-1)][ci] in fp64."""
    Cc = x_nchw.shape[1]
    return torch.nn.grad.conv2d_weight(x_nchw.double(), (Cc, Cc, 3, 3), dy_nchw.double(), padding=1).float()


def build_block_problem(nv, Cc, N, H, seed, kind):
    """One data gradient of a fused BasicBlock with every operand the plan passes (tests/test_gpu_rowconv.py has the same
    constructions for the unfused launches).  kind "g2": conv2's data gradient (block-end BatchNorm + ReLU backward while
    staging, bit mask; epilogue reduce of the interior BatchNorm with the mask recomputed; X operand = relu(bn1(y1)) recomputed),
    kind "g1": conv1's data gradient (interior BatchNorm backward while staging; masked residual; X operand = the block input).
    -> dict with the descriptor pieces and the CPU reference tensors."""
    W = 2048 // Cc
    g = torch.Generator().manual_seed(seed)
    bf = R.bf
    gin = bf(torch.randn(N, Cc, H, W, generator=g))                       # gradient of the activation that is staged
    w = bf(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc))
    xb = bf(torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3)             # BatchNorm input of the staged activation (pro_x2)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    m, inv, sc, sh, tot, cnt = R.bn_consts(xb, gamma, beta)
    act = xb * sc[None, :, None, None] + sh[None, :, None, None]
    if kind == "g2":
        on = torch.rand(N, Cc, H, W, generator=g) > 0.45                  # block-end ReLU: bit mask (depends on the residual too)
    else:
        gin = gin * (act.abs() > 1e-4)                                    # recomputed mask: no gradient where it could fall either way
        on = act > 0
    gm = gin * on
    xh = (xb - m[None, :, None, None]) * inv[None, :, None, None]
    bt = torch.cat([gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))])
    k0, k1 = bt[:Cc] / cnt, bt[Cc:] / cnt
    dyv = sc[None, :, None, None] * (gm - k0[None, :, None, None] - xh * k1[None, :, None, None])      # gradient of the conv output
    dyb = bf(dyv)
    ref_dx = F.conv_transpose2d(dyb.double(), w.double(), padding=1).float()
    out = dict(Cc=Cc, N=N, H=H, W=W, w=w, gin=gin, xb=xb, gamma=gamma, beta=beta, tot=tot, bt=bt, cnt=cnt, on=on, dyv=dyv, dyb=dyb,
               ref_dx=ref_dx, g=g, kind=kind)
    if kind == "g2":
        y1 = bf(torch.randn(N, Cc, H, W, generator=g) * 2.0 - 0.4)          # interior BatchNorm input: epilogue reduce + X operand
        gamma1, beta1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
        m1, inv1, sc1, sh1, tot1, _ = R.bn_consts(y1, gamma1, beta1)
        hact = bf(torch.relu(y1 * sc1[None, :, None, None] + sh1[None, :, None, None]))
        out.update(y1=y1, gamma1=gamma1, beta1=beta1, tot1=tot1, xop=hact, sc1=sc1, sh1=sh1, m1=m1, inv1=inv1)
    else:
        xin = bf(torch.randn(N, Cc, H, W, generator=g))
        prev = bf(torch.randn(N, Cc, H, W, generator=g))
        onr = torch.rand(N, Cc, H, W, generator=g) > 0.5
        out.update(xop=xin, prev=prev, onr=onr, ref_dx=ref_dx + prev * onr)
    out["ref_dw"] = wgrad_ref(out["xop"], dyb)
    return out


def device_operands(nv, pb):
    """Device tensors + the data-gradient descriptor (without side outputs) of a problem."""
    Cc, N, H, W = pb["Cc"], pb["N"], pb["H"], pb["W"]
    g = pb["g"]
    keep = {}
    _, wpt = R.pack(nv, pb["w"])
    keep["w"] = wpt
    keep["gin"], keep["xb"] = R.nhwc(pb["gin"]), R.nhwc(pb["xb"])
    keep["y"] = torch.zeros(N * H * W * Cc, dtype=torch.bfloat16, device=DEV)
    keep["st"], keep["bs"] = R.slots_of(pb["tot"], g), R.slots_of(pb["bt"], g)
    keep["gam"], keep["bet"] = pb["gamma"].to(DEV), pb["beta"].to(DEV)
    d = R.desc(nv, keep["gin"], wpt, keep["y"], N, H, W, Cc, transposed=True)
    d.pro_mode, d.pro_x2, d.pro_stats, d.pro_bsums = 2, keep["xb"].data_ptr(), keep["st"].data_ptr(), keep["bs"].data_ptr()
    d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps = keep["gam"].data_ptr(), keep["bet"].data_ptr(), float(pb["cnt"]), EPS
    if pb["kind"] == "g2":
        keep["mk"] = R.mask_bits(pb["on"])
        d.pro_mask = keep["mk"].data_ptr()
        keep["y1"] = R.nhwc(pb["y1"])
        keep["st1"] = R.slots_of(pb["tot1"], g)
        keep["bs1"] = torch.zeros(SLOTS * 2 * Cc, dtype=torch.float64, device=DEV)
        keep["gam1"], keep["bet1"] = pb["gamma1"].to(DEV), pb["beta1"].to(DEV)
        d.stats, d.bnb_x, d.bnb_x_pitch = keep["bs1"].data_ptr(), keep["y1"].data_ptr(), Cc
        d.bnb_stats, d.bnb_gamma, d.bnb_beta = keep["st1"].data_ptr(), keep["gam1"].data_ptr(), keep["bet1"].data_ptr()
        d.bnb_count, d.bnb_eps = float(pb["cnt"]), EPS
        keep["wgx"] = keep["y1"]
    else:
        keep["res"], keep["mkr"] = R.nhwc(pb["prev"]), R.mask_bits(pb["onr"])
        keep["y"].fill_(3.0)
        d.res, d.res_mask = keep["res"].data_ptr(), keep["mkr"].data_ptr()
        keep["wgx"] = R.nhwc(pb["xop"])
    return d, keep


def rowbw_desc(nv, d, keep, pb, dw):
    q = nv.RowBwDesc()
    C.memmove(C.byref(q.conv), C.byref(d), C.sizeof(nv.ConvDesc))
    q.conv.pro_side = None
    q.wg_x, q.dw = keep["wgx"].data_ptr(), dw.data_ptr()
    q.wg_act = 1 if pb["kind"] == "g2" else 0
    q.accumulate = 0
    return q


def check_problem(nv, pb, d, keep, dw, y_sep, dw_sep, bs_sep):
    Cc, N, H, W = pb["Cc"], pb["N"], pb["H"], pb["W"]
    got = R.from_nhwc(keep["y"], N, H, W, Cc)
    assert R.rel(got, pb["ref_dx"]) < 2e-2, ("data gradient vs torch", R.rel(got, pb["ref_dx"]))
    assert R.rel(got, y_sep) < 1e-2, ("data gradient vs the separate kernel", R.rel(got, y_sep))      # (K chunks are summed in another order)
    dwg = dw.view(Cc, Cc, 3, 3).cpu()
    assert torch.isfinite(dwg).all()
    e_ref, e_sep = R.rel(dwg, pb["ref_dw"]), R.rel(dwg, dw_sep)
    assert e_ref < 2e-3, ("weight gradient vs torch", e_ref, [R.rel(dwg[:, :, a, b], pb["ref_dw"][:, :, a, b]) for a in range(3) for b in range(3)])
    assert e_sep < 2e-3, ("weight gradient vs the separate kernel", e_sep)
    if bs_sep is not None:
        s_new = keep["bs1"].view(SLOTS, 2 * Cc).sum(0).float().cpu()
        assert R.rel(s_new, bs_sep) < 2e-3, ("epilogue reduce vs the separate kernel", R.rel(s_new, bs_sep))


def run_separate(nv, pb, d, keep):
    """hrp_conv2d_fwd (side output = gradient of the conv output) + hrp_conv2d_bwd_weight: what the fused launch replaces."""
    Cc, N, H, W = pb["Cc"], pb["N"], pb["H"], pb["W"]
    side = torch.full((N * H * W * Cc,), 7.0, dtype=torch.bfloat16, device=DEV)
    d.pro_side = side.data_ptr()
    if pb["kind"] == "g1":
        keep["y"].fill_(3.0)
    nv.call("hrp_conv2d_fwd", C.byref(d), None)
    torch.cuda.synchronize()
    y_sep = R.from_nhwc(keep["y"], N, H, W, Cc).clone()
    bs_sep = None
    if pb["kind"] == "g2":
        bs_sep = keep["bs1"].view(SLOTS, 2 * Cc).sum(0).float().cpu()
        keep["bs1"].zero_()
    xop = R.nhwc(pb["xop"])
    dw_sep = wgrad_separate(nv, xop, side, N, H, W, Cc)
    gside = R.from_nhwc(side, N, H, W, Cc)
    assert R.rel(gside, pb["dyv"]) < 1.5e-2
    d.pro_side = None
    keep["y"].fill_(3.0 if pb["kind"] == "g1" else 0.0)
    return y_sep, dw_sep, bs_sep


SHAPES = [(32, 3, 64), (64, 2, 32), (32, 1, 16), (64, 5, 8), (32, 16, 64), (64, 16, 32)]      # (C, N, H); W = 2048 / C


@pytest.mark.parametrize("kind", ["g2", "g1"])
@pytest.mark.parametrize("shape", SHAPES)
def test_rowbw_single_problem(shape, kind):
    nv = R.nvmod()
    Cc, N, H = shape
    pb = build_block_problem(nv, Cc, N, H, Cc * 41 + N + (7 if kind == "g1" else 0), kind)
    d, keep = device_operands(nv, pb)
    y_sep, dw_sep, bs_sep = run_separate(nv, pb, d, keep)
    dw = torch.full((Cc * Cc * 9,), 5.0, device=DEV)
    q = rowbw_desc(nv, d, keep, pb, dw)
    assert nv.lib().hrp_rowbw_channels(C.byref(q)) == Cc
    info, _ws = run_rowbw(nv, [q])
    assert info.total_strips == N * H // 8 and info.G[0] == info.grid
    check_problem(nv, pb, d, keep, dw, y_sep, dw_sep, bs_sep)


@pytest.mark.parametrize("kind", ["g2", "g1"])
@pytest.mark.parametrize("max_wgs", [0, 5, 1])
def test_rowbw_two_problems_one_launch(max_wgs, kind):
    """A 32-channel and a 64-channel problem in one launch, strips split evenly over the workgroups: with 5 workgroups over
    24 + 20 strips one workgroup finishes the first problem and starts the second; with 1 a single workgroup walks everything;
    accumulate = 1 adds to the existing gradient."""
    nv = R.nvmod()
    pbs = [build_block_problem(nv, 32, 3, 64, 901, kind), build_block_problem(nv, 64, 5, 32, 902, kind)]      # (one form per launch)
    ops = [device_operands(nv, pb) for pb in pbs]
    seps = [run_separate(nv, pb, d, keep) for pb, (d, keep) in zip(pbs, ops)]
    dws = [torch.full((pb["Cc"] ** 2 * 9,), 0.25, device=DEV) for pb in pbs]
    qs = [rowbw_desc(nv, d, keep, pb, dw) for pb, (d, keep), dw in zip(pbs, ops, dws)]
    for q in qs:
        q.accumulate = 1
    info, _ws = run_rowbw(nv, qs, max_wgs)
    assert info.total_strips == 24 + 20
    if max_wgs == 5:
        assert info.grid == 5 and info.G[0] + info.G[1] in (5, 6)        # (a workgroup may hold strips of both)
    for pb, (d, keep), dw, sep in zip(pbs, ops, dws, seps):
        check_problem(nv, pb, d, keep, dw - 0.25, *sep)


def test_rowbw_rejects_what_it_cannot_run():
    nv = R.nvmod()
    pb = build_block_problem(nv, 32, 1, 16, 5, "g1")
    d, keep = device_operands(nv, pb)
    dw = torch.zeros(32 * 32 * 9, device=DEV)
    q = rowbw_desc(nv, d, keep, pb, dw)
    q2 = nv.RowBwDesc()
    C.memmove(C.byref(q2), C.byref(q), C.sizeof(nv.RowBwDesc))
    q2.wg_x = None
    assert nv.lib().hrp_rowbw_channels(C.byref(q2)) == 0
    arr = (nv.RowBwDesc * 2)(q, q)                    # two 32-channel problems: not one launch
    info = nv.RowBwInfo()
    assert nv.lib().hrp_rowbw_prepare(arr, 2, 0, None, C.byref(info)) == -1
    arr1 = (nv.RowBwDesc * 1)(q)
    table = (C.c_char * int(nv.lib().hrp_rowbw_table_bytes()))()
    assert nv.lib().hrp_rowbw_prepare(arr1, 1, 0, table, C.byref(info)) == -1      # table without workspace
