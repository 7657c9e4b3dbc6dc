"""GPU: round-2 additions - batched launches against the same problems launched one by one, the Philox dropout of the
regression heads on lane streams, refreshed inference plans after FusedClipAdam / running-statistics updates, the graph
cache of PlannedModule."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

from test_gpu_model import DEV, NAMES8, build_full, load
from synth import synth_inputs

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def _conv_problem(nv, N, hw, cin, cout, k, dtype, seed):
    import bench_kernels as bk
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N * hw * hw * cin, generator=g).to(DEV).to(dtype)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV)
    wp, _ = bk.pack(w, dtype)
    d = nv.ConvDesc()
    d.x, d.w = x.data_ptr(), wp.data_ptr()
    d.dtype = nv.HRP_BF16 if dtype == torch.bfloat16 else nv.HRP_F32
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, cin, cin
    d.Ho, d.Wo, d.Cout = hw, hw, cout
    d.y_H, d.y_W, d.y_pitch, d.res_pitch = hw, hw, cout, cout
    d.out_stride, d.in_stride = 1, 1
    taps = bk.TAPS3 if k == 3 else [(0, 0)]
    d.ntaps = d.w_ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dy[i], d.dx[i], d.wtap[i] = a, b, i
    d.w_cout_pad = bk.rup(cout, 32)
    return d, (x, wp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [3, 40])
def test_batched_conv_equals_single_launches(dtype, N):
    """hrp_batch_* conv: four problems of different shapes in ONE launch give bit-identical outputs to four single
    launches (a convolution output is one dot product in a fixed K order whatever the tile), and equal batch statistics up
    to the order of the fp32 atomics.  N = 40 reaches the persistent light kernel and images-per-tile > 1."""
    from hrpe_amd import _native as nv
    shapes = [(32, 32, 64), (64, 64, 32), (128, 128, 16), (256, 256, 8)]
    descs, keep = [], []
    for i, (cin, cout, hw) in enumerate(shapes):
        d, bufs = _conv_problem(nv, N, hw, cin, cout, 3, dtype, 100 + i)
        keep.append(bufs)
        descs.append(d)
    outs = {}
    for mode in ("single", "batch"):
        ys, sts = [], []
        for d in descs:
            y = torch.full((d.N * d.Ho * d.Wo * d.Cout,), 7.0, device=DEV).to(dtype)
            st = torch.zeros(16 * d.Cout, dtype=torch.float64, device=DEV)
            d.y, d.stats = y.data_ptr(), st.data_ptr()
            ys.append(y)
            sts.append(st)
        if mode == "single":
            for d in descs:
                nv.call("hrp_conv2d_fwd", C.byref(d), None)
        else:
            arr = (nv.ConvDesc * len(descs))(*descs)
            info = nv.BatchInfo()
            host = (C.c_char * int(nv.lib().hrp_batch_table_bytes(nv.BATCH_CONV, len(descs))))()
            nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_CONV, arr, len(descs), host, C.byref(info)), "prepare")
            tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
            nv.check(nv.lib().hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
        torch.cuda.synchronize()
        outs[mode] = (ys, sts)
    for (ya, sa), (yb, sb), d in zip(zip(*outs["single"]), zip(*outs["batch"]), descs):
        assert torch.equal(ya, yb), f"conv {d.Cin}->{d.Cout} @{d.Ho}: batched output differs from the single launch"
        a, b = sa.view(8, -1).sum(0), sb.view(8, -1).sum(0)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * float(a.abs().max())), "batch statistics"
    # light-only batch (two 32-channel problems): the LIGHT kernel variant, persistent workgroups when N is large
    # (bf16 @64x64 is the row-strip kernel's shape - tests/test_gpu_rowconv.py; 80x80 keeps this on the general tile program)
    lhw = 64 if dtype == torch.float32 else 80
    l0, k0 = _conv_problem(nv, N, lhw, 32, 32, 3, dtype, 9)
    l1, k1 = _conv_problem(nv, N, lhw, 32, 32, 3, dtype, 7)
    keep += [k0, k1]
    light = [l0, l1]
    ref = []
    for d in light:
        y = torch.zeros(d.N * d.Ho * d.Wo * d.Cout, device=DEV).to(dtype)
        d.y, d.stats = y.data_ptr(), None
        nv.call("hrp_conv2d_fwd", C.byref(d), None)
        ref.append(y)
    got = []
    for d in light:
        y = torch.zeros(d.N * d.Ho * d.Wo * d.Cout, device=DEV).to(dtype)
        d.y = y.data_ptr()
        got.append(y)
    arr = (nv.ConvDesc * 2)(*light)
    info = nv.BatchInfo()
    host = (C.c_char * int(nv.lib().hrp_batch_table_bytes(nv.BATCH_CONV, 2)))()
    nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_CONV, arr, 2, host, C.byref(info)), "prepare")
    assert info.variant == 109
    tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
    nv.check(nv.lib().hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("batched", [False, True])
def test_conv_epilogue_bn_backward_reduce(dtype, batched):
    """hrp_conv_desc.bnb_*: a (data-gradient) convolution that also accumulates the two BatchNorm-backward sums of the
    activation whose gradient it writes - sum g and sum g * (x - mean) * invstd with g = y masked by the ReLU bit
    mask - gives the sums a separate pass over the stored y computes, and leaves y itself untouched."""
    from hrpe_amd import _native as nv
    vec = 4 if dtype == torch.float32 else 8
    shapes = [(32, 32, 32, 3), (64, 64, 16, 3), (64, 32, 16, 1)]
    N = 5
    probs = []
    for i, (cin, cout, hw, k) in enumerate(shapes):
        d, keep = _conv_problem(nv, N, hw, cin, cout, k, dtype, 300 + i)
        g = torch.Generator(device="cpu").manual_seed(400 + i)
        npx = N * hw * hw
        bx = torch.randn(npx, cout, generator=g).to(DEV).to(dtype)
        mask = torch.randint(0, 1 << vec, (npx, cout // vec), generator=g, dtype=torch.int32).to(torch.uint8).to(DEV)
        consts = torch.cat([torch.randn(cout, generator=g) * 0.3, torch.rand(cout, generator=g) + 0.5]).to(DEV)
        y0 = torch.zeros(npx * cout, device=DEV).to(dtype)
        y1 = torch.zeros(npx * cout, device=DEV).to(dtype)
        sums = torch.zeros(8 * 2 * cout, dtype=torch.float64, device=DEV)
        probs.append((d, keep, bx, mask, consts, y0, y1, sums, cout))
    plain, fused = [], []
    for d, keep, bx, mask, consts, y0, y1, sums, cout in probs:
        d.y, d.stats = y0.data_ptr(), None
        nv.call("hrp_conv2d_fwd", C.byref(d), None)
        f = nv.ConvDesc.from_buffer_copy(d)
        f.y, f.stats = y1.data_ptr(), sums.data_ptr()
        f.bnb_x, f.bnb_x_pitch, f.bnb_mask, f.bnb_mask_pitch, f.bnb_consts = bx.data_ptr(), cout, mask.data_ptr(), cout // vec, consts.data_ptr()
        fused.append(f)
    if batched:
        for grp in (fused[:2], fused[2:]):     # (one tap count per batch)
            arr = (nv.ConvDesc * len(grp))(*grp)
            info = nv.BatchInfo()
            host = (C.c_char * int(nv.lib().hrp_batch_table_bytes(nv.BATCH_CONV, len(grp))))()
            nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_CONV, arr, len(grp), host, C.byref(info)), "prepare")
            tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
            nv.check(nv.lib().hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
    else:
        for f in fused:
            nv.call("hrp_conv2d_fwd", C.byref(f), None)
    torch.cuda.synchronize()
    for (d, keep, bx, mask, consts, y0, y1, sums, cout), (cin, _, hw, k) in zip(probs, shapes):
        assert torch.equal(y0, y1), "the fused launch must store the same y"
        yv = y0.float().view(-1, cout)
        bits = ((mask.to(torch.int32).unsqueeze(-1) >> torch.arange(vec, device=DEV)) & 1).reshape(-1, cout).float()
        gm = yv * bits
        xh = (bx.float() - consts[:cout]) * consts[cout:]
        want = torch.cat([gm.sum(0), (gm * xh).sum(0)]).double()
        got = sums.view(8, 2 * cout).sum(0).double()
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 2e-5, f"conv {cin}->{cout} k{k} @{hw}: BatchNorm backward sums differ by {err}"
    bad = nv.ConvDesc.from_buffer_copy(fused[0])
    bad.relu = 1
    assert nv.lib().hrp_conv2d_fwd(C.byref(bad), None) == -1 and b"bnb_x" in nv.lib().hrp_last_error()


def _wgrad_problem(nv, N, hw, cin, cout, k, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(N * hw * hw * cin, generator=g).to(DEV).to(dtype)
    dy = torch.randn(N * hw * hw * cout, generator=g).to(DEV).to(dtype)
    d = nv.WgradDesc()
    d.x, d.dy, d.dtype = x.data_ptr(), dy.data_ptr(), nv.HRP_BF16 if dtype == torch.bfloat16 else nv.HRP_F32
    d.N, d.H, d.W, d.Cin, d.x_pitch = N, hw, hw, cin, cin
    d.Ho, d.Wo, d.Cout, d.dy_pitch = hw, hw, cout, cout
    d.in_stride, d.ntaps = 1, k * k
    for i, (a, b) in enumerate([(ky - k // 2, kx - k // 2) for ky in range(k) for kx in range(k)]):
        d.dy_t[i], d.dx_t[i] = a, b
    d.dw_cin = cin
    return d, (x, dy)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_wgrad_fold_equals_immediate(dtype):
    """Weight gradients in two phases (descriptor phase 1 + ONE HRP_BATCH_WGRAD_FOLD launch over problems of different
    tap counts, batched and single) are bit-identical to the one-call form: same slabs, same fold order."""
    from hrpe_amd import _native as nv
    L = nv.lib()
    shapes = [(32, 32, 32, 3), (64, 64, 16, 3), (128, 128, 8, 3), (64, 128, 16, 1)]
    probs = [_wgrad_problem(nv, 6, hw, cin, cout, k, dtype, 50 + i) for i, (cin, cout, hw, k) in enumerate(shapes)]
    descs = [p[0] for p in probs]

    def run(phase):
        dws, wss, folds = [], [], []
        # the three 3x3 problems as one batched launch, the 1x1 problem as a single launch
        arr = (nv.WgradDesc * 3)(*descs[:3])
        for d in arr:
            d.phase, d.accumulate = phase, 0
            dws.append(torch.full((d.Cout * d.dw_cin * d.ntaps,), 3.0, device=DEV))
            d.dw = dws[-1].data_ptr()
        info = nv.BatchInfo()
        nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 3, None, C.byref(info)), "query")
        for i, d in enumerate(arr):
            ws = torch.zeros(int(info.ws_bytes[i]) // 4 + 4, device=DEV)
            d.workspace, d.workspace_bytes = ws.data_ptr(), int(info.ws_bytes[i])
            wss.append(ws)
        host = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD, 3)))()
        nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD, arr, 3, host, C.byref(info)), "prepare")
        assert (info.grid2 == 0) == (phase == 1)
        tab = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(DEV)
        nv.check(L.hrp_batch_launch(tab.data_ptr(), C.byref(info), None), "launch")
        if phase == 1:
            f3 = (nv.WgradFoldDesc * 3)()
            nv.check(L.hrp_batch_wgrad_fold_descs(host, C.byref(info), f3), "fold descs")
            folds += list(f3)
        d = descs[3]
        d.phase, d.accumulate = phase, 0
        dw = torch.full((d.Cout * d.dw_cin * d.ntaps,), 3.0, device=DEV)
        d.dw = dw.data_ptr()
        need = int(L.hrp_wgrad_workspace_bytes(C.byref(d)))
        assert need > 0
        ws = torch.zeros(need // 4 + 4, device=DEV)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
        nv.call("hrp_conv2d_bwd_weight", C.byref(d), None)
        wss.append(ws)
        dws.append(dw)
        if phase == 1:
            f = nv.WgradFoldDesc()
            nv.check(L.hrp_wgrad_fold_desc_of(C.byref(d), C.byref(f)), "fold desc")
            folds.append(f)
            torch.cuda.synchronize()
            assert all(float(dw.min()) == 3.0 == float(dw.max()) for dw in dws), "phase 1 must not touch dw"
            assert all(f.G > 0 for f in folds)
            farr = (nv.WgradFoldDesc * 4)(*folds)
            finfo = nv.BatchInfo()
            fhost = (C.c_char * int(L.hrp_batch_table_bytes(nv.BATCH_WGRAD_FOLD, 4)))()
            nv.check(L.hrp_batch_prepare(nv.BATCH_WGRAD_FOLD, farr, 4, fhost, C.byref(finfo)), "fold prepare")
            ftab = torch.frombuffer(bytearray(bytes(fhost)), dtype=torch.uint8).to(DEV)
            nv.check(L.hrp_batch_launch(ftab.data_ptr(), C.byref(finfo), None), "fold launch")
        torch.cuda.synchronize()
        return dws

    a, b = run(0), run(1)
    for x, y, (cin, cout, hw, k) in zip(a, b, shapes):
        assert float(x.abs().max()) > 0 and float((x - 3.0).abs().max()) > 0
        assert torch.equal(x, y), f"wgrad {cin}->{cout} k{k} @{hw}: deferred fold differs"


def test_plan_with_deferred_folds_matches_immediate_folds():
    """One training step of the full network with the weight-gradient folds deferred (default) and folded on the spot
    (HRP_NO_WGRAD_DEFER): the same gradients (the slab sums are identical; the BN-statistic atomics are the noise)."""
    from hrpe_amd import plan as P
    m = build_full().train()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    saved = P.WGRAD_DEFER
    grads = {}
    try:
        for name, defer in (("deferred", True), ("deferred_again", True), ("immediate", False)):
            P.WGRAD_DEFER = defer
            if name != "deferred_again":
                m.invalidate_plans()
            m.load_state_dict(sd0)
            m.zero_grad()
            out = m(x_reg, x_root, kv, K)
            sum(o.float().square().mean() for o in out).backward()
            torch.cuda.synchronize()
            grads[name] = m.flat_grads()[0].clone()
            if defer:
                tp = [r.plan for r in m._plans.values() if r.plan.need_grad][-1]
                assert any(isinstance(e.op, P.BatchLaunch) and e.op.fam == "wgrad_fold" for e in tp.bwd_ops())
    finally:
        P.WGRAD_DEFER = saved
        m.invalidate_plans()
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    noise = rel(grads["deferred_again"], grads["deferred"])
    assert float(grads["deferred"].abs().max()) > 0
    assert rel(grads["immediate"], grads["deferred"]) <= max(10 * noise, 1e-4), (rel(grads["immediate"], grads["deferred"]), noise)


def test_grouped_plan_matches_one_by_one_plan():
    """The same network with batched launches (default), with the launches one by one in the same order (HRP_NO_BATCH),
    fully merged on one stream and on the round-1 lanes: eval outputs to 2e-6 (split-K atomics of the single-launch
    path), a training step within the run-to-run noise of the fp32 atomics (BN statistics, weight-gradient slabs)."""
    from hrpe_amd import plan as P
    from hrpe_amd.lib.models.backbones import HRnet
    m = build_full().eval()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
    saved = (P.PLAN_MODE, P.BATCHING, HRnet.TRUNK_LANES)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}      # (the train-mode forward moves the running statistics)
    outs = {}
    try:
        for name, mode, batching, lanes in (("batched", "hybrid", True, "nets"), ("one_by_one", "hybrid", False, "nets"),
                                            ("merged", "merged", True, "nets"), ("lanes", "lanes", False, "flat"),
                                            ("flat2", "hybrid", True, "flat2")):
            P.PLAN_MODE, P.BATCHING, HRnet.TRUNK_LANES = mode, batching, lanes
            m.invalidate_plans()
            m.load_state_dict(sd0)
            m.eval()
            with torch.no_grad():
                ev = [o.clone() for o in m(x_reg, x_root, kv, K)]
            m.train()
            m.zero_grad()
            out = m(x_reg, x_root, kv, K)
            sum(o.float().square().mean() for o in out).backward()
            torch.cuda.synchronize()
            outs[name] = (ev, m.flat_grads()[0].clone())
            if name == "batched":      # second run of the same plan: the noise floor
                m.load_state_dict(sd0)
                m.zero_grad()
                out = m(x_reg, x_root, kv, K)
                sum(o.float().square().mean() for o in out).backward()
                torch.cuda.synchronize()
                outs["batched_again"] = (ev, m.flat_grads()[0].clone())
    finally:
        P.PLAN_MODE, P.BATCHING, HRnet.TRUNK_LANES = saved
        m.invalidate_plans()
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    noise = rel(outs["batched_again"][1], outs["batched"][1])
    for name in ("one_by_one", "merged", "lanes", "flat2"):
        for n, a, b in zip(NAMES8, outs[name][0], outs["batched"][0]):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), (name, n)
        e = rel(outs[name][1], outs["batched"][1])
        assert e <= max(10 * noise, 2e-3), (name, e, noise)
    # (not bit for bit at this batch size: the single-launch path runs the small fp32 1x1 fuse convs split-K with fp32
    # atomics, the batched path does not; test_batched_conv_equals_single_launches pins bit-identity kernel by kernel)


def test_dropout_mask_on_lane_streams():
    """the dropout mask is drawn by a Philox kernel on the stream of the lane that uses it.  With
    p_dropout = 0.5 the rotation head runs on a side stream: forward y == x * mask, backward dx == dy * mask with the SAME
    mask, masks differ between steps and between the two dropout layers, and the keep rate is ~0.5."""
    from hrpe_amd import plan as P
    from hrpe_amd.lib.models import full_net as FN
    FN.FUSED_REGRESSORS = False      # (the per-layer dropout op: what the regressor variants outside the fused chain still run)
    m = build_full(p_dropout=0.5).train()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
    masks_seen = []
    drops = []
    orig = P.PlanBuilder.dropout

    def spy(self, x, prob):
        y = orig(self, x, prob)
        drops.append((x, y, y.dropout_mask))
        return y
    P.PlanBuilder.dropout = spy
    try:
        for step in range(2):
            m.zero_grad()
            out = m(x_reg, x_root, kv, K)
            sum(o.float().square().mean() for o in out).backward()
            torch.cuda.synchronize()
            assert len(drops) == 16          # 2 heads x 4 iterations x 2 layers
            for x, y, mask in drops:
                xv = x.buf.view(x.N, x.pitch)[:, :x.C]
                yv = y.buf.view(y.N, y.pitch)[:, :y.C]
                assert set(torch.unique(mask).tolist()) <= {0.0, 2.0}
                assert torch.equal(yv, xv * mask), "forward: y == x * mask"
                gx = x.grad_buf().view(x.N, x.pitch)[:, :x.C]
                gy = y.grad_buf().view(y.N, y.pitch)[:, :y.C]
                assert torch.allclose(gx, gy * mask, rtol=0, atol=0), "backward uses the forward's mask"
            allm = torch.stack([mk for _, _, mk in drops])
            assert 0.45 < float((allm > 0).float().mean()) < 0.55
            assert not torch.equal(drops[0][2], drops[1][2])
            masks_seen.append(allm.clone())
    finally:
        P.PlanBuilder.dropout = orig
        FN.FUSED_REGRESSORS = True
    assert not torch.equal(masks_seen[0], masks_seen[1]), "a new mask every step"


def test_eval_plan_follows_fused_optimizer_and_running_stats():
    """FusedClipAdam and hrp_bn_running_update change parameters / buffers through raw pointers (no
    tensor._version bump); the cached inference plan must repack and refold anyway.  eval -> train step -> eval equals a
    freshly built model with the same state dict."""
    from hrpe_amd.optim import FusedClipAdam
    m = build_full()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
    m.eval()
    with torch.no_grad():
        before = [o.clone() for o in m(x_reg, x_root, kv, K)]
    opt = FusedClipAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, max_norm=5.0)
    m.train()
    out = m(x_reg, x_root, kv, K)
    sum(o.float().square().mean() for o in out).backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        after = [o.clone() for o in m(x_reg, x_root, kv, K)]
    fresh = build_full()
    fresh.load_state_dict(m.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh(x_reg, x_root, kv, K)
    assert any(float((a - b).abs().max()) > 1e-4 for a, b in zip(after, before)), "the step must change the outputs"
    for n, a, w in zip(NAMES8, after, want):
        assert float((a - w).abs().max()) <= 5e-5 * max(1.0, float(w.abs().max())), n   # (split-K fp32 atomics: 1.0e-5 measured)
    # running statistics only (a train-mode forward without an optimizer step) also invalidate the folded BatchNorm
    m.train()
    with torch.no_grad():
        m(x_reg, x_root, kv, K)
    m.eval()
    with torch.no_grad():
        after2 = m(x_reg, x_root, kv, K)
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        want2 = fresh(x_reg, x_root, kv, K)
    for n, a, w in zip(NAMES8, after2, want2):
        assert float((a - w).abs().max()) <= 5e-5 * max(1.0, float(w.abs().max())), n   # (split-K fp32 atomics: 1.0e-5 measured)


def test_module_graph_cache_matches_eager_and_speeds_up_the_plain_loop():
    """the reference's unmodified loop - model(...), loss.backward(), optimizer.step()
    (scripts/train_full.py:53-67) - replays captured forward / backward graphs per plan after two eager warm-up steps.
    Same numbers as the eager walk, and the plain loop at B = 64 costs about what one captured whole-step graph costs."""
    import time
    from hrpe_amd import runtime as R
    from hrpe_amd.lib.core.function import full_loss
    from hrpe_amd.optim import FusedClipAdam
    from test_gpu_parity import _train_step_inputs
    g = load("golden_full_train_b8.npz")
    m = build_full().train()
    x_reg, x_root, kv, K, gt = _train_step_inputs(g, m, 8)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}

    def step():
        m.zero_grad()
        loss, _ = full_loss(m(x_reg, x_root, kv, K), gt, K)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), m.flat_grads()[0].clone()
    res = {}
    for cache in (False, True):
        R.GRAPH_CACHE = cache
        m.invalidate_plans()
        out = []
        for i in range(5):
            m.load_state_dict(sd0)
            out.append(step())
        res[cache] = out
        if cache:
            r = next(iter(m._plans.values()))
            assert r.g_fwd is not None and r.g_bwd is not None, "forward and backward graphs are captured after the warm-up"
    R.GRAPH_CACHE = True
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))   # noqa: E731
    noise = rel(res[False][1][1], res[False][0][1])
    for i in range(5):
        assert abs(res[True][i][0] - res[False][i][0]) < 1e-4 * abs(res[False][i][0])
        assert rel(res[True][i][1], res[False][i][1]) <= max(10 * noise, 2e-3)
    # speed at the benchmark's batch size, bf16: plain loop with the graph cache vs ONE captured whole step
    B = 64
    m = build_full().train().set_compute_dtype(torch.bfloat16)
    rng = np.random.Generator(np.random.PCG64(3))
    xr = torch.tensor(rng.random((B, 3, 256, 256), dtype=np.float32)).to(DEV)
    xo = torch.tensor(rng.random((B, 3, 256, 256), dtype=np.float32)).to(DEV)
    Kb, kvb = K[:1].repeat(B, 1, 1).contiguous(), kv[:1].repeat(B).contiguous()
    gtb = {k: v[:1].repeat(B, *([1] * (v.dim() - 1))).contiguous() for k, v in gt.items()}
    opt = FusedClipAdam([p for p in m.parameters() if p.requires_grad], lr=1e-5, max_norm=5.0)

    def loop_step():
        loss, _ = full_loss(m(xr, xo, kvb, Kb), gtb, Kb)
        loss.backward()
        opt.step()
    for _ in range(4):
        loop_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loop_step()
    torch.cuda.synchronize()
    t_loop = (time.perf_counter() - t0) / 10
    R.GRAPH_CACHE = False
    try:
        m.invalidate_plans()
        for _ in range(2):
            loop_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            loop_step()
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / 5
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            loop_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        whole = torch.cuda.CUDAGraph()
        with torch.cuda.graph(whole):
            loop_step()
        for _ in range(2):
            whole.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            whole.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 10
    finally:
        R.GRAPH_CACHE = True
    print(f"\nB=64 bf16 step: plain loop with graph cache {t_loop * 1e3:.1f} ms, plain loop eager {t_eager * 1e3:.1f} ms, "
          f"one captured whole-step graph {t_graph * 1e3:.1f} ms")
    assert t_loop <= 1.10 * t_graph + 1e-3, (t_loop, t_graph)


@pytest.mark.parametrize("damped", [False, True])
def test_fused_pose_loss_matches_tensor_expressions(damped):
    """hrp_pose_loss (ten terms + analytic gradient, one launch) against the tensor-expression restatement
    of lib/core/function.py:191-322 with autograd, on random predictions; both branches of the exp(-20 e) damping of the
    translation term (function.py:245-251), a zero mask entry, a prediction that equals its target (norm at 0)."""
    from hrpe_amd.lib.core.function import TERM_NAMES, full_loss, full_loss_expr
    g = torch.Generator(device="cpu").manual_seed(5 + int(damped))
    B, J, P = 37, 7, 8
    r = lambda *s: torch.randn(*s, generator=g)   # noqa: E731
    K = torch.zeros(B, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 300.0 + 200.0 * torch.rand(B, generator=g)
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    K[:, 2, 2] = 1.0
    kp3d = r(B, J, 3) * 0.3 + torch.tensor([0.0, 0.0, 1.2])
    gt = dict(pose=r(B, P), root_rot=r(B, 6), root_trans=kp3d[:, 3].clone(), root_uv=128 + 40 * r(B, 2), kp3d=kp3d,
              kp2d=128 + 60 * r(B, J, 2), mask=(torch.rand(B, J, generator=g) > 0.2).float())
    gt["mask"][0, 3] = 0.0
    pred = [r(B, P), r(B, 6), kp3d[:, 3] + (2.0 if damped else 0.05) * r(B, 3), 128 + 40 * r(B, 2), 1.2 + 0.2 * r(B, 1),
            r(B, J, 3), kp3d + 0.1 * r(B, J, 3), kp3d + 0.1 * r(B, J, 3)]
    pred[6][1, 2] = kp3d[1, 2]       # exactly on target: ||.|| = 0 -> gradient 0
    dev = lambda t: t.to(DEV)        # noqa: E731
    Kd, gtd = dev(K), {k: dev(v) for k, v in gt.items()}
    pa = [dev(t).requires_grad_(True) for t in pred]
    pb = [dev(t).requires_grad_(True) for t in pred]
    la, ta = full_loss(pa, gtd, Kd)
    lb, tb = full_loss_expr(pb, gtd, Kd)
    e_mean = float(torch.norm(pb[2] - gtd["root_trans"], dim=1).mean())
    assert (e_mean > 0.5) == damped
    for n in TERM_NAMES:
        assert abs(float(ta[n]) - float(tb[n])) <= 2e-6 * max(1.0, abs(float(tb[n]))), n
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    (3.0 * la).backward()
    (3.0 * lb).backward()
    for i, (a, b) in enumerate(zip(pa, pb)):
        if i == 5:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0     # uvd does not enter the loss
            continue
        assert torch.allclose(a.grad, b.grad, rtol=2e-5, atol=2e-7 * float(b.grad.abs().max())), i


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("M,K,N", [(64, 2056, 1024), (64, 1024, 8), (5, 2054, 1024), (130, 1024, 6), (64, 1024, 1024)])
def test_linear_kernels_match_torch(M, K, N, det):
    """hrp_linear_fwd / _bwd_data / _bwd_weight (nn.Linear of the regression heads, full_net.py:95-100) against torch in
    fp64; padded pitches, residual, accumulate into existing gradients.  det: with the workspace of the deterministic split
    reduction (what the plans pass) - the same launch twice gives the same bits; without: fp32 atomics."""
    from hrpe_amd import _native as nv
    wsb = int(nv.lib().hrp_linear_workspace_bytes(M, K, N)) if det else 0
    ws = torch.zeros(wsb // 4 + 4, device=DEV) if det else None
    wsp = ws.data_ptr() if det else None
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K + 3, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N + 5, generator=g).to(DEV)
    y = torch.full((M, N + 2), 9.0, device=DEV)
    nv.call("hrp_linear_fwd", x.data_ptr(), K + 3, w.data_ptr(), b.data_ptr(), res.data_ptr(), N + 5, y.data_ptr(), N + 2, M, K, N, wsp, wsb, None)
    if det:
        y2 = torch.full((M, N + 2), 9.0, device=DEV)
        for _ in range(3):
            nv.call("hrp_linear_fwd", x.data_ptr(), K + 3, w.data_ptr(), b.data_ptr(), res.data_ptr(), N + 5, y2.data_ptr(), N + 2, M, K, N, wsp, wsb, None)
            assert torch.equal(y, y2)
    want = (x[:, :K].double() @ w.double().t() + b.double() + res[:, :N].double())
    assert torch.allclose(y[:, :N].double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    assert float((y[:, N:] - 9.0).abs().max()) == 0.0, "columns beyond N are not touched"
    dy = torch.randn(M, N + 2, generator=g).to(DEV)
    for acc in (0, 1):
        dx = torch.full((M, K + 3), 0.5, device=DEV)
        nv.call("hrp_linear_bwd_data", dy.data_ptr(), N + 2, w.data_ptr(), dx.data_ptr(), K + 3, M, K, N, acc, wsp, wsb, None)
        want = dy[:, :N].double() @ w.double() + (0.5 if acc else 0.0)
        assert torch.allclose(dx[:, :K].double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max())), acc
        assert float((dx[:, K:] - 0.5).abs().max()) == 0.0
        dw = torch.full((N, K), 0.25, device=DEV)
        db = torch.full((N,), 0.25, device=DEV)
        nv.call("hrp_linear_bwd_weight", x.data_ptr(), K + 3, dy.data_ptr(), N + 2, dw.data_ptr(), db.data_ptr(), M, K, N, acc, None)
        want_w = dy[:, :N].double().t() @ x[:, :K].double() + (0.25 if acc else 0.0)
        want_b = dy[:, :N].double().sum(0) + (0.25 if acc else 0.0)
        assert torch.allclose(dw.double(), want_w, rtol=1e-5, atol=1e-5 * float(want_w.abs().max())), acc
        assert torch.allclose(db.double(), want_b, rtol=1e-5, atol=1e-5 * float(want_b.abs().max())), acc
