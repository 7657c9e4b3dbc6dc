"""GPU: the fused iterative regressors (csrc/regressor.hip, PlanBuilder.regressors; reference lib/models/full_net.py:318-331,
365-378) - the step kernel against torch in float64 in every option it has, the stacked weight-gradient launch, the mask
launch, and the whole chain inside the full network against the one-launch-per-layer path and against torch autograd
through the SAME dropout masks."""
import ctypes as C

import pytest
import torch

from test_gpu_model import DEV, build_full
from synth import synth_inputs

pytestmark = pytest.mark.gpu


def _close(got, want, tol=2e-5):
    want = want.double()
    scale = float(want.abs().max()) + 1e-30
    err = float((got.double() - want).abs().max()) / scale
    assert err < tol, err


@pytest.mark.parametrize("M,P,K,N,wt", [(64, 8, 1024, 1024, False), (64, 6, 1024, 1024, True), (2, 8, 256, 48, False),
                                         (70, 15, 384, 40, True), (5, 4, 128, 16, False), (64, 0, 2048, 1024, False)])
def test_regressor_step_all_options(M, P, K, N, wt):
    """One step with everything switched on: state update from (z, zw) in both stride orders, rank-P update + mask while staging,
    the product in both weight layouts with an odd leading dimension (scalar loads), bias, output mask, accumulate, out_sum,
    the saved operand and state - against float64; repeated launches are bit-identical; rows / columns beyond the problem are
    not touched."""
    from hrpe_amd import _native as nv
    g = torch.Generator(device="cpu").manual_seed(M * 7 + P * 3 + K + N + wt)
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)   # noqa: E731
    zl = 64
    ld_w = (N + 3) if wt else (K + 2)                       # leading dimension of the stored weight: not a multiple of 4
    w = rn(K, ld_w) if wt else rn(N, ld_w)
    w = w / K ** 0.5
    a, amask = rn(M, K + 4), (torch.rand(M, K, generator=g) < 0.5).float().to(DEV) * 2
    omask = (torch.rand(M, N, generator=g) < 0.5).float().to(DEV) * 2
    bias = rn(N)
    d = nv.RegStepDesc()
    d.M, d.P, d.K, d.N = M, P, K, N
    keep = []
    if P:
        u_prev, u_bias, z = rn(M, P), rn(P), rn(M, zl + 4)
        zw_kp = rn(zl, P + 1) if wt else None             # zw[k][p] rows (stride order of the backward)
        zw_pk = rn(P, zl) if not wt else None             # zw[p][k] (forward: dec.weight)
        v_kp = rn(K, P + 3) if not wt else None           # v[k][p] (forward: fc1.weight[:, F:])
        v_pk = rn(P, K) if wt else None                   # v[p][k] (backward: dec.weight^T)
        u_out = torch.full((M + 1, P), 7.0, device=DEV)
        d.u_prev, d.u_bias, d.z, d.z_len, d.z_pitch = u_prev.data_ptr(), u_bias.data_ptr(), z.data_ptr(), zl, zl + 4
        if wt:
            d.zw, d.zw_sk, d.zw_sp = zw_kp.data_ptr(), P + 1, 1
            d.v, d.v_sk, d.v_sp = v_pk.data_ptr(), 1, K
            zw, v = zw_kp[:, :P].double(), v_pk.double().t()
        else:
            d.zw, d.zw_sk, d.zw_sp = zw_pk.data_ptr(), 1, zl
            d.v, d.v_sk, d.v_sp = v_kp.data_ptr(), P + 3, 1
            zw, v = zw_pk.double().t(), v_kp[:, :P].double()
        d.u_out = u_out.data_ptr()
        keep += [u_prev, u_bias, z, zw_kp, zw_pk, v_kp, v_pk]
        u_want = u_prev.double() + u_bias.double() + z[:, :zl].double() @ zw
    a_out = torch.full((M + 1, K), 7.0, device=DEV)
    d.a, d.a_pitch, d.a_mask, d.a_out = a.data_ptr(), K + 4, amask.data_ptr(), a_out.data_ptr()
    d.w, d.bias, d.out_mask = w.data_ptr(), bias.data_ptr(), omask.data_ptr()
    d.w_sn, d.w_sk = (1, ld_w) if wt else (ld_w, 1)
    out = torch.full((M + 1, N + 5), 0.5, device=DEV)
    osum = torch.full((M + 1, N), 0.25, device=DEV)
    d.out, d.out_pitch, d.out_accumulate, d.out_sum, d.out_sum_accumulate = out.data_ptr(), N + 5, 1, osum.data_ptr(), 1
    arr = (nv.RegStepDesc * 1)(d)
    nv.call("hrp_regressor_step", arr, 1, None)
    torch.cuda.synchronize()
    ap = a[:, :K].double()
    if P:
        ap = ap + u_want @ v.t()
        _close(u_out[:M], u_want)
        assert float((u_out[M:] - 7.0).abs().max()) == 0.0
    ap = ap * amask.double()
    _close(a_out[:M], ap)
    W = w[:, :N].double().t() if wt else w[:, :K].double()       # [N][K]
    val = (ap @ W.t() + bias.double()) * omask.double()
    _close(out[:M, :N] - 0.5, val, 5e-5)
    _close(osum[:M] - 0.25, val, 5e-5)
    assert float((out[M:] - 0.5).abs().max()) == 0.0 and float((out[:, N:] - 0.5).abs().max()) == 0.0
    assert float((a_out[M:] - 7.0).abs().max()) == 0.0
    first = out.clone()
    for _ in range(3):
        out.fill_(0.5)
        osum.fill_(0.25)
        nv.call("hrp_regressor_step", arr, 1, None)
        assert torch.equal(out, first), "fixed summation order: bit-identical repeats"


def test_regressor_step_two_problems_two_sources():
    """Two problems of different state width in one launch (the two heads), a state-only problem (N = 0: the last prediction),
    and the two-source product of d xf (different leading dimensions of the two weights)."""
    from hrpe_amd import _native as nv
    g = torch.Generator(device="cpu").manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)   # noqa: E731
    M, H, F = 64, 256, 512
    ds, want, outs = [], [], []
    keep = []
    for P in (8, 6):
        u_prev, u_bias, z, zw = rn(M, P), rn(P), rn(M, H), rn(P, H)
        u_out = torch.zeros(M, P, device=DEV)
        d = nv.RegStepDesc()
        d.M, d.P, d.K, d.N = M, P, 0, 0
        d.u_prev, d.u_bias, d.z, d.z_len, d.z_pitch = u_prev.data_ptr(), u_bias.data_ptr(), z.data_ptr(), H, H
        d.zw, d.zw_sk, d.zw_sp, d.u_out = zw.data_ptr(), 1, H, u_out.data_ptr()
        ds.append(d)
        keep += [u_prev, u_bias, z, zw]
        want.append(u_prev.double() + u_bias.double() + z.double() @ zw.double().t())
        outs.append(u_out)
    nv.call("hrp_regressor_step", (nv.RegStepDesc * 2)(*ds), 2, None)
    torch.cuda.synchronize()
    for o, w_ in zip(outs, want):
        _close(o, w_)
    gA1, gA2 = rn(M, H), rn(M, H)
    W1, W2 = rn(H, F + 8) / H ** 0.5, rn(H, F + 6) / H ** 0.5
    dx = torch.full((M, F), 0.5, device=DEV)
    d = nv.RegStepDesc()
    d.M, d.P, d.K, d.N = M, 0, H, F
    d.a, d.a_pitch, d.w, d.w_sn, d.w_sk = gA1.data_ptr(), H, W1.data_ptr(), 1, F + 8
    d.a2, d.a2_pitch, d.w2, d.w2_sk = gA2.data_ptr(), H, W2.data_ptr(), F + 6
    d.out, d.out_pitch, d.out_accumulate = dx.data_ptr(), F, 1
    nv.call("hrp_regressor_step", (nv.RegStepDesc * 1)(d), 1, None)
    torch.cuda.synchronize()
    _close(dx - 0.5, gA1.double() @ W1[:, :F].double() + gA2.double() @ W2[:, :F].double(), 5e-5)


def test_linear_wgrad_batch_matches_torch():
    """Four problems of one launch (a 1024 x 1024 weight over 256 stacked rows with its bias, a column block of a wider
    parameter, an 8-row weight, a problem with ragged edges) against float64, overwrite and accumulate."""
    from hrpe_amd import _native as nv
    g = torch.Generator(device="cpu").manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)   # noqa: E731
    probs = [(256, 1024, 1024, 1024, 0, True), (64, 300, 200, 300 + 8, 0, True), (256, 8, 200, 308, 300, False), (256, 1024, 8, 1024, 0, True),
             (37, 70, 33, 70, 0, False)]
    for acc in (0, 1):
        ds, chk, keep = [], [], []
        for (M, K, N, ld, col0, wb) in probs:
            x, dy = rn(M, K + 2), rn(M, N + 1)
            dw = torch.full((N, ld), 0.25, device=DEV)
            db = torch.full((N,), 0.25, device=DEV) if wb else None
            d = nv.LinWgradDesc()
            d.x, d.x_pitch, d.dy, d.dy_pitch = x.data_ptr(), K + 2, dy.data_ptr(), N + 1
            d.dw, d.dw_ld, d.dbias = dw.data_ptr() + 4 * col0, ld, db.data_ptr() if wb else None
            d.M, d.K, d.N, d.accumulate = M, K, N, acc
            ds.append(d)
            keep += [x, dy]
            chk.append((dw, db, dy[:, :N].double().t() @ x[:, :K].double(), dy[:, :N].double().sum(0), col0, K))
        nv.call("hrp_linear_wgrad_batch", (nv.LinWgradDesc * len(ds))(*ds), len(ds), None)
        torch.cuda.synchronize()
        for dw, db, ww, wb_, col0, K in chk:
            base = 0.25 if acc else 0.0
            _close(dw[:, col0:col0 + K] - base, ww, 2e-5)
            rest = torch.cat([dw[:, :col0], dw[:, col0 + K:]], 1)
            assert rest.numel() == 0 or float((rest - 0.25).abs().max()) == 0.0, "columns outside the block are not touched"
            if db is not None:
                _close(db - base, wb_, 2e-5)


def test_dropout_masks_are_the_per_op_generator():
    """hrp_dropout_masks draws what hrp_dropout_f32 draws for the same (seed, salt, step): one generator, one counter layout."""
    from hrpe_amd import _native as nv
    n = 64 * 1024 + 8
    state = torch.tensor([1234567, 3], dtype=torch.int64, device=DEV)
    m1 = torch.zeros(n, device=DEV)
    nv.call("hrp_dropout_masks", m1.data_ptr(), n, 0.5, state.data_ptr(), 99, None)
    x = torch.ones(n, device=DEV)
    y, m2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    nv.call("hrp_dropout_f32", x.data_ptr(), n, y.data_ptr(), n, m2.data_ptr(), 1, n, 0.5, state.data_ptr(), 99, None)
    torch.cuda.synchronize()
    assert torch.equal(m1, m2)
    assert set(torch.unique(m1).tolist()) == {0.0, 2.0} and 0.49 < float((m1 > 0).float().mean()) < 0.51
    nv.call("hrp_rng_advance", state.data_ptr(), None)
    m3 = torch.zeros(n, device=DEV)
    nv.call("hrp_dropout_masks", m3.data_ptr(), n, 0.5, state.data_ptr(), 99, None)
    assert not torch.equal(m1, m3)


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def test_fused_chain_equals_the_layerwise_path():
    """The full network with the fused chain against the same network with one launch per layer (full_net.FUSED_REGRESSORS =
    False), fp32, no dropout: the 8-tuple of an eval forward and of a training step, and every gradient of the heads and of the
    trunk that feeds them (summation orders differ: 1e-5 of the tensor's scale on outputs, 2e-4 on gradients)."""
    from hrpe_amd.lib.models import full_net as FN
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(4)]
    res = {}
    for fused in (True, False):
        FN.FUSED_REGRESSORS = fused
        try:
            torch.manual_seed(0)
            m = build_full(p_dropout=0.0).set_compute_dtype(torch.float32)
            m.eval()
            with torch.no_grad():
                ev = [o.clone() for o in m(x_reg, x_root, kv, K)]
            m.train()
            m.zero_grad()
            out = m(x_reg, x_root, kv, K)
            sum((o.float() * (1 + i)).square().mean() for i, o in enumerate(out)).backward()
            torch.cuda.synchronize()
            res[fused] = (ev, [o.detach().clone() for o in out], _grads(m))
        finally:
            FN.FUSED_REGRESSORS = True
    for a, b in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        _close(a, b, 2e-5)
    ga, gb = res[True][2], res[False][2]
    assert set(ga) == set(gb)
    # (a conv bias in front of a BatchNorm has an analytically zero gradient: pure rounding noise, left out by the floor)
    floor = 1e-4 * max(float(v.abs().max()) for v in gb.values())
    worst, where = 0.0, None
    for k in ga:
        if k.startswith(("fc_", "dec")) or "final_feat_layer" in k or "stage4.2.fuse_layers.0" in k:
            scale = float(gb[k].abs().max())
            if scale < floor:
                continue
            e = float((ga[k] - gb[k]).abs().max()) / scale
            if e > worst:
                worst, where = e, k
    assert worst < 2e-4, (worst, where)


def test_fused_chain_with_dropout_against_torch_autograd():
    """p_dropout = 0.5: the chain's predictions and the gradients it produces (both heads' weights and biases, the feature) against
    torch autograd in float64 through the SAME masks, read back from the plan; a second step draws other masks; two models with
    one seed repeat bit for bit."""
    from hrpe_amd import plan as P
    m = build_full(p_dropout=0.5).set_compute_dtype(torch.float32).train()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(4)]
    seen = {}
    orig = P.PlanBuilder.regressors

    def spy(self, xf, heads, n_iter, prob):
        outs = orig(self, xf, heads, n_iter, prob)
        seen["xf"], seen["heads"], seen["outs"], seen["plan"], seen["n_iter"] = xf, heads, outs, self.plan, n_iter
        return outs
    P.PlanBuilder.regressors = spy
    try:
        masks_by_step = []
        for step in range(2):
            m.zero_grad()
            out = m(x_reg, x_root, kv, K)
            sum((o.float() * (1 + i)).square().mean() for i, o in enumerate(out)).backward()
            torch.cuda.synchronize()
            xf, heads, outs, n_iter = seen["xf"], seen["heads"], seen["outs"], seen["n_iter"]
            M, F, H = xf.N, xf.C, 1024
            masks = seen["plan"].reg_chains[0]["masks"].view(2, n_iter, 2, M, H).double()
            masks_by_step.append(masks.clone())
            assert set(torch.unique(masks).tolist()) == {0.0, 2.0} and 0.47 < float((masks > 0).double().mean()) < 0.53
            xf_t = xf.buf.view(M, xf.pitch)[:, :F].double().detach().requires_grad_(True)
            params, preds = [], []
            for h, (init, fc1, fc2, dec) in enumerate(heads):
                ws = [t.detach().double().requires_grad_(True) for t in (fc1.weight, fc1.bias, fc2.weight, fc2.bias, dec.weight, dec.bias)]
                pr = init.buf.view(M, init.pitch)[:, :init.C].double()
                for i in range(n_iter):
                    h1 = (torch.cat([xf_t, pr], 1) @ ws[0].t() + ws[1]) * masks[h, i, 0]
                    h2 = (h1 @ ws[2].t() + ws[3]) * masks[h, i, 1]
                    pr = pr + h2 @ ws[4].t() + ws[5]
                params.append(ws)
                preds.append(pr)
            loss = 0.0
            for pr, o in zip(preds, outs):
                _close(o.buf.view(M, o.pitch)[:, :o.C], pr, 2e-5)
                loss = loss + (pr * o.grad_buf().view(M, o.pitch)[:, :o.C].double()).sum()
            loss.backward()
            for ws, (init, fc1, fc2, dec) in zip(params, heads):
                for t64, t in zip(ws, (fc1.weight, fc1.bias, fc2.weight, fc2.bias, dec.weight, dec.bias)):
                    _close(t.grad, t64.grad, 1e-4)
            # d xf: the chain is the only consumer of the pooled feature
            _close(xf.grad_buf().view(M, xf.pitch)[:, :F], xf_t.grad, 1e-4)
        assert not torch.equal(masks_by_step[0], masks_by_step[1]), "a new mask every step"
    finally:
        P.PlanBuilder.regressors = orig


def test_fused_chain_launch_count():
    """the regressors cost <= 20 launches per training step (169 + 41 folds in round 5)."""
    from hrpe_amd import _native as nv
    m = build_full(p_dropout=0.5).train()
    x_reg, x_root, kv, K = [t.to(DEV) for t in synth_inputs(2)]
    out = m(x_reg, x_root, kv, K)
    sum(o.float().square().mean() for o in out).backward()      # builds the plan
    counts = {}
    names = ("hrp_regressor_step", "hrp_linear_wgrad_batch", "hrp_dropout_masks", "hrp_linear_fwd", "hrp_linear_bwd_data", "hrp_linear_bwd_weight",
             "hrp_dropout_f32", "hrp_mul_f32", "hrp_colsum")
    nv.set_profile_hook(lambda name, args, launch: (counts.__setitem__(name, counts.get(name, 0) + 1), launch()))
    try:
        m.zero_grad()
        out = m(x_reg, x_root, kv, K)
        sum(o.float().square().mean() for o in out).backward()
        torch.cuda.synchronize()
    finally:
        nv.set_profile_hook(None)
    head = {k: counts.get(k, 0) for k in names}
    assert head["hrp_regressor_step"] == 1 + 4 + 1 + 4 + 1 and head["hrp_linear_wgrad_batch"] == 1 and head["hrp_dropout_masks"] == 1
    assert head["hrp_dropout_f32"] == 0 and head["hrp_mul_f32"] == 0 and head["hrp_linear_bwd_weight"] <= 1      # (depth_layer keeps the skinny GEMM)
    # the column sums left are real bias gradients (final_layer of the heat-map head); the biases in front of a train-mode BatchNorm
    # have an identically zero gradient and no launch (PlanBuilder._conv_bwd)
    assert head["hrp_colsum"] <= 1
    assert sum(head.values()) <= 20, head


def test_copy_cols_batch_copies_accumulates_and_zero_fills():
    """hrp_copy_cols_batch: 18 problems (two launches) of different shapes and pitches - plain copies, accumulating copies, zero fills -
    against torch indexing; the columns beyond every block keep their contents."""
    from hrpe_amd import _native as nv
    g = torch.Generator(device="cpu").manual_seed(5)
    items, checks, keep = [], [], []
    for i in range(18):
        rows, cols = 1 + (7 * i) % 64, 1 + (5 * i) % 23
        sp, dp = cols + i % 3, cols + (i + 1) % 4
        src = torch.randn(rows, sp, generator=g).to(DEV)
        dst = torch.randn(rows, dp, generator=g).to(DEV)
        want = dst.clone()
        mode = i % 3
        if mode == 0:
            want[:, :cols] = src[:, :cols]
        elif mode == 1:
            want[:, :cols] += src[:, :cols]
        else:
            want[:, :cols] = 0
        items.append((None if mode == 2 else src.data_ptr(), sp, dst.data_ptr(), dp, rows, cols, 1 if mode == 1 else 0))
        checks.append((dst, want))
        keep.append(src)
    nv.copy_cols_batch(items, None)
    torch.cuda.synchronize()
    for dst, want in checks:
        assert torch.equal(dst, want)
