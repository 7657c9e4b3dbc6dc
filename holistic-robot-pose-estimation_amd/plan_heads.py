"""PlanBuilder vocabulary, part 3: the heads behind the trunks (reference lib/models/full_net.py:226-436, depth_net.py:92-137):
nn.Linear, the iterative regressors as one chain of launches, dropout, soft-argmax, pose geometry, forward kinematics and the small
column / row helpers between them."""
import ctypes as C
import math

import torch

from . import _native as nv
from . import plan as PL


class HeadOps:
    """(mixed into plan.PlanBuilder: self.plan, self.bwd_stack and the builder's other methods are the builder's)"""

    def linear(self, x, weight, bias=None, residual=None):
        """y = x W^T + b (+ residual) on fp32 [N, C] tensors: nn.Linear of the regression heads as a skinny GEMM that reads
        the PyTorch-shaped weight directly (hrp_linear_*; no packed copy, no split-K memset + conv launch)."""
        p = self.plan
        x.check_readable()
        assert x.dtype == torch.float32 and x.H == 1 and x.W == 1 and weight.shape[1] == x.C
        assert weight.dim() == 2 or (weight.dim() == 4 and weight.shape[2] == weight.shape[3] == 1)     # (a 1x1 conv on pooled features)
        M, Kf, Nf = x.N, weight.shape[1], weight.shape[0]
        y = p.new(M, 1, 1, Nf, torch.float32)
        y.requires_grad = p.need_grad
        if residual is not None:
            residual.check_readable()
        bp = bias.data_ptr() if bias is not None else None
        # workspace of the deterministic split reduction; forward and data gradient of one layer never overlap
        wsb = int(nv.lib().hrp_linear_workspace_bytes(M, Kf, Nf))
        ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device=p.device)      # (needs no initialisation, include/hrp.h)
        p.keep.append(ws)
        p.fwd.append(lambda s: nv.call("hrp_linear_fwd", x.ptr(), x.pitch, weight.data_ptr(), bp,
                                       residual.ptr() if residual is not None else None, residual.pitch if residual is not None else 0,
                                       y.ptr(), y.pitch, M, Kf, Nf, ws.data_ptr(), wsb, s))
        if p.need_grad:
            def bw():
                if not y.grad_written:
                    return
                if residual is not None and residual.requires_grad:
                    acc = residual.take_grad_slot()
                    p.bwd.append(lambda s: nv.call("hrp_copy_cols", y.gptr(), y.pitch, residual.gptr(), residual.pitch, M, Nf, acc, s))
                want_b = bias is not None and bias.requires_grad
                if weight.requires_grad:
                    gw = p.grad_of_param(weight)
                    gb = p.grad_of_param(bias) if want_b else None
                    first = id(weight) not in p.linear_grad_written
                    p.linear_grad_written.add(id(weight))
                    accw = 1 if (p.grad_arena is not None or not first) else 0
                    p.bwd.append(lambda s: nv.call("hrp_linear_bwd_weight", x.ptr(), x.pitch, y.gptr(), y.pitch, gw.data_ptr(),
                                                   gb.data_ptr() if gb is not None else None, M, Kf, Nf, accw, s))
                elif want_b:     # a frozen weight with a trainable bias: the bias gradient is a column sum of its own
                    gb = p.grad_of_param(bias)
                    first = id(bias) not in p.linear_grad_written
                    p.linear_grad_written.add(id(bias))
                    cwb = int(nv.lib().hrp_colsum_workspace_bytes(M, Nf))
                    cws = torch.empty(cwb // 4 + 4, dtype=torch.float32, device=p.device)
                    p.keep.append(cws)
                    accb = 1 if (p.grad_arena is not None or not first) else 0
                    p.bwd.append(lambda s: nv.call("hrp_colsum", y.gptr(), nv.HRP_F32, M, Nf, y.pitch, gb.data_ptr(), accb, cws.data_ptr(), cwb, s))
                if x.requires_grad:
                    acc = x.take_grad_slot()
                    p.bwd.append(lambda s: nv.call("hrp_linear_bwd_data", y.gptr(), y.pitch, weight.data_ptr(), x.gptr(), x.pitch,
                                                   M, Kf, Nf, acc, ws.data_ptr(), wsb, s))
            self.bwd_stack.append(bw)
        return y

    def regressors(self, xf, heads, n_iter, prob):
        """The iterative regressors (reference lib/models/full_net.py:318-331, 365-378) of one feature tensor as ONE chain of
        launches: p <- p + dec(drop(fc2(drop(fc1(cat(xf, p))))))  n_iter times per head, every head in every launch.
        heads: [(init [M, P] dense fp32 handle, fc1, fc2, dec modules with .weight / .bias)] -> [prediction handles, dense [M, P]].
        Forward 1 (masks) + 1 (hoisted xf product) + n_iter + 1 launches, backward n_iter + 2 (csrc/regressor.hip);
        round 5 ran 169 launches for the same arithmetic.  The sum over the iterations of a layer's weight gradient is one
        product over n_iter * M stacked rows at the end of the chain."""
        p = self.plan
        xf.check_readable()
        assert xf.dtype == torch.float32 and xf.H == 1 and xf.W == 1 and 1 <= len(heads) <= nv.REG_MAX_PROBLEMS
        M, F = xf.N, xf.C
        dev = p.device
        train_drop = p.training and prob > 0.0
        nh = len(heads)
        H = heads[0][2].weight.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        masks = None
        if train_drop:
            masks = torch.zeros(nh * n_iter * 2 * M * H, **f32)
            p.keep.append(masks)
            state = p.rng_state()
            p.n_dropout += 1
            salt = p.n_dropout * 0x9E3779B1 & 0xFFFFFFFF
            p.fwd.append(lambda s: nv.call("hrp_dropout_masks", masks.data_ptr(), masks.numel(), 1.0 - prob, state.data_ptr(), salt, s))

        def mask_ptr(h, i, layer):
            return masks.data_ptr() + 4 * (((h * n_iter + i) * 2 + layer) * M * H) if masks is not None else None

        hs = []
        for init, fc1, fc2, dec in heads:
            init.check_readable()
            P = init.C
            assert init.pitch == P and init.dtype == torch.float32 and 1 <= P <= nv.REG_MAX_P
            assert tuple(fc1.weight.shape) == (H, F + P) and tuple(fc2.weight.shape) == (H, H) and tuple(dec.weight.shape) == (P, H)
            e = dict(P=P, init=init, fc1=fc1, fc2=fc2, dec=dec, ld1=F + P,
                     A=torch.zeros(M * H, **f32), preds=torch.zeros(n_iter * M * P, **f32),
                     d1=torch.zeros(n_iter * M * H, **f32), d2=torch.zeros(n_iter * M * H, **f32))
            e["out"] = p.new(M, 1, 1, P, torch.float32, pitch=P)
            e["out"].requires_grad = p.need_grad
            p.keep += [e["A"], e["preds"], e["d1"], e["d2"]]
            hs.append(e)

        def launch(descs):
            arr = (nv.RegStepDesc * len(descs))(*descs)
            return lambda s: nv.call("hrp_regressor_step", arr, len(descs), s)

        # hoist: A = xf W1[:, :F]^T + b1
        ds = []
        for e in hs:
            d = nv.RegStepDesc()
            d.M, d.P, d.K, d.N = M, 0, F, H
            d.a, d.a_pitch = xf.ptr(), xf.pitch
            d.w, d.w_sn, d.w_sk, d.bias = e["fc1"].weight.data_ptr(), e["ld1"], 1, e["fc1"].bias.data_ptr()
            d.out, d.out_pitch = e["A"].data_ptr(), H
            ds.append(d)
        p.fwd.append(launch(ds))
        for i in range(n_iter):
            ds = []
            for h, e in enumerate(hs):
                P = e["P"]
                d = nv.RegStepDesc()
                d.M, d.P, d.K, d.N = M, P, H, H
                if i == 0:
                    d.u_prev = e["init"].ptr()
                else:
                    d.u_prev, d.u_bias = e["preds"].data_ptr() + 4 * (i - 1) * M * P, e["dec"].bias.data_ptr()
                    d.z, d.z_len, d.z_pitch = e["d2"].data_ptr() + 4 * (i - 1) * M * H, H, H
                    d.zw, d.zw_sk, d.zw_sp = e["dec"].weight.data_ptr(), 1, H
                d.u_out = e["preds"].data_ptr() + 4 * i * M * P
                d.a, d.a_pitch, d.a_mask = e["A"].data_ptr(), H, mask_ptr(h, i, 0)
                d.v, d.v_sk, d.v_sp = e["fc1"].weight.data_ptr() + 4 * F, e["ld1"], 1
                d.a_out = e["d1"].data_ptr() + 4 * i * M * H        # (the operand is staged through it; the backward reads it again)
                d.w, d.w_sn, d.w_sk, d.bias = e["fc2"].weight.data_ptr(), H, 1, e["fc2"].bias.data_ptr()
                d.out_mask = mask_ptr(h, i, 1)
                d.out, d.out_pitch = e["d2"].data_ptr() + 4 * i * M * H, H
                ds.append(d)
            p.fwd.append(launch(ds))
        ds = []
        for e in hs:       # the last state: p_n = p_{n-1} + b3 + d2_{n-1} W3^T
            P = e["P"]
            d = nv.RegStepDesc()
            d.M, d.P, d.K, d.N = M, P, 0, 0
            d.u_prev, d.u_bias = e["preds"].data_ptr() + 4 * (n_iter - 1) * M * P, e["dec"].bias.data_ptr()
            d.z, d.z_len, d.z_pitch = e["d2"].data_ptr() + 4 * (n_iter - 1) * M * H, H, H
            d.zw, d.zw_sk, d.zw_sp = e["dec"].weight.data_ptr(), 1, H
            d.u_out = e["out"].ptr()
            ds.append(d)
        p.fwd.append(launch(ds))
        p.counters["regressor_chains"] = p.counters.get("regressor_chains", 0) + 1
        p.reg_chains.append(dict(masks=masks, heads=hs, n_iter=n_iter, xf=xf))      # (tests and tools read the saved operands here)

        if p.need_grad:
            def bw():
                for e in hs:
                    if not e["out"].grad_written:
                        e["out"].grad_buf()          # (an unused prediction: a zero gradient)
                    e["gs"] = torch.zeros(n_iter * M * e["P"], **f32)          # g_1 .. g_n (g_{i+1} = the gradient of iteration i's update)
                    e["gh2"], e["gh1"] = torch.zeros(n_iter * M * H, **f32), torch.zeros(n_iter * M * H, **f32)
                    e["gA"] = torch.zeros(M * H, **f32)
                    p.keep += [e["gs"], e["gh2"], e["gh1"], e["gA"]]
                for i in range(n_iter - 1, -1, -1):
                    ds = []
                    for h, e in enumerate(hs):
                        P = e["P"]
                        d = nv.RegStepDesc()
                        d.M, d.P, d.K, d.N = M, P, H, H
                        if i == n_iter - 1:
                            d.u_prev = e["out"].gptr()
                        else:        # g_{i+1} = g_{i+2} + gh1_{i+1} W1[:, F:]
                            d.u_prev = e["gs"].data_ptr() + 4 * (i + 1) * M * P
                            d.z, d.z_len, d.z_pitch = e["gh1"].data_ptr() + 4 * (i + 1) * M * H, H, H
                            d.zw, d.zw_sk, d.zw_sp = e["fc1"].weight.data_ptr() + 4 * F, e["ld1"], 1
                        d.u_out = e["gs"].data_ptr() + 4 * i * M * P
                        d.a_mask = mask_ptr(h, i, 1)
                        d.v, d.v_sk, d.v_sp = e["dec"].weight.data_ptr(), 1, H
                        d.a_out = e["gh2"].data_ptr() + 4 * i * M * H
                        d.w, d.w_sn, d.w_sk = e["fc2"].weight.data_ptr(), 1, H
                        d.out_mask = mask_ptr(h, i, 0)
                        d.out, d.out_pitch = e["gh1"].data_ptr() + 4 * i * M * H, H
                        d.out_sum, d.out_sum_accumulate = e["gA"].data_ptr(), 0 if i == n_iter - 1 else 1
                        ds.append(d)
                    p.bwd.append(launch(ds))
                # weight / bias gradients: the n_iter iterations of a layer as one product over n_iter * M stacked rows
                wd = []
                for e in hs:
                    P = e["P"]

                    def prob_(x, xp, dy, dyp, wparam, col0, ld, bparam, rows, K_, N_, acc):
                        g = nv.LinWgradDesc()
                        g.x, g.x_pitch, g.dy, g.dy_pitch = x, xp, dy, dyp
                        g.dw, g.dw_ld = p.grad_of_param(wparam).data_ptr() + 4 * col0, ld
                        g.dbias = p.grad_of_param(bparam).data_ptr() if (bparam is not None and bparam.requires_grad) else None
                        g.M, g.K, g.N, g.accumulate = rows, K_, N_, acc
                        return g

                    def acc_of(t):       # (the arena is zeroed once per backward: every producer accumulates)
                        first = id(t) not in p.linear_grad_written
                        p.linear_grad_written.add(id(t))
                        return 1 if (p.grad_arena is not None or not first) else 0
                    if e["fc2"].weight.requires_grad:
                        wd.append(prob_(e["d1"].data_ptr(), H, e["gh2"].data_ptr(), H, e["fc2"].weight, 0, H, e["fc2"].bias, n_iter * M, H, H,
                                        acc_of(e["fc2"].weight)))
                    if e["dec"].weight.requires_grad:
                        wd.append(prob_(e["d2"].data_ptr(), H, e["gs"].data_ptr(), P, e["dec"].weight, 0, H, e["dec"].bias, n_iter * M, H, P,
                                        acc_of(e["dec"].weight)))
                    if e["fc1"].weight.requires_grad:
                        a1 = acc_of(e["fc1"].weight)       # (two column blocks of one parameter: both are its first writers)
                        wd.append(prob_(xf.ptr(), xf.pitch, e["gA"].data_ptr(), H, e["fc1"].weight, 0, e["ld1"], e["fc1"].bias, M, F, H, a1))
                        wd.append(prob_(e["preds"].data_ptr(), P, e["gh1"].data_ptr(), H, e["fc1"].weight, F, e["ld1"], None, n_iter * M, P, H, a1))
                for k in range(0, len(wd), nv.LIN_WGRAD_MAX):
                    grp = wd[k:k + nv.LIN_WGRAD_MAX]
                    arr = (nv.LinWgradDesc * len(grp))(*grp)
                    p.bwd.append(lambda s, arr=arr, n=len(grp): nv.call("hrp_linear_wgrad_batch", arr, n, s))
                # d xf = sum over the heads of gA W1[:, :F]: one launch, two sources per problem
                if xf.requires_grad:
                    acc = xf.take_grad_slot()
                    for k in range(0, nh, 2):
                        d = nv.RegStepDesc()
                        d.M, d.P, d.K, d.N = M, 0, H, F
                        d.a, d.a_pitch = hs[k]["gA"].data_ptr(), H
                        d.w, d.w_sn, d.w_sk = hs[k]["fc1"].weight.data_ptr(), 1, hs[k]["ld1"]
                        if k + 1 < nh:
                            d.a2, d.a2_pitch = hs[k + 1]["gA"].data_ptr(), H
                            d.w2, d.w2_sk = hs[k + 1]["fc1"].weight.data_ptr(), hs[k + 1]["ld1"]
                        d.out, d.out_pitch, d.out_accumulate = xf.gptr(), xf.pitch, 1 if (acc or k > 0) else 0
                        p.bwd.append(launch([d]))
            self.bwd_stack.append(bw)
        return [e["out"] for e in hs]

    # ---- heads ------------------------------------------------------------------------------------------
    def softargmax(self, heat, J, D, root, fix_root):
        """3-D soft-argmax of NHWC logits [N,H,W,J*D] -> uvd fp32 [N, J*3] (dense)."""
        p = self.plan
        heat.n_readers += 1
        N, H, W = heat.N, heat.H, heat.W
        uvd = p.new(N, 1, 1, J * 3, torch.float32, pitch=J * 3)
        ms = p.new(N, 1, 1, J * 2, torch.float32, pitch=J * 2)
        uvd.requires_grad = p.need_grad and heat.requires_grad
        dt = PL._dt(heat.dtype)
        p.fwd.append(lambda s: nv.call("hrp_softargmax3d_fwd", heat.ptr(), dt, N, J, D, H, W, heat.pitch, root,
                                       1 if fix_root else 0, uvd.ptr(), ms.ptr(), s))
        if p.need_grad:
            def bw():
                if not uvd.grad_written or not heat.requires_grad:
                    return
                assert not heat.grad_written, "heat-map gradient has a single producer"
                heat.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_softargmax3d_bwd", heat.ptr(), dt, N, J, D, H, W, heat.pitch, root,
                                               1 if fix_root else 0, uvd.ptr(), ms.ptr(), uvd.gptr(), heat.gptr(),
                                               heat.pitch, s))
            self.bwd_stack.append(bw)
        return uvd

    def pose_geometry(self, gamma, kval, uvd, Kmat, J, root, image_size, depth_factor):
        """depth = gamma*k/1000; xyz_int = uvd_to_xyz; root_uv; trans = uvz2xyz (all fp32, dense)."""
        p = self.plan
        N = gamma.N
        depth = p.new(N, 1, 1, 1, torch.float32, pitch=1)
        xyz = p.new(N, 1, 1, J * 3, torch.float32, pitch=J * 3)
        ruv = p.new(N, 1, 1, 2, torch.float32, pitch=2)
        trans = p.new(N, 1, 1, 3, torch.float32, pitch=3)
        rg = p.need_grad and (gamma.requires_grad or uvd.requires_grad)
        for t in (depth, xyz, ruv, trans):
            t.requires_grad = rg
        assert gamma.pitch == 1 and kval.pitch == 1 and Kmat.pitch == 9
        p.fwd.append(lambda s: nv.call("hrp_pose_geometry_fwd", gamma.ptr(), kval.ptr(), uvd.ptr(), Kmat.ptr(), N, J, root,
                                       float(image_size), float(depth_factor), depth.ptr(), xyz.ptr(), ruv.ptr(),
                                       trans.ptr(), s))
        if p.need_grad:
            def bw():
                if not rg:
                    return
                gp = [t.gptr() if t.grad_written else None for t in (depth, xyz, ruv, trans)]
                if not any(gp):
                    return
                dg = p.new(N, 1, 1, 1, torch.float32, pitch=1)
                du = p.new(N, 1, 1, J * 3, torch.float32, pitch=J * 3)
                p.bwd.append(lambda s: nv.call("hrp_pose_geometry_bwd", gamma.ptr(), kval.ptr(), uvd.ptr(), Kmat.ptr(), N, J,
                                               root, float(image_size), float(depth_factor), gp[0], gp[1], gp[2], gp[3],
                                               dg.ptr(), du.ptr(), s))
                for src, dst in ((dg, gamma), (du, uvd)):
                    if dst.requires_grad:
                        acc = dst.take_grad_slot()
                        p.bwd.append(lambda s, src=src, dst=dst, acc=acc: nv.call(
                            "hrp_copy_cols", src.ptr(), src.pitch, dst.gptr(), dst.pitch, N, src.C, acc, s))
            self.bwd_stack.append(bw)
        return depth, xyz, ruv, trans

    def fk(self, chain_dev, dof, nkp, q, rot, trans, root, Kmat=None, want_uv=False, want_root_rot=False):
        """Forward kinematics (+projection): q [N,dof], rot [N,6] (two matrix rows) or [N,4] (quaternion), trans [N,3] dense fp32."""
        p = self.plan
        N = q.N
        rd = rot.C
        assert q.pitch == dof and rd in (6, 4) and rot.pitch == rd and trans.pitch == 3
        xyz = p.new(N, 1, 1, nkp * 3, torch.float32, pitch=nkp * 3)
        uv = p.new(N, 1, 1, nkp * 2, torch.float32, pitch=nkp * 2) if want_uv else None
        rr = p.new(N, 1, 1, rd, torch.float32, pitch=rd) if want_root_rot else None
        rg = p.need_grad and (q.requires_grad or rot.requires_grad or trans.requires_grad)
        xyz.requires_grad = rg
        if uv is not None:
            uv.requires_grad = rg
        kp = Kmat.ptr() if Kmat is not None else None
        p.fwd.append(lambda s: nv.call("hrp_fk_project_rot_fwd", chain_dev.data_ptr(), q.ptr(), rot.ptr(), rd, trans.ptr(), kp, N, root,
                                       xyz.ptr(), uv.ptr() if uv is not None else None,
                                       rr.ptr() if rr is not None else None, s))
        if p.need_grad:
            def bw():
                gx = xyz.gptr() if xyz.grad_written else None
                gu = uv.gptr() if (uv is not None and uv.grad_written) else None
                if not rg or (gx is None and gu is None):
                    return
                dq = p.new(N, 1, 1, dof, torch.float32, pitch=dof)
                dr = p.new(N, 1, 1, rd, torch.float32, pitch=rd)
                dtv = p.new(N, 1, 1, 3, torch.float32, pitch=3)
                p.bwd.append(lambda s: nv.call("hrp_fk_project_rot_bwd", chain_dev.data_ptr(), q.ptr(), rot.ptr(), rd, trans.ptr(), kp, N,
                                               root, gx, gu, dq.ptr(), dr.ptr(), dtv.ptr(), s))
                for src, dst in ((dq, q), (dr, rot), (dtv, trans)):
                    if dst.requires_grad:
                        acc = dst.take_grad_slot()
                        p.bwd.append(lambda s, src=src, dst=dst, acc=acc: nv.call(
                            "hrp_copy_cols", src.ptr(), src.pitch, dst.gptr(), dst.pitch, N, src.C, acc, s))
            self.bwd_stack.append(bw)
        return xyz, uv, rr

    def dropout(self, x, prob):
        """Inverted dropout on an fp32 [N, C] tensor: one launch draws the Philox mask (plan seed, per-op salt, a step
        counter the forward advances) and applies it, on the lane's own stream (hrp_dropout_f32)."""
        p = self.plan
        if not p.training or prob <= 0.0:
            return x
        rows, cols = x.N, x.C
        mask = torch.zeros(rows, cols, dtype=torch.float32, device=p.device)
        p.keep.append(mask)
        y = p.new(x.N, 1, 1, x.C, torch.float32, pitch=x.pitch)
        y.requires_grad = x.requires_grad
        keep = 1.0 - prob
        state = p.rng_state()
        p.n_dropout += 1
        salt = p.n_dropout * 0x9E3779B1 & 0xFFFFFFFF
        p.fwd.append(lambda s: nv.call("hrp_dropout_f32", x.ptr(), x.pitch, y.ptr(), y.pitch, mask.data_ptr(), rows, cols, keep,
                                       state.data_ptr(), salt, s))
        if p.need_grad:
            def bw():
                if not y.grad_written or not x.requires_grad:
                    return
                acc = x.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_mul_f32", y.gptr(), y.pitch, mask.data_ptr(), cols, x.gptr(), x.pitch,
                                               rows, cols, acc, s))
            self.bwd_stack.append(bw)
        y.dropout_mask = mask
        return y

    def cat_cols(self, parts, width=None):
        """fp32 [N, sum C_i] = concatenation of [N, C_i] tensors; gradient splits back additively."""
        p = self.plan
        N = parts[0].N
        total = sum(t.C for t in parts)
        out = p.new(N, 1, 1, width or total, torch.float32)
        out.requires_grad = p.need_grad and any(t.requires_grad for t in parts)
        offs, c = [], 0
        for t in parts:
            offs.append(c)
            c += t.C
        for t, o in zip(parts, offs):
            p.fwd.append(lambda s, t=t, o=o: nv.call("hrp_copy_cols", t.ptr(), t.pitch, out.ptr() + 4 * o, out.pitch, N, t.C, 0, s))
        if p.need_grad:
            def bw():
                if not out.grad_written:
                    return
                for t, o in zip(parts, offs):
                    if t.requires_grad:
                        acc = t.take_grad_slot()
                        p.bwd.append(lambda s, t=t, o=o, acc=acc: nv.call(
                            "hrp_copy_cols", out.gptr() + 4 * o, out.pitch, t.gptr(), t.pitch, N, t.C, acc, s))
            self.bwd_stack.append(bw)
        return out

    def dense(self, x):
        """fp32 copy with pitch == C (layout the head kernels expect)."""
        p = self.plan
        if x.pitch == x.C:
            return x
        out = p.new(x.N, 1, 1, x.C, torch.float32, pitch=x.C)
        out.requires_grad = x.requires_grad
        self.copy_cols(x, out)
        return out

    def softargmax_flat(self, heat, J):
        """HeatmapIntegralJoint's core (integral.py:206-224): per channel softmax over the H*W positions of NHWC logits
        [N, H, W, J] -> E[flat index] / (H*W), fp32 [N, J] dense."""
        p = self.plan
        heat.n_readers += 1
        N, HW = heat.N, heat.H * heat.W
        coord = p.new(N, 1, 1, J, torch.float32, pitch=J)
        ms = p.new(N, 1, 1, J * 2, torch.float32, pitch=J * 2)
        coord.requires_grad = p.need_grad and heat.requires_grad
        dt = PL._dt(heat.dtype)
        p.fwd.append(lambda s: nv.call("hrp_softargmax_flat_fwd", heat.ptr(), dt, N, J, HW, heat.pitch, coord.ptr(), ms.ptr(), s))
        if p.need_grad:
            def bw():
                if not coord.grad_written or not heat.requires_grad:
                    return
                assert not heat.grad_written, "joint-map gradient has a single producer"
                heat.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_softargmax_flat_bwd", heat.ptr(), dt, N, J, HW, heat.pitch, coord.ptr(), ms.ptr(),
                                               coord.gptr(), heat.gptr(), heat.pitch, s))
            self.bwd_stack.append(bw)
        return coord

    def rot6d_compose(self, a, b):
        """out = rotmat_to_rot6d(R(a) @ R(b)) on dense fp32 [N, 6] tensors (full_net.py:362)."""
        p = self.plan
        assert a.C == 6 and b.C == 6 and a.pitch == 6 and b.pitch == 6
        N = a.N
        out = p.new(N, 1, 1, 6, torch.float32, pitch=6)
        out.requires_grad = p.need_grad and (a.requires_grad or b.requires_grad)
        p.fwd.append(lambda s: nv.call("hrp_rot6d_compose_fwd", a.ptr(), b.ptr(), out.ptr(), N, s))
        if p.need_grad:
            def bw():
                if not out.grad_written or not out.requires_grad:
                    return
                acc_a = a.take_grad_slot() if a.requires_grad else 0
                acc_b = b.take_grad_slot() if b.requires_grad else 0
                p.bwd.append(lambda s: nv.call("hrp_rot6d_compose_bwd", a.ptr(), b.ptr(), out.gptr(),
                                               a.gptr() if a.requires_grad else None, b.gptr() if b.requires_grad else None,
                                               N, acc_a, acc_b, s))
            self.bwd_stack.append(bw)
        return out

    def row_scale(self, x, kvec, into=None):
        """y[n, c] = x[n, c] * k[n, c]  (fp32; C == 1: depth = gamma * k_value, C > 1: the multi_kp depths); into: y += x * k
        on an existing row_scale result (depth += 1000 * offset, depth_net.py:127-131)."""
        p = self.plan
        Cc = x.C
        assert kvec.C == Cc
        y = into if into is not None else p.new(x.N, 1, 1, Cc, torch.float32, pitch=Cc)
        facc = 1 if into is not None else 0
        y.requires_grad = x.requires_grad or (into is not None and into.requires_grad)
        p.fwd.append(lambda s: nv.call("hrp_mul_f32", x.ptr(), x.pitch, kvec.ptr(), kvec.pitch, y.ptr(), y.pitch, x.N, Cc, facc, s))
        if p.need_grad:
            def bw():
                if not y.grad_written or not x.requires_grad:
                    return
                acc = x.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_mul_f32", y.gptr(), y.pitch, kvec.ptr(), kvec.pitch, x.gptr(), x.pitch, x.N, Cc,
                                               acc, s))
            self.bwd_stack.append(bw)
        return y
