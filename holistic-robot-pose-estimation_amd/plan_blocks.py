"""PlanBuilder vocabulary, part 2: the fused residual-block forms (reference lib/models/backbones/HRnet.py:28-98).

basic_block_eval    inference BasicBlock as ONE launch (csrc/conv_block.h)
conv_bn_relu_conv   train-mode conv -> BN -> ReLU -> conv on the row-strip kernels, with the BatchNorm passes of the block in the
                    convolutions' prologues / epilogues (csrc/conv_row.h)
bottleneck_tail     conv3 (+ projection) + BN + shortcut + ReLU of a wide Bottleneck without storing a raw 1x1 output (csrc/conv_pw.h)

The switches (ROWCONV_FUSE, BLOCK_END_*, BNECK_TAIL_FUSE ..) live in the plan module, where tests and tools patch them."""
import ctypes as C
import math

import torch

from . import _native as nv
from . import plan as PL


class BlockOps:
    """(mixed into plan.PlanBuilder: self.plan, self.bwd_stack and the builder's other methods are the builder's)"""

    def basic_block_eval(self, x, conv1_w, bn1, conv2_w, bn2):
        """out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), a whole BasicBlock without downsample (reference HRnet.py:41-57) of an
        INFERENCE plan as ONE launch (csrc/conv_block.h): BatchNorm folded to scale / shift, the intermediate stays in LDS.
        -> out, or None when the shapes are not the fused kernel's (32 / 64 channels at 64 / 32 pixels per row, bf16; caller:
        the general path - two convolutions with folded epilogues)."""
        p = self.plan
        if not (PL.BLOCK_FUSE and self.fuse_inference and not p.training and not p.need_grad and x.dtype == torch.bfloat16):
            return None
        Cc = x.C
        if Cc not in (32, 64) or tuple(conv1_w.shape) != (Cc, Cc, 3, 3) or tuple(conv2_w.shape) != (Cc, Cc, 3, 3) or x.pitch != Cc or x.offset:
            return None
        x.check_readable()
        dtype = x.dtype
        w1, w2 = p.weight(conv1_w, Cc, Cc, 9), p.weight(conv2_w, Cc, Cc, 9)
        out = p.new(x.N, x.H, x.W, Cc, dtype)
        b = nv.BlockDesc()
        d1 = self._conv_desc(x, w1, out, 1, 3, dtype, into=b.conv1)
        d2 = self._conv_desc(x, w2, out, 1, 3, dtype, into=b.conv2)
        (sc1, sh1), (sc2, sh2) = self._fold(bn1), self._fold(bn2)
        d1.scale, d1.shift, d1.relu, d1.y = sc1.data_ptr(), sh1.data_ptr(), 1, None
        d2.scale, d2.shift, d2.relu, d2.x = sc2.data_ptr(), sh2.data_ptr(), 1, None
        d2.res, d2.res_pitch = x.ptr(), x.pitch
        d1.w = d2.w = x.ptr()                      # (placeholders for the host-side shape query; final in late())
        if nv.lib().hrp_block_channels(C.byref(b)) != Cc:
            return None
        for w in (w1, w2):
            w.dtype, w.cin_used = dtype, Cc
            w.need_t = getattr(w, "need_t", False)

        def late():
            d1.w, d2.w = w1.arena.data_ptr() + w1.fwd_off * 2, w2.arena.data_ptr() + w2.fwd_off * 2
        p.late(late)
        p.fwd.append(PL.BlockLaunch(b))
        out.producer = None
        p.counters["block_fused"] = p.counters.get("block_fused", 0) + 1
        return out

    def conv_bn_relu_conv(self, x, conv1_w, bn1, conv2_w, bn2=None):
        """y2 = conv2(relu(bn1(conv1(x)))), the interior of a BasicBlock (reference HRnet.py:41-50), in a TRAINING plan on the
        row-strip kernel: conv1 as usual (statistics in its epilogue), conv2 with the BatchNorm + ReLU applied while its
        input rows are staged (the activation leaves as a side output, the operand of conv2's weight gradient) - no
        hrp_ew_fwd pass.  Backward: conv2's data gradient accumulates the BatchNorm-backward sums in its epilogue, conv1's
        data gradient applies the BatchNorm + ReLU backward while IT stages (side output: the gradient of conv1's output,
        the operand of conv1's weight gradient) - no hrp_ew_bwd_reduce / hrp_ew_bwd_apply passes.
        bn2 given: the whole BasicBlock, out = relu(bn2(y2) + x) (HRnet.py:52-56).  The block-end activation keeps its forward
        pass (hrp_ew_fwd, with the ReLU bit mask); its BACKWARD apply pass moves into conv2's data gradient (pro_mode 2 with the
        bit mask: side output = y2.grad for conv2's weight gradient, second side output = the masked gradient for the
        residual), and its reduce pass into the epilogue of the NEXT block's conv1 data gradient when that launch is the
        last producer of out.grad (the blocks of a branch stack) - no hrp_ew_bwd_apply and mostly no hrp_ew_bwd_reduce.
        -> y2 (raw, with statistics) or, with bn2, out; None when the shapes / mode are not the row-strip kernel's (caller:
        general path)."""
        p = self.plan
        if not (PL.ROWCONV_FUSE and p.training and x.dtype == torch.bfloat16):
            return None
        if not (self.bn_batch_stats(bn1) and (bn2 is None or self.bn_batch_stats(bn2))):
            return None              # (a BatchNorm in eval mode inside a training plan: general path)
        Cc = x.C
        if tuple(conv1_w.shape) != (Cc, Cc, 3, 3) or tuple(conv2_w.shape) != (Cc, Cc, 3, 3) or x.pitch != Cc or x.offset:
            return None
        if p.need_grad and not x.requires_grad:
            return None
        # x = the previous block's output whose forward pass is still held back: this block's conv1 applies it while staging
        pend = getattr(x, "pending_block_end", None)
        if pend is not None and not (PL.BLOCK_END_FWD_FUSE and x.lane_path == p.lane_path):
            pend = None
        if pend is not None:
            x.pending_block_end = None
            p.pending_block_ends.remove(x)
        x.check_readable()
        dtype = x.dtype
        w1 = p.weight(conv1_w, Cc, Cc, 9)
        probe = self._conv_desc(x, w1, x, 1, 3, dtype)       # geometry only; dummy aligned pointers
        probe.w = probe.x
        probe.pro_mode, probe.pro_stats, probe.pro_gamma, probe.pro_beta = 1, probe.x, probe.x, probe.x
        if nv.lib().hrp_conv_rowstrip_channels(C.byref(probe)) != Cc:
            if pend is not None:          # (not reachable: the previous block had this shape; keep the plan correct anyway)
                x.pending_block_end = pend
                p.pending_block_ends.append(x)
                x.materialize()
            return None
        w2 = p.weight(conv2_w, Cc, Cc, 9)
        for w in (w1, w2):
            w.dtype, w.cin_used = dtype, Cc
            w.need_t = getattr(w, "need_t", False) or p.need_grad
        esz = 2
        cnt = float(x.N * x.H * x.W)
        y1, h, y2 = p.new(x.N, x.H, x.W, Cc, dtype), p.new(x.N, x.H, x.W, Cc, dtype), p.new(x.N, x.H, x.W, Cc, dtype)
        for t in (y1, h, y2):
            t.requires_grad = p.need_grad
        y1.stats, y2.stats = p.alloc_stats(Cc), p.alloc_stats(Cc)
        p.bn_train.append((bn1, y1.stats, x.N * x.H * x.W))
        gam, bet = bn1.weight.data_ptr(), bn1.bias.data_ptr()
        d1 = self._conv_desc(x, w1, y1, 1, 3, dtype)
        d2 = self._conv_desc(y1, w2, y2, 1, 3, dtype)
        d2.pro_mode, d2.pro_gamma, d2.pro_beta, d2.pro_count, d2.pro_eps, d2.pro_side = 1, gam, bet, cnt, bn1.eps, h.ptr()
        if pend is not None:
            # conv1 stages the previous block's raw conv2 output and turns it into x = relu(bn2'(y2') + x') in place; x and its
            # ReLU bits (what the previous block's backward reads) leave as side outputs
            py2, px, pbn = pend["y2"], pend["x"], pend["bn"]
            d1.x, d1.pro_mode, d1.pro_x2 = py2.ptr(), 3, px.ptr()
            d1.pro_gamma, d1.pro_beta, d1.pro_count, d1.pro_eps = pbn.weight.data_ptr(), pbn.bias.data_ptr(), cnt, pbn.eps
            d1.pro_side, d1.pro_mask = x.ptr(), pend["mask"]
            p.counters["block_end_forward_fused"] = p.counters.get("block_end_forward_fused", 0) + 1

        def late():
            d1.w, d2.w = w1.arena.data_ptr() + w1.fwd_off * esz, w2.arena.data_ptr() + w2.fwd_off * esz
            d1.stats, d2.stats = p.stats.data_ptr() + 8 * y1.stats, p.stats.data_ptr() + 8 * y2.stats
            d2.pro_stats = d1.stats
            if pend is not None:
                d1.pro_stats = p.stats.data_ptr() + 8 * pend["y2"].stats
        p.late(late)
        p.fwd.append(PL.Launch("conv", d1))
        p.fwd.append(PL.Launch("conv", d2))
        y1.producer, y2.producer = None, ("conv", d2)
        p.counters["rowconv_fused_blocks"] = p.counters.get("rowconv_fused_blocks", 0) + 1
        out, fd = None, None
        if bn2 is not None:
            out = self.act([PL.Term(y2, bn2), PL.Term(x)], relu=True)
            fd = out.ew_desc
            act_bw = list.pop(self.bwd_stack) if p.need_grad else None      # the activation's own backward
            if PL.BLOCK_END_FWD_FUSE and p.need_grad and fd.mask:
                # hold the forward pass back: the next block of the stack runs it inside its conv1 (pro_mode 3); any other reader
                # (a fuse layer, the head, a plan output) makes it run in front of itself (TensorH.materialize)
                ent = list.pop(p.fwd)
                assert isinstance(ent.op, PL.Launch) and ent.op.desc is fd
                out.pending_block_end = dict(launch=ent.op, y2=y2, x=x, bn=bn2, mask=fd.mask)
                p.pending_block_ends.append(out)
            if not (p.need_grad and fd.mask and PL.BLOCK_END_FUSE):
                fd = None                         # ... stays (hrp_ew_bwd_reduce + hrp_ew_bwd_apply), re-pushed behind bw below
        if p.need_grad:
            def bw():
                if fd is not None:
                    if not out.grad_written:
                        return
                elif not y2.grad_written:
                    return
                # the identity shortcut's gradient (out.grad under the block-end mask): nobody has written x.grad yet -> conv1's data
                # gradient below adds it as a MASKED residual and writes x.grad once; else conv2's data gradient accumulates it as
                # a second side output
                masked_res = fd is not None and PL.MASKED_RES and not x.grad_written
                if fd is not None:
                    y2.take_grad_slot()
                wg2_first = fd is None
                if wg2_first and conv2_w.requires_grad:
                    self._wgrad_launch(h, w2, y2)
                # data gradient of conv2 -> gradient of the activation h (raw), BatchNorm-backward sums in the epilogue
                h.take_grad_slot()
                boff = p.alloc_bsums(Cc)
                p.bn_bwd.append((bn1, boff))
                g2 = nv.ConvDesc()
                self._conv_desc(y2, w2, h, 1, 3, dtype, into=g2)
                g2.x, g2.y = y2.gptr() if (fd is None) else 0, h.gptr()
                red = None
                if fd is not None:
                    # the block-end BatchNorm + ReLU backward (bn2, mask bits): staged operand of this launch
                    boff2 = p.alloc_bsums(Cc)
                    p.bn_bwd.append((bn2, boff2))
                    g2.x = out.gptr()
                    g2.pro_mode, g2.pro_x2, g2.pro_gamma, g2.pro_beta = 2, y2.ptr(), bn2.weight.data_ptr(), bn2.bias.data_ptr()
                    g2.pro_count, g2.pro_eps, g2.pro_mask = cnt, bn2.eps, fd.mask
                    g2.pro_side = y2.gptr()
                    if not masked_res:
                        g2.pro_side2, g2.pro_side2_acc = x.gptr(), x.take_grad_slot()
                    # its reduce: in the epilogue of the launch that completes out.grad when that is a row-strip data gradient
                    # of this lane accumulating onto ONE earlier producer (the next block of the stack), else a pass of its own
                    nxt = p.row_last_writer.get(out.gptr())
                    if (PL.BLOCK_END_REDUCE_FUSE and nxt is not None and not nxt[0].bnb_x and nxt[2] == len(out._grad_paths)
                            and all(q == p.lane_path for q in out._grad_paths)):
                        gn = nxt[0]
                        gn.bnb_x, gn.bnb_x_pitch, gn.bnb_mask, gn.bnb_mask_pitch = y2.ptr(), y2.pitch, fd.mask, fd.mask_pitch
                        gn.bnb_gamma, gn.bnb_beta, gn.bnb_count, gn.bnb_eps = bn2.weight.data_ptr(), bn2.bias.data_ptr(), cnt, bn2.eps
                        red = gn
                        p.counters["block_end_reduce_fused"] = p.counters.get("block_end_reduce_fused", 0) + 1
                    else:
                        b = nv.EwBwdDesc()
                        b.dout, b.out, b.dout_pitch, b.out_pitch = out.gptr(), out.ptr(), out.pitch, out.pitch
                        for f, _ in nv.EwInput._fields_:
                            setattr(b.inp, f, getattr(fd.inp[0], f))
                        b.dtype, b.N, b.H, b.W, b.C, b.relu = fd.dtype, fd.N, fd.H, fd.W, fd.C, fd.relu
                        b.mask, b.mask_pitch = fd.mask, fd.mask_pitch
                        red = b
                        p.bwd.append(PL.Launch("ew_red", b))
                    p.counters["block_end_apply_fused"] = p.counters.get("block_end_apply_fused", 0) + 1
                for i, (a, b) in enumerate(PL._TAPS3):
                    g2.dy[i], g2.dx[i], g2.wtap[i] = -a, -b, i
                g2.bnb_x, g2.bnb_x_pitch = y1.ptr(), y1.pitch
                g2.bnb_gamma, g2.bnb_beta, g2.bnb_count, g2.bnb_eps = gam, bet, cnt, bn1.eps
                # data gradient of conv1 -> x.grad; its staged operand is the BatchNorm + ReLU backward of (h.grad, y1)
                y1.take_grad_slot()
                acc = x.take_grad_slot()
                g1 = nv.ConvDesc()
                self._conv_desc(h, w1, x, 1, 3, dtype, into=g1)
                g1.x, g1.y = h.gptr(), x.gptr()
                for i, (a, b) in enumerate(PL._TAPS3):
                    g1.dy[i], g1.dx[i], g1.wtap[i] = -a, -b, i
                if fd is not None and masked_res:
                    assert not acc
                    g1.res, g1.res_pitch, g1.res_mask = out.gptr(), out.pitch, fd.mask
                    p.counters["block_end_masked_residual"] = p.counters.get("block_end_masked_residual", 0) + 1
                elif acc:
                    g1.res, g1.res_pitch = x.gptr(), x.pitch
                g1.pro_mode, g1.pro_x2, g1.pro_gamma, g1.pro_beta, g1.pro_count, g1.pro_eps = 2, y1.ptr(), gam, bet, cnt, bn1.eps
                g1.pro_side = y1.gptr()

                def late_b():
                    g2.w, g1.w = w2.arena.data_ptr() + w2.bwd_off * esz, w1.arena.data_ptr() + w1.bwd_off * esz
                    g2.stats = p.bsums.data_ptr() + 8 * boff
                    g2.bnb_stats = g1.pro_stats = p.stats.data_ptr() + 8 * y1.stats
                    g1.pro_bsums = g2.stats
                    if fd is not None:
                        sums2 = p.bsums.data_ptr() + 8 * boff2
                        g2.pro_stats, g2.pro_bsums = p.stats.data_ptr() + 8 * y2.stats, sums2
                        if isinstance(red, nv.ConvDesc):
                            red.stats, red.bnb_stats = sums2, g2.pro_stats
                        else:
                            red.sums, red.inp.stats = sums2, fd.inp[0].stats
                p.late(late_b)
                p.bwd.append(PL.Launch("conv", g2))
                if conv2_w.requires_grad and not wg2_first:
                    self._wgrad_launch(h, w2, y2)
                p.bwd.append(PL.Launch("conv", g1))
                # (so far) the last producer of x.grad: the block in front of this one may put its BatchNorm reduce here if
                # that is still so when its own backward is emitted
                if acc or (fd is not None and masked_res):
                    p.row_last_writer[x.gptr()] = (g1, p.lane_path, len((x.base if x.base is not None else x)._grad_paths))
                if conv1_w.requires_grad:
                    self._wgrad_launch(x, w1, y1)
            self.bwd_stack.append(bw)
            if bn2 is not None and fd is None:
                list.append(self.bwd_stack, act_bw)
        return out if bn2 is not None else y2

    def bottleneck_tail(self, h, conv3_w, bn3, x, proj=None):
        """out = relu(bn3(conv3(h)) + shortcut): the tail of a train-mode Bottleneck (reference HRnet.py:88-96) WITHOUT conv3's raw
        output in HBM; shortcut = x (identity) or, proj = (1x1 weight, BatchNorm), bn_d(conv_d(x)) - the projection of the first block of
        a stack (HRnet.py:139-150), whose raw output is not stored either.  At 64 x 64 the output of the 64 -> 256 layer is 134 MB per
        batch of 64, its input 33 MB: the product is recomputed wherever it is needed instead of stored and read back
        (hrp_conv_desc.tail_mode, csrc/conv_pw.h):
          forward   mode 1 (batch statistics of a product, nothing stored; one launch per product) + mode 2 / 5 (normalise, add the
                    shortcut, ReLU, bit mask) replace conv3 (+ the projection) + hrp_ew_fwd: 343 MB instead of 577 per identity block,
                    276 instead of 745 per projection block
          backward  per product mode 3 (sum g, sum g xhat) + mode 4 (gradient of the product; identity: the shortcut gradient as a
                    rider) replace hrp_ew_bwd_reduce + hrp_ew_bwd_apply: they read the 33 MB input instead of the 134 MB output
        The data and weight gradients of the 1x1 layers are the ordinary launches on mode 4's output.
        -> out, or None when the problem is not the pointwise kernel's (caller: the general path)."""
        p = self.plan
        if not (PL.BNECK_TAIL_FUSE and p.training and h.dtype == torch.bfloat16 and self.bn_batch_stats(bn3)):
            return None
        cout, cin = conv3_w.shape[0], conv3_w.shape[1]
        if (conv3_w.dim() == 4 and conv3_w.shape[2] != 1) or (x.N, x.H, x.W) != (h.N, h.H, h.W) or x.dtype != h.dtype:
            return None
        if h.pitch != cin or h.offset or h.C != cin or x.offset or x.pitch != x.C:
            return None
        if proj is None:
            if x.C != cout:
                return None
        else:
            wd_param, bnd = proj
            if tuple(wd_param.shape[:2]) != (cout, cin) or x.C != cin or (wd_param.dim() == 4 and wd_param.shape[2] != 1) or not self.bn_batch_stats(bnd):
                return None
        if p.need_grad and not (x.requires_grad and h.requires_grad and conv3_w.requires_grad and (proj is None or proj[0].requires_grad)):
            return None
        h.check_readable()
        x.check_readable()
        dtype = h.dtype
        w3 = p.weight(conv3_w, cout, cin, 1)
        wd = p.weight(proj[0], cout, cin, 1) if proj is not None else None
        y3 = p.new(h.N, h.H, h.W, cout, dtype)          # (never written by the forward; its gradient buffer is mode 4's output)
        yd = p.new(h.N, h.H, h.W, cout, dtype) if proj is not None else None
        out = p.new(h.N, h.H, h.W, cout, dtype)
        cnt = float(h.N * h.H * h.W)
        mask = torch.zeros(h.N * h.H * h.W * (cout // 8), dtype=torch.uint8, device=p.device)
        p.keep.append(mask)

        def desc(mode, second=False):
            src, w, y, bn = (x, wd, yd, proj[1]) if second else (h, w3, y3, bn3)
            d = self._conv_desc(src, w, y, 1, 1, dtype)
            d.tail_mode = mode
            d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps = bn.weight.data_ptr(), bn.bias.data_ptr(), cnt, bn.eps
            d.tail_mask = mask.data_ptr()
            return d
        dA, dB = desc(1), desc(5 if proj is not None else 2)
        dA2 = desc(1, True) if proj is not None else None
        dB.y = out.ptr()
        if proj is None:
            dB.res, dB.res_pitch = x.ptr(), x.pitch
        else:
            dB.tail_x2, dB.tail_gamma2, dB.tail_beta2 = x.ptr(), proj[1].weight.data_ptr(), proj[1].bias.data_ptr()
        # eligibility of all forms (dummy aligned pointers where the arenas are not allocated yet)
        for mode in (1, 5 if proj is not None else 2, 3, 4):
            q = desc(mode)
            q.w = q.stats = q.tail_stats = q.tail_bsums = q.tail_g = q.x
            if mode == 2:
                q.res = q.x
            if mode == 5:
                q.tail_x2 = q.tail_w2 = q.tail_stats2 = q.tail_gamma2 = q.tail_beta2 = q.x
            if not nv.lib().hrp_conv_pointwise(C.byref(q)):
                return None
        for w in (w3, wd):
            if w is not None:
                w.dtype, w.cin_used = dtype, cin
                w.need_t = getattr(w, "need_t", False) or p.need_grad
        y3.requires_grad = out.requires_grad = p.need_grad
        y3.stats = p.alloc_stats(cout)
        p.bn_train.append((bn3, y3.stats, h.N * h.H * h.W))
        if proj is not None:
            yd.requires_grad = p.need_grad
            yd.stats = p.alloc_stats(cout)
            p.bn_train.append((proj[1], yd.stats, h.N * h.H * h.W))

        def late():
            dA.w = dB.w = w3.arena.data_ptr() + w3.fwd_off * 2
            dA.stats = dB.tail_stats = p.stats.data_ptr() + 8 * y3.stats
            if proj is not None:
                dA2.w = dB.tail_w2 = wd.arena.data_ptr() + wd.fwd_off * 2
                dA2.stats = dB.tail_stats2 = p.stats.data_ptr() + 8 * yd.stats
        p.late(late)
        p.fwd.append(PL.Launch("conv", dA))
        if proj is not None:
            p.fwd.append(PL.Launch("conv", dA2))
        p.fwd.append(PL.Launch("conv", dB))
        out.producer = None
        p.counters["bottleneck_tails"] = p.counters.get("bottleneck_tails", 0) + 1
        if p.need_grad:
            def bw():
                if not out.grad_written:
                    return
                for second in ((False, True) if proj is not None else (False,)):
                    src, w, y, bn = (x, wd, yd, proj[1]) if second else (h, w3, y3, bn3)
                    y.take_grad_slot()
                    boff = p.alloc_bsums(cout)
                    p.bn_bwd.append((bn, boff))
                    dC, dD = desc(3, second), desc(4, second)
                    dC.tail_g = dD.tail_g = out.gptr()
                    dD.y = y.gptr()
                    if proj is None:
                        dD.tail_side, dD.tail_side_acc = x.gptr(), x.take_grad_slot()

                    def late_b(dC=dC, dD=dD, w=w, y=y, boff=boff):
                        dC.w = dD.w = w.arena.data_ptr() + w.fwd_off * 2
                        dC.tail_stats = dD.tail_stats = p.stats.data_ptr() + 8 * y.stats
                        dC.stats = dD.tail_bsums = p.bsums.data_ptr() + 8 * boff
                    p.late(late_b)
                    p.bwd.append(PL.Launch("conv", dC))
                    p.bwd.append(PL.Launch("conv", dD))
                self._conv_bwd(h, w3, y3, None, 1, 1, dtype, None, False)
                if proj is not None:
                    self._conv_bwd(x, wd, yd, None, 1, 1, dtype, None, False)
            self.bwd_stack.append(bw)
        return out
