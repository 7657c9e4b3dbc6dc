"""Glue between torch.nn.Module / autograd and the static plans (plan.py).

A ``PlannedModule`` keeps one plan per (input shapes, compute dtype, train/eval, grad mode).  Its
``forward`` binds the caller's tensors, runs the plan's forward list and returns fresh torch tensors;
when gradients are required the call is wrapped in ONE ``torch.autograd.Function`` whose backward runs
the plan's backward list and publishes parameter gradients into ``param.grad``.
"""
import os

import torch
import torch.nn as nn

from . import _native as nv
from .plan import Plan, PlanBuilder, _dt


def require_gpu(device):
    if device.type != "cuda":
        raise nv.HrpError("hrpe_amd runs on an MI355X (gfx950) only: tensors must live on a HIP device; "
                          "there is no CPU path")
    nv.lib()


# Graph cache: the reference's loop calls model(...) / loss.backward() eagerly (scripts/train_full.py:53-67).  Walking a
# plan in Python costs ~10 us per launch (2 200 launches per training step), so after GRAPH_WARMUP eager calls of one
# plan its forward (weight pack + launch list + running statistics) and its backward are captured in HIP graphs and
# replayed: the unmodified eager loop then runs at the speed of bench.py's captured step.  Inputs are copied into
# plan-owned staging tensors first (a captured launch bakes its pointers).  Off: HRP_NO_MODULE_GRAPH=1 or
# runtime.GRAPH_CACHE = False; never active inside somebody else's capture or with a split backward.
GRAPH_CACHE = not os.environ.get("HRP_NO_MODULE_GRAPH")
PLAN_CACHE_MAX = int(os.environ.get("HRP_PLAN_CACHE", "6"))   # plans kept per module (train / eval x a few input shapes)
GRAPH_WARMUP = 2


class Runner:
    """A built plan plus its external inputs/outputs."""

    def __init__(self, plan, in_names, outs, img_inputs):
        self.plan, self.in_names, self.outs, self.img_inputs = plan, in_names, outs, img_inputs
        self.calls = self.bwd_calls = 0
        self.g_fwd = self.g_bwd = self.static_in = None
        self.graph_failed = False

    # ---- eager pieces ---------------------------------------------------------------------------------------
    def _forward_launches(self, prep=True):
        p = self.plan
        # training plans repack the weights on every step (they change every step, and a captured HIP graph must
        # contain the pack launch); inference plans repack only when a parameter changed - eagerly, OUTSIDE their graph
        if prep or p.need_grad:
            p.run_prep(force=p.need_grad)
        p.run_forward()

    def _results(self):
        """Fresh tensors for the caller (the plan's buffers are overwritten by the next call).  The [B, C] fp32 outputs - all eight of
        RootNetwithRegInt - leave through ONE launch into one allocation and come back as views of it (eight device copies, each a
        node of its own in a captured step, sat between the forward and the loss where nothing overlaps them)."""
        p = self.plan
        res = [None] * len(self.outs)
        dense = [(i, h, shape) for i, (kind, h, shape) in enumerate(self.outs) if kind != "nchw" and h.dtype == torch.float32]
        if len(dense) > 1:
            flat = torch.empty(sum(h.N * h.C for _, h, _ in dense), dtype=torch.float32, device=p.device)
            items, off = [], 0
            for i, h, shape in dense:
                items.append((h.ptr(), h.pitch, flat.data_ptr() + 4 * off, h.C, h.N, h.C, 0))
                res[i] = flat[off:off + h.N * h.C].view(shape)
                off += h.N * h.C
            nv.copy_cols_batch(items, torch.cuda.current_stream(p.device).cuda_stream)
        for i, (kind, h, shape) in enumerate(self.outs):
            if res[i] is not None:
                continue
            if kind == "nchw":
                res[i] = h["out"].clone() if self.g_fwd is not None else h["out"]   # (graph pool memory is reused)
            else:
                res[i] = h.buf.view(-1)[h.offset: h.offset + h.N * h.pitch].view(h.N, h.pitch)[:, : h.C].reshape(shape).clone()
        return tuple(res)

    def _graphs_allowed(self):
        p = self.plan
        return (GRAPH_CACHE and not self.graph_failed and not p.split_active and not torch.cuda.is_current_stream_capturing()
                and not nv.profiling())

    def _capture(self, what):
        """Capture the forward or the backward launch list of this plan (every kernel has run eagerly before: lazy code
        loading and one-time function attributes are not capturable)."""
        p = self.plan
        try:
            g = torch.cuda.CUDAGraph()
            pool = self.g_fwd.pool() if self.g_fwd is not None else None
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                if what == "fwd":
                    self._forward_launches(prep=False)
                else:
                    # a backward graph of its own always zeroes the gradient arena itself: replayed twice behind one forward
                    # (retain_graph) it must not rely on the forward's early memset (Plan.run_prep)
                    p._arena_clean = False
                    p.run_backward(None)
            return g
        except Exception as e:   # a capture-unsafe op somewhere: stay eager for this plan
            import sys
            print(f"hrpe_amd: graph capture of the {what} plan failed ({e!r}); running it eagerly", file=sys.stderr)
            self.graph_failed = True
            return None

    # ---- public ---------------------------------------------------------------------------------------------
    def forward(self, tensors):
        p = self.plan
        self.calls += 1
        if self._graphs_allowed() and self.calls > GRAPH_WARMUP:
            if self.g_fwd is None:
                self.static_in = [torch.empty_like(t) for t in tensors]
                for n, st in zip(self.in_names, self.static_in):
                    p.dyn[n] = st
                self.g_fwd = self._capture("fwd")
            if self.g_fwd is not None and p.need_grad and self.g_bwd is None and self.bwd_calls >= GRAPH_WARMUP:
                self.g_bwd = self._capture("bwd")
        if self.g_fwd is not None and self._graphs_allowed():
            for st, t in zip(self.static_in, tensors):
                st.copy_(t)
            if not p.need_grad:
                p.run_prep(force=False)      # pack / fold only when a parameter changed (not part of the graph)
            self.g_fwd.replay()
            if p.training and p._run_tab:
                from .plan import bump_param_epoch
                bump_param_epoch()       # running statistics moved (the Python side of run_forward does not run on replay)
            return self._results()
        for n, t in zip(self.in_names, tensors):
            p.dyn[n] = t
        self._forward_launches()
        return self._results()

    def backward(self, grads):
        p = self.plan
        s = torch.cuda.current_stream(p.device).cuda_stream
        items, keep = [], []
        for (kind, h, shape), g in zip(self.outs, grads):
            t = h["handle"] if kind == "nchw" else h
            if not (t.requires_grad and t.grad_written):
                continue
            gb = t.grad_buf()
            if kind != "nchw" and t.dtype == torch.float32:
                # the [B, C] gradients coming back from the loss: one launch for all of them (None: zero fill)
                if g is not None:
                    g = g.reshape(t.N, t.C)
                    if g.dtype != torch.float32 or not g.is_contiguous():
                        g = g.contiguous().float()
                    keep.append(g)
                items.append((None if g is None else g.data_ptr(), t.C, t.gptr(), t.pitch, t.N, t.C, 0))
            elif g is None:
                gb.zero_()
            elif kind == "nchw":
                g = g.contiguous().float()
                nv.call("hrp_nchw_to_nhwc", g.data_ptr(), t.gptr(), _dt(t.dtype), t.N, t.C, t.H, t.W, t.pitch, s)
            else:
                gb.view(t.N, t.pitch)[:, : t.C].copy_(g.reshape(t.N, t.C))
        if items:
            nv.copy_cols_batch(items, s)
        self.bwd_calls += 1
        if self.g_bwd is not None and self._graphs_allowed():
            self.g_bwd.replay()
        else:
            p.run_backward("first" if p.split_active else None)   # (split: the caller runs plan.run_backward("rest"))
        p.publish_param_grads()
        gin = []
        for n in self.in_names:
            t = self.img_inputs.get(n)
            if t is not None and t.requires_grad and t.grad_written:
                out = torch.empty(t.N, t.C, t.H, t.W, dtype=torch.float32, device=p.device)
                nv.call("hrp_nhwc_to_nchw", t.gptr(), out.data_ptr(), _dt(t.dtype), t.N, t.C, t.H, t.W, t.pitch, s)
                gin.append(out)
            else:
                gin.append(None)
        return gin


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, anchor, *tensors):
        ctx.runner = runner
        ctx.n = len(tensors)
        outs = runner.forward(tensors)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        gin = ctx.runner.backward(grads)
        return (None, None) + tuple(gin)


MODE_EPOCH = [0]     # bumped by every train() / eval() call on a PlannedModule: invalidates the cached BatchNorm-mode signatures


class PlannedModule(nn.Module):
    """Base of every hrpe_amd module: owns plans, never computes with torch ops."""

    def train(self, mode=True):
        MODE_EPOCH[0] += 1
        return super().train(mode)

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_plans", {})
        object.__setattr__(self, "_bn_sig", (-1, []))
        object.__setattr__(self, "_compute_dtype", torch.float32)
        object.__setattr__(self, "_x3", False)

    # compute dtype of the convolution trunk (heads stay fp32)
    def set_compute_dtype(self, dtype):
        """torch.float32, torch.bfloat16, or "fp32x3": fp32 tensors whose convolution products are three bf16 MFMAs on split (hi + lo)
        operands - ~2^-16 relative per product at a third of the bf16 matrix rate, where the fp32 matrix cores run at a sixteenth
        (include/hrp.h HRP_F32X3; the precision between bf16 and fp32 of DESIGN 4)."""
        x3 = dtype == "fp32x3"
        if x3:
            dtype = torch.float32
        assert dtype in (torch.float32, torch.bfloat16)
        for m in self.modules():
            if isinstance(m, PlannedModule):
                object.__setattr__(m, "_compute_dtype", dtype)
                object.__setattr__(m, "_x3", x3)
                m._plans.clear()
        return self

    @property
    def compute_dtype(self):
        return self._compute_dtype

    def flat_grads(self):
        """Flat fp32 gradient arena of the training plan(s) (views of it are the parameters' .grad)."""
        return [r.plan.grad_arena for r in self._plans.values() if r.plan.grad_arena is not None]

    def enable_split_backward(self, min_frac=0.55, fracs=None):
        """Data-parallel overlap: from now on a backward through this module runs only the first part of the training
        plan's backward list; the caller runs ``plan.run_backward("rest")`` itself (e.g. in a second HIP graph) and may
        all-reduce the returned arena ranges in between - no launch of the second part touches them.
        -> (plan, [(offset, numel), ...]) or None when the plan has no suitable split.
        fracs (e.g. (0.25, 0.6, 0.9)): k cuts: the backward becomes k + 1 segments - the autograd call runs segment 0, the caller
        ``plan.run_backward(("seg", j))`` for j = 1 .. k - and the return value is (plan, [ranges_0, .., ranges_{k-1}]): ranges_j
        are final once segment j has run, so each can travel while the next segment computes and only the gradients of the
        last segment (the stems and first stages: a few per cent of the bytes) are reduced behind the backward."""
        runners = [r for r in self._plans.values() if r.plan.grad_arena is not None]
        if len(runners) != 1:
            return None
        plan = runners[0].plan
        if fracs is not None:
            cuts = plan.analyze_backward_split(fracs=fracs)
            if not cuts:
                return None
            plan.bwd_cuts, plan.bwd_split = [c for c, _ in cuts], None
            plan.split_active = True
            return plan, [r for _, r in cuts]
        info = plan.analyze_backward_split(min_frac)
        if info is None:
            return None
        plan.bwd_split, ranges = info
        plan.bwd_cuts = None
        plan.split_active = True
        return plan, ranges

    def check_split_backward(self, ranges):
        """Self-check of the split on this machine, for library users as for bench.py: call it right after a backward
        in split mode (only the first part has run).  Runs the rest of the backward and verifies bit for bit that it left the
        arena ranges handed out by enable_split_backward untouched (they may already be with RCCL) and that the arena is
        finite.  On a violation the split is switched off and a RuntimeError says so (the caller falls back to the unsplit
        backward).  With k cuts `ranges` is the list of k range lists: every later segment must leave every earlier list alone."""
        runners = [r for r in self._plans.values() if r.plan.grad_arena is not None and r.plan.split_active]
        if len(runners) != 1:
            raise RuntimeError("check_split_backward: no plan with an active split")
        plan = runners[0].plan
        arena = plan.grad_arena
        multi = plan.bwd_cuts is not None
        groups = ranges if multi else [ranges]
        nseg = len(plan.bwd_cuts) + 1 if multi else 2
        snaps = []
        ok = True
        for j in range(1, nseg):
            torch.cuda.synchronize(arena.device)
            grp = groups[j - 1]
            snaps.append((grp, torch.cat([arena[o:o + n] for o, n in grp]).clone() if grp else arena[:0].clone()))
            plan.run_backward(("seg", j))
            torch.cuda.synchronize(arena.device)
            for grp_, before in snaps:
                after = torch.cat([arena[o:o + n] for o, n in grp_]) if grp_ else arena[:0]
                ok = ok and torch.equal(before, after)
        ok = ok and bool(torch.isfinite(arena).all())
        if not ok:
            self.disable_split_backward()
            raise RuntimeError("split backward self-check failed: a later part of the backward touched gradient ranges "
                               "that were final after an earlier one")
        return True

    def disable_split_backward(self):
        for r in self._plans.values():
            r.plan.split_active = False

    def invalidate_plans(self):
        for m in self.modules():
            if isinstance(m, PlannedModule):
                m._plans.clear()

    def _apply(self, fn, *a, **k):
        # parameters moved / cast: cached plans hold stale pointers
        r = super()._apply(fn, *a, **k)
        self.invalidate_plans()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, PlannedModule):
                for pl in m._plans.values():
                    pl.plan._versions = None
        return r

    def _frozen_bn_signature(self):
        """Which BatchNorm modules of a TRAINING module are in eval() (train_sim2real.py:139-146 trains that way), as a bit mask over
        self.modules() order: part of the plan-cache key.  The key holds the IDENTITY of the frozen set (two different
        subsets of equal size - the regression trunk's BatchNorms against the DepthNet's - must not share a plan, which has every
        bn.training baked in), and it is re-read from the modules on every call: a plain `m.training = False` assignment does not
        pass through train() / eval().  Only the module LIST is cached (per MODE_EPOCH)."""
        if self._bn_sig[0] != MODE_EPOCH[0]:
            object.__setattr__(self, "_bn_sig", (MODE_EPOCH[0], [m for m in self.modules() if hasattr(m, "running_mean")]))
        sig = 0
        if self.training:
            for i, m in enumerate(self._bn_sig[1]):
                if not m.training:
                    sig |= 1 << i
        return sig

    # subclasses implement: _signature(*inputs) -> hashable ; _build(pb, *inputs) -> (in_names, outs, img_inputs)
    def _run(self, *tensors):
        dev = tensors[0].device
        require_gpu(dev)
        need_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                 or any(t.requires_grad for t in tensors))
        # uint8 inputs are raw images (dataset bytes): the plan's input kernel divides them by 255 on the way in
        # (a training module whose BatchNorm modules were switched to eval() - train_sim2real.py:139-146 - is a different plan)
        bn_eval = self._frozen_bn_signature()
        key = (tuple(tuple(t.shape) for t in tensors), self._compute_dtype, self.training, need_grad,
               tuple(bool(t.requires_grad) for t in tensors) if need_grad else (),
               tuple(t.dtype == torch.uint8 for t in tensors), bn_eval, self._x3)
        runner = self._plans.pop(key, None)
        if runner is not None:
            self._plans[key] = runner        # most recently used last
        if runner is None:
            # bounded cache: a plan owns its activation / gradient / scratch arenas (tens of GB at B = 64), and a
            # loader with a ragged last batch or varying crops would otherwise keep one set per shape alive for ever
            while len(self._plans) >= PLAN_CACHE_MAX:
                self._plans.pop(next(iter(self._plans)))
            plan = Plan(dev, self._compute_dtype, self.training, need_grad)
            plan.x3 = bool(self._x3)
            if need_grad:
                plan.preallocate_param_grads(list(self.parameters()))
            pb = PlanBuilder(plan)
            in_names, outs, img_inputs = self._build(pb, *tensors)
            for kind, h, shape in outs:
                t = h["handle"] if kind == "nchw" else h
                pb.output(t)
            pb.finish()
            runner = Runner(plan, in_names, outs, img_inputs)
            self._plans[key] = runner
        tensors = tuple(t.contiguous() if t.dtype == torch.uint8 else
                        (t.contiguous().float() if t.dtype != torch.float32 or not t.is_contiguous() else t)
                        for t in tensors)
        if need_grad:
            anchor = next((p for p in self.parameters() if p.requires_grad), None)
            if anchor is None:
                anchor = tensors[0]
            return _PlanFn.apply(runner, anchor, *tensors)
        return runner.forward(tensors)


class SingleTensorModule(PlannedModule):
    """Modules mapping one NCHW tensor to one NCHW tensor through ``emit(pb, x)`` (blocks, stages)."""

    def forward(self, x):
        return self._run(x)[0]

    def _build(self, pb, x):
        N, Cc, H, W = x.shape
        t = pb.image_input("x", N, Cc, H, W, u8=x.dtype == torch.uint8)
        t.requires_grad = pb.plan.need_grad and x.requires_grad
        y = self.emit(pb, t)
        holder = pb.nchw_output(y)
        holder["handle"] = y
        return ["x"], [("nchw", holder, None)], {"x": t}
