"""Passes over the launch lists of a built plan (holistic-robot-pose-estimation_amd/plan.py: Plan.finalize and the data-parallel step).

flatten            lanes of the virtual parallel blocks walked in lock step, launches of one merge key folded into batched launches
fuse_bn_reduce     BatchNorm-backward reduce passes moved into the epilogue of the data gradient that produces their input
sink_wgrads        weight-gradient launches of a lane regrouped into batches of WGRAD_SINK problems
insert_folds       one fold launch per WGRAD_FOLD_EVERY deferred weight-gradient problems
analyze_backward_split   cut positions of the backward list after which a given share of the gradient bytes is final

Every function takes the Plan as its first argument; the switches (BATCHING, GREEDY_MERGE, PLAN_MODE, WGRAD_SINK ..) are read from
the plan module at call time, where tests and tools patch them."""
import collections
import ctypes as C

import torch

from . import _native as nv
from . import plan as PL


def flatten(plan, entries):
    """Launch list of the merged / hybrid modes.  The lanes of every VIRTUAL parallel block (all blocks in merged mode)
    are walked in lock step - position k of every lane before position k + 1 of any - and launches of equal merge
    key at one position are folded into batched launches.  Lanes of one block are independent by construction
    (TensorH.check_readable / take_grad_slot), so any interleaving that keeps each lane's own order is a valid
    serial order.  Blocks that asked for streams (hybrid mode) keep their fork / join markers; the merge happens
    inside each of their lanes.  -> list of Entry (path unused)."""
    root = PL._Seq()
    for e in entries:
        if e.lane is None and not isinstance(e.op, PL._PackJoin):
            continue        # fork / join markers are re-created below
        seq = root
        for blk, idx in (e.path or ()):
            node = seq.blocks.get(blk)
            if node is None:
                node = PL._Par()
                node.blk = blk
                seq.blocks[blk] = node
                seq.items.append(node)
            # a block's launches must be contiguous in its parent's sequence: anything emitted after the block
            # started and before it ended would otherwise be moved behind it
            assert seq.items[-1] is node, "plan: launches of a parallel block are interleaved with its parent's"
            seq = node.lanes.setdefault(idx, PL._Seq())
        seq.items.append(e.op)

    def key_of(op):
        return op.merge_key() if PL.BATCHING and isinstance(op, (PL.Launch, PL.BatchLaunch, PL.BlockLaunch)) else None

    def emit_groups(groups, out):
        for key, ops in groups.items():
            if key[0] == "block":
                out += PL._pair_block(plan, ops)
            else:
                out += ops if len(ops) == 1 else PL._merge_ops(plan, ops)

    def greedy(kids):
        """Lanes whose launch sequences differ (the paths of a fuse layer): instead of position k of every lane, take the
        HEADS of all lanes, send the unbatchable ones out, then the largest group of equal merge key; the other lanes wait
        for partners.  Any interleaving that keeps each lane's own order is valid."""
        out, ptr = [], [0] * len(kids)
        while True:
            heads = [(i, kids[i][ptr[i]]) for i in range(len(kids)) if ptr[i] < len(kids[i])]
            if not heads:
                return out
            groups, moved = collections.OrderedDict(), False
            for i, op in heads:
                key = key_of(op)
                if key is None:
                    out.append(op)
                    ptr[i] += 1
                    moved = True
                else:
                    groups.setdefault(key, []).append((i, op))
            if moved:
                continue
            key = max(groups, key=lambda k: len(groups[k]))
            emit_groups({key: [op for _, op in groups[key]]}, out)
            for i, _ in groups[key]:
                ptr[i] += 1

    def lockstep(kids):
        if PL.GREEDY_MERGE and len({tuple(key_of(op) for op in x) for x in kids}) > 1:
            return greedy(kids)
        out = []
        for k in range(max(len(x) for x in kids)):
            groups = collections.OrderedDict()
            for x in kids:
                if k < len(x):
                    key = key_of(x[k])
                    if key is None:
                        out.append(x[k])
                    else:
                        groups.setdefault(key, []).append(x[k])
            for key, ops in groups.items():
                if key[0] == "block":
                    out += PL._pair_block(plan, ops)
                else:
                    out += ops if len(ops) == 1 else PL._merge_ops(plan, ops)
        return out

    def walk(seq, lane):
        """-> entries of this sequence, running on stream `lane`."""
        out = []
        for it in seq.items:
            if isinstance(it, PL._PackJoin):
                out.append(PL.Entry(None, (), it))
            elif not isinstance(it, PL._Par):
                out.append(PL.Entry(lane, (), it))
            else:
                real = PL.PLAN_MODE == "hybrid" and plan._block_lanes.get(it.blk) is not None
                if real:
                    ids = plan._block_lanes[it.blk]
                    kids = [(ids[idx], walk(sub, ids[idx])) for idx, sub in sorted(it.lanes.items())]
                    children = [l for l, _ in kids if l != lane]
                    out.append(PL.Entry(None, (), PL._LaneSync("fork", lane, children)))
                    for _, ents in kids:
                        out += ents
                    out.append(PL.Entry(None, (), PL._LaneSync("join", lane, children)))
                else:
                    kids = [walk(sub, lane) for _, sub in sorted(it.lanes.items())]
                    assert all(e.lane == lane for x in kids for e in x), "a virtual block cannot contain a block with streams"
                    out += [PL.Entry(lane, (), op) for op in lockstep([[e.op for e in x] for x in kids])]
        return out
    return walk(root, 0)


def fuse_bn_reduce(plan):
    """conv -> BN -> ReLU -> conv (the interior of BasicBlock / Bottleneck, HRnet.py:41-57): the gradient of the
    activation comes from exactly one data-gradient launch; its epilogue then also accumulates the two BatchNorm
    backward sums (sum g, sum g * xhat) and the separate hrp_ew_bwd_reduce launch over the same tensors goes away."""
    is_l = lambda e, fam: isinstance(e.op, PL.Launch) and e.op.fam == fam   # noqa: E731
    convs, other = {}, set()
    for i, e in enumerate(plan.bwd):
        if is_l(e, "conv"):
            convs.setdefault(e.op.desc.y, []).append((i, e))
        elif is_l(e, "ew_app"):
            other.update(x for x in (e.op.desc.din, e.op.desc.din2) if x)
    fwd_by_mask = {e.op.desc.mask: e.op for e in plan.fwd if is_l(e, "ew_fwd") and e.op.desc.mask}
    drop = set()
    for i, e in enumerate(plan.bwd):
        if not is_l(e, "ew_red"):
            continue
        b = e.op.desc
        if b.inp.mode != nv.EW_BN_TRAIN or b.inp.up != 1 or b.relu != 1 or not b.mask or b.dout in other:
            continue
        cands = convs.get(b.dout, [])
        fw = fwd_by_mask.get(b.mask)
        if len(cands) != 1 or fw is None or fw.desc.nin != 1:
            continue
        # every producer of a gradient registers through take_grad_slot - also the ones that are plain closures in the
        # launch list (copy_cols, pooling, linear layers, soft-argmax ..), which the scan above cannot see: exactly ONE
        # producer (the candidate conv) or the epilogue would reduce a partial gradient
        owner = plan.grad_owner.get(b.dout)
        if owner is None or len(owner._grad_paths) != 1:
            continue
        j, ce = cands[0]
        d = ce.op.desc
        esz = 2 if d.dtype == nv.HRP_BF16 else 4
        vec = 16 // esz
        ddt = nv.HRP_F32 if d.dtype == nv.HRP_F32X3 else d.dtype      # (fp32x3 convolutions read and write fp32 tensors)
        if j > i or ce.lane != e.lane or ce.path != e.path:
            continue
        if d.res or d.relu or d.bias or d.scale or d.stats or d.out_stride != 1 or (d.y_H, d.y_W) != (d.Ho, d.Wo) or d.pro_mode or d.bnb_x:
            continue
        if (d.N, d.Ho, d.Wo, d.Cout, d.y_pitch, ddt) != (b.N, b.H, b.W, b.C, b.dout_pitch, b.dtype) or d.Cout % vec:
            continue
        if d.y % 16 or (d.y_pitch * esz) % 16 or b.inp.ptr % 16 or (b.inp.pitch * esz) % 16:
            continue
        consts = torch.zeros(2 * b.C, dtype=torch.float32, device=plan.device)
        plan.keep.append(consts)
        fw.desc.consts_out = consts.data_ptr()
        d.bnb_x, d.bnb_x_pitch = b.inp.ptr, b.inp.pitch
        d.bnb_mask, d.bnb_mask_pitch = b.mask, b.mask_pitch
        d.bnb_consts, d.stats = consts.data_ptr(), b.sums
        drop.add(i)
    if drop:
        kept = [e for i, e in enumerate(plan.bwd) if i not in drop]
        del plan.bwd[:]
        list.extend(plan.bwd, kept)
    plan.counters["bn_reduce_fused"] = len(drop)


def sink_wgrads(plan, entries):
    """Regroup the weight-gradient launches of every lane into batches of WGRAD_SINK problems of one tap count (their
    inputs - the layer's forward input and its output gradient - stay untouched for the rest of the step)."""
    out, pend = [], {}

    def emit(lane, key, path, force):
        items = pend[lane][key]
        while items and (force or len(items) >= PL.WGRAD_SINK):
            grp, rest, seen = [], [], set()
            for it in items:
                w = it.written()
                if len(grp) < PL.WGRAD_SINK and not any(a in seen for a in w):
                    grp.append(it)
                    seen.update(w)
                else:
                    rest.append(it)
            out.append(PL.Entry(lane, path, grp[0] if len(grp) == 1 else PL.BatchLaunch(plan, grp)))
            items = rest
        pend[lane][key] = items

    def flush(lane, path):
        for key in list(pend.get(lane, {})):
            emit(lane, key, path, True)

    for e in entries:
        if e.lane is None:
            if getattr(e.op, "kind", None) == "join":
                for c in e.op.children:
                    flush(c, e.path)
            out.append(e)
            continue
        op = e.op
        if isinstance(op, (PL.Launch, PL.BatchLaunch)) and op.fam == "wgrad" and not any(it.desc.reserved for it in op.launches()):
            for it in op.launches():
                key = it.merge_key()
                if key is None:
                    out.append(PL.Entry(e.lane, e.path, it))
                    continue
                pend.setdefault(e.lane, {}).setdefault(key, []).append(it)
                if len(pend[e.lane][key]) >= PL.WGRAD_SINK:
                    emit(e.lane, key, e.path, False)
        else:
            out.append(e)
    for lane in sorted(pend):
        flush(lane, ())
    return out


def insert_folds(plan, entries):
    """Deferred weight-gradient folds: after every WGRAD_FOLD_EVERY phase-1 problems of a lane, before the lane
    joins its parent and at the end of the list, one HRP_BATCH_WGRAD_FOLD launch folds the lane's pending slabs."""
    out, pending = [], {}

    def flush(lane, path, everything=True):
        descs = pending.pop(lane, [])
        while descs and (everything or len(descs) >= nv.BATCH_MAX):
            # one launch folds problems with pairwise DISTINCT outputs only (the fold is a plain read-modify-write of dW:
            # a weight applied twice - or a tap group launched twice - must fold in consecutive launches, not race in one)
            grp, rest, seen = [], [], set()
            for f in descs:
                key = f.dw + 4 * f.dw_tap_off
                if len(grp) < nv.BATCH_MAX and key not in seen:
                    grp.append(f)
                    seen.add(key)
                else:
                    rest.append(f)
            b = PL.BatchLaunch(plan, [PL.Launch("wgrad_fold", f) for f in grp])
            b.prepare()
            out.append(PL.Entry(lane, path, b))
            descs = rest
        if descs:
            pending[lane] = descs

    for e in entries:
        if e.lane is None and getattr(e.op, "kind", None) == "join":
            for c in e.op.children:
                flush(c, e.path)
        out.append(e)
        if e.lane is not None and isinstance(e.op, (PL.Launch, PL.BatchLaunch)) and e.op.fam == "wgrad" and e.op.launches()[0].desc.phase == 1:
            pending.setdefault(e.lane, []).extend(f for f in e.op.fold_descs() if f.G > 0)
            if len(pending[e.lane]) >= min(max(PL.WGRAD_FOLD_EVERY, 1), nv.BATCH_MAX):
                flush(e.lane, e.path, everything=PL.WGRAD_FOLD_EVERY < nv.BATCH_MAX)
    for lane in sorted(pending):
        flush(lane, ())
    return out


def analyze_backward_split(plan, min_frac=0.55, fracs=None):
    """-> (split index, [(offset, numel)] arena ranges that no launch at or after the split touches) or None.
    fracs (e.g. (0.25, 0.6, 0.9)): k cuts instead -> [(split index, ranges that became final since the previous cut)], so
    that the collective left behind the last launch is a small tail (buckets in reverse registration order).

    Every backward launch is replayed against a recording stand-in for the C ABI (nothing runs); any pointer
    argument or descriptor field that points into the gradient arena marks that parameter as touched by that
    launch.  The split is the first top-level position (outside every parallel block) after which at least
    `min_frac` of the gradient bytes are final."""
    import bisect
    if plan.grad_arena is None or not plan._grad_layout:
        return None
    base, nbytes = plan.grad_arena.data_ptr(), plan.grad_arena.numel() * 4
    starts = [o * 4 for o, _ in plan._grad_layout]
    last = [-1] * len(starts)
    hits = []

    def walk(v):
        if isinstance(v, bool) or v is None:
            return
        if isinstance(v, int):
            if base <= v < base + nbytes:
                hits.append(v - base)
        elif isinstance(v, C.Structure):
            for name, _t in v._fields_:
                walk(getattr(v, name))
        elif isinstance(v, C.Array):
            if issubclass(v._type_, (C.Structure, C.Array, C.c_void_p)):
                for e in v:
                    walk(e)
        elif hasattr(v, "_obj"):          # ctypes.byref(struct)
            walk(v._obj)
        elif isinstance(v, C.c_void_p):
            walk(v.value)

    real = nv.call
    try:
        nv.call = lambda name, *args: [walk(a) for a in args] and 0
        depth, tops = 0, []
        for i, e in enumerate(plan.bwd_ops()):
            lane, op = e.lane, e.op
            if lane is None:
                if getattr(op, "kind", None) == "fork":
                    if depth == 0:
                        tops.append(i)
                    depth += 1
                elif getattr(op, "kind", None) == "join":
                    depth -= 1
                continue
            if depth == 0:
                tops.append(i)
            del hits[:]
            if isinstance(op, (PL.Launch, PL.BatchLaunch, PL.BlockLaunch, PL.BlockBatch)):
                for it in op.launches():     # the descriptors say what a launch touches
                    walk(it.desc)
            else:
                op(0)
            for h in hits:
                last[bisect.bisect_right(starts, h) - 1] = i
    finally:
        nv.call = real
    if plan._pgrad_tab:   # hrp_bn_param_grad after the list writes the BatchNorm weight / bias gradients
        bn_ptrs = set()
        for bn, _off in plan.bn_bwd:
            for t in (bn.weight, bn.bias):
                g = plan._grad_views.get(id(t))
                if g is not None:
                    bn_ptrs.add(g.data_ptr() - base)
        for k, st in enumerate(starts):
            if st in bn_ptrs:
                last[k] = len(plan.bwd_ops())
    total = sum(n for _, n in plan._grad_layout)

    def ranges_final_before(c, lo=-1):
        """arena ranges whose last toucher lies in [lo, c)"""
        ranges = []
        for (off, n), l in zip(plan._grad_layout, last):
            if not (lo <= l < c):
                continue
            n4 = PL._rup(n, 4)
            if ranges and ranges[-1][0] + ranges[-1][1] == off:
                ranges[-1][1] += n4
            else:
                ranges.append([off, n4])
        return [tuple(r) for r in ranges]

    if fracs is not None:
        # k cut positions: the first top-level position at which at least f of the gradient bytes are final, for every f
        cuts, prev = [], -1
        for f in sorted(fracs):
            c = next((c for c in tops if c > 0 and sum(n for (_, n), l in zip(plan._grad_layout, last) if l < c) >= f * total), None)
            if c is None or (cuts and c <= cuts[-1][0]):
                continue
            cuts.append((c, ranges_final_before(c, prev)))
            prev = c
        return cuts or None
    for c in tops:
        if c == 0:
            continue
        final = sum(n for (_, n), l in zip(plan._grad_layout, last) if l < c)
        if final >= min_frac * total:
            return c, ranges_final_before(c)
    return None
