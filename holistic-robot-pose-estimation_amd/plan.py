"""Static op plan: the host-side replacement for eager per-layer dispatch.

A model's ``emit(pb, ...)`` methods describe the network ONCE per (batch size, dtype, mode) as a list
of kernel launches on pre-allocated NHWC buffers; running the plan is a straight walk over that list
(ctypes calls into libhrp_hip.so on the current HIP stream), so a whole forward+backward is
hipGraph-capturable and carries no per-step Python graph building, no autograd tape and no allocator
traffic.  The backward list is generated at build time from the forward list (reverse order, static
decision of "first writer overwrites / later writers accumulate" for every gradient buffer).

PyTorch is used for device memory (torch.zeros buffers), the stream handle and the public
``torch.autograd.Function`` boundary only.
"""
import collections
import contextlib
import ctypes as C
import math
import os

import torch

from . import _native as nv

BN_EPS = 1e-5
STAT_SLOTS = 8   # HRP_STAT_SLOTS in include/hrp.h


def _rup(a, b):
    return (a + b - 1) // b * b


def _dt(dtype):
    return nv.HRP_F32 if dtype == torch.float32 else nv.HRP_BF16


def _cdt(plan, dtype):
    """dtype code of a CONVOLUTION problem / of the weight packing: fp32 tensors of a plan in the 3 x bf16 product mode run as
    HRP_F32X3 (Plan.x3; include/hrp.h); every other launch of such a plan is plain HRP_F32."""
    if dtype == torch.float32:
        return nv.HRP_F32X3 if plan.x3 else nv.HRP_F32
    return nv.HRP_BF16


class TensorH:
    """NHWC activation handle: logical shape (N,H,W,C), channel pitch, backing buffer (+ gradient)."""

    def __init__(self, plan, N, H, W, Cc, dtype, buf=None, offset=0, pitch=None, base=None):
        self.plan, self.N, self.H, self.W, self.C, self.dtype = plan, N, H, W, Cc, dtype
        self.pitch = pitch if pitch is not None else _rup(Cc, 8)
        self.offset = offset  # element offset into buf (column slices)
        self.base = base      # parent handle when this is a slice
        if buf is None:
            buf = torch.zeros(N * H * W * self.pitch, dtype=dtype, device=plan.device)
            plan.keep.append(buf)
        self.buf = buf
        self._grad = None
        self.grad_written = False
        self.requires_grad = False
        self.stats = None      # (arena offset) of forward sum/sumsq when produced by a conv in train mode
        self.producer = None
        self.lane_path = plan.lane_path   # where it is produced; consumers in a concurrent lane are an error
        self._grad_paths = []
        self.n_readers = self.n_bn_readers = 0   # forward consumers / those of them that are batch-statistics BatchNorm terms

    @property
    def esz(self):
        return self.buf.element_size()

    def ptr(self):
        return self.buf.data_ptr() + self.offset * self.esz

    def grad_buf(self):
        root = self.base if self.base is not None else self
        if root._grad is None:
            root._grad = torch.zeros_like(root.buf)
            self.plan.keep.append(root._grad)
        return root._grad

    def gptr(self):
        return self.grad_buf().data_ptr() + self.offset * self.esz

    def take_grad_slot(self):
        """-> accumulate flag for a producer of this tensor's gradient."""
        acc = self.grad_written
        self.grad_written = True
        # allocate NOW (build time, on the building stream): launch closures call gptr() when they run, and a buffer
        # first created then would be zero-filled on torch's current stream - unordered against a lane's side stream,
        # where the fill could land after the first accumulations and wipe them (first step of a plan only)
        self.grad_buf()
        root = self.base if self.base is not None else self
        root.grad_written = True     # (a column slice's gradient lives in its root's buffer: the root's producer must run)
        self.plan.grad_owner[self.gptr()] = root      # who writes this gradient, for Plan._fuse_bn_reduce
        here = self.plan.lane_path
        if any(lanes_concurrent(here, q) for q in root._grad_paths):
            raise RuntimeError("plan: a gradient is accumulated from two concurrent lanes")
        root._grad_paths.append(here)
        return 1 if acc else 0

    def check_readable(self):
        """Forward consumers must be ordered after the producer (same lane, or outside its parallel block)."""
        if lanes_concurrent(self.plan.lane_path, self.lane_path):
            raise RuntimeError("plan: a tensor is consumed in a lane concurrent with its producer")
        self.materialize()
        self.n_readers += 1

    def materialize(self):
        """A block output whose forward pass (hrp_ew_fwd: relu(bn2(y2) + x)) was held back for the next block's conv1 prologue
        (PlanBuilder.conv_bn_relu_conv, pro_mode 3) and is read by something else after all: the pass runs here, in the reader's lane,
        right in front of the reader."""
        op = getattr(self, "pending_input", None)
        if op is not None:
            # an external image whose layout conversion waited for its reader (PlanBuilder.image_input): it runs in the reader's lane
            self.pending_input = None
            self.plan.pending_inputs.remove(self)
            self.lane_path = self.plan.lane_path
            self.plan.fwd.append(op)
        pend = getattr(self, "pending_block_end", None)
        if pend is not None:
            self.pending_block_end = None
            self.plan.fwd.append(pend["launch"])
            if self in self.plan.pending_block_ends:
                self.plan.pending_block_ends.remove(self)

    def view4(self):
        return self.buf.view(self.N, self.H, self.W, self.pitch)[..., :self.C]


class Term:
    """One summand of an element-wise op: tensor, optional BatchNorm module, upsample factor."""

    def __init__(self, t, bn=None, up=1):
        self.t, self.bn, self.up = t, bn, up


class ParamW:
    """Packed copies of one conv / linear weight."""

    def __init__(self, param, cout, cin, ntaps):
        self.param, self.cout, self.cin, self.ntaps = param, cout, cin, ntaps
        self.fwd_off = self.bwd_off = None
        self.pad_t = 0      # extra (zero) tap slots per chunk of the transposed packing (hrp_pack_entry.pad_t)
        self.t_ntaps = set()   # w_ntaps of every data-gradient descriptor that reads the transposed packing (Plan.finalize checks)
        self.grad_written = False
        self.first_use = None    # index into Plan.fwd at (or before) the first launch that reads the packed copy
        self.src = None          # fp32 tensor packed instead of `param` (derived layouts, e.g. the 4x4 form of the stem)


# ReLU sign bits written by the forward element-wise pass, read by its backward instead of the activation
RELU_BITMASK = True

# True: every lane runs on the caller's stream (per-kernel timing passes, A/B measurements)
SERIAL_LANES = bool(os.environ.get("HRP_SERIAL_LANES"))


# nesting depth of parallel blocks that really fork; deeper ones stay on their parent lane.  1 = flat: a lane
# forked from a forked lane crashes hipStreamEndCapture on ROCm 7 (eager multi-stream execution is fine)
MAX_LANE_DEPTH = 1


# How a plan executes.  "merged" (default): ONE stream; the lanes of every parallel block are walked in lock step and
# the launches of one kernel family that sit at the same position of their lanes - the same layer of every branch of
# both trunks - become ONE batched launch (hrp_batch_*, include/hrp.h).  "lanes": every lane is a HIP stream forked
# from / joined into its parent (round 1; still what the measurements of the batched launches are compared with).
# "hybrid": parallel blocks that asked for streams keep them (one per trunk), the virtual blocks inside each stream are
# merged - two chains of batched launches whose ramp-up / tail phases overlap.
PLAN_MODE = os.environ.get("HRP_PLAN_MODE", "hybrid")
# merged / hybrid: weight-gradient launches write their partial slabs only (descriptor phase 1); the slabs of up to 32
# layers are folded into the gradients by ONE launch at the end of their lane (or every WGRAD_FOLD_EVERY problems)
# instead of a 6-10 us launch behind every weight-gradient launch (~660 per step of the benchmark network)
# the BatchNorm-backward reduce pass of an activation whose gradient has ONE producer, a data-gradient convolution, runs in
# that convolution's epilogue (hrp_conv_desc.bnb_*) instead of as a launch of its own over the same two tensors
FUSE_BN_REDUCE = True
WGRAD_DEFER = True
WGRAD_FOLD_EVERY = 32
# weight gradients feed nothing before the optimizer: the launches of a lane are held back until WGRAD_SINK problems of one
# tap count are pending (or the lane ends) and then run as ONE batched launch - also the layers that have no lock-step
# partner (stem, layer1, transitions, the fuse convolutions).  0: every launch stays where the layer's backward put it.
# 16 since round 5: a launch writes one set of partial slabs per workgroup whatever it computes, so half as many launches of
# twice the problems write (and the folds read) half the slab bytes - 35.17 -> 34.52 ms per step (8 / 12 / 16 / 20 / 24 / 32:
# 35.9 / 35.9 / 35.4 / 35.3* / 35.5 / 35.6 on one box, * 35.3 against 34.5 for 16 on another)
WGRAD_SINK = int(os.environ.get("HRP_WGRAD_SINK", "16"))
# train-mode BasicBlock interiors conv -> BN -> ReLU -> conv on the row-strip kernel (csrc/conv_row.h): the BatchNorm + ReLU
# runs in the second convolution's staging path, its backward in the staging path of the first convolution's data gradient
# stride-2 data gradients: parity classes padded to 4 taps (PlanBuilder._conv_bwd)
PARITY_PAD = True
ROWCONV_FUSE = True
# ... and the block-end activation's backward (apply pass into conv2's data gradient, reduce pass into the next block's)
BLOCK_END_FUSE = True
BLOCK_END_REDUCE_FUSE = True
BLOCK_FUSE = os.environ.get("HRP_BLOCK_FUSE", "1") not in ("0", "")      # fused inference BasicBlock (csrc/conv_block.h)
# ... with the shortcut's gradient added by conv1's data gradient as a masked residual (one write of the block input's gradient)
MASKED_RES = True
# ... and the block-end FORWARD pass, out = relu(bn2(y2) + x), inside the NEXT block's conv1 (row-strip pro_mode 3: the raw y2 rows are
# staged, bn + shortcut + ReLU applied in place, `out` and its ReLU bits leave as side outputs): three of the four block ends of a
# branch stack lose their hrp_ew_fwd launch and conv1 no longer reads `out` back
BLOCK_END_FWD_FUSE = os.environ.get("HRP_BLOCK_END_FWD", "1") not in ("0", "")
# train-mode Bottlenecks at >= 131 072 pixels (layer1, the first incre-module of the cls head): conv3 (+ the 1x1 projection of the
# shortcut) + BatchNorm + shortcut + ReLU as pointwise launches that never store a raw 1x1 output, the BatchNorm backward likewise
# (PlanBuilder.bottleneck_tail)
BNECK_TAIL_FUSE = os.environ.get("HRP_TAIL_FUSE", "1") not in ("0", "")
# the backward of a fuse sum pools its output gradient once for all upsampled terms (hrp_ew_pool2; PlanBuilder._act_bwd): built for
# VERDICT r5 item 1(d), correct (tests/test_gpu_kernels.py::test_hr_module_fuse_pooled_gradients) and OFF: it trades ~0.8 GB of
# re-reads for 62 small launches in the lanes' serial chains - A/B/A/B on one box 32.85 / 33.00 ms with it, 32.84 / 32.80 without
# (reduce + apply 5.02 -> 4.63 ms of kernel time, the pool launches 0.47 ms).  HRP_POOL_FUSE=1 turns it on.
POOL_FUSE_GRADS = os.environ.get("HRP_POOL_FUSE", "0") not in ("0", "")
# external images are converted (NCHW fp32 / bytes -> NHWC) in the lane of their first reader (PlanBuilder.image_input)
LAZY_INPUTS = os.environ.get("HRP_LAZY_INPUTS", "1") not in ("0", "")
ARENA_ZERO_EARLY = True
BATCHING = True      # False (tests): merged mode without batching = the same launches one by one
# lanes of DIFFERENT launch sequences (the paths of a fuse layer) merge by their heads - the largest group of equal merge key first -
# instead of by position: 646 -> 628 conv launches, 34.12 -> 33.87 ms per step (A/B/A/B on one box)
GREEDY_MERGE = True
# development aid (set by tests / tools): batch only these families ({"conv", "wgrad", "ew_fwd", "ew_red", "ew_app"})
BATCH_FAMILIES = None
# The switches above without an environment variable are module constants: tests and tools patch them (plan.X = ...) before a
# plan is built.  The environment variables that remain are the ones a user of the library or of bench.py needs: HRP_PLAN_MODE,
# HRP_SERIAL_LANES, HRP_WGRAD_SINK, HRP_BLOCK_FUSE, HRP_PLAN_STATS, HRP_DBG_SYNC.
PLAN_STATS = bool(os.environ.get("HRP_PLAN_STATS"))      # print launch counts per family when a plan is finalised
DBG_SYNC = os.environ.get("HRP_DBG_SYNC")                 # development aid: device-wide sync after every op / every fork and join

# bumped whenever parameters / BatchNorm buffers are modified behind torch's back (see Plan.params_dirty)
# largest dilation of a 3x3 convolution the tile program runs as ONE problem (its halo tile must fit LDS); beyond it
# PlanBuilder._conv_shifted_taps splits the taps into one-tap problems
MAX_TILE_DILATION = 4

PARAM_EPOCH = 0
_PLAN_SERIAL = 0      # plans of this process that drew a dropout key (Plan.rng_state)


def bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


Entry = collections.namedtuple("Entry", "lane path op")   # lane None: fork / join marker of the lanes mode


class _OpList(list):
    """Launch list; ``append(op)`` tags the op with the lane the builder is emitting into."""

    def __init__(self, plan):
        super().__init__()
        self.plan = plan

    def append(self, op):
        list.append(self, Entry(self.plan.cur_lane, self.plan.lane_path, op))


# kernel family -> (hrp_batch_family, single-launch entry point, descriptor type)
FAMILIES = {"conv": (nv.BATCH_CONV, "hrp_conv2d_fwd", nv.ConvDesc),
            "wgrad": (nv.BATCH_WGRAD, "hrp_conv2d_bwd_weight", nv.WgradDesc),
            "ew_fwd": (nv.BATCH_EW_FWD, "hrp_ew_fwd", nv.EwDesc),
            "ew_red": (nv.BATCH_EW_BWD_REDUCE, "hrp_ew_bwd_reduce", nv.EwBwdDesc),
            "ew_app": (nv.BATCH_EW_BWD_APPLY, "hrp_ew_bwd_apply", nv.EwBwdDesc),
            "wgrad_fold": (nv.BATCH_WGRAD_FOLD, None, nv.WgradFoldDesc)}


class Launch:
    """One launch of a batchable family: (family, descriptor).  Callable like the plain closures of the launch lists."""
    __slots__ = ("fam", "desc")

    def __init__(self, fam, desc):
        self.fam, self.desc = fam, desc

    def __call__(self, s):
        nv.call(FAMILIES[self.fam][1], C.byref(self.desc), s)

    def launches(self):
        return [self]

    def fold_desc(self):
        """Phase-1 weight gradient launched on its own: the fold descriptor of the tiling the launcher will choose."""
        f = nv.WgradFoldDesc()
        nv.check(nv.lib().hrp_wgrad_fold_desc_of(C.byref(self.desc), C.byref(f)), "hrp_wgrad_fold_desc_of")
        return f

    def fold_descs(self):
        return [self.fold_desc()]

    def merge_key(self):
        """Launches with equal keys may share a batched launch; None: always alone."""
        d = self.desc
        if BATCH_FAMILIES is not None and self.fam not in BATCH_FAMILIES:
            return None
        esz = 2 if d.dtype == nv.HRP_BF16 else 4
        if self.fam == "conv":
            if d.ntaps not in (1, 2, 4, 9) or (d.Cin * esz) % 32 or (d.dtype != nv.HRP_BF16 and d.H == 1 and d.W == 1 and d.Cin >= 512):
                return None
            if d.ntaps == 1 and nv.lib().hrp_conv_pointwise(C.byref(d)):
                return None      # the pointwise kernel (csrc/conv_pw.h) exists as a single launch only
            return ("conv", d.dtype, d.ntaps)
        if self.fam == "wgrad":
            return ("wgrad", d.dtype, d.ntaps, d.reserved)
        if d.C % (16 // esz):
            return None
        return (self.fam, d.dtype)

    def written(self):
        """Device addresses this launch writes (two launches of one batch must not share any)."""
        d = self.desc
        if self.fam == "conv":
            main = d.y + ((d.out_off_y * d.y_W + d.out_off_x) * d.y_pitch if d.out_stride > 1 else 0)
            return tuple(a for a in (main, d.pro_side, d.pro_side2) if a)
        if self.fam in ("wgrad", "wgrad_fold"):
            return (d.dw + 4 * d.dw_tap_off,)
        if self.fam == "ew_fwd":
            return (d.out,)
        if self.fam == "ew_red":
            return (d.sums,)
        return tuple(x for x in (d.din, d.din2) if x)


class BatchLaunch:
    """Launches of one family (one per concurrent lane) as ONE launch; built by Plan._flatten, prepared at finalize."""

    def __init__(self, plan, items):
        self.plan, self.items = plan, list(items)
        self.fam = items[0].fam
        self.info, self.table, self.singles, self.folds = None, None, False, None

    def launches(self):
        return self.items

    def merge_key(self):
        return self.items[0].merge_key()

    def ws_query(self):
        """-> workspace bytes per problem (weight gradients)."""
        n = len(self.items)
        arr = (nv.WgradDesc * n)(*[it.desc for it in self.items])
        info = nv.BatchInfo()
        nv.check(nv.lib().hrp_batch_prepare(nv.BATCH_WGRAD, arr, n, None, C.byref(info)), "hrp_batch_prepare")
        return [int(info.ws_bytes[i]) for i in range(n)]

    def prepare(self):
        famid, _, dtype = FAMILIES[self.fam]
        n = len(self.items)
        arr = (dtype * n)(*[it.desc for it in self.items])     # copies: every pointer is final by now
        info = nv.BatchInfo()
        nbytes = int(nv.lib().hrp_batch_table_bytes(famid, n))
        host = (C.c_char * nbytes)()
        try:
            nv.check(nv.lib().hrp_batch_prepare(famid, arr, n, host, C.byref(info)), "hrp_batch_prepare")
        except nv.HrpError:
            if self.fam == "wgrad_fold":
                raise
            if PLAN_STATS:
                import sys
                print(f"plan: batch of {n} {self.fam} launches runs one by one ({nv.lib().hrp_last_error().decode()})", file=sys.stderr)
            self.singles = True      # not batchable after all (scalar path, tile does not fit ..): one by one
            return
        self.info = info
        if self.fam == "wgrad" and self.items[0].desc.phase == 1:
            folds = (nv.WgradFoldDesc * n)()
            nv.check(nv.lib().hrp_batch_wgrad_fold_descs(host, C.byref(info), folds), "hrp_batch_wgrad_fold_descs")
            self.folds = list(folds)
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.plan.device)
        self.plan.keep.append(self.table)

    def fold_descs(self):
        """Phase-1 weight gradients: the fold descriptors of the problems (after prepare)."""
        if not self.singles:
            return self.folds
        return [it.fold_desc() for it in self.items]

    def __call__(self, s):
        if self.singles:
            for it in self.items:
                it(s)
        else:
            nv.call_batch(self, s)


class BlockLaunch:
    """One fused inference BasicBlock (csrc/conv_block.h, descriptor nv.BlockDesc).  The lock-step merge pairs the 32- and the
    64-channel block of one position into one hrp_block_launch; alone it is a launch of one problem."""
    fam = "block"

    def __init__(self, desc):
        self.desc = desc
        self._single = None

    def launches(self):
        return [self]

    def merge_key(self):
        return ("block",)

    def written(self):
        return (self.desc.conv2.y,)

    def __call__(self, s):
        if self._single is None:
            self._single = BlockBatch(None, [self])
        self._single(s)


class BlockBatch:
    """hrp_block_launch of one problem, or of the 32-channel + the 64-channel block of one lock-step position."""
    fam = "block"

    def __init__(self, plan, items):
        self.plan, self.items = plan, sorted(items, key=lambda it: it.desc.conv1.Cin)
        self.info, self.table = None, None

    def launches(self):
        return self.items

    def merge_key(self):
        return None

    def prepare(self):
        n = len(self.items)
        arr = (nv.BlockDesc * n)(*[it.desc for it in self.items])      # copies: every pointer is final by now
        self.info = nv.BlockInfo()
        self.table = (C.c_char * int(nv.lib().hrp_block_table_bytes()))()
        nv.check(nv.lib().hrp_block_prepare(arr, n, self.table, C.byref(self.info)), "hrp_block_prepare")

    def __call__(self, s):
        if self.table is None:
            self.prepare()
        nv.call_block(self, s)


def _pair_block(plan, ops):
    """Fused inference blocks of one lock-step position -> launches of one 32-channel + one 64-channel problem (the two
    high-resolution branches of one trunk), leftovers alone."""
    items = [l for op in ops for l in op.launches()]
    c32 = [it for it in items if it.desc.conv1.Cin == 32]
    c64 = [it for it in items if it.desc.conv1.Cin != 32]
    out = []
    while c32 and c64:
        out.append(BlockBatch(plan, [c32.pop(0), c64.pop(0)]))
    out += [BlockBatch(plan, [it]) for it in c32 + c64]
    return out


def _merge_ops(plan, ops):
    """Ops of equal merge key -> batched launches of at most BATCH_MAX problems with pairwise distinct outputs."""
    items = [l for op in ops for l in op.launches()]
    out, cur, seen = [], [], set()
    for it in items:
        w = it.written()
        if len(cur) == nv.BATCH_MAX or any(a in seen for a in w):
            out.append(cur)
            cur, seen = [], set()
        cur.append(it)
        seen.update(w)
    out.append(cur)
    return [g[0] if len(g) == 1 else BatchLaunch(plan, g) for g in out]


class _Seq:
    def __init__(self):
        self.items, self.blocks = [], {}


class _Par:
    def __init__(self):
        self.lanes = {}


class _LaneSync:
    """Fork (children wait for the parent lane) or join (parent waits for the children) between lanes."""

    def __init__(self, kind, parent, children):
        self.kind, self.parent, self.children = kind, parent, children

    def run(self, streams):
        for c in self.children:
            if self.kind == "fork":
                streams[c].wait_stream(streams[self.parent])
            else:
                streams[self.parent].wait_stream(streams[c])


class _PackJoin:
    """Forward-list entry: the main lane waits for the side stream that packs the late weights (Plan.run_prep)."""

    def __init__(self, plan):
        self.plan = plan

    def run(self, streams):
        if self.plan._pack_stream is not None:
            streams[0].wait_stream(self.plan._pack_stream)

    def __call__(self, s):   # merged mode: an ordinary entry of the flat list
        if self.plan._pack_stream is not None and not SERIAL_LANES:
            torch.cuda.current_stream(self.plan.device).wait_stream(self.plan._pack_stream)


def lanes_concurrent(a, b):
    """Lane paths are tuples of (parallel block id, lane index); two emit positions may run at the same time
    iff their paths split inside one block."""
    for (ba, la), (bb, lb) in zip(a, b):
        if ba != bb:
            return False
        if la != lb:
            return True
    return False


class Plan:
    def __init__(self, device, dtype, training, need_grad):
        self.device, self.dtype, self.training, self.need_grad = device, dtype, training, need_grad
        self.x3 = False            # fp32 plan whose convolutions form their products as three bf16 MFMAs on split operands
        self.keep = []
        # lanes: independent sub-graphs (the two backbones, the branches of an HRNet module) are emitted into
        # different lanes = HIP streams, forked from / joined into their parent lane; lane 0 is the caller's
        # stream.  Captured into a HIP graph the lanes become parallel graph branches.
        self.cur_lane, self.lane_path, self.n_lanes = 0, (), 1
        self._lane_ids, self._n_blocks, self._side_streams = {}, 0, []
        self.fwd, self.bwd = _OpList(self), _OpList(self)
        self.weights = {}          # id(param) -> ParamW
        self.weight_list = []
        self.bn_train = []         # (bn module, stats offset, count)
        self.bn_fold = {}          # id(bn) -> (scale tensor, shift tensor)
        self.bn_fold_hat = {}      # id(bn) -> (bn, invstd, -mean * invstd) of the RUNNING statistics: xhat = x * a + b (eval-mode BatchNorm backward)
        self.bn_bwd = []           # (bn module, bwd sums offset)
        self.stats_floats = 0
        self.bsums_floats = 0
        self.param_grads = {}      # id(param) -> (param, grad tensor)
        self.inputs, self.outputs = {}, {}
        self.dyn = {}              # name -> current external tensor (bound per call)
        self.built = False
        self._versions = None
        self._late = []
        self.out_handles = []      # TensorH whose gradient is seeded from outside
        self.grad_arena = None
        self._grad_views = {}
        self._grad_layout = []
        self.bwd_split, self.split_active = None, False
        self.bwd_cuts = None       # k cut positions of the backward list (PlannedModule.enable_split_backward(fracs=...))
        self.pre_pack = []         # ops run before the weight packing of every step (derived weight layouts)
        self.counters = {}         # build statistics (HRP_PLAN_STATS=1 prints them at finalize)
        self.wgrad_ws_bytes = {}   # lane -> scratch bytes shared by that lane's weight-gradient launches
        self.wgrad_ws = {}
        self.merged, self.fwd_run, self.bwd_run = False, [], []
        self._block_lanes = {}     # parallel block id -> stream of each lane (None: virtual block)
        self._rng, self.n_dropout = None, 0
        self.linear_grad_written = set()
        self.grad_owner = {}       # gradient buffer address -> TensorH root (every producer registers through take_grad_slot)
        self.reg_chains = []       # PlanBuilder.regressors: mask buffer and saved operands of every fused regressor chain
        self.pending_inputs = []       # external images whose conversion launch waits for the first reader (TensorH.materialize)
        self.pending_block_ends = []   # block outputs whose forward pass waits for a consumer (TensorH.materialize)
        self.row_last_writer = {}  # gradient buffer address -> (row-strip conv descriptor that completes it, lane path, producers of the gradient so far)

    # ---- build-time helpers -------------------------------------------------------------------
    def new(self, N, H, W, Cc, dtype=None, pitch=None):
        return TensorH(self, N, H, W, Cc, dtype or self.dtype, pitch=pitch)

    def weight(self, param, cout, cin, ntaps):
        w = self.weights.get(id(param))
        if w is None:
            w = ParamW(param, cout, cin, ntaps)
            self.weights[id(param)] = w
            self.weight_list.append(w)
        if w.first_use is None:
            w.first_use = len(self.fwd)   # every forward consumer looks the weight up right before it emits
        return w

    def preallocate_param_grads(self, params):
        """One flat fp32 arena for every parameter gradient (what the data-parallel all-reduce walks in
        large buckets); .grad tensors are views into it."""
        params = [p for p in params if p.requires_grad]
        # conv / linear weights first, in registration (= forward) order, then the vectors (BN affine parameters, biases):
        # the weight gradients of a trunk's late stages - final early in the backward - are then one contiguous range
        # (analyze_backward_split), not interleaved with BatchNorm gradients that one launch writes at the very end
        params = [p for p in params if p.dim() > 1] + [p for p in params if p.dim() <= 1]
        total = sum(_rup(p.numel(), 4) for p in params)
        self.grad_arena = torch.zeros(max(total, 4), dtype=torch.float32, device=self.device)
        off = 0
        self._grad_layout = []     # (offset, numel) in arena order
        for p in params:
            self._grad_views[id(p)] = self.grad_arena[off:off + p.numel()].view(p.shape)
            self._grad_views[id(p)]._hrp_plan_grad = True
            self._grad_layout.append((off, p.numel()))
            off += _rup(p.numel(), 4)

    def grad_of_param(self, p):
        e = self.param_grads.get(id(p))
        if e is None:
            g = self._grad_views.get(id(p))
            if g is None:
                g = torch.zeros_like(p, dtype=torch.float32)
            e = (p, g)
            self.param_grads[id(p)] = e
            self.keep.append(e[1])
        return e[1]

    def rng_state(self):
        """Device-side (seed, step) of the plan's dropout masks; the seed comes from torch's generator at build time."""
        if self._rng is None:
            # torch's seed, decorrelated per data-parallel rank (ranks launched with one manual_seed must not draw the same
            # masks for different data) and per training plan of the process (a second plan - another batch shape, a plan
            # re-created after a cache eviction - must not replay the first one's mask sequence)
            global _PLAN_SERIAL
            _PLAN_SERIAL += 1
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            seed = (int(torch.initial_seed()) ^ (rank << 48) ^ (_PLAN_SERIAL << 32)) & 0x7FFFFFFFFFFFFFFF
            self._rng = torch.tensor([seed, 0], dtype=torch.int64, device=self.device)
            self.keep.append(self._rng)
        return self._rng

    def alloc_stats(self, Cc):
        off = self.stats_floats
        self.stats_floats += 2 * Cc * STAT_SLOTS
        return off

    def alloc_bsums(self, Cc):
        off = self.bsums_floats
        self.bsums_floats += 2 * Cc * STAT_SLOTS
        return off

    # ---- finalisation -----------------------------------------------------------------------------
    def finalize(self):
        dev = self.device
        self.stats = torch.zeros(max(self.stats_floats, 2), dtype=torch.float64, device=dev)       # fp64 slots (include/hrp.h)
        self.bsums = torch.zeros(max(self.bsums_floats, 2), dtype=torch.float64, device=dev)
        # packed weights: separate arenas per element type.  Training plans with parallel blocks pack in two parts:
        # the weights of the first block (stem, layer1) on the main stream, the rest (99 % of the bytes) on a side
        # stream that the forward list joins right after that block - the 0.7 ms gather runs under the stem instead of
        # in front of it.  `cut` = index in fwd of the join.
        self._pack_tables, self._pack_tables_late = [], []
        for w in self.weight_list:
            # the tap stride of the transposed packing is ntaps + pad_t for EVERY data gradient of the weight: a use that was
            # emitted with another w_ntaps (a weight shared by a padded and an unpadded stride-2 layer) would read shifted taps
            assert all(t == w.ntaps + w.pad_t for t in w.t_ntaps), f"plan: data gradients of one weight disagree on its tap slots ({w.t_ntaps}, pad_t {w.pad_t})"
        cut = self._late_pack_cut()
        for dtype in (torch.float32, torch.bfloat16):
            ws = [w for w in self.weight_list if w.dtype == dtype]
            if not ws:
                continue
            esz = 4 if dtype == torch.float32 else 2
            ck = 32 // esz   # channels per 32-byte K chunk (csrc/conv_fwd.hip ROW, core.hip pack kernel)
            total = 0
            for w in ws:
                nf = math.ceil(w.cin_used / ck) * w.ntaps * _rup(w.cout, 32) * ck
                w.fwd_off = total
                total += _rup(nf, 64)
                w.max_elems = nf
                if w.need_t:
                    nb = math.ceil(w.cout / ck) * (w.ntaps + w.pad_t) * _rup(w.cin_used, 32) * ck
                    w.bwd_off = total
                    total += _rup(nb, 64)
                    w.max_elems = max(nf, nb)
            arena = torch.zeros(total, dtype=dtype, device=dev)
            self.keep.append(arena)
            for w in ws:
                w.arena = arena
            early = [w for w in ws if cut is None or w.first_use is None or w.first_use < cut]
            late = [w for w in ws if not (cut is None or w.first_use is None or w.first_use < cut)]
            for group, dest in ((early, self._pack_tables), (late, self._pack_tables_late)):
                if not group:
                    continue
                tab = (nv.PackEntry * len(group))()
                for i, w in enumerate(group):
                    tab[i].src = (w.src if w.src is not None else w.param).data_ptr()
                    tab[i].dst = arena.data_ptr() + w.fwd_off * esz
                    tab[i].dst_t = (arena.data_ptr() + w.bwd_off * esz) if w.need_t else None
                    tab[i].Cout, tab[i].Cin, tab[i].ntaps, tab[i].pad_t = w.cout, w.cin, w.ntaps, w.pad_t
                tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
                # compact grid: entry i owns the workgroups first[i] .. first[i + 1] - 1 (hrp_pack_weights_compact)
                cdt = _cdt(self, dtype)
                first = [0]
                for w in group:
                    first.append(first[-1] + nv.lib().hrp_pack_blocks(w.cout, w.cin, w.ntaps, cdt, 1, 1 if w.need_t else 0))
                fdev = torch.tensor(first, dtype=torch.int32, device=dev)
                self.keep += [tdev, fdev]
                dest.append((tdev, fdev, len(group), first[-1], cdt))
        self._pack_stream = None
        self._arena_clean = False      # the gradient arena was zeroed by this forward's run_prep (side stream)
        if self._pack_tables_late:
            self._pack_stream = torch.cuda.Stream(device=dev)
            list.insert(self.fwd, cut, Entry(None, (), _PackJoin(self)))
        self.merged = PLAN_MODE in ("merged", "hybrid")
        if not self.merged:
            self.wgrad_ws = {lane: torch.zeros(max(nb // 4, 4), dtype=torch.float32, device=dev)
                             for lane, nb in self.wgrad_ws_bytes.items()}
        # resolve deferred pointers (merged: the weight-gradient scratch is assigned below, once the batches are known)
        for fn in self._late:
            fn()
        self._late = []
        if FUSE_BN_REDUCE and self.need_grad:
            self._fuse_bn_reduce()
        if self.merged:
            # lock-step merge of the virtual lanes into batched launches; streams only where a block asked for them
            self.fwd_run, self.bwd_run = self._flatten(self.fwd), self._flatten(self.bwd)
            if WGRAD_SINK > 1 and WGRAD_DEFER and BATCHING:
                self.bwd_run = self._sink_wgrads(self.bwd_run)
            ops = [e.op for e in self.fwd_run + self.bwd_run if e.lane is not None]
            batches = [op for op in ops if isinstance(op, BatchLaunch)]
            wg_ops = [e for e in self.bwd_run if isinstance(e.op, (Launch, BatchLaunch)) and e.op.fam == "wgrad"]
            defer = WGRAD_DEFER and bool(wg_ops)
            # (descriptor.reserved == 1: a gradient some later launch of the list reads - folded on the spot)
            now = {id(e.op) for e in wg_ops if any(it.desc.reserved for it in e.op.launches())}
            if defer:
                # deferred folds: every launch keeps its slabs until its lane folds them, so every problem gets its own
                # scratch region (one bump allocation over the whole backward: ~4 GB for the benchmark network at B=64)
                total, ws_off = 0, {}
                for e in wg_ops:
                    op = e.op
                    for it in op.launches():
                        it.desc.phase = 0 if id(op) in now else 1
                    if isinstance(op, BatchLaunch):
                        op.ws = op.ws_query()
                    else:
                        op_need = int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(op.desc)))
                        op.desc.workspace_bytes = op_need
                    sizes = op.ws if isinstance(op, BatchLaunch) else [op.desc.workspace_bytes]
                    # (a batch that falls back to single launches: each item then wants its single-launch size)
                    singles = [int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(it.desc))) for it in op.launches()]
                    ws_off[id(op)] = []
                    for b, b1 in zip(sizes, singles):
                        ws_off[id(op)].append((total, max(b, b1)))
                        total += _rup(max(b, b1), 256)
                self.wgrad_ws = {0: torch.zeros(max(total // 4, 4), dtype=torch.float32, device=dev)}
                base = self.wgrad_ws[0].data_ptr()
                for e in wg_ops:
                    for it, (off, b) in zip(e.op.launches(), ws_off[id(e.op)]):
                        it.desc.workspace, it.desc.workspace_bytes = (base + off, b) if b else (None, 0)
                for op in batches:
                    op.prepare()
                self.bwd_run = self._insert_folds(self.bwd_run)
            else:
                # weight-gradient scratch: one buffer per stream, every launch of that stream uses it in turn
                need = {}
                for e in self.fwd_run + self.bwd_run:
                    op = e.op
                    if isinstance(op, BatchLaunch) and op.fam == "wgrad":
                        op.ws = op.ws_query()
                        need[e.lane] = max(need.get(e.lane, 0), sum(_rup(b, 256) for b in op.ws))
                    elif isinstance(op, Launch) and op.fam == "wgrad":
                        need[e.lane] = max(need.get(e.lane, 0), int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(op.desc))))
                self.wgrad_ws = {lane: torch.zeros(max(nb // 4, 4), dtype=torch.float32, device=dev) for lane, nb in need.items()}
                for e in self.fwd_run + self.bwd_run:
                    op = e.op
                    if isinstance(op, BatchLaunch) and op.fam == "wgrad":   # problems of one launch run concurrently: disjoint regions
                        off = 0
                        for it, b in zip(op.items, op.ws):
                            it.desc.workspace, it.desc.workspace_bytes = self.wgrad_ws[e.lane].data_ptr() + off, b
                            off += _rup(b, 256)
                    elif isinstance(op, Launch) and op.fam == "wgrad":
                        op.desc.workspace, op.desc.workspace_bytes = self.wgrad_ws[e.lane].data_ptr(), self.wgrad_ws[e.lane].numel() * 4
                for op in batches:
                    op.prepare()
                    if op.fam == "wgrad" and op.singles:
                        lane = next(e.lane for e in self.fwd_run + self.bwd_run if e.op is op)
                        for it in op.items:
                            it.desc.workspace, it.desc.workspace_bytes = self.wgrad_ws[lane].data_ptr(), self.wgrad_ws[lane].numel() * 4
            used = {e.lane for e in self.fwd_run + self.bwd_run if e.lane is not None}
            self._side_streams = [torch.cuda.Stream(device=dev) if (i + 1) in used else None for i in range(self.n_lanes - 1)]
        else:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_lanes - 1)]
        self._bn_tables()
        self.built = True
        if PLAN_STATS:
            import sys
            msg = f"plan: {len(self.fwd)} forward / {len(self.bwd)} backward ops, {self.n_lanes} lanes, {self.counters}"
            if self.merged:
                ops = [e.op for e in self.fwd_run + self.bwd_run if e.lane is not None]
                nb = sum(isinstance(op, BatchLaunch) and not op.singles for op in ops)
                msg += (f"; {PLAN_MODE}: {sum(e.lane is not None for e in self.fwd_run)} / {sum(e.lane is not None for e in self.bwd_run)}"
                        f" launches, {nb} of them batched, streams {sorted({e.lane for e in self.fwd_run if e.lane is not None})}")
            print(msg, file=sys.stderr)


    # ---- passes over the launch lists (plan_passes.py) -------------------------------------------------------------------
    def _flatten(self, entries):
        from . import plan_passes
        return plan_passes.flatten(self, entries)

    def _fuse_bn_reduce(self):
        from . import plan_passes
        return plan_passes.fuse_bn_reduce(self)

    def _sink_wgrads(self, entries):
        from . import plan_passes
        return plan_passes.sink_wgrads(self, entries)

    def _insert_folds(self, entries):
        from . import plan_passes
        return plan_passes.insert_folds(self, entries)

    def analyze_backward_split(self, min_frac=0.55, fracs=None):
        from . import plan_passes
        return plan_passes.analyze_backward_split(self, min_frac, fracs)

    def _late_pack_cut(self):
        """Index in self.fwd right after the first parallel block (the join back into the main lane), or None when the
        plan is not a training plan with at least two parallel blocks and a network's worth of weights."""
        if not self.need_grad or len(self.weight_list) < 64:
            return None
        joins = [i for i, e in enumerate(self.fwd) if e.lane is None and getattr(e.op, "kind", None) == "join"]
        forks = [i for i, e in enumerate(self.fwd) if e.lane is None and getattr(e.op, "kind", None) == "fork"]
        if len(forks) < 2 or not joins or forks[1] < joins[0]:
            return None           # (nested or single blocks: keep the simple order)
        return joins[0] + 1

    def patch_wgrad_ws(self, g, lane):
        """lanes mode: the lane's weight-gradient scratch (merged / hybrid modes assign it after the merge)."""
        if lane in self.wgrad_ws:
            g.workspace, g.workspace_bytes = self.wgrad_ws[lane].data_ptr(), self.wgrad_ws[lane].numel() * 4

    def late(self, fn):
        """Defer pointer patching until the arenas exist (finalize)."""
        self._late.append(fn)

    def _table(self, entries):
        tab = (nv.BnEntry * len(entries))()
        for i, e in enumerate(entries):
            for k, v in e.items():
                setattr(tab[i], k, v)
        tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
        self.keep.append(tdev)
        return tdev, len(entries)

    def _bn_tables(self):
        self._run_tab = self._fold_tab = self._pgrad_tab = None
        if self.bn_train:
            ents = []
            for bn, off, count in self.bn_train:
                ents.append(dict(stats=self.stats.data_ptr() + 8 * off, a=bn.running_mean.data_ptr(),
                                 b=bn.running_var.data_ptr(), counter=bn.num_batches_tracked.data_ptr(),
                                 C=bn.num_features, count=float(count), momentum=bn.momentum, eps=bn.eps))
            self._run_tab = self._table(ents)
        if self.bn_fold:
            ents = []
            for bn, sc, sh in self.bn_fold.values():
                ents.append(dict(a=bn.weight.data_ptr(), b=bn.bias.data_ptr(), c=bn.running_mean.data_ptr(),
                                 d=bn.running_var.data_ptr(), out_scale=sc.data_ptr(), out_shift=sh.data_ptr(),
                                 C=bn.num_features, eps=bn.eps))
            if self.bn_fold_hat:
                # the same fold with gamma = 1, beta = 0: xhat = x * invstd - mean * invstd (dgamma of an eval-mode BatchNorm)
                cmax = max(bn.num_features for bn, _, _ in self.bn_fold_hat.values())
                ones, zeros = torch.ones(cmax, device=self.device), torch.zeros(cmax, device=self.device)
                self.keep += [ones, zeros]
                for bn, inv, nmi in self.bn_fold_hat.values():
                    ents.append(dict(a=ones.data_ptr(), b=zeros.data_ptr(), c=bn.running_mean.data_ptr(),
                                     d=bn.running_var.data_ptr(), out_scale=inv.data_ptr(), out_shift=nmi.data_ptr(),
                                     C=bn.num_features, eps=bn.eps))
            self._fold_tab = self._table(ents)
        if self.bn_bwd:
            ents = []
            for bn, off in self.bn_bwd:
                ents.append(dict(stats=self.bsums.data_ptr() + 8 * off, a=self.grad_of_param(bn.weight).data_ptr(),
                                 b=self.grad_of_param(bn.bias).data_ptr(), C=bn.num_features, accumulate=0))
            self._pgrad_tab = self._table(ents)

    # ---- execution ----------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def params_dirty(self):
        # tensor versions catch torch-side updates (torch.optim, load_state_dict); PARAM_EPOCH the updates made through
        # raw pointers (FusedClipAdam.step, hrp_bn_running_update), which do not bump ._version
        v = tuple(w.param._version for w in self.weight_list) + \
            tuple(t._version for bn, _, _ in self.bn_fold.values() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + \
            (PARAM_EPOCH,)
        if v != self._versions:
            self._versions = v
            return True
        return False

    def run_prep(self, force=False):
        if not (force or self.params_dirty()):
            return
        s = self._stream()
        for op in self.pre_pack:
            op(s)
        late_s = s
        if self._pack_stream is not None and not SERIAL_LANES:   # (serial runs skip the join entry of the forward list)
            self._pack_stream.wait_stream(torch.cuda.current_stream(self.device))
            late_s = self._pack_stream.cuda_stream
        for tdev, fdev, n, nblk, dt in self._pack_tables_late:
            nv.call("hrp_pack_weights_compact", tdev.data_ptr(), fdev.data_ptr(), n, nblk, dt, late_s)
        if late_s != s and self.need_grad and self.grad_arena is not None and ARENA_ZERO_EARLY:
            # the gradient arena's memset (228 MB for the two-trunk network, 40 us) rides on the side stream under the stems instead of
            # standing between the loss and the backward, where nothing overlaps it; run_backward zeroes itself when this did not run
            with torch.cuda.stream(self._pack_stream):
                self.grad_arena.zero_()
            self._arena_clean = True
        for tdev, fdev, n, nblk, dt in self._pack_tables:
            nv.call("hrp_pack_weights_compact", tdev.data_ptr(), fdev.data_ptr(), n, nblk, dt, s)
        if self._fold_tab:
            nv.call("hrp_bn_fold", self._fold_tab[0].data_ptr(), self._fold_tab[1], s)

    def lane_id(self, parent, index):
        """Stable lane number of the index-th extra child of a parent lane."""
        key = (parent, index)
        if key not in self._lane_ids:
            self._lane_ids[key] = self.n_lanes
            self.n_lanes += 1
        return self._lane_ids[key]

    def _run_list(self, ops):
        if SERIAL_LANES:
            h = self._stream()
            for lane, _, op in ops:
                if lane is not None:
                    op(h)
            return
        streams = [torch.cuda.current_stream(self.device)] + self._side_streams
        handles = [st.cuda_stream if st is not None else None for st in streams]
        dbg = DBG_SYNC
        for lane, _, op in ops:
            if lane is None:
                op.run(streams)
                if dbg:
                    torch.cuda.synchronize(self.device)
            else:
                if dbg == "pre1" and lane == 1:
                    torch.cuda.synchronize(self.device)
                op(handles[lane])
                if dbg == "all" or (dbg == "lane1" and lane == 1) or (dbg == "lane0" and lane == 0):
                    torch.cuda.synchronize(self.device)
                elif dbg == "s1" and lane == 1:
                    streams[1].synchronize()

    def fwd_ops(self):
        return self.fwd_run if self.merged else self.fwd

    def bwd_ops(self):
        return self.bwd_run if self.merged else self.bwd

    def run_forward(self):
        global PARAM_EPOCH
        s = self._stream()
        if self._rng is not None:
            nv.call("hrp_rng_advance", self._rng.data_ptr(), s)
        if self.training and self._run_tab:
            PARAM_EPOCH += 1       # running statistics change through raw pointers below: eval plans must refold
        if self.stats_floats:
            self.stats.zero_()
        self._run_list(self.fwd_ops())
        if self._run_tab:
            nv.call("hrp_bn_running_update", self._run_tab[0].data_ptr(), self._run_tab[1], s)

    def run_backward(self, part=None):
        """part None: the whole backward.  "first" / "rest": the two halves around self.bwd_split (a top-level position
        of the launch list chosen by analyze_backward_split) - the data-parallel step all-reduces the gradients that
        are final after the first half while the second half runs.  ("seg", j): segment j of the k + 1 segments around the k
        cuts self.bwd_cuts ("first" == ("seg", 0) when cuts are set)."""
        s = self._stream()
        cuts = self.bwd_cuts if self.bwd_cuts else ([self.bwd_split] if self.bwd_split is not None else [])
        if part == "first":
            part = ("seg", 0)
        if part == "rest":
            assert len(cuts) == 1, "run_backward('rest') is the two-segment form"
            part = ("seg", 1)
        j = None if part is None else part[1]
        if j in (None, 0):
            if self.grad_arena is not None and not self._arena_clean:
                self.grad_arena.zero_()   # one memset; every weight / bias gradient kernel then accumulates
            self._arena_clean = False
            if self.bsums_floats:
                self.bsums.zero_()
        ops = self.bwd_ops()
        if j is None:
            self._run_list(ops)
        else:
            lo = 0 if j == 0 else cuts[j - 1]
            hi = cuts[j] if j < len(cuts) else None
            self._run_list(list.__getitem__(ops, slice(lo, hi)))
        if (j is None or j == len(cuts)) and self._pgrad_tab:
            nv.call("hrp_bn_param_grad", self._pgrad_tab[0].data_ptr(), self._pgrad_tab[1], s)


    def publish_param_grads(self):
        """Hand the plan-owned gradient buffers to the parameters (torch semantics: .grad holds this
        backward's gradient; an already-present foreign .grad tensor is accumulated into)."""
        for p, g in self.param_grads.values():
            # (a .grad that is ANOTHER plan's arena view - a trailing partial batch built a second training plan - holds
            # that plan's last gradient, not something to accumulate onto: this backward's gradient replaces it)
            if p.grad is None or p.grad is g or getattr(p.grad, "_hrp_plan_grad", False):
                p.grad = g
            else:
                p.grad.add_(g)


# =====================================================================================================
# Plan builder: the vocabulary modules use in emit()
# =====================================================================================================
_TAPS3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]


class _BwdStack(list):
    """Backward emitters remember the lane they were registered in."""

    def __init__(self, plan):
        super().__init__()
        self.plan = plan

    def append(self, fn):
        list.append(self, (self.plan.cur_lane, self.plan.lane_path, fn))


class _Parallel:
    def __init__(self, pb, block, lanes):
        self.pb, self.block, self.lanes = pb, block, lanes

    @contextlib.contextmanager
    def lane(self, i):
        p = self.pb.plan
        saved = (p.cur_lane, p.lane_path)
        p.cur_lane, p.lane_path = self.lanes[i], saved[1] + ((self.block, i),)
        try:
            yield
        finally:
            p.cur_lane, p.lane_path = saved


from .plan_blocks import BlockOps    # noqa: E402  (the mixins read this module's names at call time)
from .plan_heads import HeadOps      # noqa: E402


class PlanBuilder(BlockOps, HeadOps):
    def __init__(self, plan):
        self.plan = plan
        self.bwd_stack = _BwdStack(plan)   # emitters of backward ops, in forward order (run reversed at the end)
        self.fuse_inference = (not plan.training) and (not plan.need_grad)

    # ---- inputs / outputs ---------------------------------------------------------------------------
    def image_input(self, name, N, Cc, H, W, u8=False, dtype=None, lazy=False):
        """NCHW fp32 external tensor -> NHWC plan tensor (channel-padded to 8).  u8: the external tensor holds the
        dataset's bytes and is divided by 255 on the way in (reference lib/core/function.py:26,29).  dtype: element type of
        the plan tensor when it is not the plan's (a trunk that computes in fp32 inside a bf16 plan)."""
        p = self.plan
        t = p.new(N, H, W, Cc, dtype)
        dt = _dt(t.dtype)

        def op(s):
            x = p.dyn[name]
            if u8:
                assert x.dtype == torch.uint8
                nv.call("hrp_u8_nchw_to_nhwc", x.data_ptr(), t.ptr(), dt, N, Cc, H, W, t.pitch, 255.0, 0, s)
            else:
                nv.call("hrp_nchw_to_nhwc", x.data_ptr(), t.ptr(), dt, N, Cc, H, W, t.pitch, s)
        if lazy and LAZY_INPUTS:
            # (lazy: the caller's first reader is a convolution - an op that announces its reads with check_readable)
            # the conversion is emitted in front of the first reader, in ITS lane: the images of two trunks are converted side by side
            # inside the trunks' lanes instead of one after the other in front of the fork
            t.pending_input = op
            p.pending_inputs.append(t)
        else:
            p.fwd.append(op)
        t.external = name
        return t

    def image_input_s2d(self, name, N, Cc, H, W, u8=False):
        """NCHW fp32 (or, u8, byte / 255) external image -> NHWC 2x2 space-to-depth tensor [N, H/2, W/2, 4*Cc]
        (ResNet stem input)."""
        p = self.plan
        t = p.new(N, (H + 1) // 2, (W + 1) // 2, 4 * Cc, pitch=_rup(4 * Cc, 16))
        dt = _dt(t.dtype)

        def op(s):
            x = p.dyn[name]
            if u8:
                assert x.dtype == torch.uint8
                nv.call("hrp_u8_nchw_to_nhwc", x.data_ptr(), t.ptr(), dt, N, Cc, H, W, t.pitch, 255.0, 1, s)
            else:
                nv.call("hrp_nchw_to_nhwc_s2d", x.data_ptr(), t.ptr(), dt, N, Cc, H, W, t.pitch, s)
        p.fwd.append(op)
        t.external = name
        return t

    def stem7x7_s2d(self, xs, weight, want_stats=False):
        """Conv2d(3, Cout, 7, stride 2, padding 3, bias=False) (Resnet.py:21) on the space-to-depth image `xs`:
        out[oy] = sum_ky w[ky] x[2 oy + ky - 3]; with ky + 1 = 2 ty + dy this is a 4x4 stride-1 convolution over the
        12 channels (dy, dx, c) with tap offsets ty - 2, tx - 2 and w'[co][(dy,dx,c)][ty,tx] = w[co][c][2ty+dy-1][2tx+dx-1]
        (zero where the 8x8 embedding has no 7x7 entry).  The weight gradient comes back through the same index map."""
        p = self.plan
        cout, cin = weight.shape[0], weight.shape[1]
        assert tuple(weight.shape[2:]) == (7, 7) and xs.C == 4 * cin
        dev, dtype = p.device, xs.dtype
        # index maps (host, once): w' element -> flat index into w (or -1), w element -> flat index into dW'
        idx_w = torch.full((cout, 4 * cin, 16), -1, dtype=torch.int32)
        idx_g = torch.zeros((cout, cin, 49), dtype=torch.int32)
        for dy in range(2):
            for dx in range(2):
                for ty in range(4):
                    for tx in range(4):
                        ky, kx = 2 * ty + dy - 1, 2 * tx + dx - 1
                        if ky < 0 or kx < 0:
                            continue
                        for c in range(cin):
                            q = (dy * 2 + dx) * cin + c
                            for co in range(cout):
                                idx_w[co, q, ty * 4 + tx] = (co * cin + c) * 49 + ky * 7 + kx
                                idx_g[co, c, ky * 7 + kx] = (co * 4 * cin + q) * 16 + ty * 4 + tx
        idx_w, idx_g = idx_w.to(dev), idx_g.to(dev)
        w12 = torch.zeros(cout, 4 * cin, 4, 4, dtype=torch.float32, device=dev)
        gw12 = torch.zeros(cout, 4 * cin, 16, dtype=torch.float32, device=dev)
        p.keep += [idx_w, idx_g, w12, gw12]
        p.pre_pack.append(lambda s: nv.call("hrp_gather_f32", weight.data_ptr(), idx_w.data_ptr(), w12.data_ptr(),
                                            w12.numel(), 0, s))
        w = ParamW(weight, cout, 4 * cin, 16)
        w.first_use = len(p.fwd)
        w.src, w.dtype, w.cin_used, w.need_t = w12, dtype, 4 * cin, False
        p.weights[("s2d", id(weight))] = w
        p.weight_list.append(w)
        y = p.new(xs.N, xs.H, xs.W, cout, dtype)
        y.requires_grad = p.need_grad
        taps = [(ty - 2, tx - 2) for ty in range(4) for tx in range(4)]
        vec = 8 if dtype == torch.bfloat16 else 4
        d = nv.ConvDesc()
        d.x, d.y, d.dtype = xs.ptr(), y.ptr(), _cdt(p, dtype)
        d.N, d.H, d.W, d.Cin, d.x_pitch = xs.N, xs.H, xs.W, _rup(xs.C, vec), xs.pitch
        d.Ho, d.Wo, d.Cout = y.H, y.W, cout
        d.y_H, d.y_W, d.y_pitch, d.res_pitch = y.H, y.W, y.pitch, y.pitch
        d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 1, 16, 16, _rup(cout, 32)
        for i, (a, b) in enumerate(taps):
            d.dy[i], d.dx[i], d.wtap[i] = a, b, i
        if want_stats:
            y.stats = p.alloc_stats(cout)
        esz = 4 if dtype == torch.float32 else 2

        def late():
            d.w = w.arena.data_ptr() + w.fwd_off * esz
            if y.stats is not None:
                d.stats = p.stats.data_ptr() + 8 * y.stats
        p.late(late)
        p.fwd.append(Launch("conv", d))
        y.producer = ("conv", d)
        if p.need_grad and weight.requires_grad:
            def bw():
                if not y.grad_written:
                    return
                lane = p.cur_lane
                for grp in range(4):   # weight gradient of the 16 taps in 4 groups of 4 (one kernel row each)
                    g = nv.WgradDesc()
                    g.x, g.dy, g.dw, g.dtype = xs.ptr(), y.gptr(), gw12.data_ptr(), _cdt(p, dtype)
                    g.N, g.H, g.W, g.Cin, g.x_pitch = xs.N, xs.H, xs.W, _rup(xs.C, vec), xs.pitch
                    g.Ho, g.Wo, g.Cout, g.dy_pitch = y.H, y.W, cout, y.pitch
                    g.in_stride, g.ntaps = 1, 4
                    for i in range(4):
                        g.dy_t[i], g.dx_t[i] = taps[4 * grp + i]
                    g.dw_cin, g.dw_tap_stride, g.dw_tap_off, g.accumulate = xs.C, 16, 4 * grp, 0
                    g.reserved = 1    # (plan-side mark: the gather below reads gw12 right away - never a deferred fold)
                    p.wgrad_ws_bytes[lane] = max(p.wgrad_ws_bytes.get(lane, 0), int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g))))
                    p.late(lambda g=g, lane=lane: p.patch_wgrad_ws(g, lane))
                    p.bwd.append(Launch("wgrad", g))
                gp = p.grad_of_param(weight)
                acc = 1 if p.grad_arena is not None else 0
                p.bwd.append(lambda s: nv.call("hrp_gather_f32", gw12.data_ptr(), idx_g.data_ptr(), gp.data_ptr(),
                                               gp.numel(), acc, s))
            self.bwd_stack.append(bw)
        return y

    def maxpool3x3s2(self, x):
        """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (Resnet.py:25)."""
        p = self.plan
        x.check_readable()
        Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        y = p.new(x.N, Ho, Wo, x.C, x.dtype)
        y.requires_grad = p.need_grad and x.requires_grad
        arg = None
        if y.requires_grad:
            arg = torch.zeros(x.N * Ho * Wo * x.C, dtype=torch.uint8, device=p.device)
            p.keep.append(arg)
        p.fwd.append(lambda s: nv.call("hrp_maxpool3x3s2_fwd", x.ptr(), _dt(x.dtype), x.N, x.H, x.W, x.C, x.pitch, y.ptr(),
                                       y.pitch, arg.data_ptr() if arg is not None else None, s))
        if y.requires_grad:
            def bw():
                if not y.grad_written:
                    return
                acc = x.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_maxpool3x3s2_bwd", y.gptr(), y.pitch, arg.data_ptr(), x.gptr(), _dt(x.dtype),
                                               x.N, x.H, x.W, x.C, x.pitch, acc, s))
            self.bwd_stack.append(bw)
        return y

    @staticmethod
    def _class_taps(k, pad, py, px):
        """Taps of output-parity class (py, px) of a stride-2 transposed convolution / data gradient: the forward conv
        reads iy = 2 oy + ky - pad, so pixel iy = 2 a + py receives from (ky, oy = a + (py + pad - ky) / 2)."""
        ys = [(ky, (py + pad - ky) // 2) for ky in range(k) if (py + pad - ky) % 2 == 0]
        xs = [(kx, (px + pad - kx) // 2) for kx in range(k) if (px + pad - kx) % 2 == 0]
        return [(oy, ox, ky * k + kx) for (ky, oy) in ys for (kx, ox) in xs]

    def deconv4x4s2(self, x, weight, want_stats=False):
        """nn.ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1, bias=False) (full_net.py:194-216).  With V = the
        Conv2d(Cout -> Cin, 4, stride 2, padding 1) whose weight tensor is `weight` itself ([Cin, Cout, 4, 4] = V's
        [out, in, 4, 4]): forward = V's data gradient (four output-parity launches of 2x2 taps, no zero insertion),
        gradient wrt the input = V's forward (16 taps), weight gradient = V's weight gradient (4 groups of 4 taps)."""
        p = self.plan
        x.check_readable()
        cin_t, cout_t = weight.shape[0], weight.shape[1]
        assert tuple(weight.shape[2:]) == (4, 4) and x.C == cin_t
        dtype = x.dtype
        vec = 8 if dtype == torch.bfloat16 else 4
        esz = 4 if dtype == torch.float32 else 2
        w = p.weight(weight, cin_t, cout_t, 16)      # as V's weight: cout_V = Cin_T, cin_V = Cout_T
        w.dtype, w.cin_used, w.need_t = dtype, cout_t, True
        y = p.new(x.N, 2 * x.H, 2 * x.W, cout_t, dtype)
        y.requires_grad = p.need_grad
        if want_stats:
            y.stats = p.alloc_stats(cout_t)
        for (py, px) in [(0, 0), (0, 1), (1, 0), (1, 1)]:
            d = nv.ConvDesc()
            d.x, d.y, d.dtype = x.ptr(), y.ptr(), _cdt(p, dtype)
            d.N, d.H, d.W, d.Cin, d.x_pitch = x.N, x.H, x.W, _rup(x.C, vec), x.pitch
            d.Cout = cout_t
            d.y_H, d.y_W, d.y_pitch, d.res_pitch = y.H, y.W, y.pitch, y.pitch
            d.in_stride, d.out_stride, d.out_off_y, d.out_off_x = 1, 2, py, px
            d.Ho, d.Wo = (y.H - py + 1) // 2, (y.W - px + 1) // 2
            tl = self._class_taps(4, 1, py, px)
            d.ntaps = len(tl)
            for i, (a, b, t) in enumerate(tl):
                d.dy[i], d.dx[i], d.wtap[i] = a, b, t
            d.w_ntaps, d.w_cout_pad = 16, _rup(cout_t, 32)

            def late(d=d):
                d.w = w.arena.data_ptr() + w.bwd_off * esz
                if y.stats is not None:
                    d.stats = p.stats.data_ptr() + 8 * y.stats
            p.late(late)
            p.fwd.append(Launch("conv", d))
        if p.need_grad:
            self.bwd_stack.append(lambda: self._deconv_bwd(x, w, y, dtype))
        return y

    def _deconv_bwd(self, x, w, y, dtype):
        p = self.plan
        if not y.grad_written:
            return
        vec = 8 if dtype == torch.bfloat16 else 4
        esz = 4 if dtype == torch.float32 else 2
        taps = [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]
        if w.param.requires_grad:
            lane = p.cur_lane
            for grp in range(4):
                g = nv.WgradDesc()
                g.x, g.dy, g.dw, g.dtype = y.gptr(), x.ptr(), p.grad_of_param(w.param).data_ptr(), _cdt(p, dtype)
                g.N, g.H, g.W, g.Cin, g.x_pitch = y.N, y.H, y.W, _rup(y.C, vec), y.pitch
                g.Ho, g.Wo, g.Cout, g.dy_pitch = x.H, x.W, x.C, x.pitch
                g.in_stride, g.ntaps = 2, 4
                for i in range(4):
                    g.dy_t[i], g.dx_t[i] = taps[4 * grp + i]
                g.dw_cin, g.dw_tap_stride, g.dw_tap_off = y.C, 16, 4 * grp
                g.accumulate = 1 if (w.grad_written or p.grad_arena is not None) else 0
                p.wgrad_ws_bytes[lane] = max(p.wgrad_ws_bytes.get(lane, 0), int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g))))
                p.late(lambda g=g, lane=lane: p.patch_wgrad_ws(g, lane))
                p.bwd.append(Launch("wgrad", g))
            w.grad_written = True
        if x.requires_grad:
            acc = x.take_grad_slot()
            d = nv.ConvDesc()
            d.x, d.y, d.dtype = y.gptr(), x.gptr(), _cdt(p, dtype)
            d.N, d.H, d.W, d.Cin, d.x_pitch = y.N, y.H, y.W, _rup(y.C, vec), y.pitch
            d.Ho, d.Wo, d.Cout = x.H, x.W, x.C
            d.y_H, d.y_W, d.y_pitch, d.res_pitch = x.H, x.W, x.pitch, x.pitch
            d.out_stride, d.in_stride, d.ntaps, d.w_ntaps, d.w_cout_pad = 1, 2, 16, 16, _rup(x.C, 32)
            for i, (a, b) in enumerate(taps):
                d.dy[i], d.dx[i], d.wtap[i] = a, b, i
            if acc:
                d.res = x.gptr()
            p.late(lambda d=d: setattr(d, "w", w.arena.data_ptr() + w.fwd_off * esz))
            p.bwd.append(Launch("conv", d))

    def constant(self, N, Cc, value):
        t = self.plan.new(N, 1, 1, Cc, torch.float32, pitch=Cc)
        t.buf.fill_(value)
        return t

    def vector_input(self, name, N, Cc, dense=False):
        """fp32 [N, C] external tensor copied into a static buffer."""
        p = self.plan
        t = p.new(N, 1, 1, Cc, torch.float32, pitch=Cc if dense else None)

        # the vector inputs of a plan are declared next to each other (full_net._build: k_value, K, init_pose, init_rot): ONE launch
        # copies all that were declared since the last other forward op
        grp = getattr(self, "_vec_group", None)
        if grp is None or grp["at"] != len(p.fwd) or grp["lane"] != p.lane_path or len(grp["items"]) >= nv.COPY_MAX:
            grp = self._vec_group = dict(items=[], lane=p.lane_path)
            items = grp["items"]
            p.fwd.append(lambda s: nv.copy_cols_batch([(p.dyn[n_].data_ptr(), c_, t_.ptr(), t_.pitch, r_, c_, 0) for n_, r_, c_, t_ in items], s))
            grp["at"] = len(p.fwd)
        grp["items"].append((name, N, Cc, t))
        return t

    def nchw_output(self, t):
        """NHWC plan tensor -> fresh NCHW fp32 torch tensor (public API boundary)."""
        dt = _dt(t.dtype)
        holder = {}
        p = self.plan
        t.materialize()

        def op(s):
            out = torch.empty(t.N, t.C, t.H, t.W, dtype=torch.float32, device=p.device)
            nv.call("hrp_nhwc_to_nchw", t.ptr(), out.data_ptr(), dt, t.N, t.C, t.H, t.W, t.pitch, s)
            holder["out"] = out
        p.fwd.append(op)
        return holder

    def pil_resize_input(self, name, N, H, W, scale=0.5, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """External NCHW image batch with values 0 .. 255 (float32 or uint8) -> the ResNet stem's space-to-depth tensor of the
        image resized by `scale` with Pillow's bicubic filter, / 255, normalised (reference mask_inference.py:44-49: a host loop
        over PIL images).  One launch; bit-exact with PIL.Image.resize on the bytes."""
        p = self.plan
        Ho, Wo = int(H * scale), int(W * scale)
        if Ho % 2 or Wo % 2:
            raise nv.HrpError(f"pil_resize_input: the resized image {Ho} x {Wo} (image {H} x {W}, scale {scale}) must have even sides: "
                              "it is written straight into the ResNet stem's 2 x 2 space-to-depth layout")
        t = p.new(N, Ho // 2, Wo // 2, 12, pitch=16)
        tabs = []
        for n_in, n_out in ((W, Wo), (H, Ho)):
            host = (C.c_int32 * (n_out * (nv.PIL_KMAX + 2)))()
            nv.check(nv.lib().hrp_pil_resize_table(n_in, n_out, host), "hrp_pil_resize_table")
            tabs.append(torch.frombuffer(bytearray(bytes(host)), dtype=torch.int32).to(p.device))
        p.keep += tabs
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        dt = _dt(t.dtype)

        def op(s):
            x = p.dyn[name]
            assert x.dtype in (torch.uint8, torch.float32) and tuple(x.shape) == (N, 3, H, W)
            nv.call("hrp_pil_resize_normalize", x.data_ptr(), 1 if x.dtype == torch.uint8 else 0, N, H, W, tabs[0].data_ptr(),
                    tabs[1].data_ptr(), Ho, Wo, t.ptr(), dt, t.pitch, 1, m3, s3, s)
        p.fwd.append(op)
        t.external = name
        return t

    def broadcast_hw(self, vec, out):
        """out[n, y, x, :] = vec[n, :] (fp32 [N, C] -> an NHWC tensor / channel slice): F.interpolate(bilinear) of a 1 x 1 map, the
        image-pooling branch of ASPP (torchvision deeplabv3.ASPPPooling).  Inference plans."""
        p = self.plan
        assert not p.need_grad and vec.dtype == torch.float32 and vec.C == out.C and vec.N == out.N
        vec.check_readable()
        p.fwd.append(lambda s: nv.call("hrp_broadcast_hw", vec.ptr(), vec.pitch, out.ptr(), _dt(out.dtype), out.N, out.H * out.W, out.C,
                                       out.pitch, s))
        return out

    def bilinear_nchw_output(self, t, H, W, sigmoid=False):
        """NHWC plan tensor -> fresh NCHW fp32 torch tensor at H x W by bilinear interpolation (align_corners=False;
        keypoint_seg_resnet.py:147), optionally through a sigmoid (CtRNet.py:109)."""
        holder = {}
        p = self.plan
        t.check_readable()

        def op(s):
            out = torch.empty(t.N, t.C, H, W, dtype=torch.float32, device=p.device)
            nv.call("hrp_bilinear_nhwc_to_nchw", t.ptr(), _dt(t.dtype), t.N, t.H, t.W, t.C, t.pitch, out.data_ptr(), H, W,
                    1 if sigmoid else 0, s)
            holder["out"] = out
        p.fwd.append(op)
        return holder

    def channel_slice(self, t, c0, cc):
        """View of channels [c0, c0 + cc) of an NHWC tensor (writes through)."""
        assert c0 % 8 == 0 and c0 + cc <= t.C
        v = TensorH(self.plan, t.N, t.H, t.W, cc, t.dtype, buf=t.buf, offset=t.offset + c0, pitch=t.pitch, base=t.base or t)
        v.lane_path = t.lane_path
        return v

    # ---- convolution ----------------------------------------------------------------------------------
    def _conv_desc(self, x, w, y, stride, ksize, dtype, into=None, dilation=1):
        d = into if into is not None else nv.ConvDesc()
        d.x, d.y = x.ptr(), y.ptr()
        d.dtype = _cdt(self.plan, dtype)
        d.N, d.H, d.W, d.Cin, d.x_pitch = x.N, x.H, x.W, _rup(x.C, 8 if dtype == torch.bfloat16 else 4), x.pitch
        d.Ho, d.Wo, d.Cout = y.H, y.W, y.C
        d.y_H, d.y_W, d.y_pitch, d.res_pitch = y.H, y.W, y.pitch, y.pitch
        d.out_stride, d.out_off_y, d.out_off_x = 1, 0, 0
        d.in_stride = stride
        taps = _TAPS3 if ksize == 3 else [(0, 0)]
        d.ntaps = len(taps)
        for i, (a, b) in enumerate(taps):
            d.dy[i], d.dx[i], d.wtap[i] = a * dilation, b * dilation, i
        d.w_ntaps = len(taps)
        d.w_cout_pad = _rup(y.C, 32)
        return d

    def conv(self, x, weight, bias=None, stride=1, want_stats=False, out=None, residual=None, relu=False, dilation=1):
        """y = conv(x) (+bias) (+residual) (ReLU).  weight: torch parameter [Cout, Cin, k, k] or [Cout, Cin].
        dilation > 1 (3x3, stride 1, padding = dilation: the atrous layers of DeepLabv3's ResNet-50, reference
        lib/models/ctrnet/keypoint_seg_resnet.py:103-149): the same tile program with the taps dilation pixels apart."""
        p = self.plan
        x.check_readable()
        cout, cin = weight.shape[0], weight.shape[1]
        ksize = weight.shape[2] if weight.dim() == 4 else 1
        ntaps = ksize * ksize
        dtype = x.dtype
        w = p.weight(weight, cout, cin, ntaps)
        w.dtype = dtype
        w.cin_used = cin
        w.need_t = getattr(w, "need_t", False) or (p.need_grad and x.requires_grad)
        assert dilation == 1 or (ksize == 3 and stride == 1), "dilated convolutions: 3x3, stride 1"
        Ho = (x.H + 2 * (ksize // 2) - ksize) // stride + 1
        Wo = (x.W + 2 * (ksize // 2) - ksize) // stride + 1
        y = out if out is not None else p.new(x.N, Ho, Wo, cout, dtype)
        if dilation > MAX_TILE_DILATION:
            return self._conv_shifted_taps(x, w, y, bias, dilation, dtype, want_stats, residual, relu)
        y.requires_grad = p.need_grad
        d = self._conv_desc(x, w, y, stride, ksize, dtype, dilation=dilation)
        if bias is not None:
            d.bias = bias.data_ptr()
        if residual is not None:
            d.res, d.res_pitch = residual.ptr(), residual.pitch
        d.relu = 1 if relu else 0
        if want_stats:
            y.stats = p.alloc_stats(cout)
        esz = 4 if dtype == torch.float32 else 2

        def late():
            d.w = w.arena.data_ptr() + w.fwd_off * esz
            if y.stats is not None:
                d.stats = p.stats.data_ptr() + 8 * y.stats
        p.late(late)
        p.fwd.append(Launch("conv", d))
        y.producer = ("conv", d)
        if p.need_grad:
            self.bwd_stack.append(lambda: self._conv_bwd(x, w, y, bias, stride, ksize, dtype, residual, relu, dilation))
        return y

    def _conv_shifted_taps(self, x, w, y, bias, dilation, dtype, want_stats, residual, relu):
        """3x3 convolution whose taps lie further apart than the tile program's halo (ASPP rates 12 / 24 / 36 of DeepLabv3's head,
        reference lib/models/ctrnet/keypoint_seg_resnet.py:121-129): nine ONE-tap problems over the nine packed taps of the same
        weight, each on the rectangle of output pixels whose source pixel lies inside the image (the tap offset folded into the
        problem's window, so that no problem needs border handling), the centre tap first and the others accumulating onto it
        (res == y).  A tap whose rectangle is empty (rate 36 on a 30-row map: every vertical tap) is no launch at all.  Inference
        plans only; BatchNorm + ReLU behind it run as PlanBuilder.act(..., out=y) in place: the nine partial sums are rounded to
        the plan's element type between the launches."""
        p = self.plan
        if p.need_grad or want_stats or residual is not None or relu:
            raise nv.HrpError(f"conv: dilation {dilation} (beyond the tile halo) is built for inference plans without fused epilogue")
        esz = 4 if dtype == torch.float32 else 2
        first = True
        for i, (a, b) in sorted(enumerate(_TAPS3), key=lambda kv: (kv[1] != (0, 0), kv[0])):
            dt_y, dt_x = a * dilation, b * dilation
            y0, y1 = max(0, -dt_y), min(y.H, y.H - dt_y)
            x0, x1 = max(0, -dt_x), min(y.W, y.W - dt_x)
            if y1 <= y0 or x1 <= x0:
                continue
            d = self._conv_desc(x, w, y, 1, 1, dtype)
            d.Ho, d.Wo, d.out_off_y, d.out_off_x = y1 - y0, x1 - x0, y0, x0
            d.ntaps, d.w_ntaps = 1, 9
            d.dy[0], d.dx[0], d.wtap[0] = y0 + dt_y, x0 + dt_x, i
            if first:
                assert (a, b) == (0, 0)
                if bias is not None:
                    d.bias = bias.data_ptr()
            else:
                d.res, d.res_pitch = y.ptr(), y.pitch
            first = False
            p.late(lambda d=d: setattr(d, "w", w.arena.data_ptr() + w.fwd_off * esz))
            p.fwd.append(Launch("conv", d))
        y.producer = None          # (nothing to fold a BatchNorm into: the sum is complete only after the last launch)
        return y

    def _wgrad_launch(self, x, w, y, ksize=3, stride=1):
        """Weight-gradient launch of conv(x) -> y for parameter holder w (dW (+)= x^T * y.grad)."""
        p = self.plan
        dtype = x.dtype
        vec = 8 if dtype == torch.bfloat16 else 4
        taps = _TAPS3 if ksize == 3 else [(0, 0)]
        g = nv.WgradDesc()
        g.x, g.dy, g.dw = x.ptr(), y.gptr(), p.grad_of_param(w.param).data_ptr()
        g.dtype = _cdt(p, dtype)
        g.N, g.H, g.W, g.Cin, g.x_pitch = x.N, x.H, x.W, _rup(x.C, vec), x.pitch
        g.Ho, g.Wo, g.Cout, g.dy_pitch = y.H, y.W, y.C, y.pitch
        g.in_stride, g.ntaps = stride, len(taps)
        for i, (a, b) in enumerate(taps):
            g.dy_t[i], g.dx_t[i] = a, b
        g.dw_cin = w.cin
        g.accumulate = 1 if (w.grad_written or p.grad_arena is not None) else 0
        w.grad_written = True
        lane = p.cur_lane
        p.wgrad_ws_bytes[lane] = max(p.wgrad_ws_bytes.get(lane, 0), int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g))))
        p.late(lambda g=g, lane=lane: p.patch_wgrad_ws(g, lane))
        p.bwd.append(Launch("wgrad", g))


    def _conv_bwd(self, x, w, y, bias, stride, ksize, dtype, residual, relu, dilation=1):
        p = self.plan
        assert not relu, "ReLU fused in a conv epilogue is inference-only"
        if not y.grad_written:
            return  # nobody consumed this output
        esz = 4 if dtype == torch.float32 else 2
        vec = 8 if dtype == torch.bfloat16 else 4
        taps = [(a * dilation, b * dilation) for a, b in _TAPS3] if ksize == 3 else [(0, 0)]
        # residual: d_res += dY (fp32 heads only)
        if residual is not None and residual.requires_grad:
            assert dtype == torch.float32
            acc = residual.take_grad_slot()
            p.bwd.append(lambda s: nv.call("hrp_copy_cols", y.gptr(), y.pitch, residual.gptr(), residual.pitch,
                                           y.N * y.H * y.W, y.C, acc, s))
        # bias gradient.  A bias in front of a BatchNorm that normalises with batch statistics (the cls head's downsamp_modules and
        # final_feat_layer, HRnet.py:364-383) has the gradient sum_pixels dy = 0 identically - the BatchNorm backward subtracts the
        # mean of its output gradient; the reference's autograd computes that zero with fp32 rounding noise (~1e-9 of the weight
        # gradients' scale).  No launch: the gradient keeps the zero of the arena (8 column-sum launches per step, 0.3 ms one by one).
        if bias is not None and bias.requires_grad and y.n_bn_readers > 0 and y.n_readers == y.n_bn_readers:
            p.grad_of_param(bias)
            p.counters["bias_grad_zero_by_bn"] = p.counters.get("bias_grad_zero_by_bn", 0) + 1
        elif bias is not None and bias.requires_grad:
            gb = p.grad_of_param(bias)
            cwb = int(nv.lib().hrp_colsum_workspace_bytes(y.N * y.H * y.W, y.C))      # deterministic: partial sums folded in order
            cws = torch.empty(cwb // 4 + 4, dtype=torch.float32, device=p.device)
            p.keep.append(cws)
            p.bwd.append(lambda s: nv.call("hrp_colsum", y.gptr(), _dt(dtype), y.N * y.H * y.W, y.C, y.pitch,
                                           gb.data_ptr(), 1 if p.grad_arena is not None else 0, cws.data_ptr(), cwb, s))
        # weight gradient
        if w.param.requires_grad:
            g = nv.WgradDesc()
            g.x, g.dy, g.dw = x.ptr(), y.gptr(), p.grad_of_param(w.param).data_ptr()
            g.dtype = _cdt(p, dtype)
            g.N, g.H, g.W, g.Cin, g.x_pitch = x.N, x.H, x.W, _rup(x.C, vec), x.pitch
            g.Ho, g.Wo, g.Cout, g.dy_pitch = y.H, y.W, y.C, y.pitch
            g.in_stride, g.ntaps = stride, len(taps)
            for i, (a, b) in enumerate(taps):
                g.dy_t[i], g.dx_t[i] = a, b
            g.dw_cin = w.cin
            g.accumulate = 1 if (w.grad_written or p.grad_arena is not None) else 0
            w.grad_written = True
            lane = p.cur_lane
            p.wgrad_ws_bytes[lane] = max(p.wgrad_ws_bytes.get(lane, 0), int(nv.lib().hrp_wgrad_workspace_bytes(C.byref(g))))
            p.late(lambda g=g, lane=lane: p.patch_wgrad_ws(g, lane))
            p.bwd.append(Launch("wgrad", g))
        # data gradient
        if x.requires_grad:
            acc = x.take_grad_slot()
            classes = [(0, 0)] if stride == 1 else [(0, 0), (0, 1), (1, 0), (1, 1)]
            if stride == 2 and ksize == 1 and not acc:
                # only the even pixels are written below: clear the buffer, then accumulate into it
                gb_bytes = x.N * x.H * x.W * x.pitch * esz
                p.bwd.append(lambda s: nv.call("hrp_fill_zero", x.gptr(), gb_bytes, s))
                acc = 1
            # stride-2 3x3 layers below ~4 GFLOP of data gradient (the fuse / transition layers of the branches: launch-latency
            # bound): the 1- / 2- / 2- / 4-tap parity classes all become 4-tap problems - missing taps point at a zero tap
            # slot of the transposed packing - and share ONE batched launch instead of three (246 -> ~100 launches per step)
            pad4 = (PARITY_PAD and stride == 2 and ksize == 3 and dtype == torch.bfloat16 and
                    2.0 * 9 * y.N * y.H * y.W * y.C * x.C < 4e9)
            if pad4:
                w.pad_t = 1
            par = None
            if len(classes) > 1:   # the parity classes write disjoint pixels: virtual lanes, one batched launch per tap count
                vp = self.parallel(len(classes), virtual=True)
                par = vp.__enter__()
            for ci, (py, px) in enumerate(classes):
                d = nv.ConvDesc()
                d.x, d.y = y.gptr(), x.gptr()
                d.dtype = _cdt(p, dtype)
                d.N, d.H, d.W, d.Cin, d.x_pitch = y.N, y.H, y.W, _rup(y.C, vec), y.pitch
                d.Cout = x.C
                d.y_H, d.y_W, d.y_pitch = x.H, x.W, x.pitch
                d.in_stride = 1
                if stride == 1:
                    d.Ho, d.Wo = x.H, x.W
                    d.out_stride, d.out_off_y, d.out_off_x = 1, 0, 0
                    tl = [(-a, -b, i) for i, (a, b) in enumerate(taps)]
                else:
                    assert ksize in (1, 3) and stride == 2
                    d.Ho, d.Wo = (x.H - py + 1) // 2, (x.W - px + 1) // 2
                    d.out_stride, d.out_off_y, d.out_off_x = 2, py, px
                    # forward: iy = 2*oy + ky - pad  ->  pixel 2a+py receives from (ky, oy = a + (py + pad - ky) / 2)
                    tl = self._class_taps(ksize, ksize // 2, py, px)
                    if not tl:       # 1x1 stride 2: the odd pixels get no gradient from this conv
                        continue
                if pad4:
                    tl = tl + [(tl[0][0], tl[0][1], len(taps))] * (4 - len(tl))      # (zero slot; any offset inside the halo)
                d.ntaps = len(tl)
                for i, (a, b, t) in enumerate(tl):
                    d.dy[i], d.dx[i], d.wtap[i] = a, b, t
                d.w_ntaps = len(taps) + (1 if pad4 else 0)
                w.t_ntaps.add(int(d.w_ntaps))
                d.w_cout_pad = _rup(x.C, 32)
                p.late(lambda d=d: setattr(d, "w", w.arena.data_ptr() + w.bwd_off * esz))
                if acc:
                    d.res, d.res_pitch = x.gptr(), x.pitch
                if d.Ho > 0 and d.Wo > 0:
                    if par is not None:
                        with par.lane(ci):
                            p.bwd.append(Launch("conv", d))
                    else:
                        p.bwd.append(Launch("conv", d))
            if par is not None:
                vp.__exit__(None, None, None)

    # ---- element-wise ---------------------------------------------------------------------------------
    def _fold(self, bn):
        p = self.plan
        e = p.bn_fold.get(id(bn))
        if e is None:
            sc = torch.zeros(bn.num_features, dtype=torch.float32, device=p.device)
            sh = torch.zeros(bn.num_features, dtype=torch.float32, device=p.device)
            e = (bn, sc, sh)
            p.bn_fold[id(bn)] = e
        return e[1], e[2]

    def _fold_hat(self, bn):
        p = self.plan
        e = p.bn_fold_hat.get(id(bn))
        if e is None:
            e = (bn, torch.zeros(bn.num_features, dtype=torch.float32, device=p.device),
                 torch.zeros(bn.num_features, dtype=torch.float32, device=p.device))
            p.bn_fold_hat[id(bn)] = e
        return e[1], e[2]

    def bn_batch_stats(self, bn):
        """Does this BatchNorm normalise with batch statistics in this plan?  (train_sim2real.py:139-146 trains the network with
        every BatchNorm module switched to eval(): running statistics in the forward pass, gradients through them.)"""
        return self.plan.training and bn.training

    def act(self, terms, relu, out=None):
        """out = act(sum_j BN_j(t_j) upsampled).  In inference plans a single conv+BN(+identity residual)
        is folded into the producing conv's epilogue instead.  out: write into this tensor (a channel slice of a
        concatenation; may be terms[0].t itself: the pass is one-to-one per element) instead of a new one."""
        p = self.plan
        for tm in terms:
            tm.t.check_readable()
        t0 = terms[0]
        ref = max((tm.t for tm in terms), key=lambda t: t.H)  # output geometry = largest input x its up
        H = max(tm.t.H * tm.up for tm in terms)
        W = max(tm.t.W * tm.up for tm in terms)
        Cc, N, dtype = t0.t.C, t0.t.N, t0.t.dtype
        leaky = relu == "leaky"      # nn.LeakyReLU() (slope 0.01): descriptor relu = 2, element-wise kernels only
        if self.fuse_inference and not leaky and t0.bn is not None and t0.up == 1 and t0.t.producer is not None \
                and t0.t.producer[0] == "conv" and not getattr(t0.t, "consumed", False) \
                and len(terms) <= 2 and all(tm.bn is None and tm.up == 1 for tm in terms[1:]):
            d = t0.t.producer[1]
            sc, sh = self._fold(t0.bn)
            d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
            if len(terms) == 2:
                d.res, d.res_pitch = terms[1].t.ptr(), terms[1].t.pitch
            d.relu = 1 if relu else 0
            t0.t.consumed = True
            t0.t.producer = None
            return t0.t
        if out is None:
            out = p.new(N, H, W, Cc, dtype)
        else:
            assert (out.N, out.H, out.W, out.C, out.dtype) == (N, H, W, Cc, dtype) and not p.need_grad
        out.requires_grad = p.need_grad and any(tm.t.requires_grad for tm in terms)
        d = nv.EwDesc()
        d.nin = len(terms)
        d.out, d.out_pitch, d.dtype = out.ptr(), out.pitch, _dt(dtype)
        d.N, d.H, d.W, d.C, d.relu = N, H, W, Cc, 2 if leaky else (1 if relu else 0)
        ins = []
        for j, tm in enumerate(terms):
            e = d.inp[j]
            e.ptr, e.pitch, e.up = tm.t.ptr(), tm.t.pitch, tm.up
            assert tm.t.H * tm.up == H and tm.t.W * tm.up == W and tm.t.C == Cc
            if tm.bn is None:
                e.mode = nv.EW_IDENTITY
            elif self.bn_batch_stats(tm.bn):
                assert tm.t.stats is not None, "train-mode BN needs conv statistics"
                e.mode = nv.EW_BN_TRAIN
                tm.t.n_bn_readers += 1           # (read by nothing else, its producer's bias has an identically zero gradient: _conv_bwd)
                e.a, e.b = tm.bn.weight.data_ptr(), tm.bn.bias.data_ptr()
                e.count, e.eps = float(tm.t.N * tm.t.H * tm.t.W), tm.bn.eps
                p.bn_train.append((tm.bn, tm.t.stats, tm.t.N * tm.t.H * tm.t.W))
                off = tm.t.stats
                p.late(lambda e=e, off=off: setattr(e, "stats", p.stats.data_ptr() + 8 * off))
            else:
                sc, sh = self._fold(tm.bn)
                e.mode = nv.EW_AFFINE
                e.a, e.b = sc.data_ptr(), sh.data_ptr()
            ins.append(e)
        vec = 16 // out.esz
        if relu and out.requires_grad and RELU_BITMASK and Cc % vec == 0 and all(tm.t.pitch % vec == 0 for tm in terms):
            # the backward needs only the sign of the output: one byte per 16-byte vector instead of the tensor
            mask = torch.zeros(N * H * W * (Cc // vec), dtype=torch.uint8, device=p.device)
            p.keep.append(mask)
            d.mask, d.mask_pitch = mask.data_ptr(), Cc // vec
        p.fwd.append(Launch("ew_fwd", d))
        out.ew_desc = d
        if p.need_grad:
            self.bwd_stack.append(lambda: self._act_bwd(terms, out, relu, d))
        return out

    def _act_bwd(self, terms, out, relu, fd):
        p = self.plan
        if not out.grad_written:
            return
        # an identity term (residual) at the output resolution rides along with a BN / affine term of the same
        # activation: one apply launch writes both gradients (hrp_ew_bwd_desc.din2)
        # upsampled terms (the 2 / 4 / 8-fold inputs of a fuse sum, HRnet.py:197-208): the masked output gradient is summed over the
        # 2 x 2 windows ONCE (hrp_ew_pool2; 4 x 4 and 8 x 8 from the level below) and every term's reduce and apply pass read their
        # level instead of pooling out.grad under the mask again - six passes over the [N, 64, 64, 32] gradient per stage-4 output
        pooled = {}
        ups = sorted({tm.up for tm in terms if tm.up > 1 and tm.t.requires_grad})
        if POOL_FUSE_GRADS and ups and relu and relu != "leaky" and fd.mask and fd.C % 8 == 0 and all(u in (2, 4, 8) for u in ups):
            src, sdt, spitch, msk, mp = out.gptr(), fd.dtype, out.pitch, fd.mask, fd.mask_pitch
            Hc, Wc, lvl = fd.H, fd.W, 1
            while lvl < ups[-1]:
                lvl *= 2
                buf = torch.zeros(fd.N * (Hc // 2) * (Wc // 2) * fd.C, dtype=torch.float32, device=p.device)
                p.keep.append(buf)
                p.bwd.append(lambda s, src=src, sdt=sdt, spitch=spitch, msk=msk, mp=mp, Hc=Hc, Wc=Wc, buf=buf: nv.call(
                    "hrp_ew_pool2", src, sdt, spitch, msk, mp, fd.N, Hc, Wc, fd.C, buf.data_ptr(), s))
                pooled[lvl] = buf
                src, sdt, spitch, msk, mp, Hc, Wc = buf.data_ptr(), nv.HRP_F32, fd.C, None, 0, Hc // 2, Wc // 2
            p.counters["fuse_grad_pools"] = p.counters.get("fuse_grad_pools", 0) + 1
        host = next((j for j, tm in enumerate(terms) if tm.t.requires_grad and tm.bn is not None and tm.up == 1), None)
        rider = next((j for j, tm in enumerate(terms) if tm.t.requires_grad and tm.bn is None and tm.up == 1
                      and fd.inp[j].mode == nv.EW_IDENTITY), None) if host is not None else None
        for j, tm in enumerate(terms):
            if not tm.t.requires_grad or j == rider:
                continue
            bn_eval = tm.bn is not None and not self.bn_batch_stats(tm.bn)
            b = nv.EwBwdDesc()
            b.dout, b.out = out.gptr(), out.ptr()
            b.dout_pitch, b.out_pitch = out.pitch, out.pitch
            src = fd.inp[j]
            for f, _ in nv.EwInput._fields_:
                setattr(b.inp, f, getattr(src, f))
            # the statistics pointer of the forward descriptor is patched at finalize: copy it then
            p.late(lambda b=b, src=src: setattr(b.inp, "stats", src.stats))
            b.dtype, b.N, b.H, b.W, b.C, b.relu = fd.dtype, fd.N, fd.H, fd.W, fd.C, fd.relu
            b.mask, b.mask_pitch = fd.mask, fd.mask_pitch
            b.din, b.din_pitch = tm.t.gptr(), tm.t.pitch
            b.accumulate = tm.t.take_grad_slot()
            if tm.up in pooled:
                b.pooled = pooled[tm.up].data_ptr()
            if j == host and rider is not None:
                rt = terms[rider].t
                # the check above guarantees the aligned vector path is the same for both outputs
                b.din2, b.din2_pitch = rt.gptr(), rt.pitch
                b.accumulate2 = rt.take_grad_slot()
            if bn_eval:
                # eval-mode BatchNorm (affine with the running statistics): dx = g * scale (the apply launch below, affine input as
                # in the forward pass); dbeta = sum g, dgamma = sum g * xhat with xhat = x * invstd - mean * invstd from a reduce
                # launch of its own descriptor (affine input (invstd, -mean * invstd): elementwise.hip channel_consts)
                if tm.bn.weight.requires_grad or tm.bn.bias.requires_grad:
                    r = nv.EwBwdDesc()
                    C.memmove(C.byref(r), C.byref(b), C.sizeof(nv.EwBwdDesc))
                    inv, nmi = self._fold_hat(tm.bn)
                    r.inp.mode, r.inp.a, r.inp.b = nv.EW_AFFINE, inv.data_ptr(), nmi.data_ptr()
                    r.din, r.din2 = None, None
                    off = p.alloc_bsums(fd.C)
                    p.bn_bwd.append((tm.bn, off))
                    p.late(lambda r=r, off=off: setattr(r, "sums", p.bsums.data_ptr() + 8 * off))
                    p.bwd.append(Launch("ew_red", r))
            elif tm.bn is not None:
                off = p.alloc_bsums(fd.C)
                p.bn_bwd.append((tm.bn, off))
                p.late(lambda b=b, off=off: setattr(b, "sums", p.bsums.data_ptr() + 8 * off))
                p.bwd.append(Launch("ew_red", b))
            p.bwd.append(Launch("ew_app", b))

    # ---- pooling / heads -----------------------------------------------------------------------------
    def avgpool(self, x, out=None):
        """Global average pool -> fp32 [N, C] (optionally into a column slice of `out`)."""
        p = self.plan
        x.n_readers += 1
        y = out if out is not None else p.new(x.N, 1, 1, x.C, torch.float32)
        y.requires_grad = p.need_grad and x.requires_grad
        p.fwd.append(lambda s: nv.call("hrp_avgpool_fwd", x.ptr(), _dt(x.dtype), x.N, x.H * x.W, x.C, x.pitch,
                                       y.ptr(), y.pitch, s))
        if p.need_grad:
            def bw():
                if not y.grad_written or not x.requires_grad:
                    return
                acc = x.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_avgpool_bwd", y.gptr(), y.pitch, x.gptr(), _dt(x.dtype), x.N,
                                               x.H * x.W, x.C, x.pitch, acc, s))
            self.bwd_stack.append(bw)
        return y

    def cast(self, x, dtype):
        """NHWC tensor in another element type (bf16 trunk -> fp32 head): two layout launches through an fp32 NCHW scratch
        (hrp_nhwc_to_nchw + hrp_nchw_to_nhwc: both exist for the public boundary), the gradient the same way back.  The tensor
        must have no other gradient producer before this one (the gradient is written, not accumulated)."""
        p = self.plan
        if x.dtype == dtype:
            return x
        x.check_readable()
        y = p.new(x.N, x.H, x.W, x.C, dtype)
        y.requires_grad = p.need_grad and x.requires_grad
        tmp = torch.empty(x.N * x.C * x.H * x.W, dtype=torch.float32, device=p.device)
        p.keep.append(tmp)
        p.fwd.append(lambda s: (nv.call("hrp_nhwc_to_nchw", x.ptr(), tmp.data_ptr(), _dt(x.dtype), x.N, x.C, x.H, x.W, x.pitch, s),
                                nv.call("hrp_nchw_to_nhwc", tmp.data_ptr(), y.ptr(), _dt(dtype), x.N, x.C, x.H, x.W, y.pitch, s)))
        if y.requires_grad:
            def bw():
                if not y.grad_written:
                    return
                if x.take_grad_slot():
                    raise RuntimeError("plan: cast() needs to be the first producer of its input's gradient")
                p.bwd.append(lambda s: (nv.call("hrp_nhwc_to_nchw", y.gptr(), tmp.data_ptr(), _dt(dtype), x.N, x.C, x.H, x.W, y.pitch, s),
                                        nv.call("hrp_nchw_to_nhwc", tmp.data_ptr(), x.gptr(), _dt(x.dtype), x.N, x.C, x.H, x.W, x.pitch, s)))
            self.bwd_stack.append(bw)
        return y

    def new_like(self, t):
        return self.plan.new(t.N, t.H, t.W, t.C, t.dtype, pitch=t.pitch)

    def copy_cols(self, src, dst):
        """dst[:, :src.C] = src (fp32), gradient flows back additively."""
        p = self.plan
        src.n_readers += 1
        rows = src.N * src.H * src.W
        dst.requires_grad = dst.requires_grad or src.requires_grad
        p.fwd.append(lambda s: nv.call("hrp_copy_cols", src.ptr(), src.pitch, dst.ptr(), dst.pitch, rows, src.C, 0, s))
        if p.need_grad:
            def bw():
                if not dst.grad_written or not src.requires_grad:
                    return
                acc = src.take_grad_slot()
                p.bwd.append(lambda s: nv.call("hrp_copy_cols", dst.gptr(), dst.pitch, src.gptr(), src.pitch, rows,
                                               src.C, acc, s))
            self.bwd_stack.append(bw)

    def finish(self):
        """Emit the backward list (reverse forward order) and resolve pointers."""
        p = self.plan
        assert p.cur_lane == 0 and p.lane_path == ()
        for t in list(p.pending_inputs):          # (an input nothing read)
            t.materialize()
        for t in list(p.pending_block_ends):      # block outputs nobody read in the forward (plan outputs are handled in output())
            t.materialize()
        for lane, path, emit in reversed(self.bwd_stack):
            if lane is None:
                list.append(p.bwd, Entry(None, (), emit))    # lane fork / join marker (already mirrored)
                continue
            p.cur_lane, p.lane_path = lane, path
            emit()
        p.cur_lane, p.lane_path = 0, ()
        p.finalize()

    @contextlib.contextmanager
    def parallel(self, n, virtual=False):
        """``with pb.parallel(n) as par: with par.lane(i): ...`` - emit n independent sub-graphs into n lanes
        (lane 0 stays on the current lane).  Tensors produced before the block may be read by every lane;
        nothing produced or whose gradient is written inside one lane may be touched by a sibling.
        virtual: the lanes exist for the lock-step merge only (independent launches of ONE chain - the parity classes
        of a stride-2 data gradient, the paths of a fuse layer); in lanes mode they stay on the parent's stream."""
        p = self.plan
        p._n_blocks += 1
        parent = p.cur_lane
        if len(p.lane_path) >= MAX_LANE_DEPTH or virtual:
            n = 1   # deeper blocks stay on their parent lane
        n = min(n, 64)
        children = [p.lane_id(parent, i) for i in range(1, n)]
        par = _Parallel(self, p._n_blocks, ([parent] + children + [parent] * 64) if children else [parent] * 64)
        p._block_lanes[p._n_blocks] = ([parent] + children) if children else None
        if children:
            list.append(p.fwd, Entry(None, (), _LaneSync("fork", parent, children)))
            list.append(self.bwd_stack, (None, None, _LaneSync("join", parent, children)))
        yield par
        if children:
            list.append(p.fwd, Entry(None, (), _LaneSync("join", parent, children)))
            list.append(self.bwd_stack, (None, None, _LaneSync("fork", parent, children)))

    # ---- outputs ----------------------------------------------------------------------------------------
    def output(self, t):
        """Mark a tensor as a plan output whose gradient is provided by the caller."""
        p = self.plan
        if p.need_grad and t.requires_grad:
            t.grad_written = True
            t.grad_buf()
        p.out_handles.append(t)
        return t


