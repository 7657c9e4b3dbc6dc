"""Optimizer step of the reference's training loops on the HIP path.

The reference trains with ``torch.optim.Adam`` preceded by ``torch.nn.utils.clip_grad_norm_``
(scripts/train_full.py:42, configs/panda/full.yaml:39 ``clip_gradient: 5``).  ``FusedClipAdam`` is that pair
as two table-driven launches over ALL parameters (hrp_opt_grad_sumsq, hrp_opt_adam_step in include/hrp.h):
the multi-tensor torch path spends ~70 launches and 3 ms per step on the 57 M parameters of the full
network, this one ~0.6 ms.  Same arithmetic as torch (fp32 state, bias correction from a device-side step
counter so that the step is capturable into a HIP graph), no weight decay / amsgrad (the reference uses
neither).  There is no CPU path.

It is a ``torch.optim.Optimizer``: ``param_groups`` / ``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's
layout (per-parameter ``step``, ``exp_avg``, ``exp_avg_sq``), so the ``optimizer_state_dict`` of the reference's
checkpoints (lib/utils/utils.py:192-267 ``resume_run`` / ``save_checkpoint``) loads into it and what it saves loads
into ``torch.optim.Adam``; ``LambdaLR`` (utils.py:160-189) can drive ``param_groups[0]["lr"]`` - outside a captured
HIP graph, where the learning rate is a baked launch argument.
"""
import ctypes as C

import torch

from . import _native as nv


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedClipAdam: no parameters")
        if any(isinstance(p, dict) for p in params):
            raise ValueError("FusedClipAdam: one parameter group only")
        super().__init__(params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                      weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=True,
                                      differentiable=False, fused=None))
        self.params = self.param_groups[0]["params"]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise nv.HrpError("FusedClipAdam runs on an MI355X (gfx950) only; there is no CPU path")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FusedClipAdam: parameters must be contiguous fp32 tensors on one device")
        self.device = dev
        self.max_norm = float(max_norm) if max_norm else 0.0
        total = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self._m = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.state_views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.state_views.append((self._m[off:off + n].view_as(p), self._v[off:off + n].view_as(p)))
            off += (n + 3) // 4 * 4
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self._publish_state()
        self._slots = torch.zeros(8, dtype=torch.float32, device=dev)
        self._chunk_sums = None
        self._grad_ptrs = None
        self._tensors = self._chunks = None
        self._nchunks = 0

    def _publish_state(self):
        """torch.optim.Adam's per-parameter state, as views of the flat moment buffers and the shared step counter."""
        for p, (m, v) in zip(self.params, self.state_views):
            self.state[p] = {"step": self.step_count.view(()), "exp_avg": m, "exp_avg_sq": v}

    def state_dict(self):
        """torch.optim.Adam's layout; every parameter gets its OWN step tensor (the foreach Adam increments them
        one by one, aliases of the shared counter would advance N steps at a time)."""
        sd = super().state_dict()
        sd["state"] = {k: dict(v, step=v["step"].clone()) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's (or this class's) state_dict: moments are copied into the flat buffers."""
        if len(state_dict["param_groups"]) != 1:
            raise ValueError("FusedClipAdam: one parameter group only")
        g = state_dict["param_groups"][0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FusedClipAdam: weight_decay / amsgrad / maximize are not supported")
        super().load_state_dict(state_dict)
        steps = set()
        with torch.no_grad():
            for p, (m, v) in zip(self.params, self.state_views):
                st = self.state.get(p)
                if not st:               # torch leaves parameters that never stepped without state
                    m.zero_()
                    v.zero_()
                    continue
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps.add(float(st["step"]))
            if len(steps) > 1:
                raise ValueError(f"FusedClipAdam: parameters at different step counts {sorted(steps)}")
            self.step_count.fill_(steps.pop() if steps else 0.0)
        self.param_groups[0]["capturable"] = True
        self._publish_state()

    def _build_tables(self):
        ptrs = tuple(p.grad.data_ptr() for p in self.params)
        if ptrs == self._grad_ptrs:
            return
        tab = (nv.OptTensor * len(self.params))()
        chunks = []
        for i, (p, (m, v)) in enumerate(zip(self.params, self.state_views)):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("FusedClipAdam: gradients must be contiguous fp32")
            tab[i].param, tab[i].grad, tab[i].exp_avg, tab[i].exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            tab[i].numel = p.numel()
            chunks += [(i, k) for k in range((p.numel() + nv.OPT_CHUNK - 1) // nv.OPT_CHUNK)]
        ck = (nv.OptChunk * len(chunks))()
        for j, (i, k) in enumerate(chunks):
            ck[j].tensor, ck[j].offset = i, k
        self._tensors = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
        self._chunks = torch.frombuffer(bytearray(bytes(ck)), dtype=torch.uint8).to(self.device)
        self._nchunks = len(chunks)
        self._chunk_sums = torch.zeros(self._nchunks, dtype=torch.float32, device=self.device)
        self._grad_ptrs = ptrs

    def prepare(self):
        """Build the device tables (call once outside a HIP graph capture; gradients must exist)."""
        if any(p.grad is None for p in self.params):
            raise ValueError("FusedClipAdam: every parameter needs a gradient (run one backward first)")
        self._build_tables()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise ValueError("FusedClipAdam: closures are not supported")
        self.prepare()
        g = self.param_groups[0]
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.step_count += 1
        if self.max_norm > 0:
            # per-chunk sums folded in a fixed order (no atomics): every data-parallel rank derives the same clip
            # coefficient, bit for bit, from the same averaged gradients - replicas stay identical
            nv.call("hrp_opt_grad_sumsq", self._tensors.data_ptr(), self._chunks.data_ptr(), self._nchunks,
                    self._slots.data_ptr(), self._chunk_sums.data_ptr(), s)
        nv.call("hrp_opt_adam_step", self._tensors.data_ptr(), self._chunks.data_ptr(), self._nchunks,
                self._slots.data_ptr(), self.max_norm, self.step_count.data_ptr(),
                float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), s)
        # the parameters changed through raw pointers (no tensor._version bump): inference plans must repack / refold
        from .plan import bump_param_epoch
        bump_param_epoch()

    def total_norm(self):
        """Gradient norm of the last step (before clipping), as clip_grad_norm_ returns it."""
        return self._slots.sum().sqrt()

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
