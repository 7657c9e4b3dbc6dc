"""Optimizer step of the reference's training loops on the HIP path.

The reference trains with ``torch.optim.Adam`` preceded by ``torch.nn.utils.clip_grad_norm_``
(scripts/train_full.py:42, configs/panda/full.yaml:39 ``clip_gradient: 5``).  ``FusedClipAdam`` is that pair
as two table-driven launches over ALL parameters (hrp_opt_grad_sumsq, hrp_opt_adam_step in include/hrp.h):
the multi-tensor torch path spends ~70 launches and 3 ms per step on the 57 M parameters of the full
network, this one ~0.6 ms.  Same arithmetic as torch (fp32 state, bias correction from a device-side step
counter so that the step is capturable into a HIP graph), no weight decay / amsgrad (the reference uses
neither).  There is no CPU path.
"""
import ctypes as C

import torch

from . import _native as nv


class FusedClipAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedClipAdam: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise nv.HrpError("FusedClipAdam runs on an MI355X (gfx950) only; there is no CPU path")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("FusedClipAdam: parameters must be contiguous fp32 tensors on one device")
        self.device = dev
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.max_norm = float(max_norm) if max_norm else 0.0
        total = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        self._m = torch.zeros(total, dtype=torch.float32, device=dev)
        self._v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.state, off = [], 0
        for p in self.params:
            n = p.numel()
            self.state.append((self._m[off:off + n].view_as(p), self._v[off:off + n].view_as(p)))
            off += (n + 3) // 4 * 4
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self._slots = torch.zeros(8, dtype=torch.float32, device=dev)
        self._grad_ptrs = None
        self._tensors = self._chunks = None
        self._nchunks = 0

    def _build_tables(self):
        ptrs = tuple(p.grad.data_ptr() for p in self.params)
        if ptrs == self._grad_ptrs:
            return
        tab = (nv.OptTensor * len(self.params))()
        chunks = []
        for i, (p, (m, v)) in enumerate(zip(self.params, self.state)):
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
        self._grad_ptrs = ptrs

    def prepare(self):
        """Build the device tables (call once outside a HIP graph capture; gradients must exist)."""
        if any(p.grad is None for p in self.params):
            raise ValueError("FusedClipAdam: every parameter needs a gradient (run one backward first)")
        self._build_tables()

    @torch.no_grad()
    def step(self):
        self.prepare()
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.step_count += 1
        if self.max_norm > 0:
            self._slots.zero_()
            nv.call("hrp_opt_grad_sumsq", self._tensors.data_ptr(), self._chunks.data_ptr(), self._nchunks,
                    self._slots.data_ptr(), s)
        nv.call("hrp_opt_adam_step", self._tensors.data_ptr(), self._chunks.data_ptr(), self._nchunks,
                self._slots.data_ptr(), self.max_norm, self.step_count.data_ptr(),
                self.lr, self.betas[0], self.betas[1], self.eps, s)

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
