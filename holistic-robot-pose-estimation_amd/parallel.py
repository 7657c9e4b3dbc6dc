"""Data parallelism: one process per GPU, gradients averaged with RCCL over xGMI.

Replaces the reference's per-iteration ``torch.nn.DataParallel`` re-wrap (lib/core/function.py:100-102:
parameter broadcast from GPU 0 on every forward + gradient reduce to GPU 0) by the standard
process-per-GPU scheme: identical replicas, per-rank data, ONE all-reduce per step over the flat fp32
gradient arena the plan writes into (no flatten/unflatten copies), issued in a few large buckets -
xGMI is point-to-point, so few large messages beat many small ones.  BatchNorm statistics stay
per-replica exactly as in the reference (no SyncBN under DataParallel).
"""
import os

import torch
import torch.distributed as dist


def force_world1():
    """HRP_DIST_WORLD1=1: create the process group and issue every collective even with ONE rank - the only execution of the
    RCCL path (stream ordering of graph replays against RCCL's stream in the k-cut step) a one-GPU box can produce."""
    return os.environ.get("HRP_DIST_WORLD1", "") not in ("", "0")


def collectives_active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_world1())


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_world1()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm; HRP_DIST_BACKEND=gloo lets two ranks share one GPU in a functional test
            backend = os.environ.get("HRP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_module(module, src=0):
    """Start-up: every replica takes rank `src`'s parameters and buffers."""
    if not collectives_active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class GradAllReducer:
    """Averages flat gradient buffers across ranks in buckets of `bucket_mb` MB.
    payload "bf16" (HRP_GRAD_PAYLOAD=bf16): the ranges travel as bf16 copies - half the bytes on the xGMI links (162 instead of
    323 MB for the benchmark network) - and come back into the fp32 arena, where the mean is taken in fp32.  The sum across ranks
    itself is then formed in bf16 by the collective (8 mantissa bits): an option for link-bound scaling, not the default."""

    def __init__(self, bucket_mb=64, payload=None):
        self.bucket_elems = int(bucket_mb * 1024 * 1024 // 4)
        self.payload = payload or os.environ.get("HRP_GRAD_PAYLOAD", "fp32")
        assert self.payload in ("fp32", "bf16")
        self._pending = []      # (fp32 view, bf16 copy) of ranges in flight

    def buckets(self, flat):
        n = flat.numel()
        return [flat[i:min(i + self.bucket_elems, n)] for i in range(0, n, self.bucket_elems)]

    def start(self, flat, ranges):
        """Asynchronous all-reduce (sum) of the (offset, numel) ranges of one flat buffer, in buckets; -> work handles.
        The collective is ordered after the work already enqueued on the current stream."""
        if not collectives_active():
            return []
        works = []
        for off, n in ranges:
            for b in self.buckets(flat[off:off + n]):
                if self.payload == "bf16":
                    c = b.to(torch.bfloat16)
                    self._pending.append((b, c))
                    works.append(dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True))
                else:
                    works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))
        return works

    def finish(self, works, flats):
        """Wait for `works` (the current stream waits) and turn the sums into means."""
        for w in works:
            w.wait()
        for b, c in self._pending:
            b.copy_(c)
        self._pending = []
        if collectives_active():
            for flat in flats:
                flat.div_(dist.get_world_size())

    @staticmethod
    def complement(ranges, total):
        """Ranges of [0, total) not covered by the sorted, disjoint `ranges`."""
        out, pos = [], 0
        for off, n in sorted(ranges):
            if off > pos:
                out.append((pos, off - pos))
            pos = max(pos, off + n)
        if pos < total:
            out.append((pos, total - pos))
        return out

    def __call__(self, flats):
        if not collectives_active():
            return
        works = []
        for flat in flats:
            works += self.start(flat, [(0, flat.numel())])
        self.finish(works, flats)
