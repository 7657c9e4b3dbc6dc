"""Deterministic weight synthesis keyed on state-dict names (test data rule).

Weights are not committed (hundreds of MB); instead both the reference (in gen_golden.py) and the
implementation under test load ``synth_state_dict(model.state_dict())``: every tensor is drawn from
a numpy PCG64 stream seeded with crc32(key), so the values depend only on the key name and shape.

The reference's own initialisers are unsuitable as parity data: full_net.py:167-173 re-initialises
every conv to N(0, sqrt(2/n)) which saturates the soft-argmax, and HRnet.py:577 (std 0.001) makes
every feature ~0.
"""
import zlib

import numpy as np
import torch

# logits of the 448-channel heat-map must stay O(1) so that the soft-argmax is sensitive
_FINAL_LAYER_GAIN = 3.0


def _rng(key):
    return np.random.Generator(np.random.PCG64(zlib.crc32(key.encode())))


def synth_tensor(key, ref):
    shape = tuple(ref.shape)
    g = _rng(key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if key.endswith(("init_pose", "init_rot")):
        return ref.clone()
    if leaf == "running_mean":
        v = g.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        v = g.uniform(0.5, 1.5, shape)
    elif leaf == "weight" and len(shape) == 1:            # BN gamma
        # last BN of a residual branch gets a smaller gamma so 100+ blocks do not blow up
        if ((".bn2" in key and "branches" in key) or ".bn3" in key or "fuse_layers" in key
                or "downsamp_modules" in key):
            lo, hi = 0.15, 0.35
        else:
            lo, hi = 0.5, 1.5
        v = g.uniform(lo, hi, shape)
    elif key.endswith("depth_layer.bias"):
        v = np.full(shape, 0.3)
    elif leaf == "bias":
        v = g.normal(0.0, 0.1, shape)
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if "final_layer" in key:
            gain = _FINAL_LAYER_GAIN
        elif key.startswith(("decpose", "decrot")) or ".decpose" in key or ".decrot" in key:
            gain = 0.05
        elif "depth_layer" in key:
            gain = 0.1
        v = g.normal(0.0, gain / np.sqrt(fan_in), shape)
    else:
        v = g.normal(0.0, 0.1, shape)
    return torch.as_tensor(np.asarray(v), dtype=ref.dtype).reshape(shape)


def synth_state_dict(ref_sd):
    return {k: synth_tensor(k, v) for k, v in ref_sd.items()}


def synth_inputs(B, seed=808):
    """Synthetic batch shaped like the reference's: images U[0,1), crop intrinsics,
    k_value."""
    g = np.random.Generator(np.random.PCG64(seed))
    x_reg = torch.as_tensor(g.random((B, 3, 256, 256), dtype=np.float32))
    x_root = torch.as_tensor(g.random((B, 3, 256, 256), dtype=np.float32))
    s = g.uniform(0.8, 2.5, B).astype(np.float32)
    K = np.zeros((B, 3, 3), np.float32)
    K[:, 0, 0] = 320.0 * s
    K[:, 1, 1] = 320.0 * s
    K[:, 0, 2] = 128.0
    K[:, 1, 2] = 128.0
    K[:, 2, 2] = 1.0
    side = g.uniform(80.0, 240.0, B).astype(np.float32)
    k_value = np.sqrt(K[:, 0, 0] * K[:, 1, 1] * 1000.0 * 1000.0 / side ** 2).astype(np.float32)
    return x_reg, x_root, torch.as_tensor(k_value), torch.as_tensor(K)
