"""Shared input generators for the parity tests."""
import numpy as np


def edge_floats(rng, H, W, nbits):
    """Random floats around the quantiser's range plus every edge case of kernel.h:39-44,68."""
    hi = float(2 ** nbits)
    x = rng.uniform(-1.0, hi + 1.5, size=(H, W)).astype(np.float32)
    specials = [np.nan, -0.0, 0.0, 0.5, 1.5, 2.5, 0.51, hi, hi - 0.5, hi + 1e-3, -1e-3, 100.0 * hi,
                np.inf, -np.inf, hi - 1.0, 3.5]
    flat = x.reshape(-1)
    for i, s in enumerate(specials):
        if i < flat.size:
            flat[(i * 7919) % flat.size] = s
    return x


def rand_q(rng, H, W, nbits, density=None):
    q = rng.integers(0, 2 ** nbits, size=(H, W), dtype=np.int64)
    if density is not None:
        q = q * (rng.random((H, W)) < density)
    return q.astype(np.int32)


def to_dev(torch, arr_u32, shape, device="cuda"):
    """flat uint32 numpy -> int32 torch tensor with the reference's reported shape."""
    return torch.from_numpy(np.ascontiguousarray(arr_u32).view(np.int32).reshape(shape)).to(device)


def to_np_u32(t):
    return t.detach().cpu().numpy().view(np.uint32).reshape(-1)
