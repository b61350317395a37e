"""Portable deterministic tensors for the G9 full-width fixtures: weights and inputs are REGENERATED on both
sides from a counter-based 32-bit integer hash (exact integer arithmetic, identical on every platform) instead of being
stored, so a full-width layer (256 -> 256 3x3: 2.4 MB of weights) costs a fixture only its sampled outputs."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def u01(n: int, seed: int) -> np.ndarray:
    """n float32 values in [0, 1): murmur3 finaliser of (index + seed * golden-ratio), top 24 bits."""
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B1)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return (x >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)


def tensor(shape, seed: int, lo: float = -1.0, hi: float = 1.0):
    import torch
    n = int(np.prod(shape))
    return torch.from_numpy((u01(n, seed) * np.float32(hi - lo) + np.float32(lo)).reshape(shape).copy())


def fill_state(module, seed: int) -> None:
    """Overwrite every floating tensor of module.state_dict() (seeded by its key): conv weights uniform with variance 2/fan_in,
    norm weights / running_var in [0.75, 1.25], biases / running_mean in [-0.1, 0.1], ScaleExp scales in [0.8, 1.2]."""
    import torch
    with torch.no_grad():
        import zlib
        for k, t in module.state_dict().items():
            if not t.is_floating_point():
                continue
            s = (seed * 1000003 + zlib.crc32(k.encode())) & 0x7FFFFFFF      # by key NAME: independent of registration order
            if t.dim() == 4:
                fan = t.shape[1] * t.shape[2] * t.shape[3]
                a = (6.0 / fan) ** 0.5
                v = tensor(t.shape, s, -a, a)
            elif k.endswith("running_var") or (k.endswith("weight") and t.dim() == 1):
                v = tensor(t.shape, s, 0.75, 1.25)
            elif k.endswith("scale"):
                v = tensor(t.shape, s, 0.8, 1.2)
            else:
                v = tensor(t.shape, s, -0.1, 0.1)
            t.copy_(v)
