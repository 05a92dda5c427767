"""Deterministic weights for fixtures whose models are too large to store (tiny width x FULL depth).

Every parameter is drawn from its own generator seeded by a CRC of its state-dict name, so the values depend on neither the
order a framework enumerates parameters in nor on which other parameters exist; values are rounded to bf16 (the reference then
runs on them in fp32, the product in bf16: only the arithmetic differs).  The fixture stores a checksum of what the generating
run saw (`checksum`); the tests re-derive the weights with this function and assert the checksum before comparing anything.
Pure test infrastructure: data generation only, no model code."""
import zlib

import torch


def seeded_weight(name: str, shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) <= 1:
        w = 1.0 + 0.1 * torch.randn(shape, generator=g) if "norm" in name else 0.02 * torch.randn(shape, generator=g)
    elif "bridge" in name and name.endswith("weight_B"):
        w = torch.randn(shape, generator=g) * 0.05          # zero-initialised upstream: made numerically live
    else:
        w = torch.randn(shape, generator=g) * (1.0 / shape[-1] ** 0.5)
    return w.to(torch.bfloat16).float()


def seeded_state(named_shapes, seed: int):
    """{name: fp32 tensor holding bf16-representable values} for (name, shape) pairs."""
    return {n: seeded_weight(n, s, seed) for n, s in named_shapes}


def checksum(sd) -> dict:
    """Order-independent fingerprint: float64 sum and abs-sum over everything + the first value of a few named tensors."""
    tot = sum(float(v.double().sum()) for v in sd.values())
    atot = sum(float(v.double().abs().sum()) for v in sd.values())
    names = sorted(sd)
    pick = names[:: max(1, len(names) // 8)][:8]
    return {"sum": tot, "abs_sum": atot, "n": len(names), "first": {n: float(sd[n].reshape(-1)[0]) for n in pick}}
