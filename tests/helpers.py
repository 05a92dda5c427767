"""Shared test helpers (fixture loading, tolerances)."""
import json
import os

import torch
from safetensors import safe_open

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    t = {}
    with safe_open(path, framework="pt") as f:
        meta = json.loads(f.metadata()["meta"])
        for k in f.keys():
            t[k] = f.get_tensor(k)
    return t, meta


def sub(t, prefix):
    return {k[len(prefix):]: v for k, v in t.items() if k.startswith(prefix)}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  — the max-norm relative error SURVEY §8(d) gates on."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
