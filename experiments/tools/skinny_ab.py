"""The generation step's weight-streaming GEMMs at the Libra-11B shapes: output digests (two builds must agree bit for bit) and time
per launch over a pool of distinct weight buffers (so the Infinity Cache cannot serve the weights).
    python tools/skinny_ab.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from libra_amd import kernels as K  # noqa: E402

BF = torch.bfloat16


def digest(t):
    return hashlib.sha256(t.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(BF)
    POOL = 6
    total = 0.0
    for name, M, N, Kd, sw in (("o", 8, 4096, 4096, False), ("qkv", 8, 12352, 4096, False), ("down", 8, 4096, 11008, False),
                               ("gate|up+swiglu", 8, 11008, 4096, True), ("o M=5 gathered", 5, 4096, 4096, False),
                               ("small", 8, 520, 1024, False), ("sw ragged", 3, 1002, 576, True)):
        a = rnd(M + 3, Kd, sc=0.5)
        rows = torch.arange(M + 3, device="cuda")[torch.randperm(M + 3, generator=torch.Generator().manual_seed(1))[:M]].to(torch.int32)
        ws = [rnd(2 * N if sw else N, Kd, sc=0.05) for _ in range(POOL)]
        res = rnd(M, N)
        if sw:
            fn = lambda w: K.gemm_swiglu_skinny(a, w, a_rows=rows)
        else:
            fn = lambda w: K.gemm_nt(a, w, a_rows=rows, c_rows=torch.arange(M, device="cuda", dtype=torch.int32), out=torch.empty(M, N, dtype=BF, device="cuda"), resid=res)
        d = digest(fn(ws[0]))
        for w in ws:
            fn(w)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 5
        s.record()
        for _ in range(it):
            for w in ws:
                fn(w)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / (it * POOL) * 1e3
        gb = (2 * N if sw else N) * Kd * 2 / 1e9
        print(f"{name:18s} M={M} N={N} K={Kd}: {d}  {us:7.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s", flush=True)
        if name in ("o", "qkv", "down", "gate|up+swiglu"):
            total += us
    print(f"four weight streams of a layer: {total:.1f} us", flush=True)


if __name__ == "__main__":
    main()
