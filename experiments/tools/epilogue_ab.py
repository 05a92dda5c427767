"""What a fused epilogue costs on each GEMM tile structure (256^2 one workgroup per CU vs W = 256x128 two per CU): the same problem
plain, with a residual, and with bias + quick-GELU + pre-activation store (the ViT fc1 epilogue: two outputs), pinned to each
structure.  The question behind it (round-3 review, item 3): does a second workgroup on the CU hide an epilogue's cost?

    python tools/epilogue_ab.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from libra_amd import kernels as K  # noqa: E402

SHAPES = [(18464, 4096, 1024), (18464, 1024, 4096), (11760, 4096, 11008), (4624, 4096, 1024), (4624, 11008, 1024)]
TILES = [("256", 2), ("W", 3), ("auto", 0)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, Kd in SHAPES:
        a = torch.randn(M, Kd, device="cuda", generator=g).to(torch.bfloat16)
        b = (torch.randn(N, Kd, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        modes = {"plain": lambda t: K.gemm_nt(a, b, out=out, tile=t),
                 "resid": lambda t: K.gemm_nt(a, b, out=out, resid=res, tile=t),
                 "bias+qgelu+preact": lambda t: K.gemm_nt(a, b, out=out, bias=bias, quick_gelu=True, preact_out=pre, tile=t),
                 "qgelu_grad": lambda t: K.gemm_nt(a, b, out=out, qgelu_grad_of=res, tile=t)}
        us = {}
        for rnd in range(3):
            for mname, fn in modes.items():
                for tname, t in TILES:
                    fn(t); torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(iters):
                        fn(t)
                    e.record(); torch.cuda.synchronize()
                    k = (mname, tname)
                    us[k] = min(us.get(k, 1e30), s.elapsed_time(e) / iters * 1e3)
        print(f"{M}x{N}x{Kd}")
        for mname in modes:
            base = {t: us[("plain", t)] for t, _ in TILES}
            print("   " + f"{mname:20s} " + "  ".join(f"{t}={us[(mname, t)]:7.1f}us (+{us[(mname, t)] - base[t]:5.1f})" for t, _ in TILES), flush=True)


if __name__ == "__main__":
    main()
