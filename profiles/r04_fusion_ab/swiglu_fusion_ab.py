"""Same-process A/B of the headline step with and without SwiGLU fused into the text gate | up GEMM's epilogue
(decoder_engine.FUSE_SWIGLU), alternating; also checks loss / gradient-norm agreement of the two paths.

    python tools/swiglu_fusion_ab.py [reps] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from libra_amd import decoder_engine as DE  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda", 0)
    w = bench.make_bridge(dev, 8, 2048, 1, None)

    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            loss = w.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)

    def gnorm():
        return float(torch.sqrt(sum((p.grad.float() ** 2).sum() for _, p in w.named if p.grad is not None)))

    for fused in (True, False):
        DE.FUSE_SWIGLU = fused
        run(2)
    for r in range(reps):
        for fused in (False, True):
            DE.FUSE_SWIGLU = fused
            ms, loss = run(steps)
            print(f"fused={int(fused)} ms_per_step {ms:.2f} loss {loss:.6f} grad_norm {gnorm():.6f}", flush=True)


if __name__ == "__main__":
    main()
