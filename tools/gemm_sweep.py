"""Every GEMM shape of the benchmarked steps under each tile structure (AUTO = the library's cost model, 128, 256, W) - the table
the planner's cost model in gemm_bf16.hip is fitted on (profiles/r04_gemm_tile_fit.txt).

    python tools/gemm_sweep.py [headline|vit|cfg3|cfg4|all] [iters]

Shapes are the `roofline.by_shape` rows of the bench lines (M x N x K, aT / bT = reduction-major operands, Gx[..] = grouped launch);
each (shape, structure) is timed in `rounds` interleaved rounds of `iters` back-to-back launches (HIP events), the minimum is kept.
One JSON line per shape on stdout, a readable table on stderr."""
import faulthandler
import json
import os
import re
import sys

faulthandler.enable()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from libra_amd import kernels as K  # noqa: E402

# (shape string, launches per step) of the headline step (configs[2]), the ViT leg (configs[1]) and the configs[3] / [4] shaped steps
HEADLINE = [("11760x22016x4096", 32), ("11760x4096x22016 bT", 32), ("11760x12352x4096", 32), ("11760x4096x12352 bT", 32),
            ("11760x11008x4096 bT", 32), ("11760x4096x11008", 32), ("2x[4624x2752x11008 bT]", 32), ("11008x2752x4672 aT bT", 64),
            ("2x[4624x11008x2752]", 32), ("11760x4096x4096", 32), ("11760x4096x4096 bT", 32), ("5504x4096x4672 aT bT", 32),
            ("4624x4096x5504 bT", 32), ("4624x5504x4096", 32), ("4624x4096x3136 bT", 32), ("3x[4624x4096x1024]", 32),
            ("4624x11008x1024 bT", 32), ("4624x4096x1024", 64), ("4624x1024x4096 bT", 64), ("1024x11008x4672 aT bT", 32),
            ("4624x1024x11008", 32), ("3136x4096x4672 aT bT", 32), ("3x[4096x1024x4672 aT bT]", 32), ("4624x3136x4096", 32),
            ("3x[4624x1024x4096 bT]", 32), ("2x[4096x1024x4672 aT bT]", 32), ("4624x1024x4096", 32), ("4624x4096x1024 bT", 32),
            ("1024x4096x4672 aT bT", 32), ("4616x1024x4096", 24), ("4616x4096x1024", 24), ("4616x3072x1024", 24),
            ("4616x1024x1024", 24)]
VIT = [("18464x4096x1024 bT", 23), ("18464x1024x4096", 24), ("18464x4096x1024", 24), ("18464x1024x4096 bT", 23),
       ("1024x4096x18496 aT bT", 23), ("4096x1024x18496 aT bT", 23), ("18464x3072x1024", 24), ("18464x1024x3072 bT", 23),
       ("3072x1024x18496 aT bT", 23), ("18464x1024x1024", 24), ("1024x1024x18496 aT bT", 23), ("18464x1024x1024 bT", 23)]
CFG3 = [("976x4096x22016 bT", 32), ("976x22016x4096", 32), ("976x4096x12352 bT", 32), ("976x4096x11008", 32), ("976x12352x4096", 32),
        ("976x11008x4096 bT", 32), ("976x4096x4096", 32), ("976x4096x4096 bT", 32)]
CFG4 = [("7036x22016x4096", 64), ("22016x4096x7040 aT bT", 32), ("7036x4096x22016 bT", 32), ("7036x4096x4096", 64),
        ("1156x1024x11008", 64), ("2x[1156x11008x2752]", 64), ("2x[1156x2752x11008 bT]", 32), ("11008x2752x1216 aT bT", 64),
        ("1156x5504x4096", 64), ("1156x3136x4096", 64), ("1156x1024x4096 bT", 64), ("1156x1024x4096", 64), ("1156x4096x5504 bT", 32),
        ("3x[1156x4096x1024]", 64), ("1156x4096x1024", 96), ("5504x4096x1216 aT bT", 32), ("3x[1156x1024x4096 bT]", 32)]
TEXT = HEADLINE[:6] + HEADLINE[9:11] + [("4624x4096x1024", 64), ("4624x5504x4096", 32), ("11008x2752x4672 aT bT", 64)]
SETS = {"text": TEXT, "headline": HEADLINE, "vit": VIT, "cfg3": CFG3, "cfg4": CFG4}
TILES = [("auto", 0), ("128", 1), ("256", 2), ("W", 3)]        # (+ ("X", 4) with experiments/r04/gemm_bf16_x_*.hip built in)
if os.environ.get("SWEEP_TILES"):
    TILES = [t for t in TILES if t[0] in os.environ["SWEEP_TILES"].split(",")]


def parse(spec):
    m = re.match(r"(?:(\d)x\[)?(\d+)x(\d+)x(\d+)((?: aT)?)((?: bT)?)\]?$", spec)
    G = int(m.group(1)) if m.group(1) else 1
    return G, int(m.group(2)), int(m.group(3)), int(m.group(4)), bool(m.group(5)), bool(m.group(6))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rounds = 3
    names = list(SETS) if which == "all" else [which]
    gen = torch.Generator(device="cuda").manual_seed(0)
    for setname in names:
        tot = {t: 0.0 for t, _ in TILES}
        best_tot = 0.0
        for spec, launches in SETS[setname]:
            G, M, N, Kd, a_t, b_t = parse(spec)
            Kp = (Kd + 63) // 64 * 64
            As = [torch.randn((Kp, M) if a_t else (M, Kp), device="cuda", generator=gen).to(torch.bfloat16) for _ in range(G)]
            Bs = [torch.randn((Kp, N) if b_t else (N, Kp), device="cuda", generator=gen).to(torch.bfloat16) for _ in range(G)]
            Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(G)]

            def run(tile):
                if G == 1:
                    K.gemm_nt(As[0], Bs[0], out=Cs[0], a_t=a_t, b_t=b_t, tile=tile)
                else:
                    K.gemm_nt_grouped(As, Bs, Cs, a_t=a_t, b_t=b_t, tile=tile)

            us = {}
            for r in range(rounds):
                for tname, tile in TILES:
                    if r == 0:
                        print(f"[run] {spec} tile={tname}", file=sys.stderr, flush=True)
                    run(tile)
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); s.record()
                    for _ in range(iters):
                        run(tile)
                    e.record(); torch.cuda.synchronize()
                    t = s.elapsed_time(e) / iters * 1e3
                    us[tname] = min(us.get(tname, 1e30), t)
            flop = 2.0 * M * N * Kp * G
            row = dict(set=setname, shape=spec, launches=launches, us={k: round(v, 1) for k, v in us.items()},
                       tflops={k: round(flop / v / 1e6, 1) for k, v in us.items()})
            best = min(us, key=lambda k: us[k] if k != "auto" else 1e30)
            row["best"] = best
            print(json.dumps(row), flush=True)
            for k in us:
                tot[k] += us[k] * launches / 1e3
            best_tot += us[best] * launches / 1e3
            print(f"{setname:8s} {spec:28s} x{launches:3d} " + " ".join(f"{k}={us[k]:8.1f}us/{flop / us[k] / 1e6:6.0f}TF" for k in us) +
                  f"  best={best}", file=sys.stderr, flush=True)
            del As, Bs, Cs
        print(json.dumps(dict(set=setname, total_ms_per_step={k: round(v, 2) for k, v in tot.items()}, best_of_pinned_ms=round(best_tot, 2))),
              flush=True)
        print(f"{setname}: ms/step by structure {tot}  best-of-pinned {best_tot:.2f}", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
