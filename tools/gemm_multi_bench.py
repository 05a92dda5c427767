"""Multi-problem GEMM launches against the separate launches they replace, at the decoder's benchmark shape
(B = 8 x 2048: 11 760 text rows, 4 624 vision rows; H 4096, I 11008, r 1024, rg 2752).

    python tools/gemm_multi_bench.py [iters]

Each group is what one stage of decoder_engine.layer_forward / layer_backward issues; "separate" = the launches of the round-5
engine (gemm_nt with the library's tile planner, gemm_nt_grouped), "multi" = one libra_gemm_bf16_multi launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

BF = torch.bfloat16
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NL, NV, H, I, R, RG = 11760, 4624, 4096, 11008, 1024, 2752
N = NL + NV
dev = "cuda"


def rn(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(BF)


g = torch.Generator().manual_seed(0)
perm = torch.randperm(N, generator=g)
lang = perm[:NL].sort().values.to(torch.int32).to(dev)
vis = perm[NL:].sort().values.to(torch.int32).to(dev)
NVP = K.round_up(NV, 64)


def P(a, b, **kw):
    return (a, b, kw)


def groups():
    x = rn(N, H, scale=0.5)
    out = {}
    # ---------------- forward
    wqkv = rn(3 * H + 64, H, scale=0.02)
    t = rn(NV, 3 * R, scale=0.5)
    wb = [rn(H, R, scale=0.02) for _ in range(3)]
    qkv = torch.empty((N, 3 * H + 64), dtype=BF, device=dev)
    out["F2 text qkv + 3 vision B"] = (
        [P(x, wqkv, out=qkv, a_rows=lang, c_rows=lang)] + [P(t[:, j * R:(j + 1) * R], wb[j], out=qkv[:, j * H:(j + 1) * H], c_rows=vis) for j in range(3)],
        [[0], [1, 2, 3]])
    wo = rn(H, H, scale=0.02)
    to = rn(NV, R, scale=0.5)
    wob = rn(H, R, scale=0.02)
    xm = torch.empty((N, H), dtype=BF, device=dev)
    out["F4 text o + vision o_B"] = ([P(x, wo, out=xm, a_rows=lang, c_rows=lang, resid=x), P(to, wob, out=xm, c_rows=vis, resid=x)], [[0], [1]])
    wgu = rn(2 * I, H, scale=0.02)
    tg = rn(NV, 2 * RG, scale=0.5)
    wgb = [rn(I, RG, scale=0.02) for _ in range(2)]
    gu = torch.empty((NL, 2 * I), dtype=BF, device=dev)
    guv = torch.empty((NV, 2 * I), dtype=BF, device=dev)
    out["F6 text gate|up + 2 vision B"] = (
        [P(x, wgu, out=gu, a_rows=lang)] + [P(tg[:, j * RG:(j + 1) * RG], wgb[j], out=guv[:, j * I:(j + 1) * I]) for j in range(2)],
        [[0], [1, 2]])
    act = rn(NL, I, scale=0.5)
    wd = rn(H, I, scale=0.02)
    td = rn(NV, R, scale=0.5)
    wdb = rn(H, R, scale=0.02)
    xo = torch.empty((N, H), dtype=BF, device=dev)
    out["F8 text down + vision down_B"] = ([P(act, wd, out=xo, c_rows=lang, resid=x), P(td, wdb, out=xo, c_rows=vis, resid=x)], [[0], [1]])
    # ---------------- backward (frozen language: text dgrads only; vision dgrads + weight gradients)
    dxo_v = K.alloc_rows(NV, H, dev)[:NV]; dxo_v.copy_(rn(NV, H, scale=0.5))
    dtd = K.alloc_rows(NV, R, dev)[:NV]; dtd.copy_(rn(NV, R, scale=0.5))
    actv = K.alloc_rows(NV, I, dev)[:NV]; actv.copy_(rn(NV, I, scale=0.5))
    tdp = K.alloc_rows(NV, R, dev)[:NV]; tdp.copy_(td)
    full = lambda tt: torch.as_strided(tt, (K.round_up(tt.shape[0], 64), tt.shape[1]), tt.stride(), tt.storage_offset())
    wda = rn(R, I, scale=0.02)
    dact = torch.empty((NL, I), dtype=BF, device=dev)
    dactv = torch.empty((NV, I), dtype=BF, device=dev)
    out["B1 text dact + vision dactv + dW down_B + dW down_A"] = (
        [P(x, wd, out=dact, b_t=True, a_rows=lang), P(dtd, wda, out=dactv, b_t=True),
         P(full(dxo_v), full(tdp), a_t=True, b_t=True), P(full(dtd), full(actv), a_t=True, b_t=True)],
        [[0], [1], [2], [3]])
    dgu = rn(NL, 2 * I, scale=0.5)
    dguv = K.alloc_rows(NV, 2 * I, dev)[:NV]; dguv.copy_(rn(NV, 2 * I, scale=0.5))
    tgp = K.alloc_rows(NV, 2 * RG, dev)[:NV]; tgp.copy_(tg)
    dh2 = torch.empty((N, H), dtype=BF, device=dev)
    dtg = torch.empty((NV, 2 * RG), dtype=BF, device=dev)
    out["B2 text dh2 + 2 vision dtg + 2 dW gate/up_B"] = (
        [P(dgu, wgu, out=dh2, b_t=True, c_rows=lang)]
        + [P(dguv[:, j * I:(j + 1) * I], wgb[j], out=dtg[:, j * RG:(j + 1) * RG], b_t=True) for j in range(2)]
        + [P(full(dguv[:, j * I:(j + 1) * I]), full(tgp[:, j * RG:(j + 1) * RG]), a_t=True, b_t=True) for j in range(2)],
        [[0], [1, 2], [3], [4]])
    agu = rn(2 * RG, H, scale=0.02)
    dtgp = K.alloc_rows(NV, 2 * RG, dev)[:NV]; dtgp.copy_(rn(NV, 2 * RG, scale=0.5))
    h2v = K.alloc_rows(NV, H, dev)[:NV]; h2v.copy_(rn(NV, H, scale=0.5))
    out["B3 vision dh2 + dW agu"] = ([P(dtgp, agu, out=dh2, b_t=True, c_rows=vis), P(full(dtgp), full(h2v), a_t=True, b_t=True)], [[0], [1]])
    dto = K.alloc_rows(NV, R, dev)[:NV]; dto.copy_(rn(NV, R, scale=0.5))
    woa = rn(R, H, scale=0.02)
    top = K.alloc_rows(NV, R, dev)[:NV]; top.copy_(to)
    ov = K.alloc_rows(NV, H, dev)[:NV]; ov.copy_(rn(NV, H, scale=0.5))
    do = torch.empty((N, H), dtype=BF, device=dev)
    out["B4 text do + vision do + dW o_B + dW o_A"] = (
        [P(x, wo, out=do, b_t=True, a_rows=lang, c_rows=lang), P(dto, woa, out=do, b_t=True, c_rows=vis),
         P(full(dxo_v), full(top), a_t=True, b_t=True), P(full(dto), full(ov), a_t=True, b_t=True)],
        [[0], [1], [2], [3]])
    dqkvt = rn(N, 3 * H + 64, scale=0.5)
    dqkv_v = K.alloc_rows(NV, 3 * H, dev)[:NV]; dqkv_v.copy_(rn(NV, 3 * H, scale=0.5))
    tp = K.alloc_rows(NV, 3 * R, dev)[:NV]; tp.copy_(t)
    dh = torch.empty((N, H), dtype=BF, device=dev)
    dt = torch.empty((NV, 3 * R), dtype=BF, device=dev)
    out["B5 text dh + 3 vision dt + 3 dW qkv_B"] = (
        [P(dqkvt, wqkv, out=dh, b_t=True, a_rows=lang, c_rows=lang)]
        + [P(dqkv_v[:, j * H:(j + 1) * H], wb[j], out=dt[:, j * R:(j + 1) * R], b_t=True) for j in range(3)]
        + [P(full(dqkv_v[:, j * H:(j + 1) * H]), full(tp[:, j * R:(j + 1) * R]), a_t=True, b_t=True) for j in range(3)],
        [[0], [1, 2, 3], [4, 5, 6]])
    aqkv = rn(3 * R + 64, H, scale=0.02)
    dte = K.alloc_rows(NV, 3 * R + 64, dev)[:NV]; dte.copy_(rn(NV, 3 * R + 64, scale=0.5))
    hv = K.alloc_rows(NV, H, dev)[:NV]; hv.copy_(rn(NV, H, scale=0.5))
    out["B6 vision dh + dW aqkv"] = ([P(dte, aqkv, out=dh, b_t=True, c_rows=vis), P(full(dte), full(hv), a_t=True, b_t=True)], [[0], [1]])
    return out


def run_separate(probs, launches):
    for idx in launches:
        if len(idx) == 1:
            a, b, kw = probs[idx[0]]
            K.gemm_nt(a, b, **kw)
        else:
            kw0 = dict(probs[idx[0]][2])
            outs = [probs[i][2]["out"] for i in idx]
            kw0.pop("out")
            K.gemm_nt_grouped([probs[i][0] for i in idx], [probs[i][1] for i in idx], outs, **kw0)


def run_multi(probs):
    K.gemm_multi([K.gemm_spec(a, b, **kw) for a, b, kw in probs])


def timeit(fn):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(IT):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / IT * 1e3


tot_s = tot_m = 0.0
for name, (probs, launches) in groups().items():
    for p in probs:                        # weight gradients without an explicit out: allocate once
        a, b, kw = p
        if "out" not in kw:
            kw["out"] = torch.empty((a.shape[1], b.shape[1]), dtype=BF, device=dev)
    fl = 0.0
    for a, b, kw in probs:
        sp = K.gemm_spec(a, b, **kw)
        fl += sp.work[0]
    us_s = timeit(lambda: run_separate(probs, launches))
    us_m = timeit(lambda: run_multi(probs))
    us_s2 = timeit(lambda: run_separate(probs, launches))
    us_m2 = timeit(lambda: run_multi(probs))
    us_s, us_m = min(us_s, us_s2), min(us_m, us_m2)
    tot_s += us_s; tot_m += us_m
    print(f"{name:55s} separate {us_s:8.1f} us ({fl / us_s / 1e6:7.1f} TF)   multi {us_m:8.1f} us ({fl / us_m / 1e6:7.1f} TF)   {us_m - us_s:+8.1f} us", flush=True)
print(f"{'sum per layer':55s} separate {tot_s:8.1f} us   multi {tot_m:8.1f} us   {tot_m - tot_s:+8.1f} us  (x32 layers = {(tot_m - tot_s) * 32 / 1e3:+.2f} ms / step)")
