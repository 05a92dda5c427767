"""Do the text-stream and the vision-stream GEMM chains of one decoder MLP overlap when launched on two HIP streams?
(wave quantisation: the vision GEMMs have 1.1-1.9 waves of 256^2 tiles; the text GEMMs 15.)  Prints sequential vs concurrent time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import kernels as K

BF = torch.bfloat16
H, I, r, rg = 4096, 11008, 1024, 2752
nl, nv = 11760, 4624
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(BF)
h2 = rn(nl + nv, H)
lang = torch.arange(nl, device=dev, dtype=torch.int32)
vis = torch.arange(nl, nl + nv, device=dev, dtype=torch.int32)
wgu, wdown = rn(2 * I, H), rn(H, I)
agu, bg, bu, ad, bd = rn(2 * rg, H), rn(I, rg), rn(I, rg), rn(r, I), rn(H, r)
x_mid = rn(nl + nv, H)
x_out = torch.empty_like(x_mid)
gu = torch.empty(nl, 2 * I, dtype=BF, device=dev); act = torch.empty(nl, I, dtype=BF, device=dev)
tg = torch.empty(nv, 2 * rg, dtype=BF, device=dev); guv = torch.empty(nv, 2 * I, dtype=BF, device=dev)
actv = torch.empty(nv, I, dtype=BF, device=dev); td = torch.empty(nv, r, dtype=BF, device=dev)


def text():
    K.gemm_nt(h2, wgu, a_rows=lang, out=gu)
    K.swiglu(gu[:, :I], gu[:, I:], out=act)
    K.gemm_nt(act, wdown, out=x_out, c_rows=lang, resid=x_mid)


def vision():
    K.gemm_nt(h2, agu, a_rows=vis, out=tg)
    K.gemm_nt_grouped([tg[:, :rg], tg[:, rg:]], [bg, bu], [guv[:, :I], guv[:, I:]])
    K.swiglu(guv[:, :I], guv[:, I:], out=actv)
    K.gemm_nt(actv, ad, out=td)
    K.gemm_nt(td, bd, out=x_out, c_rows=vis, resid=x_mid)


side = torch.cuda.Stream()


def seq():
    text(); vision()


def conc():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        vision()
    text()
    ev2 = torch.cuda.Event(); ev2.record(side)
    main.wait_event(ev2)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(3):
    print(f"text {timeit(text):.0f} us  vision {timeit(vision):.0f} us  sequential {timeit(seq):.0f} us  two streams {timeit(conc):.0f} us", flush=True)
