"""rope_bridge (+ bwd) at the benchmark shape: output digests (two builds must agree bit for bit) and time per launch.
    python tools/rope_ab.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from libra_amd import kernels as K  # noqa: E402

BF = torch.bfloat16


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.contiguous().view(torch.int16).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(BF)
    for (B, S, H) in ((8, 2048, 32), (3, 100, 2)):
        N, D = B * S, H * 128
        qkv0 = rnd(N, 3 * D)
        tb = torch.zeros(N, 64, dtype=BF, device="cuda"); tb[:, :16] = rnd(N, 16)
        w = [rnd(D, 8, sc=0.3) for _ in range(4)]
        flag = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
        flag[:, 1:min(579, S // 2)] = 1                       # one contiguous image per sequence, as in the benchmark batch
        flag = flag.reshape(N).contiguous()
        ang = torch.arange(S, device="cuda")[:, None] * torch.exp(-torch.arange(64, device="cuda") / 10.0)[None]
        cos = torch.cat([ang.cos(), ang.cos()], 1).to(BF).contiguous(); sin = torch.cat([ang.sin(), ang.sin()], 1).to(BF).contiguous()
        qkv = qkv0.clone()
        kc, vc = K.rope_bridge(qkv, tb, *w, flag, cos, sin, S, H)
        d_f = digest(qkv, kc, vc)
        # backward
        dq, dks, dkc, dvs, dvc = (rnd(N, D) for _ in range(5))
        dqkv = torch.empty(N, 3 * D, dtype=BF, device="cuda"); dkb = torch.empty(N, D, dtype=BF, device="cuda")
        dtb = torch.zeros(N, 64, dtype=BF, device="cuda")
        wt = tuple(t.t().contiguous() for t in w)
        K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, H, dqkv, dkb, bridge_b=wt, flag=flag, dtb=dtb)
        d_b = digest(dqkv, dkb); d_t = digest(dtb[:, :16])

        def t(fn, it=20):
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(it):
                fn()
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / it * 1e3
        tf = t(lambda: K.rope_bridge(qkv, tb, *w, flag, cos, sin, S, H))
        tbw = t(lambda: K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, H, dqkv, dkb, bridge_b=wt, flag=flag, dtb=dtb))
        print(f"B={B} S={S} H={H}: fwd {d_f} {tf:7.1f} us | bwd {d_b} dtb {d_t} {tbw:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
