"""Forward / backward schedule of the CLIP ViT encoder on the gfx950 kernels.

This is the host-side "graph" of the hot path a1-a7 of SURVEY.md §8: which kernel runs on which
buffer and what is saved for backward.  No operand is ever copied into another layout: dgrad reads the weights and
wgrad reads dY / X where they lie, as reduction-major GEMM operands (LDS transpose loads).  All math is in
``libra_amd/csrc``; torch supplies memory and the stream only.

Reference semantics being reproduced (file:line relative to /root/reference/libra/models/clip):
  modeling_clip.py:193-228 embeddings, :893 pre_layrnorm, :390-428 encoder layer (pre-LN residual),
  :287-363 attention (q scaled by hd^-0.5, softmax over keys), :374-378 MLP with quick_gelu,
  :664-694 collection of all hidden states.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import dp
from . import kernels as K

BF16 = torch.bfloat16
P = "vision_model."


@dataclass
class VitDims:
    hidden: int
    inter: int
    layers: int
    heads: int
    patch: int
    image: int
    channels: int = 3
    eps: float = 1e-5

    @property
    def grid(self):
        return self.image // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1

    @property
    def kpe(self):           # im2col width, padded to the GEMM's K granule
        return K.round_up(self.channels * self.patch * self.patch, 64)


class _Packed:
    """Device-side operand copies derived from the parameters: fused [q;k;v] weight/bias and the zero-padded patch-embedding
    matrix.  Allocated once and refreshed IN PLACE: a slice is re-copied when its source parameter is trainable (an optimizer
    may have updated it through `.data`, which bumps neither `_version` nor `data_ptr` - DeepSpeed, master-weight
    optimizers) or when its (data_ptr, _version) key changed.  Version counters alone are never trusted."""

    def __init__(self):
        self.fwd = None
        self._slices: List[tuple] = []
        self._keys: Dict[int, tuple] = {}

    def get(self, params: Dict[str, torch.Tensor], dims: VitDims):
        dev = params[P + "embeddings.patch_embedding.weight"].device
        if self.fwd is None or self.fwd["wpe"].device != dev:
            self._build(dims, dev)
            force = True
        else:
            force = False
        with torch.no_grad():
            dsts, srcs = [], []
            for k, (dst, name, shape) in enumerate(self._slices):
                p = params[name]
                key = (p.data_ptr(), p._version)
                if force or p.requires_grad or self._keys.get(k) != key:
                    dsts.append(dst)
                    srcs.append(p.detach().reshape(shape))
                    self._keys[k] = key
            if dsts:
                torch._foreach_copy_(dsts, srcs)              # a few multi-tensor launches instead of 145 small copies per forward
        return self.fwd

    def _build(self, dims: VitDims, dev):
        D = dims.hidden
        kk = dims.channels * dims.patch * dims.patch
        wpad = torch.zeros((D, dims.kpe), dtype=BF16, device=dev)
        self.fwd = {"wpe": wpad, "layers": []}
        self._slices = [(wpad[:, :kk], P + "embeddings.patch_embedding.weight", (D, kk))]
        self._keys = {}
        for i in range(dims.layers):
            pre = f"{P}encoder.layers.{i}.self_attn."
            w = torch.empty((3 * D, D), dtype=BF16, device=dev)
            bq = torch.empty((3 * D,), dtype=BF16, device=dev)
            for j, n in enumerate("qkv"):
                self._slices.append((w[j * D:(j + 1) * D], pre + f"{n}_proj.weight", (D, D)))
                self._slices.append((bq[j * D:(j + 1) * D], pre + f"{n}_proj.bias", (D,)))
            self.fwd["layers"].append({"wqkv": w, "bqkv": bq})


def pack_forward(params: Dict[str, torch.Tensor], dims: VitDims):
    return _Packed().get(params, dims)


def forward(params: Dict[str, torch.Tensor], packed, pixel: torch.Tensor, dims: VitDims, *, save: bool,
            n_layers: Optional[int] = None):
    """-> (hidden_states: list of L+1 tensors [B*T, D] bf16, saved-for-backward dict or None)."""
    if pixel.dim() != 4 or pixel.shape[1] != dims.channels or pixel.shape[2] != dims.image or pixel.shape[3] != dims.image:
        raise ValueError(f"pixel_values must be [B,{dims.channels},{dims.image},{dims.image}], got {tuple(pixel.shape)}")
    B = pixel.shape[0]
    T, D, H = dims.tokens, dims.hidden, dims.heads
    if D // H != 64 or D % H:
        raise ValueError("the gfx950 ViT attention kernel is specialised for head_dim 64 (CLIP ViT-L/14)")
    scale = 64 ** -0.5
    L = dims.layers if n_layers is None else n_layers
    pixel = pixel.to(BF16).contiguous()
    cols = K.patch_im2col(pixel, dims.patch, dims.kpe)
    patches = K.gemm_nt(cols, packed["wpe"])
    emb, x, mean0, rstd0 = K.vit_embed_ln(patches, params[P + "embeddings.class_embedding"],
                                          params[P + "embeddings.position_embedding.weight"],
                                          params[P + "pre_layrnorm.weight"], params[P + "pre_layrnorm.bias"], B, T,
                                          dims.eps, save=save)
    hs: List[torch.Tensor] = [x]
    saved = {"B": B, "cols": cols, "emb": emb, "mean0": mean0, "rstd0": rstd0, "layers": []} if save else None
    for i in range(L):
        pre = f"{P}encoder.layers.{i}."
        pk = packed["layers"][i]
        M = B * T
        rows = (lambda c: K.alloc_rows(M, c, x.device)[:M]) if save else (lambda c: None)
        xn1, m1, r1 = K.layernorm_fwd(x, params[pre + "layer_norm1.weight"], params[pre + "layer_norm1.bias"], dims.eps,
                                      save_stats=save, out=rows(D))
        qkv = K.gemm_nt(xn1, pk["wqkv"], bias=pk["bqkv"])
        o_lo = torch.empty((M, D), dtype=BF16, device=x.device) if save else None      # rounding residual of o: the backward's D term
        o, lse = K.vit_attn_fwd(qkv, B, T, H, scale, need_lse=save, out=rows(D), out_lo=o_lo)
        x_mid = K.gemm_nt(o, params[pre + "self_attn.out_proj.weight"], bias=params[pre + "self_attn.out_proj.bias"],
                          resid=x)
        xn2, m2, r2 = K.layernorm_fwd(x_mid, params[pre + "layer_norm2.weight"], params[pre + "layer_norm2.bias"],
                                      dims.eps, save_stats=save, out=rows(D))
        hpre = torch.empty((B * T, dims.inter), dtype=BF16, device=x.device) if save else None
        act = K.gemm_nt(xn2, params[pre + "mlp.fc1.weight"], bias=params[pre + "mlp.fc1.bias"], quick_gelu=True,
                        preact_out=hpre, out=rows(dims.inter))
        x_out = K.gemm_nt(act, params[pre + "mlp.fc2.weight"], bias=params[pre + "mlp.fc2.bias"], resid=x_mid)
        if save:
            saved["layers"].append(dict(x=x, xn1=xn1, m1=m1, r1=r1, qkv=qkv, o=o, o_lo=o_lo, lse=lse, x_mid=x_mid, xn2=xn2,
                                        m2=m2, r2=r2, hpre=hpre, act=act))
        x = x_out
        hs.append(x)
    return hs, saved


def _full(t: torch.Tensor, m_pad: int) -> torch.Tensor:
    """The zero-padded [m_pad, C] allocation behind a [M, C] row view made by K.alloc_rows."""
    return torch.as_strided(t, (m_pad, t.shape[1]), t.stride(), t.storage_offset())


def _wgrad(dy: torch.Tensor, x: torch.Tensor, m_pad: int, name: Optional[str] = None) -> torch.Tensor:
    """dW[N_out, K_in] = sum_m dY[m, N_out] X[m, K_in]: both operands are read token-major as they lie in HBM.
    With a capturing data-parallel gradient store the GEMM writes straight into the parameter's bucket slot."""
    return K.gemm_nt(_full(dy, m_pad), _full(x, m_pad), a_t=True, b_t=True, out=dp.grad_out(name) if name else None)


# Round 6: each dgrad of the backward shares ONE multi-problem launch (K.gemm_multi) with the weight gradient that reads the same dY:
# the weight gradient (64 / 16 / 48 tiles of 256^2 with an 18 496-step reduction at bs 32) is cut into K slices - tile-list entries of
# its own - that fill the dgrad's last wave instead of a K-sliced launch of their own.  False = the round-5 schedule (bench.py --no-multi).
MULTI = True


def _wgrad_slices(n_out: int, n_in: int, k_tokens: int) -> int:
    """K slices for a weight gradient inside a multi-problem launch: about one slice per compute unit, at least 8 K tiles each."""
    tiles = ((n_out + 255) // 256) * ((n_in + 255) // 256)
    if tiles >= 128 or n_in % 8:
        return 1
    return max(1, min(256 // tiles, (k_tokens // 64) // 8, 32))


def _wgrad_spec(dy: torch.Tensor, x: torch.Tensor, m_pad: int, name: Optional[str] = None):
    return K.gemm_spec(_full(dy, m_pad), _full(x, m_pad), a_t=True, b_t=True, out=dp.grad_out(name) if name else None,
                       splitk=_wgrad_slices(dy.shape[1], x.shape[1], m_pad))


def emit_group(name: str, n_layers: int) -> int:
    """Backward-order group of a parameter for the data-parallel bucket layout (dp.GradBuckets): the weight matrices of
    encoder layer i are finished at step L-1-i of the backward; every bias / LayerNorm vector (one fp32 arena converted at
    the end) and the embedding gradients come last."""
    parts = name.split(".")
    if "layers" in parts and name.endswith(("proj.weight", "fc1.weight", "fc2.weight")):
        return n_layers - 1 - int(parts[parts.index("layers") + 1])
    return n_layers


def backward(params, packed, saved, dhs: Sequence[Optional[torch.Tensor]], dims: VitDims,
             *, need_pixel_grad: bool = True):
    """Given d(loss)/d(hidden_states[i]) (None = zero) return (d_pixel or None, {param name: bf16 grad}).

    Residual-stream gradient flows from the last layer that has a non-zero cotangent downwards; layers
    above it are skipped entirely (their gradient is exactly zero)."""
    B = saved["B"]
    T, D, H, I = dims.tokens, dims.hidden, dims.heads, dims.inter
    M = B * T
    m_pad = K.round_up(M, 64)
    scale = 64 ** -0.5
    dev = saved["cols"].device
    grads: Dict[str, torch.Tensor] = {}
    L = len(saved["layers"])
    top = max((i for i, g in enumerate(dhs) if g is not None), default=-1)
    if top < 0:
        return None, grads
    # Gradient buffers that later serve as reduction-major wgrad operands (zero pad rows) are allocated ONCE per backward and
    # reused by every layer (one stream: a layer's kernels are enqueued after every reader of the previous contents).
    dev_rows = lambda c: K.alloc_rows(M, c, dev)[:M]
    dx_ring = [dev_rows(D), dev_rows(D)]                       # dx of layer i is read while layer i-1 writes its own
    dh_buf, dxmid_buf, dqkv_buf = dev_rows(I), dev_rows(D), dev_rows(3 * D)
    dx = dx_ring[0]
    dx.copy_(dhs[top].reshape(M, D))
    # one zeroed fp32 arena for every small (bias / LayerNorm) gradient of this backward; converted to bf16 once
    n_small = (min(top, L)) * (9 * D + I) + 2 * D
    arena = torch.zeros(n_small, dtype=torch.float32, device=dev)
    small: List = []          # (param name, offset, numel)
    cursor = [0]

    def f32(n, name=None):
        o = cursor[0]
        cursor[0] += n
        if name is not None:
            small.append((name, o, n))
        return arena[o:o + n]

    def param_grads(wname, bname, dy, x, nb, bias_slice=None, dgrad=None):
        """weight gradient (+ the bias gradient = column sum of dy, unless the kernel that produced dy already summed it
        into `bias_slice`).  dgrad: (weight, kwargs) of the dgrad GEMM that reads the same dy - with MULTI both go out as one
        launch; -> the dgrad's result (or None)."""
        bslice = f32(nb, bname) if bias_slice is None else None

        # (Weight gradients used to run on a second stream; once the kernels were tuned that overlap measured as no gain -
        # 56.6 vs 56.5 ms/step - so the whole backward is one stream and per-launch timings mean what they say.)
        dx_ = None
        if MULTI and dgrad is not None and K.gemm_multi_ok(dy.device):
            dx_, grads[wname] = K.gemm_multi([K.gemm_spec(dy, dgrad[0], b_t=True, **dgrad[1]), _wgrad_spec(dy, x, m_pad, wname)])
        else:
            grads[wname] = _wgrad(dy, x, m_pad, wname)
            if dgrad is not None:
                dx_ = K.gemm_nt(dy, dgrad[0], b_t=True, **dgrad[1])
        if bslice is not None:
            K.colsum(dy, bslice)
        return dx_

    emitted: set = set()
    dx_sum = None          # fp32 column sum of dx when the LayerNorm backward that produced dx already took it
    for i in range(min(top, L) - 1, -1, -1):
        pre = f"{P}encoder.layers.{i}."
        s = saved["layers"][i]
        # ---- MLP: x_out = x_mid + fc2(quick_gelu(fc1(LN2(x_mid))))
        dh = param_grads(pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", dx, s["act"], D, bias_slice=dx_sum,
                         dgrad=(params[pre + "mlp.fc2.weight"], dict(qgelu_grad_of=s["hpre"], out=dh_buf)))
        dxn2 = param_grads(pre + "mlp.fc1.weight", pre + "mlp.fc1.bias", dh, s["xn2"], I,
                           dgrad=(params[pre + "mlp.fc1.weight"], {}))                        # [M, D]
        dg2, dbt2 = f32(D, pre + "layer_norm2.weight"), f32(D, pre + "layer_norm2.bias")
        bo = f32(D, pre + "self_attn.out_proj.bias")              # = column sum of dx_mid, taken by the LN backward itself
        dx_mid = K.layernorm_bwd(dxn2, s["x_mid"], params[pre + "layer_norm2.weight"], s["m2"], s["r2"], dres=dx,
                                 dgamma=dg2, dbeta=dbt2, dxsum=bo, out=dxmid_buf)
        # ---- attention: x_mid = x + out_proj(attn(LN1(x)))
        do = param_grads(pre + "self_attn.out_proj.weight", pre + "self_attn.out_proj.bias", dx_mid, s["o"], D, bias_slice=bo,
                         dgrad=(params[pre + "self_attn.out_proj.weight"], {}))                # [M, D]
        dqkv = K.vit_attn_bwd(s["qkv"], s["o"], do, s["lse"], B, T, H, scale, out_dqkv=dqkv_buf, out_lo=s["o_lo"])
        dxn1 = param_grads(pre + "self_attn.qkv_packed", None, dqkv, s["xn1"], 3 * D,            # [3D, D]
                           dgrad=(packed["layers"][i]["wqkv"], {}))
        dwqkv = grads.pop(pre + "self_attn.qkv_packed")
        o0 = cursor[0] - 3 * D
        for j, n in enumerate("qkv"):
            grads[pre + f"self_attn.{n}_proj.weight"] = dwqkv[j * D:(j + 1) * D]
            small.append((pre + f"self_attn.{n}_proj.bias", o0 + j * D, D))
        dg1, dbt1 = f32(D, pre + "layer_norm1.weight"), f32(D, pre + "layer_norm1.bias")
        # this dx is the dY of layer i-1's fc2: its column sum is that bias gradient, unless a hidden-state cotangent is
        # still to be added to dx below (then the separate column-sum pass runs on the final dx)
        fuse = i > 0 and dhs[i] is None
        dx_sum = f32(D, f"{P}encoder.layers.{i - 1}.mlp.fc2.bias") if fuse else None
        dx = K.layernorm_bwd(dxn1, s["x"], params[pre + "layer_norm1.weight"], s["m1"], s["r1"], dres=dx_mid,
                             dgamma=dg1, dbeta=dbt1, dxsum=dx_sum, out=dx_ring[1] if dx is dx_ring[0] else dx_ring[0])
        if dhs[i] is not None:
            K.add_(dx, dhs[i].reshape(M, D).to(BF16).contiguous())
        dp.emit_new(grads, emitted)            # data parallel: this layer's weight gradients start their all-reduce now
    # ---- embeddings: hs0 = LN(emb); emb = [cls ; patches] + pos
    dg0, db0 = f32(D, P + "pre_layrnorm.weight"), f32(D, P + "pre_layrnorm.bias")
    demb = K.layernorm_bwd(dx, saved["emb"], params[P + "pre_layrnorm.weight"], saved["mean0"], saved["rstd0"],
                           dgamma=dg0, dbeta=db0)
    small_bf16 = K.f32_to_bf16(arena)
    for name, o, n in small:
        grads[name] = small_bf16[o:o + n]
    demb3 = demb.view(B, T, D)
    # tiny batch reductions (B x 577 x 1024 -> 577 x 1024): plumbing, not the hot path
    dpos = demb3.float().sum(0)
    grads[P + "embeddings.position_embedding.weight"] = dpos.to(BF16)
    grads[P + "embeddings.class_embedding"] = demb3[:, 0].float().sum(0).to(BF16)
    npatch = B * (T - 1)
    np_pad = K.round_up(npatch, 64)
    dpatch = K.alloc_rows(npatch, D, dev)[:npatch]
    dpatch.view(B, T - 1, D).copy_(demb3[:, 1:])
    cols = saved["cols"]
    if cols.shape[0] != np_pad:                     # forward allocates it padded (see patch_im2col); be safe
        cpad = K.alloc_rows(npatch, cols.shape[1], dev)
        cpad[:npatch].copy_(cols)
        cols = cpad[:npatch]
    dwpe = K.gemm_nt(_full(dpatch, np_pad), _full(cols, np_pad), a_t=True, b_t=True)     # [D, kpe]
    kk = dims.channels * dims.patch * dims.patch
    grads[P + "embeddings.patch_embedding.weight"] = dwpe[:, :kk].reshape(D, dims.channels, dims.patch, dims.patch)
    dpixel = None
    if need_pixel_grad:
        dcols = K.gemm_nt(dpatch, packed["wpe"], b_t=True)                               # [Np, kpe]
        dpixel = K.patch_col2im(dcols, B, dims.channels, dims.image, dims.image, dims.patch)
    dp.emit_new(grads, emitted)                # the bias / LayerNorm arena and the embedding gradients
    return dpixel, grads
