"""CPU oracle for the VQ/LFQ encode tail of the Libra vision tokenizer.

TEST INFRASTRUCTURE ONLY (see vit_oracle.py header for the import rule).

Restates, as plain torch functions (dtype-agnostic):

  * VQModel.encode                /root/reference/libra/models/libra/taming/models/vqgan.py:106-114
      quant_conv = Conv2d(C_feat, E, 1) (bias)                        vqgan.py:74
  * LFQ.forward, eval branch      .../taming/modules/quantization/lookup_free_quantization.py:160-280
      project_in Linear(E->Q*9) iff E != Q*9 (:79-81); sign quantise (:195-196, :204);
      index = sum (x>0) * 2^[8..0]  MSB first (:111, :208); project_out (:261); aux = 0 (:234-248)
      return contract is POSITIONAL: [0]=quant [1]=aux_loss [2]=indices (:275, vqgan.py:109)
  * ImageTokenizer.encode         /root/reference/libra/models/libra/image_tokenizer.py:74-95
      ids = indices.permute(3,0,1,2)+offset, BOI/EOI framing (:44-49), encoder_feat [B,hw,C]

State-dict keys follow VQModel: ``quant_conv.{weight,bias}``,
``quantize.project_in.{weight,bias}``, ``quantize.project_out.{weight,bias}``.

The sign decision is taken on the value *after* it is rounded to the working
dtype, exactly as the reference does (``x > 0`` on the Linear's output tensor).
``margins`` (|x| before the sign) are returned so tests can report how close to
zero any disagreeing bit was.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def quant_conv(sd: Dict[str, torch.Tensor], feat_bchw: torch.Tensor) -> torch.Tensor:
    """1x1 conv == per-pixel Linear (vqgan.py:108)."""
    w = sd["quant_conv.weight"]            # [E, C, 1, 1]
    b = sd["quant_conv.bias"]
    return F.conv2d(feat_bchw, w, b)


def lfq_eval(sd: Dict[str, torch.Tensor], h_bchw: torch.Tensor, *, num_codebooks: int = 2,
             codebook_dim: int = 9) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """LFQ.forward in eval mode. Returns (quant [B,E,h,w], aux(0), indices int64 [B,h,w,Q], x_pre_sign)."""
    B, E, H, W = h_bchw.shape
    x = h_bchw.permute(0, 2, 3, 1).reshape(B, H * W, E)                      # b d h w -> b (hw) d
    cd = num_codebooks * codebook_dim
    has_proj = E != cd
    if has_proj:
        x = F.linear(x, sd["quantize.project_in.weight"], sd["quantize.project_in.bias"])
    x = x.reshape(B, H * W, num_codebooks, codebook_dim)
    pre = x
    pos = x > 0
    quantized = torch.where(pos, torch.ones_like(x), -torch.ones_like(x))    # codebook_scale = 1
    mask = 2 ** torch.arange(codebook_dim - 1, -1, -1, dtype=torch.int32)    # MSB first
    indices = (pos.to(torch.int32) * mask).sum(-1).to(torch.int64)           # einops reduce 'sum' on int32 -> int64
    q = quantized.reshape(B, H * W, cd)
    if has_proj:
        q = F.linear(q, sd["quantize.project_out.weight"], sd["quantize.project_out.bias"])
    quant = q.reshape(B, H, W, E).permute(0, 3, 1, 2)
    indices = indices.reshape(B, H, W, num_codebooks)
    aux = torch.zeros((), dtype=h_bchw.dtype)
    return quant, aux, indices, pre.reshape(B, H, W, cd)


def vq_encode(sd, feat_bchw, **kw):
    """VQModel.encode(x, return_encoder_feat=True) given the tower output.
    -> (quant, aux, indices, encoder_feat, x_pre_sign)"""
    h = quant_conv(sd, feat_bchw)
    quant, aux, idx, pre = lfq_eval(sd, h, **kw)
    return quant, aux, idx, feat_bchw, pre


def image_tokenizer_encode(indices_bhwq: torch.Tensor, feat_bchw: torch.Tensor, *, offset: int,
                           codebook_size: int = 512):
    """ImageTokenizer.encode (image_tokenizer.py:74-95) from VQ outputs.
    -> input_ids int64 [Q,B,hw+2], attention_mask [B,hw+2], encoder_feat [B,hw,C]"""
    boi = offset + codebook_size          # token_offset + len(self) - 2   (:46)
    eoi = offset + codebook_size + 1      # (:47)
    ids = indices_bhwq.permute(3, 0, 1, 2) + offset
    ids = ids.flatten(2, 3)
    Q, B, _ = ids.shape
    ids = torch.cat([torch.full((Q, B, 1), boi, dtype=ids.dtype), ids,
                     torch.full((Q, B, 1), eoi, dtype=ids.dtype)], dim=-1)
    attn = torch.ones(ids[0].shape, dtype=torch.long)
    feat = feat_bchw.flatten(2, 3).permute(0, 2, 1).contiguous()
    return ids, attn, feat


def random_vq_state_dict(*, c_feat: int, embed_dim: int, num_codebooks: int = 2, codebook_dim: int = 9,
                         seed: int = 43, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    cd = num_codebooks * codebook_dim

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    sd = {"quant_conv.weight": rn(embed_dim, c_feat, 1, 1, std=c_feat ** -0.5),
          "quant_conv.bias": rn(embed_dim, std=0.05)}
    if embed_dim != cd:
        sd["quantize.project_in.weight"] = rn(cd, embed_dim, std=embed_dim ** -0.5)
        sd["quantize.project_in.bias"] = rn(cd, std=0.05)
        sd["quantize.project_out.weight"] = rn(embed_dim, cd, std=cd ** -0.5)
        sd["quantize.project_out.bias"] = rn(embed_dim, std=0.05)
    return sd
