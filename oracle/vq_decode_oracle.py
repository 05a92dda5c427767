"""CPU oracle for the VQ image decoder: token ids -> LFQ codes -> post_quant_conv -> taming Decoder -> image.

TEST INFRASTRUCTURE ONLY (see vit_oracle.py header for the import rule); no product code exists for this row yet.

Clean-room functional restatement (plain torch ops on the reference's state-dict keys), following
  * ImageTokenizer.decode            /root/reference/libra/models/libra/image_tokenizer.py:97-124
  * VQModel.decode / decode_code     /root/reference/libra/models/libra/taming/models/vqgan.py:122-130
  * LFQ.indices_to_codes             /root/reference/libra/models/libra/taming/modules/quantization/lookup_free_quantization.py:129-158
    (bits MSB-first per codebook -> +-1 codes, `project_out` iff embed_dim != num_codebooks * log2(codebook_size))
  * Decoder.forward, ResnetBlock, AttnBlock (single head), Upsample, Normalize = GroupNorm(32, eps 1e-6), swish
                                     /root/reference/libra/models/libra/taming/modules/diffusionmodules/model.py:28-232, :474-588
Pinned by tests/test_oracle_golden.py against tests/golden/vq_decode_tiny.safetensors (tests/golden/make_golden_vq_decode.py
runs the reference's own classes).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def swish(x):
    return x * torch.sigmoid(x)                                            # model.py:28-31


def group_norm(sd, pre, x):
    return F.group_norm(x, 32, sd[pre + "weight"], sd[pre + "bias"], eps=1e-6)     # Normalize, :34-35


def conv(sd, pre, x, padding=0):
    return F.conv2d(x, sd[pre + "weight"], sd.get(pre + "bias"), stride=1, padding=padding)


def indices_to_codes(sd, indices: torch.Tensor, codebook_size: int, dtype=torch.float32) -> torch.Tensor:
    """indices [B,h,w,Q] in [0, codebook_size) -> codes [B,E,h,w]  (:129-158)."""
    nbits = int(math.log2(codebook_size))
    mask = 2 ** torch.arange(nbits - 1, -1, -1)                            # MSB first (:111)
    bits = ((indices[..., None].int() & mask) != 0).to(dtype)              # [B,h,w,Q,nbits]
    codes = (bits * 2 - 1).flatten(-2)                                     # bits_to_codes, '... c d -> ... (c d)'
    if "quantize.project_out.weight" in sd:
        codes = F.linear(codes, sd["quantize.project_out.weight"], sd.get("quantize.project_out.bias"))
    return codes.permute(0, 3, 1, 2).contiguous()                          # 'b ... d -> b d ...'


def resnet_block(sd, pre, x):
    h = conv(sd, pre + "conv1.", swish(group_norm(sd, pre + "norm1.", x)), padding=1)
    h = conv(sd, pre + "conv2.", swish(group_norm(sd, pre + "norm2.", h)), padding=1)        # dropout p = 0
    if pre + "nin_shortcut.weight" in sd:
        x = conv(sd, pre + "nin_shortcut.", x)
    elif pre + "conv_shortcut.weight" in sd:
        x = conv(sd, pre + "conv_shortcut.", x, padding=1)
    return x + h                                                           # :116-138


def attn_block(sd, pre, x):
    """Single-head spatial self-attention over the h*w positions (:170-196)."""
    hn = group_norm(sd, pre + "norm.", x)
    q, k, v = conv(sd, pre + "q.", hn), conv(sd, pre + "k.", hn), conv(sd, pre + "v.", hn)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    p = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)           # [b, hw(query), hw(key)]
    o = torch.bmm(v.reshape(b, c, h * w), p.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + conv(sd, pre + "proj_out.", o)


def decoder(sd, z, *, ch_mult: Sequence[int], num_res_blocks: int, resolution: int, pre: str = "decoder."):
    """Decoder.forward (:554-588).  Which levels carry attention / an upsample conv is read off the state dict."""
    h = conv(sd, pre + "conv_in.", z, padding=1)
    h = resnet_block(sd, pre + "mid.block_1.", h)
    h = attn_block(sd, pre + "mid.attn_1.", h)
    h = resnet_block(sd, pre + "mid.block_2.", h)
    nres = len(ch_mult)
    curr = resolution // 2 ** (nres - 1)
    for lvl in reversed(range(nres)):
        for blk in range(num_res_blocks + 1):
            h = resnet_block(sd, f"{pre}up.{lvl}.block.{blk}.", h)
            if f"{pre}up.{lvl}.attn.{blk}.norm.weight" in sd:
                h = attn_block(sd, f"{pre}up.{lvl}.attn.{blk}.", h)
        if lvl != 0:
            # :531-536: levels > 1 double; level 1 jumps to the output resolution
            scale = 2.0 if lvl > 1 else resolution / curr
            if lvl > 1:
                curr *= 2
            h = F.interpolate(h, scale_factor=scale, mode="nearest")
            if f"{pre}up.{lvl}.upsample.conv.weight" in sd:
                h = conv(sd, f"{pre}up.{lvl}.upsample.conv.", h, padding=1)
    return conv(sd, pre + "conv_out.", swish(group_norm(sd, pre + "norm_out.", h)), padding=1)


def decode_code(sd, indices, *, codebook_size: int, ch_mult, num_res_blocks: int, resolution: int):
    """VQModel.decode_code (:127-130) -> (codes, z, image)."""
    codes = indices_to_codes(sd, indices, codebook_size, sd["post_quant_conv.weight"].dtype)
    z = conv(sd, "post_quant_conv.", codes)
    return codes, z, decoder(sd, z, ch_mult=ch_mult, num_res_blocks=num_res_blocks, resolution=resolution)


def token_ids_to_indices(ids: torch.Tensor, *, offset: int, boi_token_id: int) -> torch.Tensor:
    """ImageTokenizer.decode's id handling (:101-121): [Q,B,N(+2)] token ids -> [B,h,w,Q] code indices (square images only)."""
    if ids.dim() == 2:
        ids = ids[None]
    if bool((ids == boi_token_id).any()):
        ids = ids[:, :, 1:-1]
    Q, B, N = ids.shape
    side = math.isqrt(N)
    if side * side != N:
        raise ValueError("Input images are invalid. Currently, the image decoder only support square images.")
    return ids.reshape(Q, B, side, side).permute(1, 2, 3, 0) - offset
