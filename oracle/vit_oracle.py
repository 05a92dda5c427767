"""CPU oracle for the CLIP-ViT encoder half of the Libra hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``libra_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

This is a clean-room, functional restatement (plain torch ops, dtype-agnostic:
run it in float64 / float32 for the accumulate-exact oracle, or in bfloat16 to
reproduce the reference's op-by-op rounding points) of

  * CLIPVisionEmbeddings.forward   /root/reference/libra/models/clip/modeling_clip.py:193-228
  * nn.LayerNorm (eps 1e-5)        modeling_clip.py:386-388, :866
  * CLIPAttention.forward          modeling_clip.py:287-363
  * CLIPMLP.forward (quick_gelu)   modeling_clip.py:374-378
  * CLIPEncoderLayer.forward       modeling_clip.py:390-428
  * CLIPEncoder.forward            modeling_clip.py:615-700  (collects all hidden states)
  * CLIPVisionTransformer.forward  modeling_clip.py:872-914  (embeddings -> pre_layrnorm -> encoder)

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here
against fixtures produced by running the reference's own modules in the build
container (``tests/golden/make_golden.py``).

Weights are addressed by the reference's state-dict key names
(``vision_model.encoder.layers.{i}.self_attn.q_proj.weight`` ...), SURVEY §8(b).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

P = "vision_model."


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    # HF ACT2FN["quick_gelu"]: x * sigmoid(1.702 x)   (config.hidden_act, modeling_clip.py:370)
    return x * torch.sigmoid(1.702 * x)


def patch_embed(sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, patch: int) -> torch.Tensor:
    """modeling_clip.py:193-228. Conv2d(k=s=patch, no bias) == im2col + GEMM."""
    w = sd[P + "embeddings.patch_embedding.weight"]  # [D,3,p,p]
    B, C, H, W = pixel_values.shape
    gh, gw = H // patch, W // patch
    x = pixel_values.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    x = x.reshape(B, gh * gw, C * patch * patch)            # im2col rows, (c,ky,kx) fastest
    pe = x @ w.reshape(w.shape[0], -1).t()                  # [B, gh*gw, D]
    cls = sd[P + "embeddings.class_embedding"].expand(B, 1, -1)
    emb = torch.cat([cls, pe], dim=1)
    return emb + sd[P + "embeddings.position_embedding.weight"][None, : emb.shape[1]]


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(sd, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """modeling_clip.py:287-363 (no masks on the vision tower, dropout p=0)."""
    B, N, D = x.shape
    hd = D // heads
    scale = hd ** -0.5
    q = F.linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]) * scale   # :299
    k = F.linear(x, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])
    v = F.linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                                  # :308
    p = torch.softmax(s, dim=-1)                                                 # :335 (input dtype)
    o = (p @ v).transpose(1, 2).reshape(B, N, D)                                 # :348-358
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def mlp(sd, pre: str, x: torch.Tensor) -> torch.Tensor:
    """modeling_clip.py:374-378."""
    h = F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    return F.linear(quick_gelu(h), sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def encoder_layer(sd, i: int, x: torch.Tensor, heads: int, eps: float) -> torch.Tensor:
    """modeling_clip.py:390-428 — pre-norm residual block."""
    pre = f"{P}encoder.layers.{i}."
    h = layer_norm(x, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"], eps)
    x = x + attention(sd, pre + "self_attn.", h, heads)
    h = layer_norm(x, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"], eps)
    return x + mlp(sd, pre + "mlp.", h)


def vit_hidden_states(sd, pixel_values: torch.Tensor, *, patch: int, heads: int, layers: int,
                      eps: float = 1e-5) -> List[torch.Tensor]:
    """All ``layers+1`` hidden states exactly as CLIPEncoder collects them
    (modeling_clip.py:664-665, :693-694): hs[0] is the pre_layrnorm output."""
    x = patch_embed(sd, pixel_values, patch)
    x = layer_norm(x, sd[P + "pre_layrnorm.weight"], sd[P + "pre_layrnorm.bias"], eps)   # :893 (sic)
    hs = [x]
    for i in range(layers):
        x = encoder_layer(sd, i, x, heads, eps)
        hs.append(x)
    return hs


def feature_select(hs: Sequence[torch.Tensor], select_layer, square: bool = True) -> torch.Tensor:
    """CLIPVisionTower.feature_select + reshape_to_square
    (/root/reference/libra/models/libra/clip_encoder.py:31-51): channel-concat of the
    selected hidden states, CLS dropped, -> [B, C, g, g]."""
    if isinstance(select_layer, (list, tuple)):
        f = torch.cat([hs[i] for i in select_layer], dim=-1)
    else:
        f = hs[select_layer]
    f = f[:, 1:]
    if not square:
        return f
    B, N, C = f.shape
    g = int(math.isqrt(N))
    assert g * g == N
    return f.view(B, g, g, C).permute(0, 3, 1, 2)


def cast_sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def random_vit_state_dict(*, hidden: int, inter: int, layers: int, patch: int, image: int,
                          seed: int = 42, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random-init ViT weights with CLIP's initialiser shapes/scales
    (modeling_clip.py:442-493: factor 1.0).  Deterministic across machines for a
    given torch version; used by the GPU parity tests and by bench.py so the GPU
    path and the CPU baseline see identical weights."""
    g = torch.Generator().manual_seed(seed)
    n_pos = (image // patch) ** 2 + 1
    sd = {}

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    sd[P + "embeddings.class_embedding"] = rn(hidden, std=hidden ** -0.5)
    sd[P + "embeddings.patch_embedding.weight"] = rn(hidden, 3, patch, patch, std=0.02)
    sd[P + "embeddings.position_embedding.weight"] = rn(n_pos, hidden, std=0.02)
    for nm in ("pre_layrnorm", "post_layernorm"):
        sd[P + nm + ".weight"] = (1.0 + rn(hidden, std=0.1)).to(dtype)
        sd[P + nm + ".bias"] = rn(hidden, std=0.1)
    in_std = hidden ** -0.5 * (2 * layers) ** -0.5
    out_std = hidden ** -0.5
    fc_std = (2 * hidden) ** -0.5
    for i in range(layers):
        pre = f"{P}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            sd[pre + f"self_attn.{nm}.weight"] = rn(hidden, hidden, std=in_std * 4)
            sd[pre + f"self_attn.{nm}.bias"] = rn(hidden, std=0.05)
        sd[pre + "self_attn.out_proj.weight"] = rn(hidden, hidden, std=out_std)
        sd[pre + "self_attn.out_proj.bias"] = rn(hidden, std=0.05)
        sd[pre + "mlp.fc1.weight"] = rn(inter, hidden, std=fc_std)
        sd[pre + "mlp.fc1.bias"] = rn(inter, std=0.05)
        sd[pre + "mlp.fc2.weight"] = rn(hidden, inter, std=in_std)
        sd[pre + "mlp.fc2.bias"] = rn(hidden, std=0.05)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[pre + nm + ".weight"] = (1.0 + rn(hidden, std=0.1)).to(dtype)
            sd[pre + nm + ".bias"] = rn(hidden, std=0.1)
    return sd
