"""CLIPVisionTower — mirror of /root/reference/libra/models/libra/clip_encoder.py:8-96 on the gfx950 kernels.

Same constructor (vision_tower, delay_load, square_output, select_layer), ``feature_select``,
``reshape_to_square``, ``forward`` and properties.  ``forward`` runs under ``no_grad`` exactly like the
reference (:53); ``forward_flat`` is the fused entry the VQ path uses (one feature-select kernel, output
already in the [B, hw, C] layout ImageTokenizer returns) and can run under autograd (``allow_grad=True``
— an extension beyond the reference).
"""
import math
from collections.abc import Iterable

import torch
import torch.nn as nn

from .. import kernels as K
from ..clip import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel


class _SelectFn(torch.autograd.Function):
    """cat([hs[i] for i in select], -1)[:, 1:]  as one gather kernel (+ its scatter backward)."""

    @staticmethod
    def forward(ctx, B, T, *hs):
        ctx.B, ctx.T, ctx.D, ctx.n = B, T, hs[0].shape[-1], len(hs)
        flat = [h.reshape(B * T, -1).contiguous() for h in hs]
        return K.feature_select(flat, B, T)

    @staticmethod
    def backward(ctx, dfeat):
        B, T, D, n = ctx.B, ctx.T, ctx.D, ctx.n
        dhs = [torch.empty((B, T, D), dtype=torch.bfloat16, device=dfeat.device) for _ in range(n)]
        K.feature_select_bwd(dfeat.contiguous(), dhs, [False] * n, B, T)
        return (None, None, *dhs)


class CLIPVisionTower(nn.Module):
    def __init__(self, vision_tower, delay_load=False, square_output=False, select_layer=-2, model=None):
        super().__init__()
        self.square_output = square_output
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = select_layer
        self.select_feature = 'patch'
        self.allow_grad = False
        if model is not None:                      # test / bench hook: adopt an already-built CLIPVisionModel
            self.image_processor = None
            self.vision_tower = model
            self.vision_tower.requires_grad_(False)
            self.is_loaded = True
        elif not delay_load:
            self.load_model()
        else:
            self.cfg_only = CLIPVisionConfig.from_pretrained(self.vision_tower_name)

    def load_model(self):
        self.image_processor = CLIPImageProcessor.from_pretrained(self.vision_tower_name)
        self.vision_tower = CLIPVisionModel.from_pretrained(self.vision_tower_name)
        self.vision_tower.requires_grad_(False)     # clip_encoder.py:27
        self.is_loaded = True

    def _select_list(self):
        return list(self.select_layer) if isinstance(self.select_layer, Iterable) else [self.select_layer]

    def feature_select(self, image_forward_outs):
        hs = image_forward_outs.hidden_states
        if self.select_feature not in ('patch', 'cls_patch'):
            raise ValueError(f'Unexpected select feature: {self.select_feature}')
        sel = [hs[i] for i in self._select_list()]
        B, T, _ = sel[0].shape
        if self.select_feature == 'patch':
            return _SelectFn.apply(B, T, *sel).view(B, T - 1, -1)
        return torch.cat(sel, dim=-1)

    def reshape_to_square(self, feats):
        """[B, hw, C] -> [B, C, h, w] for a square patch grid (the reference's helper of the same name, clip_encoder.py:47-51)."""
        batch, tokens, channels = feats.shape
        side = math.isqrt(tokens)
        if side * side != tokens:
            raise AssertionError(f"{tokens} patch tokens do not form a square grid")
        return feats.view(batch, side, side, channels).permute(0, 3, 1, 2)

    def forward_flat(self, images):
        """-> [B, hw, C_total] contiguous (the ``encoder_feat`` layout of image_tokenizer.py:93)."""
        with torch.set_grad_enabled(self.allow_grad and torch.is_grad_enabled()):
            outs = self.vision_tower(images.to(device=self.device, dtype=self.dtype), output_hidden_states=True)
            return self.feature_select(outs).to(self.dtype)

    @torch.no_grad()
    def forward(self, images, square_output=None):
        square_output = self.square_output if square_output is None else square_output
        if type(images) is list:
            feats = []
            for image in images:
                f = self.forward_flat(image.unsqueeze(0))
                feats.append(self.reshape_to_square(f) if square_output else f)
            return feats
        f = self.forward_flat(images)
        return self.reshape_to_square(f) if square_output else f

    # ---- read-only surface of the reference class (clip_encoder.py:72-96): everything is answered by the wrapped tower ----
    dtype = property(lambda self: self.vision_tower.dtype)
    device = property(lambda self: self.vision_tower.device)
    hidden_size = property(lambda self: self.config.hidden_size)
    num_patches = property(lambda self: (self.config.image_size // self.config.patch_size) ** 2)

    @property
    def config(self):
        """The tower's config once it is loaded, before that the one read from `vision_tower` at construction (delay_load)."""
        if self.is_loaded:
            return self.vision_tower.config
        return self.cfg_only

    @property
    def dummy_feature(self):
        return torch.zeros((1, self.hidden_size), dtype=self.dtype, device=self.device)
