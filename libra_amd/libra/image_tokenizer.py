"""ImageTokenizer — mirror of /root/reference/libra/models/libra/image_tokenizer.py:12-139 (encode side).

``encode`` returns the same dict (``input_ids`` int64 [Q,B,hw+2] framed by BOI/EOI, ``image_size``,
``attention_mask``, ``encoder_feat`` [B,hw,C]); the ids are written directly in their final layout by the
fused LFQ kernel (no permute / += offset / cat passes).
"""
import math
import os

import torch

from .vqgan import VQModel


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    g = getattr(cfg, "get", None)
    return g(key, default) if g is not None else getattr(cfg, key, default)


def disabled_train(self, mode=True):
    return self


class ImageTokenizer(torch.nn.Module):
    def __init__(self, cfg, token_offset, vision_model=None, **kwargs):
        super().__init__()
        params = _get(cfg, "params")
        self.codebook_size = _get(params, "codebook_size")
        self.num_codebook = _get(params, "num_codebook")
        self.model = VQModel(ignore_keys=["loss."], vision_model=vision_model, **dict(params))
        if _get(cfg, 'ckpt_path', None) is not None:
            raise NotImplementedError("Please load image tokenizer weight through params.ckpt_path.")
        ckpt_path = _get(params, "ckpt_path") or ""
        base = os.path.basename(ckpt_path)
        if "_f16_" in base and "_f8_" in base:
            raise NotImplementedError
        self.downsample_ratio = 16 if "_f16_" in base else (8 if "_f8_" in base else None)
        if _get(cfg, "freeze", True):
            self.model.eval()
            self.model.train = disabled_train
            for _, p in self.model.named_parameters():
                p.requires_grad = False
        self.offset = token_offset
        self.boi_token_id = token_offset + len(self) - 2
        self.eoi_token_id = token_offset + len(self) - 1
        self.max_vision_token_length = _get(cfg, "max_vision_token_length")
        self.vocab_size = self.codebook_size + 2

    @property
    def device(self):
        return self.model.device

    @property
    def dtype(self):
        return self.model.dtype

    def __len__(self) -> int:
        return self.codebook_size + 2

    def get_token_length(self, images: torch.Tensor):
        if self.downsample_ratio is None:
            return self.max_vision_token_length
        _, _, H, W = images.shape
        assert H == W
        return (H // self.downsample_ratio) ** 2 + 2

    @torch.no_grad()
    def forward(self, x, add_boi_token=True, add_eoi_token=True, return_tensors=True, return_encoder_feat=True):
        return self.encode(x, add_boi_token=add_boi_token, add_eoi_token=add_eoi_token, return_tensors=return_tensors,
                           return_encoder_feat=return_encoder_feat)

    @torch.no_grad()
    def encode(self, x, add_boi_token=True, add_eoi_token=True, return_tensors=True, return_encoder_feat=True):
        assert return_tensors
        feat, _, _, ids, _, _ = self.model.encode_flat(x, offset=self.offset, boi=self.boi_token_id,
                                                       eoi=self.eoi_token_id, want_ids=True, want_quant=False)
        lo = 0 if add_boi_token else 1
        hi = ids.shape[-1] if add_eoi_token else ids.shape[-1] - 1
        input_ids = ids if (lo == 0 and hi == ids.shape[-1]) else ids[..., lo:hi].contiguous()
        B, hw, _ = feat.shape
        g = int(math.isqrt(hw))
        attention_mask = torch.ones(input_ids[0].shape, dtype=torch.long, device=feat.device)
        return {"input_ids": input_ids, "image_size": [g, g], "attention_mask": attention_mask, "encoder_feat": feat}

    @torch.no_grad()
    def decode(self, x):
        """token ids [Q, B, N (+2 with BOI / EOI)] (or [B, N] for one codebook) -> image [B, out_ch, H, W]
        (image_tokenizer.py:97-124): strip the frame tokens, undo the text-vocabulary offset, square the sequence, VQ-decode."""
        if len(x) == 0 or len(x[0]) == 0:
            return x
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, dtype=torch.long, device=self.device)
        if x.dim() == 2:
            x = x[None, ...]
        elif x.dim() != 3:
            raise NotImplementedError
        if bool((x == self.boi_token_id).any()):
            x = x[:, :, 1:-1]                                    # exclude <img> and <\img> tokens
        Q, B, N = x.shape
        side = math.isqrt(N)
        if side * side != N:
            raise ValueError('Input images are invalid. Currently, the image decoder only support square images.')
        idx = (x.reshape(Q, B, side, side).permute(1, 2, 3, 0) - self.offset).contiguous()
        return self.model.decode_code(idx)

    @classmethod
    def from_config(cls, config, **kwargs):
        return cls(config, **kwargs)
