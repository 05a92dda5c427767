"""LFQ (eval branch) — mirror of
/root/reference/libra/models/libra/taming/modules/quantization/lookup_free_quantization.py:51-280.

Same constructor arguments, parameters (``project_in`` / ``project_out`` Linear when dim != Q*9), buffers
(``mask``, ``zero``, non-persistent ``codebook``) and POSITIONAL return contract
``(quantized, aux_loss, indices)`` (:275 — the namedtuple field names are misleading upstream; every
caller unpacks positionally).  Libra freezes the tokenizer in eval mode (image_tokenizer.py:37-42), so
only the eval branch exists here; it is one fused gfx950 kernel (csrc/vq.hip).
"""
from collections import namedtuple
from math import ceil, log2

import torch
from torch import nn

from .. import kernels as K

Return = namedtuple('Return', ['quantized', 'indices', 'entropy_aux_loss'])   # sic, see module docstring


class LFQ(nn.Module):
    def __init__(self, *, dim=None, codebook_size=None, entropy_loss_weight=0.1, commitment_loss_weight=0.25,
                 diversity_gamma=1., straight_through_activation=nn.Identity(), num_codebooks=1,
                 keep_num_codebooks_dim=None, codebook_scale=1.):
        super().__init__()
        assert dim is not None or codebook_size is not None, 'either dim or codebook_size must be specified for LFQ'
        assert codebook_size is None or log2(codebook_size).is_integer(), (
            f'your codebook size must be a power of 2 for lookup free quantization (suggested {2 ** ceil(log2(codebook_size))})')
        codebook_size = codebook_size if codebook_size is not None else 2 ** dim
        codebook_dim = int(log2(codebook_size))
        codebook_dims = codebook_dim * num_codebooks
        dim = dim if dim is not None else codebook_dims
        has_projections = dim != codebook_dims
        self.project_in = nn.Linear(dim, codebook_dims) if has_projections else nn.Identity()
        self.project_out = nn.Linear(codebook_dims, dim) if has_projections else nn.Identity()
        self.has_projections = has_projections
        self.dim, self.codebook_dim, self.num_codebooks = dim, codebook_dim, num_codebooks
        keep_num_codebooks_dim = keep_num_codebooks_dim if keep_num_codebooks_dim is not None else num_codebooks > 1
        assert not (num_codebooks > 1 and not keep_num_codebooks_dim)
        self.keep_num_codebooks_dim = keep_num_codebooks_dim
        self.activation = straight_through_activation
        self.diversity_gamma, self.entropy_loss_weight = diversity_gamma, entropy_loss_weight
        self.codebook_scale = codebook_scale
        self.commitment_loss_weight = commitment_loss_weight
        if codebook_dim != 9 or codebook_scale != 1.:
            raise NotImplementedError("the fused kernel packs 9-bit codes with scale 1 (Libra: codebook_size=512)")
        self.register_buffer('mask', 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self.register_buffer('zero', torch.tensor(0.), persistent=False)
        all_codes = torch.arange(codebook_size)
        bits = ((all_codes[..., None].int() & self.mask) != 0).float()
        self.register_buffer('codebook', bits * codebook_scale * 2 - codebook_scale, persistent=False)

    @property
    def dtype(self):
        return self.codebook.dtype

    def encode_flat(self, h2d: torch.Tensor, B: int, hw: int, *, offset: int = 0, boi: int = 0, eoi: int = 0,
                    want_ids=False, want_quant=True, want_xpre=False):
        """h2d [B*hw, dim] bf16 -> (indices int64 [B*hw, Q], ids [Q,B,hw+2] | None, xpre | None, quant2d | None)."""
        if self.training:
            raise NotImplementedError("LFQ training branch (entropy/commitment losses) is outside the Libra hot path; "
                                      "the tokenizer is frozen in eval mode (image_tokenizer.py:37-42)")
        pi, po = self.project_in, self.project_out
        w_in = pi.weight if self.has_projections else None
        b_in = pi.bias if self.has_projections else None
        w_out = po.weight if self.has_projections else None
        b_out = po.bias if self.has_projections else None
        return K.lfq_encode(h2d, w_in, b_in, w_out, b_out, B=B, hw=hw, Q=self.num_codebooks, offset=offset, boi=boi,
                            eoi=eoi, want_ids=want_ids, want_xpre=want_xpre, want_quant=want_quant)

    @torch.no_grad()
    def indices_to_codes_flat(self, indices2d: torch.Tensor) -> torch.Tensor:
        """indices int64 [M, Q] -> codes [M, dim] bf16 (NHWC rows): +-1 per bit, MSB first, then `project_out` when the
        tokenizer has projections (lookup_free_quantization.py:129-158).  The +-1 matrix is zero-padded to the GEMM K granule."""
        kp = K.round_up(self.num_codebooks * self.codebook_dim, 64)
        codes = K.lfq_codes(indices2d.contiguous(), self.codebook_dim, kp)
        if not self.has_projections:
            return codes[:, :self.dim]
        po = self.project_out
        key = (po.weight.data_ptr(), po.weight._version)
        if getattr(self, "_po_key", None) != key:
            w = torch.zeros((self.dim, kp), dtype=torch.bfloat16, device=po.weight.device)
            w[:, :po.weight.shape[1]] = po.weight.detach()
            self._po_key, self._po_w = key, w
        return K.gemm_nt(codes, self._po_w, bias=po.bias.detach().to(torch.bfloat16))

    @torch.no_grad()
    def indices_to_codes(self, indices, project_out=True):
        """[B, h, w, Q] -> codes [B, dim, h, w]  ('b ... d -> b d ...', :129-158)."""
        if not project_out:
            raise NotImplementedError("indices_to_codes(project_out=False)")
        if not self.keep_num_codebooks_dim:
            indices = indices[..., None]
        B, H, W, Q = indices.shape
        c = self.indices_to_codes_flat(indices.reshape(B * H * W, Q))
        return c.reshape(B, H, W, -1).permute(0, 3, 1, 2)

    @torch.no_grad()
    def forward(self, x, inv_temperature=100., return_loss_breakdown=False, mask=None):
        """x [B, dim, h, w] -> Return(quant [B,dim,h,w], aux_loss (0), indices int64 [B,h,w,Q])  (positional!)."""
        if x.ndim != 4:
            raise NotImplementedError("image-shaped input [B, dim, h, w] expected on this path")
        B, E, H, W = x.shape
        assert E == self.dim, f'expected dimension of {self.dim} but received {E}'
        h2d = x.permute(0, 2, 3, 1).reshape(B * H * W, E).to(torch.bfloat16).contiguous()
        idx, _, _, q2d = self.encode_flat(h2d, B, H * W)
        quant = q2d.view(B, H, W, E).permute(0, 3, 1, 2)
        indices = idx.view(B, H, W, self.num_codebooks)
        if not self.keep_num_codebooks_dim:
            indices = indices[..., 0]
        aux = self.zero * self.entropy_loss_weight + self.zero * self.commitment_loss_weight
        ret = Return(quant, aux, indices)
        return ret if not return_loss_breakdown else (ret, None)
