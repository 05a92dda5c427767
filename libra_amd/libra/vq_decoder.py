"""VQ image decoder on the gfx950 kernels: token ids -> LFQ codes -> post_quant_conv -> taming Decoder -> image.

Module / state-dict surface of the reference's `Decoder` (conv_in, mid.{block_1, attn_1, block_2}, up.{i}.{block, attn, upsample},
norm_out, conv_out; /root/reference/libra/models/libra/taming/modules/diffusionmodules/model.py:474-588, ResnetBlock :79-138,
AttnBlock :141-230, Upsample :38-56, Normalize = GroupNorm(32, eps 1e-6) :34-35, swish :28-31).  The sub-modules only own the
parameters; `forward` is one kernel schedule over NHWC activations ([pixels, channels] bf16):

  * 1x1 convs and the (im2col-gathered) 3x3 convs are bf16 MFMA GEMMs (`libra_gemm_bf16_nt`, bias and residual add fused);
  * what precedes a conv upstream - GroupNorm + swish, the nearest-neighbour upsample - is applied inside the gather
    (`libra_conv_gather`), with the reference's bf16 rounding points; GroupNorm costs one statistics pass (`libra_groupnorm_affine`);
  * the single-head spatial attention is two GEMMs per image around a row softmax (`libra_softmax_rows`), the score matrix
    in bf16 exactly where the reference's `torch.bmm` rounds it.
The 3x3 operand is materialised (9x the activation bytes): HBM-bound at the top resolution, fine for an inference-only path.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import kernels as K

BF16 = torch.bfloat16


def Normalize(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only owns parameters in libra_amd: the decoder runs as one fused schedule - "
                           "call VQModel.decode / decode_code (or Decoder.forward)")


class ResnetBlock(_Holder):                                   # model.py:79-138 (temb_channels = 0, dropout 0)
    def __init__(self, in_channels, out_channels=None, conv_shortcut=False):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class AttnBlock(_Holder):                                     # model.py:141-230
    def __init__(self, in_channels, num_attn_head=1):
        super().__init__()
        assert in_channels % num_attn_head == 0
        self.in_channels, self.num_head = in_channels, num_attn_head
        self.norm = Normalize(in_channels)
        for n in ("q", "k", "v", "proj_out"):
            setattr(self, n, nn.Conv2d(in_channels, in_channels, 1, 1, 0))


class Upsample(_Holder):                                      # model.py:38-56
    def __init__(self, in_channels, with_conv, scale_factor=None):
        super().__init__()
        self.with_conv = with_conv
        self.scale_factor = 2.0 if scale_factor is None else scale_factor
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)


class _Act:
    """An NHWC activation [B*H*W, C] plus a pending nearest-neighbour upsample that the next gather applies."""

    def __init__(self, t, B, H, W, scale=1.0):
        self.t, self.B, self.H, self.W, self.scale = t, B, H, W, scale

    @property
    def out_hw(self):
        # F.interpolate(scale_factor=s): output size = floor(input * s)
        return (int(math.floor(self.H * self.scale)), int(math.floor(self.W * self.scale))) if self.scale != 1.0 else (self.H, self.W)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, initial_resolution=None, num_attn_head=1, norm_first=False,
                 **ignorekwargs):
        super().__init__()
        if dropout:
            raise NotImplementedError("dropout in the VQ decoder (0 in the released tokenizer)")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.give_pre_end, self.norm_first = resolution, in_channels, give_pre_end, norm_first
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = initial_resolution if initial_resolution is not None else resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        if norm_first:
            self.first_norm = Normalize(z_channels)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in, num_attn_head)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in, num_attn_head))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level > 1:                                   # :531-536: levels > 1 double, level 1 jumps to the output resolution
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            elif i_level == 1:
                up.upsample = Upsample(block_in, resamp_with_conv, scale_factor=resolution / curr_res)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
        self._packed = {}

    # ---- packed weights: [Cout, taps * Cin] tap-major, K zero-padded to the GEMM granule (64); rebuilt when a tensor changes ----
    def _w(self, conv: nn.Conv2d, key: str, n_pad: int = 0):
        w = conv.weight
        k = (w.data_ptr(), w._version, n_pad)
        hit = self._packed.get(key)
        if hit is None or hit[0] != k:
            co, ci, kh, kw = w.shape
            kp = K.round_up(kh * kw * ci, 64)
            wp = torch.zeros((max(co, n_pad), kp), dtype=BF16, device=w.device)
            wp[:co, :kh * kw * ci] = w.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
            b = torch.zeros((max(co, n_pad),), dtype=BF16, device=w.device)
            if conv.bias is not None:
                b[:co] = conv.bias.detach()
            hit = (k, wp, b)
            self._packed[key] = hit
        return hit[1], hit[2]

    def _gn(self, a: _Act, norm: nn.GroupNorm):
        if a.scale != 1.0:
            raise RuntimeError("GroupNorm directly after an upsample without conv is not a configuration of the reference decoder")
        return K.groupnorm_affine(a.t, norm.weight, norm.bias, a.B, a.H * a.W, norm.num_groups, norm.eps)

    def _conv(self, a: _Act, conv: nn.Conv2d, key: str, *, norm: Optional[nn.GroupNorm] = None, swish=False, resid=None) -> _Act:
        """[GroupNorm (+ swish)] -> [pending upsample] -> conv (+ bias, + residual): gather + one GEMM."""
        sc = sh = None
        if norm is not None:
            sc, sh = self._gn(a, norm)
        wp, b = self._w(conv, key)
        ks = conv.kernel_size[0]
        H, W = a.out_hw
        if ks == 1 and norm is None and a.scale == 1.0 and a.t.shape[1] == wp.shape[1]:
            op = a.t                                         # a 1x1 conv on a K-aligned activation needs no gather
        else:
            op = K.conv_gather(a.t, a.B, a.H, a.W, H, W, ks, wp.shape[1], scale=sc, shift=sh, swish=swish, inv_scale=1.0 / a.scale)
        co = conv.out_channels
        out = torch.empty((a.B * H * W, K.round_up(co, 8)), dtype=BF16, device=a.t.device)
        K.gemm_nt(op, wp, out=out[:, :co], bias=b, resid=resid)
        if co % 8:
            out[:, co:].zero_()
        return _Act(out if co % 8 == 0 else out, a.B, H, W)

    def _resnet(self, a: _Act, blk: ResnetBlock, key: str) -> _Act:
        h = self._conv(a, blk.conv1, key + ".conv1", norm=blk.norm1, swish=True)
        if blk.in_channels != blk.out_channels:
            skip = self._conv(a, blk.conv_shortcut if blk.use_conv_shortcut else blk.nin_shortcut, key + ".shortcut").t
        else:
            skip = a.t
        return self._conv(h, blk.conv2, key + ".conv2", norm=blk.norm2, swish=True, resid=skip)

    def _attn(self, a: _Act, blk: AttnBlock, key: str) -> _Act:
        B, HW, C, nh = a.B, a.H * a.W, blk.in_channels, blk.num_head
        ch = C // nh
        chp = K.round_up(ch, 64)                             # per-head q / k width padded to the GEMM K granule with zero columns
        hwp = K.round_up(HW, 64)
        dev = a.t.device
        # fused [q | k | v] 1x1 conv on GroupNorm(x): q, k head blocks padded with zero weight rows
        pk = self._packed.get(key)
        kk = tuple((getattr(blk, n).weight.data_ptr(), getattr(blk, n).weight._version) for n in "qkv")
        if pk is None or pk[0] != kk:
            kp = K.round_up(C, 64)
            w = torch.zeros((2 * nh * chp + C, kp), dtype=BF16, device=dev)
            bias = torch.zeros((2 * nh * chp + C,), dtype=BF16, device=dev)
            for i, n in enumerate("qk"):
                cw, cb = getattr(blk, n).weight.detach().view(C, C), getattr(blk, n).bias.detach()
                for h in range(nh):
                    w[(i * nh + h) * chp:(i * nh + h) * chp + ch, :C] = cw[h * ch:(h + 1) * ch]
                    bias[(i * nh + h) * chp:(i * nh + h) * chp + ch] = cb[h * ch:(h + 1) * ch]
            w[2 * nh * chp:, :C] = blk.v.weight.detach().view(C, C)
            bias[2 * nh * chp:] = blk.v.bias.detach()
            pk = (kk, w, bias)
            self._packed[key] = pk
        _, wqkv, bqkv = pk
        sc, sh = self._gn(a, blk.norm)
        hn = K.conv_gather(a.t, B, a.H, a.W, a.H, a.W, 1, wqkv.shape[1], scale=sc, shift=sh, swish=False)
        M = B * HW
        qkv = torch.zeros((M + 64, wqkv.shape[0]), dtype=BF16, device=dev)      # zero pad rows: V tail of the last image's K tile
        K.gemm_nt(hn, wqkv, out=qkv[:M], bias=bqkv)
        o = torch.zeros((M, K.round_up(C, 64)), dtype=BF16, device=dev)
        voff = 2 * nh * chp
        s = torch.empty((HW, hwp), dtype=BF16, device=dev)
        for b in range(B):
            r0 = b * HW
            for h in range(nh):
                q = qkv[r0:r0 + HW, h * chp:(h + 1) * chp]
                k = qkv[r0:r0 + HW, (nh + h) * chp:(nh + h + 1) * chp]
                K.gemm_nt(q, k, out=s[:, :HW])                                        # w_[i, j] = q_i . k_j   (bf16, like torch.bmm)
                K.softmax_rows_(s, HW, float(int(ch) ** -0.5))                       # * c^-0.5, softmax over keys; pad columns -> 0
                v = qkv[r0:r0 + hwp, voff + h * ch:voff + (h + 1) * ch]               # [keys (+ finite pad rows), ch]
                K.gemm_nt(s, v, b_t=True, out=o[r0:r0 + HW, h * ch:(h + 1) * ch])     # h_[i, :] = sum_j w_[i, j] v_j
        wp, bp = self._w(blk.proj_out, key + ".proj")
        out = torch.empty((M, C), dtype=BF16, device=dev)
        K.gemm_nt(o, wp, out=out, bias=bp, resid=a.t)
        return _Act(out, B, a.H, a.W)

    @torch.no_grad()
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, z_channels, h, w] (or an NHWC `_Act`) -> image [B, out_ch, H, W] bf16."""
        if isinstance(z, _Act):
            a = z
        else:
            if not z.is_cuda:
                raise RuntimeError("libra_amd VQ Decoder runs on MI355X only; got a CPU tensor (no CPU fallback)")
            B, C, H, W = z.shape
            a = _Act(z.to(BF16).permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous(), B, H, W)
        if self.norm_first:
            raise NotImplementedError("norm_first (off in the released tokenizer config)")
        a = self._conv(a, self.conv_in, "conv_in")
        a = self._resnet(a, self.mid.block_1, "mid.block_1")
        a = self._attn(a, self.mid.attn_1, "mid.attn_1")
        a = self._resnet(a, self.mid.block_2, "mid.block_2")
        for i_level in reversed(range(self.num_resolutions)):
            up = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                a = self._resnet(a, up.block[i_block], f"up.{i_level}.block.{i_block}")
                if len(up.attn) > 0:
                    a = self._attn(a, up.attn[i_block], f"up.{i_level}.attn.{i_block}")
            if i_level != 0:
                a = _Act(a.t, a.B, a.H, a.W, scale=float(up.upsample.scale_factor))
                if up.upsample.with_conv:
                    a = self._conv(a, up.upsample.conv, f"up.{i_level}.upsample")
                else:                                        # materialise the upsampled activation (identity 1x1 gather)
                    H, W = a.out_hw
                    a = _Act(K.conv_gather(a.t, a.B, a.H, a.W, H, W, 1, a.t.shape[1], inv_scale=1.0 / a.scale), a.B, H, W)
        if self.give_pre_end:
            out, co = a, a.t.shape[1]
        else:
            out, co = self._conv(a, self.conv_out, "conv_out", norm=self.norm_out, swish=True), self.conv_out.out_channels
        return out.t[:, :co].reshape(out.B, out.H, out.W, co).permute(0, 3, 1, 2).contiguous()
