"""VQModel (encode path) — mirror of /root/reference/libra/models/libra/taming/models/vqgan.py:26-114.

``encode`` = CLIP tower -> quant_conv (1x1 conv == GEMM with bias) -> LFQ, all on gfx950 kernels.
``decode`` / ``decode_code`` (image generation, SURVEY §8f-2, vqgan.py:122-130) = LFQ codes -> post_quant_conv -> taming
Decoder on the kernels of ``vq_decoder.py``; built when the ddconfig carries the decoder's fields (``ch``, ``ch_mult`` ...).
The tokenizer *training* parts of the reference class (losses, training_step) are outside the hot path.
"""
import torch
import torch.nn as nn

from .. import kernels as K
from .clip_encoder import CLIPVisionTower
from .lookup_free_quantization import LFQ
from .vq_decoder import Decoder, _Act


class VQModel(nn.Module):
    def __init__(self, ddconfig, embed_dim, lossconfig=None, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, codebook_size=512, num_codebook=2, disable_loss=False,
                 vision_model=None):
        super().__init__()
        self.image_key = image_key
        self.encoder_name = ddconfig.get("encoder_name", "default")
        self.use_clip = "clip" in self.encoder_name
        if not self.use_clip and vision_model is None:
            raise NotImplementedError("only the CLIP-tower encoder is on the Libra hot path (vqgan.py:44-56; the "
                                      "conv Encoder is dead code there)")
        select_layer = ddconfig.get("select_layer", -2)
        self.encoder = CLIPVisionTower(vision_tower=self.encoder_name, square_output=True, select_layer=select_layer,
                                       model=vision_model)
        self.quantize = LFQ(dim=embed_dim, codebook_size=codebook_size, num_codebooks=num_codebook,
                            entropy_loss_weight=0.1, commitment_loss_weight=1., diversity_gamma=2.5)
        n_sel = len(self.encoder._select_list())
        self.quant_conv = nn.Conv2d(self.encoder.vision_tower.config.hidden_size * n_sel, embed_dim, 1)
        self.embed_dim = embed_dim
        self.has_decoder = all(k in ddconfig for k in ("ch", "out_ch", "num_res_blocks", "attn_resolutions", "resolution", "z_channels"))
        if self.has_decoder:                                              # vqgan.py:58,:75
            dd = {k: v for k, v in dict(ddconfig).items() if k not in ("encoder_name", "select_layer", "only_auto_encoder")}
            dd.setdefault("in_channels", 3)
            self.decoder = Decoder(**dd)
            self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    @property
    def device(self):
        return self.quant_conv.weight.device

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        skipped = [] if self.has_decoder else [k for k in sd if k.startswith(("decoder.", "post_quant_conv."))]
        for k in skipped:
            del sd[k]
        self.load_state_dict(sd, strict=True)
        print(f"Restored from {path}" + (f" ({len(skipped)} decoder-side keys skipped: no decoder fields in ddconfig)" if skipped else ""))

    def _quant_conv(self, feat2d, w):
        E = self.embed_dim
        out = None
        if E % 8:                          # e.g. E == 18 (no LFQ projection): 16-byte row alignment for the GEMM store
            out = torch.empty((feat2d.shape[0], K.round_up(E, 8)), dtype=feat2d.dtype, device=feat2d.device)[:, :E]
        return K.gemm_nt(feat2d, w, bias=self.quant_conv.bias, out=out)

    def encode_flat(self, x, *, offset=0, boi=0, eoi=0, want_ids=False, want_quant=True, want_xpre=False):
        """Fused entry: -> (feat [B,hw,C], h2d [B*hw,E], indices [B*hw,Q], ids|None, xpre|None, quant2d|None)."""
        feat = self.encoder.forward_flat(x)                              # [B, hw, C]
        B, hw, Cf = feat.shape
        w = self.quant_conv.weight.view(self.embed_dim, Cf)
        h2d = self._quant_conv(feat.view(B * hw, Cf), w)                                # vqgan.py:108
        idx, ids, xpre, q2d = self.quantize.encode_flat(h2d, B, hw, offset=offset, boi=boi, eoi=eoi,
                                                        want_ids=want_ids, want_quant=want_quant, want_xpre=want_xpre)
        return feat, h2d, idx, ids, xpre, q2d

    @torch.no_grad()
    def encode(self, x, return_encoder_feat=False):
        feat, h2d, idx, _, _, q2d = self.encode_flat(x)
        B, hw, Cf = feat.shape
        g = int(round(hw ** 0.5))
        quant = q2d.view(B, g, g, self.embed_dim).permute(0, 3, 1, 2)
        info = idx.view(B, g, g, self.quantize.num_codebooks)
        emb_loss = self.quantize.zero * 0.1 + self.quantize.zero * 1.
        if return_encoder_feat:
            return quant, emb_loss, info, feat.view(B, g, g, Cf).permute(0, 3, 1, 2)
        return quant, emb_loss, info

    def encode_without_quant(self, x):
        feat = self.encoder.forward_flat(x)
        B, hw, Cf = feat.shape
        g = int(round(hw ** 0.5))
        h2d = self._quant_conv(feat.view(B * hw, Cf), self.quant_conv.weight.view(self.embed_dim, Cf))
        return h2d.view(B, g, g, self.embed_dim).permute(0, 3, 1, 2), None, None

    def _decode_act(self, codes2d, B, H, W):
        if not self.has_decoder:
            raise NotImplementedError("this VQModel was built without the decoder's ddconfig fields (ch, ch_mult, ...): encode only")
        wp, b = self.decoder._w(self.post_quant_conv, "post_quant_conv")
        op = codes2d if codes2d.shape[1] == wp.shape[1] and codes2d.is_contiguous() else \
            K.conv_gather(codes2d.contiguous(), B, H, W, H, W, 1, wp.shape[1])
        zc = self.post_quant_conv.out_channels
        z = torch.empty((B * H * W, K.round_up(zc, 8)), dtype=torch.bfloat16, device=codes2d.device)
        K.gemm_nt(op, wp, out=z[:, :zc], bias=b)                                          # vqgan.py:123
        if zc % 8:
            raise NotImplementedError("z_channels must be a multiple of 8")
        return self.decoder(_Act(z, B, H, W))

    @torch.no_grad()
    def decode(self, quant):
        """quant [B, embed_dim, h, w] -> image [B, out_ch, H, W]  (vqgan.py:122-125)."""
        B, E, H, W = quant.shape
        return self._decode_act(quant.to(torch.bfloat16).permute(0, 2, 3, 1).reshape(B * H * W, E).contiguous(), B, H, W)

    @torch.no_grad()
    def decode_code(self, code_b):
        """code indices int64 [B, h, w, Q] -> image  (vqgan.py:127-130)."""
        B, H, W, Q = code_b.shape
        return self._decode_act(self.quantize.indices_to_codes_flat(code_b.reshape(B * H * W, Q)), B, H, W)
