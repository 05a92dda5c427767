"""MI355X-native CLIP vision tower with the module / state-dict surface of the reference's
``libra/models/clip/modeling_clip.py`` (vision half).

Same class names, constructor arguments, parameter names (``vision_model.embeddings.*``,
``vision_model.pre_layrnorm.*`` [sic], ``vision_model.encoder.layers.{i}.*``,
``vision_model.post_layernorm.*``), ``forward`` signature and output container as
``CLIPVisionModel`` (/root/reference/libra/models/clip/modeling_clip.py:921-972), so reference
checkpoints load with ``from_pretrained`` / ``load_state_dict`` and ``CLIPVisionTower`` can use it
unchanged.  The sub-modules only *own* the parameters; the compute is one schedule of hand-written
gfx950 kernels (``libra_amd/vit_engine.py``), forward and backward, exposed to autograd through a
single ``torch.autograd.Function``.  There is no eager / CPU fallback: calling it without the HIP
library or on a non-GPU tensor raises.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn
from transformers import CLIPVisionConfig, PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithPooling

from .. import kernels as K
from .. import vit_engine as E

__all__ = ["CLIPVisionModel", "CLIPVisionConfig", "CLIPVisionTransformer"]


# ---- parameter holders (names/shapes == reference; never called) ------------------------------------
class CLIPVisionEmbeddings(nn.Module):          # modeling_clip.py:170-191
    def __init__(self, config):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.class_embedding = nn.Parameter(torch.randn(self.embed_dim))
        self.patch_embedding = nn.Conv2d(config.num_channels, self.embed_dim, kernel_size=config.patch_size,
                                         stride=config.patch_size, bias=False)
        self.num_patches = (config.image_size // config.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = nn.Embedding(self.num_positions, self.embed_dim)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)), persistent=False)


class CLIPAttention(nn.Module):                 # modeling_clip.py:262-285
    def __init__(self, config):
        super().__init__()
        d = config.hidden_size
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class CLIPMLP(nn.Module):                       # modeling_clip.py:366-372
    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)


class CLIPEncoderLayer(nn.Module):              # modeling_clip.py:381-388
    def __init__(self, config):
        super().__init__()
        self.self_attn = CLIPAttention(config)
        self.layer_norm1 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.mlp = CLIPMLP(config)
        self.layer_norm2 = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class CLIPEncoder(nn.Module):                   # modeling_clip.py:600-613
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.gradient_checkpointing = False


class CLIPVisionTransformer(nn.Module):         # modeling_clip.py:859-870
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = CLIPVisionEmbeddings(config)
        self.pre_layrnorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.encoder = CLIPEncoder(config)
        self.post_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


# ---- autograd bridge -------------------------------------------------------------------------------
class _VitFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, pixel_values, *params):
        names = model._param_names
        pd = dict(zip(names, params))
        dims = model._dims
        need_grad = any(ctx.needs_input_grad)      # grad mode is always off inside Function.forward
        packed = model._packed_forward(pd)
        hs, saved = E.forward(pd, packed, pixel_values, dims, save=need_grad)
        B = pixel_values.shape[0]
        ctx.model, ctx.saved, ctx.pd, ctx.packed = model, saved, pd, packed
        ctx.pixel_needs_grad = pixel_values.requires_grad
        # hidden states nobody differentiated arrive as None, not as zero tensors: the engine then starts the backward at the
        # last layer that has a cotangent (select_layer = [-2, -3] never reaches the last encoder layer), adds nothing for
        # the others and keeps the fused bias-gradient path
        ctx.set_materialize_grads(False)
        return tuple(h.view(B, dims.tokens, dims.hidden) for h in hs)

    @staticmethod
    def backward(ctx, *dhs):
        model = ctx.model
        if ctx.saved is None:
            raise RuntimeError("backward through CLIPVisionModel requested but no activations were saved")
        dhs = [None if g is None else g for g in dhs]
        dpixel, grads = E.backward(ctx.pd, ctx.packed, ctx.saved, dhs, model._dims,
                                   need_pixel_grad=ctx.pixel_needs_grad)
        out = []
        from .. import dp
        for n, p in ctx.pd.items():
            g = None if dp.is_captured(n) else grads.get(n)       # captured gradients live in the data-parallel buckets
            if g is not None and g.shape != p.shape:
                g = g.reshape(p.shape)
            out.append(g if (g is not None and p.requires_grad) else None)
        ctx.saved = None
        return (None, dpixel, *out)


class CLIPVisionModel(PreTrainedModel):
    """Drop-in for the reference ``CLIPVisionModel`` (modeling_clip.py:921-972)."""
    config_class = CLIPVisionConfig
    base_model_prefix = "clip"
    main_input_name = "pixel_values"
    supports_gradient_checkpointing = False
    _no_split_modules = ["CLIPEncoderLayer"]

    def __init__(self, config: CLIPVisionConfig):
        super().__init__(config)
        if config.hidden_act != "quick_gelu":
            raise ValueError("only hidden_act='quick_gelu' (CLIP ViT-L/14) is implemented in the fused GEMM epilogue")
        self.vision_model = CLIPVisionTransformer(config)
        self._dims = E.VitDims(hidden=config.hidden_size, inter=config.intermediate_size,
                               layers=config.num_hidden_layers, heads=config.num_attention_heads,
                               patch=config.patch_size, image=config.image_size, channels=config.num_channels,
                               eps=config.layer_norm_eps)
        self._pack = E._Packed()
        self.post_init()
        self._param_names = [n for n, _ in self.named_parameters()]

    # reference initialiser (modeling_clip.py:442-493), factor = config.initializer_factor
    def _init_weights(self, module):
        f = self.config.initializer_factor
        c = self.config
        if isinstance(module, CLIPVisionEmbeddings):
            nn.init.normal_(module.class_embedding, mean=0.0, std=module.embed_dim ** -0.5 * f)
            nn.init.normal_(module.patch_embedding.weight, std=c.initializer_range * f)
            nn.init.normal_(module.position_embedding.weight, std=c.initializer_range * f)
        elif isinstance(module, CLIPAttention):
            d = c.hidden_size
            in_std = (d ** -0.5) * ((2 * c.num_hidden_layers) ** -0.5) * f
            for m in (module.q_proj, module.k_proj, module.v_proj):
                nn.init.normal_(m.weight, std=in_std)
            nn.init.normal_(module.out_proj.weight, std=(d ** -0.5) * f)
        elif isinstance(module, CLIPMLP):
            d = c.hidden_size
            nn.init.normal_(module.fc1.weight, std=(2 * d) ** -0.5 * f)
            nn.init.normal_(module.fc2.weight, std=(d ** -0.5) * ((2 * c.num_hidden_layers) ** -0.5) * f)
        if isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def get_input_embeddings(self) -> nn.Module:
        return self.vision_model.embeddings.patch_embedding

    # ---- packed operand copies (refreshed in place every forward for trainable parameters: vit_engine._Packed) ----
    def _packed_forward(self, pd):
        return self._pack.get(pd, self._dims)

    def forward(self, pixel_values: Optional[torch.FloatTensor] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                ) -> Union[Tuple, BaseModelOutputWithPooling]:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")            # modeling_clip.py:889-890
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised by the flash kernel")
        if not pixel_values.is_cuda:
            raise RuntimeError("libra_amd CLIPVisionModel runs on MI355X only; got a CPU tensor (no CPU fallback)")
        params = [p for _, p in self.named_parameters()]
        if any(p.dtype != torch.bfloat16 for p in params):
            raise RuntimeError("parameters must be bfloat16 (the reference casts the whole model: train.py:31-32)")
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else self.config.output_hidden_states)
        return_dict = return_dict if return_dict is not None else getattr(self.config, "return_dict", True)
        hs = _VitFunction.apply(self, pixel_values, *params)
        last = hs[-1]
        # pooled output = post_layernorm(CLS)  (modeling_clip.py:902-904); unused by Libra, not differentiable here
        with torch.no_grad():
            cls = last[:, 0, :].contiguous()
            pooled, _, _ = K.layernorm_fwd(cls, self.vision_model.post_layernorm.weight,
                                           self.vision_model.post_layernorm.bias, self._dims.eps, save_stats=False)
        if not return_dict:
            return (last, pooled) + ((tuple(hs),) if output_hidden_states else ())
        return BaseModelOutputWithPooling(last_hidden_state=last, pooler_output=pooled,
                                          hidden_states=tuple(hs) if output_hidden_states else None, attentions=None)
