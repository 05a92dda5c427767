"""MI355X-native routed decoder with the module / state-dict surface of the reference's
``libra/models/libra/modeling_libra.py`` (LibraLinear :150-206, LibraAttention :245-265, LibraMLP :208-238,
LibraDecoderLayer :416-435, LibraModel :524-600, MultiLMHead :834-843, LibraForCausalLM :845-940, forward :1069-1188).

The sub-modules only own the parameters (same names and shapes as the reference checkpoints: SURVEY §8b); the
compute is the kernel schedule in ``libra_amd/decoder_engine.py`` (forward and hand-written backward, exposed to
autograd through one ``torch.autograd.Function``), parity-tested against the reference fixtures.
Built: use_bridge on / off, concatenated (normed or not) or added signals, vision position embedding, dropout 0, with 1d or 2d RoPE (`use_2d_rope`), routed or unified heads
(`unified_head`), 1d or 2d vision prediction (`vision_prediction_mode`); anything else raises NotImplementedError rather
than silently diverging.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn
from transformers import PreTrainedModel
from transformers.modeling_outputs import CausalLMOutputWithPast

from .. import decoder_engine as DE
from ..common.registry import registry
from ..llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding     # noqa: F401  (re-exported: trainer.py:3 imports the name)
from .configuration_libra import LibraConfig
from .generation import LibraGenerationMixin


class _EngineOwned(nn.Module):
    """Base of the parameter-holder sub-modules: their arithmetic is the fused kernel schedule of
    ``libra_amd/decoder_engine.py`` driven by ``LibraForCausalLM.forward`` - calling one on its own has no fused equivalent."""

    def forward(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__} only owns parameters in libra_amd: the routed decoder runs as one fused "
                           "schedule - call LibraForCausalLM(...) (or libra_amd.decoder_engine.layer_forward) instead")


class LibraLinear(_EngineOwned):           # modeling_libra.py:150-206
    def __init__(self, in_features: int, out_features: int, bias: bool = False, down_ratio=4, rank=None):
        super().__init__()
        assert in_features % down_ratio == 0
        assert bias is False, "Not checked yet."
        self.in_features, self.out_features, self.down_ratio, self.rank = in_features, out_features, down_ratio, rank
        mid = rank if rank is not None else out_features // down_ratio
        self.weight_A = nn.Parameter(torch.empty((mid, in_features)))
        self.weight_B = nn.Parameter(torch.empty((out_features, mid)))
        self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight_A, a=math.sqrt(5))
        if rank is not None:
            nn.init.constant_(self.weight_B, 0.0)
        else:
            nn.init.kaiming_uniform_(self.weight_B, a=math.sqrt(5))


class LibraAttention(_EngineOwned):        # modeling_libra.py:245-265 (+ LlamaAttention.__init__ modeling_llama.py:207-228)
    def __init__(self, c: LibraConfig):
        super().__init__()
        H = c.hidden_size
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self, n, nn.Linear(H, H, bias=False))
        for n in ("vision_q_proj", "vision_k_proj", "vision_v_proj", "vision_o_proj"):
            setattr(self, n, LibraLinear(H, H, down_ratio=c.vision_down_ratio))
        # modeling_llama.py:224 - its persistent `inv_freq` buffer is part of every reference checkpoint's key set
        self.rotary_emb = LlamaRotaryEmbedding(H // c.num_attention_heads, max_position_embeddings=c.max_position_embeddings)
        if c.use_bridge:                   # :258 - without the bridge the layer is plain routed attention (no such parameters)
            for n in ("vision_v_bridge_on_language", "vision_v_bridge_on_vision", "vision_k_bridge_on_language",
                      "vision_k_bridge_on_vision"):
                setattr(self, n, LibraLinear(H, H, rank=c.bridge_rank))


class LibraMLP(_EngineOwned):              # modeling_libra.py:208-238 (+ LlamaMLP modeling_llama.py:185-201)
    def __init__(self, c: LibraConfig):
        super().__init__()
        H, I = c.hidden_size, c.intermediate_size
        self.gate_proj = nn.Linear(H, I, bias=False)
        self.down_proj = nn.Linear(I, H, bias=False)
        self.up_proj = nn.Linear(H, I, bias=False)
        self.vision_gate_proj = LibraLinear(H, I, down_ratio=c.vision_down_ratio)
        self.vision_down_proj = LibraLinear(I, H, down_ratio=c.vision_down_ratio)
        self.vision_up_proj = LibraLinear(H, I, down_ratio=c.vision_down_ratio)


class LibraDecoderLayer(_EngineOwned):     # modeling_libra.py:416-435
    def __init__(self, c: LibraConfig):
        super().__init__()
        self.self_attn = LibraAttention(c)
        self.mlp = LibraMLP(c)
        self.input_layernorm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)
        self.vision_input_layernorm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)
        self.vision_post_attention_layernorm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)


class LibraModel(_EngineOwned):            # modeling_libra.py:524-600 (parameter holder)
    def __init__(self, c: LibraConfig):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size, c.pad_token_id)
        self.layers = nn.ModuleList([LibraDecoderLayer(c) for _ in range(c.num_hidden_layers)])
        self.norm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)
        assert c.hidden_size % c.vision_codebook_num == 0
        self.vision_embed_tokens = nn.ModuleList([nn.Embedding(c.vision_vocab_size, c.hidden_size // c.vision_codebook_num)
                                                  for _ in range(c.vision_codebook_num)])
        self.vision_norm = LlamaRMSNorm(c.hidden_size, eps=c.rms_norm_eps)
        if c.concat_signals:               # :556-562 - the signal rides next to the codebook embeddings through one Linear ...
            self.vision_contiguous_signal_processor = nn.Linear(c.contiguous_signal_size + c.hidden_size, c.hidden_size, bias=False)
            if c.norm_signals:
                self.vision_signal_norm = LlamaRMSNorm(c.contiguous_signal_size + c.hidden_size, eps=c.rms_norm_eps)
        else:                              # ... or is projected on its own and added to the embeddings (:753-754)
            self.vision_contiguous_signal_processor = nn.Linear(c.contiguous_signal_size, c.hidden_size, bias=False)
        if c.use_vision_position_embedding:                                  # :564-566
            self.vision_position_embedding = nn.Embedding(c.max_vision_token_length, c.hidden_size)
        # modeling_libra.py:597 - PreTrainedModel.gradient_checkpointing_enable() flips it (both recipes do:
        # libra_pretrain.yaml:120); the engine then keeps only each layer's input and recomputes the layer in backward
        self.gradient_checkpointing = False


class MultiLMHead(_EngineOwned):           # modeling_libra.py:834-843
    def __init__(self, head_num, input_dim, output_dim):
        super().__init__()
        self.heads = nn.ModuleList([nn.Linear(input_dim, output_dim, bias=False) for _ in range(head_num)])


@dataclass
class LibraCausalLMOutputWithPast(CausalLMOutputWithPast):     # modeling_libra.py:98-109
    """`.logits` is the reference's [Q,B,S,V+514] tensor.  Under autograd (a training step, where HF Trainer reads only
    `.loss`) the 2.1 GB tensor (B=8, S=2048) is built on FIRST ACCESS of `.logits` / `["logits"]` from the compact per-modality
    logits the fused CE consumed; without autograd (evaluation / prediction) it is materialised eagerly, so `items()`,
    `to_tuple()` and `Trainer.prediction_step` see it.  It carries no grad_fn: a custom loss on `.logits` is not supported
    (the loss is fused - pass `labels`)."""
    past_hidden_states: Optional[torch.FloatTensor] = None
    past_vision_flag: Optional[torch.BoolTensor] = None

    def _materialize(self):
        lazy = self.__dict__.get("_lazy_logits")
        if lazy is None:
            return None
        val = DE.dense_logits(*lazy)
        self.__dict__["_lazy_logits"] = None
        self.logits = val                      # ModelOutput.__setattr__ also registers the dict key
        return val

    def __getattribute__(self, name):
        if name == "logits":
            d = object.__getattribute__(self, "__dict__")
            if d.get("_lazy_logits") is not None:
                return object.__getattribute__(self, "_materialize")()
        return super().__getattribute__(name)

    def __getitem__(self, k):
        if k == "logits" and self.__dict__.get("_lazy_logits") is not None:
            return self._materialize()
        return super().__getitem__(k)


class LibraForCausalLM(LibraGenerationMixin, PreTrainedModel):
    config_class = LibraConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = True
    _no_split_modules = ["LibraDecoderLayer"]

    def __init__(self, config: LibraConfig):
        super().__init__(config)
        c = config
        if c.vision_prediction_mode not in ("1d", "2d"):
            raise ValueError(f"vision_prediction_mode={c.vision_prediction_mode!r}: '1d' or '2d' (modeling_libra.py:857-861)")
        pred_2d = c.vision_prediction_mode == "2d"
        if pred_2d and c.unified_head:
            raise NotImplementedError("unified_head with vision_prediction_mode='2d' (the reference asserts it away, :1055)")
        if (pred_2d or c.use_2d_rope) and c.image_feature_resolution ** 2 + 2 != c.max_vision_token_length:
            raise ValueError("max_vision_token_length must be image_feature_resolution ** 2 + 2 (modeling_libra.py:572, :866)")
        if c.use_2d_rope and c.num_attention_heads % 2:
            raise ValueError("use_2d_rope alternates (row, column) over the heads: the head count must be even (:47-48)")
        for nm in ("resid_pdrop", "attn_pdrop", "embd_pdrop", "vision_resid_pdrop", "vision_embd_pdrop"):
            if getattr(c, nm, 0.0):
                raise NotImplementedError(f"{nm}={getattr(c, nm)}: dropout is 0 in both Libra recipes and is not built into "
                                          "the fused kernels")
        if c.hidden_size // c.num_attention_heads != 128:
            raise NotImplementedError("the fused bridge attention kernel is specialised for head_dim 128 (LLaMA-2-7B)")
        self.model = LibraModel(c)
        self.lm_head = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        self.vision_lm_head = MultiLMHead(c.vision_codebook_num, c.hidden_size * (2 if pred_2d else 1), c.vision_vocab_size)   # :858-861
        self.vision_hidden_placeholder = nn.Parameter(torch.empty(c.hidden_size))
        self.vision_hidden_placeholder.data.normal_(mean=0.0, std=c.initializer_range)
        self.max_vision_token_length = c.max_vision_token_length
        # the reference's four persistent buffers (modeling_libra.py:870-882): part of the checkpoint key set.  The fused
        # CE / logits kernels write the -inf padding and the EOI -> newline rule themselves; the buffers are carried, saved and loaded.
        self.register_buffer("naive_placeholder", torch.zeros(c.hidden_size))
        self.register_buffer("vision_logits_placeholder", torch.full([1, c.vision_vocab_size], -float("inf")))
        self.register_buffer("language_logits_placeholder", torch.full([1, c.vocab_size], -float("inf")))
        e2n = torch.full([1, 1, 1, c.vocab_size + c.vision_vocab_size], -float("inf"))
        e2n[:, :, :, c.newline_token_id] = float("inf")
        self.register_buffer("eoi_to_newline_logits_placeholder", e2n)
        self._dims = DE.DecDims(hidden=c.hidden_size, inter=c.intermediate_size, layers=c.num_hidden_layers,
                                heads=c.num_attention_heads, vocab=c.vocab_size, vision_vocab=c.vision_vocab_size,
                                codebooks=c.vision_codebook_num, max_vision_len=c.max_vision_token_length,
                                signal=c.contiguous_signal_size, rank=c.bridge_rank, down_ratio=c.vision_down_ratio,
                                eps=c.rms_norm_eps, max_pos=c.max_position_embeddings, rope_2d=bool(c.use_2d_rope),
                                unified_head=bool(c.unified_head), pred_2d=pred_2d, res=int(c.image_feature_resolution),
                                bridge=bool(c.use_bridge), concat=bool(c.concat_signals),
                                norm_sig=bool(c.concat_signals and c.norm_signals), vis_pos=bool(c.use_vision_position_embedding),
                                addition=bool(c.addition_mode))
        self._packed: Optional[DE.PackedOperands] = None
        self.post_init()

    def _init_weights(self, module):        # modeling_libra.py:502-519
        std = self.config.initializer_range
        if isinstance(module, LibraLinear):
            module.weight_A.data.normal_(mean=0.0, std=std)
            if module.rank is not None or self.config.addition_mode:
                module.weight_B.data.zero_()
            else:
                module.weight_B.data.normal_(mean=0.0, std=std)
        elif isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def _refresh_packed(self, sd, volatile: bool = True):
        """Fused operand copies ([q;k;v|bridge A], [gate;up], bridge B / B^T): built once, then refreshed in place - always
        for trainable parameters (optimizers that write through `.data`, e.g. DeepSpeed ZeRO, bump no version counter),
        by (data_ptr, _version) for frozen ones.  See DE.PackedOperands."""
        dev = sd["model.embed_tokens.weight"].device
        if self._packed is None or self._packed.device != dev:
            self._packed = DE.PackedOperands(sd, self._dims)
            self._packed.adopt(sd)       # this model's own parameters become views of the fused operands: nothing to re-copy per step
        else:
            self._packed.refresh(sd, volatile=volatile)
        return self._packed

    def invalidate_packed(self):
        """Force a full rebuild of the fused operand copies at the next forward (e.g. after writing FROZEN weights through
        `.data`; trainable ones are refreshed every forward anyway), and of the cached parameter walk."""
        self._packed = None
        self.__dict__.pop("_np_cache", None)

    def _named_params(self):
        """(names, parameters) in `named_parameters()` order without its per-call string building: the (module, key) owners are
        walked once and cached - the Parameter OBJECTS are looked up again on every call, so replaced parameters are seen; a module
        added or removed after the first forward needs `invalidate_packed()`.  (`named_parameters()` costs 1.2 ms for this model
        and ran twice per forward, right after the tensor assembly's host syncs, i.e. with the GPU idle.)"""
        c = self.__dict__.get("_np_cache")
        if c is None:
            names = [n for n, _ in self.named_parameters()]
            owners, seen = [], set()
            for mod in self.modules():
                for k, p in mod._parameters.items():
                    if p is not None and id(p) not in seen:
                        seen.add(id(p))
                        owners.append((mod, k))
            if len(owners) != len(names):
                raise RuntimeError("parameter walk disagrees with named_parameters()")
            c = self.__dict__["_np_cache"] = (names, owners)
        names, owners = c
        return names, [mod._parameters[k] for mod, k in owners]

    def _state(self, volatile: bool = True):
        names, params = self._named_params()
        sd = dict(zip(names, params))
        return sd, self._refresh_packed(sd, volatile)

    def _ptr_key(self):
        return tuple(p.data_ptr() for p in self.parameters())

    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids=None, past_key_values=None, inputs_embeds=None, labels: Optional[torch.LongTensor] = None,
                use_cache: Optional[bool] = None, output_attentions=None, output_hidden_states=None, return_dict=None,
                contiguous_signal: Optional[torch.Tensor] = None, vision_indices: Optional[torch.LongTensor] = None,
                past_hidden_states=None, past_vision_flag=None, max_cache_len: Optional[int] = None):
        if input_ids is not None and inputs_embeds is not None:                        # modeling_libra.py:703-704
            raise ValueError("You cannot specify both decoder_input_ids and decoder_inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:                                # :715-716
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        if vision_indices is None:
            raise ValueError("You have to specify vision_indices [B,S] (the modality of every position, :1116)")
        if output_attentions:
            raise NotImplementedError("attention maps are not produced by the fused kernels")
        lead = input_ids if input_ids is not None else inputs_embeds
        if not lead.is_cuda:
            raise RuntimeError("libra_amd LibraForCausalLM runs on MI355X only; got CPU tensors (no CPU fallback)")
        if input_ids is not None:
            assert len(input_ids) == self.config.vision_codebook_num                  # :705
        elif past_key_values is not None or use_cache:
            raise NotImplementedError("inputs_embeds with use_cache / past_key_values (generation feeds token ids)")
        if past_key_values is not None or use_cache:
            if labels is not None:
                raise ValueError("labels with use_cache / past_key_values: the cached path is inference only (:1142)")
            return self._forward_cached(input_ids, attention_mask, position_ids, past_key_values, contiguous_signal, vision_indices,
                                        max_cache_len)
        if position_ids is not None:
            raise NotImplementedError("custom position_ids on the training path (positions are arange(S), :736-739)")
        B, S = lead.shape[1:3] if input_ids is not None else lead.shape[:2]
        if attention_mask is None:
            attention_mask = torch.ones((B, S), dtype=torch.bool, device=lead.device)
        names, params = self._named_params()
        holder = {}
        # (grad mode is always off inside Function.forward and needs_input_grad ignores no_grad(): read it here)
        holder["grad_on"] = torch.is_grad_enabled()
        loss = _LibraFunction.apply(self, holder, names, input_ids, attention_mask, vision_indices, contiguous_signal, labels,
                                    bool(output_hidden_states), inputs_embeds, *params)
        out = holder["out"]
        if labels is None:
            loss = None
        hs = None
        if output_hidden_states:
            # the reference's tuple (modeling_libra.py:781-813): the input of every layer (embeddings, outputs of layers 0..L-2)
            # and the NORMED output of the last layer - its raw output never appears
            hs = tuple(h.view(B, S, -1) for h in out["hidden_states"][:-1]) + (out["hidden"],)
        training_step = out["saved"] is not None              # autograd will call back: the Trainer reads only .loss
        logits = None if training_step else DE.dense_logits(out, self._dims, B, S)
        res = LibraCausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=None, hidden_states=hs, attentions=None)
        res.__dict__["_lazy_logits"] = (out, self._dims, B, S) if training_step else None
        res.__dict__["_engine_out"] = out
        return res

    @torch.no_grad()
    def _forward_cached(self, input_ids, attention_mask, position_ids, past, signal, vision_indices, max_cache_len=None):
        """Generation path (LibraForCausalLM.forward with use_cache / past_key_values, modeling_libra.py:1118-1188):
        `past` None = prefill of the prompt (returns a filled DE.KVCache as `.past_key_values`); otherwise one new token per
        sequence.  Batched prompts are padded on the LEFT (positions = attention_mask.cumsum - 1, :1204-1207; pad keys are
        masked by the kernels through the cache's per-sequence start).  `.logits` is materialised ([Q,B,q,V+Vv], q = the tokens
        of this call) with the cached branch's rule that an EOI input token predicts nothing but a newline (:1141-1144).
        The KV cache holds `max_cache_len` tokens (default max_position_embeddings; generate() passes its max_length)."""
        Q, B, S = input_ids.shape
        dims = self._dims
        # A decode step continues the generation its prefill started: the operand copies were checked and refreshed there, and
        # as long as no parameter storage moved since (pack_key) the step reuses them and the parameter dict as they are - walking
        # ~800 parameters and ~1000 operand slices in Python per generated token costs more than a millisecond of an 8 ms step.
        ptr_key = self._ptr_key()
        if isinstance(past, DE.KVCache) and past.sd is not None and self._packed is not None and past.pack_key == ptr_key:
            sd, packed = past.sd, self._packed
        else:
            sd, packed = self._state(volatile=past is None)
        dev = input_ids.device
        if past is None:
            if attention_mask is None:
                attention_mask = torch.ones((B, S), dtype=torch.bool, device=dev)
            if position_ids is not None and dims.rope_2d:
                want = DE.positions_2d(vision_indices, dims, attention_mask).permute(0, 2, 1)
                if not torch.equal(position_ids.reshape(B, 2, S).to(dev).to(want.dtype), want):
                    raise NotImplementedError("prefill positions other than get_2d_position_ids(vision_indices, attention_mask)")
            elif position_ids is not None:
                am = attention_mask.to(torch.long)
                want = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
                if not torch.equal(position_ids.reshape(B, S).to(dev), want):
                    raise NotImplementedError("prefill positions other than attention_mask.cumsum(-1) - 1")
            cap = max(int(max_cache_len), S + 1) if max_cache_len else max(dims.max_pos, S + 1)
            cache = DE.KVCache(dims.layers, B, cap, dims.hidden, dev)
            cache.pack_key = ptr_key
            cache.sd = sd
            out = DE.forward(sd, packed, dims, input_ids, attention_mask, vision_indices, signal, None, cache=cache)
        else:
            if not isinstance(past, DE.KVCache):
                raise TypeError("past_key_values must be the KVCache returned by a previous call of this model")
            if S != 1:
                raise ValueError("only support generating token by token")             # cal_vision_logits_inference, :912
            cache = past
            if cache.length >= cache.capacity:
                raise ValueError(f"the KV cache is full ({cache.capacity} tokens): pass a larger max_cache_len / max_length")
            if cache.pack_key != ptr_key:                # parameter storage moved since the graphs were captured
                cache.graphs.clear()                     # (values may change freely: operands are refreshed in place)
                cache.pack_key, cache.sd = ptr_key, sd
            if dims.rope_2d:
                if position_ids is not None:                                            # [B, 2, 1] (:1199-1201) -> [B, 2]
                    position_ids = position_ids.reshape(B, 2)
            elif position_ids is None:                                                  # attention_mask.cumsum(-1) - 1, :1207
                position_ids = torch.full((B, 1), cache.length, dtype=torch.long, device=dev)
                if cache.start is not None:
                    position_ids = position_ids - cache.start.long()[:, None]
            out = DE.decode_step(sd, packed, dims, cache, input_ids, vision_indices, position_ids,
                                 use_graph=getattr(self, "decode_graphs", True))      # (False: launch the step kernel by kernel)
        logits = DE.dense_logits(out, dims, B, S)
        if past is not None:
            eoi = vision_indices[:, -1] == self.max_vision_token_length - 1
            forced = torch.full((logits.shape[-1],), float("-inf"), dtype=logits.dtype, device=dev)
            forced[self.config.newline_token_id] = float("inf")
            logits[:, :, -1, :] = torch.where(eoi[None, :, None], forced[None, None, :], logits[:, :, -1, :])   # no host read
        return LibraCausalLMOutputWithPast(loss=None, logits=logits, past_key_values=cache, hidden_states=None, attentions=None)

    # ---- generation glue (the reference's custom greedy_search / sample call these, modeling_libra_utils.py:61,:330) ----
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      vision_indices=None, contiguous_signal=None, past_hidden_states=None, past_vision_flag=None,
                                      **kwargs):
        """modeling_libra.py:1190-1231: with a cache only the last token, its position (attention_mask.cumsum - 1) and its
        vision index are fed, and no encoder signal."""
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds: the embedding gather is part of the fused path")
        if past_key_values:
            input_ids = input_ids[:, :, -1:]
        position_ids = kwargs.get("position_ids", None)
        dims = getattr(self, "_dims", None)
        if attention_mask is not None and position_ids is None and dims is not None and dims.rope_2d:
            # :1199-1201.  A cached step passes None: the cache carries get_2d_position_ids' running position, so the step's
            # (row, column) comes out in O(1) instead of a cumsum over the whole sequence - same values.
            if not past_key_values:
                position_ids = DE.positions_2d(vision_indices, dims, attention_mask).permute(0, 2, 1).long()
        elif attention_mask is not None and position_ids is None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, -1].unsqueeze(-1)
        if past_key_values:
            vision_indices = vision_indices[:, -1].unsqueeze(-1)
            contiguous_signal = None
        return {"input_ids": input_ids, "position_ids": position_ids, "past_key_values": past_key_values,
                "use_cache": kwargs.get("use_cache"), "attention_mask": attention_mask, "vision_indices": vision_indices,
                "contiguous_signal": contiguous_signal, "past_hidden_states": past_hidden_states,
                "past_vision_flag": past_vision_flag}

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder: bool = False,
                                            standardize_cache_format: bool = False):
        """modeling_libra.py:1233-1282: carry the cache, grow the attention mask by one, and advance the vision index - inside
        an image it counts up, after EOI (or in text) it stays at max_vision_token_length."""
        model_kwargs["past_hidden_states"] = getattr(outputs, "past_hidden_states", None)
        model_kwargs["past_vision_flag"] = getattr(outputs, "past_vision_flag", None)
        model_kwargs["past_key_values"] = outputs.past_key_values
        if "attention_mask" in model_kwargs:
            am = model_kwargs["attention_mask"]
            model_kwargs["attention_mask"] = torch.cat([am, am.new_ones((am.shape[0], 1))], dim=-1)
        if "vision_indices" in model_kwargs:
            vi = model_kwargs["vision_indices"]
            nxt = (vi[:, -1].clone() + 1).clamp_(max=self.max_vision_token_length)
            model_kwargs["vision_indices"] = torch.cat([vi, nxt[..., None]], dim=-1)
        return model_kwargs

    @staticmethod
    def _reorder_cache(past_key_values, beam_idx):
        """modeling_libra.py:1284-1289 for the KVCache container (beam search re-gathers the batch dimension)."""
        past_key_values.layers = [tuple(t.index_select(0, beam_idx) for t in layer) for layer in past_key_values.layers]
        past_key_values.flag = past_key_values.flag.index_select(0, beam_idx)
        for nm in ("start", "run2d", "hid"):
            t = getattr(past_key_values, nm, None)
            if t is not None:
                setattr(past_key_values, nm, t.index_select(0, beam_idx))
        past_key_values.B = int(beam_idx.numel())
        past_key_values.graphs.clear()                                   # captured steps point at the buffers just replaced
        return past_key_values

    @staticmethod
    def materialize_logits(output) -> torch.Tensor:
        """[Q,B,S,V+Vv] exactly as the reference's `.logits` (text rows [lm_head | -inf], vision rows [-inf | head_q])."""
        return output.logits


class _LibraFunction(torch.autograd.Function):
    """Autograd bridge: forward = DE.forward (saving activations when any parameter needs a gradient), backward =
    DE.backward (hand-written kernel schedule).  Only the parameters that require gradients get one (the reference's
    freeze policy, modeling_libra.py:1342-1369, is expressed through requires_grad exactly as upstream)."""

    @staticmethod
    def forward(ctx, model, holder, names, input_ids, attention_mask, vision_indices, signal, labels, want_hs, inputs_embeds, *params):
        sd = dict(zip(names, params))
        packed = model._refresh_packed(sd)
        need = labels is not None and holder.get("grad_on", True) and any(ctx.needs_input_grad[9:])     # ([9] = inputs_embeds)
        out = DE.forward(sd, packed, model._dims, input_ids, attention_mask, vision_indices, signal, labels,
                         want_hidden_states=want_hs, save=need,
                         recompute=need and model.model.gradient_checkpointing and model.training, inputs_embeds=inputs_embeds)
        holder["out"] = out
        ctx.model, ctx.sd, ctx.out, ctx.names = model, sd, out, names
        ctx.want = {n for n, ng in zip(names, ctx.needs_input_grad[10:]) if ng}
        ctx.embeds_dtype = inputs_embeds.dtype if inputs_embeds is not None and ctx.needs_input_grad[9] else None
        loss = out["loss"]
        return loss if loss is not None else torch.zeros((), device=vision_indices.device)

    @staticmethod
    def backward(ctx, gloss):
        if ctx.out.get("saved") is None:
            raise RuntimeError("backward through LibraForCausalLM requested but no activations were saved")
        # the incoming scalar (1.0, or a loss scale / 1/accum factor) is folded into the logits gradient ON THE DEVICE: no
        # elementwise kernel per parameter and no host read (which would drain the whole forward before backward is queued)
        grads = DE.backward(ctx.sd, ctx.model._packed, ctx.model._dims, ctx.out, ctx.want, gscale=gloss)
        res = []
        from .. import dp
        for n in ctx.names:
            gr = grads.get(n) if n in ctx.want and not dp.is_captured(n) else None      # captured: lives in the DP bucket
            if gr is not None:
                gr = gr.reshape(ctx.sd[n].shape)
            res.append(gr)
        ctx.out["saved"] = None
        d_emb = ctx.out.pop("d_inputs_embeds", None)
        d_emb = d_emb.to(ctx.embeds_dtype) if (d_emb is not None and ctx.embeds_dtype is not None) else None
        return (None,) * 9 + (d_emb,) + tuple(res)


def _cfg_get(cfg, key, default=None):
    g = getattr(cfg, "get", None)
    return g(key, default) if g is not None else getattr(cfg, key, default)


@registry.register_model("libra_train_wrapper")
class LibraTrainWrapper(PreTrainedModel):
    """What `train.py` builds (`registry.get_model_class("libra_train_wrapper").from_config(cfg)`, train.py:29-30) and HF
    Trainer steps: tokenizer -> labels -> LibraForCausalLM, with the reference's freeze switches.  Mirrors
    modeling_libra.py:1292-1437 (cfg fields `pretrained`, `custom_kwargs`, `tokenizer_kwargs`, `model_kwargs.{frozen_language,
    freeze_vision_value, freeze_text_embedding, freeze_vision_embedding, debug}`, `pretrained_weight`).

    `module=` / `tokenizer=` inject pre-built parts (random-init benchmarks and tests: no checkpoint exists offline)."""
    config_class = LibraConfig
    base_model_prefix = "module"
    supports_gradient_checkpointing = True
    # HF Trainer (>= 4.46) hands `num_items_in_batch` to any model whose forward has **kwargs; the wrapper's loss is the
    # reference's own mean over the label counts (modeling_libra.py:1160-1174) and takes no such argument
    accepts_loss_kwargs = False

    def __init__(self, config, *, module: Optional["LibraForCausalLM"] = None, tokenizer=None):
        from .tokenization_libra import LibraTokenizer, apply_freeze_policy
        pretrained = _cfg_get(config, "pretrained")
        if module is not None:
            libra_config = module.config
        else:
            import json
            import os
            with open(os.path.join(pretrained, "config.json")) as f:
                libra_config = LibraConfig(**json.load(f))
        super().__init__(libra_config)
        self.module = module if module is not None else LibraForCausalLM.from_pretrained(
            pretrained, **dict(_cfg_get(config, "custom_kwargs", {}) or {}))
        self.tokenizer = tokenizer if tokenizer is not None else LibraTokenizer(
            pretrained, **dict(_cfg_get(config, "tokenizer_kwargs", {}) or {}))
        tt = self.tokenizer.text_tokenizer
        self.change_pad_token_to_eos(pad_token_id=tt.pad_token_id, eos_token_id=tt.eos_token_id)
        weight = _cfg_get(config, "pretrained_weight", None)
        if weight is not None:
            sd = torch.load(weight, map_location="cpu")
            wrapped = any(k.startswith("model.model.") for k in sd)
            mod_wrapped = any(k.startswith("module.model.") for k in sd)
            assert not (wrapped and mod_wrapped), "'has_wrapper' and 'has_module_wrapper' cannot both be True."
            if mod_wrapped:                                                     # :1325-1331 (only this branch re-binds upstream)
                sd = {k[7:]: v for k, v in sd.items() if k.startswith("module.")}
            missing, unexpected = self.module.load_state_dict(sd, strict=False)
            print("missing keys: ", missing)
            print("unexpected keys: ", unexpected)
        mk = dict(_cfg_get(config, "model_kwargs", {}) or {})
        apply_freeze_policy(self.module, frozen_language=mk.get("frozen_language", False),
                            freeze_vision_value=mk.get("freeze_vision_value", False),
                            freeze_text_embedding=mk.get("freeze_text_embedding", False),
                            freeze_vision_embedding=mk.get("freeze_vision_embedding", False), debug=mk.get("debug", False))

    def _init_weights(self, module):
        pass                                           # the wrapped LibraForCausalLM initialises / loads its own weights

    @classmethod
    def get_model_from_config(cls, config):
        from .tokenization_libra import LibraTokenizer
        pretrained = _cfg_get(config, "pretrained")
        model = LibraForCausalLM.from_pretrained(pretrained, torch_dtype="auto", **dict(_cfg_get(config, "custom_kwargs", {}) or {}))
        return model, LibraTokenizer(pretrained, **dict(_cfg_get(config, "tokenizer_kwargs", {}) or {}))

    def change_pad_token_to_eos(self, pad_token_id=0, eos_token_id=2):
        """:1383-1388 - the padding row of the text embedding gets the EOS row's values (same scale as real tokens)."""
        emb = self.module.get_input_embeddings()
        emb.weight.data[pad_token_id] = emb.weight.data[eos_token_id].clone()

    def get_labels(self, inputs, label_mask_position_map):
        from .tokenization_libra import get_labels
        return get_labels(inputs, label_mask_position_map, boi_token_id=self.tokenizer.image_tokenizer.boi_token_id,
                          bos_token_id=self.tokenizer.text_tokenizer.bos_token_id)

    def forward(self, samples, return_loss=None, **kwargs):
        kwargs.pop("num_items_in_batch", None)
        inputs = self.tokenizer(samples, return_tensors="pt", padding="longest",
                                max_length=self.tokenizer.text_tokenizer.model_max_length, truncation=True)
        labels = self.get_labels(inputs, samples["label_mask_position_map"])
        return self.module(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                           vision_indices=inputs["vision_indices"], contiguous_signal=inputs["coninous_signal"],
                           labels=labels, use_cache=False, **kwargs)

    @classmethod
    def from_config(cls, config):
        return cls(config)
