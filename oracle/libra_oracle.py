"""CPU oracle for the routed ("bridge") LLaMA decoder half of the Libra hot path.

TEST INFRASTRUCTURE ONLY (see vit_oracle.py header for the import rule).

Clean-room functional restatement (plain torch, dtype-agnostic) in *closed form* — the reference routes
by boolean-mask gather/scatter and duplicates the attention matmuls; here every routed op is written as
a per-token select, and attention as

    S_ij = q_i . (k_j + [m_i != m_j] kb_j) / sqrt(d)  + mask_ij ,   O_i = sum_j P_ij (v_j + [m_i != m_j] vb_j)

which is algebraically what the reference computes:

  * cal_language_vision            /root/reference/libra/models/libra/modeling_libra.py:111-147
  * LibraLinear.forward            :192-199   (x A^T B^T; rank-8 bridges have rank != None)
  * LlamaRMSNorm.forward           /root/reference/libra/models/llama/modeling_llama.py:127-132
  * rotate_half / apply_rotary_pos_emb   modeling_libra.py:32-61 ; LlamaRotaryEmbedding  modeling_llama.py:135-164
  * LibraAttention.forward + attn_with_bridge     modeling_libra.py:267-414
  * LibraMLP.forward               :227-238
  * LibraDecoderLayer.forward      :437-491
  * LibraModel.get_inputs_embeds_from_multicodebook :625-661, forward :680-831,
    _prepare_decoder_attention_mask :602-623, _make_causal_mask/_expand_mask modeling_llama.py:44-73
  * LibraForCausalLM.cal_vl_logits :1018-1052, loss :1160-1174
  * LibraTrainWrapper.get_labels   :1397-1411
  * LibraTokenizer.forward tensor assembly  /root/reference/libra/models/libra/tokenization_libra.py:250-316

Pinned by tests/test_oracle_libra_golden.py against fixtures generated from the reference's own
LibraForCausalLM (tests/golden/make_golden_libra.py).  State-dict keys are the reference's.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """modeling_llama.py:127-132: variance in fp32, product promoted to fp32, `weight * x.to(input_dtype)`."""
    dt = x.dtype
    xf = x.to(torch.float32) if dt != torch.float64 else x
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def routed(x: torch.Tensor, flag: torch.Tensor, f_lang, f_vis, addition: bool = False) -> torch.Tensor:
    """cal_language_vision (modeling_libra.py:111-137): language fn on ~flag rows, vision fn on flag rows; with addition_mode
    (:112-127, the q / k / v / o projections only) the language fn on EVERY row and the vision fn added on the flag rows."""
    yl, yv = f_lang(x), f_vis(x)
    return torch.where(flag.unsqueeze(-1), yl + yv if addition else yv, yl)


def libra_linear(x, sd, pre):
    return F.linear(F.linear(x, sd[pre + "weight_A"]), sd[pre + "weight_B"])


def rope_tables(dim: int, n_pos: int, base: float = 10000.0, dtype=torch.float32):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(n_pos, dtype=inv.dtype)
    freqs = torch.einsum("i,j->ij", t, inv)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)       # cached in fp32, cast to x.dtype (modeling_llama.py:161-164)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin, position_ids):
    """position_ids [B,S] (1d) or [B,2,S] (use_2d_rope, modeling_libra.py:43-49: heads alternate between the row and the column
    position, `cos[position_ids].repeat(1, heads // 2, 1, 1)`)."""
    if position_ids.dim() == 3:
        rep_ = x.shape[1] // 2
        c = cos[position_ids].repeat(1, rep_, 1, 1)         # [B, heads, S, d]
        s = sin[position_ids].repeat(1, rep_, 1, 1)
    else:
        c = cos[position_ids].unsqueeze(1)      # [B,1,S,d]
        s = sin[position_ids].unsqueeze(1)
    return x * c + rotate_half(x) * s


def position_ids_2d(vision_indices: torch.Tensor, max_vision_token_length: int, res: int, attention_mask=None) -> torch.Tensor:
    """LibraModel.get_2d_position_ids (modeling_libra.py:663-678) -> [B, 2, S]: text / BOI tokens advance a running position by
    one, EOI by res + 1, grid tokens sit at (running position of their BOI) + (row, column) in 1..res."""
    L = max_vision_token_length
    m = (vision_indices == L) | (vision_indices == 0)
    if attention_mask is not None:
        m = m & (attention_mask != 0)
    m = m.long()
    m[vision_indices == L - 1] = res + 1
    pos = (m.cumsum(-1) - 1)[..., None].expand(-1, -1, 2)
    hh = torch.arange(1, res + 1)[:, None].expand(-1, res)
    ww = torch.arange(1, res + 1)[None, :].expand(res, -1)
    off = torch.cat([torch.zeros(1, 2, dtype=torch.long), torch.stack([hh, ww], -1).reshape(-1, 2), torch.zeros(2, 2, dtype=torch.long)], 0)
    pos = pos + off[vision_indices]
    if attention_mask is not None:
        pos = pos.clone()
        pos[attention_mask == 0] = 1
    return pos.permute(0, 2, 1)


def additive_mask(attention_mask: torch.Tensor, S: int, dtype) -> torch.Tensor:
    """[B,1,S,S]: causal (finfo.min above the diagonal) + padding (finfo.min on masked KEYS), summed as the
    reference does (modeling_libra.py:602-623)."""
    mn = torch.finfo(dtype).min
    causal = torch.full((S, S), mn, dtype=dtype)
    causal = torch.triu(causal, diagonal=1)
    B = attention_mask.shape[0]
    exp = attention_mask[:, None, None, :].expand(B, 1, S, S).to(dtype)
    inv = 1.0 - exp
    pad = inv.masked_fill(inv.to(torch.bool), mn)
    return pad + causal[None, None]


def attention(sd, pre, x, flag, mask, position_ids, heads: int, cos, sin, addition: bool = False) -> torch.Tensor:
    B, S, Hd = x.shape
    d = Hd // heads
    q = routed(x, flag, lambda t: F.linear(t, sd[pre + "q_proj.weight"]), lambda t: libra_linear(t, sd, pre + "vision_q_proj."), addition)
    k = routed(x, flag, lambda t: F.linear(t, sd[pre + "k_proj.weight"]), lambda t: libra_linear(t, sd, pre + "vision_k_proj."), addition)
    v = routed(x, flag, lambda t: F.linear(t, sd[pre + "v_proj.weight"]), lambda t: libra_linear(t, sd, pre + "vision_v_proj."), addition)
    if pre + "vision_k_bridge_on_language.weight_A" in sd:
        kb = routed(x, flag, lambda t: libra_linear(t, sd, pre + "vision_k_bridge_on_language."),
                    lambda t: libra_linear(t, sd, pre + "vision_k_bridge_on_vision."))
        vb = routed(x, flag, lambda t: libra_linear(t, sd, pre + "vision_v_bridge_on_language."),
                    lambda t: libra_linear(t, sd, pre + "vision_v_bridge_on_vision."))
    else:                                          # config.use_bridge = False (:258, :311-317, :394): plain routed attention
        kb, vb = torch.zeros_like(k), torch.zeros_like(v)
    # the reference adds the bridge to K *before* RoPE and rotates both variants (:320-340); RoPE is linear, so
    # rope(k + kb) = rope(k) + rope(kb) up to rounding — keep the reference's order of operations.
    k_cross = k + kb

    def heads_(t):
        return t.view(B, S, heads, d).transpose(1, 2)
    q, k, k_cross, v, vb = map(heads_, (q, k, k_cross, v, vb))
    q = apply_rope(q, cos, sin, position_ids)
    k = apply_rope(k, cos, sin, position_ids)
    k_cross = apply_rope(k_cross, cos, sin, position_ids)
    cross = (flag[:, :, None] != flag[:, None, :]).unsqueeze(1)            # [B,1,S,S]: m_i != m_j
    s_same = q @ k.transpose(-1, -2) / (d ** 0.5)
    s_cross = q @ k_cross.transpose(-1, -2) / (d ** 0.5)
    s = torch.where(cross, s_cross, s_same) + mask
    s = torch.max(s, torch.tensor(torch.finfo(s.dtype).min, dtype=s.dtype))     # :385-388
    p = torch.softmax(s, dim=-1, dtype=torch.float32 if s.dtype != torch.float64 else torch.float64).to(q.dtype)
    o = p @ v + (p * cross.to(p.dtype)) @ vb                                    # attn_with_bridge :267-296
    o = o.transpose(1, 2).reshape(B, S, Hd)
    return routed(o, flag, lambda t: F.linear(t, sd[pre + "o_proj.weight"]), lambda t: libra_linear(t, sd, pre + "vision_o_proj."), addition)


def mlp(sd, pre, x, flag):
    def lang(t):
        return F.linear(F.silu(F.linear(t, sd[pre + "gate_proj.weight"])) * F.linear(t, sd[pre + "up_proj.weight"]),
                        sd[pre + "down_proj.weight"])

    def vis(t):
        return libra_linear(F.silu(libra_linear(t, sd, pre + "vision_gate_proj.")) * libra_linear(t, sd, pre + "vision_up_proj."),
                            sd, pre + "vision_down_proj.")
    return routed(x, flag, lang, vis)


def decoder_layer(sd, i, x, flag, mask, position_ids, heads, eps, cos, sin, addition: bool = False):
    pre = f"model.layers.{i}."
    h = routed(x, flag, lambda t: rms_norm(t, sd[pre + "input_layernorm.weight"], eps),
               lambda t: rms_norm(t, sd[pre + "vision_input_layernorm.weight"], eps))
    x = x + attention(sd, pre + "self_attn.", h, flag, mask, position_ids, heads, cos, sin, addition)
    h = routed(x, flag, lambda t: rms_norm(t, sd[pre + "post_attention_layernorm.weight"], eps),
               lambda t: rms_norm(t, sd[pre + "vision_post_attention_layernorm.weight"], eps))
    return x + mlp(sd, pre + "mlp.", h, flag)


def input_embeds(sd, input_ids: torch.Tensor, flag: torch.Tensor, signal: Optional[torch.Tensor], vocab: int, eps: float,
                 vision_indices: Optional[torch.Tensor] = None):
    """get_inputs_embeds_from_multicodebook (:625-661) + the un-concatenated signal path of LibraModel.forward (:753-754).  The
    configuration is read off the state dict, as the reference's constructor shapes it (:553-566): a `vision_position_embedding`
    table = use_vision_position_embedding; a `vision_signal_norm` weight = concat_signals and norm_signals; a processor whose
    input width is the signal width alone = concat_signals False."""
    Q = input_ids.shape[0]
    lang_ids = torch.where(flag, torch.zeros_like(input_ids[0]), input_ids[0])
    lang = F.embedding(lang_ids, sd["model.embed_tokens.weight"])
    vis = []
    for q in range(Q):
        vid = torch.where(flag, input_ids[q] - vocab, torch.zeros_like(input_ids[q]))
        vis.append(F.embedding(vid, sd[f"model.vision_embed_tokens.{q}.weight"]))
    vis = torch.cat(vis, dim=-1)
    if "model.vision_position_embedding.weight" in sd:                         # :636-638
        pos = torch.where(flag, vision_indices, torch.zeros_like(vision_indices))
        vis = vis + F.embedding(pos, sd["model.vision_position_embedding.weight"])
    wp = sd["model.vision_contiguous_signal_processor.weight"]
    concat = wp.shape[1] > vis.shape[-1]                                        # Linear(Cs + H -> H) vs Linear(Cs -> H), :556-562
    if concat:
        if signal is None:
            signal = vis.new_zeros(vis.shape[:-1] + (wp.shape[1] - vis.shape[-1],))          # :646-653
        ve = torch.cat([vis, signal.to(vis.dtype)], dim=-1)
        if "model.vision_signal_norm.weight" in sd:
            ve = rms_norm(ve, sd["model.vision_signal_norm.weight"], eps)
        ve = F.linear(ve, wp)
        return torch.where(flag.unsqueeze(-1), ve, lang)
    x = torch.where(flag.unsqueeze(-1), vis, lang)
    if signal is not None:                                                      # :753-754, every position (the signal is 0 off-image)
        x = x + F.linear(signal.to(x.dtype), wp)
    return x


def model_forward(sd, input_ids, attention_mask, vision_indices, signal, *, layers: int, heads: int, vocab: int,
                  max_vision_token_length: int, eps: float = 1e-6, max_pos: int = 2048, rope_2d_res: Optional[int] = None,
                  hidden_states: Optional[list] = None, addition: bool = False):
    """LibraForCausalLM up to the final routed norm: -> hidden [B,S,H], vision_flag.  `rope_2d_res` = image_feature_resolution
    switches on use_2d_rope (position ids from `position_ids_2d`, :731-733).  `hidden_states` (a list) receives the reference's
    `output_hidden_states` tuple: the embeddings and every layer's output, before the final norm (:781-807)."""
    flag = vision_indices < max_vision_token_length                      # :1118
    assert torch.equal(flag, input_ids[0] >= vocab)                        # :707-710
    B, S = input_ids.shape[1:]
    x = input_embeds(sd, input_ids, flag, signal, vocab, eps, vision_indices)
    d = x.shape[-1] // heads
    if rope_2d_res is not None:
        pos = position_ids_2d(vision_indices, max_vision_token_length, rope_2d_res)
        cos, sin = rope_tables(d, max(max_pos, int(pos.max()) + 1), dtype=x.dtype)
    else:
        cos, sin = rope_tables(d, max(max_pos, S), dtype=x.dtype)
        pos = torch.arange(S).unsqueeze(0).expand(B, S)
    mask = additive_mask(attention_mask, S, x.dtype)
    if hidden_states is not None:
        hidden_states.append(x)
    for i in range(layers):
        x = decoder_layer(sd, i, x, flag, mask, pos, heads, eps, cos, sin, addition)
        if hidden_states is not None:
            hidden_states.append(x)
    x = routed(x, flag, lambda t: rms_norm(t, sd["model.norm.weight"], eps),
               lambda t: rms_norm(t, sd["model.vision_norm.weight"], eps))
    return x, flag


def vl_logits(sd, hidden, flag, Q: int):
    """cal_vl_logits (unified_head off, 1d): [Q,B,S,V+Vv]; text rows: [lm_head | -inf], vision rows: [-inf | head_q]."""
    lang = F.linear(hidden, sd["lm_head.weight"])
    V = lang.shape[-1]
    outs = []
    for q in range(Q):
        vis = F.linear(hidden, sd[f"vision_lm_head.heads.{q}.weight"])
        Vv = vis.shape[-1]
        neg_v = torch.full(vis.shape, float("-inf"), dtype=lang.dtype)
        neg_l = torch.full(lang.shape, float("-inf"), dtype=lang.dtype)
        row_l = torch.cat([lang, neg_v], -1)
        row_v = torch.cat([neg_l, vis], -1)
        outs.append(torch.where(flag.unsqueeze(-1), row_v, row_l))
    return torch.stack(outs)


def vl_logits_unified(sd, hidden, Q: int):
    """cal_vl_logits with unified_head (modeling_libra.py:1054-1064, training / uncached): every row gets [lm_head | head_q]."""
    lang = F.linear(hidden, sd["lm_head.weight"])
    return torch.stack([torch.cat([lang, F.linear(hidden, sd[f"vision_lm_head.heads.{q}.weight"])], -1) for q in range(Q)])


def vision_features_2d(sd, hidden, flag, max_vision_token_length: int, res: int):
    """cal_vision_logits_train (modeling_libra.py:942-1014), complete images only: the input of the vision heads in
    vision_prediction_mode="2d".  The row at the position that PREDICTS grid cell (i, j) is cat(up, left) = (hidden of cell
    (i-1, j) | hidden of cell (i, j-1)), the learned placeholder where there is none - except left of cell (0, 0), which is BOI;
    the last grid token's row (predicting EOI) is (itself | placeholder), the EOI row (no loss) likewise. -> [n_vision, 2C]."""
    L, C = max_vision_token_length, hidden.shape[-1]
    vis = hidden[flag]
    assert vis.shape[0] % L == 0, "vision_prediction_mode='2d' assumes complete images"
    n = vis.shape[0] // L
    vis = vis.view(n, L, C)
    ph = sd["vision_hidden_placeholder"].to(hidden.dtype)
    amap = ph[None, None, None, :].repeat(n, res + 1, res + 1, 1)
    amap[:, 1, 0, :] = vis[:, 0, :]
    amap[:, 1:, 1:, :] = vis[:, 1:-1, :].reshape(n, res, res, C)
    grid = torch.cat([amap[:, :-1, 1:, :], amap[:, 1:, :-1, :]], -1).reshape(n, L - 2, 2 * C)
    phn = ph[None, None, :].expand(n, 1, C)
    to_eoi = torch.cat([vis[:, -2:-1, :], phn], -1)
    eoi = torch.cat([vis[:, -1:, :], phn], -1)
    return torch.cat([grid, to_eoi, eoi], 1).reshape(-1, 2 * C)


def vl_logits_2d(sd, hidden, flag, Q: int, max_vision_token_length: int, res: int):
    """cal_vl_logits with vision_prediction_mode="2d": text rows as in 1d; vision rows = head_q(cat(up, left)), heads [Vv, 2C]."""
    lang = F.linear(hidden, sd["lm_head.weight"])
    V = lang.shape[-1]
    feats = vision_features_2d(sd, hidden, flag, max_vision_token_length, res)
    outs = []
    for q in range(Q):
        vis = F.linear(feats, sd[f"vision_lm_head.heads.{q}.weight"])
        Vv = vis.shape[-1]
        full = torch.full(hidden.shape[:-1] + (V + Vv,), float("-inf"), dtype=lang.dtype)
        full[..., :V] = torch.where(flag.unsqueeze(-1), torch.full_like(lang, float("-inf")), lang)
        row_v = torch.cat([torch.full((vis.shape[0], V), float("-inf"), dtype=lang.dtype), vis], -1)
        full = full.clone()
        full[flag] = row_v
        outs.append(full)
    return torch.stack(outs)


# ---- KV-cache decode path ------------------------------------------------------------------------------
# LibraAttention.forward with past_key_value (modeling_libra.py:344-361), LibraForCausalLM.forward's cached branch
# (:1139-1144) and prepare_inputs_for_generation's conventions (:1190-1231): the per-layer cache is
# ([K_for_vision, K_for_language] (roped), V, V_bridge, vision_flag).  Here it is kept in the closed-form operands
# k_same = rope(k), k_cross = rope(k + kb), v, vb with  K_for_vision[j] = flag_j ? k_same : k_cross  and
# K_for_language[j] = flag_j ? k_cross : k_same;  `as_reference_cache` converts for comparison with the fixture.
# Pinned by tests/test_oracle_libra_golden.py against tests/golden/libra_tiny_decode.safetensors
# (tests/golden/make_golden_libra_decode.py runs the reference's own cached forward).

def attention_step(sd, pre, x, flag, cache: Optional[dict], position_ids, key_valid, heads: int, cos, sin, addition: bool = False):
    """x [B,q,H] = the NEW tokens (q = prompt length at prefill, 1 afterwards), flag [B,q], position_ids [B,q],
    key_valid [B, past + q] bool (the attention_mask).  -> (out [B,q,H], new cache)."""
    B, q_len, Hd = x.shape
    d = Hd // heads
    lin = lambda name: (lambda t: F.linear(t, sd[pre + name + ".weight"]))
    low = lambda name: (lambda t: libra_linear(t, sd, pre + name + "."))
    q = routed(x, flag, lin("q_proj"), low("vision_q_proj"), addition)
    k = routed(x, flag, lin("k_proj"), low("vision_k_proj"), addition)
    v = routed(x, flag, lin("v_proj"), low("vision_v_proj"), addition)
    if pre + "vision_k_bridge_on_language.weight_A" in sd:
        kb = routed(x, flag, low("vision_k_bridge_on_language"), low("vision_k_bridge_on_vision"))
        vb = routed(x, flag, low("vision_v_bridge_on_language"), low("vision_v_bridge_on_vision"))
    else:                                          # use_bridge = False
        kb, vb = torch.zeros_like(k), torch.zeros_like(v)
    hd = lambda t: t.view(B, q_len, heads, d).transpose(1, 2)
    q, k_same, k_cross, v, vb = hd(q), hd(k), hd(k + kb), hd(v), hd(vb)
    q = apply_rope(q, cos, sin, position_ids)
    k_same = apply_rope(k_same, cos, sin, position_ids)
    k_cross = apply_rope(k_cross, cos, sin, position_ids)
    new = dict(k_same=k_same, k_cross=k_cross, v=v, vb=vb, flag=flag, pos=position_ids)
    if cache is not None:
        new = {n: torch.cat([cache[n], new[n]], dim=(1 if n in ("flag", "pos") else 2)) for n in new}
    kf, kpos = new["flag"], new["pos"]                                       # [B, L]
    cross = (flag[:, :, None] != kf[:, None, :]).unsqueeze(1)                 # [B,1,q,L]: m_i != m_j
    s = torch.where(cross, q @ new["k_cross"].transpose(-1, -2), q @ new["k_same"].transpose(-1, -2)) / (d ** 0.5)
    mn = torch.finfo(s.dtype).min
    allowed = (kpos[:, None, :] <= position_ids[:, :, None]) & key_valid[:, None, :]      # causal by position + padding
    s = torch.max(s + torch.where(allowed, 0.0, mn).to(s.dtype).unsqueeze(1), torch.tensor(mn, dtype=s.dtype))
    p = torch.softmax(s, dim=-1, dtype=torch.float32 if s.dtype != torch.float64 else torch.float64).to(q.dtype)
    o = p @ new["v"] + (p * cross.to(p.dtype)) @ new["vb"]
    o = o.transpose(1, 2).reshape(B, q_len, Hd)
    return routed(o, flag, lin("o_proj"), low("vision_o_proj"), addition), new


def model_step(sd, input_ids, vision_indices, signal, caches: Optional[list], position_ids, key_valid, *, layers: int, heads: int,
               vocab: int, max_vision_token_length: int, eps: float = 1e-6, max_pos: int = 2048, addition: bool = False):
    """One cached forward over the NEW tokens input_ids [Q,B,q]: -> (hidden [B,q,H], flag [B,q], caches)."""
    flag = vision_indices < max_vision_token_length
    assert torch.equal(flag, input_ids[0] >= vocab)
    x = input_embeds(sd, input_ids, flag, signal, vocab, eps, vision_indices)   # signal None -> zeros (:646-653)
    cos, sin = rope_tables(x.shape[-1] // heads, max(max_pos, int(position_ids.max()) + 1), dtype=x.dtype)
    out_caches = []
    for i in range(layers):
        pre = f"model.layers.{i}."
        h = routed(x, flag, lambda t: rms_norm(t, sd[pre + "input_layernorm.weight"], eps),
                   lambda t: rms_norm(t, sd[pre + "vision_input_layernorm.weight"], eps))
        a, c = attention_step(sd, pre + "self_attn.", h, flag, None if caches is None else caches[i], position_ids, key_valid,
                              heads, cos, sin, addition)
        out_caches.append(c)
        x = x + a
        h = routed(x, flag, lambda t: rms_norm(t, sd[pre + "post_attention_layernorm.weight"], eps),
                   lambda t: rms_norm(t, sd[pre + "vision_post_attention_layernorm.weight"], eps))
        x = x + mlp(sd, pre + "mlp.", h, flag)
    x = routed(x, flag, lambda t: rms_norm(t, sd["model.norm.weight"], eps), lambda t: rms_norm(t, sd["model.vision_norm.weight"], eps))
    return x, flag, out_caches


def cached_logits(sd, hidden, flag, vision_indices, Q: int, *, had_past: bool, max_vision_token_length: int, newline_token_id: int):
    """cal_vl_logits + the cached branch's EOI rule: when the step's last input token is EOI (vision index L-1) its logits
    are replaced by "append a newline" (+inf at newline_token_id, -inf elsewhere; :1141-1144)."""
    z = vl_logits(sd, hidden, flag, Q)
    if had_past:
        eoi = vision_indices[:, -1] == max_vision_token_length - 1
        forced = torch.full((z.shape[-1],), float("-inf"), dtype=z.dtype)
        forced[newline_token_id] = float("inf")
        z[:, eoi, -1, :] = forced
    return z


def as_reference_cache(c: dict):
    """-> (K_for_vision, K_for_language, V, V_bridge, flag) in the reference's cache layout [B, heads, L, d]."""
    f = c["flag"][:, None, :, None]
    return (torch.where(f, c["k_same"], c["k_cross"]), torch.where(f, c["k_cross"], c["k_same"]), c["v"], c["vb"], c["flag"])


def causal_lm_loss(logits_q: torch.Tensor, labels_q: torch.Tensor) -> torch.Tensor:
    """mean over codebooks of shift-by-one CE with ignore_index -100 (:1160-1174)."""
    Q = logits_q.shape[0]
    loss = 0.0
    for q in range(Q):
        sl = logits_q[q][..., :-1, :].reshape(-1, logits_q.shape[-1])
        tl = labels_q[q][..., 1:].reshape(-1)
        loss = loss + F.cross_entropy(sl.float() if sl.dtype not in (torch.float64,) else sl, tl)
    return loss / Q


def get_labels(input_ids, attention_mask, label_mask_position_map, *, boi_token_id: int, bos_token_id: int):
    """LibraTrainWrapper.get_labels (:1397-1411)."""
    labels = input_ids.clone()
    labels[:, attention_mask == 0] = -100
    labels[labels == boi_token_id] = -100
    labels[labels == bos_token_id] = -100
    labels = labels.permute(1, 2, 0)
    for label, spans in zip(labels, label_mask_position_map):
        for (start, end) in spans:
            label[start:end] = -100
    return labels.permute(2, 0, 1)


def assemble_inputs(text_ids: torch.Tensor, attention_mask: torch.Tensor, image_ids: Optional[torch.Tensor],
                    encoder_feat: Optional[torch.Tensor], *, img_ph_token_id: int, img_gen_token_id: int,
                    boi_token_id: int, num_codebook: int, max_vision_token_length: int,
                    contiguous_ignore_signs: Optional[Sequence[bool]] = None, max_length: Optional[int] = None):
    """The tensor-assembly half of LibraTokenizer.forward (tokenization_libra.py:250-316), given the text
    tokenizer's ids (with <img_ph> placeholders already expanded to hw+2 slots per image) and the
    ImageTokenizer outputs.  -> input_ids [Q,B,S], attention_mask, vision_indices, coninous_signal [sic]."""
    ids = text_ids.clone()
    ph = ids == img_ph_token_id
    gen = ids == img_gen_token_id
    ids[gen] = boi_token_id
    ids = ids[None].repeat(num_codebook, 1, 1)
    has_images = image_ids is not None
    if has_images:
        ids[:, ph] = image_ids.flatten(1, 2)
    vi = torch.full(attention_mask.shape, max_vision_token_length, dtype=torch.long)
    signal = None
    if has_images:
        n_img, L = image_ids.shape[1], image_ids.shape[2]
        vi[ph] = torch.arange(L).expand(n_img, -1).flatten(0, 1)
        z = torch.zeros(encoder_feat.shape[0], 1, encoder_feat.shape[2], dtype=encoder_feat.dtype)
        cont = torch.cat([z, encoder_feat, z], dim=1)
        if contiguous_ignore_signs is not None:
            cont[torch.tensor(list(contiguous_ignore_signs), dtype=torch.bool)] = 0
        signal = torch.zeros(ids.shape[1], ids.shape[2], cont.shape[-1], dtype=cont.dtype)
        signal[ph] = cont.flatten(0, 1)
    else:
        vi[gen] = 0
    if max_length is not None:
        ids, attention_mask, vi = ids[:, :, :max_length], attention_mask[:, :max_length], vi[:, :max_length]
        if signal is not None:
            signal = signal[:, :max_length]
    return ids.contiguous(), attention_mask.contiguous(), vi.contiguous(), signal


def random_layer_state_dict(*, hidden: int = 4096, inter: int = 11008, rank: int = 8, down_ratio: int = 4, layer: int = 0,
                            seed: int = 5, dtype=torch.bfloat16, bridge_b_std: float = 0.3) -> Dict[str, torch.Tensor]:
    """Random weights of ONE LibraDecoderLayer under the reference's state-dict keys, fan-in scaled so that
    activations stay O(1); bridge weight_B is drawn non-zero (zero-initialised upstream, modeling_libra.py:184) so the
    bridge path is numerically live.  Used by the full-width parity tests and bench.py's CPU baseline."""
    g = torch.Generator().manual_seed(seed)
    H, I, r, rg = hidden, inter, hidden // down_ratio, inter // down_ratio
    sd: Dict[str, torch.Tensor] = {}

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(dtype)
    p = f"model.layers.{layer}."
    for n in ("q", "k", "v", "o"):
        sd[p + f"self_attn.{n}_proj.weight"] = rn(H, H, std=H ** -0.5)
        sd[p + f"self_attn.vision_{n}_proj.weight_A"] = rn(r, H, std=H ** -0.5)
        sd[p + f"self_attn.vision_{n}_proj.weight_B"] = rn(H, r, std=r ** -0.5)
    for kv in ("k", "v"):
        for w in ("language", "vision"):
            sd[p + f"self_attn.vision_{kv}_bridge_on_{w}.weight_A"] = rn(rank, H, std=H ** -0.5)
            sd[p + f"self_attn.vision_{kv}_bridge_on_{w}.weight_B"] = rn(H, rank, std=bridge_b_std)
    sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = rn(I, H, std=H ** -0.5), rn(I, H, std=H ** -0.5)
    sd[p + "mlp.down_proj.weight"] = rn(H, I, std=I ** -0.5)
    for n in ("gate", "up"):
        sd[p + f"mlp.vision_{n}_proj.weight_A"] = rn(rg, H, std=H ** -0.5)
        sd[p + f"mlp.vision_{n}_proj.weight_B"] = rn(I, rg, std=rg ** -0.5)
    sd[p + "mlp.vision_down_proj.weight_A"], sd[p + "mlp.vision_down_proj.weight_B"] = rn(r, I, std=I ** -0.5), rn(H, r, std=r ** -0.5)
    for n in ("input_layernorm", "post_attention_layernorm", "vision_input_layernorm", "vision_post_attention_layernorm"):
        sd[p + n + ".weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    return sd


# ======================================================================================================
# Generation loop: the reference's custom greedy_search over [Q,B,S] ids
# (/root/reference/libra/models/libra/modeling_libra_utils.py:61-328) with its ValidImageLogitsProcessor
# (/root/reference/libra/models/llama/modeling_llama_utils.py:23-76), restated on the cached step above.
# Pinned by tests/test_oracle_libra_golden.py against tests/golden/libra_tiny_generate.safetensors
# (tests/golden/make_golden_libra_generate.py runs the reference's own greedy_search, left-padded batch).
# ======================================================================================================
def valid_image_scores(ids_q: torch.Tensor, scores_q: torch.Tensor, *, valid_image_token_length: int, boi: int, eoi: int,
                       offset: int) -> torch.Tensor:
    """One codebook: ids_q [B,S], scores_q [B,V'].  n = trailing vision ids; 0 < n < len+1 -> only code ids; n == len+1 -> only EOI."""
    full = valid_image_token_length + 2
    n = (torch.cumsum(torch.flip(ids_q, dims=[-1]) < offset, dim=-1) == 0).sum(-1)
    if bool((n > full).any()):
        raise ValueError("You have generated an invalid image.")
    out = scores_q.clone()
    cols = torch.arange(scores_q.shape[-1])
    code = (cols >= offset) & (cols != boi) & (cols != eoi)
    body, close = (n > 0) & (n < full - 1), n == full - 1
    out[body[:, None] & ~code[None, :]] = float("-inf")
    out[close[:, None] & (cols != eoi)[None, :]] = float("-inf")
    return out


def greedy_generate(sd, input_ids, attention_mask, vision_indices, signal, *, steps: int, Q: int, layers: int, heads: int, vocab: int,
                    max_vision_token_length: int, newline_token_id: int, pad_token_id: int, eos_token_id: int, image_rule: dict,
                    eps: float = 1e-6, max_pos: int = 2048, addition: bool = False):
    """-> (sequences [Q,B,S+steps], processed scores [steps,Q,B,V']).  position_ids = attention_mask.cumsum(-1) - 1 (pads -> 1,
    modeling_libra.py:1204-1207); the vision index of a new token counts up inside an image and stays at L otherwise (:1270-1278);
    finished sequences emit pad, codebook by codebook (modeling_libra_utils.py:276-296)."""
    kw = dict(layers=layers, heads=heads, vocab=vocab, max_vision_token_length=max_vision_token_length, eps=eps, max_pos=max_pos, addition=addition)
    am, vi, ids = attention_mask.clone(), vision_indices.clone(), input_ids.clone()
    B = ids.shape[1]
    unfinished = torch.ones(B, dtype=torch.long)
    caches, all_scores = None, []
    for t in range(steps):
        pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
        if caches is None:
            hid, flag, caches = model_step(sd, ids, vi, signal, None, pos, am.bool(), **kw)
            z = cached_logits(sd, hid, flag, vi, Q, had_past=False, max_vision_token_length=max_vision_token_length,
                              newline_token_id=newline_token_id)
        else:
            hid, flag, caches = model_step(sd, ids[:, :, -1:], vi[:, -1:], None, caches, pos[:, -1:], am.bool(), **kw)
            z = cached_logits(sd, hid, flag, vi[:, -1:], Q, had_past=True, max_vision_token_length=max_vision_token_length,
                              newline_token_id=newline_token_id)
        sc = torch.stack([valid_image_scores(ids[q], z[q, :, -1, :], **image_rule) for q in range(Q)])
        all_scores.append(sc)
        nxt = sc.argmax(-1)
        cols = []
        for q in range(Q):
            tok = nxt[q] * unfinished + pad_token_id * (1 - unfinished)
            unfinished = unfinished * (tok != eos_token_id).long()
            cols.append(tok)
        ids = torch.cat([ids, torch.stack(cols)[:, :, None]], dim=-1)
        am = torch.cat([am, am.new_ones((B, 1))], dim=-1)
        vi = torch.cat([vi, (vi[:, -1] + 1).clamp(max=max_vision_token_length)[:, None]], dim=-1)
    return ids, torch.stack(all_scores)
