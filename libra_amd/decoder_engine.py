"""Forward schedule of Libra's routed ("bridge") decoder on the gfx950 kernels — SURVEY §8 rows a12-a21.

Routing without permutation passes: activations stay in sequence order [B*S, H]; the two modality streams
are index lists (``lang_idx``, ``vis_idx``) that the GEMM uses to gather its A rows and scatter its C rows
(libra_gemm_bf16_nt_routed), so `cal_language_vision`'s boolean-mask gather/scatter (≈10 per layer, each a
host sync in the reference: modeling_libra.py:111-147) costs ONE nonzero() per batch.

Rounding points follow the reference's bf16 execution (every Linear output, norm, RoPE term is bf16);
accumulation is fp32 in the MFMA.  Reference lines: see oracle/libra_oracle.py (same structure).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import kernels as K

BF16 = torch.bfloat16


@dataclass
class DecDims:
    hidden: int
    inter: int
    layers: int
    heads: int
    vocab: int
    vision_vocab: int
    codebooks: int
    max_vision_len: int
    signal: int
    rank: int = 8
    down_ratio: int = 4
    eps: float = 1e-6
    max_pos: int = 2048

    @property
    def r(self):
        return self.hidden // self.down_ratio

    @property
    def rg(self):
        return self.inter // self.down_ratio


def rope_tables(dim: int, n_pos: int, device, base: float = 10000.0):
    """LlamaRotaryEmbedding cache (models/llama/modeling_llama.py:135-164): fp32 tables, cast to bf16 on use."""
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16).to(device).contiguous(), emb.sin().to(BF16).to(device).contiguous()


def pack(sd: Dict[str, torch.Tensor], d: DecDims):
    """Fused / padded operand copies derived from the parameters (one per layer):
    [q;k;v] dense and low-rank-A stacks, [gate;up] stacks, rank-8 bridge A's padded to the GEMM's 64-wide K/N granule."""
    dev = sd["model.embed_tokens.weight"].device
    H = d.hidden
    out = []
    for i in range(d.layers):
        a = f"model.layers.{i}.self_attn."
        m = f"model.layers.{i}.mlp."
        cat = lambda names: torch.cat([sd[n].detach() for n in names], 0).contiguous()

        def bridge_a(which):
            t = torch.zeros((64, H), dtype=BF16, device=dev)
            t[0:d.rank] = sd[a + f"vision_k_bridge_on_{which}.weight_A"].detach()
            t[8:8 + d.rank] = sd[a + f"vision_v_bridge_on_{which}.weight_A"].detach()
            return t

        def bridge_b(name):
            t = torch.zeros((H, 8), dtype=BF16, device=dev)
            t[:, :d.rank] = sd[a + name + ".weight_B"].detach()
            return t
        out.append(dict(
            wqkv=cat([a + "q_proj.weight", a + "k_proj.weight", a + "v_proj.weight"]),
            aqkv=cat([a + "vision_q_proj.weight_A", a + "vision_k_proj.weight_A", a + "vision_v_proj.weight_A"]),
            ab_l=bridge_a("language"), ab_v=bridge_a("vision"),
            bk_l=bridge_b("vision_k_bridge_on_language"), bk_v=bridge_b("vision_k_bridge_on_vision"),
            bv_l=bridge_b("vision_v_bridge_on_language"), bv_v=bridge_b("vision_v_bridge_on_vision"),
            wgu=cat([m + "gate_proj.weight", m + "up_proj.weight"]),
            agu=cat([m + "vision_gate_proj.weight_A", m + "vision_up_proj.weight_A"]),
        ))
    return out


def route(vision_indices: torch.Tensor, attention_mask: torch.Tensor, d: DecDims):
    """-> flag uint8 [N], lang_idx / vis_idx int32, kv_len int32 [B].  One host sync (nonzero) per batch."""
    B, S = vision_indices.shape
    flag = (vision_indices < d.max_vision_len).reshape(-1)
    lang_idx = torch.nonzero(~flag).squeeze(1).to(torch.int32)
    vis_idx = torch.nonzero(flag).squeeze(1).to(torch.int32)
    am = attention_mask.to(torch.bool)
    lens = am.sum(1).to(torch.int32)
    ar = torch.arange(S, device=am.device)[None, :]
    if not bool(torch.equal(am, ar < lens[:, None])):
        raise NotImplementedError("the fused attention kernel handles right padding only (the reference pads right: "
                                  "tokenization_libra.py, padding='longest'); got a non-suffix attention_mask")
    return flag.to(torch.uint8).contiguous(), lang_idx.contiguous(), vis_idx.contiguous(), lens.contiguous()


def embed(sd, d: DecDims, input_ids, flag, lang_idx, vis_idx, signal):
    """get_inputs_embeds_from_multicodebook (modeling_libra.py:625-661) -> x [N, H]."""
    Q, B, S = input_ids.shape
    N, H = B * S, d.hidden
    dev = input_ids.device
    x = torch.empty((N, H), dtype=BF16, device=dev)
    ids0 = input_ids[0].reshape(-1)
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    if n_l:
        tmp = torch.empty((n_l, H), dtype=BF16, device=dev)
        K.gather_rows(sd["model.embed_tokens.weight"], ids0, 0, lang_idx, n_l, tmp, 0)
        x.index_copy_(0, lang_idx.long(), tmp)              # plumbing copy; language rows only
    if n_v:
        Cs = d.signal
        ve = torch.empty((n_v, H + Cs), dtype=BF16, device=dev)
        for q in range(Q):
            K.gather_rows(sd[f"model.vision_embed_tokens.{q}.weight"], input_ids[q].reshape(-1), d.vocab, vis_idx, n_v, ve,
                          q * (H // Q))
        if signal is not None:
            K.copy_rows(signal.reshape(N, Cs).to(BF16), vis_idx, n_v, ve, H)
        else:
            ve[:, H:].zero_()
        ven = K.rmsnorm_routed(ve, sd["model.vision_signal_norm.weight"], None, None, d.eps)
        K.gemm_nt(ven, sd["model.vision_contiguous_signal_processor.weight"], out=x, c_rows=vis_idx)
    return x


def layer_forward(sd, pk, i: int, d: DecDims, x, flag, lang_idx, vis_idx, lens, cos, sin, B: int, S: int):
    H, I, r, rg = d.hidden, d.inter, d.r, d.rg
    N = B * S
    dev = x.device
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    pre = f"model.layers.{i}."
    a, m = pre + "self_attn.", pre + "mlp."
    # ---- attention block
    h = K.rmsnorm_routed(x, sd[pre + "input_layernorm.weight"], sd[pre + "vision_input_layernorm.weight"], flag, d.eps)
    qkv = torch.empty((N, 3 * H), dtype=BF16, device=dev)
    tb = torch.empty((N, 64), dtype=BF16, device=dev)
    if n_l:
        K.gemm_nt(h, pk["wqkv"], out=qkv, a_rows=lang_idx, c_rows=lang_idx)
        K.gemm_nt(h, pk["ab_l"], out=tb, a_rows=lang_idx, c_rows=lang_idx)
    if n_v:
        t = K.gemm_nt(h, pk["aqkv"], a_rows=vis_idx)                                   # [n_v, 3r]
        for j, nm in enumerate(("q", "k", "v")):
            K.gemm_nt(t[:, j * r:(j + 1) * r], sd[a + f"vision_{nm}_proj.weight_B"], out=qkv[:, j * H:(j + 1) * H],
                      c_rows=vis_idx)
        K.gemm_nt(h, pk["ab_v"], out=tb, a_rows=vis_idx, c_rows=vis_idx)
    kc, vc = K.rope_bridge(qkv, tb, pk["bk_l"], pk["bk_v"], pk["bv_l"], pk["bv_v"], flag, cos, sin, S, d.heads)
    o, _ = K.bridge_attn_fwd(qkv[:, :H], qkv[:, H:2 * H], kc, qkv[:, 2 * H:], vc, flag, lens, B, S, d.heads,
                             (H // d.heads) ** -0.5)
    x_mid = torch.empty_like(x)
    if n_l:
        K.gemm_nt(o, sd[a + "o_proj.weight"], out=x_mid, a_rows=lang_idx, c_rows=lang_idx, resid=x)
    if n_v:
        t = K.gemm_nt(o, sd[a + "vision_o_proj.weight_A"], a_rows=vis_idx)
        K.gemm_nt(t, sd[a + "vision_o_proj.weight_B"], out=x_mid, c_rows=vis_idx, resid=x)
    # ---- MLP block
    h2 = K.rmsnorm_routed(x_mid, sd[pre + "post_attention_layernorm.weight"],
                          sd[pre + "vision_post_attention_layernorm.weight"], flag, d.eps)
    x_out = torch.empty_like(x)
    if n_l:
        gu = K.gemm_nt(h2, pk["wgu"], a_rows=lang_idx)                                 # [n_l, 2I]
        act = K.swiglu(gu[:, :I], gu[:, I:])
        K.gemm_nt(act, sd[m + "down_proj.weight"], out=x_out, c_rows=lang_idx, resid=x_mid)
    if n_v:
        tg = K.gemm_nt(h2, pk["agu"], a_rows=vis_idx)                                  # [n_v, 2 rg]
        guv = torch.empty((n_v, 2 * I), dtype=BF16, device=dev)
        K.gemm_nt(tg[:, :rg], sd[m + "vision_gate_proj.weight_B"], out=guv[:, :I])
        K.gemm_nt(tg[:, rg:], sd[m + "vision_up_proj.weight_B"], out=guv[:, I:])
        actv = K.swiglu(guv[:, :I], guv[:, I:])
        td = K.gemm_nt(actv, sd[m + "vision_down_proj.weight_A"])
        K.gemm_nt(td, sd[m + "vision_down_proj.weight_B"], out=x_out, c_rows=vis_idx, resid=x_mid)
    return x_out


def forward(sd, packed, d: DecDims, input_ids, attention_mask, vision_indices, signal, labels=None, *,
            want_hidden_states: bool = False):
    """-> dict(hidden [B,S,H], flag, lang_idx, vis_idx, z_lang [n_l,V], z_vis list of [n_v,Vv], loss or None)."""
    Q, B, S = input_ids.shape
    dev = input_ids.device
    flag, lang_idx, vis_idx, lens = route(vision_indices, attention_mask, d)
    if not torch.equal(flag.view(B, S).bool(), input_ids[0] >= d.vocab):
        raise AssertionError("Inconsistent input_ids and vision_flag")                 # modeling_libra.py:707-710
    cos, sin = rope_tables(d.hidden // d.heads, max(d.max_pos, S), dev)
    x = embed(sd, d, input_ids, flag, lang_idx, vis_idx, signal)
    hs = [x] if want_hidden_states else None
    for i in range(d.layers):
        x = layer_forward(sd, packed[i], i, d, x, flag, lang_idx, vis_idx, lens, cos, sin, B, S)
        if want_hidden_states:
            hs.append(x)
    hidden = K.rmsnorm_routed(x, sd["model.norm.weight"], sd["model.vision_norm.weight"], flag, d.eps)
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    z_lang = K.gemm_nt(hidden, sd["lm_head.weight"], a_rows=lang_idx) if n_l else None
    z_vis = []
    vv_ld = K.round_up(d.vision_vocab, 8)
    for q in range(Q):
        if n_v:
            buf = torch.empty((n_v, vv_ld), dtype=BF16, device=dev)
            z_vis.append(K.gemm_nt(hidden, sd[f"vision_lm_head.heads.{q}.weight"], a_rows=vis_idx, out=buf[:, :d.vision_vocab]))
        else:
            z_vis.append(None)
    loss = None
    if labels is not None:
        N = B * S
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        for q in range(Q):
            tgt = torch.full((B, S), -100, dtype=torch.int64, device=dev)
            tgt[:, :-1] = labels[q][:, 1:]                                             # shift so that tokens < n predict n
            tgt = tgt.reshape(N)
            tot = torch.zeros((), dtype=torch.float32, device=dev)
            if n_l:
                tot = tot + K.ce_rows(z_lang, tgt.index_select(0, lang_idx.long()).contiguous(), 0).sum()
            if n_v:
                tot = tot + K.ce_rows(z_vis[q], tgt.index_select(0, vis_idx.long()).contiguous(), d.vocab).sum()
            loss = loss + tot / (tgt >= 0).sum().clamp_min(1)
        loss = loss / Q
    return dict(hidden=hidden.view(B, S, d.hidden), flag=flag, lang_idx=lang_idx, vis_idx=vis_idx, z_lang=z_lang,
                z_vis=z_vis, loss=loss, hidden_states=hs)


def dense_logits(out, d: DecDims, B: int, S: int):
    """Materialise the reference's padded logits tensor [Q,B,S,V+Vv] (cal_vl_logits, modeling_libra.py:1018-1052):
    text rows = [lm_head | -inf], vision rows = [-inf | head_q].  Memory-heavy; built only when asked for."""
    Q = len(out["z_vis"])
    V, Vv = d.vocab, d.vision_vocab
    dev = out["hidden"].device
    res = torch.full((Q, B * S, V + Vv), float("-inf"), dtype=BF16, device=dev)
    for q in range(Q):
        if out["z_lang"] is not None:
            res[q, out["lang_idx"].long(), :V] = out["z_lang"]
        if out["z_vis"][q] is not None:
            res[q, out["vis_idx"].long(), V:] = out["z_vis"][q]
    return res.view(Q, B, S, V + Vv)
