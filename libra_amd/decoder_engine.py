"""Forward and backward schedule of Libra's routed ("bridge") decoder on the gfx950 kernels — SURVEY §8 rows a12-a21.

Routing without permutation passes: activations stay in sequence order [B*S, H]; the two modality streams
are index lists (``lang_idx``, ``vis_idx``) that the GEMM uses to gather its A rows and scatter its C rows
(libra_gemm_bf16_nt_routed), so `cal_language_vision`'s boolean-mask gather/scatter (≈10 per layer, each a
host sync in the reference: modeling_libra.py:111-147) costs ONE nonzero() per batch.

Rounding points follow the reference's bf16 execution (every Linear output, norm, RoPE term is bf16);
accumulation is fp32 in the MFMA.  Reference file:line citations: include/libra_hip.h and DESIGN.md.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import dp
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
    rope_2d: bool = False          # config.use_2d_rope (modeling_libra.py:43-49, :663-678)
    unified_head: bool = False     # config.unified_head (:1054-1064)
    pred_2d: bool = False          # config.vision_prediction_mode == "2d" (:942-1014)
    res: int = 0                   # config.image_feature_resolution (max_vision_len == res * res + 2)
    bridge: bool = True            # config.use_bridge (:258); False: no bridge parameters, K_cross = K_same, V_cross = V_same
    concat: bool = True            # config.concat_signals (:556-562); False: processor(signal) is ADDED to the embeddings (:753-754)
    norm_sig: bool = True          # config.norm_signals (with concat): RMSNorm over [codebook embeddings | signal] (:558, :641-644)
    vis_pos: bool = False          # config.use_vision_position_embedding (:564-566, :636-638)
    addition: bool = False         # config.addition_mode (cal_language_vision :111-127): the q / k / v / o language projections run on
                                   # EVERY row and the vision low-rank projections are added on the vision rows (MLP, norms, bridges stay routed)

    @property
    def r(self):
        return self.hidden // self.down_ratio

    @property
    def rg(self):
        return self.inter // self.down_ratio


_ROPE_TABLES: Dict[tuple, tuple] = {}


def rope_tables(dim: int, n_pos: int, device, base: float = 10000.0):
    """LlamaRotaryEmbedding cache (models/llama/modeling_llama.py:135-164): fp32 tables, cast to bf16 on use.
    Built once per (dim, n_pos, device): the generation loop asks for them at every step."""
    key = (dim, n_pos, str(device), base)
    if key not in _ROPE_TABLES:
        _ROPE_TABLES[key] = _rope_tables(dim, n_pos, device, base)
    return _ROPE_TABLES[key]


def rope_rows(d: "DecDims", n_tokens: int) -> int:
    """Rows of the cos / sin tables: every position a sequence of n_tokens can reach (2d positions run at most res ahead)."""
    return max(d.max_pos, n_tokens + (d.res + 2 if d.rope_2d else 0))


def _rope_tables(dim: int, n_pos: int, device, base: float):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16).to(device).contiguous(), emb.sin().to(BF16).to(device).contiguous()


class PackedOperands:
    """Fused / padded operand copies derived from the parameters, one dict per layer:
    [q;k;v | bridge A] dense and low-rank-A stacks, [gate;up] stacks, rank-8 bridge B's padded to 8 columns (+ transposes).

    The buffers are allocated once and REFRESHED IN PLACE (so hipGraphs captured against them stay valid and see new values).
    A slice is re-copied from its source parameter when the parameter is *volatile* (``requires_grad`` - an optimizer may
    have updated it through ``.data``, which bumps neither ``_version`` nor ``data_ptr``: DeepSpeed ZeRO, apex, any
    master-weight optimizer) or when its (data_ptr, _version) key changed (load_state_dict, .to()).  Version counters alone
    are never trusted for trainable parameters."""

    def __init__(self, sd: Dict[str, torch.Tensor], d: DecDims):
        dev = sd["model.embed_tokens.weight"].device
        H, r, rg, I = d.hidden, d.r, d.rg, d.inter
        self.device = dev
        self.layers: List[dict] = []
        self.arena = K.RowArena()               # zero-padded row buffers of this model's training passes (see _arows)
        self._slices: List[tuple] = []          # (dest view, source name, transposed?)
        self._keys: Dict[int, tuple] = {}
        z = lambda *shape: torch.zeros(shape, dtype=BF16, device=dev)
        for i in range(d.layers):
            a = f"model.layers.{i}.self_attn."
            m = f"model.layers.{i}.mlp."
            L = dict(wqkv_ab=z(3 * H + 64, H), aqkv_ab=z(3 * r + 64, H), wgu=z(2 * I, H), agu=z(2 * rg, H))
            for j, nm in enumerate(("q", "k", "v")):
                self._slices.append((L["wqkv_ab"][j * H:(j + 1) * H], a + f"{nm}_proj.weight", False))
                self._slices.append((L["aqkv_ab"][j * r:(j + 1) * r], a + f"vision_{nm}_proj.weight_A", False))
            # the rank-8 bridge A's ride along as 64 extra output rows of the q/k/v projection of their modality: one GEMM
            # produces [q | k | v | t_k t_v 0...] (and one dgrad / wgrad GEMM handles both in the backward)
            # use_bridge = False: the same operand layout with the bridge rows / matrices left at zero (kb = vb = 0, so the kernels
            # compute plain routed attention; the configuration is in neither recipe and is not separately tuned)
            for key, which, base in (("wqkv_ab", "language", 3 * H), ("aqkv_ab", "vision", 3 * r)):
                if d.bridge:
                    self._slices.append((L[key][base:base + d.rank], a + f"vision_k_bridge_on_{which}.weight_A", False))
                    self._slices.append((L[key][base + 8:base + 8 + d.rank], a + f"vision_v_bridge_on_{which}.weight_A", False))
            for kv in ("k", "v"):
                for which, tag in (("language", "l"), ("vision", "v")):
                    b, bT = z(H, 8), z(8, H)      # B [H, rank] and the same transposed [8, H]: the backward takes
                    L[f"b{kv}_{tag}"], L[f"b{kv}T_{tag}"] = b, bT       # B^T dkb / B^T dvb with v_dot2 over channel pairs
                    src = a + f"vision_{kv}_bridge_on_{which}.weight_B"
                    if d.bridge:
                        self._slices.append((b[:, :d.rank], src, False))
                        self._slices.append((bT[:d.rank], src, True))
            for j, nm in enumerate(("gate", "up")):
                self._slices.append((L["wgu"][j * I:(j + 1) * I], m + f"{nm}_proj.weight", False))
                self._slices.append((L["agu"][j * rg:(j + 1) * rg], m + f"vision_{nm}_proj.weight_A", False))
            self.layers.append(L)
        # unified_head (modeling_libra.py:1054-1064): per codebook the row-concatenated head [lm_head; head_q], its row count
        # zero-padded to the dgrad's 64 granule - one refreshed slice like the rest instead of a torch.cat per codebook in the
        # forward and a zeros + two copies in the backward (~1 GB of extra HBM traffic per step at the 11B shape)
        self.heads: List[torch.Tensor] = []
        if d.unified_head:
            V, Vv = d.vocab, d.vision_vocab
            for q in range(d.codebooks):
                wcat = z((V + Vv + 63) // 64 * 64, H)
                self._slices.append((wcat[:V], "lm_head.weight", False))
                self._slices.append((wcat[V:V + Vv], f"vision_lm_head.heads.{q}.weight", False))
                self.heads.append(wcat)
        self.refresh(sd, force=True)

    def refresh(self, sd: Dict[str, torch.Tensor], force: bool = False, volatile: bool = True) -> int:
        """Re-copy stale slices; returns the number of slices copied.  volatile=False: trust the (data_ptr, _version) key also for
        trainable parameters - only for calls between which no optimizer can have run (the decode steps of ONE generation,
        whose prefill did the full check: re-copying 10 GB of operands per generated token made a step 15 ms instead of 8.7)."""
        dsts, srcs = [], []
        with torch.no_grad():
            for k, (dst, name, tr) in enumerate(self._slices):
                p = sd[name]
                if p.data_ptr() == dst.data_ptr() and tr is False:
                    continue                                  # adopted (see adopt()): the parameter IS this slice
                key = (p.data_ptr(), p._version)
                if force or (volatile and p.requires_grad) or self._keys.get(k) != key:
                    src = p.detach()
                    dsts.append(dst)
                    srcs.append(src.t() if tr else src)
                    self._keys[k] = key
            if dsts:
                torch._foreach_copy_(dsts, srcs)          # a handful of multi-tensor launches instead of ~500 small copies / step
        return len(dsts)

    def adopt(self, sd: Dict[str, torch.Tensor]) -> int:
        """Make every parameter that fills exactly one whole-row slice of a fused operand a VIEW of that slice (`p.data = slice`, values
        preserved): q / k / v, the low-rank A's, the bridge A's, gate / up, the unified heads.  From then on an optimizer step writes
        the fused operand directly and refresh() has nothing to copy for them - at Libra-11B the per-forward refresh of the trainable
        vision A stacks was ~500 small copies = 2.2 GB read + written (2.6 ms of GPU time and as much host time per step), and
        the frozen text q|k|v / gate|up weights stop existing twice (13.5 GB).  Anything that re-points `.data` later (`.to()`, a
        flat-buffer optimizer such as dp.FlatAdamW) simply falls back to the copying path: refresh() compares data_ptr.  A parameter
        that ALREADY is a view of a larger storage when adopt() runs (the optimizer was built before the model's first forward)
        is left alone for the same reason: its owner's buffer is the truth.
        Called by the owning model (LibraForCausalLM._refresh_packed) - a PackedOperands built on somebody else's tensors copies.
        -> number of parameters adopted."""
        count: Dict[str, int] = {}
        for _, name, _ in self._slices:
            count[name] = count.get(name, 0) + 1
        n = 0
        with torch.no_grad():
            for dst, name, tr in self._slices:
                p = sd[name]
                if (tr is not False or count[name] != 1 or not isinstance(p, torch.nn.Parameter) or tuple(dst.shape) != tuple(p.shape)
                        or not dst.is_contiguous() or p.dtype != dst.dtype or p.device != dst.device or p.data_ptr() == dst.data_ptr()):
                    continue
                # A parameter that already views a LARGER storage belongs to somebody's flat buffer (dp.FlatAdamW binds `p.data`
                # to its `pf` buckets, ZeRO / FSDP style wrappers do the same): its owner keeps writing that buffer, so re-pointing
                # the parameter here would silently detach it from the optimizer.  It stays a copy refreshed per forward.
                # (dp.FlatAdamW also marks what it owns - a bucket of ONE parameter is exactly that parameter's size)
                if getattr(p, "_libra_flat_owner", None) is not None or p.untyped_storage().nbytes() != p.numel() * p.element_size():
                    continue
                dst.copy_(p.detach())
                p.data = dst
                n += 1
        return n

    def __getitem__(self, i: int) -> dict:
        return self.layers[i]

    def __len__(self) -> int:
        return len(self.layers)


def pack(sd: Dict[str, torch.Tensor], d: DecDims) -> PackedOperands:
    return PackedOperands(sd, d)


def route(vision_indices: torch.Tensor, attention_mask: torch.Tensor, d: DecDims, *, allow_left: bool = False):
    """-> flag uint8 [N], lang_idx / vis_idx int32, kv_len int32 [B] (end of the valid tokens), kv_start int32 [B] (first valid
    token; None unless some sequence is left-padded, which needs `allow_left`).  One host sync per batch.
    Training pads RIGHT (libra_pretrain.yaml:17 `padding_side: right`); batched generation pads LEFT (demo/libra_demo.ipynb sets
    `padding_side = 'left'`): the inference forward accepts one contiguous block of valid tokens per sequence."""
    B, S = vision_indices.shape
    flag = (vision_indices < d.max_vision_len).reshape(-1)
    lang_idx = torch.nonzero(~flag).squeeze(1).to(torch.int32)
    vis_idx = torch.nonzero(flag).squeeze(1).to(torch.int32)
    am = attention_mask.to(torch.bool)
    n = am.sum(1).to(torch.int32)
    ar = torch.arange(S, device=am.device)[None, :]
    first = torch.where(am.any(1), am.to(torch.int32).argmax(1), torch.zeros_like(n)).to(torch.int32)
    ends = first + n
    holes = (am != ((ar >= first[:, None]) & (ar < ends[:, None]))).any()
    code = int((holes.to(torch.int32) + 2 * (first > 0).any().to(torch.int32)).item())          # the one host read
    if code & 1:
        raise NotImplementedError("the fused attention kernels take one contiguous block of valid tokens per sequence (right "
                                  "padding when training, left or right padding at inference); got an attention_mask with holes")
    left = bool(code & 2)
    if left and not allow_left:
        raise NotImplementedError("left-padded batches are an inference-only path: the backward kernels assume right padding, "
                                  "as the training recipes pad (libra_pretrain.yaml:17 `padding_side: right`)")
    return (flag.to(torch.uint8).contiguous(), lang_idx.contiguous(), vis_idx.contiguous(), ends.contiguous(),
            first.contiguous() if left else None)


def embed(sd, d: DecDims, input_ids, flag, lang_idx, vis_idx, signal, sv=None, vision_indices=None):
    """get_inputs_embeds_from_multicodebook (modeling_libra.py:625-661) -> x [N, H].  The recipes' configuration (signal
    concatenated to the codebook embeddings, RMSNorm, one Linear) is the tuned path; `norm_sig` False skips the norm, `concat`
    False projects the signal on its own and adds it to every row (:753-754), `vis_pos` adds a learned embedding of the in-image
    index to the codebook embeddings (:636-638; `vision_indices` [B, S] int64 then needed)."""
    Q, B, S = input_ids.shape
    N, H = B * S, d.hidden
    dev = input_ids.device
    x = torch.empty((N, H), dtype=BF16, device=dev)
    ids0 = input_ids[0].reshape(-1)
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    save = sv is not None
    if n_l:
        tmp = torch.empty((n_l, H), dtype=BF16, device=dev)
        K.gather_rows(sd["model.embed_tokens.weight"], ids0, 0, lang_idx, n_l, tmp, 0)
        x.index_copy_(0, lang_idx.long(), tmp)              # plumbing copy; language rows only
    Cs = d.signal
    if n_v:
        # codebook embeddings [n_v, H]: straight into their final place unless the position embedding has to be added first
        ve = (K.alloc_rows(n_v, H + Cs, dev)[:n_v] if save and not d.norm_sig else torch.empty((n_v, H + Cs), dtype=BF16, device=dev)) \
            if d.concat else None
        direct = ve if (d.concat and not d.vis_pos) else torch.empty((n_v, H), dtype=BF16, device=dev)
        for q in range(Q):
            K.gather_rows(sd[f"model.vision_embed_tokens.{q}.weight"], input_ids[q].reshape(-1), d.vocab, vis_idx, n_v, direct,
                          q * (H // Q))
        if d.vis_pos:
            pe = torch.empty((n_v, H), dtype=BF16, device=dev)
            K.gather_rows(sd["model.vision_position_embedding.weight"], vision_indices.reshape(-1), 0, vis_idx, n_v, pe, 0)
            K.add_(direct, pe)                              # bf16 add, as the reference's `vision_concat + vision_position_embedding`
            if d.concat:
                K.copy_rows(direct, None, n_v, ve, 0)
        if d.concat:
            if signal is not None:
                K.copy_rows(signal.reshape(N, Cs).to(BF16), vis_idx, n_v, ve, H)
            else:
                ve[:, H:].zero_()
            rstd_e = None
            if d.norm_sig:
                ven = _rows(n_v, H + Cs, dev, save)             # (once per step: not worth an arena slot)
                _, rstd_e = K.rmsnorm_routed(ve, sd["model.vision_signal_norm.weight"], None, None, d.eps, out=ven, save_rstd=True)
            else:
                ven = ve
            K.gemm_nt(ven, sd["model.vision_contiguous_signal_processor.weight"], out=x, c_rows=vis_idx)
            if save:
                sv.update(ve=ve, ven=ven, rstd_e=rstd_e)
        else:
            x.index_copy_(0, vis_idx.long(), direct)        # plumbing copy; vision rows
    if not d.concat and signal is not None:                 # inputs_embeds + processor(contiguous_signal), every position (:753-754)
        sig2 = signal.reshape(N, Cs).to(BF16).contiguous()
        K.gemm_nt(sig2, sd["model.vision_contiguous_signal_processor.weight"], out=x, resid=x)
        if save:
            sv.update(sig_all=sig2)
    return x


_ARENA: Optional["K.RowArena"] = None      # the running pass's row arena (set by forward / backward from packed.arena), or None
_ARENA_PREFIX = ""                          # forward: "L<i>." (saved per layer) or "R." (recompute: one layer alive at a time)


def _arows(tag: str, n: int, c: int, dev) -> torch.Tensor:
    """[n, c] rows of a 64-row zero-padded buffer: from the pass's arena when there is one (no fill, no allocation), else fresh."""
    if _ARENA is not None:
        return _ARENA.rows(tag, n, c, dev)
    return K.alloc_rows(n, c, dev)[:n]


def _rows(n: int, c: int, dev, save: bool, tag: Optional[str] = None):
    """Compact per-modality buffer; when it will later be a reduction-major wgrad operand it needs zeroed pad rows."""
    if not save:
        return torch.empty((n, c), dtype=BF16, device=dev)
    return _arows(_ARENA_PREFIX + tag, n, c, dev) if tag is not None and _ARENA_PREFIX else K.alloc_rows(n, c, dev)[:n]


def positions_2d(vision_indices: torch.Tensor, d: DecDims, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LibraModel.get_2d_position_ids (modeling_libra.py:663-678) -> int32 [B, S, 2] (row, column): text and <img> tokens advance
    a running position by one, </img> by res + 1, a grid token sits at (running position) + (its row, its column) in 1..res;
    with `attention_mask` (generation, left padding) masked tokens do not advance it and sit at position 1."""
    L, res = d.max_vision_len, d.res
    vi = vision_indices.long()
    step = (vi == L) | (vi == 0)
    if attention_mask is not None:
        step = step & (attention_mask != 0)
    step = step.long()
    step = torch.where(vi == L - 1, torch.full_like(step, res + 1), step)
    run = step.cumsum(-1) - 1
    grid = (vi >= 1) & (vi <= L - 2)
    cell = (vi - 1).clamp_(0, max(res * res - 1, 0))
    row = torch.where(grid, cell // max(res, 1) + 1, torch.zeros_like(vi))
    col = torch.where(grid, cell % max(res, 1) + 1, torch.zeros_like(vi))
    pos = torch.stack((run + row, run + col), -1)
    if attention_mask is not None:
        pos = pos.masked_fill((attention_mask == 0).unsqueeze(-1), 1)
    return pos.to(torch.int32).contiguous()


def pred2d_sources(vision_indices_flat: torch.Tensor, vis_idx: torch.Tensor, d: DecDims, none_row: int):
    """vision_prediction_mode="2d" (cal_vision_logits_train, modeling_libra.py:942-1014) in closed form.  The head input of the
    vision row at flat position p whose in-image index is k (0 = <img>, 1..res^2 = grid cells in raster order, L-1 = </img>) is
    cat(hidden[src_a], hidden[src_b]); rows k < L-2 predict grid cell k = (i, j): src_a = the cell above (p + 1 - res, exists
    when i >= 1), src_b = the cell to the left (p itself, when j >= 1; <img> itself for cell (0,0)); rows L-2 and L-1:
    (p, none).  `none_row` = the row of the learned placeholder.  Every source is at or before p (causal), so the same
    formula serves truncated images and token-by-token generation (the reference pads to complete images instead, :944-966,
    :905-938).  -> (src_a, src_b) int32 [n_v]."""
    L, res = d.max_vision_len, d.res
    p = vis_idx.long()
    k = vision_indices_flat.long().index_select(0, p)
    none = torch.full_like(p, none_row)
    tail = k >= L - 2
    src_a = torch.where(tail, p, torch.where(k >= res, p + 1 - res, none))
    src_b = torch.where(tail, none, torch.where((k % res >= 1) | (k == 0), p, none))
    return src_a.to(torch.int32).contiguous(), src_b.to(torch.int32).contiguous()


class KVCache:
    """Per-layer key / value cache of the generation path (the reference's 4-tuple ([K_for_vision, K_for_language], V,
    V_bridge, vision_flag), modeling_libra.py:344-361), held as the four row buffers the kernels produce anyway:
    K_same = rope(k), K_cross = rope(k + kb), V_same = v, V_cross = v + vb, each [B, capacity, H] bf16, plus the modality
    flag [B, capacity] of every cached token.  `length` tokens are valid in every sequence (no padding inside the cache)."""

    def __init__(self, layers: int, B: int, capacity: int, H: int, device):
        self.B, self.capacity, self.length = B, capacity, 0
        self.layers = [tuple(torch.empty((B, capacity, H), dtype=BF16, device=device) for _ in range(4)) for _ in range(layers)]
        self.flag = torch.zeros((B, capacity), dtype=torch.uint8, device=device)
        self.start: Optional[torch.Tensor] = None          # int32 [B] first valid slot of each sequence (left-padded prompts)
        self.graphs: Dict[tuple, tuple] = {}               # routing pattern of a decode step -> (hipGraph, static buffers, outputs)
        self.pack_key = None                               # version of the packed weights the graphs were captured against
        self.sd = None                                     # the owner model's parameter dict at prefill (reused by its decode steps)
        self.run2d: Optional[torch.Tensor] = None          # use_2d_rope: int64 [B] running position (get_2d_position_ids' cumsum) at the last token
        self.hid: Optional[torch.Tensor] = None            # vision_prediction_mode="2d": final hidden state of every cached token [B, capacity, H]

    def get_seq_length(self) -> int:
        return self.length


# Round 6: a text GEMM and the vision GEMMs / weight gradients that do not depend on it go out as ONE multi-problem launch
# (K.gemm_multi -> libra_gemm_bf16_multi): the low-rank vision branch's 0.3 - 1.9-wave launches fill the text GEMM's last wave
# instead of each paying their own.  False = the round-5 schedule (one launch per GEMM / grouped launch), kept for same-process A/Bs
# (bench.py --no-multi) and as the schedule of addition_mode and of generation steps.
MULTI = True
# ... and the FIRST low-rank stage of a vision pair (x A^T: 76 tiles = 0.3 waves as a launch of its own) rides in the same launch as
# its second stage, which waits for it on the device (gemm_spec(reads=...)).  Off until measured on the chip.
CHAIN = False


def _multi_ok(d: "DecDims", n_l: int, n_v: int, dev=None) -> bool:
    # (K.gemm_multi_ok: one-time on-device acceptance check of the launch mechanism, cached per device)
    return MULTI and not d.addition and n_l > 16 and n_v > 0 and (dev is None or K.gemm_multi_ok(dev))


def layer_forward(sd, pk, i: int, d: DecDims, x, flag, lang_idx, vis_idx, lens, cos, sin, B: int, S: int, sv=None, *,
                  cache: Optional[KVCache] = None, positions: Optional[torch.Tensor] = None,
                  slot: Optional[torch.Tensor] = None, need_out: bool = True, kv_start: Optional[torch.Tensor] = None):
    """One LibraDecoderLayer (modeling_libra.py:437-491).  `sv` (dict) collects what the backward needs.
    `positions` (int32 [N]): explicit RoPE positions (left-padded prompts, decode steps); None = arange(S) per sequence.
    With `cache`: slot None = prefill (the S prompt tokens' K/V rows are stored at slots [0, S)); slot = one cached decode step
    (S == 1): the new rows appended at slot cache.length, attention of the single query over the cached tokens
    [kv_start, cache_len] (bridge_attn_decode); `slot` is the cache slot as a device tensor [1] int64 so that the step
    contains no host-side shape or index and can be captured in a hipGraph.  `kv_start` int32 [B]: first valid key."""
    H, I, r, rg = d.hidden, d.inter, d.r, d.rg
    N = B * S
    dev = x.device
    save = sv is not None
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    pre = f"model.layers.{i}."
    a, m = pre + "self_attn.", pre + "mlp."
    # ---- attention block
    h, rstd1 = K.rmsnorm_routed(x, sd[pre + "input_layernorm.weight"], sd[pre + "vision_input_layernorm.weight"], flag, d.eps,
                                save_rstd=True)
    qkvt = torch.empty((N, 3 * H + 64), dtype=BF16, device=dev)       # [q | k | v | bridge low-rank activations t_k t_v 0..]
    qkv, tb = qkvt[:, :3 * H], qkvt[:, 3 * H:]
    t = None
    multi = slot is None and _multi_ok(d, n_l, n_v, dev)         # (a cached decode step keeps its own, graph-captured schedule)
    G = K.gemm_spec
    if multi:
        # vision A stage first (the expansions need it); then text q|k|v (+ bridge A) and the three rank-r expansions together
        t_ext = _rows(n_v, 3 * r + 64, dev, save, "t_ext")                                                         # [n_v, 3r + 64]
        t = t_ext[:, :3 * r]
        pa = G(h, pk["aqkv_ab"], a_rows=vis_idx, out=t_ext) if CHAIN else None
        if not CHAIN:                                            # (a launch of its own goes through the library's tile planner)
            K.gemm_nt(h, pk["aqkv_ab"], a_rows=vis_idx, out=t_ext)
        K.gemm_multi(([pa] if CHAIN else []) + [G(h, pk["wqkv_ab"], out=qkvt, a_rows=lang_idx, c_rows=lang_idx)]
                     + [G(t[:, j * r:(j + 1) * r], sd[a + f"vision_{nm}_proj.weight_B"], out=qkv[:, j * H:(j + 1) * H], c_rows=vis_idx,
                          reads=pa if CHAIN else None) for j, nm in enumerate(("q", "k", "v"))])
        tb.index_copy_(0, vis_idx.long(), t_ext[:, 3 * r:])
    elif d.addition:
        K.gemm_nt(h, pk["wqkv_ab"], out=qkvt)                        # language projections (+ language bridge A) on every row
    elif n_l:
        K.gemm_nt(h, pk["wqkv_ab"], out=qkvt, a_rows=lang_idx, c_rows=lang_idx)
    if n_v and not multi:
        t_ext = K.gemm_nt(h, pk["aqkv_ab"], a_rows=vis_idx, out=_rows(n_v, 3 * r + 64, dev, save, "t_ext"))      # [n_v, 3r + 64]
        t = t_ext[:, :3 * r]
        tb.index_copy_(0, vis_idx.long(), t_ext[:, 3 * r:])          # 64 columns of the vision rows: plumbing copy (the bridges stay routed)
        if d.addition:                                               # language[vis] + vision (:126): accumulate on the vision rows
            for j, nm in enumerate(("q", "k", "v")):
                col = qkv[:, j * H:(j + 1) * H]
                K.gemm_nt(t[:, j * r:(j + 1) * r], sd[a + f"vision_{nm}_proj.weight_B"], out=col, c_rows=vis_idx, resid=col)
        else:
            # the three rank-r expansions share one launch (each alone is 1.2 waves of 256^2 tiles)
            K.gemm_nt_grouped([t[:, j * r:(j + 1) * r] for j in range(3)],
                              [sd[a + f"vision_{nm}_proj.weight_B"] for nm in ("q", "k", "v")],
                              [qkv[:, j * H:(j + 1) * H] for j in range(3)], c_rows=vis_idx)
    fused_append = cache is not None and slot is not None and positions is not None
    if positions is None:
        kc, vc = K.rope_bridge(qkv, tb, pk["bk_l"], pk["bk_v"], pk["bv_l"], pk["bv_v"], flag, cos, sin, S, d.heads)
    else:                                                   # (decode step: the new token's four cache rows go out of the same launch)
        kc, vc = K.rope_bridge_pos(qkv, tb, pk["bk_l"], pk["bk_v"], pk["bv_l"], pk["bv_v"], flag, cos, sin, positions, d.heads,
                                   append=(cache.layers[i], slot) if fused_append else None)
    if cache is not None:                                   # the cache slots
        new_rows = (qkv[:, H:2 * H], kc, qkv[:, 2 * H:], vc)
        if slot is None:
            for buf, rows in zip(cache.layers[i], new_rows):
                buf[:, :S].copy_(rows.view(B, S, H))         # prefill: plumbing copies
        elif not fused_append:
            K.kv_cache_append(new_rows, cache.layers[i], slot)
    if slot is None:
        o_lo = torch.empty((N, H), dtype=BF16, device=dev) if save else None   # rounding residual of o (the backward's D = dO.O)
        o, lse = K.bridge_attn_fwd(qkv[:, :H], qkv[:, H:2 * H], kc, qkv[:, 2 * H:], vc, flag, lens, B, S, d.heads,
                                   (H // d.heads) ** -0.5, need_lse=save, kv_start=kv_start, out_lo=o_lo)
    else:
        ks_c, kc_c, vs_c, vc_c = cache.layers[i]
        o_lo = None
        o, lse = K.bridge_attn_decode(qkv[:, :H], ks_c, kc_c, vs_c, vc_c, cache.flag, flag, lens, d.heads,
                                      (H // d.heads) ** -0.5, kv_start=kv_start), None
    x_mid = torch.empty_like(x)
    to = None
    if multi:
        to = _rows(n_v, r, dev, save, "to")
        pa = G(o, sd[a + "vision_o_proj.weight_A"], a_rows=vis_idx, out=to) if CHAIN else None
        if not CHAIN:
            K.gemm_nt(o, sd[a + "vision_o_proj.weight_A"], a_rows=vis_idx, out=to)
        K.gemm_multi(([pa] if CHAIN else []) + [G(o, sd[a + "o_proj.weight"], out=x_mid, a_rows=lang_idx, c_rows=lang_idx, resid=x),
                                                 G(to, sd[a + "vision_o_proj.weight_B"], out=x_mid, c_rows=vis_idx, resid=x,
                                                   reads=pa if CHAIN else None)])
    elif d.addition:
        K.gemm_nt(o, sd[a + "o_proj.weight"], out=x_mid, resid=x)    # every row
    elif n_l:
        K.gemm_nt(o, sd[a + "o_proj.weight"], out=x_mid, a_rows=lang_idx, c_rows=lang_idx, resid=x)
    if n_v and not multi:
        to = K.gemm_nt(o, sd[a + "vision_o_proj.weight_A"], a_rows=vis_idx, out=_rows(n_v, r, dev, save, "to"))
        K.gemm_nt(to, sd[a + "vision_o_proj.weight_B"], out=x_mid, c_rows=vis_idx, resid=x_mid if d.addition else x)
    # ---- MLP block
    h2, rstd2 = K.rmsnorm_routed(x_mid, sd[pre + "post_attention_layernorm.weight"],
                                 sd[pre + "vision_post_attention_layernorm.weight"], flag, d.eps, save_rstd=True)
    x_out = torch.empty_like(x) if need_out else None      # (a recompute pass stops before the last down projections)
    gu = act = tg = guv = actv = td = None
    if multi:
        tg = _rows(n_v, 2 * rg, dev, save, "tg")                                                               # [n_v, 2 rg]
        guv = torch.empty((n_v, 2 * I), dtype=BF16, device=dev)
        pa = G(h2, pk["agu"], a_rows=vis_idx, out=tg) if CHAIN else None
        if not CHAIN:
            K.gemm_nt(h2, pk["agu"], a_rows=vis_idx, out=tg)
        ptxt = G(h2, pk["wgu"], a_rows=lang_idx)                                                              # [n_l, 2I]
        K.gemm_multi(([pa] if CHAIN else []) + [ptxt,
                     G(tg[:, :rg], sd[m + "vision_gate_proj.weight_B"], out=guv[:, :I], reads=pa if CHAIN else None),
                     G(tg[:, rg:], sd[m + "vision_up_proj.weight_B"], out=guv[:, I:], reads=pa if CHAIN else None)])
        gu = ptxt.out
        act = K.swiglu(gu[:, :I], gu[:, I:], out=_rows(n_l, I, dev, save, "act"))
        actv = K.swiglu(guv[:, :I], guv[:, I:], out=_rows(n_v, I, dev, save, "actv"))
        td = _rows(n_v, r, dev, save, "td")
        pa = G(actv, sd[m + "vision_down_proj.weight_A"], out=td) if CHAIN and need_out else None
        if pa is None:
            K.gemm_nt(actv, sd[m + "vision_down_proj.weight_A"], out=td)
        if need_out:
            K.gemm_multi(([pa] if CHAIN else []) + [G(act, sd[m + "down_proj.weight"], out=x_out, c_rows=lang_idx, resid=x_mid),
                         G(td, sd[m + "vision_down_proj.weight_B"], out=x_out, c_rows=vis_idx, resid=x_mid, reads=pa if CHAIN else None)])
    elif n_l and n_l <= 16 and not save:                     # generation step: gate | up GEMM + SwiGLU as one weight-streaming launch
        act = K.gemm_swiglu_skinny(h2, pk["wgu"], a_rows=lang_idx)
    elif n_l:
        gu = K.gemm_nt(h2, pk["wgu"], a_rows=lang_idx)                                 # [n_l, 2I]
        act = K.swiglu(gu[:, :I], gu[:, I:], out=_rows(n_l, I, dev, save, "act"))
    if n_l and not multi:
        if need_out:
            K.gemm_nt(act, sd[m + "down_proj.weight"], out=x_out, c_rows=lang_idx, resid=x_mid)
    if n_v and not multi:
        tg = K.gemm_nt(h2, pk["agu"], a_rows=vis_idx, out=_rows(n_v, 2 * rg, dev, save, "tg"))               # [n_v, 2 rg]
        guv = torch.empty((n_v, 2 * I), dtype=BF16, device=dev)
        K.gemm_nt_grouped([tg[:, :rg], tg[:, rg:]], [sd[m + "vision_gate_proj.weight_B"], sd[m + "vision_up_proj.weight_B"]],
                          [guv[:, :I], guv[:, I:]])
        actv = K.swiglu(guv[:, :I], guv[:, I:], out=_rows(n_v, I, dev, save, "actv"))
        td = K.gemm_nt(actv, sd[m + "vision_down_proj.weight_A"], out=_rows(n_v, r, dev, save, "td"))
        if need_out:
            K.gemm_nt(td, sd[m + "vision_down_proj.weight_B"], out=x_out, c_rows=vis_idx, resid=x_mid)
    if save:
        sv.update(x=x, rstd1=rstd1, h=h, qkv=qkv, tb=tb, kc=kc, vc=vc, o=o, o_lo=o_lo, lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2, gu=gu,
                  act=act, t=t, to=to, tg=tg, guv=guv, actv=actv, td=td)
    return x_out


def check_ids(input_ids, flag_bs, d: DecDims, also=()):
    """One fused device-side validation + ONE host read: vision_flag == (ids[0] >= V) (modeling_libra.py:707-710), text ids
    inside the text table and every codebook's vision id inside the vision table (nn.Embedding device-asserts upstream;
    gather_rows itself does not bound-check).  `also`: 0-d integer tensors that ride along on the same read -> their values
    (the loss's target counts: reading them here keeps backward() free of a host synchronisation)."""
    if input_ids.dtype != torch.int64:
        raise TypeError(f"input_ids must be int64 (torch.long), got {input_ids.dtype}")
    V, Vv = d.vocab, d.vision_vocab
    ids0 = input_ids[0]
    bad_flag = (flag_bs != (ids0 >= V)).any()
    bad_text = ((ids0 < 0) & ~flag_bs).any()
    # (no boolean-mask gather: `input_ids[:, flag_bs]` is a hidden nonzero() + host sync)
    bad_vis = (flag_bs[None] & ((input_ids < V) | (input_ids >= V + Vv))).any()
    code = bad_flag.to(torch.int64) + 2 * bad_text.to(torch.int64) + 4 * bad_vis.to(torch.int64)
    code, *rest = torch.stack([code, *[a.to(torch.int64) for a in also]]).tolist()
    if code & 1:
        raise AssertionError("Inconsistent input_ids and vision_flag")                 # modeling_libra.py:707-710
    if code & 2:
        raise IndexError("negative token id in input_ids")
    if code & 4:
        raise IndexError(f"a vision token id lies outside [{V}, {V + Vv}) in one of the codebooks")
    return rest


def heads_forward(sd, d: DecDims, hidden, flag, lang_idx, vis_idx, Q: int, *, unified: bool = False, feats=None, packed=None):
    """cal_vl_logits (modeling_libra.py:1018-1064) without the -inf padding: -> (z_lang [n_l, V] or None, z_vis list of
    [n_v, Vv] or None, z_all list of [N, V+Vv] or None).
    default: text rows through lm_head, vision rows through head_q.
    unified (unified_head, uncached forward :1054-1064): EVERY row gets [lm_head | head_q] - one GEMM per codebook against the
    row-concatenated weight.
    feats [n_v, 2H] (vision_prediction_mode "2d"): the vision rows' head input cat(up, left); head_q is [Vv, 2H]."""
    dev = hidden.device
    N, H = hidden.shape
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    if unified:
        z_all = []
        ld = K.round_up(d.vocab + d.vision_vocab, 8)
        for q in range(Q):
            wcat = packed.heads[q][:d.vocab + d.vision_vocab]           # [lm_head; head_q], refreshed in place with the operands
            buf = torch.empty((N, ld), dtype=BF16, device=dev)
            z_all.append(K.gemm_nt(hidden, wcat, out=buf[:, :d.vocab + d.vision_vocab]))
        return None, [None] * Q, z_all
    z_lang = None
    if n_l:                                                   # (row stride padded to 8 elements: any vocabulary size)
        zl = torch.empty((n_l, K.round_up(d.vocab, 8)), dtype=BF16, device=dev)
        z_lang = K.gemm_nt(hidden, sd["lm_head.weight"], a_rows=lang_idx, out=zl[:, :d.vocab])
    z_vis = []
    vv_ld = K.round_up(d.vision_vocab, 8)
    for q in range(Q):
        if n_v:
            buf = torch.empty((n_v, vv_ld), dtype=BF16, device=dev)
            wq = sd[f"vision_lm_head.heads.{q}.weight"]
            if feats is not None:
                z_vis.append(K.gemm_nt(feats, wq, out=buf[:, :d.vision_vocab]))
            else:
                z_vis.append(K.gemm_nt(hidden, wq, a_rows=vis_idx, out=buf[:, :d.vision_vocab]))
        else:
            z_vis.append(None)
    return z_lang, z_vis, None


def forward(sd, packed, d: DecDims, input_ids, attention_mask, vision_indices, signal, labels=None, *,
            want_hidden_states: bool = False, save: bool = False, cache: Optional[KVCache] = None,
            recompute: bool = False, inputs_embeds: Optional[torch.Tensor] = None):
    """-> dict(hidden [B,S,H], flag, lang_idx, vis_idx, z_lang [n_l,V], z_vis list of [n_v,Vv], loss or None, saved).
    `inputs_embeds` [B,S,H] with input_ids None (LibraModel.forward, modeling_libra.py:703-716, :748-754): the decoder runs on the
    given embeddings - no table lookup, no signal processing (upstream applies both only when it computes the embeddings itself);
    backward() then leaves the gradient w.r.t. them in out["d_inputs_embeds"].
    `cache` (empty KVCache): prefill - every layer's K/V rows of the prompt are stored for decode_step.
    `recompute` (with save): gradient checkpointing per decoder layer (modeling_libra.py:787-797) - only each layer's
    INPUT is kept; backward() re-runs layer_forward before layer_backward (4.3 GB instead of 61 GB at B=8, S=2048)."""
    if (input_ids is None) == (inputs_embeds is None):
        raise ValueError("exactly one of input_ids and inputs_embeds")
    if input_ids is not None:
        Q, B, S = input_ids.shape
        dev = input_ids.device
    else:
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[2] != d.hidden or tuple(inputs_embeds.shape[:2]) != tuple(vision_indices.shape):
            raise ValueError(f"inputs_embeds must be [B, S, {d.hidden}] matching vision_indices {tuple(vision_indices.shape)}")
        if cache is not None:
            raise NotImplementedError("inputs_embeds with a KV cache (generation feeds token ids)")
        Q, (B, S), dev = d.codebooks, inputs_embeds.shape[:2], inputs_embeds.device
    if labels is not None and labels.dtype != torch.int64:
        raise TypeError(f"labels must be int64 (torch.long), got {labels.dtype}")
    if save:
        K.errors.check(dev)                     # the previous backward's device error word (copied back asynchronously): raises if set
    flag, lang_idx, vis_idx, lens, starts = route(vision_indices, attention_mask, d, allow_left=not save)
    positions = None
    if d.rope_2d:                               # (row, column) positions; the mask only matters for left-padded generation prompts
        pos2 = positions_2d(vision_indices, d, attention_mask if starts is not None else None)
        positions = pos2.view(-1, 2)
    elif starts is not None:                    # left padding: position_ids = attention_mask.cumsum(-1) - 1, pads -> 1 (:1204-1207)
        am = attention_mask.to(torch.long)
        positions = (am.cumsum(-1) - 1).masked_fill(am == 0, 1).reshape(-1).to(torch.int32).contiguous()
    if cache is not None:
        if cache.length != 0 or S > cache.capacity or B != cache.B:
            raise ValueError("prefill needs an empty KV cache of the batch size with capacity >= the prompt length")
        if not bool((lens == S).all()):
            raise NotImplementedError("cached generation continues every sequence at slot S: pad the prompts on the LEFT "
                                      "(tokenizer.padding_side = 'left', as the reference demo does); got right padding")
        cache.start = starts
        if d.rope_2d:                           # the running position after the prompt's last token: decode steps advance it
            vl = vision_indices[:, -1].long()
            cell = (vl - 1).clamp(0, max(d.res * d.res - 1, 0))
            row = torch.where((vl >= 1) & (vl <= d.max_vision_len - 2), cell // d.res + 1, torch.zeros_like(vl))
            cache.run2d = pos2[:, -1, 0].long() - row
    # the loss's per-codebook target counts (shifted labels >= 0) are known before the first layer: read with the id check
    cnts = [(labels[q][:, 1:] >= 0).sum().clamp_min(1) for q in range(Q)] if labels is not None else []
    if input_ids is not None:
        counts = check_ids(input_ids, flag.view(B, S).bool(), d, also=cnts)
    else:
        counts = torch.stack(cnts).tolist() if cnts else []                        # (no ids to validate: the read carries the counts only)
    cos, sin = rope_tables(d.hidden // d.heads, rope_rows(d, S), dev)
    saved = dict(layers=[], emb={}, recompute=bool(recompute)) if save else None
    if input_ids is not None:
        x = embed(sd, d, input_ids, flag, lang_idx, vis_idx, signal, saved["emb"] if save else None, vision_indices=vision_indices)
    else:
        x = inputs_embeds.detach().reshape(B * S, d.hidden).to(BF16).contiguous()
    hs = [x] if want_hidden_states else None
    # per-layer saved row buffers come from the model's arena unless an earlier saved forward still waits for its backward
    global _ARENA, _ARENA_PREFIX
    arena = getattr(packed, "arena", None) if save and not recompute else None
    if arena is not None and arena.busy:            # an earlier saved forward is still alive: fresh buffers for this one
        arena.warn_busy()
        arena = None
    try:
        _ARENA = arena
        for i in range(d.layers):
            sv = {} if save and not recompute else None
            x_in = x
            _ARENA_PREFIX = f"L{i}." if arena is not None else ""
            x = layer_forward(sd, packed[i], i, d, x, flag, lang_idx, vis_idx, lens, cos, sin, B, S, sv, cache=cache,
                              positions=positions, kv_start=starts)
            if save:
                saved["layers"].append(sv if not recompute else {"x": x_in})
            if want_hidden_states:
                hs.append(x)
    finally:
        _ARENA, _ARENA_PREFIX = None, ""
    if arena is not None:                           # the lease lives (and dies) with this forward's saved state
        saved["arena_lease"] = arena.lease()
    if cache is not None:
        cache.flag[:, :S] = flag.view(B, S)
        cache.length = S
    hidden, rstd_f = K.rmsnorm_routed(x, sd["model.norm.weight"], sd["model.vision_norm.weight"], flag, d.eps, save_rstd=True)
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    N = B * S
    unified = d.unified_head and cache is None           # with use_cache the reference masks the foreign half: the routed heads
    src2d = feats = None
    if d.pred_2d:
        if cache is not None:
            cache.hid = torch.zeros((B, cache.capacity, d.hidden), dtype=BF16, device=dev)
            cache.hid[:, :S] = hidden.view(B, S, d.hidden)
        if n_v:
            hidden_src = torch.cat([hidden, sd["vision_hidden_placeholder"].to(BF16).view(1, -1)], 0)      # row N = the placeholder
            src2d = pred2d_sources(vision_indices.reshape(-1), vis_idx, d, N)
            feats = K.alloc_rows(n_v, 2 * d.hidden, dev)[:n_v]          # 64-row padded: the head wgrad's reduction operand
            K.copy_rows(hidden_src, src2d[0], n_v, feats, 0)
            K.copy_rows(hidden_src, src2d[1], n_v, feats, d.hidden)
    z_lang, z_vis, z_all = heads_forward(sd, d, hidden, flag, lang_idx, vis_idx, Q, unified=unified, feats=feats, packed=packed)
    loss = None
    tgts = []
    if labels is not None:
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        for q in range(Q):
            tgt = torch.full((B, S), -100, dtype=torch.int64, device=dev)
            tgt[:, :-1] = labels[q][:, 1:]                                             # shift so that tokens < n predict n
            tgt = tgt.reshape(N).contiguous()
            tl = tgt.index_select(0, lang_idx.long()).contiguous() if n_l else None
            tv = tgt.index_select(0, vis_idx.long()).contiguous() if n_v else None
            tot = torch.zeros((), dtype=torch.float32, device=dev)
            if unified:
                tl = tgt                                                               # one softmax over [V | Vv] on every row
                tot = tot + K.ce_rows(z_all[q], tgt, 0).sum()
            else:
                if n_l:
                    tot = tot + K.ce_rows(z_lang, tl, 0).sum()
                if n_v:
                    tot = tot + K.ce_rows(z_vis[q], tv, d.vocab).sum()
            loss = loss + tot / counts[q]
            tgts.append((tl, tv))
        loss = loss / Q
    if save:
        saved.update(x_last=x, rstd_f=rstd_f, hidden=hidden, tgts=tgts, counts=counts, cos=cos, sin=sin, lens=lens, B=B, S=S,
                     Q=Q, input_ids=input_ids, positions=positions, unified=unified, src2d=src2d, feats=feats,
                     vision_indices=vision_indices)
    return dict(hidden=hidden.view(B, S, d.hidden), flag=flag, lang_idx=lang_idx, vis_idx=vis_idx, z_lang=z_lang,
                z_vis=z_vis, z_all=z_all, loss=loss, hidden_states=hs, saved=saved)


def _decode_core(sd, packed, d: DecDims, cache: KVCache, st: dict):
    """The device work of one cached step on static buffers `st` (ids [Q,B,1], flag [B], lang_idx, vis_idx, positions [B]
    int32, slot [1] int64, kv_len [B] int32): no host synchronisation, no host-side index - capturable."""
    Q, B = st["ids"].shape[0], st["ids"].shape[1]
    flag, lang_idx, vis_idx = st["flag"], st["lang_idx"], st["vis_idx"]
    dev = flag.device
    cos, sin = rope_tables(d.hidden // d.heads, rope_rows(d, cache.capacity), dev)
    cache.flag.index_copy_(1, st["slot"], flag.view(B, 1))
    x = embed(sd, d, st["ids"], flag, lang_idx, vis_idx, None, vision_indices=st["vi"].view(B, 1))
    for i in range(d.layers):
        x = layer_forward(sd, packed[i], i, d, x, flag, lang_idx, vis_idx, st["kv_len"], cos, sin, B, 1, None, cache=cache,
                          positions=st["positions"], slot=st["slot"], kv_start=cache.start)
    hidden, _ = K.rmsnorm_routed(x, sd["model.norm.weight"], sd["model.vision_norm.weight"], flag, d.eps, save_rstd=True)
    feats = None
    if d.pred_2d:                                           # cal_vision_logits_inference (:905-938) through pred2d_sources' formula
        Hd, L, res = d.hidden, d.max_vision_len, d.res
        cache.hid.index_copy_(1, st["slot"], hidden.view(B, 1, Hd))
        if vis_idx.numel():
            rows = vis_idx.long()
            k = st["vi"].index_select(0, rows)
            own = hidden.index_select(0, rows)
            up = cache.hid[rows, (st["slot"] + 1 - res).clamp_min(0).expand(rows.numel())]      # the cell above: res - 1 tokens back
            ph = sd["vision_hidden_placeholder"].to(BF16).view(1, Hd).expand_as(own)
            tail = (k >= L - 2).unsqueeze(1)
            a = torch.where(tail, own, torch.where((k >= res).unsqueeze(1), up, ph))
            b = torch.where(tail, ph, torch.where(((k % res >= 1) | (k == 0)).unsqueeze(1), own, ph))
            feats = torch.cat([a, b], 1)
    z_lang, z_vis, _ = heads_forward(sd, d, hidden, flag, lang_idx, vis_idx, Q, feats=feats)
    return dict(hidden=hidden.view(B, 1, d.hidden), flag=flag, lang_idx=lang_idx, vis_idx=vis_idx, z_lang=z_lang, z_vis=z_vis,
                z_all=None, loss=None, hidden_states=None, saved=None)


MAX_DECODE_GRAPHS = 8       # routing patterns kept per cache (in practice 2: "all text" and "all inside an image")


@torch.no_grad()
def decode_step(sd, packed, d: DecDims, cache: KVCache, input_ids, vision_indices, position_ids, *, use_graph: bool = True):
    """One cached generation step (LibraForCausalLM.forward with past_key_values, modeling_libra.py:1118-1144): input_ids
    [Q,B,1] are the NEW tokens, position_ids [B] their RoPE positions ([B,2] (row, column) with use_2d_rope; None = continue from
    the cache's running 2d position); decoded vision tokens have no encoder signal
    (prepare_inputs_for_generation, :1216-1218 -> zeros, :646-653).  Appends to `cache`.  Same return dict as forward().

    A step is ~40 small launches per layer on B rows - launch-bound by two orders of magnitude against its HBM floor - so it
    is captured ONCE per routing pattern (which of the B new tokens are vision tokens) into a hipGraph over static buffers and
    replayed; the pattern is the only host read of the step.  The outputs live in the graph's buffers: consume them before the
    next step."""
    Q, B, S = input_ids.shape
    if S != 1 or B != cache.B:
        raise ValueError("decode_step takes exactly one new token per cached sequence")
    if cache.length == 0 or cache.length >= cache.capacity:
        raise ValueError("decode_step needs a prefilled KV cache with a free slot")
    dev = input_ids.device
    flagb = (vision_indices < d.max_vision_len).reshape(-1)
    tok = input_ids[:, :, 0]                                                            # [Q, B]
    bad = (flagb & ((tok < d.vocab) | (tok >= d.vocab + d.vision_vocab)).any(0)) | (~flagb & (tok[0] < 0))
    pattern = tuple(torch.stack([flagb, tok[0] >= d.vocab, bad]).tolist())               # the one host read
    if pattern[0] != pattern[1]:
        raise AssertionError("Inconsistent input_ids and vision_flag")                 # modeling_libra.py:707-710
    if any(pattern[2]):
        raise IndexError("a token id lies outside its embedding table (vision ids: every codebook in [V, V + Vv))")
    key = tuple(pattern[0])
    entry = cache.graphs.get(key) if use_graph else None
    if entry is None:
        st = dict(ids=input_ids.clone(), flag=flagb.to(torch.uint8).contiguous(),
                  lang_idx=torch.nonzero(~flagb).squeeze(1).to(torch.int32).contiguous(),
                  vis_idx=torch.nonzero(flagb).squeeze(1).to(torch.int32).contiguous(),
                  positions=torch.empty((B, 2) if d.rope_2d else (B,), dtype=torch.int32, device=dev),
                  slot=torch.empty(1, dtype=torch.int64, device=dev), kv_len=torch.empty(B, dtype=torch.int32, device=dev),
                  vi=torch.empty(B, dtype=torch.int64, device=dev))
    else:
        graph, st, out = entry
        st["ids"].copy_(input_ids)
    vi_new = vision_indices.reshape(B).long()
    st["vi"].copy_(vi_new)
    if d.rope_2d:                            # get_2d_position_ids (:663-678) one token on: advance the running position
        L, res = d.max_vision_len, d.res
        step = ((vi_new == L) | (vi_new == 0)).long()
        cache.run2d = cache.run2d + torch.where(vi_new == L - 1, torch.full_like(step, res + 1), step)
        if position_ids is None:
            grid = (vi_new >= 1) & (vi_new <= L - 2)
            cell = (vi_new - 1).clamp(0, res * res - 1)
            zero = torch.zeros_like(vi_new)
            position_ids = torch.stack((cache.run2d + torch.where(grid, cell // res + 1, zero),
                                        cache.run2d + torch.where(grid, cell % res + 1, zero)), -1)
        st["positions"].copy_(position_ids.reshape(B, 2))
    else:
        st["positions"].copy_(position_ids.reshape(B))
    st["slot"].fill_(cache.length)
    st["kv_len"].fill_(cache.length + 1)
    if not use_graph:
        out = _decode_core(sd, packed, d, cache, st)
    elif entry is None:
        side = torch.cuda.Stream(device=dev)                                            # warm-up off the capture (first-use
        side.wait_stream(torch.cuda.current_stream(dev))                                # kernel attributes, allocator pools);
        with torch.cuda.stream(side):                                                   # it writes the same slot the replay does
            _decode_core(sd, packed, d, cache, st)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = _decode_core(sd, packed, d, cache, st)
        while len(cache.graphs) >= MAX_DECODE_GRAPHS:               # bounded: each graph pins its static buffers and outputs
            cache.graphs.pop(next(iter(cache.graphs)))              # (dicts keep insertion order: drop the oldest pattern)
        cache.graphs[key] = (graph, st, out)
        graph.replay()
    else:
        graph.replay()
    cache.length += 1
    return out


def dense_logits(out, d: DecDims, B: int, S: int):
    """Materialise the reference's padded logits tensor [Q,B,S,V+Vv] (cal_vl_logits, modeling_libra.py:1018-1052):
    text rows = [lm_head | -inf], vision rows = [-inf | head_q].  Memory-heavy; built only when asked for."""
    Q = len(out["z_vis"])
    V, Vv = d.vocab, d.vision_vocab
    dev = out["hidden"].device
    if out.get("z_all") is not None:                         # unified head, uncached: every row is dense already (:1054-1064)
        return torch.stack([z.contiguous() for z in out["z_all"]]).view(Q, B, S, V + Vv)
    res = torch.full((Q, B * S, V + Vv), float("-inf"), dtype=BF16, device=dev)
    for q in range(Q):
        if out["z_lang"] is not None:
            res[q, out["lang_idx"].long(), :V] = out["z_lang"]
        if out["z_vis"][q] is not None:
            res[q, out["vis_idx"].long(), V:] = out["z_vis"][q]
    return res.view(Q, B, S, V + Vv)


# ======================================================================================================
# Backward schedule (hand-written; replaces autograd through modeling_libra.py:437-491, :625-661, :1018-1174)
# ======================================================================================================
def _full(t: torch.Tensor) -> torch.Tensor:
    """The zero-padded [round_up(rows,64), C] allocation behind a K.alloc_rows row view (column slices allowed)."""
    return torch.as_strided(t, (K.round_up(t.shape[0], 64), t.shape[1]), t.stride(), t.storage_offset())


def _compact(t2d: torch.Tensor, idx: torch.Tensor, tag: Optional[str] = None) -> torch.Tensor:
    """Rows `idx` of a sequence-order tensor, gathered into a 64-row-padded compact buffer (reduction-major wgrad operand).
    `tag`: the backward-arena slot of this temporary (layer_backward's compacts: one slot serves every layer)."""
    n = idx.numel()
    buf = _arows("b." + tag, n, t2d.shape[1], t2d.device) if tag is not None else K.alloc_rows(n, t2d.shape[1], t2d.device)[:n]
    K.copy_rows(t2d, idx, n, buf, 0)
    return buf


def _padded(t2d: torch.Tensor, tag: str) -> torch.Tensor:
    """A sequence-order tensor as a reduction-major wgrad operand over ALL its rows: itself when its row count is a multiple of
    64 (every training shape), otherwise a zero-padded arena copy."""
    n = t2d.shape[0]
    if n % 64 == 0:
        return t2d
    buf = _arows("b." + tag, n, t2d.shape[1], t2d.device)
    buf.copy_(t2d)
    return buf


def _wg(dy_c: torch.Tensor, x_c: torch.Tensor, post=None, name: Optional[str] = None):
    """dW[out, in] = sum_tokens dy[t, out] x[t, in]; both operands token-major, padded (LIBRA_GEMM_A_T | _B_T).
    `name`: the parameter this gradient belongs to as a whole - when a data-parallel gradient store is capturing, the GEMM
    writes straight into the parameter's slot of the flat bucket (dp.grad_out), no re-packing copy afterwards.
    (Measured: running these on a second stream, as the ViT engine does, is 1.5-3 % SLOWER here - the decoder's
    memory-bound row kernels and the big dgrad GEMMs leave no idle CUs to fill - so everything stays on one stream.)"""
    out = dp.grad_out(name) if name is not None else None
    o = K.gemm_nt(_full(dy_c), _full(x_c), a_t=True, b_t=True, out=out)
    return post(o) if post is not None else o


def _wg_grouped(dys, xs, names):
    """Up to 4 weight gradients of identical shape and operand strides as ONE launch (libra_gemm_bf16_nt_grouped): the low-rank
    [4096 x 1024] gradients are 64 tiles of 256^2 each - a quarter of the chip per launch (0.58 PFLOP/s inside the step); three of
    them together fill three quarters of one wave.  -> list of gradients (bucket views when a gradient store is capturing)."""
    outs = []
    for dy, x, n in zip(dys, xs, names):
        o = dp.grad_out(n)
        outs.append(o if o is not None else torch.empty((dy.shape[1], x.shape[1]), dtype=BF16, device=dy.device))
    same = lambda ts: len({(t.stride(0), tuple(t.shape)) for t in ts}) == 1
    if not (same(outs) and same(dys) and same(xs)):
        return [_wg(dy, x, name=n) for dy, x, n in zip(dys, xs, names)]
    K.gemm_nt_grouped([_full(dy) for dy in dys], [_full(x) for x in xs], outs, a_t=True, b_t=True)
    return outs


def _norm_wgrad(dy, x, rstd, flag, lang_idx, vis_idx, want_l: bool, want_v: bool, H: int):
    """(dw_lang, dw_vis) bf16 of a routed RMSNorm.  When only one modality's weight is trainable (frozen-language pretraining)
    only that modality's rows are read (28 % of the tokens at the Libra-11B shape)."""
    dev = dy.device
    acc = torch.zeros(2 * H, dtype=torch.float32, device=dev)          # [dw_lang | dw_vis]: one fill, one conversion
    dl, dv = acc[:H], acc[H:]
    sel = None
    if want_v and not want_l:
        sel = vis_idx
    elif want_l and not want_v:
        sel = lang_idx
    K.rmsnorm_routed_wgrad(dy, x, rstd, flag, dl, dv, rows_sel=sel)
    out = K.f32_to_bf16(acc)
    return out[:H], out[H:]


def backward(sd, packed, d: DecDims, out, want, gscale=1.0):
    """Gradients of out["loss"] w.r.t. every parameter name in `want` (a set) -> {name: bf16 grad}.
    gscale: the upstream gradient of the loss - a float, or a one-element device tensor (autograd's incoming scalar), which is
    applied inside the logits-gradient kernel and never read by the host."""
    sv = out["saved"]
    flag, lang_idx, vis_idx = out["flag"], out["lang_idx"], out["vis_idx"]
    B, S, Q = sv["B"], sv["S"], sv["Q"]
    N, H, I, r, rg = B * S, d.hidden, d.inter, d.r, d.rg
    dev = flag.device
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    cos, sin, lens = sv["cos"], sv["sin"], sv["lens"]
    g: Dict[str, torch.Tensor] = {}
    f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
    w = lambda name: name in want

    # ---- loss -> logits -> final hidden
    hidden = sv["hidden"]
    dhid = torch.zeros((N, H), dtype=BF16, device=dev)
    gdev = None
    if isinstance(gscale, torch.Tensor):
        gdev, gscale = gscale.detach().reshape(1).to(device=dev, dtype=torch.float32), 1.0
    coef = [float(gscale / (sv["counts"][q] * Q)) for q in range(Q)]       # upstream gradient folded into dlogits
    def head_pad(name):
        """Head weight with its vocab (the dgrad reduction length) zero-padded to the GEMM's 64 granule."""
        W = sd[name]
        V = W.shape[0]
        Vp = K.round_up(V, 64)
        if Vp == V:
            return W, V, Vp
        wp = torch.zeros((Vp, W.shape[1]), dtype=BF16, device=dev)
        wp[:V] = W
        return wp, V, Vp

    def dlogits(z, t0, t1, sub, c0, c1, Vp):
        n, V = z.shape
        full = torch.zeros((K.round_up(n, 64), Vp), dtype=BF16, device=dev)      # zero pad rows AND pad columns
        K.ce_rows_bwd(z, t0, t1, sub, c0, c1, full[:n, :V], scale=gdev)
        return full

    if Q > 2:
        raise NotImplementedError("more than two codebooks")
    if sv["unified"]:
        # unified_head: per codebook one head [V + Vv, H] = [lm_head; head_q] over every row (modeling_libra.py:1054-1064)
        V, Vv = d.vocab, d.vision_vocab
        Vp = K.round_up(V + Vv, 64)
        hid_c = _compact(hidden, torch.arange(N, dtype=torch.int32, device=dev))       # 64-row padded: the wgrad's reduction axis
        gl = None
        for q in range(Q):
            name = f"vision_lm_head.heads.{q}.weight"
            wcat = packed.heads[q]                                       # [Vp, H]: the forward's operand, pad rows zero
            dzf = dlogits(out["z_all"][q], sv["tgts"][q][0], None, 0, coef[q], 0.0, Vp)
            K.gemm_nt(dzf[:N], wcat, b_t=True, out=dhid, resid=dhid if q > 0 else None)
            if w(name) or w("lm_head.weight"):
                gw = K.gemm_nt(dzf, _full(hid_c), a_t=True, b_t=True)
                if w(name):
                    g[name] = gw[V:V + Vv].contiguous()
                if w("lm_head.weight"):
                    gl = gw[:V].float() if gl is None else gl + gw[:V].float()
        if gl is not None:
            g["lm_head.weight"] = gl.to(BF16)
    if n_l and not sv["unified"]:
        tl = [sv["tgts"][q][0] for q in range(Q)]
        wl, V, Vp = head_pad("lm_head.weight")
        dzf = dlogits(out["z_lang"], tl[0], tl[1] if Q > 1 else None, 0, coef[0], coef[1] if Q > 1 else 0.0, Vp)
        K.gemm_nt(dzf[:n_l], wl, b_t=True, out=dhid, c_rows=lang_idx)
        if w("lm_head.weight"):
            g["lm_head.weight"] = K.gemm_nt(dzf, _full(_compact(hidden, lang_idx)), a_t=True, b_t=True)[:V]
    if n_v and not sv["unified"] and sv["src2d"] is None:
        hv = None
        for q in range(Q):
            name = f"vision_lm_head.heads.{q}.weight"
            wq, V, Vp = head_pad(name)
            dzf = dlogits(out["z_vis"][q], sv["tgts"][q][1], None, d.vocab, coef[q], 0.0, Vp)
            K.gemm_nt(dzf[:n_v], wq, b_t=True, out=dhid, c_rows=vis_idx, resid=dhid if q > 0 else None)
            if w(name):
                hv = _compact(hidden, vis_idx) if hv is None else hv
                g[name] = K.gemm_nt(dzf, _full(hv), a_t=True, b_t=True)[:V]
    if n_v and sv["src2d"] is not None:
        # vision_prediction_mode "2d": z = head_q(cat(hidden[src_a], hidden[src_b])); dfeat scatters back onto its two sources
        feats, (src_a, src_b) = sv["feats"], sv["src2d"]
        dfeat = torch.empty((n_v, 2 * H), dtype=BF16, device=dev)
        for q in range(Q):
            name = f"vision_lm_head.heads.{q}.weight"
            wq, V, Vp = head_pad(name)
            dzf = dlogits(out["z_vis"][q], sv["tgts"][q][1], None, d.vocab, coef[q], 0.0, Vp)
            K.gemm_nt(dzf[:n_v], wq, b_t=True, out=dfeat, resid=dfeat if q > 0 else None)
            if w(name):
                g[name] = K.gemm_nt(dzf, _full(feats), a_t=True, b_t=True)[:V]
        # each hidden row is a source at most twice (left of its own row, above of the row res - 1 further): bf16 index_add of
        # two terms onto zeros is order-independent; the placeholder collects thousands of rows -> fp32 masked sum.  (plumbing)
        ph = torch.zeros(H, dtype=torch.float32, device=dev)
        for src, half in ((src_a, dfeat[:, :H]), (src_b, dfeat[:, H:])):
            real = src < N
            dhid.index_add_(0, src[real].long(), half[real])
            ph += (half.float() * (~real).unsqueeze(1)).sum(0)
        if w("vision_hidden_placeholder"):
            g["vision_hidden_placeholder"] = ph.to(BF16)
    dx = K.rmsnorm_routed_bwd(dhid, sv["x_last"], sd["model.norm.weight"], sd["model.vision_norm.weight"], flag, sv["rstd_f"])
    if w("model.norm.weight") or w("model.vision_norm.weight"):
        g["model.norm.weight"], g["model.vision_norm.weight"] = _norm_wgrad(
            dhid, sv["x_last"], sv["rstd_f"], flag, lang_idx, vis_idx, w("model.norm.weight"), w("model.vision_norm.weight"), H)

    emitted: set = set()
    groups = want_groups(want, d.layers, () if d.pred_2d else _NO_GRAD_NAMES)
    _zero_fill(g, groups[0], sd)               # a head whose modality is absent from this batch still gets a (zero) gradient
    dp.emit_new(g, emitted)                    # heads + final norm
    global _ARENA, _ARENA_PREFIX
    arena = getattr(packed, "arena", None)
    try:
        _ARENA = arena                         # the layer temporaries ("b.*": one slot for all layers) and, under gradient
        _ARENA_PREFIX = "R." if arena is not None else ""      # checkpointing, the rebuilt activations ("R.*") live in the arena
        for i in range(d.layers - 1, -1, -1):
            svi = sv["layers"][i]
            if sv["recompute"]:                # gradient checkpointing: rebuild this layer's activations from its input
                x_in, svi = svi["x"], {}
                layer_forward(sd, packed[i], i, d, x_in, flag, lang_idx, vis_idx, lens, cos, sin, B, S, svi, need_out=False,
                              positions=sv["positions"])
            dx = layer_backward(sd, packed[i], i, d, svi, dx, flag, lang_idx, vis_idx, lens, cos, sin, B, S, g, w,
                                positions=sv["positions"])
            svi.clear()
            sv["layers"][i] = None
            _zero_fill(g, groups[1 + (d.layers - 1 - i)], sd)
            dp.emit_new(g, emitted)            # data parallel: this layer's gradients start their all-reduce now
    finally:
        _ARENA, _ARENA_PREFIX = None, ""
        if arena is not None:
            arena.release(sv.pop("arena_lease", None))
        K.errors.poll_async(dev)               # sticky device error word of the bounded in-kernel waits: read back without a stall

    # ---- embeddings (modeling_libra.py:625-661)
    if sv["input_ids"] is None:                # the forward ran on caller-provided embeddings: their gradient is the result
        out["d_inputs_embeds"] = dx.view(B, S, H)
        _zero_fill(g, groups[-1], sd)
        dp.emit_new(g, emitted)
        return g
    e = sv["emb"]
    proc = "model.vision_contiguous_signal_processor.weight"
    if not d.concat and "sig_all" in e and w(proc):         # x += processor(signal) on every row: dW = dx^T signal
        allr = torch.arange(N, dtype=torch.int32, device=dev)
        g[proc] = _wg(_compact(dx, allr), _compact(e["sig_all"], allr), name=proc)
    if n_v:
        if d.concat:
            dven = K.gemm_nt(dx, sd[proc], b_t=True, a_rows=vis_idx)                                   # [n_v, H+Cs]
            if w(proc):
                g[proc] = _wg(_compact(dx, vis_idx), e["ven"], name=proc)
            if d.norm_sig:
                dve = K.rmsnorm_routed_bwd(dven, e["ve"], sd["model.vision_signal_norm.weight"], None, None, e["rstd_e"])
                if w("model.vision_signal_norm.weight"):
                    dn = f32(H + d.signal)
                    K.rmsnorm_routed_wgrad(dven, e["ve"], e["rstd_e"], None, dn, None)
                    g["model.vision_signal_norm.weight"] = K.f32_to_bf16(dn)
            else:
                dve = dven
        else:
            dve = dx.index_select(0, vis_idx.long())                                                   # plumbing gather
        ids = sv["input_ids"]
        for q in range(Q):
            name = f"model.vision_embed_tokens.{q}.weight"
            if w(name):
                # embedding gradient = scatter-add of rows by token id: tiny (n_v x H/Q), done with torch index_add_ (plumbing)
                acc = torch.zeros(sd[name].shape, dtype=torch.float32, device=dev)
                tok = (ids[q].reshape(-1).index_select(0, vis_idx.long()) - d.vocab)
                acc.index_add_(0, tok, dve[:, q * (H // Q):(q + 1) * (H // Q)].float())
                g[name] = acc.to(BF16)
        name = "model.vision_position_embedding.weight"
        if d.vis_pos and w(name):                           # same scatter-add, by the in-image index
            acc = torch.zeros(sd[name].shape, dtype=torch.float32, device=dev)
            acc.index_add_(0, sv["vision_indices"].reshape(-1).index_select(0, vis_idx.long()), dve[:, :H].float())
            g[name] = acc.to(BF16)
    if n_l and w("model.embed_tokens.weight"):
        acc = torch.zeros(sd["model.embed_tokens.weight"].shape, dtype=torch.float32, device=dev)
        tok = sv["input_ids"][0].reshape(-1).index_select(0, lang_idx.long())
        acc.index_add_(0, tok, dx.index_select(0, lang_idx.long()).float())
        g["model.embed_tokens.weight"] = acc.to(BF16)
    _zero_fill(g, groups[-1], sd)
    dp.emit_new(g, emitted)
    return g


_NO_GRAD_NAMES = ("vision_hidden_placeholder",)       # read only by vision_prediction_mode="2d" (d.pred_2d): no gradient otherwise


def emit_group(name: str, n_layers: int) -> int:
    """Backward-order group of a parameter: 0 = heads + final norms, 1 + (L-1-i) = decoder layer i, L+1 = embedding stage.
    Rank-independent: the data-parallel buckets (dp.GradBuckets) are laid out by it."""
    if name.startswith("model.layers."):
        return 1 + (n_layers - 1 - int(name.split(".")[2]))
    if name in ("model.norm.weight", "model.vision_norm.weight", "vision_hidden_placeholder") or \
            name.startswith(("lm_head.", "vision_lm_head.")):
        return 0
    return n_layers + 1


def want_groups(want, n_layers: int, skip=_NO_GRAD_NAMES):
    groups = [[] for _ in range(n_layers + 2)]
    for n in sorted(want):
        if n not in skip:
            groups[emit_group(n, n_layers)].append(n)
    return groups


def _zero_fill(g, names, sd):
    """Parameters of a modality that has no token in this micro-batch: the reference runs the module on an empty tensor
    and autograd yields ZERO gradients (not None) - and every data-parallel rank must emit the same gradient set."""
    for n in names:
        if n not in g:
            buf = dp.grad_out(n)
            g[n] = buf.zero_() if buf is not None else torch.zeros_like(sd[n], dtype=BF16)


def _layer_backward_multi(sd, pk, i, d: DecDims, sv, dx_out, flag, lang_idx, vis_idx, lens, cos, sin, B, S, g, w, positions=None):
    """layer_backward with the round-6 launch schedule (both modalities present, no addition_mode): per stage ONE multi-problem
    launch = the text dgrad + the vision dgrads that do not wait for one another + the weight gradients whose operands exist by
    then (nothing in the layer consumes a weight gradient, so they ride wherever they fill a launch's tail).  Only the first
    low-rank stage of each vision pair, which everything else of the stage waits for, stays a launch of its own:
        dtd | B1 {dact_l, dact_v, dW down_B, dW down_A} | swiglu' x2 | B2 {dh2_l, dtg x2, dW gate_B, dW up_B} | B3 {dh2_v, dW gate|up_A}
        | norm' | dto | B4 {do_l, do_v, dW o_B, dW o_A} | attention' | rope' | B5 {dh_l, dt x3, dW q|k|v_B} | B6 {dh_v, dW q|k|v_A + bridge A} | norm'
    Same kernels' arithmetic per problem, same results bit for bit as the one-launch-per-GEMM schedule (tests/test_decoder_model_gpu.py)."""
    H, I, r, rg = d.hidden, d.inter, d.r, d.rg
    N = B * S
    dev = dx_out.device
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    pre = f"model.layers.{i}."
    a, m = pre + "self_attn.", pre + "mlp."
    any_l = lambda names: any(w(n) for n in names)
    G = K.gemm_spec

    def WG(dy_c, x_c, name=None, reads=None):   # weight-gradient problem: dW[out, in] = dy^T x, straight into the bucket slot when capturing
        return G(_full(dy_c), _full(x_c), a_t=True, b_t=True, out=dp.grad_out(name) if name is not None else None, reads=reads)

    # ================= MLP =================
    dh2 = torch.empty((N, H), dtype=BF16, device=dev)
    h2, gu = sv["h2"], sv["gu"]
    tg, guv, actv, td = sv["tg"], sv["guv"], sv["actv"], sv["td"]
    dxo_v = _compact(dx_out, vis_idx, "dxo_v")
    dtd = _arows("b.dtd", n_v, r, dev)
    p0 = rd = G(dxo_v, sd[m + "vision_down_proj.weight_B"], b_t=True, out=dtd) if CHAIN else None
    if not CHAIN:
        K.gemm_nt(dxo_v, sd[m + "vision_down_proj.weight_B"], b_t=True, out=dtd)
    probs, post = [G(dx_out, sd[m + "down_proj.weight"], b_t=True, a_rows=lang_idx),                            # dact  [n_l, I]
                   G(dtd, sd[m + "vision_down_proj.weight_A"], b_t=True, reads=rd)], []                          # dactv [n_v, I]
    if w(m + "vision_down_proj.weight_B"):
        probs.append(WG(dxo_v, td, m + "vision_down_proj.weight_B")); post.append(m + "vision_down_proj.weight_B")
    if w(m + "vision_down_proj.weight_A"):
        probs.append(WG(dtd, actv, m + "vision_down_proj.weight_A", reads=rd)); post.append(m + "vision_down_proj.weight_A")
    if w(m + "down_proj.weight"):
        probs.append(WG(_compact(dx_out, lang_idx, "dxo_l"), sv["act"], m + "down_proj.weight")); post.append(m + "down_proj.weight")
    outs = K.gemm_multi(([p0] if CHAIN else []) + probs)[1 if CHAIN else 0:]     # (the producer first: its tiles lead the equal-K group)
    dact, dactv = outs[0], outs[1]
    for nm, o in zip(post, outs[2:]):
        g[nm] = o
    dgu = _arows("b.dgu", n_l, 2 * I, dev)
    K.swiglu_bwd(dact, gu[:, :I], gu[:, I:], dgu[:, :I], dgu[:, I:])
    dguv = _arows("b.dguv", n_v, 2 * I, dev)
    K.swiglu_bwd(dactv, guv[:, :I], guv[:, I:], dguv[:, :I], dguv[:, I:])
    dtg = _arows("b.dtg", n_v, 2 * rg, dev)
    probs, post = [G(dgu, pk["wgu"], b_t=True, out=dh2, c_rows=lang_idx),
                   G(dguv[:, :I], sd[m + "vision_gate_proj.weight_B"], b_t=True, out=dtg[:, :rg]),
                   G(dguv[:, I:], sd[m + "vision_up_proj.weight_B"], b_t=True, out=dtg[:, rg:])], []
    for nm, dy_c, x_c in ((m + "vision_gate_proj.weight_B", dguv[:, :I], tg[:, :rg]), (m + "vision_up_proj.weight_B", dguv[:, I:], tg[:, rg:])):
        if w(nm):
            probs.append(WG(dy_c, x_c, nm)); post.append(nm)
    want_gu = any_l([m + "gate_proj.weight", m + "up_proj.weight"])
    if want_gu:
        probs.append(WG(dgu, _compact(h2, lang_idx, "h2_l")))
    outs = K.gemm_multi(probs)
    for nm, o in zip(post, outs[3:]):
        g[nm] = o
    if want_gu:
        g[m + "gate_proj.weight"], g[m + "up_proj.weight"] = outs[-1][:I], outs[-1][I:]
    probs = [G(dtg, pk["agu"], b_t=True, out=dh2, c_rows=vis_idx)]
    ga_n, ua_n = m + "vision_gate_proj.weight_A", m + "vision_up_proj.weight_A"
    want_agu = any_l([ga_n, ua_n])
    # a capturing gradient store hands out bucket slots: the packed [gate_A; up_A] gradient then goes out as one problem per
    # parameter, each written straight into its slot (same tiles, no re-packing copy: 26 % of the vision gradient bytes were copies)
    split_agu = want_agu and w(ga_n) and w(ua_n) and dp.grad_out(ga_n) is not None and dp.grad_out(ua_n) is not None
    if want_agu:
        h2v = _compact(h2, vis_idx, "h2_v")
        probs += [WG(dtg[:, :rg], h2v, ga_n), WG(dtg[:, rg:], h2v, ua_n)] if split_agu else [WG(dtg, h2v)]
    outs = K.gemm_multi(probs)
    if split_agu:
        g[ga_n], g[ua_n] = outs[1], outs[2]
    elif want_agu:
        g[ga_n], g[ua_n] = outs[1][:rg], outs[1][rg:]
    ln_l, ln_v = pre + "post_attention_layernorm.weight", pre + "vision_post_attention_layernorm.weight"
    dx_mid = K.rmsnorm_routed_bwd(dh2, sv["x_mid"], sd[ln_l], sd[ln_v], flag, sv["rstd2"], dres=dx_out)
    if w(ln_l) or w(ln_v):
        g[ln_l], g[ln_v] = _norm_wgrad(dh2, sv["x_mid"], sv["rstd2"], flag, lang_idx, vis_idx, w(ln_l), w(ln_v), H)

    # ================= attention =================
    o = sv["o"]
    do = torch.empty((N, H), dtype=BF16, device=dev)
    dxm_v = _compact(dx_mid, vis_idx, "dxm_v")
    dto = _arows("b.dto", n_v, r, dev)
    p0 = rd = G(dxm_v, sd[a + "vision_o_proj.weight_B"], b_t=True, out=dto) if CHAIN else None
    if not CHAIN:
        K.gemm_nt(dxm_v, sd[a + "vision_o_proj.weight_B"], b_t=True, out=dto)
    probs, post = [G(dx_mid, sd[a + "o_proj.weight"], b_t=True, a_rows=lang_idx, c_rows=lang_idx, out=do),
                   G(dto, sd[a + "vision_o_proj.weight_A"], b_t=True, out=do, c_rows=vis_idx, reads=rd)], []
    if w(a + "vision_o_proj.weight_B"):
        probs.append(WG(dxm_v, sv["to"], a + "vision_o_proj.weight_B")); post.append(a + "vision_o_proj.weight_B")
    if w(a + "vision_o_proj.weight_A"):
        probs.append(WG(dto, _compact(o, vis_idx, "o_v"), a + "vision_o_proj.weight_A", reads=rd)); post.append(a + "vision_o_proj.weight_A")
    if w(a + "o_proj.weight"):
        probs.append(WG(_compact(dx_mid, lang_idx, "dxm_l"), _compact(o, lang_idx, "o_l"), a + "o_proj.weight")); post.append(a + "o_proj.weight")
    outs = K.gemm_multi(([p0] if CHAIN else []) + probs)[1 if CHAIN else 0:]
    for nm, o_ in zip(post, outs[2:]):
        g[nm] = o_
    qkv, kc, vc, tb = sv["qkv"], sv["kc"], sv["vc"], sv["tb"]
    dq, dks, dkc, dvs, dvc = K.bridge_attn_bwd(qkv[:, :H], qkv[:, H:2 * H], kc, qkv[:, 2 * H:], vc, o, do, flag, lens,
                                               sv["lse"], B, S, d.heads, (H // d.heads) ** -0.5, out_lo=sv["o_lo"])
    dqkvt = torch.empty((N, 3 * H + 64), dtype=BF16, device=dev)     # [dq | dk | dv | dt_k dt_v 0..], mirrors the forward's qkvt
    dqkv, dtb = dqkvt[:, :3 * H], dqkvt[:, 3 * H:]
    dtb.zero_()
    dkb = torch.empty((N, H), dtype=BF16, device=dev)
    K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, d.heads, dqkv, dkb,
                      bridge_b=(pk["bkT_l"], pk["bkT_v"], pk["bvT_l"], pk["bvT_v"]), flag=flag, dtb=dtb, positions=positions)
    dvb = dvc
    h = sv["h"]
    bnames = {kv: [a + f"vision_{kv}_bridge_on_{which}.weight_B" for which in ("language", "vision")] for kv in "kv"}
    if d.rank == 8:
        for kv, xg, col0 in (("k", dkb, 0), ("v", dvb, 8)):
            nl_, nv_ = bnames[kv]
            if w(nl_) or w(nv_):
                gl_, gv_ = K.rank_outer_wgrad(xg, tb[:, col0:col0 + 8], flag, transpose_out=True, want_l=w(nl_), want_v=w(nv_))
                if w(nl_):
                    g[nl_] = gl_
                if w(nv_):
                    g[nv_] = gv_
    else:
        for wi, (idx, which) in enumerate(((lang_idx, "language"), (vis_idx, "vision"))):
            nk, nv = bnames["k"][wi], bnames["v"][wi]
            if w(nk) or w(nv):
                tbc = _compact(tb, idx, "tb_" + which)
                if w(nk):
                    g[nk] = _wg(_compact(dkb, idx, "dkb_" + which), tbc[:, 0:8], post=lambda o_: o_[:, :d.rank].contiguous())
                if w(nv):
                    g[nv] = _wg(_compact(dvb, idx, "dvb_" + which), tbc[:, 8:16], post=lambda o_: o_[:, :d.rank].contiguous())
    dh = torch.empty((N, H), dtype=BF16, device=dev)
    dt_ext = _arows("b.dt_ext", n_v, 3 * r + 64, dev)                # [dt_q | dt_k | dt_v | dt_bridge], mirrors the forward's t_ext
    K.copy_rows(dtb, vis_idx, n_v, dt_ext, 3 * r)                    # the vision rows' 64 bridge columns
    t = sv["t"]
    dqkv_v = _compact(dqkv, vis_idx, "dqkv_v")                        # [n_v, 3H]
    dt = dt_ext[:, :3 * r]
    bq = [a + f"vision_{nm}_proj.weight_B" for nm in ("q", "k", "v")]
    probs, post = [G(dqkvt, pk["wqkv_ab"], b_t=True, a_rows=lang_idx, c_rows=lang_idx, out=dh)] + \
                  [G(dqkv_v[:, j * H:(j + 1) * H], sd[bq[j]], b_t=True, out=dt[:, j * r:(j + 1) * r]) for j in range(3)], []
    for j, nm in enumerate(bq):
        if w(nm):
            probs.append(WG(dqkv_v[:, j * H:(j + 1) * H], t[:, j * r:(j + 1) * r], nm)); post.append(nm)
    nk, nv = a + "vision_k_bridge_on_language.weight_A", a + "vision_v_bridge_on_language.weight_A"
    want_txt = any_l([a + "q_proj.weight", a + "k_proj.weight", a + "v_proj.weight"])
    if want_txt:
        probs.append(WG(_compact(dqkvt, lang_idx, "dqkvt_l"), _compact(h, lang_idx, "h_l")))                   # [3H + 64, H]
    outs = K.gemm_multi(probs)
    for nm, o_ in zip(post, outs[4:]):
        g[nm] = o_
    if want_txt:
        dw = outs[-1]
        for j, nm in enumerate(("q", "k", "v")):
            g[a + f"{nm}_proj.weight"] = dw[j * H:(j + 1) * H]
        if w(nk) or w(nv):
            g[nk], g[nv] = dw[3 * H:3 * H + d.rank].contiguous(), dw[3 * H + 8:3 * H + 8 + d.rank].contiguous()
    elif (w(nk) or w(nv)) and d.rank == 8:
        ga, _ = K.rank_outer_wgrad(h, dtb[:, 0:16], flag, transpose_out=False, want_l=True, want_v=False)
        g[nk], g[nv] = ga[0:8], ga[8:16]
    elif w(nk) or w(nv):
        g[nk], g[nv] = _wg(_compact(dtb, lang_idx, "dtb_l"), _compact(h, lang_idx, "h_l"),                       # [64, H]
                           post=lambda o_: (o_[0:d.rank].contiguous(), o_[8:8 + d.rank].contiguous()))
    probs = [G(dt_ext, pk["aqkv_ab"], b_t=True, out=dh, c_rows=vis_idx)]                                          # K = 3r + 64
    nk, nv = a + "vision_k_bridge_on_vision.weight_A", a + "vision_v_bridge_on_vision.weight_A"
    an = [a + f"vision_{nm}_proj.weight_A" for nm in "qkv"]
    want_a = any_l(an)
    split_a = all(w(n) for n in an) and all(dp.grad_out(n) is not None for n in an)      # (as above: one problem per bucket slot)
    if want_a or w(nk) or w(nv):
        hv = _compact(h, vis_idx, "h_v")
        if split_a:
            probs += [WG(dt_ext[:, j * r:(j + 1) * r], hv, an[j]) for j in range(3)]
            if w(nk) or w(nv):
                probs.append(WG(dt_ext[:, 3 * r:], hv))                                                           # the bridge rows [64, H]
        else:
            probs.append(WG(dt_ext, hv))                                                                          # [3r + 64, H]
    outs = K.gemm_multi(probs)
    if split_a:
        for j in range(3):
            g[an[j]] = outs[1 + j]
        if w(nk) or w(nv):
            g[nk], g[nv] = outs[4][0:d.rank].contiguous(), outs[4][8:8 + d.rank].contiguous()
    elif want_a or w(nk) or w(nv):
        da = outs[1]
        if want_a:
            for j, nm in enumerate(("q", "k", "v")):
                g[a + f"vision_{nm}_proj.weight_A"] = da[j * r:(j + 1) * r]
        if w(nk) or w(nv):
            g[nk], g[nv] = da[3 * r:3 * r + d.rank].contiguous(), da[3 * r + 8:3 * r + 8 + d.rank].contiguous()
    ln_l, ln_v = pre + "input_layernorm.weight", pre + "vision_input_layernorm.weight"
    dx = K.rmsnorm_routed_bwd(dh, sv["x"], sd[ln_l], sd[ln_v], flag, sv["rstd1"], dres=dx_mid)
    if w(ln_l) or w(ln_v):
        g[ln_l], g[ln_v] = _norm_wgrad(dh, sv["x"], sv["rstd1"], flag, lang_idx, vis_idx, w(ln_l), w(ln_v), H)
    return dx


def layer_backward(sd, pk, i, d: DecDims, sv, dx_out, flag, lang_idx, vis_idx, lens, cos, sin, B, S, g, w, positions=None):
    if _multi_ok(d, lang_idx.numel(), vis_idx.numel(), dx_out.device):
        return _layer_backward_multi(sd, pk, i, d, sv, dx_out, flag, lang_idx, vis_idx, lens, cos, sin, B, S, g, w, positions)
    H, I, r, rg = d.hidden, d.inter, d.r, d.rg
    N = B * S
    dev = dx_out.device
    n_l, n_v = lang_idx.numel(), vis_idx.numel()
    pre = f"model.layers.{i}."
    a, m = pre + "self_attn.", pre + "mlp."
    f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
    any_l = lambda names: any(w(n) for n in names)
    defer_b = None                 # (dy, x, name) of vision_down_proj.weight_B until it can share a launch with vision_o_proj.weight_B

    # ================= MLP: x_out = x_mid + down(silu(gate(h2)) * up(h2)), routed =================
    dh2 = torch.empty((N, H), dtype=BF16, device=dev)
    h2 = sv["h2"]
    if n_l:
        gu = sv["gu"]
        dact = K.gemm_nt(dx_out, sd[m + "down_proj.weight"], b_t=True, a_rows=lang_idx)                 # [n_l, I]
        if w(m + "down_proj.weight"):
            g[m + "down_proj.weight"] = _wg(_compact(dx_out, lang_idx, "dxo_l"), sv["act"], name=m + "down_proj.weight")
        dgu = _arows("b.dgu", n_l, 2 * I, dev)
        # (SwiGLU' as an epilogue of the dgrad above was built and measured: the 256^2 kernel's un-overlapped epilogue moves the
        #  extra 3 tiles of traffic slower than this HBM-rate kernel does - +13.3 ms of GEMM for 9.5 ms of row kernel per step;
        #  profiles/r03_fusion_ab.txt)
        K.swiglu_bwd(dact, gu[:, :I], gu[:, I:], dgu[:, :I], dgu[:, I:])
        K.gemm_nt(dgu, pk["wgu"], b_t=True, out=dh2, c_rows=lang_idx)
        if any_l([m + "gate_proj.weight", m + "up_proj.weight"]):
            dwgu = _wg(dgu, _compact(h2, lang_idx, "h2_l"))
            g[m + "gate_proj.weight"], g[m + "up_proj.weight"] = dwgu[:I], dwgu[I:]
    if n_v:
        tg, guv, actv, td = sv["tg"], sv["guv"], sv["actv"], sv["td"]
        dxo_v = _compact(dx_out, vis_idx, "dxo_v")
        dtd = K.gemm_nt(dxo_v, sd[m + "vision_down_proj.weight_B"], b_t=True, out=_arows("b.dtd", n_v, r, dev))
        if w(m + "vision_down_proj.weight_B"):              # [H, r] like vision_o_proj.weight_B: both in one launch further down
            defer_b = (dxo_v, td, m + "vision_down_proj.weight_B")
        dactv = K.gemm_nt(dtd, sd[m + "vision_down_proj.weight_A"], b_t=True)                            # [n_v, I]
        if w(m + "vision_down_proj.weight_A"):
            g[m + "vision_down_proj.weight_A"] = _wg(dtd, actv, name=m + "vision_down_proj.weight_A")
        dguv = _arows("b.dguv", n_v, 2 * I, dev)
        K.swiglu_bwd(dactv, guv[:, :I], guv[:, I:], dguv[:, :I], dguv[:, I:])
        dtg = _arows("b.dtg", n_v, 2 * rg, dev)
        K.gemm_nt_grouped([dguv[:, :I], dguv[:, I:]], [sd[m + "vision_gate_proj.weight_B"], sd[m + "vision_up_proj.weight_B"]],
                          [dtg[:, :rg], dtg[:, rg:]], b_t=True)
        if w(m + "vision_gate_proj.weight_B"):
            g[m + "vision_gate_proj.weight_B"] = _wg(dguv[:, :I], tg[:, :rg], name=m + "vision_gate_proj.weight_B")
        if w(m + "vision_up_proj.weight_B"):
            g[m + "vision_up_proj.weight_B"] = _wg(dguv[:, I:], tg[:, rg:], name=m + "vision_up_proj.weight_B")
        K.gemm_nt(dtg, pk["agu"], b_t=True, out=dh2, c_rows=vis_idx)
        if any_l([m + "vision_gate_proj.weight_A", m + "vision_up_proj.weight_A"]):
            dagu = _wg(dtg, _compact(h2, vis_idx, "h2_v"))
            g[m + "vision_gate_proj.weight_A"], g[m + "vision_up_proj.weight_A"] = dagu[:rg], dagu[rg:]
    ln_l, ln_v = pre + "post_attention_layernorm.weight", pre + "vision_post_attention_layernorm.weight"
    dx_mid = K.rmsnorm_routed_bwd(dh2, sv["x_mid"], sd[ln_l], sd[ln_v], flag, sv["rstd2"], dres=dx_out)
    if w(ln_l) or w(ln_v):
        g[ln_l], g[ln_v] = _norm_wgrad(dh2, sv["x_mid"], sv["rstd2"], flag, lang_idx, vis_idx, w(ln_l), w(ln_v), H)

    # ================= attention: x_mid = x + o_proj(attn(rope(q), K_same/K_cross, V_same/V_cross)) =================
    o = sv["o"]
    do = torch.empty((N, H), dtype=BF16, device=dev)
    add = d.addition               # addition_mode: the language q / k / v / o projections saw every row; vision terms ACCUMULATE
    if add:
        K.gemm_nt(dx_mid, sd[a + "o_proj.weight"], b_t=True, out=do)
        if w(a + "o_proj.weight"):
            g[a + "o_proj.weight"] = _wg(_padded(dx_mid, "dxm_a"), _padded(o, "o_a"), name=a + "o_proj.weight")
    elif n_l:
        K.gemm_nt(dx_mid, sd[a + "o_proj.weight"], b_t=True, a_rows=lang_idx, c_rows=lang_idx, out=do)
        if w(a + "o_proj.weight"):
            g[a + "o_proj.weight"] = _wg(_compact(dx_mid, lang_idx, "dxm_l"), _compact(o, lang_idx, "o_l"), name=a + "o_proj.weight")
    if n_v:
        dxm_v = _compact(dx_mid, vis_idx, "dxm_v")
        dto = K.gemm_nt(dxm_v, sd[a + "vision_o_proj.weight_B"], b_t=True, out=_arows("b.dto", n_v, r, dev))
        if w(a + "vision_o_proj.weight_B"):
            if defer_b is not None:
                g[a + "vision_o_proj.weight_B"], g[defer_b[2]] = _wg_grouped([dxm_v, defer_b[0]], [sv["to"], defer_b[1]],
                                                                             [a + "vision_o_proj.weight_B", defer_b[2]])
                defer_b = None
            else:
                g[a + "vision_o_proj.weight_B"] = _wg(dxm_v, sv["to"], name=a + "vision_o_proj.weight_B")
        K.gemm_nt(dto, sd[a + "vision_o_proj.weight_A"], b_t=True, out=do, c_rows=vis_idx, resid=do if add else None)
        if w(a + "vision_o_proj.weight_A"):
            g[a + "vision_o_proj.weight_A"] = _wg(dto, _compact(o, vis_idx, "o_v"), name=a + "vision_o_proj.weight_A")
    if defer_b is not None:
        g[defer_b[2]] = _wg(defer_b[0], defer_b[1], name=defer_b[2])
    qkv, kc, vc, tb = sv["qkv"], sv["kc"], sv["vc"], sv["tb"]
    dq, dks, dkc, dvs, dvc = K.bridge_attn_bwd(qkv[:, :H], qkv[:, H:2 * H], kc, qkv[:, 2 * H:], vc, o, do, flag, lens,
                                               sv["lse"], B, S, d.heads, (H // d.heads) ** -0.5, out_lo=sv["o_lo"])
    dqkvt = torch.empty((N, 3 * H + 64), dtype=BF16, device=dev)     # [dq | dk | dv | dt_k dt_v 0..], mirrors the forward's qkvt
    dqkv, dtb = dqkvt[:, :3 * H], dqkvt[:, 3 * H:]
    dtb.zero_()
    dkb = torch.empty((N, H), dtype=BF16, device=dev)
    # (dt_k = B_k^T dkb, dt_v = B_v^T dvb land in dtb[:, 0:16] from the same kernel: no skinny GEMMs re-reading dkb / dvb)
    K.rope_bridge_bwd(dq, dks, dkc, dvs, dvc, cos, sin, S, d.heads, dqkv, dkb,
                      bridge_b=(pk["bkT_l"], pk["bkT_v"], pk["bvT_l"], pk["bvT_v"]), flag=flag, dtb=dtb, positions=positions)
    dvb = dvc
    # rank-8 bridges: kb = B_k[m] t_k, vb = B_v[m] t_v, t = [A_k[m]; A_v[m]] h
    h = sv["h"]
    bnames = {kv: [a + f"vision_{kv}_bridge_on_{which}.weight_B" for which in ("language", "vision")] for kv in "kv"}
    if d.rank == 8:
        # dB[m][c][j] = sum_{t in m} dkb[t][c] t_k[t][j]: one pass over dkb / dvb for both modalities (libra_rank_outer_wgrad)
        for kv, xg, col0 in (("k", dkb, 0), ("v", dvb, 8)):
            nl_, nv_ = bnames[kv]
            if w(nl_) or w(nv_):
                gl_, gv_ = K.rank_outer_wgrad(xg, tb[:, col0:col0 + 8], flag, transpose_out=True, want_l=w(nl_), want_v=w(nv_))
                if w(nl_):
                    g[nl_] = gl_
                if w(nv_):
                    g[nv_] = gv_
    else:
        for wi, (idx, which) in enumerate(((lang_idx, "language"), (vis_idx, "vision"))):
            if idx.numel() == 0:
                continue
            nk, nv = bnames["k"][wi], bnames["v"][wi]
            if w(nk) or w(nv):
                tbc = _compact(tb, idx, "tb_" + which)
                if w(nk):
                    g[nk] = _wg(_compact(dkb, idx, "dkb_" + which), tbc[:, 0:8], post=lambda o: o[:, :d.rank].contiguous())
                if w(nv):
                    g[nv] = _wg(_compact(dvb, idx, "dvb_" + which), tbc[:, 8:16], post=lambda o: o[:, :d.rank].contiguous())
    dh = torch.empty((N, H), dtype=BF16, device=dev)
    dt_ext = None
    if n_v:
        dt_ext = _arows("b.dt_ext", n_v, 3 * r + 64, dev)            # [dt_q | dt_k | dt_v | dt_bridge], mirrors the forward's t_ext
        K.copy_rows(dtb, vis_idx, n_v, dt_ext, 3 * r)                # the vision rows' 64 bridge columns
        if add:                    # the vision rows' bridge columns belong to the VISION bridge A only: with them zeroed, one GEMM /
            dtb.index_fill_(0, vis_idx.long(), 0)                    # wgrad over every row serves q / k / v (all rows) + language bridge A
    if add or n_l:
        if add:
            K.gemm_nt(dqkvt, pk["wqkv_ab"], b_t=True, out=dh)                                       # K = 3H + 64, every row
        else:
            K.gemm_nt(dqkvt, pk["wqkv_ab"], b_t=True, a_rows=lang_idx, c_rows=lang_idx, out=dh)   # K = 3H + 64
        nk, nv = a + "vision_k_bridge_on_language.weight_A", a + "vision_v_bridge_on_language.weight_A"
        if any_l([a + "q_proj.weight", a + "k_proj.weight", a + "v_proj.weight"]):
            hl = _padded(h, "h_a") if add else _compact(h, lang_idx, "h_l")
            dw, gk, gv = _wg(_padded(dqkvt, "dqkvt_a") if add else _compact(dqkvt, lang_idx, "dqkvt_l"), hl,   # [3H + 64, H]
                             post=lambda o: (o, o[3 * H:3 * H + d.rank].contiguous(), o[3 * H + 8:3 * H + 8 + d.rank].contiguous()))
            for j, nm in enumerate(("q", "k", "v")):
                g[a + f"{nm}_proj.weight"] = dw[j * H:(j + 1) * H]
            if w(nk) or w(nv):
                g[nk], g[nv] = gk, gv
        elif (w(nk) or w(nv)) and d.rank == 8:                 # frozen language projections (pretraining): bridge A's only -
            # dA[j][c] = sum_{text t} dt[t][j] h[t][c], j = (k: 0-7, v: 8-15): one pass over h, no compacted copy of it
            ga, _ = K.rank_outer_wgrad(h, dtb[:, 0:16], flag, transpose_out=False, want_l=True, want_v=False)
            g[nk], g[nv] = ga[0:8], ga[8:16]
        elif w(nk) or w(nv):
            hl = _compact(h, lang_idx, "h_l")
            g[nk], g[nv] = _wg(_compact(dtb, lang_idx, "dtb_l"), hl,                                  # [64, H]
                               post=lambda o: (o[0:d.rank].contiguous(), o[8:8 + d.rank].contiguous()))
    if n_v:
        t = sv["t"]
        dqkv_v = _compact(dqkv, vis_idx, "dqkv_v")                                                    # [n_v, 3H]
        dt = dt_ext[:, :3 * r]
        K.gemm_nt_grouped([dqkv_v[:, j * H:(j + 1) * H] for j in range(3)],
                          [sd[a + f"vision_{nm}_proj.weight_B"] for nm in ("q", "k", "v")],
                          [dt[:, j * r:(j + 1) * r] for j in range(3)], b_t=True)
        bq = [a + f"vision_{nm}_proj.weight_B" for nm in ("q", "k", "v")]
        if all(w(n) for n in bq):                                     # the three [H, r] gradients: one grouped launch
            for n, gr in zip(bq, _wg_grouped([dqkv_v[:, j * H:(j + 1) * H] for j in range(3)],
                                             [t[:, j * r:(j + 1) * r] for j in range(3)], bq)):
                g[n] = gr
        else:
            for j, n in enumerate(bq):
                if w(n):
                    g[n] = _wg(dqkv_v[:, j * H:(j + 1) * H], t[:, j * r:(j + 1) * r], name=n)
        K.gemm_nt(dt_ext, pk["aqkv_ab"], b_t=True, out=dh, c_rows=vis_idx, resid=dh if add else None)   # K = 3r + 64
        nk, nv = a + "vision_k_bridge_on_vision.weight_A", a + "vision_v_bridge_on_vision.weight_A"
        want_a = any_l([a + f"vision_{nm}_proj.weight_A" for nm in "qkv"])
        if want_a or w(nk) or w(nv):
            hv = _compact(h, vis_idx, "h_v")
            da, gk, gv = _wg(dt_ext, hv,                                                   # [3r + 64, H]
                             post=lambda o: (o, o[3 * r:3 * r + d.rank].contiguous(), o[3 * r + 8:3 * r + 8 + d.rank].contiguous()))
            if want_a:
                for j, nm in enumerate(("q", "k", "v")):
                    g[a + f"vision_{nm}_proj.weight_A"] = da[j * r:(j + 1) * r]
            if w(nk) or w(nv):
                g[nk], g[nv] = gk, gv
    ln_l, ln_v = pre + "input_layernorm.weight", pre + "vision_input_layernorm.weight"
    dx = K.rmsnorm_routed_bwd(dh, sv["x"], sd[ln_l], sd[ln_v], flag, sv["rstd1"], dres=dx_mid)
    if w(ln_l) or w(ln_v):
        g[ln_l], g[ln_v] = _norm_wgrad(dh, sv["x"], sv["rstd1"], flag, lang_idx, vis_idx, w(ln_l), w(ln_v), H)
    return dx
