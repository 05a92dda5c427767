"""Generation loop over Libra's multi-codebook ids — the product-side counterpart of the reference's custom
`greedy_search` / `sample` (/root/reference/libra/models/libra/modeling_libra_utils.py:61-328, :330-620) and of its image-shape
logits rule `ValidImageLogitsProcessor` (/root/reference/libra/models/llama/modeling_llama_utils.py:23-76).

`input_ids` are [Q, B, S] (Q codebooks); every step feeds the last token of every sequence through the cached decoder
(`LibraForCausalLM._forward_cached` -> `decoder_engine.decode_step`, a replayed hipGraph), takes `logits[:, :, -1, :]`
([Q, B, V + Vv]), runs the processors / warpers per codebook, picks one token per codebook and sequence, and appends.
Batched prompts are padded on the LEFT (`tokenizer.padding_side = 'left'`, as the reference's demo notebook does).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch
from transformers.utils import ModelOutput


class ValidImageLogitsProcessor:
    """Keeps a generated image well-formed: once a BOI has been emitted, exactly `valid_image_token_length` code tokens follow
    (only code ids are allowed, neither BOI / EOI nor text), then EOI is forced.  State is read off the ids themselves: the
    number of trailing ids >= `image_logits_offset` (the text vocabulary size).

    __call__(input_ids [Q,B,S], scores [Q,B,V']) -> scores, each codebook judged on its own ids like the reference's loop."""

    def __init__(self, valid_image_token_length: int, boi_token_id: int, eoi_token_id: int, image_logits_offset: int,
                 logits_size: int):
        if math.isqrt(valid_image_token_length) ** 2 != valid_image_token_length:
            raise AssertionError("only support square images, and valid_image_token_length does not consider <img> and "
                                 "<\\img> tokens")
        self.full = valid_image_token_length + 2            # BOI + codes + EOI
        self.boi, self.eoi, self.offset = boi_token_id, eoi_token_id, image_logits_offset
        code_only = torch.zeros(logits_size, dtype=torch.bool)
        code_only[image_logits_offset:] = True
        code_only[boi_token_id] = False
        code_only[eoi_token_id] = False
        self._code_only = code_only                         # allowed while the image body is being generated
        eoi_only = torch.zeros(logits_size, dtype=torch.bool)
        eoi_only[eoi_token_id] = True
        self._eoi_only = eoi_only                           # allowed when the body is complete

    def _trailing_image_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        """[B,S] -> [B]: how many ids at the end of each sequence are vision ids."""
        is_text = torch.flip(ids, dims=[-1]) < self.offset
        return (torch.cumsum(is_text, dim=-1) == 0).sum(-1)

    def process_score(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        if input_ids.dim() != 2:
            raise AssertionError("Currently, input_ids must shaped as (batch_size, num_seq)")
        n = self._trailing_image_tokens(input_ids)
        body = (n > 0) & (n < self.full - 1)                # next token: a code
        close = n == self.full - 1                          # next token: EOI
        done = n == self.full                               # a complete image must end with EOI
        if bool((n > self.full).any()):
            raise ValueError("You have generated an invalid image.")
        if bool((input_ids[done][:, -1] != self.eoi).any()):
            raise ValueError("Find images that do not end with <\\img> tokens")
        dev = scores.device
        ninf = torch.tensor(float("-inf"), dtype=scores.dtype, device=dev)
        scores = torch.where(body[:, None] & ~self._code_only.to(dev)[None, :], ninf, scores)
        scores = torch.where(close[:, None] & ~self._eoi_only.to(dev)[None, :], ninf, scores)
        return scores

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        return torch.stack([self.process_score(i, s) for i, s in zip(input_ids, scores)])


class NoNewlineLogitsProcessor:
    """After a newline only EOS may follow (modeling_llama_utils.py:9-21)."""

    def __init__(self, newline_token_id: int, eos_token_id: int):
        if not isinstance(newline_token_id, int) or not isinstance(eos_token_id, int):
            raise ValueError("token ids have to be an integer")
        self.newline, self.eos = newline_token_id, eos_token_id

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        after_nl = input_ids[..., -1] == self.newline
        keep = torch.zeros(scores.shape[-1], dtype=torch.bool, device=scores.device)
        keep[self.eos] = True
        return torch.where(after_nl[..., None] & ~keep, torch.tensor(float("-inf"), dtype=scores.dtype, device=scores.device), scores)


@dataclass
class LibraGenerateOutput(ModelOutput):
    sequences: torch.LongTensor = None                      # [Q, B, S_total]
    scores: Optional[Tuple[torch.FloatTensor]] = None       # per step [Q, B, V + Vv] (processed)
    logits: Optional[Tuple[torch.FloatTensor]] = None       # per step, raw


def _run_all(fns, input_ids, scores):
    if fns is None:
        return scores
    if callable(fns) and not isinstance(fns, (list, tuple)):
        return fns(input_ids, scores)
    for f in fns:
        scores = f(input_ids, scores)
    return scores


class LibraGenerationMixin:
    """greedy_search / sample / generate for LibraForCausalLM (uses its prepare_inputs_for_generation,
    _update_model_kwargs_for_generation and cached forward)."""

    def _generation_loop(self, input_ids, choose: Callable, *, logits_processor=None, logits_warper=None, stopping_criteria=None,
                         max_length: Optional[int] = None, pad_token_id: Optional[int] = None,
                         eos_token_id: Optional[Union[int, Sequence[int]]] = None, output_scores: bool = False,
                         output_logits: bool = False, return_dict_in_generate: bool = False, streamer=None, **model_kwargs):
        if input_ids.dim() != 3:
            raise ValueError("input_ids must be [Q, B, S] (one row of ids per codebook)")
        cfg = self.config
        pad_token_id = pad_token_id if pad_token_id is not None else getattr(cfg, "pad_token_id", None)
        eos_token_id = eos_token_id if eos_token_id is not None else getattr(cfg, "eos_token_id", None)
        if isinstance(eos_token_id, int):
            eos_token_id = [eos_token_id]
        if eos_token_id is not None and pad_token_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")
        dev = input_ids.device
        eos = torch.tensor(eos_token_id, device=dev) if eos_token_id is not None else None
        B = input_ids.shape[1]
        unfinished = torch.ones(B, dtype=torch.long, device=dev)
        model_kwargs.setdefault("use_cache", True)
        if max_length is not None:
            model_kwargs.setdefault("max_cache_len", max_length)      # sizes the KV cache (instead of max_position_embeddings)
        scores_t, logits_t = [], []

        def stop(ids0) -> bool:
            if max_length is not None and ids0.shape[-1] >= max_length:
                return True
            if stopping_criteria is None:
                return False
            crit = stopping_criteria if isinstance(stopping_criteria, (list, tuple)) else [stopping_criteria]
            return any(bool(torch.as_tensor(c(ids0, None)).all()) for c in crit)

        while True:
            model_inputs = self.prepare_inputs_for_generation(input_ids, **model_kwargs)
            model_inputs.pop("past_hidden_states", None)
            model_inputs.pop("past_vision_flag", None)
            max_cache_len = model_kwargs.get("max_cache_len")
            outputs = self(**model_inputs, return_dict=True, **({"max_cache_len": max_cache_len} if max_cache_len else {}))
            raw = outputs.logits[:, :, -1, :]
            nxt_scores = _run_all(logits_warper, input_ids, _run_all(logits_processor, input_ids, raw))
            if output_scores:
                scores_t.append(nxt_scores)
            if output_logits:
                logits_t.append(raw)
            next_tokens = choose(nxt_scores)                             # [Q, B]
            cols = []
            for q in range(input_ids.shape[0]):                          # codebook by codebook, like the reference: a sequence that
                tok = next_tokens[q]                                     # ends in codebook q is already padded in codebook q + 1
                if eos is not None:
                    tok = tok * unfinished + pad_token_id * (1 - unfinished)
                    unfinished = unfinished * (tok[None, :] != eos[:, None]).all(0).long()
                if streamer is not None:
                    streamer.put(tok.cpu())
                cols.append(tok)
            model_kwargs = self._update_model_kwargs_for_generation(outputs, model_kwargs)
            input_ids = torch.cat([input_ids, torch.stack(cols)[:, :, None]], dim=-1)
            done = eos is not None and int(unfinished.max()) == 0
            if done or stop(input_ids[0]):
                break
        if streamer is not None:
            streamer.end()
        if return_dict_in_generate:
            return LibraGenerateOutput(sequences=input_ids, scores=tuple(scores_t) if output_scores else None,
                                       logits=tuple(logits_t) if output_logits else None)
        return input_ids

    @torch.no_grad()
    def greedy_search(self, input_ids, logits_processor=None, stopping_criteria=None, max_length=None, pad_token_id=None,
                      eos_token_id=None, output_scores=False, return_dict_in_generate=False, streamer=None, **model_kwargs):
        """argmax per codebook and sequence (modeling_libra_utils.py:61-328)."""
        return self._generation_loop(input_ids, lambda s: torch.argmax(s, dim=-1), logits_processor=logits_processor,
                                     stopping_criteria=stopping_criteria, max_length=max_length, pad_token_id=pad_token_id,
                                     eos_token_id=eos_token_id, output_scores=output_scores,
                                     return_dict_in_generate=return_dict_in_generate, streamer=streamer, **model_kwargs)

    @torch.no_grad()
    def sample(self, input_ids, logits_processor=None, stopping_criteria=None, logits_warper=None, max_length=None,
               pad_token_id=None, eos_token_id=None, output_scores=False, output_logits=False, return_dict_in_generate=False,
               streamer=None, generator: Optional[torch.Generator] = None, **model_kwargs):
        """one multinomial draw per codebook and sequence from softmax(scores) (modeling_libra_utils.py:330-620)."""
        def choose(scores):
            # a forced token is scored +inf (the EOI -> newline rule): clamp so that its probability is 1, not NaN
            probs = torch.softmax(torch.nan_to_num(scores.float(), posinf=torch.finfo(torch.float32).max), dim=-1)
            return torch.stack([torch.multinomial(p, num_samples=1, generator=generator).squeeze(1) for p in probs])
        return self._generation_loop(input_ids, choose, logits_processor=logits_processor, logits_warper=logits_warper,
                                     stopping_criteria=stopping_criteria, max_length=max_length, pad_token_id=pad_token_id,
                                     eos_token_id=eos_token_id, output_scores=output_scores, output_logits=output_logits,
                                     return_dict_in_generate=return_dict_in_generate, streamer=streamer, **model_kwargs)

    @torch.no_grad()
    def generate(self, input_ids, *, attention_mask=None, vision_indices=None, contiguous_signal=None, max_new_tokens: int = 20,
                 max_length: Optional[int] = None, do_sample: bool = False, temperature: float = 1.0, top_k: int = 0,
                 logits_processor=None, **kwargs):
        """Convenience front end: greedy (default) or temperature / top-k sampling until `max_length` / EOS."""
        if attention_mask is None:
            attention_mask = torch.ones(input_ids.shape[1:], dtype=torch.long, device=input_ids.device)
        if vision_indices is None:
            raise ValueError("generate() needs the prompt's vision_indices (LibraTokenizer output)")
        max_length = max_length if max_length is not None else input_ids.shape[-1] + max_new_tokens
        kw = dict(attention_mask=attention_mask, vision_indices=vision_indices, contiguous_signal=contiguous_signal, **kwargs)
        if not do_sample:
            return self.greedy_search(input_ids, logits_processor=logits_processor, max_length=max_length, **kw)
        warpers: List[Callable] = []
        if temperature != 1.0:
            warpers.append(lambda ids, s: s / temperature)
        if top_k > 0:
            def _topk(ids, s):
                kth = torch.topk(s, min(top_k, s.shape[-1]), dim=-1).values[..., -1:]
                return torch.where(s < kth, torch.tensor(float("-inf"), dtype=s.dtype, device=s.device), s)
            warpers.append(_topk)
        return self.sample(input_ids, logits_processor=logits_processor, logits_warper=warpers, max_length=max_length, **kw)
