"""Tensor-assembly half of LibraTokenizer.forward and the label builder — mirrors of
/root/reference/libra/models/libra/tokenization_libra.py:250-316 and LibraTrainWrapper.get_labels
(/root/reference/libra/models/libra/modeling_libra.py:1397-1411), plus the freeze policy (:1342-1369).

Integer / indexing glue that runs on whatever device its inputs live on (plain torch indexing — plumbing, no
arithmetic hot path).  The text half of the reference tokenizer (sentencepiece LLaMA tokenizer, prompt templates)
is outside the hot path: callers hand in the text ids with every ``<img_ph>`` already expanded to
``max_vision_token_length`` placeholder slots, exactly what ``self.text_tokenizer(texts, ...)`` returns upstream (:245).
"""
from typing import Optional, Sequence

import torch


def assemble_inputs(text_ids: torch.Tensor, attention_mask: torch.Tensor, image_inputs: Optional[dict], *,
                    img_ph_token_id: int, img_gen_token_id: int, boi_token_id: int, num_codebook: int,
                    max_vision_token_length: int, contiguous_ignore_signs=None, has_image_flag=None,
                    truncation: bool = False, max_length: Optional[int] = None) -> dict:
    """-> {"input_ids" [Q,B,S], "attention_mask", "vision_indices", "coninous_signal" [sic]} (same keys, incl. the
    upstream spelling, that LibraTrainWrapper.forward consumes: modeling_libra.py:1425-1430)."""
    dev = text_ids.device
    ph = text_ids == img_ph_token_id                                            # :250
    ids = text_ids.clone()
    gen = ids == img_gen_token_id
    ids[gen] = boi_token_id                                                     # :253-254
    ids = ids[None, ...].repeat(num_codebook, 1, 1)                             # :256
    has_images = image_inputs is not None
    if has_images:
        img_ids, feat = image_inputs["input_ids"], image_inputs["encoder_feat"]
        if has_image_flag is not None:                                          # :262-264
            img_ids, feat = img_ids[:, has_image_flag], feat[has_image_flag]
        ids[:, ph] = img_ids.flatten(1, 2)                                      # :266
    vi = torch.full(attention_mask.shape, max_vision_token_length, dtype=torch.long, device=dev)      # :270
    signal = None
    if has_images:
        L = img_ids.shape[2]
        vi[ph] = torch.arange(L, device=dev).expand(img_ids.shape[1], -1).flatten(0, 1)                 # :273
        z = torch.zeros([feat.shape[0], 1, feat.shape[2]], device=feat.device, dtype=feat.dtype)
        cont = torch.cat([z, feat, z], dim=1)                                   # :279-286
        if contiguous_ignore_signs is not None:
            sel = torch.as_tensor(contiguous_ignore_signs, device=cont.device, dtype=torch.bool)
            cont[sel] = 0                                                       # :288-289
        signal = torch.zeros([ids.shape[1], ids.shape[2], cont.shape[-1]], dtype=cont.dtype, device=cont.device)
        signal[ph] = cont.flatten(0, 1).contiguous()                            # :291-292
    else:
        vi[gen] = 0                                                             # :275
    if truncation and max_length is not None:                                   # :296-301
        ids, attention_mask, vi = ids[:, :, :max_length], attention_mask[:, :max_length], vi[:, :max_length]
        if signal is not None:
            signal = signal[:, :max_length]
    return {"input_ids": ids.contiguous(), "attention_mask": attention_mask.contiguous(),
            "vision_indices": vi.contiguous(), "coninous_signal": signal}


def get_labels(inputs: dict, label_mask_position_map: Sequence[Sequence], *, boi_token_id: int, bos_token_id: int):
    """LibraTrainWrapper.get_labels: ids with -100 at padding, BOI, BOS and the given (start, end) spans."""
    labels = inputs["input_ids"].clone()
    labels[:, inputs["attention_mask"] == 0] = -100
    labels[labels == boi_token_id] = -100
    labels[labels == bos_token_id] = -100
    labels = labels.permute(1, 2, 0)
    for label, spans in zip(labels, label_mask_position_map):
        for start, end in spans:
            label[start:end] = -100
    return labels.permute(2, 0, 1)


def apply_freeze_policy(module: torch.nn.Module, *, frozen_language: bool = False, freeze_vision_value: bool = False,
                        freeze_text_embedding: bool = False, freeze_vision_embedding: bool = False, debug: bool = False):
    """LibraTrainWrapper.__init__ freeze switches (modeling_libra.py:1342-1369), by parameter NAME exactly as upstream."""
    if frozen_language:
        for n, p in module.named_parameters():
            if "vision" not in n:
                p.requires_grad = False
    if freeze_vision_value:
        for n, p in module.named_parameters():
            if "vision_v_proj" in n:
                p.requires_grad = False
    if freeze_text_embedding:
        for n, p in module.named_parameters():
            if ".embed_tokens" in n:
                p.requires_grad = False
    if freeze_vision_embedding:
        for n, p in module.named_parameters():
            if ".vision_embed_tokens" in n:
                p.requires_grad = False
    if debug:
        for n, p in module.named_parameters():
            if "vision_lm_head" not in n:
                p.requires_grad = False
    return module
