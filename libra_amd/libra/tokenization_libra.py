"""Tensor-assembly half of LibraTokenizer.forward and the label builder — mirrors of
/root/reference/libra/models/libra/tokenization_libra.py:250-316 and LibraTrainWrapper.get_labels
(/root/reference/libra/models/libra/modeling_libra.py:1397-1411), plus the freeze policy (:1342-1369).

Integer / indexing glue that runs on whatever device its inputs live on (plain torch indexing — plumbing, no
arithmetic hot path).  The text half of the reference tokenizer (sentencepiece LLaMA tokenizer, prompt templates)
is outside the hot path: callers hand in the text ids with every ``<img_ph>`` already expanded to
``max_vision_token_length`` placeholder slots, exactly what ``self.text_tokenizer(texts, ...)`` returns upstream (:245).
"""
import logging
import os
from typing import Optional, Sequence

import torch
from transformers import BatchEncoding

MAX_TOKEN_LENGTH = 2048                                                         # tokenization_libra.py:15


_PLAN_STREAMS = {}


def plan_assembly(text_ids: torch.Tensor, *, img_ph_token_id: int, side_stream: bool = False):
    """Where the `<img_ph>` placeholders are: (mask [B,S], (batch index, position) of every placeholder in row-major order).
    This is the ONE host-synchronising step of the tensor assembly (a nonzero) and it does not depend on the image encoder's
    output - a training loop calls it BEFORE queueing the encoder and hands the result to assemble_inputs(plan=...), so the host
    never waits for the encoder between the encoder and the decoder (the boolean-mask scatters upstream uses,
    tokenization_libra.py:266,273,292, each hide such a wait: ~2 ms of idle GPU per step at the benchmark shape).
    side_stream: run it on a private stream, so that the host read waits for THIS little work only and not for whatever the
    current stream still holds (the previous step's backward).  Only valid when `text_ids` is already complete - a tensor
    made earlier and synchronised since (the benchmark's static batch), or made on the host (then the upload rides the same stream)."""
    if not (side_stream and text_ids.is_cuda):
        ph = text_ids == img_ph_token_id                                        # :250
        return ph, ph.nonzero(as_tuple=True)
    dev = text_ids.device
    side = _PLAN_STREAMS.get(dev)
    if side is None:
        side = _PLAN_STREAMS[dev] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        ph = text_ids == img_ph_token_id
        pb, ps = ph.nonzero(as_tuple=True)                                      # the host waits for the side stream only
    for t in (ph, pb, ps):
        t.record_stream(cur)                                                    # allocated on the side stream, consumed on the current one
    cur.wait_stream(side)
    return ph, (pb, ps)


def assemble_inputs(text_ids: torch.Tensor, attention_mask: torch.Tensor, image_inputs: Optional[dict], *,
                    img_ph_token_id: int, img_gen_token_id: int, boi_token_id: int, num_codebook: int,
                    max_vision_token_length: int, contiguous_ignore_signs=None, has_image_flag=None,
                    truncation: bool = False, max_length: Optional[int] = None, plan=None) -> dict:
    """-> {"input_ids" [Q,B,S], "attention_mask", "vision_indices", "coninous_signal" [sic]} (same keys, incl. the
    upstream spelling, that LibraTrainWrapper.forward consumes: modeling_libra.py:1425-1430).
    plan: plan_assembly(text_ids, ...) computed earlier (default: computed here)."""
    dev = text_ids.device
    if plan is None and image_inputs is not None:
        plan = plan_assembly(text_ids, img_ph_token_id=img_ph_token_id)
    pb = ps = None                                                              # text-only batch: nothing to place, no host read
    if plan is not None:
        _, (pb, ps) = plan
    gen = text_ids == img_gen_token_id
    ids = text_ids.masked_fill(gen, boi_token_id)                               # :253-254
    ids = ids[None, ...].repeat(num_codebook, 1, 1)                             # :256
    has_images = image_inputs is not None
    if has_images:
        img_ids, feat = image_inputs["input_ids"], image_inputs["encoder_feat"]
        if has_image_flag is not None:                                          # :262-264
            img_ids, feat = img_ids[:, has_image_flag], feat[has_image_flag]
        ids[:, pb, ps] = img_ids.flatten(1, 2)                                  # :266 (a placeholder / image-token count mismatch raises here, as upstream)
    vi = torch.full(attention_mask.shape, max_vision_token_length, dtype=torch.long, device=dev)      # :270
    signal = None
    if has_images:
        L = img_ids.shape[2]
        vi[pb, ps] = torch.arange(L, device=dev).expand(img_ids.shape[1], -1).flatten(0, 1)             # :273
        z = torch.zeros([feat.shape[0], 1, feat.shape[2]], device=feat.device, dtype=feat.dtype)
        cont = torch.cat([z, feat, z], dim=1)                                   # :279-286
        if contiguous_ignore_signs is not None:
            sel = torch.as_tensor(contiguous_ignore_signs, device=cont.device, dtype=torch.bool)
            cont[sel] = 0                                                       # :288-289
        signal = torch.zeros([ids.shape[1], ids.shape[2], cont.shape[-1]], dtype=cont.dtype, device=cont.device)
        signal[pb, ps] = cont.flatten(0, 1)                                     # :291-292
    else:
        vi.masked_fill_(gen, 0)                                                 # :275
    if truncation and max_length is not None:                                   # :296-301
        ids, attention_mask, vi = ids[:, :, :max_length], attention_mask[:, :max_length], vi[:, :max_length]
        if signal is not None:
            signal = signal[:, :max_length]
    return {"input_ids": ids.contiguous(), "attention_mask": attention_mask.contiguous(),
            "vision_indices": vi.contiguous(), "coninous_signal": signal}


def get_labels(inputs: dict, label_mask_position_map: Sequence[Sequence], *, boi_token_id: int, bos_token_id: int):
    """LibraTrainWrapper.get_labels: ids with -100 at padding, BOI, BOS and the given (start, end) spans."""
    ids = inputs["input_ids"]
    # (masked_fill, not boolean-mask assignment: `labels[:, mask] = v` is a hidden nonzero + host synchronisation)
    drop = (inputs["attention_mask"] == 0)[None] | (ids == boi_token_id) | (ids == bos_token_id)
    labels = ids.masked_fill(drop, -100)
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


class LibraTokenizer(torch.nn.Module):
    """The reference's multimodal tokenizer module (tokenization_libra.py:109-316): `.text_tokenizer` (LLaMA tokenizer with the
    `<img_ph>` / `<img_gen>` tokens added, pad = unk, :135-146), `.image_tokenizer` (CLIP ViT -> VQ encode on the gfx950 kernels),
    `.device` / `.dtype` (:126-133), and `forward(samples, **tokenizer_kwargs)` -> the four tensors LibraTrainWrapper.forward
    consumes (:305-316).  The sample parsing follows :170-219; the tensor assembly is `assemble_inputs` above.

    `pretrained_model_path` is the checkpoint directory (tokenizer files + `vision_tokenizer_config.yaml`, :149-160).
    Pre-built tokenizers can be injected instead (`text_tokenizer=`, `image_tokenizer=`): no LLaMA `tokenizer.model` exists
    in the offline build / test environments, so tests pass a word-level `PreTrainedTokenizerFast`."""

    def __init__(self, pretrained_model_path=None, vision_config_overwrite={}, *, text_tokenizer=None, image_tokenizer=None,
                 **kwargs):
        super().__init__()
        self.raw_output = kwargs.pop("raw_output", False)
        self.text_tokenizer = (self._prepare_text_tokenizer(text_tokenizer) if text_tokenizer is not None
                               else self.init_text_tokenizer(pretrained_model_path, **kwargs))
        self.image_tokenizer_offset = self.text_tokenizer.vocab_size
        self.image_tokenizer = (image_tokenizer if image_tokenizer is not None else
                                self.init_image_tokenizer(pretrained_model_path, self.image_tokenizer_offset,
                                                          vision_config_overwrite))
        L = self.image_tokenizer.max_vision_token_length
        self.register_buffer("img_indices_ph", torch.arange(0, L, dtype=torch.long)[None, :])
        self.num_codebook = self.image_tokenizer.num_codebook

    @property
    def device(self):
        return self.image_tokenizer.device

    @property
    def dtype(self):
        return self.image_tokenizer.dtype

    @staticmethod
    def _prepare_text_tokenizer(tok):
        tok.add_tokens("<img_ph>")
        tok.add_tokens("<img_gen>")
        tok.img_ph_token_id = tok.convert_tokens_to_ids("<img_ph>")
        tok.img_gen_token_id = tok.convert_tokens_to_ids("<img_gen>")
        tok.pad_token = tok.unk_token                                           # :143
        return tok

    @classmethod
    def init_text_tokenizer(cls, pretrained_model_path, **kwargs):
        from transformers import LlamaTokenizerFast
        return cls._prepare_text_tokenizer(LlamaTokenizerFast.from_pretrained(pretrained_model_path, **kwargs))

    @classmethod
    def init_image_tokenizer(cls, pretrained_model_path, offset, vision_config_overwrite={}):
        import yaml
        from .image_tokenizer import ImageTokenizer
        with open(os.path.join(pretrained_model_path, "vision_tokenizer_config.yaml")) as f:
            config = yaml.safe_load(f)
        if config.get("ckpt_path") is not None:
            config["ckpt_path"] = os.path.join(pretrained_model_path, config["ckpt_path"])
        params = config["params"]
        if params.get("ckpt_path") is not None:
            params["ckpt_path"] = os.path.join(pretrained_model_path, params["ckpt_path"])
        if params["ddconfig"].get("encoder_name") is not None:
            params["ddconfig"]["encoder_name"] = os.path.join(pretrained_model_path, params["ddconfig"]["encoder_name"])
        config.update(**vision_config_overwrite)
        return ImageTokenizer.from_config(config, token_offset=offset)

    @staticmethod
    def _flatten(samples, key):
        out = []
        for sample in samples:
            v = sample.get(key, None)
            if v is not None:
                out += list(v) if isinstance(v, (list, tuple)) else [v]
        return out

    @torch.no_grad()
    def forward(self, samples, **kwargs):
        if not isinstance(samples, (list, tuple)):
            samples = [samples]
        texts = self._flatten(samples, "language")
        images = self._flatten(samples, "vision")
        signs = self._flatten(samples, "contiguous_ignore_sign")
        dev = self.device
        if images:
            images = [img.to(dev) for img in images]
            if images[0].dim() == 3:
                images = torch.stack(images)
            elif images[0].dim() == 4:
                images = torch.cat(images)
            else:
                raise ValueError("Invalid vision inputs.")                      # :196
        else:
            images = None
        if signs:
            signs = torch.cat(signs) if isinstance(signs[0], torch.Tensor) else torch.tensor(signs, device=dev)
        else:
            signs = None
        has_image_flag = samples[-1].get("has_image", None)                      # (:209 reads the LAST sample's key)
        if has_image_flag is not None:
            has_image_flag = torch.tensor(has_image_flag, device=dev, dtype=torch.bool)
        if not texts and images is None:
            raise ValueError("Empty inputs")                                    # :229
        if not texts:
            raise NotImplementedError                                           # :231
        if kwargs.pop("return_tensors", "pt") != "pt":
            raise ValueError("return_tensors = \"pt\" is fixed, and should not be specified to other values.")
        truncation = kwargs.pop("truncation", False)
        max_length = kwargs.pop("max_length", self.text_tokenizer.model_max_length)
        text_inputs = self.text_tokenizer(texts, return_tensors="pt", return_length=True, **kwargs).to(dev)     # :245
        if (text_inputs["length"] > MAX_TOKEN_LENGTH).sum():
            logging.warning("The input token length ecceeds the max number that the model can hold. This may cause "
                            "performance degradation or OOM.")
        # placeholder positions first (the assembly's one host read), THEN the image encoder: the host does not wait for the encoder
        # (text-only batch: no plan, no host read - assemble_inputs has nothing to place)
        plan = None if images is None else plan_assembly(text_inputs["input_ids"], img_ph_token_id=self.text_tokenizer.img_ph_token_id)
        image_inputs = None
        if images is not None:
            image_inputs = self.image_tokenizer(images.to(self.dtype))                                          # :258-259
        out = assemble_inputs(text_inputs["input_ids"], text_inputs["attention_mask"], image_inputs,
                              img_ph_token_id=self.text_tokenizer.img_ph_token_id,
                              img_gen_token_id=self.text_tokenizer.img_gen_token_id,
                              boi_token_id=self.image_tokenizer.boi_token_id, num_codebook=self.image_tokenizer.num_codebook,
                              max_vision_token_length=self.image_tokenizer.max_vision_token_length,
                              contiguous_ignore_signs=signs, has_image_flag=has_image_flag, truncation=truncation,
                              max_length=max_length, plan=plan)
        return out if self.raw_output else BatchEncoding(out)
