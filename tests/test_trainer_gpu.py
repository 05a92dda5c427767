"""A real `transformers.Trainer` stepping the MI355X LibraTrainWrapper: what /root/reference/train.py:108-118 does
with /root/reference/trainer.py:8-85 - a Trainer subclass whose only change is the optimizer's parameter grouping
(`get_decay_parameter_names` with LlamaRMSNorm in the no-decay layer list) - on the tiny model with the recipe's switches
(`bf16: True`, `gradient_checkpointing: True`, `max_grad_norm: 1.0`, AdamW; libra_pretrain.yaml:83-120), fed by a collater
shaped like the reference's (`{"samples": {key: [per-sample values]}}`, caption_datasets.py:112-118)."""
import pytest
import torch
from torch import nn

from helpers import word_level_tokenizer

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _tiny_wrapper():
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    from libra_amd.libra import ImageTokenizer, LibraConfig, LibraForCausalLM, LibraTokenizer, LibraTrainWrapper
    from oracle import vit_oracle as VO, vq_oracle as QO
    torch.manual_seed(0)
    vcfg = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14)
    clip = CLIPVisionModel(CLIPVisionConfig(**vcfg))
    clip.load_state_dict(dict(VO.random_vit_state_dict(hidden=128, inter=256, layers=3, patch=14, image=56)), strict=False)
    clip = clip.to(BF).cuda()
    tt = word_level_tokenizer("a photo of cat dog on the grass some text follows here".split(), model_max_length=64)
    L = 18
    tcfg = {"params": {"ddconfig": {"encoder_name": "clip_tiny", "select_layer": [-2, -3]}, "embed_dim": 32,
                       "codebook_size": 512, "num_codebook": 2}, "max_vision_token_length": L}
    it = ImageTokenizer(tcfg, token_offset=tt.vocab_size, vision_model=clip)
    it.model.load_state_dict(QO.random_vq_state_dict(c_feat=256, embed_dim=32), strict=False)
    it = it.to(BF).cuda()
    tok = LibraTokenizer(text_tokenizer=tt, image_tokenizer=it).cuda()
    cfg = LibraConfig(vocab_size=tt.vocab_size, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      max_position_embeddings=64, vision_vocab_size=514, max_vision_token_length=L, contiguous_signal_size=256,
                      image_feature_resolution=4)
    lm = LibraForCausalLM(cfg)
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "bridge" in n and n.endswith("weight_B"):
                p.normal_(0, 0.02)
    lm = lm.to(BF).cuda()
    return LibraTrainWrapper({"pretrained": None, "model_kwargs": {"frozen_language": True}}, module=lm, tokenizer=tok), L


class _Captions(torch.utils.data.Dataset):
    def __init__(self, L, n=12):
        g = torch.Generator().manual_seed(1)
        ph = " ".join(["<img_ph>"] * L)
        texts = [(f"a photo of cat {ph} some text follows", 5), (f"{ph} dog on the grass", 1), (f"the {ph} text here follows a cat", 2)]
        self.items = []
        for i in range(n):
            text, first = texts[i % 3]
            self.items.append({"vision": torch.randn(3, 56, 56, generator=g), "language": text, "contiguous_ignore_sign": i % 3 == 1,
                               "label_mask_position_map": [(first + L, first + 1 + L)]})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _collater(samples):
    out = {k: [] for k in samples[0]}
    for s in samples:
        for k, v in s.items():
            out[k].append(v)
    return {"samples": out}


def test_hf_trainer_steps_the_wrapper_with_gradient_checkpointing(tmp_path):
    from transformers import Trainer, TrainingArguments
    from transformers.trainer import get_parameter_names
    from libra_amd.libra import LlamaRMSNorm

    class LibraShapedTrainer(Trainer):
        """trainer.py:8-85: weight decay skips LayerNorm / LlamaRMSNorm weights and biases."""
        def get_decay_parameter_names(self, model):
            names = get_parameter_names(model, [nn.LayerNorm, LlamaRMSNorm])
            return [n for n in names if "bias" not in n]

    model, L = _tiny_wrapper()
    lm = model.module
    assert not lm.model.gradient_checkpointing
    before = {n: p.detach().clone() for n, p in lm.named_parameters()}
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=3, max_steps=4, learning_rate=2e-3,
                             weight_decay=0.01, adam_beta2=0.99, max_grad_norm=1.0, bf16=True, gradient_checkpointing=True,
                             lr_scheduler_type="constant", logging_strategy="steps", logging_steps=1, save_strategy="no",
                             report_to=[], remove_unused_columns=False, dataloader_num_workers=0, dataloader_pin_memory=False,
                             seed=0)
    trainer = LibraShapedTrainer(model=model, args=args, train_dataset=_Captions(L), data_collator=_collater)
    trainer.train()
    losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert len(losses) == 4 and all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    assert lm.model.gradient_checkpointing                      # TrainingArguments.gradient_checkpointing reached the engine
    # the parameter grouping of trainer.py: norm weights in the no-decay group, 2-D vision weights in the decay group
    groups = trainer.optimizer.param_groups
    decay, no_decay = {id(p) for p in groups[0]["params"]}, {id(p) for p in groups[1]["params"]}
    assert groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0
    named = dict(model.named_parameters())
    norm_w = [n for n in named if n.endswith("layernorm.weight") or n.endswith("norm.weight")]
    assert norm_w and all(id(named[n]) in no_decay for n in norm_w if named[n].requires_grad)
    assert all(id(named[n]) in decay for n in named if named[n].requires_grad and named[n].ndim == 2)
    # frozen language stream untouched, trainable vision stream moved
    moved = [n for n, p in lm.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert moved and all("vision" in n for n in moved), moved[:5]
    assert all(not p.requires_grad for n, p in lm.named_parameters() if "vision" not in n)
