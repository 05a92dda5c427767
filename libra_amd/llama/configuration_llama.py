"""LlamaConfig — the base class of LibraConfig, as in the reference
(/root/reference/libra/models/llama/configuration_llama.py:84-117: the reference carries its own copy of the LLaMA
configuration, pinned to transformers 4.38; `transformers.LlamaConfig` of the installed 5.x adds rope-parameter
validation this model does not use)."""
from transformers import PretrainedConfig


class LlamaConfig(PretrainedConfig):
    model_type = "llama"
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, hidden_act="silu", max_position_embeddings=2048, initializer_range=0.02,
                 rms_norm_eps=1e-6, use_cache=True, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                 tie_word_embeddings=False, **kwargs):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        super().__init__(pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                         tie_word_embeddings=tie_word_embeddings, **kwargs)
