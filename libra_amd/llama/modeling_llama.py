"""`LlamaRMSNorm` under the import path the reference's trainer uses (/root/reference/trainer.py:3 imports it from
`libra.models.llama.modeling_llama` to build the no-weight-decay parameter list).  Weight holder: the arithmetic
(/root/reference/libra/models/llama/modeling_llama.py:118-132) is `libra_rmsnorm_routed_fwd/_bwd` in the engine."""
import torch
from torch import nn

from .. import kernels as K


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        """Stand-alone use (inference helper): the unrouted RMSNorm kernel on a bf16 cuda tensor [..., hidden]."""
        if not hidden_states.is_cuda or hidden_states.dtype != torch.bfloat16:
            raise RuntimeError("libra_amd LlamaRMSNorm runs on MI355X bf16 tensors only (no CPU fallback)")
        shape = hidden_states.shape
        x = hidden_states.reshape(-1, shape[-1]).contiguous()
        with torch.no_grad():
            y = K.rmsnorm_routed(x, self.weight.detach().to(torch.bfloat16), None, None, self.variance_epsilon)
        return y.view(shape)


class LlamaRotaryEmbedding(nn.Module):
    """State-dict holder of the reference's per-layer `self_attn.rotary_emb.inv_freq`, a PERSISTENT buffer upstream
    (/root/reference/libra/models/llama/modeling_llama.py:136-139; instantiated per attention layer at :224): a real Libra checkpoint
    carries one per layer and `load_state_dict(strict=True)` expects it.  The cos / sin tables themselves (upstream: non-persistent
    `cos_cached` / `sin_cached`, :146-148, built in fp32 at construction and cast with the model) are built by
    `decoder_engine.rope_tables` from the same formula - from constants, not from this buffer, exactly as upstream's cached tables
    do not follow a later `.to(bfloat16)` of `inv_freq`."""

    def __init__(self, dim, max_position_embeddings=2048, base=10000, device=None):
        super().__init__()
        self.dim, self.max_position_embeddings, self.base = dim, max_position_embeddings, base
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float().to(device) / dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, *args, **kwargs):
        raise RuntimeError("LlamaRotaryEmbedding only owns the `inv_freq` buffer in libra_amd: RoPE runs inside libra_rope_bridge")
