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
