from .configuration_llama import LlamaConfig
from .modeling_llama import LlamaRMSNorm

__all__ = ["LlamaConfig", "LlamaRMSNorm"]
