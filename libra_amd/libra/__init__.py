from .clip_encoder import CLIPVisionTower
from .image_tokenizer import ImageTokenizer
from .lookup_free_quantization import LFQ
from .vqgan import VQModel
from .configuration_libra import LibraConfig
from .modeling_libra import LibraForCausalLM, LlamaRMSNorm
from .tokenization_libra import apply_freeze_policy, assemble_inputs, get_labels

__all__ = ["CLIPVisionTower", "ImageTokenizer", "LFQ", "VQModel", "LibraConfig", "LibraForCausalLM", "LlamaRMSNorm"]
