from .clip_encoder import CLIPVisionTower
from .image_tokenizer import ImageTokenizer
from .lookup_free_quantization import LFQ
from .vqgan import VQModel
from .configuration_libra import LibraConfig
from .modeling_libra import LibraForCausalLM, LibraTrainWrapper, LlamaRMSNorm
from .tokenization_libra import LibraTokenizer, apply_freeze_policy, assemble_inputs, get_labels, plan_assembly

__all__ = ["CLIPVisionTower", "ImageTokenizer", "LFQ", "VQModel", "LibraConfig", "LibraForCausalLM", "LibraTrainWrapper",
           "LibraTokenizer", "LlamaRMSNorm", "apply_freeze_policy", "assemble_inputs", "get_labels", "plan_assembly"]
