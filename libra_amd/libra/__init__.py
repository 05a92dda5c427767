from .clip_encoder import CLIPVisionTower
from .image_tokenizer import ImageTokenizer
from .lookup_free_quantization import LFQ
from .vqgan import VQModel

__all__ = ["CLIPVisionTower", "ImageTokenizer", "LFQ", "VQModel"]
