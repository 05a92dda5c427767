# mirrors /root/reference/libra/models/clip/__init__.py
from transformers import CLIPImageProcessor  # CPU preprocessing is out of scope; HF's class is the same vendored code

from .modeling_clip import CLIPVisionConfig, CLIPVisionModel

__all__ = ["CLIPVisionModel", "CLIPImageProcessor", "CLIPVisionConfig"]
from .image_pipeline import CLIPImagePipeline  # noqa: E402,F401
