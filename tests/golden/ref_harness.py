"""Import harness for the *reference* Libra sources at /root/reference.

TEST INFRASTRUCTURE — runs only in the build container (the reference does not
exist on the GPU box).  It is used by ``make_golden.py`` to emit tensor-only
fixtures and by nothing else.

The reference's own ``import libra.models`` fails under the installed
transformers, so empty package skeletons are registered in
``sys.modules`` (``__path__`` pointing at the real directories, no
``__init__.py`` executed) and a handful of symbols that the leaf files import
but never use on the hot path are stubbed.  No reference source is copied.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("LIBRA_REFERENCE", "/root/reference")


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install():
    if "libra.models.libra.modeling_libra" in sys.modules:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    sys.dont_write_bytecode = True
    import torch  # noqa: F401
    import transformers

    # ---- package skeletons (no __init__ side effects) ----
    _pkg("libra", f"{REF}/libra")
    _pkg("libra.common", f"{REF}/libra/common")
    _pkg("libra.models", f"{REF}/libra/models")
    clip = _pkg("libra.models.clip", f"{REF}/libra/models/clip")
    _pkg("libra.models.llama", f"{REF}/libra/models/llama")
    _pkg("libra.models.libra", f"{REF}/libra/models/libra")
    _pkg("libra.models.libra.taming", f"{REF}/libra/models/libra/taming")
    _pkg("libra.models.libra.taming.models", f"{REF}/libra/models/libra/taming/models")
    _pkg("libra.models.libra.taming.modules", f"{REF}/libra/models/libra/taming/modules")
    _pkg("libra.models.libra.taming.modules.quantization",
         f"{REF}/libra/models/libra/taming/modules/quantization")
    _pkg("libra.models.libra.taming.modules.diffusionmodules",
         f"{REF}/libra/models/libra/taming/modules/diffusionmodules")

    # ---- stubs for imports that are dead on the hot path ----
    if "transformers.onnx" not in sys.modules:
        onnx = types.ModuleType("transformers.onnx")
        onnx.OnnxConfig = type("OnnxConfig", (), {})
        sys.modules["transformers.onnx"] = onnx
        transformers.onnx = onnx

    tok = types.ModuleType("libra.models.libra.tokenization_libra")
    tok.LibraTokenizer = type("LibraTokenizer", (), {})
    sys.modules["libra.models.libra.tokenization_libra"] = tok

    from transformers.modeling_utils import PreTrainedModel
    mlu = types.ModuleType("libra.models.libra.modeling_libra_utils")
    mlu.BaseLibraPreTrainedModel = PreTrainedModel
    sys.modules["libra.models.libra.modeling_libra_utils"] = mlu

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvt.Compose = lambda fs: fs
        tvt.Normalize = lambda mean, std: (mean, std)
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt

    # registry is a leaf (no heavy imports)
    importlib.import_module("libra.common.registry")

    cfg = importlib.import_module("libra.models.clip.configuration_clip")
    mc = importlib.import_module("libra.models.clip.modeling_clip")
    clip.CLIPVisionModel = mc.CLIPVisionModel
    clip.CLIPVisionConfig = cfg.CLIPVisionConfig

    class _FakeProcessor:
        image_mean = [0.48145466, 0.4578275, 0.40821073]
        image_std = [0.26862954, 0.26130258, 0.27577711]

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

    clip.CLIPImageProcessor = _FakeProcessor


def clip_modules():
    install()
    cfg = importlib.import_module("libra.models.clip.configuration_clip")
    mc = importlib.import_module("libra.models.clip.modeling_clip")
    return cfg, mc


def vq_modules():
    install()
    vq = importlib.import_module("libra.models.libra.taming.models.vqgan")
    lfq = importlib.import_module(
        "libra.models.libra.taming.modules.quantization.lookup_free_quantization")
    it = importlib.import_module("libra.models.libra.image_tokenizer")
    return vq, lfq, it


def libra_modules():
    install()
    cfg = importlib.import_module("libra.models.libra.configuration_libra")
    ml = importlib.import_module("libra.models.libra.modeling_libra")
    ll = importlib.import_module("libra.models.llama.modeling_llama")
    return cfg, ml, ll
