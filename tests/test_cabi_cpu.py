"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/libra_hip.h declares (no compute without a GPU); the host loader's prototype table matches the header."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "libra_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(libra_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = args
    return out


@pytest.fixture(scope="module")
def lib_path():
    from libra_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libra_amd", "csrc"), "-j8"], check=True)
    return _lib.LIB_PATH


def test_header_declares_the_expected_surface():
    d = _declared()
    for name in ("libra_gemm_bf16_nt", "libra_layernorm_fwd", "libra_layernorm_bwd", "libra_vit_attn_fwd",
                 "libra_vit_attn_bwd", "libra_lfq_encode", "libra_patch_im2col", "libra_gemm_bf16_nt_grouped"):
        assert name in d
    assert len(d) >= 16


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/libra_hip.h but not exported"
    lib.libra_hip_abi_version.restype = ctypes.c_int
    from libra_amd import _lib as _host
    assert lib.libra_hip_abi_version() == _host.ABI_VERSION >= 4


def test_host_prototypes_match_header(lib_path):
    from libra_amd import _lib
    d = _declared()
    assert set(d) == set(_lib.SIGNATURES), set(d) ^ set(_lib.SIGNATURES)
    for name, args in d.items():
        assert len(args) == len(_lib.SIGNATURES[name]), (name, len(args), len(_lib.SIGNATURES[name]))
        for a, ct in zip(args, _lib.SIGNATURES[name]):
            if "libra_gemm_problem*" in a:                      # the one struct of the ABI: a typed pointer, layout checked below
                assert ct is ctypes.POINTER(_lib.GemmProblem), (name, a)
            elif "*" in a:
                assert ct is ctypes.c_void_p, (name, a)
            elif a.startswith("size_t"):
                assert ct is ctypes.c_size_t, (name, a)
            elif a.startswith("int64_t"):
                assert ct is ctypes.c_int64, (name, a)
            elif a.startswith("float"):
                assert ct is ctypes.c_float, (name, a)
            elif a.startswith("int "):
                assert ct is ctypes.c_int, (name, a)
    _lib.load()


def test_gemm_problem_struct_matches_header():
    """`libra_gemm_problem` (the per-problem record of libra_gemm_bf16_multi): the ctypes Structure has the header's fields, in the
    header's order, with the header's types - and therefore its layout (both sides use the platform's natural alignment)."""
    from libra_amd import _lib
    src = open(os.path.join(ROOT, "include", "libra_hip.h")).read()
    body = re.search(r"typedef struct libra_gemm_problem \{(.*?)\} libra_gemm_problem;", src, re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const void\*|void\*|const int32_t\*|int64_t|int32_t|float) (.*)", decl)
        assert m, decl
        for nm in m.group(2).split(","):
            fields.append((nm.strip(), m.group(1)))
    ctype = {"const void*": ctypes.c_void_p, "void*": ctypes.c_void_p, "const int32_t*": ctypes.c_void_p, "int64_t": ctypes.c_int64,
             "int32_t": ctypes.c_int32, "float": ctypes.c_float}
    assert [(n, ctype[t]) for n, t in fields] == list(_lib.GemmProblem._fields_)
    assert ctypes.sizeof(_lib.GemmProblem) == 192          # = static_assert in gemm_bf16_multi.hip
    assert int(re.search(r"#define LIBRA_GEMM_MULTI_MAX (\d+)", src).group(1)) == _lib.GEMM_MULTI_MAX


def test_shipped_library_reads_no_environment_variable(lib_path):
    """Round-3 review: experiment switches (`LIBRA_ATTN_FWD`, `LIBRA_ATTN_DBG`, `LIBRA_ATTN_DKV`, `LIBRA_GEMM_KERNEL`) selected
    structures - some with deliberately wrong results - inside the product library.  The shipped .so must not import getenv at
    all and must carry no LIBRA_* variable name; a caller that wants a specific GEMM tile structure says so through
    libra_gemm_bf16_nt_tile."""
    allow = set()                                                     # environment variables the product library may read: none
    nm = subprocess.run(["nm", "-D", "--undefined-only", lib_path], capture_output=True, text=True, check=True).stdout
    env_syms = [l.split()[-1] for l in nm.splitlines() if re.search(r"\b(secure_)?getenv\b", l)]
    assert not env_syms, f"liblibra_hip.so imports {env_syms}"
    blob = open(lib_path, "rb").read()
    names = set(m.decode() for m in re.findall(rb"LIBRA_[A-Z][A-Z0-9_]{2,}(?=\x00)", blob))
    # (error-code / flag macros are compile-time integers and leave no strings; anything that remains would be a getenv key)
    assert names <= allow, f"environment-variable-like names in the shipped library: {sorted(names - allow)}"
    srcs = os.path.join(ROOT, "libra_amd", "csrc")
    for f in sorted(os.listdir(srcs)):
        if f.endswith((".hip", ".hpp")):
            assert "getenv" not in open(os.path.join(srcs, f)).read(), f"{f}: getenv in a product source"


def test_product_path_has_no_oracle_or_cpu_fallback():
    """The shipped package must never import the oracle or fall back to torch math."""
    for dp, _, files in os.walk(os.path.join(ROOT, "libra_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert "oracle" not in s.replace("# oracle", ""), (f, "imports/mentions the oracle")


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from transformers import CLIPVisionConfig
    from libra_amd.clip import CLIPVisionModel
    m = CLIPVisionModel(CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1,
                                         num_attention_heads=2, image_size=28, patch_size=14)).to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 28, 28))
