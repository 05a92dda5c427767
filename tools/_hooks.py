"""Tools-only: point the package's loader at the bench-hooks build of the kernel library.

`make -C libra_amd/csrc bench-hooks` builds libra_amd/lib/liblibra_hip_hooks.so from the same sources with
-DLIBRA_BENCH_HOOKS: the only build that reads LIBRA_GEMM_KERNEL (forces a GEMM tile structure for A/B timing).  The product
library reads no environment variable (tests/test_cabi_cpu.py), and nothing under libra_amd/ ever loads the hooks build -
a tool opts in by importing this module BEFORE its first kernel call:

    import _hooks          # noqa: F401   (tools/ is on sys.path when a tool runs as a script)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libra_amd import _lib  # noqa: E402

_HOOKS = os.path.join(os.path.dirname(_lib.LIB_PATH), "liblibra_hip_hooks.so")
if os.environ.get("LIBRA_GEMM_KERNEL"):
    if not os.path.exists(_HOOKS):
        raise SystemExit(f"{_HOOKS} not found: build it with `make -C libra_amd/csrc bench-hooks`")
    if _lib._lib is not None:
        raise SystemExit("tools/_hooks.py must be imported before the first kernel call")
    _lib.LIB_PATH = _HOOKS
