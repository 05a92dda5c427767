"""Which physical compute units does bit i of a hipExtStreamCreateWithCUMask mask enable on this GPU?  For a few single-bit and
block masks: launch 1024 one-per-CU workgroups on the masked stream and list the distinct (XCC, SE, SH, CU) they ran on."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libra_amd import _lib, kernels as K

lib = _lib.lib()
NW = (K.cu_count() + 31) // 32


def where(bits):
    mask = (ctypes.c_uint32 * NW)()
    for b in bits:
        mask[b >> 5] |= 1 << (b & 31)
    h = ctypes.c_void_p()
    _lib.check(lib.libra_stream_create_cu_mask(mask, NW, ctypes.byref(h)), "stream_create_cu_mask")
    out = torch.zeros(2 * 1024, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(lib.libra_debug_cu_map(out.data_ptr(), 1024, 40, h.value), "debug_cu_map")
    torch.cuda.synchronize()
    _lib.check(lib.libra_stream_destroy(h.value), "stream_destroy")
    w = out.cpu().numpy().astype("int64") & 0xffffffff
    seen = sorted({(int(w[2 * i + 1]) & 15, (int(w[2 * i]) >> 13) & 7, (int(w[2 * i]) >> 12) & 1, (int(w[2 * i]) >> 8) & 15) for i in range(1024)})
    return seen


print("cus", K.cu_count(), "mask words", NW)
for b in (0, 1, 2, 3, 7, 8, 9, 16, 31, 32, 33, 64, 128, 255):
    print(json.dumps({"bit": b, "xcc_se_sh_cu": where([b])}))
full = where(range(K.cu_count()))
print(json.dumps({"all_bits": len(full), "per_xcc": {x: sum(1 for s in full if s[0] == x) for x in range(8)}}))
for name, bits in (("bits 0..31", range(32)), ("bits 0..7", range(8)), ("every 8th bit", range(0, 256, 8))):
    s = where(bits)
    print(json.dumps({"mask": name, "n": len(s), "per_xcc": {x: sum(1 for t in s if t[0] == x) for x in range(8)}}))
