"""Audit compiled kernels for compiler-inserted `s_waitcnt vmcnt` / scratch traffic inside loops (they drain the hidden
LDS-DMA queue, see hip_common.hpp glds16), for ANY scratch use (spills, or a kernel-argument array indexed dynamically and
therefore copied to private memory) and for compiler uses of M0 (the LDS-DMA asm owns it).  usage: audit_waits.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
func, in_asm, in_loop = None, False, False
loop_lbl = re.compile(r"^\.LBB\d+_\d+:.*(Loop Header|in Loop)")
any_lbl = re.compile(r"^\.LBB\d+_\d+:")
for i, l in enumerate(lines, 1):
    s = l.strip()
    m = re.match(r"^(_Z\w+):", l)
    if m:
        func = m.group(1)[:70]; in_loop = False
        continue
    if "#ASMSTART" in s: in_asm = True
    elif "#ASMEND" in s: in_asm = False
    if any_lbl.match(l):
        in_loop = bool(loop_lbl.match(l))
    if not in_asm and (s.startswith("scratch_") or re.search(r"[ ,]m0\b", s)):
        print(f"{func}  L{i}: [anywhere] {s[:80]}")
    if in_asm or not in_loop:
        continue
    if ("s_waitcnt" in s and "vmcnt" in s) or s.startswith("scratch_") or s.startswith("global_load") or s.startswith("buffer_load"):
        print(f"{func}  L{i}: {s[:80]}")
