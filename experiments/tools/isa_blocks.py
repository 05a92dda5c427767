"""Per-kernel summary of a hipcc `-S --cuda-device-only` listing: registers, spills, and the basic blocks that carry the MFMAs
(instruction mix per block) - a CPU-side check that a kernel's main loop is the instruction stream it was written to be.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only x.hip -o x.s && python tools/isa_blocks.py x.s [name-filter] [min-mfma]"""
import re
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split("\n")
    name, body = None, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1); body[name] = []
        elif name is not None:
            body[name].append(l)
            if l.strip() == "s_endpgm":
                name = None
    meta = {}
    cur = None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1); meta[cur] = {}
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", l)
        if m and cur:
            meta[cur][m.group(1)] = int(m.group(2))
    for k, ls in body.items():
        if filt not in k:
            continue
        print(k[:100], meta.get(k, {}))
        blocks, cur, label = [], [], "entry"
        for l in ls:
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                blocks.append((label, cur)); cur, label = [], m.group(1)
            elif re.match(r"\s+[a-z]", l) and not l.strip().startswith("."):
                cur.append(l.strip())
        blocks.append((label, cur))
        tot = sum(len(b) for _, b in blocks)
        print(f"  instructions {tot}, blocks {len(blocks)}")
        for label, b in blocks:
            op = [i.split()[0] for i in b]
            n_mfma = sum(o.startswith("v_mfma") for o in op)
            if n_mfma < min_mfma:
                continue
            c = lambda pre: sum(o.startswith(pre) for o in op)
            waits = [i for i in b if i.startswith("s_waitcnt")]
            print(f"  {label:12s} n={len(b):5d} mfma={n_mfma:3d} ds_read={c('ds_read'):3d} ds_write={c('ds_write'):3d} lds_dma={sum('lds' in i and i.startswith(('global_load','buffer_load')) for i in b):3d} "
                  f"vmem={c('global_')+c('buffer_'):3d} salu={c('s_')-c('s_waitcnt')-c('s_barrier')-c('s_nop'):3d} valu={c('v_')-n_mfma:3d} "
                  f"barrier={c('s_barrier'):2d} nop={c('s_nop'):2d} scratch={c('scratch_'):2d} waits={len(waits)} vm0={sum('vmcnt(0)' in w for w in waits)}")


if __name__ == "__main__":
    main()
