"""Aggregate FETCH_SIZE / WRITE_SIZE (KB per dispatch, rocprofv3 counter_collection.csv) per kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B -> doubled here; WRITE_SIZE is
reported as is (uncalibrated).  Writes gpurun_out/hbm_<wl>.json with the per-launch mean over the GEMM kernels."""
import csv, glob, json, os, subprocess, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "vit"
acc = collections.defaultdict(lambda: {"n": 0, "rd": 0.0, "wr": 0.0, "nw": 0})
for kind in ("rd", "wr"):
    for f in glob.glob(f"gpurun_out/hbm_{wl}_{kind}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            v = float(row["Counter_Value"]) * 1024.0
            a = acc[k]
            if row["Counter_Name"] == "FETCH_SIZE":
                a["rd"] += 2.0 * v; a["n"] += 1
            elif row["Counter_Name"] == "WRITE_SIZE":
                a["wr"] += v; a["nw"] += 1
tot = sorted(acc.items(), key=lambda kv: -(kv[1]["rd"] + kv[1]["wr"]))
print(f"{'kernel':60s} {'launches':>8s} {'read MB/launch':>15s} {'write MB/launch':>16s}")
g = {"n": 0, "rd": 0.0, "wr": 0.0, "nw": 0}
for k, a in tot[:30]:
    n, nw = max(a["n"], 1), max(a["nw"], 1)
    print(f"{k:60s} {a['n']:8d} {a['rd'] / n / 1e6:15.2f} {a['wr'] / nw / 1e6:16.2f}")
    if "gemm_bf16_nt" in k or "gemm_bf16_multi" in k:        # (round 6: the multi-problem launches are GEMM launches too)
        for f in g: g[f] += a[f]
out = {"workload": wl, "gemm_launches": g["n"], "gemm_read_bytes_per_launch": g["rd"] / max(g["n"], 1),
       "gemm_write_bytes_per_launch": g["wr"] / max(g["nw"], 1),
       "note": "L2 memory-side counters; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE uncalibrated; "
               "includes Infinity-Cache hits"}
out["gemm_bytes_per_launch"] = out["gemm_read_bytes_per_launch"] + out["gemm_write_bytes_per_launch"]
# provenance: which GEMM sources these bytes were measured on (bench.py marks the number stale when they differ from the running tree)
import bench  # noqa: E402
out["gemm_sources_sha"] = bench.gemm_sources_sha()
out["head"] = os.environ.get("LIBRA_HEAD")       # the GPU box has no .git: the visit's command line passes `git rev-parse --short HEAD`
if not out["head"]:
    try:
        out["head"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=bench.ROOT).stdout.strip() or None
    except OSError:
        out["head"] = None
json.dump(out, open(f"gpurun_out/hbm_{wl}.json", "w"), indent=1)
print(json.dumps(out))
