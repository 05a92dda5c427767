"""Effective shader clock per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run: cycles / duration."""
import collections
import csv
import glob
import sys
d = sys.argv[1]
dur, cyc = collections.defaultdict(list), collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "libra" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0].split("::")[-1][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "libra" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Kernel_Name"].split("(")[0].split("::")[-1][:48]].append(float(r["Counter_Value"]))
for k in sorted(dur):
    if k in cyc:
        us, c = sum(dur[k]) / len(dur[k]), sum(cyc[k]) / len(cyc[k])
        print(f"{k:48s} {us:9.1f} us  {c:12.0f} GRBM_GUI_ACTIVE  -> {c / us / 1e3:5.2f} GHz  (n={len(dur[k])})")
