"""Average the rocprofv3 counter_collection.csv rows per (kernel, counter) over dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "libra" not in k:
                continue
            name = k.split("(")[0].split("::")[-1][:40]
            a = acc[(name, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
kern = sorted({k for k, _ in acc})
for k in kern:
    print(k)
    for (kk, c), (s, n) in sorted(acc.items()):
        if kk == k:
            print(f"   {c:32s} {s / n:16.0f}   (n={n})")
