"""Summarise rocprofv3 --pmc passes: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, os, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("dptk::", "")
            if k.startswith("void "): k = k[5:]
            k = k.split("(")[0]   # k_trav<0, false, 4, false>
            a = agg[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    print(k)
    for c in sorted(agg[k]):
        s, n = agg[k][c]
        print("   %-32s mean/dispatch %.4g   total %.5g   (n=%d)" % (c, s / n, s, n))
