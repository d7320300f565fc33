"""Sums every counter of a rocprofv3 --pmc output directory over the path kernels' dispatches: python3 tools/pmc_summary.py <dir> ..."""
import csv, collections, sys, glob
for d in sys.argv[1:]:
    agg = collections.defaultdict(float)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_path" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(f"{k:32s} {v:.6g}")
