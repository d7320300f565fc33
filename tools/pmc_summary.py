"""Sums every counter of a rocprofv3 --pmc output directory over the path kernels' dispatches (PMC_KERNEL=<substring> selects other kernels): python3 tools/pmc_summary.py <dir> ..."""
import csv, collections, os, sys, glob
KERNEL = os.environ.get("PMC_KERNEL", "k_path")
for d in sys.argv[1:]:
    agg = collections.defaultdict(float)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(f"{k:32s} {v:.6g}")
