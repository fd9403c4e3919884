"""Mean PMC counter values per dispatch for the kernels whose name contains one of the given substrings.
usage: python pmc_by_kernel.py <dir with rocprofv3 --pmc csv output> <substring> [<substring> ...]"""
import csv, glob, os, sys
from collections import defaultdict
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        for key in sys.argv[2:]:
            if key in name:
                a = acc[key][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for key, d in acc.items():
    print("==", key)
    for cn, (tot, n) in sorted(d.items()):
        print("  %-28s %14.1f  (%d dispatches)" % (cn, tot / n, n))
