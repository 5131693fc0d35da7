"""average of every counter over the dispatches of one kernel in a rocprofv3 counter_collection.csv
usage: pmc_avg.py <csv> <kernel name substring> [Grid_Size]   (Grid_Size = threads of the launch: picks one launch shape of a kernel)"""
import csv, sys, collections
f, pat = sys.argv[1], sys.argv[2]
grid = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: [0.0, 0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if pat in row.get("Kernel_Name", "") and (grid is None or row.get("Grid_Size") == grid):
            k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in acc.items():
    print("  %-40s avg/dispatch %14.1f  (n=%d)" % (k, s / n, n))
