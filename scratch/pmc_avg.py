import csv, sys, collections
f, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if pat in row.get("Kernel_Name", ""):
            k = row["Counter_Name"]; acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in acc.items():
    print("  %-40s avg/dispatch %14.1f  (n=%d)" % (k, s / n, n))
