"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"][:70], row["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    for (name, ctr), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if "vsde" in name:
            print(f"{name:<72} {ctr:<12} dispatches={n:<4} mean={tot/n:>14.1f}")
