"""Shorten the kernel names of a rocprofv3 kernel_stats.csv (dev tool)."""
import csv
import sys
w = csv.writer(sys.stdout)
for i, row in enumerate(csv.reader(open(sys.argv[1]))):
    if i:
        row[0] = row[0].split('(')[0][:70]
    w.writerow(row)
