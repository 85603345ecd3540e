"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name (dev tool)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        key = (row['Kernel_Name'].split('(')[0][:60], row['Counter_Name'])
        acc[key][0] += float(row['Counter_Value'])
        acc[key][1] += 1
print('kernel,counter,mean_per_dispatch,dispatches')
for (k, c), (s, n) in sorted(acc.items()):
    print(f'"{k}",{c},{s / n:.1f},{n}')
