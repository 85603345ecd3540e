"""dev tool: the top rows of a rocprofv3 kernel_stats.csv (name shortened, calls, total ms, average us, percent)."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(files[0])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):8d} {int(r['TotalDurationNs']) / 1e6:9.1f} ms {float(r['AverageNs']) / 1e3:8.2f} us {float(r['Percentage']):6.2f} %")
