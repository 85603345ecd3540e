cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/scripts/microbench.py --gen 200 > /dev/null 2>&1
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:8.2f} us", f"{float(r['Percentage']):6.2f}%")
PY
