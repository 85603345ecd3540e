cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
ACMI_FFN2_HALF=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$mode -- python $GRAFT_REPO_ROOT/scripts/microbench.py --gen 200 > /dev/null 2>&1
f=$(find /tmp/p$mode -name "*kernel_stats.csv" | head -1)
echo "== HALF=$mode"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:8.2f} us", f"{float(r['Percentage']):6.2f}%")
PY
done
