cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/fitp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/fitp
EG_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fitp -o f -- python $GRAFT_REPO_ROOT/tools/fit_once.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/fitp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:16]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>5}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:90]}')
PY
