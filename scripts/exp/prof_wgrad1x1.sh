cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p1
rm -rf $O; mkdir -p $O
cd $R
for t in "128,128" "256,256" "128,256"; do
  BD_WGRAD1R_TILE=$t rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$t -- python3 scripts/micro_wgrad1x1_ring.py 1 > $O/out_$t.txt 2>&1
  f=$(find $O/kt_$t -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per (kernel, grid) average duration
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "wgrad" not in n: continue
    key = (n.split("(")[0][-60:], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("LDS_Block_Size") or "")
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== tile", sys.argv[2])
for k, v in agg.items():
    print(f"{k[0]:62s} grid {k[1]:>8s} n={len(v):3d} avg {sum(v)/len(v):8.1f} us  min {min(v):8.1f}")
PY
done
