"""Which HSA queue / HIP stream every kernel of a rocprofv3 --kernel-trace run used: python scripts/queue_map.py <dir>"""
import collections
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
busy = collections.defaultdict(float)
names = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = (r["Queue_Id"], r.get("Stream_Id", "?"))
    busy[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    names[k][r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:36]] += 1
for k in sorted(busy, key=lambda k: -busy[k]):
    print(k, round(busy[k], 2), "ms", names[k].most_common(4))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows if "pad_normalize" in r["Kernel_Name"]]
print("step ms:", [round((b - a) / 1e6, 2) for a, b in zip(st, st[1:])])
if len(st) >= 3:
    a, b = st[-2], st[-1]
    step = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
    main_q = "stream 0"
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in step if r.get("Stream_Id") == "0")
    print("main-stream busy ms:", round(sum(e - s for s, e, _ in iv) / 1e6, 3), "kernels", len(iv), "span", round((iv[-1][1] - iv[0][0]) / 1e6, 3))
    gaps = sorted(((s2 - e1, n1[:50], n2[:50]) for (s1, e1, n1), (s2, e2, n2) in zip(iv, iv[1:]) if s2 > e1), reverse=True)
    print("main-queue idle ms:", round(sum(g[0] for g in gaps) / 1e6, 3), "largest gaps (us):")
    for g in gaps[:8]:
        print("   ", round(g[0] / 1e3, 1), g[1].replace("(anonymous namespace)::", ""), "->", g[2].replace("(anonymous namespace)::", ""))
