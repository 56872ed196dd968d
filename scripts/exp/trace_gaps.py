"""Timeline analysis of a rocprofv3 --kernel-trace CSV (last step of the run): time when NO kernel runs, time when only 'small' kernels
(< 256 workgroups) run, and the small kernels themselves -- what a saturated step can still hide.  python scripts/exp/trace_gaps.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nwg = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(wg, 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nwg, r["Kernel_Name"], r["Queue_Id"]))
ev.sort()
# last step: from the last sgd_kernel backwards to the previous one
sgd = [i for i, e in enumerate(ev) if "sgd_kernel" in e[3]]
a, b = sgd[-2], sgd[-1]
step = ev[a + 1:b + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"step: {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
pts = []
for s, e, n, name, q in step:
    pts.append((s, 1, n >= 256)); pts.append((e, -1, n >= 256))
pts.sort()
big = small = 0
idle = only_small = 0
last = t0
for t, d, isbig in pts:
    if big == 0 and small == 0: idle += t - last
    elif big == 0: only_small += t - last
    last = t
    if isbig: big += d
    else: small += d
print(f"no kernel running: {idle / 1e6:.3f} ms; only kernels of < 256 workgroups running: {only_small / 1e6:.3f} ms")
# which small kernels run while no big one does
alone = {}
bigs = sorted((s, e) for s, e, n, name, q in step if n >= 256)
def big_cover(s, e):
    c = 0
    for bs, be in bigs:
        if be <= s: continue
        if bs >= e: break
        c += min(e, be) - max(s, bs)
    return c
for s, e, n, name, q in step:
    if n < 256:
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        alone[k] = alone.get(k, 0) + (e - s) - min(e - s, big_cover(s, e))
print("small kernels, time not covered by any big kernel:")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v / 1e6:7.3f} ms  {k}")
agg = {}
for s, e, n, name, q in step:
    if n < 256:
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        a = agg.setdefault(k, [0, 0, n]); a[0] += e - s; a[1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:15]:
    print(f"  {v[0] / 1e6:7.3f} ms  x{v[1]:3d}  wgs={v[2]:5d}  {k}")
