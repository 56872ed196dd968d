"""Regenerates the round-3 measurement tables of DESIGN.md section 5 (between the R3_TABLES markers) from profiles/r03_*.
python scripts/design_tables.py"""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(P("r03_bench.json")))
rows = [d["roofline"]] + d["roofline_others"]
sq = json.load(open(P("r03_pmc_sq.json")))["kernels"]
ks = list(csv.DictReader(open(P("r03_bench_kernel_stats.csv"))))


def rocprof_avg(name):
    tot = n = 0
    for r in ks:
        nm = r["Name"]
        if name.split("<")[0] not in nm:
            continue
        if "igemm" in name and (("<32" in nm) != ("<32>" in name)):
            continue
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
    return tot / n / 1e3 if n else None


W = {}
for l in open(P("r03_workloads.txt")).read().strip().splitlines():
    k, v, ms = l.rsplit(" ", 2)
    W[k] = (float(v), float(ms))
r101 = json.load(open(P("r03_bench_r101_fp8.json"))); r101b = json.load(open(P("r03_bench_r101_bf16_b32.json")))
t = []
t.append("### Headline and protocol legs (`profiles/r03_bench.json`: the default `python bench.py`, 100 timed steps after 20 warm-up)\n")
t.append("| leg | img/s | ms/step (mean; p50 / p95) | note |\n|---|---|---|---|")
t.append(f"| RetinaNet-R50-FPN, inputs resident in HBM (**`value`**) | **{d['value']:.1f}** (round 2: 577.5; driver-run 588.1) | {d['ms_per_step']:.2f}; {d['step_ms_p50']:.2f} / {d['step_ms_p95']:.2f} | `whole_step_mfma_frac` {d['config']['whole_step_mfma_frac']:.3f} |")
rp = d["reference_protocol"]
t.append(f"| same step, the reference harness's protocol (`tools/benchmark.py:125-133`: float64 host batch → fp32 → H2D inside the step, device sync around every step) — PCIe-inclusive, never `value` | **{rp['images_per_sec']:.0f}** (round 2: 275–321) | {rp['ms_per_step_mean']:.1f}; {rp['ms_p50']:.1f} / {rp['ms_p95']:.1f} (round 2: 46.9 / 90.8) | `bd_h2d_submit`: threaded conversion into pinned chunks, per-chunk DMA |")
cb = d["cpu_baseline"]
_m = re.search(r"([0-9.]+) s/iter", cb["sample"])
s_iter = _m.group(1) if _m else "?"
t.append(f"| CPU baseline, oracle (`kind: \"port\"`), {cb['cpu']}, {cb['cores']} threads: batch 2 × 800×1344 | {cb['value']:.3f} | {s_iter} s / iter | C1 (R18, 2 × 512×512) in full: {cb['c1_retinanet_r18_2x512x512']['value']:.2f} img/s |")
t.append("")
t.append("### Per kernel (HIP events on one step in 25; `traffic` = PMC `FETCH_SIZE` / `WRITE_SIZE` child passes of the same run; rocprofv3 `--kernel-trace --stats` of `bench.py --steps 10 --warmup 3 --serial-wgrad` in `profiles/r03_bench_kernel_stats.csv`)\n")
t.append("| kernel | ms / step | launches | roof | achieved | frac | HBM traffic vs algorithmic per launch | avg launch: events vs rocprofv3 | MFMA busy / wait_any (SQ) |\n|---|---|---|---|---|---|---|---|---|")
for i, r in enumerate(rows):
    k = r["kernel"]; s = sq.get(k, {}); ra = rocprof_avg(k)
    t.append(f"| `{k}`{' (**dominant: `roofline`**)' if i == 0 else ''} | {r['ms_per_step']:.2f} | {r['launches_per_step']} | {r['bound'].upper()} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}** | "
             f"{(r['traffic'] or 0) / 1e6:.0f} vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {r['avg_launch_us']:.1f} vs {ra:.1f} µs | {s.get('mfma_busy_frac_of_simd_cycles', '—')} / {s.get('wait_any_frac', '—')} |"
             if ra else f"| `{k}` | {r['ms_per_step']:.2f} | {r['launches_per_step']} | {r['bound'].upper()} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}** | {(r['traffic'] or 0) / 1e6:.0f} vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {r['avg_launch_us']:.1f} µs | — |")
t.append("")
t.append("The two weight-gradient rows time the whole `bd_conv2d_wgrad` call with events (kernel + its `wgrad_reduce_kernel` launch), rocprofv3 the kernel alone.\n")
t.append("### Other workloads (`profiles/r03_workloads.txt`, one box, `--no-roofline`, 20–30 steps; builder-run)\n")
t.append("| workload | img/s | ms/step |\n|---|---|---|")
for k, (v, ms) in W.items():
    t.append(f"| `{k}` | {v:.1f} | {ms:.2f} |")
t.append("")
t.append(f"BASELINE config 5 on one GPU (`profiles/r03_bench_r101_fp8.json`, batch 32): **{r101['value']:.1f} img/s**, {r101['ms_per_step']:.1f} ms/step "
         f"(fp8 forward + e5m2 data gradients under per-group delayed scales, the default) against bf16 {r101b['value']:.1f} ({r101b['ms_per_step']:.1f} ms); "
         f"dominant kernel `{r101['roofline']['kernel']}` {r101['roofline']['ms_per_step']:.1f} ms at {r101['roofline']['frac']:.3f} of the {r101['roofline']['peak']:.0f} {r101['roofline']['unit']} roof.\n")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s = re.sub(r"<!-- R3_TABLES_BEGIN -->.*<!-- R3_TABLES_END -->", "<!-- R3_TABLES_BEGIN -->\n" + "\n".join(t).replace("\\", "\\\\") + "\n<!-- R3_TABLES_END -->", s, flags=re.S)
open(path, "w").write(s)
print("DESIGN.md section 5 tables regenerated")
