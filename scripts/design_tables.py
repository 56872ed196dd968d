"""Regenerates the round-4 measurement tables of DESIGN.md section 5 (between the R4_TABLES markers) and the fp8 stability rows of section 7
from profiles/r04_*.
python scripts/design_tables.py"""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(P("r04_bench.json")))
rows = [d["roofline"]] + d["roofline_others"]
sq = json.load(open(P("r04_pmc_sq.json")))["kernels"]
ks = list(csv.DictReader(open(P("r04_bench_kernel_stats.csv"))))


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
for l in open(P("r04_workloads.txt")).read().strip().splitlines():
    k, v, ms = l.rsplit(" ", 2)
    W[k] = (float(v), float(ms))
r101 = json.load(open(P("r04_bench_r101_fp8.json"))); r101b = json.load(open(P("r04_bench_r101_bf16_b32.json")))
t = []
t.append("### Headline and protocol legs (`profiles/r04_bench.json`: the default `python bench.py`, 100 timed steps after 20 warm-up)\n")
t.append("| leg | img/s | ms/step (mean; p50 / p95) | note |\n|---|---|---|---|")
t.append(f"| RetinaNet-R50-FPN, inputs resident in HBM (**`value`**) | **{d['value']:.1f}** (round 3: 614.2; driver-run 611.8) | {d['ms_per_step']:.2f}; {d['step_ms_p50']:.2f} / {d['step_ms_p95']:.2f} | `whole_step_mfma_frac` {d['config']['whole_step_mfma_frac']:.3f} |")
if os.path.exists(P("r04_bench_box3.json")):
    d3 = json.load(open(P("r04_bench_box3.json")))
    t.append(f"| the same build and command on another box (`profiles/r04_bench_box3.json`) | {d3['value']:.1f} | {d3['ms_per_step']:.2f}; {d3['step_ms_p50']:.2f} / {d3['step_ms_p95']:.2f} | `{d3['roofline']['kernel']}` {d3['roofline']['frac']:.3f}; reference-harness protocol {d3['reference_protocol']['images_per_sec']:.0f} img/s |")
d2 = json.load(open(P("r04_bench_box2.json")))
t.append(f"| the build BEFORE `conv1x1_ring_kernel` on two other boxes of the pool (`profiles/r04_bench_box2.json`, and 601.6 on the slowest box; spread of one build over the round's boxes: 601–631 img/s) | {d2['value']:.1f} | {d2['ms_per_step']:.2f}; {d2['step_ms_p50']:.2f} / {d2['step_ms_p95']:.2f} | dominant kernel `{d2['roofline']['kernel']}` {d2['roofline']['frac']:.3f}, `conv_wgrad3x3_ring_kernel` {[e for e in d2['roofline_others'] if e['kernel'] == 'conv_wgrad3x3_ring_kernel'][0]['frac']:.3f} |")
ab = {k: v for k, v in W.items()}
base = [v for k, v in ab.items() if k.strip() == "retinanet_r50 --batch 16"]
old_k = ab.get("retinanet_r50 --batch 16 --wgrad-knob 5")
if base and old_k:
    t.append(f"| same-box A/B of this round's weight-gradient kernels (`profiles/r04_workloads.txt`, box of this profile, 30 steps, uninstrumented): ring kernels + bucket reduce vs `--wgrad-knob 5` (the round-3 kernels) | **{base[-1][0]:.1f} vs {old_k[0]:.1f}** (+{(base[-1][0] / old_k[0] - 1) * 100:.1f} %) | {base[-1][1]:.2f} vs {old_k[1]:.2f} | R101 batch 16: {ab['retinanet_r101 --batch 16'][0]:.1f} vs {ab['retinanet_r101 --batch 16 --wgrad-knob 5'][0]:.1f} img/s (+{(ab['retinanet_r101 --batch 16'][0] / ab['retinanet_r101 --batch 16 --wgrad-knob 5'][0] - 1) * 100:.1f} %) |")
ring_off = ab.get("retinanet_r50 --batch 16 BD_DENSE1X1_RING=0")
ring_on = [v for k, v in ab.items() if k.startswith("retinanet_r50 --batch 16") and (k.strip() == "retinanet_r50 --batch 16" or "(again)" in k)]
if ring_off and ring_on:
    on = sum(v[0] for v in ring_on) / len(ring_on); onms = sum(v[1] for v in ring_on) / len(ring_on)
    t.append(f"| same-box A/B of `conv1x1_ring_kernel` (same file): default dispatch vs `BD_DENSE1X1_RING=0` (every dense 1×1 launch on `conv1x1_dense_kernel`) | **{on:.1f} vs {ring_off[0]:.1f}** (+{(on / ring_off[0] - 1) * 100:.1f} %) | {onms:.2f} vs {ring_off[1]:.2f} | an earlier alternation on another box: 638.1 / 638.0 vs 624.2 / 625.4 (+2.1 %) |")
rp = d["reference_protocol"]
t.append(f"| same step, the reference harness's protocol (`tools/benchmark.py:125-133`: float64 host batch → fp32 → H2D inside the step, device sync around every step) — PCIe-inclusive, never `value` | **{rp['images_per_sec']:.0f}** (round 3: 488) | {rp['ms_per_step_mean']:.1f}; {rp['ms_p50']:.1f} / {rp['ms_p95']:.1f} | `bd_h2d_submit`: threaded conversion into pinned chunks, per-chunk DMA |")
cb = d["cpu_baseline"]
_m = re.search(r"([0-9.]+) s/iter", cb["sample"])
s_iter = _m.group(1) if _m else "?"
t.append(f"| CPU baseline, oracle (`kind: \"port\"`), {cb['cpu']}, {cb['cores']} threads: batch 2 × 800×1344 | {cb['value']:.3f} | {s_iter} s / iter | C1 (R18, 2 × 512×512) in full: {cb['c1_retinanet_r18_2x512x512']['value']:.2f} img/s |")
t.append("")
t.append("### Per kernel (HIP events on one step in 25; `traffic` = PMC `FETCH_SIZE` / `WRITE_SIZE` child passes of the same run; rocprofv3 `--kernel-trace --stats` of `bench.py --steps 10 --warmup 3 --serial-wgrad` in `profiles/r04_bench_kernel_stats.csv`)\n")
t.append("| kernel | ms / step | launches | roof | achieved | frac | HBM traffic vs algorithmic per launch | avg launch: events vs rocprofv3 | MFMA busy / wait_any (SQ) |\n|---|---|---|---|---|---|---|---|---|")
for i, r in enumerate(rows):
    k = r["kernel"]; s = sq.get(k, {}); ra = rocprof_avg(k)
    t.append(f"| `{k}`{' (**dominant: `roofline`**)' if i == 0 else ''} | {r['ms_per_step']:.2f} | {r['launches_per_step']} | {r['bound'].upper()} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}** | "
             f"{(r['traffic'] or 0) / 1e6:.0f} vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {r['avg_launch_us']:.1f} vs {ra:.1f} µs | {s.get('mfma_busy_frac_of_simd_cycles', '—')} / {s.get('wait_any_frac', '—')} |"
             if ra else f"| `{k}` | {r['ms_per_step']:.2f} | {r['launches_per_step']} | {r['bound'].upper()} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}** | {(r['traffic'] or 0) / 1e6:.0f} vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {r['avg_launch_us']:.1f} µs | — |")
t.append("")
t.append("The weight-gradient rows time the whole `bd_conv2d_wgrad` call with events (kernel + its reduce: instrumented steps run one reduce per layer), rocprofv3 the kernel alone; "
         "their `traffic` is the kernel's (slab writes included), the reduce's reads are `wgrad_batch_reduce_kernel`'s.\n")
t.append("### Other workloads (`profiles/r04_workloads.txt`, one box, `--no-roofline`, 20–30 steps; builder-run)\n")
t.append("| workload | img/s | ms/step |\n|---|---|---|")
for k, (v, ms) in W.items():
    t.append(f"| `{k}` | {v:.1f} | {ms:.2f} |")
t.append("")
t.append(f"BASELINE config 5 on one GPU (`profiles/r04_bench_r101_fp8.json`, batch 32): **{r101['value']:.1f} img/s**, {r101['ms_per_step']:.1f} ms/step "
         f"(fp8 forward + e5m2 data gradients + one-byte 3×3 weight gradients under per-group delayed scales, the default) against bf16 {r101b['value']:.1f} ({r101b['ms_per_step']:.1f} ms); "
         f"dominant kernel `{r101['roofline']['kernel']}` {r101['roofline']['ms_per_step']:.1f} ms at {r101['roofline']['frac']:.3f} of the {r101['roofline']['peak']:.0f} {r101['roofline']['unit']} roof.\n")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s = re.sub(r"<!-- R4_TABLES_BEGIN -->.*<!-- R4_TABLES_END -->", "<!-- R4_TABLES_BEGIN -->\n" + "\n".join(t).replace("\\", "\\\\") + "\n<!-- R4_TABLES_END -->", s, flags=re.S)
open(path, "w").write(s)
# fp8 stability rows (section 7)
rows = []
import glob
for f in sorted(glob.glob(P("r04_fp8_stability_lr*.txt")), reverse=True):
    lr = re.search(r"lr([0-9.]+)\.txt", f).group(1)
    by = {}
    for l in open(f):
        if not l.startswith("seed"):
            continue
        _, name, rc, losses, fin = [x.strip() for x in l.split("|")]
        fl = fin.replace("final/img_s", "").split()
        ok = rc == "rc 0" and fl and fl[0] not in ("None", "nan") and float(fl[0]) < 1.0        # (a finite plateau at 11.7 is a diverged run too)
        by.setdefault(name, []).append((ok, float(fl[0]) if ok else None, float(fl[1]) if ok and len(fl) > 1 else None))
    def cell(name):
        v = by.get(name)
        if not v:
            return "—"
        good = [x for x in v if x[0]]
        fin = ", ".join(f"{x[1]:.3f}" if x[0] else "diverged" for x in v)
        ips = sum(x[2] for x in good) / len(good) if good else 0
        return f"**{len(good)} / {len(v)}** finish ({fin}); {ips:.0f} img/s"
    rows.append(f"| {lr} (`profiles/{os.path.basename(f)}`) | {cell('bf16')} | {cell('fp8 default (e5m2 dgrad, group scales)')} | {cell('fp8 forward only')} |")
s = open(path).read()
if "FP8_ROWS" in s:
    s = s.replace("FP8_ROWS", "\n".join(rows))
open(path, "w").write(s)
print("DESIGN.md section 5 tables and fp8 rows regenerated")
