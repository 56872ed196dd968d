"""Regenerates the measurement tables of DESIGN.md section 5 (between the R6_TABLES markers) from profiles/r06_*.
python scripts/design_tables.py"""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
d = json.load(open(P("r06_bench.json")))
rows = [d["roofline"]] + d["roofline_others"]
sq = json.load(open(P("r06_pmc_sq.json")))["kernels"]
ks = list(csv.DictReader(open(P("r06_bench_kernel_stats.csv"))))


def rocprof_avg(name, table=ks):
    tot = n = 0
    for r in table:
        nm = r["Name"]
        if name.split("<")[0] not in nm:
            continue
        if "igemm" in name and (("<32" in nm) != ("<32>" in name)):
            continue
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
    return tot / n / 1e3 if n else None


def per_step(table, steps, pats):
    return sum(float(r["TotalDurationNs"]) for r in table if any(p in r["Name"] for p in pats)) / 1e6 / steps


W = []
for l in open(P("r06_workloads.txt")).read().strip().splitlines():
    k, v, ms = l.rsplit(" ", 2)
    W.append((k.strip(), float(v), float(ms)))
r101 = json.load(open(P("r06_bench_r101_fp8.json"))); r101b = json.load(open(P("r06_bench_r101_bf16_b32.json")))
fcos = json.load(open(P("r06_bench_fcos_r50.json"))); frc = json.load(open(P("r06_bench_faster_rcnn_r50.json")))
pm = d.get("mfma_peak_measured") or {}
t = []
t.append("### Headline and protocol legs (`profiles/r06_bench.json`: the default `python bench.py`, 100 timed steps after 20 warm-up)\n")
t.append("| leg | img/s | ms/step (mean; p50 / p95) | note |\n|---|---|---|---|")
t.append(f"| RetinaNet-R50-FPN, inputs resident in HBM (**`value`**) | **{d['value']:.1f}** (round 5: 638.5; driver-run 635.9) | {d['ms_per_step']:.2f}; {d['step_ms_p50']:.2f} / {d['step_ms_p95']:.2f} | `whole_step_mfma_frac` {d['config']['whole_step_mfma_frac']:.3f} |")
rp = d["reference_protocol"]
t.append(f"| same step, the reference harness's protocol (`tools/benchmark.py:125-133`: float64 host batch → fp32 → H2D inside the step, device sync around every step) — PCIe-inclusive, never `value` | **{rp['images_per_sec']:.0f}** | {rp['ms_per_step_mean']:.1f}; {rp['ms_p50']:.1f} / {rp['ms_p95']:.1f} | `bd_h2d_submit`: threaded conversion into pinned chunks, per-chunk DMA |")
cb = d["cpu_baseline"]
_m = re.search(r"([0-9.]+) s/iter", cb["sample"])
t.append(f"| CPU baseline, oracle (`kind: \"port\"`), {cb['cpu']}, {cb['cores']} threads: batch 2 × 800×1344 | {cb['value']:.3f} | {_m.group(1) if _m else '?'} s / iter | C1 (R18, 2 × 512×512) in full: {cb['c1_retinanet_r18_2x512x512']['value']:.2f} img/s |")
if pm:
    t.append(f"| what the matrix pipes of this device sustain on the kernels' register-level pattern (`bd_probe_mfma_rate`, in the same run; `roofline.peak_measured`) | — | — | **{pm['tflops']:.0f} TFLOP/s** at an in-kernel clock of {pm['clock_mhz']:.0f} MHz (vendor peak 2 500 at 2 400 MHz) |")
t.append("")
t.append("### Per kernel (HIP events on one step in 25; `traffic` = PMC `FETCH_SIZE` / `WRITE_SIZE` child passes of the same run; kernel names from `bd_conv_last_kernel`; rocprofv3 `--kernel-trace --stats` of `bench.py --steps 10 --warmup 3 --serial-wgrad` in `profiles/r06_bench_kernel_stats.csv`)\n")
t.append("| kernel | ms / step | launches | roof | achieved | frac (of the measured MFMA peak) | clock held in the step → frac in cycles | HBM traffic vs algorithmic per launch | avg launch: events vs rocprofv3 | MFMA busy / wait_any (SQ) |\n|---|---|---|---|---|---|---|---|---|---|")
for i, r in enumerate(rows):
    k = r["kernel"]; s = sq.get(k, {}); ra = rocprof_avg(k)
    fm = f" ({r['frac_of_measured']:.3f})" if r.get("frac_of_measured") else ""
    t.append(f"| `{k}`{' (**dominant: `roofline`**)' if i == 0 else ''} | {r['ms_per_step']:.2f} | {r['launches_per_step']} | {r['bound'].upper()} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}**{fm} | " +
             (f"{r['clock_mhz']:.0f} MHz{' (PMC pass, serialised: reads high)' if str(r.get('clock_src', '')).startswith('rocprofv3') else ''} → {r['frac_in_cycles']:.3f}" if r.get("clock_mhz") else "—") + " | " +
             f"{(r['traffic'] or 0) / 1e6:.0f} vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB | {r['avg_launch_us']:.1f}" + (f" vs {ra:.1f} µs" if ra else " µs") +
             f" | {s.get('mfma_busy_frac_of_simd_cycles', '—')} / {s.get('wait_any_frac', '—')} |")
t.append("")
t.append("The weight-gradient rows time the whole `bd_conv2d_wgrad` call with events (kernel + its reduce: instrumented steps run one reduce per layer), rocprofv3 the kernel alone; "
         "their `traffic` is the kernel's (slab writes included).  `conv_wgrad1x1_kernel` (the stride-2 shortcuts): the algorithmic bytes now count the quarter of the input it reads (round 4 charged all of it: 0.486 → "
         f"{[r for r in rows if r['kernel'] == 'conv_wgrad1x1_kernel'][0]['frac']:.3f}).\n")
# FCOS / Faster R-CNN
kf = list(csv.DictReader(open(P("r06_fcos_r50_800x1344_kernel_stats.csv"))))
kr = list(csv.DictReader(open(P("r06_faster_rcnn_r50_800x1344_kernel_stats.csv"))))
steps = 13
t.append("### C3 / C4 (builder-run; `profiles/r06_bench_fcos_r50.json`, `r06_bench_faster_rcnn_r50.json`: 50 timed steps with `roofline`; kernel sums from `profiles/r06_{fcos,faster_rcnn}_r50_800x1344_kernel_stats.csv`, 13 steps)\n")
t.append("| workload | img/s (instrumented line) | dominant kernel, frac | config-specific kernels, ms per step (rocprofv3) |\n|---|---|---|---|")
gn = {n: per_step(kf, steps, (n,)) for n in ("gn_stats_partial", "gn_stats_final", "gn_apply", "gn_bwd_partial", "gn_bwd_finals", "gn_bwd_apply")}
t.append(f"| FCOS-R50-FPN, batch 16 | **{fcos['value']:.1f}** ({fcos['ms_per_step']:.2f} ms; round 5: 645.8) | `{fcos['roofline']['kernel']}` {fcos['roofline']['frac']:.3f} | "
         f"GroupNorm **{sum(gn.values()):.2f}** (round 5: 2.35): " + ", ".join(f"`{k}` {v:.2f}" for k, v in gn.items()) + " |")
box = {n: per_step(kr, steps, (n,)) for n in ("roi_align_bwd_tile", "roi_tile_list", "roi_tile_scan", "roi_foot", "conv1x1_thin_bwd", "conv1x1_thin_reduce",
                                               "conv1x1_thin_fwd", "roi_align_fwd", "rcnn_sample", "rcnn_loss", "sample_labels", "gt_rowmax", "retina_assign",
                                               "segment_topk", "nmsl_prepare", "nmsl_mask", "nmsl_scan", "nmsl_merge")}
roi_bwd = box["roi_align_bwd_tile"] + box["roi_tile_list"] + box["roi_tile_scan"] + box["roi_foot"]
thin = box["conv1x1_thin_bwd"] + box["conv1x1_thin_reduce"] + box["conv1x1_thin_fwd"]
main_chain = roi_bwd + sum(box[n] for n in ("roi_align_fwd", "rcnn_loss"))
prop = sum(box[n] for n in ("segment_topk", "nmsl_prepare", "nmsl_mask", "nmsl_scan", "nmsl_merge", "rcnn_sample"))
early = sum(box[n] for n in ("sample_labels", "gt_rowmax", "retina_assign"))
t.append(f"| Faster R-CNN R50-FPN, batch 16 | **{frc['value']:.1f}** ({frc['ms_per_step']:.2f} ms; round 5: 648.3) | `{frc['roofline']['kernel']}` {frc['roofline']['frac']:.3f} | "
         f"box operators on the main chain **{main_chain:.2f}** (round 5: 0.85): RoIAlign backward {roi_bwd:.2f} (`roi_align_bwd_tile` {box['roi_align_bwd_tile']:.2f} + its list kernels), " + ", ".join(f"`{n}` {box[n]:.2f}" for n in ("roi_align_fwd", "rcnn_loss")) +
         f"; the RPN prediction layer on its own kernels {thin:.2f} (`conv1x1_thin_fwd` {box['conv1x1_thin_fwd']:.2f}, `conv1x1_thin_bwd` {box['conv1x1_thin_bwd']:.2f} + reduce)"
         f"; proposal chain + RoI sampling on the side stream {prop:.2f} (round 5: 1.27): " + ", ".join(f"`{n}` {box[n]:.2f}" for n in ("segment_topk", "nmsl_prepare", "nmsl_mask", "nmsl_scan", "nmsl_merge", "rcnn_sample")) +
         f"; RPN targets under the forward pass {early:.2f}: " + ", ".join(f"`{n}` {box[n]:.2f}" for n in ("sample_labels", "gt_rowmax", "retina_assign")) + " |")
t.append("")
t.append("### Other workloads (`profiles/r06_workloads.txt`, one box, `--no-roofline`, 20–30 steps; builder-run)\n")
t.append("| workload | img/s | ms/step |\n|---|---|---|")
for k, v, ms in W:
    t.append(f"| `{k}` | {v:.1f} | {ms:.2f} |")
t.append("")
t.append(f"BASELINE config 5 on one GPU (`profiles/r06_bench_r101_fp8.json`, batch 32): **{r101['value']:.1f} img/s**, {r101['ms_per_step']:.1f} ms/step "
         f"(fp8 forward + e5m2 data gradients + one-byte 3×3 weight gradients under per-group delayed scales, the default) against bf16 {r101b['value']:.1f} ({r101b['ms_per_step']:.1f} ms); "
         f"dominant kernel `{r101['roofline']['kernel']}` {r101['roofline']['ms_per_step']:.1f} ms at {r101['roofline']['frac']:.3f} of the {r101['roofline']['peak']:.0f} {r101['roofline']['unit']} roof.  Round 6 (section 0 item 5): a persistent rebuild of `conv3x3_pp8_kernel` measured -0.8 % per step and was removed, 16 address VGPRs recovered (+0.4 %); the 1 x 1 launches move the same bytes in both runs (72.2 against 72.9 GB per step: a one-byte twin out for every one-byte operand in), which is why the fp8 step gains only on its 3 x 3 layers.\n")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s = re.sub(r"<!-- R6_TABLES_BEGIN -->.*<!-- R6_TABLES_END -->", "<!-- R6_TABLES_BEGIN -->\n" + "\n".join(t).replace("\\", "\\\\") + "\n<!-- R6_TABLES_END -->", s, flags=re.S)
open(path, "w").write(s)
print("DESIGN.md section 5 tables regenerated")
