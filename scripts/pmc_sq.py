"""Aggregate one rocprofv3 --pmc pass of SQ counters over a bench run into per-kernel fractions:
python scripts/pmc_sq.py <dir> <out.json>.  *_frac = counter / SQ_WAVE_CYCLES; mfma_busy_frac_of_simd_cycles =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) (GRBM_GUI_ACTIVE is summed over the XCDs)."""
import collections, csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
    m = re.match(r"conv_igemm_kernel<(\d+)", name)
    name = f"conv_igemm_kernel<{m.group(1)}>" if m else re.split(r"[<(]", name)[0].strip()
    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key); launches[name] += 1
out = {}
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if not wc or not ("conv" in k or "bottleneck" in k):
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    out[k] = {"launches": launches[k],
              "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
              "active_inst_any_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
              "lds_bank_conflict_cycles": int(c.get("SQ_LDS_BANK_CONFLICT", 0)), "lds_bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / wc, 4),
              "mfma_busy_frac_of_simd_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024), 3) if gui else None,
              "valu_insts_per_mfma_inst": round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2) if c.get("SQ_INSTS_MFMA") else None}
src = ("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT "
       "SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "
       "--serial-wgrad` (retinanet_r50_800x1344, batch 16); sums over all launches of a kernel")
json.dump({"source": src, "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
