"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/r02_pmc_traffic.json (with the digest of the kernel sources it was collected on: bench.py refuses it for another build): HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests
at 64 B -- MI355X_MICROARCH.md, HBM section).   python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        m = re.match(r"conv_igemm_kernel<(\d+)", name)      # the BK = 32 (1x1) and BK = 64 instances are reported separately
        name = f"conv_igemm_kernel<{m.group(1)}>" if m else re.split(r"[<(]", name)[0].strip()
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def load_clock(d):
    """Effective clock per kernel from a pass that collected GRBM_GUI_ACTIVE: the counter is summed over the 8 XCDs, so
    MHz = sum(GRBM_GUI_ACTIVE) / 8 / sum(dispatch duration in us) (MI355X_MICROARCH.md, DVFS give-back: reads high on dispatches well under
    0.3 ms, within 3 % of the in-kernel clock on long ones).  Returns {kernel: (mhz, mean dispatch us)}."""
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, d
    gui, dur, n = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        m = re.match(r"conv_igemm_kernel<(\d+)", name)
        name = f"conv_igemm_kernel<{m.group(1)}>" if m else re.split(r"[<(]", name)[0].strip()
        gui[name] += float(r["Counter_Value"])
        dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        n[name] += 1
    return {k: (gui[k] / 8.0 / dur[k], dur[k] / n[k]) for k in gui if dur[k] > 0}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    f, fc = load(fetch_dir, "FETCH_SIZE")
    w, wc = load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0.0))):
        n = fc[k]
        fk, wk = f[k] / n, w.get(k, 0.0) / max(1, wc.get(k, 0))
        kernels[k] = {"launches": n, "fetch_size_kb": round(fk, 1), "write_size_kb": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 2 --warmup 1 "
           "--no-cpu-baseline --no-roofline --ref-protocol-steps 0`, retinanet_r50_800x1344 batch 16; values in KB per launch (mean over all launches of "
           "the kernel); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B, "
           "MI355X_MICROARCH.md section HBM); aggregated by scripts/pmc_traffic.py")
    from basedet_amd import build as _b
    json.dump({"source": src, "build_digest": _b._digest(), "kernels": kernels}, open(out, "w"), indent=1)
    for k in list(kernels)[:8]:
        print(k, kernels[k])


if __name__ == "__main__":
    main()
