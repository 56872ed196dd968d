"""First contact with a box that has >= 2 GPUs (SURVEY 8e / DESIGN 6: N > 1 has never run on hardware).  One command:

    python scripts/multigpu_check.py [--max-gpus 8] [--steps 30] [--warmup 10] [--skip-tests]

1. runs tests/test_dist_gpu.py (the two-rank RCCL tests are skipped on a one-GPU box);
2. runs `bench.py --gpus N` for N = 1, 2, 4, 8 (up to the devices present) for BASELINE configs C2 (RetinaNet-R50) and C3 (FCOS-R50),
   16 images per GPU (weak scaling), and prints img/s, the speed-up over N = 1, the exposed all-reduce time and the RCCL rank count;
3. writes the raw result lines to gpurun_out/multigpu_check.jsonl.
Counting devices does not initialise HIP in this process (torch.cuda.device_count), so the bench parents can still start fresh ranks."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--skip-tests", action="store_true")
    ap.add_argument("--workloads", default="retinanet_r50_800x1344,fcos_r50_800x1344")
    a = ap.parse_args()
    import torch
    ndev = torch.cuda.device_count()
    print(f"devices: {ndev}", flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc = 0
    if not a.skip_tests:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_dist_gpu.py"), "-q", "-m", "gpu", "-rs"],
                           cwd=ROOT, env=env)
        rc = r.returncode
        print(f"tests/test_dist_gpu.py: exit {rc}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rows = []
    with open(os.path.join(ROOT, "gpurun_out", "multigpu_check.jsonl"), "w") as log:
        for wl in a.workloads.split(","):
            base = None
            for n in (1, 2, 4, 8):
                if n > min(ndev, a.max_gpus):
                    break
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup),
                       "--workload", wl, "--no-pmc", "--no-cpu-baseline", "--no-roofline", "--ref-protocol-steps", "0"]
                r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
                line = next((l for l in r.stdout.splitlines()[::-1] if l.startswith("{")), None)
                if r.returncode != 0 or line is None:
                    print(f"{wl} --gpus {n}: FAILED (exit {r.returncode})", flush=True)
                    rc = rc or r.returncode or 1
                    break
                log.write(line + "\n")
                log.flush()
                d = json.loads(line)
                base = base or d["value"]
                c = d.get("comm", {})
                rows.append((wl, n, d["value"], d["value"] / base, d["ms_per_step"], c.get("allreduce_exposed_ms"), c.get("rccl_ranks")))
    print(f"{'workload':28s} {'gpus':>4s} {'img/s':>9s} {'x N=1':>6s} {'ms/step':>8s} {'exposed all-reduce ms':>22s} {'rccl_ranks':>10s}")
    for wl, n, v, sp, ms, ex, rk in rows:
        print(f"{wl:28s} {n:4d} {v:9.1f} {sp:6.2f} {ms:8.2f} {str(ex):>22s} {str(rk):>10s}")
    sys.exit(rc)


if __name__ == "__main__":
    main()
