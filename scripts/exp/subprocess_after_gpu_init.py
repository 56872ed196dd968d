import subprocess, sys, torch
torch.zeros(1, device="cuda").sum().item()
print("gpu initialised", flush=True)
r = subprocess.run(["/bin/echo", "child ran"], capture_output=True, text=True)
print("rc", r.returncode, "out", r.stdout.strip(), "err", r.stderr.strip()[:300], flush=True)
r = subprocess.run([sys.executable, "-c", "print('python child ran')"], capture_output=True, text=True)
print("rc", r.returncode, "out", r.stdout.strip(), "err", r.stderr.strip()[:300], flush=True)
