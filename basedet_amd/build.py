"""Builds libbasedet_hip.so (gfx950) in-tree with hipcc.  `python -m basedet_amd.build [--force]`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, os.environ.get("BD_LIB_NAME", "libbasedet_hip.so"))      # BD_LIB_NAME: side-by-side experimental build
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# per-file extra flags: the box ops must not contract a*b+c into FMA (bit-exact parity with the numpy oracle)
SOURCES = {
    "image_ops.hip": [],
    "h2d.hip": [],
    "conv_igemm.hip": [],
    "conv3x3.hip": [],
    "conv3x3_pp.hip": [],
    "conv3x3_pp128.hip": [],
    "conv1x1.hip": [],
    "conv1x1_ring.hip": [],
    "conv1x1_thin.hip": [],
    "bottleneck_fused.hip": [],
    "conv_fp8.hip": [],
    "conv3x3_pp8.hip": [],
    "conv_wgrad.hip": [],
    "conv_wgrad3x3.hip": [],
    "conv_wgrad3x3_ring.hip": [],
    "conv_wgrad3x3_fp8.hip": [],
    "conv_wgrad1x1.hip": [],
    "conv_wgrad1x1_ring.hip": [],
    "stem.hip": [],
    "boxops.hip": ["-ffp-contract=off"],
    "rcnn_ops.hip": ["-ffp-contract=off"],
    "postprocess.hip": ["-ffp-contract=off"],
    "freeanchor.hip": ["-ffp-contract=off"],
    "ota.hip": ["-ffp-contract=off"],
    "losses.hip": [],
    "norm.hip": [],
    "comm.hip": [],
    "layer_ops.hip": ["-ffp-contract=off"],
    "probe.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]
COMMON += os.environ.get("BD_EXTRA_FLAGS", "").split()      # diagnostic builds (-DBD_PP_STAMP, -DBD_W3_STAMP) next to BD_LIB_NAME


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/basedet_hip.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    flags = [f for f in COMMON if not os.path.isabs(f)]        # (the include directory is an absolute path: not part of the build's identity)
    h.update(repr(sorted(SOURCES.items())).encode() + repr(flags).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp = os.path.join(OUT_DIR, "build.stamp" if "BD_LIB_NAME" not in os.environ else os.environ["BD_LIB_NAME"] + ".stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        obj = os.path.join(OUT_DIR, os.environ.get("BD_LIB_NAME", "") + src.replace(".hip", ".o"))
        cmd = [HIPCC, *COMMON, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


def build_driver(verbose=False):
    """tools/abi_driver.cpp: a C++ caller of the C ABI with no Python / torch in the process (links the in-tree library by rpath)."""
    src = os.path.join(HERE, "..", "tools", "abi_driver.cpp")
    name = os.path.basename(LIB)
    out = os.path.join(OUT_DIR, "abi_driver" if name == "libbasedet_hip.so" else "abi_driver." + name)      # (diagnostic builds keep their own)
    lib = build(force=False, verbose=verbose)
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return out
    cmd = [HIPCC, "-O2", "-std=c++17", "-x", "hip", f"--offload-arch={ARCH}", "-I", os.path.join(HERE, "..", "include"), src,
           "-o", out, "-L", OUT_DIR, "-l:" + os.path.basename(LIB), "-Wl,-rpath,$ORIGIN", "-Wno-unused-result", "-Wno-unused-value"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_driver(verbose=True))
