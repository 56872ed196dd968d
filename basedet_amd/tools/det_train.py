#!/usr/bin/env python3
"""`basedet_train` (basedet/tools/det_train.py:24-141): `-f <config.py>` holds a `Cfg` class; `-n` GPUs; trailing KEY VALUE pairs
override config entries.  One process per GPU: with -n > 1 the rank processes are started HERE, before this process touches the GPU
(the reference forks them through megengine.distributed.launcher, det_train.py:136-141); each rank brings up the bd_comm_*
communicator (RCCL) and runs `cfg.build_trainer().train()`."""
import argparse
import importlib.util
import os
import socket
import subprocess
import sys
import time

# RCCL shares buffers between the rank processes through dmabuf IPC on this driver stack; the legacy IPC mode fails with
# `hipIpcGetMemHandle: invalid argument`.  Must be in the environment before the first HIP call of the process -- also when the ranks
# were started by torch.distributed.run rather than by the -n launcher below.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))      # run as a script: repo root


def default_parser():
    p = argparse.ArgumentParser()
    p.add_argument("-f", "--file", type=str, required=True, help="training process description file (defines Cfg)")
    p.add_argument("-n", "--ngpus", type=int, default=1, help="number of GPUs (processes) on this node")
    p.add_argument("--iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    p.add_argument("--synthetic", action="store_true",
                   help="train on the synthetic DummyLoader batches (utils/dummy.py) -- the ONLY data source of this build: dataset "
                        "readers are outside the hot-path scope, so without this flag (or DATA.BUILDER_NAME DummyLoader) the entry refuses to run")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE pairs merged into the config")
    return p


def load_cfg(path):
    """det_train.py:125-127: import the description file and instantiate its Cfg."""
    path = os.path.abspath(path)
    sys.path.append(os.path.dirname(path))
    spec = importlib.util.spec_from_file_location(os.path.basename(path).split(".")[0], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Cfg()


def worker(args):
    import torch
    from basedet_amd import comm
    cfg = load_cfg(args.file)
    if args.opts:
        cfg.merge(args.opts)                                            # det_train.py:71
    if args.synthetic:
        cfg.DATA.BUILDER_NAME = "DummyLoader"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        comm.set_comm(comm.Comm.from_env())
    else:
        torch.cuda.set_device(0)
    trainer = cfg.build_trainer()
    out = trainer.train(max_iters=args.iters)
    torch.cuda.synchronize()
    if comm.get_comm() is not None:
        comm.get_comm().barrier()
        comm.get_comm().destroy()                                       # ncclCommDestroy before the process exits
        comm.set_comm(None)
    return out


def wait_ranks(procs, poll_s=0.2):
    """Poll every rank process; when one exits non-zero, end the survivors by their exact PIDs (they would otherwise block forever in
    an RCCL collective) and return that code."""
    rc = 0
    alive = set(range(len(procs)))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for o in alive:
                    procs[o].kill()
        if alive:
            time.sleep(poll_s)
    return rc


def main():
    args = default_parser().parse_args()
    if args.ngpus > 1 and "WORLD_SIZE" not in os.environ:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(args.ngpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ngpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        sys.exit(wait_ranks(procs))
    worker(args)


if __name__ == "__main__":
    main()
