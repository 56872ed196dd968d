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

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))      # run as a script: repo root


def default_parser():
    p = argparse.ArgumentParser()
    p.add_argument("-f", "--file", type=str, required=True, help="training process description file (defines Cfg)")
    p.add_argument("-n", "--ngpus", type=int, default=1, help="number of GPUs (processes) on this node")
    p.add_argument("--iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
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
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        comm.set_comm(comm.Comm.from_env())
    else:
        torch.cuda.set_device(0)
    trainer = cfg.build_trainer()
    out = trainer.train(max_iters=args.iters)
    torch.cuda.synchronize()
    if comm.get_comm() is not None:
        comm.get_comm().barrier()
    return out


def main():
    args = default_parser().parse_args()
    if args.ngpus > 1 and "WORLD_SIZE" not in os.environ:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(args.ngpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ngpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        for p in procs:
            rc = p.wait() or rc
        sys.exit(rc)
    worker(args)


if __name__ == "__main__":
    main()
