"""Minimal experiment logger with the directory layout and attributes the drivers use from the reference's vendored
smilelogging.Logger (/root/reference/smilelogging/logger.py:234-288): Experiments/<name>_SERVER<id>-<time>/
{weights, gen_img, log}, .info(), .ExpID, .weights_path, .gen_img_path, .log_path.  No network probe, no pynvml,
no code cache; only rank 0 writes files."""
import os
import sys
import time

import yaml


class Logger:
    def __init__(self, args, rank=0):
        self.rank = rank
        stamp = time.strftime("%Y%m%d-%H%M%S")
        try:  # one experiment folder per job: every rank uses rank 0's time stamp
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                box = [stamp]
                dist.broadcast_object_list(box, src=0)
                stamp = box[0]
        except ImportError:
            pass
        server = os.environ.get("R2L_SERVER_ID", "000")
        self.ExpID = "SERVER%s-%s" % (server, stamp)
        root = "Debug_Dir" if getattr(args, "debug", False) else getattr(args, "experiments_dir", "Experiments")
        name = getattr(args, "experiment_name", "") or "r2l"
        self.exp_path = os.path.join(root, "%s_%s" % (name, self.ExpID))
        self.weights_path = os.path.join(self.exp_path, "weights")
        self.gen_img_path = os.path.join(self.exp_path, "gen_img")
        self.log_path = os.path.join(self.exp_path, "log")
        self._fh = None
        if rank == 0:
            for d in (self.weights_path, self.gen_img_path, self.log_path):
                os.makedirs(d, exist_ok=True)
            self._fh = open(os.path.join(self.log_path, "log.txt"), "a")
            with open(os.path.join(self.log_path, "args.yaml"), "w") as f:
                yaml.safe_dump({k: (vars(v) if hasattr(v, "__dict__") else v) for k, v in vars(args).items()}, f)
            self.info("cmd: python " + " ".join(sys.argv))

    def info(self, *msg, unprefix=False, acc=False, **_):
        if self.rank != 0:
            return
        text = " ".join(str(m) for m in msg)
        line = text if unprefix else "[%s %s] %s" % (self.ExpID[-6:], time.strftime("%Y/%m/%d-%H:%M:%S"), text)
        print(line, flush=True)
        if self._fh is not None:
            self._fh.write(line + "\n")
            self._fh.flush()

    log_printer = property(lambda self: self)

    def __call__(self, *a, **k):
        self.info(*a, **k)
