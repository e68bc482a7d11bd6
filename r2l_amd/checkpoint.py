"""`.tar` checkpoint surface of the R2L path, byte-compatible with the reference (main.py:481-509, 1516-1542;
utils/run_nerf_raybased_helpers.py:333-425).

A checkpoint is a torch.save'd dict: global_step, best_psnr, best_psnr_step, network_fn_state_dict (no `module.`
prefix), optimizer_state_dict (torch.optim.Adam format) and — for R2L — network_fn = the whole pickled module,
which the loader uses IN PLACE OF the freshly built one.  Un-pickling resolves model.nerf_raybased.{NeRF_v3_2,ResMLP}
(our classes), argparse.Namespace and an EmptyClass for args.trial; torch >= 2.6 needs weights_only=False."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn


def parse_expid_iter(path):
    """'.../name_SERVER142-20210704-150540/weights/200000.tar' -> ('SERVER142-20210704-150540', '200000')."""
    # (the reference indexes path.split('_SERVER')[1] whenever 'SERVER' occurs anywhere in the path and dies with an IndexError
    # on e.g. /data/SERVER3/ckpt.tar; here such paths fall through to 'Unknown')
    if "_SERVER" in path:
        return "SERVER" + path.split("_SERVER")[1].split("/")[0], path.split("/")[-1].split(".tar")[0]
    return "Unknown", "Unknown"


def undataparallel(obj):
    """Strip a DataParallel wrapper (module) or its 'module.' key prefix (state dict)."""
    if isinstance(obj, nn.Module):
        return obj.module if hasattr(obj, "module") else obj
    if isinstance(obj, OrderedDict):
        out = OrderedDict()
        for k, v in obj.items():
            if k.startswith("module."):
                assert k.count("module.") == 1
                k = k[len("module."):]
            out[k] = v
        return out
    raise NotImplementedError(type(obj))


def load_ckpt(path, map_location=None):
    if map_location is None:
        map_location = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.load(path, map_location=map_location, weights_only=False)


def load_weights(model, ckpt_path, key):
    """Load ckpt[key] into model, tolerating 'module.' prefixes (teacher checkpoints, helpers:347-359)."""
    ckpt = load_ckpt(ckpt_path)
    sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in ckpt[key].items())
    model.load_state_dict(sd)
    return ckpt_path, ckpt


def load_weights_v2(model, ckpt, key):
    """Strict variant (helpers:362-382): model and state dict must agree on the 'module.' prefix."""
    model_dp = any(name.startswith("module.") for name, _ in model.named_modules())
    sd = ckpt[key]
    sd_dp = any(k.startswith("module.") for k in sd)
    if model_dp != sd_dp:
        raise NotImplementedError("DataParallel prefix mismatch between model and checkpoint")
    model.load_state_dict(sd)


def save_ckpt(path, global_step, model, optimizer_state_dict, best_psnr, best_psnr_step, model_name="R2L",
              model_fine=None, r2l_config=None):
    """Write a reference-layout checkpoint (main.py:1516-1542).  r2l_config: the arithmetic the run trained on
    ({'precision', 'dw_mode', 'requested'}, driver.apply_arithmetic), stored under a key of its own — the reference's loaders
    read the keys they know by name (main.py:481-509) and never see it."""
    model = undataparallel(model)
    to_save = {
        "global_step": global_step,
        "best_psnr": best_psnr,
        "best_psnr_step": best_psnr_step,
        "network_fn_state_dict": OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items()),
        "optimizer_state_dict": optimizer_state_dict,
    }
    if model_name == "nerf" and model_fine is not None:
        to_save["network_fine_state_dict"] = undataparallel(model_fine).state_dict()
    if model_name in ("nerf_v3.2", "R2L"):
        to_save["network_fn"] = model  # pickled whole, engine state excluded by NeRF_v3_2.__getstate__
    if r2l_config is not None:
        to_save["r2l_config"] = dict(r2l_config)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(to_save, path)
    return path
