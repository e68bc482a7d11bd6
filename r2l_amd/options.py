"""Command-line / config-file surface of the R2L drivers (main.py, utils/create_data.py).

Keeps the flag names, defaults and file syntax of the reference's option.py + smilelogging argparser
(/root/reference/option.py:6-358, smilelogging/__init__.py:8-42) for every flag that parameterises the hot path
(SURVEY.md appendix A): `--config file` with `key = value` lines (# comments, True/False for switches), command
line overrides the file, argparse prefix abbreviations work (README spells --num_worker), dotted `--trial.*`
flags fold into `args.trial` iff `--trial.ON` (slutils.py:176-188).  configargparse is not installed on the
target image, so this is a small table-driven parser of our own.
"""
import argparse
import glob
import os
import sys

from utils import EmptyClass

# name: (type | "flag", default)            -- "flag" = store_true switch
FLAGS = {
    # experiment bookkeeping (smilelogging argparser)
    "experiment_name": (str, ""), "experiments_dir": (str, "Experiments"), "debug": ("flag", False),
    "resume_TimeID": (str, ""),
    # data
    "expname": (str, None), "basedir": (str, "./logs/"), "datadir": (str, "./data/nerf_synthetic/lego"),
    "dataset_type": (str, "blender"), "testskip": (int, 8), "white_bkgd": ("flag", False), "half_res": ("flag", False),
    "datadir_kd": (str, ""), "data_mode": (str, "images"), "num_workers": (int, 8), "pseudo_ratio": (float, -1.),
    "pseudo_data_hold_ratio": (float, 0.), "i_update_data": (int, 1000000000), "focal_scale": (float, 1.),
    # networks
    "model_name": (str, "R2L"), "netdepth": (int, 8), "netwidth": (int, 256), "netdepth_fine": (int, 8),
    "netwidth_fine": (int, 256), "n_sample_per_ray": (int, 192), "multires": (int, 10), "multires_views": (int, 4),
    "i_embed": (int, 0), "use_viewdirs": ("flag", False), "use_residual": ("flag", False),
    "linear_tail": ("flag", False), "layerwise_netwidths": (str, ""), "act": (str, "relu"),
    "freeze_pretrained": ("flag", False),
    # rendering / sampling
    "N_samples": (int, 64), "N_importance": (int, 0), "perturb": (float, 1.), "perturb_test": (float, 0.),
    "raw_noise_std": (float, 0.), "chunk": (int, 1024 * 32), "netchunk": (int, 1024 * 64), "lindisp": ("flag", False),
    "no_ndc": ("flag", False), "render_only": ("flag", False), "render_test": ("flag", False),
    "render_factor": (float, 0), "n_pose_video": (str, "40"), "video_tag": (str, ""), "benchmark": ("flag", False),
    # optimisation
    "N_rand": (int, 32 * 32 * 4), "N_iters": (int, 200000), "lrate": (float, 5e-4), "lrate_decay": (int, 250),
    "warmup_lr": (str, ""), "lw_rgb": (float, 1.), "hard_ratio": (str, ""), "hard_mul": (float, 1.),
    "no_batching": ("flag", False), "precrop_iters": (int, 0), "precrop_frac": (float, .5),
    # checkpoints / logging cadence
    "pretrained_ckpt": (str, ""), "resume": ("flag", False), "test_pretrained": ("flag", False),
    "save_intermediate_models": ("flag", False), "i_print": (int, 100), "i_img": (int, 500), "i_weights": (int, 10000),
    "i_testset": (int, 2000), "i_video": (int, 10000), "no_reload": ("flag", False), "ft_path": (str, None),
    # teacher / pseudo-data generation (utils/create_data.py)
    "teacher_ckpt": (str, None), "test_teacher": ("flag", False), "create_data": (str, "spiral_evenly_spaced"),
    "n_pose_kd": (str, "100"), "create_data_chunk": (int, 100), "rm_existing_data": ("flag", False),
    "max_save": (int, 40000),
    # arithmetic of the HIP kernels (this build's own keys; include/r2l_hip.h r2l_config.precision / .dw_mode): which kernel
    # family every launch of the run uses.  auto = the library default (fp16x2 products, fp16 weight-gradient operands);
    # fp32_mfma = the reference's arithmetic (exact fp32 products and accumulation) — the graded numbers of bench.py
    "r2l_precision": (str, "auto"), "r2l_dw_mode": (str, "auto"),
    # new-architecture switches (dotted group)
    "trial.ON": ("flag", False), "trial.body_arch": (str, "mlp"), "trial.res_scale": (float, 1.),
    "trial.n_learnable": (int, 2), "trial.inact": (str, "relu"), "trial.outact": (str, "none"),
    "trial.n_block": (int, -1), "trial.near": (float, -1.), "trial.far": (float, -1.),
}
CHOICES = {"r2l_precision": ["auto", "fp16x2", "bf16x3", "fp32_mfma"], "r2l_dw_mode": ["auto", "fp16", "exact"],
           "model_name": ["nerf", "nerf_v3.2", "R2L"], "data_mode": ["images", "rays"], "act": ["relu", "lrelu"],
           "trial.body_arch": ["mlp", "resmlp"], "trial.inact": ["none", "relu", "lrelu"],
           "trial.outact": ["none", "relu", "lrelu"]}
# flags of reference variants outside the accelerated path: accepted so old command lines / configs still parse
IGNORED = {"plucker": ("flag", False), "learn_depth": (str, ""), "shuffle_input": ("flag", False),
           "given_render_path_rays": (str, ""), "convert_to_onnx": ("flag", False), "lpips_net": (str, "alex"),
           "trans_origin": (str, ""), "select_pixel_mode": (str, "rand_pixel"), "factor": (int, 8),
           "spherify": ("flag", False), "llffhold": (int, 8), "shape": (str, "greek"), "lw_depth": (float, 0.1),
           "no_cache": ("flag", False), "no_scp": ("flag", False), "cache_code": (str, ""), "skips": (str, "4")}


def _build_parser():
    p = argparse.ArgumentParser(description="R2L on MI355X", allow_abbrev=True)
    p.add_argument("--config", type=str, default=None, help="config file: key = value per line")
    for table in (FLAGS, IGNORED):
        for name, (typ, default) in table.items():
            if typ == "flag":
                p.add_argument("--" + name, action="store_true", default=default)
            else:
                p.add_argument("--" + name, type=typ, default=default, choices=CHOICES.get(name))
    p.add_argument("--no_rand_focal", dest="use_rand_focal", action="store_false", default=True)
    return p


def read_config_file(path):
    """`key = value` lines -> argv-style list ('#' starts a comment; True/False toggle store_true switches)."""
    argv = []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line or (line.startswith("[") and line.endswith("]")):  # blank / [section] headers are cosmetic
                continue
            if "=" in line:
                key, val = [s.strip() for s in line.split("=", 1)]
            else:
                key, val = line, "True"
            typ = (FLAGS.get(key) or IGNORED.get(key) or (str,))[0]
            if typ == "flag" or key == "no_rand_focal":
                if val.lower() in ("true", "1", "yes"):
                    argv.append("--" + key)
                elif val.lower() not in ("false", "0", "no"):
                    raise ValueError("%s: switch %s needs True/False, got %r" % (path, key, val))
            else:
                argv += ["--" + key, val]
    return argv


def check_path(pattern):
    """Expand a glob to exactly one existing path (smilelogging/utils.py:424-432 behaviour); '' stays ''."""
    if not pattern:
        return pattern
    hits = sorted(glob.glob(pattern))
    if len(hits) != 1:
        raise FileNotFoundError("%r matched %d paths (need exactly 1): %s" % (pattern, len(hits), hits[:5]))
    return hits[0]


def _n_pose(v):
    if v.lower() == "none":
        return None
    return int(v) if v.isdigit() else v.split(",")


def parse_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    p = _build_parser()
    pre, _ = p.parse_known_args(argv)
    if pre.config:
        argv = read_config_file(pre.config) + argv  # later (command-line) occurrences win
    args = p.parse_args(argv)
    # post-processing, as option.py:362-386
    if args.video_tag == "":
        args.video_tag = "pose%s" % args.n_pose_video
    args.n_pose_kd = _n_pose(args.n_pose_kd)
    args.n_pose_video = _n_pose(args.n_pose_video)
    args.pretrained_ckpt = check_path(args.pretrained_ckpt)
    if args.hard_ratio != "":
        hr = [float(x) for x in args.hard_ratio.split(",")]
        args.hard_ratio = hr[0] if len(hr) == 1 else hr
    # fold dotted groups: args.'trial.x' -> args.trial.x when trial.ON
    for key in [k for k in vars(args) if "." in k]:
        group, name = key.split(".")
        if getattr(args, group + ".ON"):
            if not hasattr(args, group):
                setattr(args, group, EmptyClass())
            setattr(getattr(args, group), name, getattr(args, key))
    for key in [k for k in vars(args) if "." in k]:
        delattr(args, key)
    return args


def validate_accelerated(args):
    """Fail loudly for reference variants this build does not accelerate (out of scope)."""
    for name in ("plucker", "learn_depth", "shuffle_input", "given_render_path_rays", "convert_to_onnx"):
        if getattr(args, name):
            raise NotImplementedError("--%s is a reference variant outside the accelerated R2L path" % name)
    if args.dataset_type != "blender":
        raise NotImplementedError("only --dataset_type blender is on the accelerated path (got %s)" % args.dataset_type)
