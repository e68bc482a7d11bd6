"""Write configs/<scene>[_noview][_800x800].txt for the eight Blender scenes of the NeRF-synthetic set, in this repo's
config syntax (options.read_config_file: [section] headers, key=value, '#' comments).  Same flag values as the configs
the reference ships for those scenes: teacher = view-dependent NeRF with 64+128 samples, `_noview` = the R2L student
(no view directions); `_800x800` = full resolution (half_res off)."""
import os

SCENES = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")

TEACHER = '''# NeRF teacher, Blender "{scene}" at {res}: view-dependent, 64 coarse + 128 importance samples per ray.
# Used by utils/create_data.py to render pseudo training rays for the R2L student.
# Syntax: key=value per line, '#' comments, True/False for switches; command-line flags override this file.

[scene]
dataset_type=blender
datadir=./data/nerf_synthetic/{scene}
half_res={half}
white_bkgd=True
use_viewdirs=True

[sampling]
N_samples=64
N_importance=128

[optimisation]
lrate_decay=500
N_rand=1024
no_batching=True
precrop_iters=500
precrop_frac=0.5

[bookkeeping]
expname=blender_paper_{scene}
basedir=./logs
'''

STUDENT = '''# R2L student, Blender "{scene}" at {res} ({note}, composited on white).
# The student maps a ray (16 sample points) to RGB: it takes NO view-direction input.
# Syntax: key=value per line, '#' comments, True/False for switches; command-line flags override this file.

[scene]
dataset_type=blender
datadir=./data/nerf_synthetic/{scene}
half_res={half}
white_bkgd=True
use_viewdirs=False

[optimisation]        # lr 5e-4 decayed 10x every lrate_decay*1000 = 500k iterations
lrate_decay=500
N_rand=1024
no_batching=True
precrop_iters=500
precrop_frac=0.5

[teacher-style sampling]   # only used when a NeRF is rendered from this config
N_samples=64
N_importance=128

[bookkeeping]
expname=blender_paper_{scene}
basedir=./logs
'''


def write_configs(scenes, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for scene in scenes:
        for student in (False, True):
            for full_res in (False, True):
                name = scene + ("_noview" if student else "") + ("_800x800" if full_res else "") + ".txt"
                kw = dict(scene=scene, res="800x800" if full_res else "400x400", half="False" if full_res else "True",
                          note="the full-size renders" if full_res else "half_res of the 800x800 renders")
                with open(os.path.join(out_dir, name), "w") as f:
                    f.write((STUDENT if student else TEACHER).format(**kw))


def main():
    """The repo tracks only the lego files (the scene BASELINE.json names); `python tools/gen_configs.py all [dir]` writes
    the other seven scenes when they are wanted."""
    import sys
    scenes = SCENES if "all" in sys.argv[1:] else ["lego"]
    dirs = [a for a in sys.argv[1:] if a != "all"]
    out = dirs[0] if dirs else ROOT
    write_configs(scenes, out)
    print(len(os.listdir(out)), "config files in", out)


if __name__ == "__main__":
    main()
