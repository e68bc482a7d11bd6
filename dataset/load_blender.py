"""Drop-in module path of the reference's Blender data layer; implementation in r2l_amd/data.py."""
from r2l_amd.data import (BlenderDataset_v2, get_novel_poses, get_rand_pose, load_blender_data, pose_spherical)  # noqa: F401
