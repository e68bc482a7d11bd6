"""Drop-in module path of the reference's model layer: `from model.nerf_raybased import NeRF_v3_2, ...` and the
pickled `network_fn` inside reference `.tar` checkpoints both resolve here; the implementation is r2l_amd's."""
from r2l_amd.nerf_raybased import (NeRF, NeRF_v3_2, PointSampler, PositionalEmbedder, ResMLP, get_activation)  # noqa: F401
