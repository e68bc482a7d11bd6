"""Drop-in module path of the reference's helper module; implementations live in r2l_amd."""
from r2l_amd.checkpoint import load_weights, load_weights_v2, parse_expid_iter, undataparallel  # noqa: F401
from r2l_amd.metrics import img2mse, mse2psnr, to8b, to_array, to_tensor  # noqa: F401
from r2l_amd.render import get_embedder, get_rays, get_rays_np, raw2outputs, sample_pdf  # noqa: F401
