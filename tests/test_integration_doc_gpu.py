"""The binding INTEGRATION.md tells a maintainer of the reference to add (section B, `model/r2l_hip.py`) is executed VERBATIM —
only the library name is pointed at the in-tree build — on a plain torch module of the reference's architecture, and checked
against the oracle: documentation that does not run is worse than none."""
import os
import re

import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "class R2LHip" in b]
    assert len(stub) == 1
    return stub[0]


def test_documented_ctypes_stub_renders_a_frame():
    from r2l_amd import _lib
    from tests.test_forward_gpu import build_model
    _lib.load()  # (torch's HIP runtime first, as r2l_amd/_lib.py explains)
    src = _stub_source().replace('ctypes.CDLL("libr2l_hip.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH)
    ns = {}
    exec(compile(src, "INTEGRATION.md:model/r2l_hip.py", "exec"), ns)
    nb = 3
    sd = O.make_state_dict(n_block=nb, seed=12)
    # a PLAIN torch module with the reference's attribute layout (head / body / tail): the stub must not need this repo's classes
    m = build_model(sd, nb)
    hip = ns["R2LHip"](m)
    H, W, focal = 37, 41, 50.
    c2w = torch.from_numpy(O.pose_spherical(30., -20., 4.)[:3, :4])
    z = O.z_vals(16, 2., 6.)
    rgb = hip.render_pose(c2w, H, W, focal, z)
    torch.cuda.synchronize()
    ref = O.r2l_forward(sd, O.positional_embed(O.sample_test(O.pixel_dirs(H, W, focal), z, c2w), 10))
    assert rgb.shape == (H * W, 3)
    assert (rgb.cpu() - ref).abs().max().item() < 1e-4
    # the parameters are views of the flat buffer now: an in-place change + repack() is seen by the next frame
    with torch.no_grad():
        m.tail[0].bias.add_(0.25)
    hip.repack()
    rgb2 = hip.render_pose(c2w, H, W, focal, z)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["tail.0.bias"] += 0.25
    ref2 = O.r2l_forward(sd2, O.positional_embed(O.sample_test(O.pixel_dirs(H, W, focal), z, c2w), 10))
    assert (rgb2.cpu() - ref2).abs().max().item() < 1e-4 and (rgb2 - rgb).abs().max().item() > 1e-3
