"""Teacher (NeRF 64+128) render timing on the GPU box: one 400x400 frame through r2l_amd.render.render (dev/profiling aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import r2l_oracle as O
from model.nerf_raybased import NeRF
from r2l_amd.render import render

csd, fsd = O.make_teacher_state_dicts(11, 2, alpha_bias=0.5)
nets = []
for sd in (csd, fsd):
    m = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    m.load_state_dict(sd); nets.append(m.cuda())
H = W = 400; focal = 555.5555155968841
c2w = torch.from_numpy(O.pose_spherical(30., -30., 4.)[:3, :4]).cuda()
kw = dict(network_fn=nets[0], network_query_fn=None, N_samples=64, N_importance=128, network_fine=nets[1],
          white_bkgd=True, perturb=1., ndc=False, near=2., far=6., use_viewdirs=True)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
with torch.no_grad():
    for _ in range(1): render(H, W, focal, chunk=chunk, c2w=c2w, **kw)
    torch.cuda.synchronize(); t0 = time.time(); K = 3
    for _ in range(K): out = render(H, W, focal, chunk=chunk, c2w=c2w, **kw)
    torch.cuda.synchronize(); dt = (time.time() - t0) / K
rays = H * W
print("teacher frame (chunk %d): %.1f ms  %.3f Mrays/s  %.1f TFLOP/s (303.82 MFLOP/ray) = %.1f%% of 157.3" %
      (chunk, dt * 1e3, rays / dt / 1e6, rays * 303.82e6 / dt / 1e12, rays * 303.82e6 / dt / 157.3e12 * 100))
