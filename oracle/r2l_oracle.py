"""CPU oracle for the R2L hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional fp32 restatement (torch CPU ops, same op order as the reference) of the algorithm the HIP kernels
implement.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under r2l_amd/ does.  Each function cites the reference file:line (paths into /root/reference) it restates.

Pinning: the reference ships no tests or known-answer vectors, so this oracle is pinned against
outputs of the reference itself, imported in the dev container by tests/golden/gen_golden.py and frozen as
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against those vectors (bit-exact
where the op sequence is identical, else to 1e-6).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------------------
# a1-a3  PointSampler   (model/nerf_raybased.py:78-126)
# ---------------------------------------------------------------------------------------------------------------


def pixel_dirs(H, W, focal):
    """PointSampler.__init__ dirs[H,W,3]  (nerf_raybased.py:80-86; same as helpers get_rays:233-240)."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    return torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -torch.ones_like(i)], dim=-1)


def z_vals(n_sample, near, far):
    """PointSampler.__init__ z_vals[n_sample]  (nerf_raybased.py:88-90)."""
    t = torch.linspace(0., 1., steps=n_sample)
    return near * (1 - t) + far * t


def rays_from_pose(dirs, c2w):
    """rays_o, rays_d [H*W,3] as in sample_test (nerf_raybased.py:95-99) / get_rays (helpers:243-247)."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    rays_d = torch.sum(dirs.unsqueeze(dim=-2) * c2w[:3, :3], dim=-1).view(-1, 3)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def sample_test(dirs, z, c2w):
    """PointSampler.sample_test -> pts[H*W, 3*n_sample]  (nerf_raybased.py:94-102)."""
    rays_o, rays_d = rays_from_pose(dirs, c2w)
    zt = z[None, :].expand(rays_d.shape[0], z.shape[0])
    pts = rays_o[..., None, :] + rays_d[..., None, :] * zt[..., :, None]
    return pts.reshape(pts.shape[0], -1)


def stratified_bounds(z):
    """lower, upper of the jitter strata (nerf_raybased.py:119-121)."""
    mids = .5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], dim=-1)
    lower = torch.cat([z[..., :1], mids], dim=-1)
    return lower, upper


def sample_train(rays_o, rays_d, z, perturb=0., t_rand=None):
    """PointSampler.sample_train -> pts[N, 3*n_sample]  (nerf_raybased.py:114-126).  t_rand replaces torch.rand."""
    zz = z[None, :].expand(rays_o.shape[0], z.shape[0])
    if perturb > 0.:
        lower, upper = stratified_bounds(zz)
        zz = lower + (upper - lower) * t_rand
    pts = rays_o[..., None, :] + rays_d[..., None, :] * zz[..., :, None]
    return pts.reshape(pts.shape[0], -1)


# ---------------------------------------------------------------------------------------------------------------
# a4  PositionalEmbedder   (model/nerf_raybased.py:191-208): per coordinate [sin(2^0x)..sin(2^9x), cos.., x]
# ---------------------------------------------------------------------------------------------------------------


def positional_embed(x, L=10):
    w = 2**torch.linspace(0, L - 1, steps=L)
    y = x[..., None] * w
    y = torch.cat([torch.sin(y), torch.cos(y)], dim=-1)
    y = torch.cat([y, x.unsqueeze(dim=-1)], dim=-1)
    return y.view(y.shape[0], -1)


# ---------------------------------------------------------------------------------------------------------------
# a5-a7  NeRF_v3_2 / ResMLP forward on a state_dict   (model/nerf_raybased.py:461-465, 539-544)
# ---------------------------------------------------------------------------------------------------------------


def n_block_of(sd):
    return len([k for k in sd if k.startswith("body.") and k.endswith(".body.0.weight")])


def r2l_forward(sd, emb, res_scale=1., use_residual=True, return_acts=False):
    """rgb[N,3] = tail(body(head(emb)) + head(emb)); optionally the per-block activations the backward needs."""
    h0 = F.relu(F.linear(emb, sd["head.0.weight"], sd["head.0.bias"]))
    x = h0
    xs, ts = [h0], []
    for b in range(n_block_of(sd)):
        t = F.relu(F.linear(x, sd["body.%d.body.0.weight" % b], sd["body.%d.body.0.bias" % b]))
        x = F.linear(t, sd["body.%d.body.2.weight" % b], sd["body.%d.body.2.bias" % b]).mul(res_scale) + x
        ts.append(t)
        xs.append(x)
    y = x + h0 if use_residual else x
    rgb = torch.sigmoid(F.linear(y, sd["tail.0.weight"], sd["tail.0.bias"]))
    if return_acts:
        return rgb, xs, ts
    return rgb


def img2mse(x, y):
    """helpers:19"""
    return torch.mean((x - y)**2)


def mse2psnr(x):
    """helpers:20"""
    return -10. * torch.log(x) / torch.log(torch.tensor([10.]))


def r2l_loss_and_grads(sd, emb, target):
    """loss = mean((rgb-target)^2) and d loss / d every tensor via autograd  (main.py:1374-1380, 1403-1404)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    rgb = r2l_forward(p, emb)
    loss = img2mse(rgb, target)
    loss.backward()
    return loss.detach(), rgb.detach(), {k: v.grad for k, v in p.items()}


def lr_schedule(step, lrate, lrate_decay, warmup_lr=""):
    """main.py:1181-1193 (decay_rate 0.1, decay_steps = lrate_decay*1000, optional 'start_lr,end_iter' warm-up)."""
    decay_rate, decay_steps = 0.1, lrate_decay * 1000
    if warmup_lr:
        start_lr, end_iter = [float(v) for v in warmup_lr.split(",")]
        if step < end_iter:
            return (lrate - start_lr) / end_iter * step + start_lr
        return lrate * (decay_rate**((step - end_iter) / decay_steps))
    return lrate * (decay_rate**(step / decay_steps))


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (main.py:465-467) single-tensor update, step counted from 1; returns new (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def hard_ray_indices(rgb, target, n_hard_in):
    """main.py:1411-1414: indices of the n_hard_in rays with the largest per-ray MSE (ascending sort, take the tail)."""
    _, indices = torch.sort(torch.mean((rgb - target)**2, dim=1))
    return indices[-n_hard_in:]


# ---------------------------------------------------------------------------------------------------------------
# a12-a13  teacher embedder + NeRF MLP   (helpers:24-74; model/nerf_raybased.py:377-401)
# ---------------------------------------------------------------------------------------------------------------


def nerf_embed(x, multires):
    """get_embedder: [x, sin(2^0x), cos(2^0x), ..., sin(2^(L-1)x), cos(2^(L-1)x)], each over all 3 dims (helpers:30-56)."""
    freqs = 2.**torch.linspace(0., multires - 1, steps=multires)
    out = [x]
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def nerf_forward(sd, x, D=8, skips=(4,), input_ch=63, input_ch_views=27):
    """NeRF.forward with use_viewdirs=True -> [n,4] = [rgb(3), alpha]  (nerf_raybased.py:377-401)."""
    input_pts, input_views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    h = input_pts
    for i in range(D):
        h = F.relu(F.linear(h, sd["pts_linears.%d.weight" % i], sd["pts_linears.%d.bias" % i]))
        if i in skips:
            h = torch.cat([input_pts, h], -1)
    alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    feature = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    h = torch.cat([feature, input_views], -1)
    h = F.relu(F.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
    rgb = F.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    return torch.cat([rgb, alpha], -1)


def run_network(sd, pts, viewdirs, multires=10, multires_views=4):
    """run_network (create_data.py:55-77): embed points and (expanded) view directions, apply the MLP."""
    flat = pts.reshape(-1, 3)
    emb = nerf_embed(flat, multires)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    emb = torch.cat([emb, nerf_embed(dirs, multires_views)], -1)
    out = nerf_forward(sd, emb)
    return out.reshape(list(pts.shape[:-1]) + [4])


# ---------------------------------------------------------------------------------------------------------------
# a16  raw2outputs   (create_data.py:335-402 == main.py:556-621 == nerf_raybased.py:226-295)
# ---------------------------------------------------------------------------------------------------------------


def raw2outputs(raw, z, rays_d, noise=None, white_bkgd=False):
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10]).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1. - torch.exp(-F.relu(sigma) * dists)
    weights = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z, -1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(weights, -1))
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# ---------------------------------------------------------------------------------------------------------------
# a17  sample_pdf   (helpers:283-330); u is given explicitly (det -> linspace, else the caller's uniforms)
# ---------------------------------------------------------------------------------------------------------------


def sample_pdf(bins, weights, N_samples, det=False, u=None):
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0., 1., steps=N_samples)
        u = u.expand(list(cdf.shape[:-1]) + [N_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


# ---------------------------------------------------------------------------------------------------------------
# a15  render_rays   (create_data.py:405-544): coarse -> raw2outputs -> sample_pdf -> sort-merge -> fine
# ---------------------------------------------------------------------------------------------------------------


def render_rays(ray_batch, sd_coarse, sd_fine, N_samples=64, N_importance=128, perturb=0., t_rand=None, u=None,
                white_bkgd=True):
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    bounds = torch.reshape(ray_batch[..., 6:8], [-1, 1, 2])
    near, far = bounds[..., 0], bounds[..., 1]
    t_vals = torch.linspace(0., 1., steps=N_samples)
    z = near * (1. - t_vals) + far * t_vals
    z = z.expand([N_rays, N_samples])
    if perturb > 0.:
        lower, upper = stratified_bounds(z)
        z = lower + (upper - lower) * t_rand
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = run_network(sd_coarse, pts, viewdirs)
    rgb0, disp0, acc0, weights, depth0 = raw2outputs(raw, z, rays_d, None, white_bkgd)
    z_mid = .5 * (z[..., 1:] + z[..., :-1])
    z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.), u=u)
    z_all, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
    raw = run_network(sd_fine, pts, viewdirs)
    rgb, disp, acc, weights_f, depth = raw2outputs(raw, z_all, rays_d, None, white_bkgd)
    return {
        "rgb_map": rgb, "disp_map": disp, "acc_map": acc, "depth_map": depth, "rgb0": rgb0, "disp0": disp0,
        "acc0": acc0, "z_std": torch.std(z_samples, dim=-1, unbiased=False), "z_samples": z_samples, "z_vals": z_all,
        "weights0": weights,
    }


# ---------------------------------------------------------------------------------------------------------------
# a19  pose utilities   (dataset/load_blender.py:22-28, 359-368)
# ---------------------------------------------------------------------------------------------------------------


def pose_spherical(theta, phi, radius):
    trans_t = lambda t: np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=np.float32)
    rot_phi = lambda ph: np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                                   [0, 0, 0, 1]], dtype=np.float32)
    rot_theta = lambda th: np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                                     [0, 0, 0, 1]], dtype=np.float32)
    c2w = trans_t(radius)
    c2w = rot_phi(phi / 180. * np.pi) @ c2w
    c2w = rot_theta(theta / 180. * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32) @ c2w
    return c2w


# ---------------------------------------------------------------------------------------------------------------
# helpers for tests / bench: seeded weights in the reference's construction order
# ---------------------------------------------------------------------------------------------------------------


def make_state_dict(n_block=43, W=256, input_dim=1008, seed=0, D=None):
    """The weights NeRF_v3_2(args, input_dim, 3) gets under torch.manual_seed(seed), as a state_dict.

    Replays the reference constructor's nn.Linear creation order (nerf_raybased.py:500-537): head, then the legacy
    plain-MLP body that is built and discarded when trial.body_arch == 'resmlp' (:503-505, D-2 Linear(W,W) draws),
    then n_block ResMLP blocks of two Linear(W,W) (:517-524), then the tail.  Same torch build => same numbers
    (the golden fixture stores checksums of these tensors to prove it)."""
    D = 2 * n_block + 2 if D is None else D
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        sd = {}
        head = torch.nn.Linear(input_dim, W)
        for _ in range(1, D - 1):
            torch.nn.Linear(W, W)  # legacy body, discarded by the reference
        blocks = [(torch.nn.Linear(W, W), torch.nn.Linear(W, W)) for _ in range(n_block)]
        tail = torch.nn.Linear(W, 3)
    sd["head.0.weight"], sd["head.0.bias"] = head.weight.detach(), head.bias.detach()
    for b, (l0, l2) in enumerate(blocks):
        sd["body.%d.body.0.weight" % b], sd["body.%d.body.0.bias" % b] = l0.weight.detach(), l0.bias.detach()
        sd["body.%d.body.2.weight" % b], sd["body.%d.body.2.bias" % b] = l2.weight.detach(), l2.bias.detach()
    sd["tail.0.weight"], sd["tail.0.bias"] = tail.weight.detach(), tail.bias.detach()
    return sd


def flatten_state_dict(sd):
    """state_dict-order flat fp32 vector (the layout include/r2l_hip.h calls `params`)."""
    return torch.cat([v.reshape(-1) for v in sd.values()])


def make_teacher_state_dicts(seed, n_nets=2, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), alpha_bias=0.0):
    """State dicts of `n_nets` NeRF(D,W,63,27,use_viewdirs=True) teachers built one after another under
    torch.manual_seed(seed), replaying NeRF.__init__'s nn.Linear creation order (nerf_raybased.py:357-373)."""
    out = []
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        for _ in range(n_nets):
            pts = [torch.nn.Linear(input_ch, W)] + [
                torch.nn.Linear(W, W) if i not in skips else torch.nn.Linear(W + input_ch, W) for i in range(D - 1)
            ]
            views = torch.nn.Linear(input_ch_views + W, W // 2)
            feature = torch.nn.Linear(W, W)
            alpha = torch.nn.Linear(W, 1)
            rgb = torch.nn.Linear(W // 2, 3)
            sd = {}
            for i, l in enumerate(pts):
                sd["pts_linears.%d.weight" % i], sd["pts_linears.%d.bias" % i] = l.weight.detach(), l.bias.detach()
            sd["views_linears.0.weight"], sd["views_linears.0.bias"] = views.weight.detach(), views.bias.detach()
            sd["feature_linear.weight"], sd["feature_linear.bias"] = feature.weight.detach(), feature.bias.detach()
            sd["alpha_linear.weight"], sd["alpha_linear.bias"] = alpha.weight.detach(), alpha.bias.detach() + alpha_bias
            sd["rgb_linear.weight"], sd["rgb_linear.bias"] = rgb.weight.detach(), rgb.bias.detach()
            out.append(sd)
    return out


# ---- SSIM (test-set metric) -------------------------------------------------------------------------------------------
def ssim_window(window_size=11, sigma=1.5):
    """utils/ssim_torch.py:11-25: fp32 Gaussian normalised in fp32; 2-D window = outer product (fp32)."""
    import math
    g = torch.tensor([math.exp(-(x - window_size // 2)**2 / float(2 * sigma**2)) for x in range(window_size)])
    g = g / g.sum()
    return g[:, None] @ g[None, :]


def ssim(img1, img2, window_size=11):
    """utils/ssim_torch.py:28-56 + 86-94 as main.py:46 calls it.  img1, img2: [H, W, C] in [0,1] -> scalar tensor."""
    import torch.nn.functional as F
    a, b = img1.permute(2, 0, 1)[None], img2.permute(2, 0, 1)[None]
    C = a.shape[1]
    w = ssim_window(window_size).to(a)[None, None].expand(C, 1, window_size, window_size).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=window_size // 2, groups=C)
    mu1, mu2 = conv(a), conv(b)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = conv(a * a) - mu1_sq, conv(b * b) - mu2_sq, conv(a * b) - mu12
    C1, C2 = 0.01**2, 0.03**2
    return (((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()
