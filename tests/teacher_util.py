"""Test infrastructure: a NeRF teacher whose weights have LEFT the init distribution.

No trained lego checkpoint can be downloaded here, so the teacher is fitted for a couple of thousand Adam steps (plain torch
autograd on the oracle's functional forward, on whatever device the caller picks) to an analytic scene with the statistics a
trained NeRF shows: sigma pre-activations from about -10^2 in empty space to 10^2 .. 10^3 inside surfaces, a sharp shell and
an opaque blob, view-dependent colour.  The kernels under test never see this code; they get the resulting state dict."""
import torch

from oracle import r2l_oracle as O


def analytic_targets(pts, dirs):
    """(logit-rgb [n,3], sigma pre-activation [n]) of the analytic scene at points / unit view directions."""
    r = pts.norm(dim=-1)
    shell = 600.0 * torch.exp(-((r - 1.0) / 0.06)**2)
    blob = 1000.0 * torch.sigmoid((0.35 - (pts - torch.tensor([0.3, -0.2, 0.4], device=pts.device)).norm(dim=-1)) / 0.02)
    sigma = shell + blob - 80.0 * torch.sigmoid((r - 1.3) / 0.1) - 20.0
    base = torch.stack([torch.sin(3.1 * pts[:, 0] + 0.3), torch.cos(2.3 * pts[:, 1] - 0.5), torch.sin(1.7 * pts[:, 2] + pts[:, 0])], -1)
    spec = (dirs * torch.tensor([0.6, -0.3, 0.5], device=pts.device)).sum(-1, keepdim=True)
    return 2.5 * base + 1.5 * spec, sigma


def fit_teacher(seed, steps=1500, n=8192, device="cpu", lr=5e-4):
    """state dict (CPU tensors) of one NeRF(D=8, W=256, skips=[4], use_viewdirs) fitted to analytic_targets."""
    sd0 = O.make_teacher_state_dicts(seed, 1, alpha_bias=0.5)[0]
    sd = {k: v.clone().to(device).requires_grad_(True) for k, v in sd0.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=lr)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for it in range(steps):
        pts = ((torch.rand(n, 3, generator=g) * 2 - 1) * 1.6).to(device)
        dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(device)
        rgb_t, sig_t = analytic_targets(pts, dirs)
        emb = torch.cat([O.nerf_embed(pts, 10), O.nerf_embed(dirs, 4)], -1)
        out = O.nerf_forward(sd, emb)
        loss = ((out[:, :3] - rgb_t)**2).mean() + ((out[:, 3] - sig_t)**2).mean() * 1e-3
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    return {k: v.detach().cpu().contiguous() for k, v in sd.items()}


def teacher_stats(sd, n=4096, seed=0):
    """(max |hidden activation|, min sigma pre-activation, max sigma pre-activation) on random points of the scene box."""
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(n, 3, generator=g) * 2 - 1) * 1.6
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    emb = torch.cat([O.nerf_embed(pts, 10), O.nerf_embed(dirs, 4)], -1)
    h, hmax = emb[:, :63], 0.
    for i in range(8):
        h = torch.relu(torch.nn.functional.linear(h, sd["pts_linears.%d.weight" % i], sd["pts_linears.%d.bias" % i]))
        hmax = max(hmax, h.abs().max().item())
        if i == 4:
            h = torch.cat([emb[:, :63], h], -1)
    out = O.nerf_forward(sd, emb)
    return hmax, out[:, 3].min().item(), out[:, 3].max().item()
