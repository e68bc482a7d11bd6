"""Host-side mirror of the reference's model layer (/root/reference/model/nerf_raybased.py) for the R2L hot path.

Same public names, constructor arguments, attribute layout and state_dict keys as the reference, so that
  * main.py-style drivers call it unchanged (PointSampler / PositionalEmbedder / NeRF_v3_2 / NeRF),
  * reference `.tar` checkpoints — which pickle the whole NeRF_v3_2 module (main.py:1534-1536) — un-pickle into
    these classes (they are exported as `model.nerf_raybased.*`; __init__ does not run on un-pickle, so all HIP
    state is built lazily by r2l_amd.engine).
On a ROCm device every forward goes through the hand-written HIP kernels (libr2l_hip.so); there is no PyTorch-op
fallback on the GPU.  On CPU tensors (BASELINE config 0, "plumbing, no GPU") the modules run as plain nn.Modules.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine as _engine

_MOD = "model.nerf_raybased"  # the module path the reference pickles its classes under


def default_device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


# ---------------------------------------------------------------------------------------------------------------
# ray -> sample points   (reference: PointSampler, nerf_raybased.py:76-126)
# ---------------------------------------------------------------------------------------------------------------
class PointSampler:
    """Pixel directions of an H x W pinhole camera and n_sample depths in [near, far].

    sample_test(c2w)            -> pts [H*W, 3*n_sample]   (whole frame, reference :94-102)
    sample_train(o, d, perturb) -> pts [N,   3*n_sample]   (explicit rays, stratified jitter if perturb>0, :114-126)
    The fused HIP forward does not call these (it samples in-kernel from the same `dirs`/`z_vals` definition); they
    are the module-boundary API and the CPU path.
    """

    def __init__(self, H, W, focal, n_sample, near, far, device=None):
        dev = default_device() if device is None else torch.device(device)
        self.H, self.W, self.focal = H, W, focal
        self.n_sample, self.near, self.far = n_sample, near, far
        cols = torch.arange(W, dtype=torch.float32).expand(H, W)  # pixel x index i
        rows = torch.arange(H, dtype=torch.float32).unsqueeze(1).expand(H, W)  # pixel y index j
        self.dirs = torch.stack([(cols - W * .5) / focal, -(rows - H * .5) / focal, -torch.ones(H, W)], dim=-1).to(dev)
        t = torch.linspace(0., 1., steps=n_sample)
        self.z_vals = (near * (1 - t) + far * t).to(dev)
        self.z_vals_test = self.z_vals[None, :].expand(H * W, n_sample)

    def rays(self, c2w):
        """rays_o, rays_d [H*W,3] for a camera-to-world matrix c2w[3,4]."""
        c2w = torch.as_tensor(c2w, dtype=torch.float32, device=self.dirs.device)
        rays_d = (self.dirs.unsqueeze(-2) * c2w[:3, :3]).sum(-1).reshape(-1, 3)
        rays_o = c2w[:3, -1].expand(rays_d.shape)
        return rays_o, rays_d

    def sample_test(self, c2w):
        rays_o, rays_d = self.rays(c2w)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * self.z_vals_test[:, :, None]
        return pts.reshape(pts.shape[0], -1)

    def strata(self):
        """(lower, upper) bounds of the per-sample jitter intervals."""
        z = self.z_vals
        mids = .5 * (z[1:] + z[:-1])
        return torch.cat([z[:1], mids]), torch.cat([mids, z[-1:]])

    def sample_train(self, rays_o, rays_d, perturb, t_rand=None):
        n = rays_o.shape[0]
        z = self.z_vals[None, :].expand(n, self.n_sample)
        if perturb > 0.:
            lower, upper = self.strata()
            if t_rand is None:
                t_rand = torch.rand(n, self.n_sample).to(z.device)  # CPU generator stream, as the reference (:122)
            z = lower + (upper - lower) * t_rand
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
        return pts.reshape(n, -1)


# ---------------------------------------------------------------------------------------------------------------
# positional encoding   (reference: PositionalEmbedder, nerf_raybased.py:191-208)
# ---------------------------------------------------------------------------------------------------------------
class PositionalEmbedder:
    """x[N,C] -> [N, C*(2L+1)], per coordinate [sin(2^0 x)..sin(2^(L-1) x), cos(2^0 x)..cos(2^(L-1) x), x]."""

    def __init__(self, L, include_input=True, device=None):
        dev = default_device() if device is None else torch.device(device)
        self.L = L
        self.weights = (2**torch.linspace(0, L - 1, steps=L)).to(dev)
        self.include_input = include_input
        self.embed_dim = 2 * L + 1 if include_input else 2 * L

    def __call__(self, x):
        w = self.weights.to(x.device)
        ang = x.unsqueeze(-1) * w
        parts = [torch.sin(ang), torch.cos(ang)]
        if self.include_input:
            parts.append(x.unsqueeze(-1))
        return torch.cat(parts, dim=-1).reshape(x.shape[0], -1)


# ---------------------------------------------------------------------------------------------------------------
# the student network   (reference: ResMLP :443-465, NeRF_v3_2 :480-544)
# ---------------------------------------------------------------------------------------------------------------
def get_activation(act):
    act = act.lower()
    if act == "relu":
        return nn.ReLU(inplace=True)
    if act == "lrelu":
        return nn.LeakyReLU(inplace=True)
    if act == "none":
        return None
    raise NotImplementedError(act)


class ResMLP(nn.Module):
    """x + res_scale * body(x), body = Linear [act Linear]*(n_learnable-1); optional output activation."""

    def __init__(self, width, inact=nn.ReLU(True), outact=None, res_scale=1, n_learnable=2):
        super().__init__()
        layers = [nn.Linear(width, width)]
        for _ in range(n_learnable - 1):
            if inact is not None:  # --trial.inact none: Linear, Linear with nothing in between (reference :453-455)
                layers.append(inact)
            layers.append(nn.Linear(width, width))
        self.body = nn.Sequential(*layers)
        self.res_scale = res_scale
        self.outact = outact

    def forward(self, x):
        y = self.body(x).mul(self.res_scale) + x
        return y if self.outact is None else self.outact(y)


class NeRF_v3_2(nn.Module):
    """R2L student: head Linear(input_dim,W)+act -> body -> (+ head output if use_residual) -> tail Linear(W,3)+Sigmoid.

    Parameters are named head.0.*, body.<b>.body.{0,2}.*, tail.0.* exactly as the reference so state_dicts,
    optimizers and checkpoint helpers interoperate.  `args` needs: netdepth, netwidth, act, use_residual,
    linear_tail, layerwise_netwidths and (optionally) args.trial.{body_arch,inact,outact,res_scale,n_learnable,n_block}.
    """

    def __init__(self, args, input_dim, output_dim):
        super().__init__()
        self.args = args
        D, W = args.netdepth, args.netwidth
        widths = [int(v) for v in args.layerwise_netwidths.split(",")] + [3] if getattr(
            args, "layerwise_netwidths", "") else [W] * (D - 1) + [3]
        act = get_activation(args.act)
        self.input_dim = input_dim
        self.head = nn.Sequential(nn.Linear(input_dim, widths[0]), act)
        def plain_body():  # the reference default (not accelerated: CPU / torch ops only)
            layers = []
            for i in range(1, D - 1):
                layers += [nn.Linear(widths[i - 1], widths[i]), act]
            return layers

        # The reference always constructs the plain D-2 layer body first and drops it when a --trial body replaces it
        # (model/nerf_raybased.py:502-505, then 508-530).  Those nn.Linear initialisers consume RNG draws, so the same
        # construction is replayed here: under torch.manual_seed(s) every parameter then coincides with the reference's.
        body = plain_body()
        trial = getattr(args, "trial", None)
        if trial is not None and trial.body_arch == "resmlp":
            n_block = trial.n_block if trial.n_block > 0 else (D - 2) // 2
            inact, outact = get_activation(trial.inact), get_activation(trial.outact)
            body = [ResMLP(W, inact=inact, outact=outact, res_scale=trial.res_scale, n_learnable=trial.n_learnable)
                    for _ in range(n_block)]
        elif trial is not None and trial.body_arch == "mlp":
            body = plain_body()  # built a second time there too
        self.body = nn.Sequential(*body)
        if getattr(args, "linear_tail", False):
            self.tail = nn.Linear(input_dim, output_dim)
        else:
            self.tail = nn.Sequential(nn.Linear(widths[D - 2], output_dim), nn.Sigmoid())

    # -- pickling: never serialise the engine (device buffers, ctypes handles) ---------------------------------
    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_r2l_engine", None)
        return state

    # -- torch-op path: CPU plumbing only ------------------------------------------------------------------------
    def _forward_torch(self, x):
        h = self.head(x)
        y = self.body(h) + h if self.args.use_residual else self.body(h)
        return self.tail(y)

    def engine(self):
        return _engine.get_engine(self)

    def forward(self, x):
        """x: embedded input [N, input_dim] (or NHWC-permutable [N,C,H,W], as the reference's ONNX path)."""
        if x.shape[-1] != self.input_dim:
            x = x.permute(0, 2, 3, 1)
        if not x.is_cuda:
            return self._forward_torch(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import R2LEmbFunction
            return R2LEmbFunction.apply(self, x, *list(self.parameters()))
        lead = x.shape[:-1]
        return self.engine().forward_emb(x.reshape(-1, self.input_dim)).reshape(*lead, 3)

    # -- fused entry points (what the build's own drivers call) -----------------------------------------------------
    def forward_rays(self, rays_o, rays_d, point_sampler, perturb=0., t_rand=None):
        """rgb[N,3] = self(embed(point_sampler.sample_train(o, d, perturb))) with sampling + encoding fused in-kernel."""
        if not rays_o.is_cuda:
            pe = PositionalEmbedder(_engine.L_PE, device=rays_o.device)
            return self._forward_torch(pe(point_sampler.sample_train(rays_o, rays_d, perturb, t_rand)))
        return self.engine().forward_rays(rays_o, rays_d, point_sampler.z_vals, perturb, t_rand)

    def render_pose(self, c2w, point_sampler):
        """rgb[H*W,3] for a frame: self(embed(point_sampler.sample_test(c2w))) fused (main.py:401-404 render_func)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            pe = PositionalEmbedder(_engine.L_PE, device=dev)
            return self._forward_torch(pe(point_sampler.sample_test(c2w)))
        ps = point_sampler
        return self.engine().forward_pose(c2w, ps.H, ps.W, ps.focal, ps.z_vals)

    def render_poses(self, c2ws, point_sampler):
        """rgb[K, H*W, 3] for K frames in one launch (the test-set loop of main.py:300-309 without per-frame launches)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return torch.stack([self.render_pose(c, point_sampler) for c in c2ws], 0)
        ps = point_sampler
        return self.engine().forward_poses(c2ws, ps.H, ps.W, ps.focal, ps.z_vals)


# ---------------------------------------------------------------------------------------------------------------
# the teacher network   (reference: NeRF, nerf_raybased.py:337-401)
# ---------------------------------------------------------------------------------------------------------------
class NeRF(nn.Module):
    """Original NeRF MLP: D pts layers of width W (input re-concatenated after layers in `skips`), then either a
    single output layer or (use_viewdirs) alpha / feature heads + a W//2 view-dependent layer + rgb."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        layers = [nn.Linear(input_ch, W)]
        for i in range(D - 1):
            layers.append(nn.Linear(W + input_ch if i in skips else W, W))
        self.pts_linears = nn.ModuleList(layers)
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def forward(self, x):
        pts, views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = pts
        for i, lin in enumerate(self.pts_linears):
            h = F.relu(lin(h))
            if i in self.skips:
                h = torch.cat([pts, h], -1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), views], -1)
        for lin in self.views_linears:
            h = F.relu(lin(h))
        return torch.cat([self.rgb_linear(h), alpha], -1)


for _cls in (ResMLP, NeRF_v3_2, NeRF):
    _cls.__module__ = _MOD  # pickle these under the reference's module path (checkpoint compatibility)
