"""R2L driver: the `python main.py --model_name R2L ...` surface (render_only / render_test / --benchmark / distillation
training with hard-ray mining / checkpoints), mirroring the control flow of the reference's main.py:888-1542 for the
accelerated path only.  One process per GPU; under torchrun the ray shards, test frames and log output are
rank-partitioned and the only collective is the flat-gradient all-reduce inside R2LTrainer.
"""
import ctypes
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import data as D
from .checkpoint import load_ckpt, load_weights_v2, parse_expid_iter, save_ckpt
from .logger import Logger
from .metrics import img2mse, mse2psnr, ssim, to8b
from .nerf_raybased import NeRF_v3_2, PointSampler, PositionalEmbedder
from .options import parse_args, validate_accelerated
from .dist_utils import split_shards
from .train_step import R2LTrainer, lr_schedule


# ---------------------------------------------------------------------------------------------------------------
def init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
        device = torch.device("cuda", local % torch.cuda.device_count())
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # (R2L_DIST_BACKEND=gloo: two ranks on ONE GPU in tests — RCCL refuses a device twice, gloo moves CUDA tensors through the host)
        dist.init_process_group(os.environ.get("R2L_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo"))
    return rank, world, device


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize()


class HardRayPool:
    """Hard-example pool of main.py:1164-1165, 1325-1347, 1410-1425: after each step the hard_ratio*B rays with the
    largest per-ray MSE enter the pool (appended until it holds >= B*hard_mul rows, then replacing random rows), and
    n_hard_out random pool rows are appended to every batch once the pool is full.

    The pool ([rows, 9] = o,d,rgb; 59 MB at the README sizes) lives on the device of the rays it is fed, is allocated
    once at its final size, and on a GPU the random row choice is a device randperm: the reference's host-side
    np.random.permutation(1.6 M) per step costs as much as a whole MI355X training step."""

    def __init__(self, hard_ratio, hard_mul, rng=None, seed=0):
        self.ratio, self.mul = hard_ratio, hard_mul
        self.pool = None      # rows filled so far (a view of _store once allocated)
        self._store = None
        self.full = False
        self.rng = rng or np.random
        self.seed = seed
        self._draws = 0
        self._ix_out = None

    def sizes(self, batch_size):
        if isinstance(self.ratio, list):
            n_in, n_out = int(self.ratio[0] * batch_size), int(self.ratio[1] * batch_size)
        else:
            n_in = n_out = int(self.ratio * batch_size)
        return min(n_in, n_out), n_out

    def _pick(self, n_rows, n_out, device):
        if device.type == "cuda":
            # n_out distinct rows, every row equally likely: a keyed bijection of [0, n_rows) evaluated at 0 .. n_out-1
            # (include/r2l_hip.h r2l_pool_pick) instead of a 1.6 M-key randperm per step; key = f(seed, draw counter)
            from . import _lib
            lib = _lib.load()
            ix = torch.empty(n_out, dtype=torch.int64, device=device)
            self._draws += 1
            key = (self.seed * 0x9E3779B97F4A7C15 + self._draws * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            _lib.check(lib.r2l_pool_pick(ctypes.c_void_p(ix.data_ptr()), n_out, n_rows, key,
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "r2l_pool_pick")
            return ix
        return torch.as_tensor(self.rng.permutation(n_rows)[:n_out], device=device)

    @staticmethod
    def _rows(x):
        """(tensor, row stride in floats) of a [N,3] fp32 tensor whose rows are contiguous (column slices of a shard batch)."""
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != 3 or x.stride(1) != 1:
            x = x.float().contiguous()
        return x, x.stride(0)

    def extra_rays(self, batch_size, updates_done):
        """Rays augment() appends to a batch of `batch_size` after `updates_done` update() calls on batches of that size — a
        pure function of the two, so every rank of a data-parallel run can tell every other rank's step size without a
        collective (driver.train: global ray count of a step when --N_rand does not divide by the world size)."""
        n_in, n_out = self.sizes(batch_size)
        if n_in <= 0:
            return 0
        steps_to_full = max(1, -(-int(np.ceil(batch_size * self.mul)) // n_in))
        return n_out if updates_done >= steps_to_full else 0

    def augment(self, rays_o, rays_d, target):
        if not self.full:
            return rays_o, rays_d, target
        _, n_out = self.sizes(rays_o.shape[0])
        self._ix_out = self._pick(self.pool.shape[0], n_out, self.pool.device)
        if self.pool.is_cuda:  # one kernel: batch rows (possibly column slices of the [B,9] shard batch) + picked pool rows
            from . import _lib
            lib, B = _lib.load(), rays_o.shape[0]
            (o, so), (d, sd), (t, st) = self._rows(rays_o), self._rows(rays_d), self._rows(target)
            out = torch.empty(3, B + n_out, 3, dtype=torch.float32, device=self.pool.device)
            p = lambda x: ctypes.c_void_p(x.data_ptr())
            _lib.check(lib.r2l_pool_augment(p(o), p(d), p(t), so, sd, st, p(self.pool), p(self._ix_out), B, n_out, p(out[0]),
                                            p(out[1]), p(out[2]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "r2l_pool_augment")
            return out[0], out[1], out[2]
        picked = self.pool[self._ix_out]
        return (torch.cat([rays_o, picked[:, :3]], 0), torch.cat([rays_d, picked[:, 3:6]], 0),
                torch.cat([target, picked[:, 6:]], 0))

    def update(self, rgb, rays_o, rays_d, target, batch_size):
        n_in, _ = self.sizes(batch_size)
        if n_in <= 0:
            return
        err = torch.mean((rgb[:batch_size] - target[:batch_size])**2, dim=1)
        _, order = torch.sort(err)
        hard = order[-n_in:]
        if self._store is None:  # final size: the first multiple of n_in that reaches batch_size * hard_mul
            steps = max(1, -(-int(np.ceil(batch_size * self.mul)) // n_in))
            self._store = torch.empty(steps * n_in, 9, dtype=torch.float32, device=rays_o.device)
            self._n = 0
        if rays_o.is_cuda:  # one kernel: gather the hard rows [o, d, rgb] and put them where they go (replace / append)
            from . import _lib
            lib = _lib.load()
            (o, so), (d, sd), (t, st) = self._rows(rays_o), self._rows(rays_d), self._rows(target)
            hard = hard.contiguous()
            dst = self._ix_out[:n_in].contiguous() if self.full else None
            p = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
            _lib.check(lib.r2l_pool_store(p(o), p(d), p(t), so, sd, st, p(hard), p(self._store), p(dst), 0 if self.full else self._n,
                                          n_in, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "r2l_pool_store")
            if self.full:
                return
        else:
            rows = torch.cat([rays_o[hard], rays_d[hard], target[hard]], dim=-1)
            if self.full:
                self.pool[self._ix_out[:n_in]] = rows
                return
            self._store[self._n:self._n + n_in] = rows
        self._n += n_in
        self.pool = self._store[:self._n]
        if self._n >= batch_size * self.mul:
            self.full = True


# ---------------------------------------------------------------------------------------------------------------
def create_r2l(args, device, logger):
    """Model + (optional) checkpoint, following main.py:455-509: a checkpoint that carries the pickled `network_fn`
    REPLACES the freshly constructed module before its state dict is loaded."""
    embedder = PositionalEmbedder(L=args.multires, device=device)
    input_dim = args.n_sample_per_ray * 3 * embedder.embed_dim
    model = NeRF_v3_2(args, input_dim, 3).to(device)
    history = {"start": 0, "best_psnr": 0, "best_psnr_step": 0}
    ckpt = None
    if args.pretrained_ckpt:
        ckpt = load_ckpt(args.pretrained_ckpt, map_location=device)
        if "network_fn" in ckpt:
            model = ckpt["network_fn"].to(device)
            logger.info('Use model arch saved in checkpoint "%s"' % args.pretrained_ckpt)
        load_weights_v2(model, ckpt, "network_fn_state_dict")
        logger.info('Load pretrained ckpt successfully: "%s".' % args.pretrained_ckpt)
        if args.resume:
            history.update(start=ckpt["global_step"], best_psnr=ckpt.get("best_psnr", 0),
                           best_psnr_step=ckpt.get("best_psnr_step", 0))
    n_params = sum(p.numel() for p in model.parameters())
    macs = sum(m.in_features * m.out_features for m in model.modules() if isinstance(m, torch.nn.Linear))
    logger.info("Model complexity per pixel: FLOPs %.10fM, Params %.10fM" % (macs / 1e6, n_params / 1e6))
    return model, embedder, history, ckpt


def create_nerf_teacher(args, device, logger, near, far):
    """The `model_name in ['nerf']` branch of create_nerf (main.py:407-453, 481-509, 511-541) for rendering: coarse NeRF (+ the
    fine one when N_importance > 0) built from --netdepth/--netwidth(/_fine), weights from --pretrained_ckpt through
    load_weights_v2 ('network_fn_state_dict' / 'network_fine_state_dict'; a checkpoint that carries pickled modules replaces the
    constructed ones), and render_kwargs_test: perturb = --perturb_test, raw_noise_std = 0, near / far of the blender scenes.
    Teacher TRAINING is out of scope: no optimizer, parameters frozen."""
    from .nerf_raybased import NeRF
    from .render import get_embedder, run_network
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = (get_embedder(args.multires_views, args.i_embed) if args.use_viewdirs else (None, 0))
    output_ch = 5 if args.N_importance > 0 else 4
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=[4],
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch, skips=[4],
                          input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    if args.pretrained_ckpt:
        ckpt = load_ckpt(args.pretrained_ckpt, map_location=device)
        if "network_fn" in ckpt:
            model = ckpt["network_fn"].to(device)
            if model_fine is not None:
                assert "network_fine" in ckpt
                model_fine = ckpt["network_fine"].to(device)
            logger.info('Use model arch saved in checkpoint "%s"' % args.pretrained_ckpt)
        load_weights_v2(model, ckpt, "network_fn_state_dict")
        if model_fine is not None:
            load_weights_v2(model_fine, ckpt, "network_fine_state_dict")
        logger.info('Load pretrained ckpt successfully: "%s".' % args.pretrained_ckpt)
    for net in (model, model_fine):
        if net is not None:
            for p in net.parameters():
                p.requires_grad = False
    qfn = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                   netchunk=args.netchunk)
    n_params = sum(p.numel() for p in model.parameters())
    macs = sum(m.in_features * m.out_features for m in model.modules() if isinstance(m, torch.nn.Linear))
    macs *= args.N_samples + args.N_samples + args.N_importance  # as main.py:547-549
    logger.info("Model complexity per pixel: FLOPs %.10fM, Params %.10fM" % (macs / 1e6, n_params / 1e6))
    # render_kwargs_test (main.py:511-541) + bds_dict (main.py:977-982)
    return dict(network_query_fn=qfn, perturb=args.perturb_test, N_importance=args.N_importance, network_fine=model_fine,
                N_samples=args.N_samples, network_fn=model, use_viewdirs=args.use_viewdirs, white_bkgd=args.white_bkgd,
                raw_noise_std=0., ndc=False, lindisp=args.lindisp, near=near, far=far)


def apply_arithmetic(args, device, logger, student=None, teachers=()):
    """--r2l_precision / --r2l_dw_mode (options.py; include/r2l_hip.h r2l_config) -> the engines of this run, through
    R2LEngine.set_config / TeacherEngine.set_config — arguments of the *_cfg entry points, not environment switches.  Returns the
    record {'precision', 'dw_mode', 'requested': {...}} that is logged here and stored in every checkpoint the run writes (key
    'r2l_config': the reference's loaders never look at it, main.py:481-509).  CPU (config 0, plumbing): torch fp32 ops."""
    req = {"precision": args.r2l_precision, "dw_mode": args.r2l_dw_mode}
    if device.type != "cuda":
        rec = {"precision": "torch fp32 (CPU plumbing)", "dw_mode": "autograd fp32", "requested": req}
    else:
        from . import engine as _engine
        cfg = _lib_config(req)
        if student is not None:
            student.engine().set_config(precision=req["precision"], dw_mode=req["dw_mode"])
        for net in teachers:
            if net is not None:
                from .render import teacher_engine
                teacher_engine(net).set_config(precision=req["precision"])
        rec = dict(_engine.arithmetic(cfg), requested=req)
    logger.info("r2l_config: precision %s, dw_mode %s (requested: --r2l_precision %s --r2l_dw_mode %s)" %
                (rec["precision"], rec["dw_mode"], req["precision"], req["dw_mode"]))
    return rec


def _lib_config(req):
    from . import _lib
    return _lib.make_config(precision=req["precision"], dw_mode=req["dw_mode"])


POSES_PER_LAUNCH = 9


def render_frame(model, point_sampler, c2w):
    """One frame [H,W,3] (main.py:300-324 R2L branch): fused sample -> encode -> network."""
    with torch.no_grad():
        rgb = model.render_pose(c2w[:3, :4], point_sampler)
    return rgb.view(point_sampler.H, point_sampler.W, 3)


class _FrameWriter:
    """PNG writing off the render loop (replaces the inline imageio.imwrite of main.py:337-344): the frame is quantised to 8
    bit on the device (same rounding as to8b), copied non-blocking into a pinned buffer, and handed — with the event behind
    the copy — to the library's encoder threads (include/r2l_hip.h r2l_png_writer_*: native threads, zlib level 1, no GIL).
    Rounds 1 - 3 encoded with PIL in Python threads: ~15 ms per 400x400 PNG with the GIL held around zlib, which made the
    test-set loop encoder-bound (6.5 ms/frame against 4.5 without images, profiles/r03_e2e_render.txt)."""

    # With 9 frames per launch 27 images arrive at once (frame, target, error): the pinned staging slots must cover two groups, or
    # save() blocks on the encoders before the next group's render is launched.
    def __init__(self, device, workers=None, slots=6 * POSES_PER_LAUNCH + 4, level=1):
        from . import _lib
        self._lib, self.lib = _lib, _lib.load()
        self.device, self.slots = device, slots
        n = workers or max(4, min(32, os.cpu_count() or 8))
        self._h = ctypes.c_void_p()
        _lib.check(self.lib.r2l_png_writer_open(n, level, ctypes.byref(self._h)), "r2l_png_writer_open")
        self.inflight = []  # (job id, what must stay alive until it is done, recyclable pinned buffer or None)
        self.free = {}

    def _submit(self, path, ptr, shape, event):
        H, W = int(shape[0]), int(shape[1])
        C = int(shape[2]) if len(shape) == 3 else 1
        job = ctypes.c_int64()
        self._lib.check(self.lib.r2l_png_writer_submit(self._h, os.fsencode(path), ctypes.c_void_p(ptr), H, W, C,
                                                       ctypes.c_void_p(event) if event else None, ctypes.byref(job)),
                        "r2l_png_writer_submit")
        return job.value

    def _retire(self, n_keep):
        while len(self.inflight) > n_keep:
            job, _, host = self.inflight.pop(0)
            try:
                self._lib.check(self.lib.r2l_png_writer_wait(self._h, job), "r2l_png_writer_wait")
            finally:  # (a failed write still returns its pinned staging slot: the writer stays usable)
                if host is not None:
                    self.free.setdefault(tuple(host.shape), []).append(host)

    def save(self, img, path):
        """img: float [H,W,3] in [0,1] (device or host tensor / array)."""
        if isinstance(img, torch.Tensor) and img.is_cuda:
            q = (255 * torch.clamp(img, 0, 1)).to(torch.uint8).contiguous()  # float -> uint8 truncates, as numpy's astype does
            key = tuple(q.shape)
            free = self.free.get(key)
            if free is None:  # ONE pinned allocation for all staging slots of this frame size (44 separate ones cost ~100 ms)
                slab = torch.empty((self.slots,) + key, dtype=torch.uint8, pin_memory=True)
                free = self.free[key] = [slab[i] for i in range(self.slots)]
            if not free:
                self._retire(self.slots - 1)
            if not free:  # (every slot of this size still in flight behind jobs of another size: wait for all)
                self._retire(0)
            host = free.pop()
            host.copy_(q, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.inflight.append((self._submit(path, host.data_ptr(), key, ev.cuda_event), (ev, q), host))
        else:
            arr = np.ascontiguousarray(to8b(img))
            self.inflight.append((self._submit(path, arr.ctypes.data, arr.shape, None), arr, None))

    def flush(self):
        """Every frame handed over so far is on disk (the writer stays usable: its threads and pinned slots are kept)."""
        self._retire(0)

    def close(self):
        if not self._h:
            return
        self._retire(0)
        self._lib.check(self.lib.r2l_png_writer_close(self._h), "r2l_png_writer_close")
        self._h = ctypes.c_void_p()

    _shared = {}

    @classmethod
    def shared(cls, device, workers=None):
        """One writer per device and process, kept across render_path calls (the test set is evaluated every i_testset
        iterations: thread start-up and the pinned allocation are paid once)."""
        key = (str(device), workers)
        w = cls._shared.get(key)
        if w is None:
            import atexit
            w = cls._shared[key] = cls(device, workers=workers)
            atexit.register(w.close)
        return w


def save_video(rgbs, logger, expid, iter_, tag, rank=0, world=1, device=None):
    """`video_<expid>_iter<k>_<tag>` of main.py:1096-1097 / 1483-1484 from the frames render_path returned (this rank's
    poses[rank::world]; with world > 1 the ranks' frames are gathered to rank 0 and re-interleaved).  Container: Motion-JPEG
    AVI instead of the reference's mp4 (video.py says why).  Returns the path on rank 0, None elsewhere."""
    from .video import write_mjpeg_avi
    frames = to8b(rgbs) if rgbs.numel() else np.zeros((0, 0, 0, 3), np.uint8)
    if world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(frames, parts, dst=0)  # only rank 0 needs (and holds) the other ranks' frames
        if rank != 0:
            return None
        total = sum(p.shape[0] for p in parts)
        hw = next((p.shape[1:] for p in parts if p.shape[0]), (0, 0, 3))
        allf = np.zeros((total,) + tuple(hw), np.uint8)
        for r, p in enumerate(parts):
            if p.shape[0]:  # (a rank beyond the number of poses rendered nothing)
                allf[r::world] = p
        frames_np = allf
    else:
        frames_np = frames
    path = os.path.join(logger.gen_img_path, "video_%s_iter%s_%s.avi" % (expid, iter_, tag))
    os.makedirs(logger.gen_img_path, exist_ok=True)
    write_mjpeg_avi(path, frames_np, fps=30, quality=8)
    return path


def render_path(poses, model, point_sampler, device, logger, gt_imgs=None, savedir=None, rank=0, world=1, teacher=None):
    """Render poses[rank::world]; returns (rgbs [n,H,W,3], misc with test_loss/test_psnr/test_psnr_v2 over ALL frames)
    and test_ssim — main.py:189-398 (LPIPS/FLIP need network weights / packages that are absent: out of scope).  No host
    sync inside the loop: metrics stay on the device, frames are written by _FrameWriter, per-frame times come from device
    events and are logged after the loop.
    teacher = None: the R2L branch (main.py:284-324), `model` = the student.
    teacher = dict(hwf=(H, W, focal), chunk=, render_kwargs=): the `model_name in ['nerf']` branch (main.py:275-282): every
    frame is render(H, W, focal, chunk, c2w=pose[:3,:4], **render_kwargs) of r2l_amd/render.py (coarse + fine NeRF on the
    teacher kernels); `model` = render_kwargs['network_fn'], `point_sampler` unused."""
    model.eval()
    mine = list(range(rank, len(poses), world))
    rgbs, sq_err, psnrs, ssims, events, errors = [], [], [], [], [], []
    if savedir is not None:
        os.makedirs(savedir, exist_ok=True)  # every rank writes its own frames: none may rely on rank 0's mkdir
    writer = _FrameWriter.shared(device, workers=int(os.environ.get("R2L_PNG_WORKERS", "0")) or None) if savedir is not None else None
    on_gpu = device.type == "cuda"
    t_loop = time.time()
    def account(i, rgb):
        rgbs.append(rgb)
        if gt_imgs is not None:
            gt = gt_imgs[i].to(rgb.device, non_blocking=True)
            if gt.shape[:2] != rgb.shape[:2]:  # --render_factor (teacher branch): the reference CROPS the target, main.py:329-333
                gt = gt[:rgb.shape[0], :rgb.shape[1]].contiguous()
            errors.append((rgb - gt).abs())  # misc['errors'] (main.py:330, 386): this rank's frames; main() writes their video
            mse = img2mse(rgb, gt)
            sq_err.append(mse)
            psnrs.append(mse2psnr(mse))
            ssims.append(ssim(rgb, gt))
        if writer is not None:
            writer.save(rgb, os.path.join(savedir, "%03d.png" % i))
            if gt_imgs is not None:
                writer.save(gt_imgs[i], os.path.join(savedir, "%03d_gt.png" % i))
                # |rgb - gt| as an image, as the reference saves it beside every test frame (main.py:330, 342-344)
                writer.save(errors[-1], os.path.join(savedir, "%03d_error.png" % i))

    if teacher is not None:
        from .render import render
        H, W, focal = teacher["hwf"]
        rf = teacher.get("render_factor", 0)
        if rf != 0:  # "Render downsampled for speed" (main.py:197-201); the R2L branch's sampler is built for the full frame
            H, W, focal = int(H / rf), int(W / rf), focal / rf
        # a whole frame per launch on the GPU (as create_data.main: --chunk is a memory work-around of the op-by-op path)
        chunk = max(int(teacher["chunk"]), H * W) if on_gpu else int(teacher["chunk"])
        for i in mine:
            t0 = time.time()
            if on_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            c2w = torch.as_tensor(poses[i], dtype=torch.float32)[:3, :4].to(device)
            with torch.no_grad():
                rgb, _disp, _acc, _extras = render(H, W, focal, chunk=chunk, c2w=c2w, **teacher["render_kwargs"])
            if on_gpu:
                e1.record()
                events.append(([i], e0, e1))
            else:
                logger.info("[#%d] frame, rendering done, time for this frame: %.4fs" % (i, time.time() - t0))
            account(i, rgb)
    elif on_gpu:
        # POSES_PER_LAUNCH frames per launch (engine.forward_poses: no launch gap and no partly filled last round of
        # workgroups per frame; 9 x 1250 workgroups = 43.95 rounds of the 256 CUs at 400x400)
        for g0 in range(0, len(mine), POSES_PER_LAUNCH):
            idx = mine[g0:g0 + POSES_PER_LAUNCH]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            c2ws = torch.stack([torch.as_tensor(poses[i], dtype=torch.float32)[:3, :4] for i in idx], 0)
            with torch.no_grad():
                frames = model.render_poses(c2ws, point_sampler).view(len(idx), point_sampler.H, point_sampler.W, 3)
            e1.record()
            events.append((idx, e0, e1))
            for k, i in enumerate(idx):
                account(i, frames[k])
    else:
        for i in mine:
            t0 = time.time()
            rgb = render_frame(model, point_sampler, poses[i])
            logger.info("[#%d] frame, rendering done, time for this frame: %.4fs" % (i, time.time() - t0))
            account(i, rgb)
    if writer is not None:
        writer.flush()  # (shared writer: threads and pinned slots stay for the next evaluation)
    if on_gpu:
        sync(device)
        for idx, e0, e1 in events:
            for i in idx:  # (one launch per group of frames: its time divided evenly)
                logger.info("[#%d] frame, rendering done, time for this frame: %.4fs" % (i, e0.elapsed_time(e1) * 1e-3 / len(idx)))
        if mine:
            logger.info("%d frames in %.3fs wall (%.1f ms/frame incl. metrics and image writing)" %
                        (len(mine), time.time() - t_loop, (time.time() - t_loop) * 1e3 / len(mine)))
    rgbs = torch.stack(rgbs, 0) if rgbs else torch.empty(0)
    misc = {}
    if gt_imgs is not None:
        sums = [torch.stack(v).double().sum() if v else torch.zeros((), dtype=torch.float64, device=device)
                for v in (sq_err, psnrs, ssims)]
        stats = torch.stack([sums[0].reshape(()), sums[1].reshape(()),
                             torch.tensor(float(len(mine)), dtype=torch.float64, device=device),
                             sums[2].reshape(())]).to(device)
        if world > 1:
            dist.all_reduce(stats)  # host-side metric gather (4 scalars), not on the data path
        stats = stats.tolist()
        misc["test_loss"] = torch.tensor(stats[0] / max(stats[2], 1))
        misc["test_psnr"] = mse2psnr(misc["test_loss"].float()).squeeze()
        misc["test_psnr_v2"] = torch.tensor(stats[1] / max(stats[2], 1))
        misc["test_ssim"] = torch.tensor(stats[3] / max(stats[2], 1))
        misc["errors"] = torch.stack(errors, 0) if errors else torch.empty(0)
    model.train()
    return rgbs, misc


# ---------------------------------------------------------------------------------------------------------------
def main(argv=None):
    args = parse_args(argv)
    validate_accelerated(args)
    is_teacher = args.model_name == "nerf"
    if is_teacher and (not args.render_only or args.benchmark):
        # (--benchmark times render_func = the R2L student's forward, main.py:401-404,1124-1133)
        raise NotImplementedError("--model_name nerf: the accelerated path renders a pretrained teacher (--render_only, with or "
                                  "without --render_test / --test_pretrained) and uses it in utils/create_data.py; TRAINING "
                                  "the teacher is out of scope (SURVEY.md §2)")
    rank, world, device = init_distributed()
    np.random.seed(0)
    # every rank must build the same student: torch's default generator is seeded per process otherwise (the reference had
    # ONE module that nn.DataParallel re-broadcast every step, main.py:472-479); R2LTrainer additionally broadcasts rank 0's
    # flat parameter buffer once, so a checkpoint-less start is identical on all ranks by construction
    torch.manual_seed(int(os.environ.get("R2L_SEED", "0")))
    logger = Logger(args, rank)

    images, poses, render_poses, hwf, i_split = D.load_blender_data(args.datadir, args.half_res, args.testskip)
    logger.info("Loaded blender", tuple(images.shape), tuple(poses.shape), hwf, args.datadir)
    i_train, i_val, i_test = i_split
    near, far = 2., 6.
    if hasattr(args, "trial") and args.trial.near > 0:
        assert args.trial.far > args.trial.near
        near, far = args.trial.near, args.trial.far
    images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:]) if args.white_bkgd else images[..., :3]
    H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    if args.focal_scale > 0:
        focal *= args.focal_scale

    teacher = None
    if is_teacher:
        kwargs_test = create_nerf_teacher(args, device, logger, near, far)
        model, point_sampler = kwargs_test["network_fn"], None
        teacher = dict(hwf=(H, W, focal), chunk=args.chunk, render_kwargs=kwargs_test, render_factor=args.render_factor)
        history = {"start": 0, "best_psnr": 0, "best_psnr_step": 0}
        r2l_config = apply_arithmetic(args, device, logger, teachers=(model, kwargs_test["network_fine"]))
    else:
        model, embedder, history, ckpt = create_r2l(args, device, logger)
        point_sampler = PointSampler(H, W, focal, args.n_sample_per_ray, near, far, device=device)
        r2l_config = apply_arithmetic(args, device, logger, student=model)
        if ckpt is not None and ckpt.get("r2l_config"):
            was = ckpt["r2l_config"]
            logger.info("checkpoint was trained with r2l_config: precision %s, dw_mode %s" % (was.get("precision"), was.get("dw_mode")))
    test_poses, test_images = poses[i_test], images[i_test]
    video_poses = D.get_novel_poses(args, n_pose=args.n_pose_video)
    start, best_psnr, best_psnr_step = history["start"], history["best_psnr"], history["best_psnr_step"]

    if args.test_pretrained:
        _, misc = render_path(test_poses, model, point_sampler, device, logger, gt_imgs=test_images, rank=rank,
                              world=world, teacher=teacher)
        logger.info("Pretrained test: TestLoss %.4f TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" %
                    (misc["test_loss"].item(), misc["test_psnr"].item(), misc["test_psnr_v2"].item(),
                     misc["test_ssim"].item()))

    if args.render_only:
        logger.info("RENDER ONLY")
        expid, iter_ = parse_expid_iter(args.pretrained_ckpt)
        t_ = time.time()
        if args.render_test:
            rgbs, misc = render_path(test_poses, model, point_sampler, device, logger, gt_imgs=test_images,
                                     savedir=logger.gen_img_path if rank == 0 or world > 1 else None, rank=rank,
                                     world=world, teacher=teacher)
            logger.info("[TEST] TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f" %
                        (misc["test_psnr"].item(), misc["test_psnr_v2"].item(), misc["test_ssim"].item()))
        else:
            rgbs, misc = render_path(video_poses, model, point_sampler, device, logger, savedir=logger.gen_img_path,
                                     rank=rank, world=world, teacher=teacher)
        n_rays = rgbs.numel() // 3 if rgbs.numel() else 0  # (frames x rendered pixels: --render_factor shrinks the teacher's frames)
        dt = time.time() - t_
        logger.info("Rendered %d frames (%d rays) in %.2fs on rank %d = %.0f rays/s incl. I/O; frames in %s" %
                    (rgbs.shape[0], n_rays, dt, rank, n_rays / max(dt, 1e-9), logger.gen_img_path))
        # (the reference writes the video of whichever frames it rendered — test views too, main.py:1096-1097)
        video_path = save_video(rgbs, logger, expid, iter_, args.video_tag, rank, world, device)
        if "errors" in misc:  # (main.py:1098-1102: the |rgb - gt| frames as a second video)
            save_video(misc["errors"], logger, expid, iter_, args.video_tag + "_error", rank, world, device)
        return {"misc": misc, "rgbs": rgbs, "logger": logger, "video_path": video_path, "r2l_config": r2l_config}

    if args.benchmark:
        # torch.utils.benchmark.Timer('render_func(model, pose)').timeit(100) in the reference (main.py:1124-1133)
        pose = video_poses[0]
        for _ in range(3):
            render_frame(model, point_sampler, pose)
        sync(device)
        t0 = time.time()
        n = 100 if device.type == "cuda" else 2
        for _ in range(n):
            render_frame(model, point_sampler, pose)
        sync(device)
        per = (time.time() - t0) / n
        logger.info("render_func(model, pose): %.3f ms per %dx%d frame = %.3f M rays/s" % (per * 1e3, H, W,
                                                                                           H * W / per / 1e6))
        return {"ms_per_frame": per * 1e3, "logger": logger}

    # ---------------- distillation training (data_mode rays) ----------------
    if args.data_mode != "rays" or not args.datadir_kd:
        raise NotImplementedError("training on the accelerated path needs --data_mode rays --datadir_kd <dir of "
                                  "[4096,9] .npy ray shards> (README step 3/5)")
    if device.type != "cuda":
        raise RuntimeError("R2L training runs on the HIP path and needs a ROCm GPU")
    datadir_kd = args.datadir_kd.split(":")[1] if ":" in args.datadir_kd else args.datadir_kd
    files = D.list_ray_shards(datadir_kd, args.pseudo_ratio, args.pseudo_data_hold_ratio)
    # --N_rand is the GLOBAL batch in shard files per step, as in the reference (its DataLoader builds one batch that
    # nn.DataParallel then splits over the GPUs, main.py:794-806,1374): each rank loads N_rand / world shards, so the
    # README command keeps its optimisation schedule (lrate, N_iters, hard-ray pool size) at any GPU count
    # (not divisible by the rank count — the README's --N_rand 20 on 8 GPUs —: the first N_rand % world ranks take one shard
    # more and every rank's gradient is weighted by its share of the step's rays, below; nn.DataParallel scattered uneven
    # batches the same way, main.py:1374)
    try:
        shards = split_shards(args.N_rand, world)
    except ValueError as e:
        raise SystemExit(str(e))
    loader = D.RayShardLoader(files, shards[rank], rank=rank, world=world, device=device,
                              threads=max(1, min(args.num_workers, 16)))
    uneven = len(set(shards)) > 1
    if uneven:
        logger.info("--N_rand %d over %d ranks: %s shard files per rank and step; gradients weighted by ray share" %
                    (args.N_rand, world, shards))
    logger.info("Loaded data. Now total #train files: %d (this rank: %d)" % (len(files), len(loader.files)))
    trainer = R2LTrainer(model, point_sampler, lw_rgb=args.lw_rgb)
    if ckpt is not None and args.resume:
        trainer.load_optimizer_state_dict(ckpt["optimizer_state_dict"])
        logger.info("Resume optimizer successfully.")
    pool = HardRayPool(args.hard_ratio, args.hard_mul, seed=1000 + rank) if args.hard_ratio else None
    hist_psnr, t_data, t_batch = 0., 0., 0.
    logger.info("Begin training")
    for i in range(start + 1, args.N_iters + 1):
        t0 = time.time()
        lr = lr_schedule(i, args.lrate, args.lrate_decay, args.warmup_lr)
        batch = loader.next()  # device tensor; its H2D copy (36 B/ray) ran on the loader's stream during the last step
        rays_o, rays_d, target = batch[:, :3], batch[:, 3:6], batch[:, 6:9]
        batch_size = rays_o.shape[0]
        if pool is not None:
            rays_o, rays_d, target = pool.augment(rays_o, rays_d, target)
        t_data = time.time() - t0
        n_global = None
        if uneven:  # this step's rays over all ranks: shard rows + what each rank's hard-ray pool appends (deterministic)
            rows = [q * loader.rows_per_file for q in shards]
            n_global = [r + (pool.extra_rays(r, i - start - 1) if pool is not None else 0) for r in rows]
            # the ray count PREDICTED for this rank is the weight its gradient gets: it must be the batch it really holds (a
            # shard with another row count, a pool that fills differently: silently wrong weights, and ranks that disagree
            # on the collective sequence — ADVICE r3)
            if rays_o.shape[0] != n_global[rank]:
                raise RuntimeError("uneven --N_rand: rank %d holds %d rays in iteration %d, the share computed on every rank "
                                   "says %d (shards with differing row counts?)" % (rank, rays_o.shape[0], i, n_global[rank]))
        rgb, loss_out = trainer.step(rays_o, rays_d, target, lr, perturb=args.perturb, n_global=n_global)
        if pool is not None:
            pool.update(rgb, rays_o, rays_d, target, batch_size)
        t_batch = time.time() - t0
        if i % args.i_print == 0:
            loss, psnr = loss_out.tolist()  # the only host sync of the loop, every i_print iterations
            hist_psnr = psnr if hist_psnr == 0. else hist_psnr * 0.95 + psnr * 0.05
            logger.info("[TRAIN] Iter %d data_time %.4f batch_time %.4f loss %.6f psnr %.4f hist_psnr %.4f LR %.10f" %
                        (i, t_data, t_batch, loss, psnr, hist_psnr, lr))
            # where the fp16 kernels stand with this model (include/r2l_hip.h range control): largest |activation| / |chain
            # gradient| of the last step, the power-of-two scales the streams run on, head-room to the guard, fallbacks so far
            ri = trainer.range_info()
            logger.info("[RANGE] Iter %d act_amax %.4g act_scale %g headroom x%.3g | grad_amax %.4g grad_scale %g headroom x%.3g | "
                        "fallbacks fwd %d bwd %d" % (i, ri["amax"], ri["scale"], ri["headroom"], ri.get("grad_amax", 0.),
                                                     ri.get("grad_scale", 1.), ri.get("grad_headroom", float("inf")),
                                                     ri["trips"], ri.get("bwd_trips", 0)))
        if i % args.i_testset == 0:
            savedir = os.path.join(logger.gen_img_path, "testset_%s_iter%d" % (logger.ExpID, i))
            os.makedirs(savedir, exist_ok=True)
            t_ = time.time()
            _, misc = render_path(test_poses, model, point_sampler, device, logger, gt_imgs=test_images,
                                  savedir=savedir, rank=rank, world=world)
            if misc["test_psnr_v2"] > best_psnr:
                best_psnr, best_psnr_step = misc["test_psnr_v2"].item(), i
                if rank == 0:
                    save_ckpt(os.path.join(logger.weights_path, "ckpt_best.tar"), i, model,
                              trainer.optimizer_state_dict(lr), best_psnr, best_psnr_step, r2l_config=r2l_config)
            logger.info("[TEST] Iter %d TestPSNR %.4f TestPSNRv2 %.4f TestSSIM %.4f BestPSNRv2 %.4f (Iter %d) "
                        "TrainHistPSNR %.4f LR %.8f Time %.1fs" %
                        (i, misc["test_psnr"].item(), misc["test_psnr_v2"].item(), misc["test_ssim"].item(), best_psnr,
                         best_psnr_step, hist_psnr, lr, time.time() - t_))
        if i % args.i_video == 0:
            # test: using novel poses (main.py:1473-1484)
            logger.info("Iter %d Rendering video... (n_pose: %d)" % (i, len(video_poses)))
            t_ = time.time()
            rgbs, _ = render_path(video_poses, model, point_sampler, device, logger, rank=rank, world=world)
            path = save_video(rgbs, logger, logger.ExpID, i, args.video_tag, rank, world, device)
            if rank == 0:
                logger.info('Iter %d Save video: "%s" (time: %.2fs)' % (i, path, time.time() - t_))
        if i % args.i_weights == 0 and rank == 0:
            name = "ckpt_%d.tar" % i if args.save_intermediate_models else "ckpt.tar"
            path = save_ckpt(os.path.join(logger.weights_path, name), i, model, trainer.optimizer_state_dict(lr),
                             best_psnr, best_psnr_step, r2l_config=r2l_config)
            logger.info('Iter %d Save checkpoint: "%s".' % (i, path))
    loader.close()
    if world > 1 and os.environ.get("R2L_CHECK_SYNC"):  # tests: the replicas must have stayed bit-identical
        from .dist_utils import parameters_in_sync
        ok = parameters_in_sync(trainer.eng.flat)
        logger.info("replicas in sync after %d iterations: %s (skipped steps: %d)" % (args.N_iters, ok, trainer.drain()))
        if not ok:
            raise RuntimeError("data-parallel replicas diverged")
    return {"trainer": trainer, "logger": logger, "model": model, "r2l_config": r2l_config}
