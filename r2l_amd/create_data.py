"""Teacher pseudo-data generation — the `python utils/create_data.py --create_data rand ...` surface
(/root/reference/utils/create_data.py:606-872, 'rand' branch 777-872): a pretrained NeRF (coarse + fine) renders
random poses; the rays [o, d, rgb] are shuffled and written as data_<k>.npy shards of 4096 rays.

Per pose the whole stack runs on the GPU through libr2l_hip.so: get_rays -> stratified z -> fused embed+MLP (coarse)
-> raw2outputs -> sample_pdf + sort -> fused embed+MLP (fine) -> raw2outputs.  Poses are partitioned over ranks
(i % world == rank) with rank-disjoint shard indices and NO collective; each rank uses its own
RandomState, so the reference's single global np.random replay is distributional, not bitwise.
"""
import os
import re
import queue
import shutil
import threading
import time

import numpy as np
import torch

from . import data as D
from .checkpoint import load_weights
from .driver import apply_arithmetic, init_distributed, sync
from .logger import Logger
from .nerf_raybased import NeRF
from .options import parse_args, validate_accelerated
from .render import get_embedder, get_rays, render, run_network


def create_teacher(args, device):
    """Coarse + fine NeRF(D=8, W=256, 63+27) with frozen weights from --teacher_ckpt (create_data.py:250-280)."""
    nets = []
    for key in ("network_fn_state_dict", "network_fine_state_dict"):
        net = NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27,
                   use_viewdirs=args.use_viewdirs).to(device)
        net.eval()
        for p in net.parameters():
            p.requires_grad = False
        if args.teacher_ckpt:
            load_weights(net, args.teacher_ckpt, key)
        nets.append(net)
    return nets


def teacher_render_kwargs(args, coarse, fine):
    """render_kwargs_train of create_data.py:302-318 with the teacher pair plugged in (create_data.py:802-808)."""
    embed_fn, _ = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, _ = get_embedder(args.multires_views, args.i_embed)
    qfn = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                   netchunk=args.netchunk)
    return dict(network_query_fn=qfn, perturb=args.perturb, N_importance=args.N_importance, network_fine=fine,
                N_samples=args.N_samples, network_fn=coarse, use_viewdirs=args.use_viewdirs, white_bkgd=args.white_bkgd,
                raw_noise_std=args.raw_noise_std, ndc=False, lindisp=args.lindisp)


def render_pose_rows(pose, H, W, focal, near, far, chunk, render_kwargs):
    """[H*W, 9] rows [o, d, rgb] of one teacher-rendered pose (create_data.py:819-840)."""
    rays_o, rays_d = get_rays(H, W, focal, pose[:3, :4])
    with torch.no_grad():
        rgb, *_ = render(H, W, focal, chunk=chunk, rays=torch.stack([rays_o, rays_d], 0), near=near, far=far,
                         **render_kwargs)
    return torch.cat([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), rgb.reshape(-1, 3)], dim=-1)


def shard_index_base(rank, world, n_pose, chunk_poses, files_per_flush, n_existing=0):
    """First data_<k>.npy index of a rank.  Every rank gets the same-sized index range, wide enough for the LARGEST
    per-rank flush count (ranks own i % world == rank of 1..n_pose, so their pose counts differ by one and with them,
    sometimes, their flush counts: n_pose=301, world=3, chunk=100 -> 1, 2, 1 flushes), and numbering continues after the
    files a kept directory already holds, as the reference's does (create_data.py:789-792: `split = len(npys)`)."""
    chunk_poses = max(int(chunk_poses), 1)
    most_poses = (n_pose + world - 1) // world
    most_flushes = (most_poses + chunk_poses - 1) // chunk_poses
    return n_existing + rank * most_flushes * files_per_flush


def next_free_shard_index(names):
    """First unused data_<k>.npy index of a kept directory: one past the LARGEST index present.  (The reference counts the
    files, create_data.py:789-792, which is the same thing for its gap-free single-process numbering; a multi-rank run here
    leaves gaps between the rank ranges — shard_index_base sizes them for the largest flush count — so a count would make
    the next run overwrite existing shards.)"""
    used = [int(m.group(1)) for m in (re.fullmatch(r"data_(\d+)\.npy", x) for x in names) if m]
    return max(used) + 1 if used else 0


def main(argv=None):
    args = parse_args(argv)
    validate_accelerated(args)
    if args.create_data != "rand":
        raise NotImplementedError("only --create_data rand (the README pipeline) is on the accelerated path")
    rank, world, device = init_distributed()
    logger = Logger(args, rank)
    rng = np.random.RandomState(1000003 * rank)  # per-rank pose / focal / shuffle stream

    scene = None
    if os.path.exists(os.path.join(args.datadir, "transforms_train.json")):
        scene = D.load_blender_data(args.datadir, args.half_res, args.testskip)
        hwf = scene[3]
        H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    else:  # intrinsics of the 400x400 lego setting when no scene directory is present (main.py:927 comment)
        H, W, focal = 400, 400, 555.5555155968841
    near, far = 2., 6.
    coarse, fine = create_teacher(args, device)
    apply_arithmetic(args, device, logger, teachers=(coarse, fine))  # --r2l_precision: the teacher kernels' arithmetic
    kwargs = teacher_render_kwargs(args, coarse, fine)
    if args.test_teacher:
        # "Testing teacher..." (create_data.py:723-741): the test views through render_path with render_kwargs_test (perturb =
        # --perturb_test, raw_noise_std = 0), Loss / PSNR logged before any data is generated; frames sharded over the ranks
        from .driver import render_path
        if not args.teacher_ckpt or scene is None:
            raise SystemExit("--test_teacher needs --teacher_ckpt and a scene directory (--datadir) with test views")
        images, poses, _, _, i_split = scene
        images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:]) if args.white_bkgd else images[..., :3]
        kw_test = dict(kwargs, perturb=args.perturb_test, raw_noise_std=0., near=near, far=far)
        _, misc = render_path(poses[i_split[2]], coarse, None, device, logger, gt_imgs=images[i_split[2]], rank=rank, world=world,
                              teacher=dict(hwf=(H, W, focal), chunk=args.chunk, render_kwargs=kw_test, render_factor=args.render_factor))
        logger.info("Teacher test: Loss %.4f PSNR %.4f" % (misc["test_loss"].item(), misc["test_psnr"].item()))

    datadir_new = args.datadir_kd.split(":")[-1]
    if rank == 0:
        if os.path.exists(datadir_new) and args.rm_existing_data:
            shutil.rmtree(datadir_new)
        os.makedirs(datadir_new, exist_ok=True)
    if world > 1:
        torch.distributed.barrier()
    n_existing = next_free_shard_index(os.listdir(datadir_new))  # kept directory: numbering continues behind it
    if world > 1:
        torch.distributed.barrier()  # every rank has counted before any rank writes
    n_pose = args.n_pose_kd if isinstance(args.n_pose_kd, int) else int(args.n_pose_kd[0])
    mine = [i for i in range(1, n_pose + 1) if i % world == rank]
    rays_per_file = 4096
    # rank-disjoint shard index ranges: each flush of `create_data_chunk` poses yields at most this many files
    files_per_flush = (args.create_data_chunk * H * W) // rays_per_file
    next_index = shard_index_base(rank, world, n_pose, args.create_data_chunk, files_per_flush,
                                  n_existing)
    # Rows of a flush are copied pose by pose (non-blocking) into one of two pinned staging buffers; a writer thread
    # shuffles and saves a full buffer while the GPU renders into the other one.  The only host syncs are one per flush.
    chunk_poses = max(args.create_data_chunk, 1)
    pin = device.type == "cuda"
    stage = [torch.empty(chunk_poses * H * W, 9, dtype=torch.float32, pin_memory=pin) for _ in range(2)]
    jobs, errors = queue.Queue(maxsize=1), []
    free = [threading.Event(), threading.Event()]  # set while no flush job is reading that staging buffer
    for ev in free:
        ev.set()

    def writer():
        while True:
            job = jobs.get()
            if job is None:
                return
            try:
                rows, seed, first_index, slot = job
                r = np.random.RandomState(seed)
                p1, p2 = r.permutation(rows.shape[0]), r.permutation(rows.shape[0])
                # the reference permutes twice (create_data.py:854-859): rows[p1][p2] == rows[p1[p2]]
                D.write_ray_shards(rows[p1[p2]], datadir_new, first_index, rays_per_file)
            except Exception as e:  # surfaced by the main thread
                errors.append(e)
            finally:
                free[job[3]].set()

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    t0, n_rays, k, filled = time.time(), 0, 0, 0
    for j, i in enumerate(mine, 1):
        pose = D.get_rand_pose(rng).to(device)
        focal_ = focal * (1 + rng.rand()) if args.use_rand_focal else focal  # focal x U[1,2) (create_data.py:816)
        # whole frame per launch on the GPU: --chunk (32 768 rays in the reference's configs) is a memory work-around of
        # the op-by-op path; the fused kernels need 0.5 GB of scratch for a 400x400 frame at 192 samples
        chunk = max(args.chunk, H * W) if device.type == "cuda" else args.chunk
        rows = render_pose_rows(pose, H, W, focal_, near, far, chunk, kwargs)
        if filled == 0:
            free[k].wait()  # the flush that last used this buffer has been written
        stage[k][filled:filled + H * W].copy_(rows, non_blocking=True)
        filled += H * W
        n_rays += H * W
        if j % chunk_poses == 0 or j == len(mine):
            sync(device)  # the staged rows have landed
            if errors:
                raise errors[0]
            free[k].clear()
            jobs.put((stage[k][:filled].numpy(), int(rng.randint(0, 2**31 - 1)), next_index, k))
            next_index += filled // rays_per_file
            k, filled = 1 - k, 0
            dt = time.time() - t0
            logger.info("[%d/%d poses on rank %d] %d rays in %.1fs = %.0f rays/s; shards up to data_%d.npy" %
                        (j, len(mine), rank, n_rays, dt, n_rays / dt, next_index - 1))
    jobs.put(None)
    th.join()
    if errors:
        raise errors[0]
    return {"n_rays": n_rays, "datadir": datadir_new, "logger": logger}
