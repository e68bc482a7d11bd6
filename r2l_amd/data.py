"""Data layer on the R2L path: Blender-format scenes, pose generators and the [4096,9] `.npy` ray-shard format
(mirror of /root/reference/dataset/load_blender.py:22-28, 31-120, 257-368; PIL replaces imageio/cv2).

Ray shards: NumPy v1 `.npy`, float32 C-order [n_ray, 9] rows [o(3), d(3), rgb(3)], named data_<k>.npy (teacher
pseudo data, utils/create_data.py:854-872) or train_<k>.npy (real images).  RayShardLoader streams them through the
native reader threads of libr2l_hip.so into pinned memory, rank-sharded (files[rank::world]) — the per-process
replacement of the reference's DataLoader(BlenderDataset_v2, batch_size=N_rand, InfiniteSampler) (main.py:759-808)."""
import json
import os

import numpy as np
import torch

# ---- poses -------------------------------------------------------------------------------------------------------


def _trans_t(t):
    return torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=torch.float32)


def _rot_phi(phi):
    c, s = np.cos(phi), np.sin(phi)
    return torch.tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=torch.float32)


def _rot_theta(th):
    c, s = np.cos(th), np.sin(th)
    return torch.tensor([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=torch.float32)


_FLIP = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)


def pose_spherical(theta, phi, radius):
    """Camera-to-world [4,4] on a sphere: angles in degrees (load_blender.py:22-28)."""
    return _FLIP @ (_rot_theta(theta / 180. * np.pi) @ (_rot_phi(phi / 180. * np.pi) @ _trans_t(radius)))


def get_rand_pose(rng=np.random):
    """theta ~ U[-180,180), phi ~ U[-90,0), radius 4 (load_blender.py:359-368).  rng: np.random or a RandomState."""
    theta = -180 + rng.rand() * 360
    phi = -90 + rng.rand() * 90
    return pose_spherical(theta, phi, 4)


def get_novel_poses(args, n_pose, theta1=-180, theta2=180, phi1=-90, phi2=0):
    """Evenly spaced video poses (load_blender.py:327-356): int -> thetas at phi=-30, r=4; list -> grid spec."""
    near, far = 2, 6
    if isinstance(n_pose, int):
        thetas, phis, radii = np.linspace(theta1, theta2, n_pose + 1)[:-1], [-30], [4]
    elif ":" not in n_pose[0]:
        n = [int(x) for x in n_pose]
        thetas = np.linspace(theta1, theta2, n[0] + 1)[:-1]
        phis = np.linspace(phi1, phi2, n[1] + 2)[1:-1]
        radii = np.linspace(near, far, n[2] + 2)[1:-1]
    else:
        def axis(spec, lo, hi, closed):
            mode, value = spec.split(":")
            if mode != "sample":
                return [float(value)]
            return np.linspace(lo, hi, int(value) + 1)[:-1] if closed else np.linspace(lo, hi, int(value) + 2)[1:-1]
        thetas, phis, radii = axis(n_pose[0], theta1, theta2, True), axis(n_pose[1], phi1, phi2, False), axis(
            n_pose[2], near, far, False)
    return torch.stack([pose_spherical(t, p, r) for r in radii for p in phis for t in thetas], 0)


# ---- Blender scenes ------------------------------------------------------------------------------------------------


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path))


def load_blender_data(basedir, half_res=False, testskip=1, n_pose=40):
    """images[N,H,W,C] in [0,1], poses[N,4,4], render_poses[n_pose,4,4], [H,W,focal], i_split  (load_blender.py:31-120).
    half_res halves H and W by 2x2 area averaging (what cv2.INTER_AREA computes for an exact factor of 2)."""
    metas = {}
    for s in ("train", "val", "test"):
        with open(os.path.join(basedir, "transforms_%s.json" % s)) as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    for s in ("train", "val", "test"):
        skip = 1 if (s == "train" or testskip == 0) else testskip
        frames = metas[s]["frames"][::skip]
        imgs = np.stack([_read_png(os.path.join(basedir, f["file_path"] + ".png")) for f in frames])
        imgs = (imgs / 255.).astype(np.float32)
        poses = np.array([f["transform_matrix"] for f in frames]).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    meta = metas["test"]
    if "camera_angle_x" in meta:
        camera_angle_x = float(meta["camera_angle_x"])
    else:
        with open(os.path.join(basedir, "dataset_info.json")) as fp:
            camera_angle_x = float(json.load(fp)["camera_angle_x"])
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    render_poses = torch.stack([pose_spherical(t, -30., 4.) for t in np.linspace(-180, 180, n_pose + 1)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = imgs[:, :2 * H, :2 * W].reshape(imgs.shape[0], H, 2, W, 2, imgs.shape[-1]).mean(axis=(2, 4))
    return torch.from_numpy(np.ascontiguousarray(imgs, dtype=np.float32)), torch.from_numpy(poses), render_poses, [
        H, W, focal], i_split


# ---- ray shards -------------------------------------------------------------------------------------------------------


def list_ray_shards(datadir, pseudo_ratio=-1., hold_ratio=0., rng=np.random):
    """Shard file list with the reference's selection rules (BlenderDataset_v2.__init__, load_blender.py:271-296)."""
    names = sorted(x for x in os.listdir(datadir) if x.endswith(".npy"))
    pseudo = [os.path.join(datadir, x) for x in names if not x.startswith("train_")]
    original = [os.path.join(datadir, x) for x in names if x.startswith("train_")]
    assert 0 <= pseudo_ratio <= 1 or pseudo_ratio == -1
    if pseudo_ratio != -1:
        n_pseudo = int(len(original) / (1. - pseudo_ratio)) - len(original)
        pseudo = rng.choice(pseudo, n_pseudo).tolist()
    files = pseudo + original
    assert 0 <= hold_ratio < 1
    if hold_ratio > 0:
        files = list(rng.choice(files, int(len(files) * (1 - hold_ratio))))
    return files


class BlenderDataset_v2(torch.utils.data.Dataset):
    """Index -> (rays_o, rays_d, rgb) of one `.npy` shard  (load_blender.py:257-324)."""

    def __init__(self, datadir, dim_dir=3, dim_rgb=3, hold_ratio=0, pseudo_ratio=1.):
        self.all_splits = list_ray_shards(datadir, pseudo_ratio, hold_ratio)
        self.dim_dir, self.dim_rgb = dim_dir, dim_rgb

    def __getitem__(self, index):
        d = torch.from_numpy(np.load(self.all_splits[index]).astype(np.float32, copy=False))
        return d[..., :3], d[..., 3:3 + self.dim_dir], d[..., 3 + self.dim_dir:3 + self.dim_dir + self.dim_rgb]

    def __len__(self):
        return len(self.all_splits)


def shard_for_rank(files, rank, world):
    """Disjoint, near-equal file subsets per rank (files[rank::world])."""
    return files[rank::world]


class RayShardLoader:
    """Infinite stream of batches of `n_files` shards as one [n_files*rays_per_file, 9] tensor, drawn by random
    permutations over this rank's files (the InfiniteSampler of main.py:759-767).

    The reading is done by the native reader of libr2l_hip.so (csrc/r2l_shard_reader.hip): `threads` host threads pread()
    shard payloads straight into a ring of `prefetch` pinned buffers owned by this object.  With `device` set, next()
    returns a DEVICE tensor: the host->device copy of batch i+1 runs on a side stream while step i computes (two
    device buffers), so neither the file reads nor the 36 B/ray PCIe copy sit on the training stream.  Without a
    device, next() returns the pinned host tensor; either way the tensor is valid until the second-next call."""

    def __init__(self, files, n_files, rank=0, world=1, seed=0, prefetch=3, pin=None, device=None, threads=4):
        import ctypes
        from . import _lib
        self.files = shard_for_rank(list(files), rank, world)
        if not self.files:
            raise ValueError("rank %d got no ray shards (have %d files, world %d)" % (rank, len(files), world))
        self.n_files = n_files
        self._L = _lib.load()
        rows, cols = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.r2l_npy_shape(self.files[0].encode(), ctypes.byref(rows), ctypes.byref(cols)),
                   "r2l_npy_shape")
        self.rows_per_file, self.cols = rows.value, cols.value
        depth = max(3, int(prefetch))
        self.host = torch.empty(depth, n_files * rows.value, cols.value, dtype=torch.float32)
        if torch.cuda.is_available() if pin is None else pin:
            self.host = self.host.pin_memory()
        self._paths = (ctypes.c_char_p * len(self.files))(*[f.encode() for f in self.files])
        slots = (ctypes.c_void_p * depth)(*[self.host[i].data_ptr() for i in range(depth)])
        self._h = ctypes.c_void_p()
        _lib.check(self._L.r2l_reader_open(ctypes.cast(self._paths, ctypes.c_void_p), len(self.files), n_files,
                                           max(1, int(threads)), seed + 9973 * rank, ctypes.cast(slots, ctypes.c_void_p),
                                           depth, ctypes.byref(self._h)), "r2l_reader_open")
        self._held = []  # [(slot, copy-done event | None)] checked out of the ring, oldest first
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type != "cuda":
            self.device = None
        if self.device is not None:
            self._copy_stream = torch.cuda.Stream(self.device)
            self._dev = [torch.empty(n_files * rows.value, cols.value, device=self.device) for _ in range(2)]
            self._ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._k = 0
            self._issue_copy()

    def _take_slot(self):
        import ctypes
        from . import _lib
        # hand finished slots back to the reader: all but the newest on the host path, copies that completed on the
        # device path (blocking on the oldest only if the ring would otherwise run dry)
        while self._held:
            slot, ev = self._held[0]
            if ev is None:
                if len(self._held) < 2:
                    break
            elif not ev.query():
                if len(self._held) < self.host.shape[0] - 1:
                    break
                ev.synchronize()
            self._held.pop(0)
            _lib.check(self._L.r2l_reader_release(self._h, slot), "r2l_reader_release")
        slot = ctypes.c_int()
        _lib.check(self._L.r2l_reader_next(self._h, ctypes.byref(slot)), "r2l_reader_next")
        return slot.value

    def _issue_copy(self):
        k = self._k
        slot = self._take_slot()
        cur = torch.cuda.current_stream(self.device)
        self._copy_stream.wait_stream(cur)  # the step that read _dev[k] (two calls ago) is enqueued before this point
        with torch.cuda.stream(self._copy_stream):
            self._dev[k].copy_(self.host[slot], non_blocking=True)
            self._ready[k].record(self._copy_stream)
        self._held.append((slot, self._ready[k]))

    def next(self):
        if self._h is None:
            raise RuntimeError("RayShardLoader is closed")
        if self.device is None:
            slot = self._take_slot()
            self._held.append((slot, None))
            return self.host[slot]
        k = self._k
        torch.cuda.current_stream(self.device).wait_event(self._ready[k])
        out = self._dev[k]
        self._k ^= 1
        self._ready[self._k] = torch.cuda.Event()
        self._issue_copy()
        return out

    __next__ = next

    def __iter__(self):
        return self

    def files_read(self):
        import ctypes
        n = ctypes.c_int64()
        self._L.r2l_reader_info(self._h, None, None, ctypes.byref(n))
        return n.value

    def close(self):
        if getattr(self, "_h", None) is not None:
            if self.device is not None:
                self._copy_stream.synchronize()
            self._L.r2l_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ray_shards(rows, outdir, start_index, rays_per_file=4096, prefix="data_"):
    """Save [n,9] rows as <prefix><k>.npy files of `rays_per_file` rays; leftover rows are dropped as in
    create_data.py:862-872.  Returns the next free index."""
    rows = np.asarray(rows, dtype=np.float32)
    n_files = rows.shape[0] // rays_per_file
    for k in range(n_files):
        np.save(os.path.join(outdir, "%s%d.npy" % (prefix, start_index + k)),
                rows[k * rays_per_file:(k + 1) * rays_per_file])
    return start_index + n_files


def convert_images_to_ray_shards(datadir, splits=("train",), suffix="", ignore=(), full_res=False, white_bkgd=True,
                                 rays_per_file=4096, rng=np.random):
    """Real images -> shuffled [4096,9] ray shards `<splits>_<k>.npy` in `<datadir>_real_<splits><suffix>/` for the
    fine-tuning stage (README step 4; reference utils/convert_original_data_to_rays_blender.py:116-235, Blender
    branch; the DONERF ray convention is out of scope).  Returns (savedir, n_files)."""
    from .render import get_rays
    prefix = "".join(splits)
    savedir = "%s_real_%s%s" % (os.path.normpath(datadir), prefix, suffix)
    os.makedirs(savedir, exist_ok=True)
    ignore = set(str(i) for i in ignore)
    imgs, poses, meta = [], [], None
    for s in splits:
        with open(os.path.join(datadir, "transforms_%s.json" % s)) as fp:
            meta = json.load(fp)
        for frame in meta["frames"]:
            if frame["file_path"].split("_")[-1] in ignore:  # e.g. "./train/r_3" -> "3"
                continue
            imgs.append(_read_png(os.path.join(datadir, frame["file_path"] + ".png")))
            poses.append(np.array(frame["transform_matrix"], dtype=np.float32))
    imgs = (np.stack(imgs) / 255.).astype(np.float32)
    H, W = imgs.shape[1:3]
    if "camera_angle_x" in meta:
        camera_angle_x = float(meta["camera_angle_x"])
    else:
        with open(os.path.join(datadir, "dataset_info.json")) as fp:
            camera_angle_x = float(json.load(fp)["camera_angle_x"])
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    if not full_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        imgs = imgs[:, :2 * H, :2 * W].reshape(imgs.shape[0], H, 2, W, 2, imgs.shape[-1]).mean(axis=(2, 4))
    if imgs.shape[-1] == 4 and white_bkgd:
        imgs = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    imgs = imgs[..., :3]
    rows = []
    for im, po in zip(imgs, poses):
        ro, rd = get_rays(H, W, focal, torch.from_numpy(po[:3, :4]))
        rows.append(np.concatenate([ro.reshape(-1, 3).numpy(), rd.reshape(-1, 3).numpy(), im.reshape(-1, 3)], -1))
    rows = np.concatenate(rows, 0).astype(np.float32)
    rows = rows[rng.permutation(rows.shape[0])][rng.permutation(rows.shape[0])]  # shuffled twice, as the reference
    n_files = rows.shape[0] // rays_per_file
    for k in range(n_files):
        np.save(os.path.join(savedir, "%s_%d.npy" % (prefix, k + 1)), rows[k * rays_per_file:(k + 1) * rays_per_file])
    return savedir, n_files
