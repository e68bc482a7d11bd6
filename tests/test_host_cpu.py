"""Host logic on CPU: options/config parsing, checkpoint un-pickling of a REFERENCE-written .tar, data layer,
LR schedule, C-ABI library exports."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from r2l_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "r2l_hip.h")).read()
    declared = set(re.findall(r"\b(r2l_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.r2l_param_count(43) == 5917187
    assert lib.r2l_teacher_param_count() == sum(v.numel() for v in O.make_teacher_state_dicts(0, 1)[0].values())
    assert lib.r2l_num_tiles(33) == 2


def test_explicit_config_dispatch(monkeypatch):
    """r2l_config (include/r2l_hip.h): the *_cfg queries answer for the config they are given — NULL / all-zero = the plain
    forms (environment switches apply), a non-zero field wins over the environment — and leave no state behind."""
    import ctypes
    from r2l_amd import _lib
    lib = _lib.load()
    for k in ("R2L_FORCE_VARIANT", "R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2", "R2L_COOPF_TILES"):
        monkeypatch.delenv(k, raising=False)
    ref = ctypes.byref
    auto = _lib.make_config()
    assert ctypes.sizeof(_lib.Config) == 32
    for n in (32, 4096, 20000, 98304, 160000):
        assert lib.r2l_variant_for_cfg(n, None) == lib.r2l_variant_for_cfg(n, ref(auto)) == lib.r2l_variant_for(n)
        assert lib.r2l_forward_layout_for_cfg(n, 1, ref(auto)) == lib.r2l_forward_layout_for(n, 1)
        assert lib.r2l_backward_layout_for_cfg(n, None) == lib.r2l_backward_layout_for(n)
        assert lib.r2l_coop_tiles_for_cfg(n, 43, ref(auto)) == lib.r2l_coop_tiles_for(n, 43)
    bf = _lib.make_config(precision="bf16x3")
    f32 = _lib.make_config(precision="fp32_mfma")
    f16 = _lib.make_config(precision="fp16x2")
    assert lib.r2l_forward_layout_for_cfg(98304, 1, ref(bf)) == 3 and lib.r2l_backward_layout_for_cfg(98304, ref(bf)) == 3
    assert lib.r2l_forward_layout_for_cfg(160000, 0, ref(bf)) == 3
    assert lib.r2l_forward_layout_for_cfg(98304, 1, ref(f32)) == 32 and lib.r2l_backward_layout_for_cfg(98304, ref(f32)) == 32
    assert lib.r2l_variant_for_cfg(4096, ref(bf)) == 2 and lib.r2l_coop_tiles_for_cfg(4096, 43, ref(bf)) == 0
    # tiling and tiles per workgroup
    assert lib.r2l_variant_for_cfg(98304, ref(_lib.make_config(tiling="coop16"))) == 2
    assert lib.r2l_variant_for_cfg(4096, ref(_lib.make_config(tiling="main"))) == 0
    assert lib.r2l_coop_tiles_for_cfg(4096, 43, ref(_lib.make_config(tiling="main"))) == 0
    assert lib.r2l_coop_tiles_for_cfg(98304, 43, ref(_lib.make_config(tiling="coopf"))) == 2
    assert lib.r2l_coop_tiles_for_cfg(4096, 43, ref(_lib.make_config(coop_tiles=2))) == 2
    assert lib.r2l_coop_tiles_for_cfg(12288, 43, ref(_lib.make_config(coop_tiles=1))) == 1
    assert lib.r2l_coop_tiles_for_cfg(12288, 43, ref(_lib.make_config(coop_tiles=2))) == 2
    assert lib.r2l_coop_tiles_for_cfg(12288, 43, ref(_lib.make_config(coop_tiles=3))) == 3
    # explicit fields beat the environment; AUTO fields follow it; nothing sticks after the call
    monkeypatch.setenv("R2L_NO_FWD2", "1")
    monkeypatch.setenv("R2L_FORCE_VARIANT", "coop16")
    assert lib.r2l_forward_layout_for(98304, 1) == 16 and lib.r2l_variant_for(98304) == 2
    both = _lib.make_config(precision="fp16x2", tiling="main")
    assert lib.r2l_forward_layout_for_cfg(98304, 1, ref(both)) == 2 and lib.r2l_variant_for_cfg(98304, ref(both)) == 0
    assert lib.r2l_variant_for_cfg(98304, ref(f16)) == 2  # tiling AUTO: the environment's coop16
    assert lib.r2l_forward_layout_for(98304, 1) == 16 and lib.r2l_variant_for(98304) == 2
    # the 32-ray fp32-MFMA cooperative family (tiling value 2) was retired in round 5: the value stays reserved and is refused
    retired = _lib.make_config()
    retired.tiling = 2
    assert lib.r2l_variant_for_cfg(4096, ref(retired)) == -1
    assert b"retired" in lib.r2l_last_error()
    with pytest.raises(ValueError, match="retired"):  # ... and the Python binding no longer offers the name
        _lib.make_config(tiling="coop")


def test_invalid_config_is_rejected():
    """A field outside its enum / range, or a non-zero reserved word: the queries return -1, the launch entry points fail
    with hipErrorInvalidValue before touching anything (so this runs without a GPU), r2l_last_error names the field."""
    import ctypes
    from r2l_amd import _lib
    lib = _lib.load()
    bad = []
    for field, value in (("precision", 4), ("precision", -1), ("tiling", 5), ("coop_tiles", 4), ("reserve_cus", -2),
                         ("dw_mode", 3)):
        c = _lib.make_config()
        setattr(c, field, value)
        bad.append((field, c))
    c = _lib.make_config()
    c.reserved[1] = 7
    bad.append(("reserved", c))
    for field, c in bad:
        r = ctypes.byref(c)
        assert lib.r2l_variant_for_cfg(4096, r) == -1 and lib.r2l_forward_layout_for_cfg(4096, 1, r) == -1, field
        assert lib.r2l_backward_layout_for_cfg(4096, r) == -1 and lib.r2l_coop_tiles_for_cfg(4096, 43, r) == -1, field
        assert lib.r2l_chain_segments_ok_cfg(4096, 43, r) == -1, field
        rc = lib.r2l_forward_rays_cfg(None, None, None, None, None, None, 43, None, None, None, 4096, None, r)
        assert rc == 1 and field in lib.r2l_last_error().decode(), (field, rc, lib.r2l_last_error())  # hipErrorInvalidValue
        assert lib.r2l_forward_pose_cfg(None, 4, 4, 1., None, None, None, 43, None, None, r) == 1
        assert lib.r2l_forward_poses_cfg(None, 1, 4, 4, 1., None, None, None, 43, None, None, r) == 1
        assert lib.r2l_teacher_mlp_cfg(None, None, None, None, None, None, None, 4, 4, None, r) == 1
    ok = _lib.make_config(reserve_cus=-1, coop_tiles=2, dw_mode="exact")
    assert lib.r2l_variant_for_cfg(4096, ctypes.byref(ok)) >= 0


def test_bad_arguments_are_error_codes_not_device_faults():
    """NULL required pointers and out-of-range sizes are rejected with hipErrorInvalidValue before anything is launched (so
    this needs no GPU); empty inputs are successful no-ops."""
    from r2l_amd import _lib
    lib = _lib.load()
    INVALID = 1
    one = ctypes.c_void_p(64)  # any non-NULL value: the checks below fail before it is ever dereferenced
    assert lib.r2l_forward_rays(None, None, None, None, None, None, 43, None, None, None, 0, None) == 0  # N = 0
    assert lib.r2l_forward_rays(None, one, None, one, one, one, 43, one, None, None, 32, None) == INVALID
    assert b"r2l_forward_rays" in lib.r2l_last_error()
    assert lib.r2l_forward_rays(one, one, None, one, one, one, 43, one, one, None, 32, None) == INVALID  # save_x without save_t
    assert lib.r2l_forward_rays(one, one, None, one, one, one, -1, one, None, None, 32, None) == INVALID
    assert lib.r2l_forward_rays(one, one, None, one, one, one, 43, one, None, None, -5, None) == INVALID
    assert lib.r2l_forward_pose(one, 0, 400, 555., one, one, one, 43, one, None) == INVALID
    assert lib.r2l_forward_pose(None, 400, 400, 555., one, one, one, 43, one, None) == INVALID
    assert lib.r2l_forward_poses_cfg(one, 0, 400, 400, 555., one, one, one, 43, one, None, None) == 0  # K = 0
    assert lib.r2l_forward_poses_cfg(one, 2, 400, 400, 555., None, one, one, 43, one, None, None) == INVALID
    assert lib.r2l_forward_emb(None, one, one, 43, one, None, None, 32, None) == INVALID
    bf = _lib.make_config(precision="bf16x3")
    assert lib.r2l_forward_emb_cfg(None, one, one, 43, one, None, None, 32, one, None, ctypes.byref(bf)) == INVALID      # emb (bf16x3 path)
    assert lib.r2l_forward_emb_cfg(None, one, one, 43, one, None, None, 32, None, None, ctypes.byref(bf)) == INVALID     # ... -> r2l_forward_emb
    assert lib.r2l_forward_emb_cfg(one, one, one, 5000, one, None, None, 32, one, None, ctypes.byref(bf)) == INVALID     # n_block
    assert lib.r2l_forward_emb_cfg(None, None, None, 43, None, None, None, 0, None, None, ctypes.byref(bf)) == 0         # N = 0
    assert lib.r2l_pack_forward(None, 43, one, None) == INVALID and lib.r2l_pack_backward(one, 43, None, None) == INVALID
    assert lib.r2l_pack_forward_layout(one, 43, one, 5, None) == INVALID and lib.r2l_pack_backward_layout(one, 5000, one, 2, None) == INVALID
    args = [one] * 12 + [43, 1e-5] + [one] * 6 + [4096, None]
    assert lib.r2l_backward_part(*args, 0, 0, 86) == INVALID and lib.r2l_backward_part(*args, 64, 0, 86) == INVALID  # parts
    bad = list(args)
    bad[5] = None  # rgb
    assert lib.r2l_backward_part(*bad, 15, 0, 86) == INVALID
    bad = list(args)
    bad[4] = bad[0] = None  # neither emb nor rays
    assert lib.r2l_backward_part(*bad, 15, 0, 86) == INVALID
    bad = list(args)
    bad[20] = 0  # N = 0
    assert lib.r2l_backward_part(*bad, 15, 0, 86) == 0
    assert lib.r2l_adam_step(None, one, one, one, 10, 1e-3, .9, .999, 1e-8, 1, 1., None) == INVALID
    assert lib.r2l_adam_step(one, one, one, one, 10, 1e-3, .9, .999, 1e-8, 0, 1., None) == INVALID  # step counts from 1
    assert lib.r2l_adam_step(None, None, None, None, 0, 1e-3, .9, .999, 1e-8, 1, 1., None) == 0
    assert lib.r2l_adam_step_packed(one, one, one, one, 43, 1e-3, .9, .999, 1e-8, 1, 1., None, None, one, None) == INVALID  # wstream_fwd
    assert lib.r2l_adam_step_packed(one, one, one, one, 43, 1e-3, .9, .999, 1e-8, 0, 1., None, one, one, None) == INVALID   # step
    assert lib.r2l_adam_step_packed(one, one, one, one, 5000, 1e-3, .9, .999, 1e-8, 1, 1., None, one, one, None) == INVALID
    assert lib.r2l_loss_finish(one, 3, 1., None, None) == INVALID
    assert lib.r2l_pack_teacher(None, one, None) == INVALID
    assert lib.r2l_teacher_mlp(one, one, one, one, one, one, None, 4, 64, None) == INVALID
    assert lib.r2l_teacher_mlp(None, None, None, None, None, None, None, 0, 64, None) == 0
    assert lib.r2l_stratified_z(None, one, 1, one, None, one, 4, 64, None) == INVALID
    assert lib.r2l_raw2outputs(one, one, one, None, 1, one, one, one, None, None, 4, 64, None) == INVALID  # depth_map
    assert lib.r2l_raw2outputs(one, one, one, None, 1, one, one, one, None, one, 4, 300, None) == INVALID
    assert lib.r2l_sample_pdf_sort(one, one, None, 0, one, one, None, 4, 64, 128, None) == INVALID  # u
    assert lib.r2l_ssim(one, None, 8, 8, 3, None, one, one, None) == INVALID


def test_dispatch_and_buffer_size_helpers(monkeypatch):
    """Host-side decisions of the library (no device work): which kernel family / stream layout an N-ray launch takes under
    the environment switches, and the caller-side buffer sizes that go with them."""
    from r2l_amd import _lib
    lib = _lib.load()
    for k in ("R2L_FORCE_VARIANT", "R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2"):
        monkeypatch.delenv(k, raising=False)
    # defaults: the fp16 trio everywhere (layout 2 = fp16x2 stages with the bf16x3 stream behind them): its cooperative
    # kernels (one 32-ray tile per workgroup) up to 16 384 rays, one wave per tile above — both are the MAIN family (0)
    assert lib.r2l_variant_for(4096) == 0 and lib.r2l_variant_for(16385) == 0 and lib.r2l_variant_for(160000) == 0
    assert lib.r2l_forward_layout_for(4096, 1) == 2 and lib.r2l_backward_layout_for(4096) == 2
    assert lib.r2l_forward_layout_for(98304, 1) == 2 and lib.r2l_forward_layout_for(160000, 0) == 2
    assert lib.r2l_backward_layout_for(98304) == 2
    # ... cooperative: one tile per workgroup up to one tile per CU, two above, up to 16 384 rays; and again (two tiles) where
    # the one-wave-per-tile kernels would run a half-empty second round.  (The MIXED grid, 3, is opt-in: measured slower in
    # round 6, profiles/r06_mixed_coopf_ab.txt)
    assert [lib.r2l_coop_tiles_for(n, 43) for n in (32, 4096, 8192, 8193, 12288, 16352, 16353, 16384, 16385, 32768, 32769, 49152,
                                                     49153, 98304, 160000)] == [1, 1, 1, 2, 2, 2, 2, 2, 0, 0, 2, 2, 0, 0, 0]
    monkeypatch.setenv("R2L_COOPF_TILES", "3")  # mixed pinned: outside its band one tile below, two above
    assert [lib.r2l_coop_tiles_for(n, 43) for n in (4096, 8193, 12288, 16352, 16353, 16384)] == [1, 3, 3, 3, 2, 2]
    monkeypatch.setenv("R2L_COOPF_TILES", "2")
    assert lib.r2l_coop_tiles_for(4096, 43) == 2 and lib.r2l_coop_tiles_for(98304, 43) == 0
    monkeypatch.delenv("R2L_COOPF_TILES")
    monkeypatch.setenv("R2L_FORCE_VARIANT", "coopf")
    assert lib.r2l_variant_for(98304) == 0 and lib.r2l_forward_layout_for(98304, 1) == 2
    assert lib.r2l_coop_tiles_for(98304, 43) == 2
    monkeypatch.setenv("R2L_FORCE_VARIANT", "main")
    assert lib.r2l_coop_tiles_for(4096, 43) == 0
    monkeypatch.delenv("R2L_FORCE_VARIANT")
    monkeypatch.setenv("R2L_NO_DW2", "1")  # no fp16 trio, no cooperative fp16x2 kernels
    assert lib.r2l_coop_tiles_for(4096, 43) == 0
    monkeypatch.delenv("R2L_NO_DW2")
    # any switch of the fp16 training trio puts the WHOLE step on the bf16x3 trio (one stash format per step); forward-only
    # launches look at R2L_NO_FWD2 alone
    for k in ("R2L_NO_BWD2", "R2L_NO_DW2"):
        monkeypatch.setenv(k, "1")
        assert lib.r2l_forward_layout_for(98304, 1) == 3 and lib.r2l_backward_layout_for(98304) == 3
        assert lib.r2l_forward_layout_for(160000, 0) == 2
        monkeypatch.delenv(k)
    monkeypatch.setenv("R2L_NO_FWD2", "1")
    assert lib.r2l_forward_layout_for(98304, 1) == 3 and lib.r2l_backward_layout_for(98304) == 3
    assert lib.r2l_forward_layout_for(160000, 0) == 3
    # without the fp16 trio small launches go to the 16-ray cooperative fp32-MFMA kernels again
    assert lib.r2l_variant_for(4096) == 2 and lib.r2l_forward_layout_for(4096, 1) == 16 and lib.r2l_backward_layout_for(4096) == 16
    assert lib.r2l_variant_for(4097) == 0
    monkeypatch.setenv("R2L_NO_FWD3", "1")  # everything on the fp32 MFMA: the small-batch kernels win up to 20 480 rays again
    assert lib.r2l_forward_layout_for(98304, 1) == 32 and lib.r2l_backward_layout_for(98304) == 32
    assert lib.r2l_variant_for(20480) == 2 and lib.r2l_variant_for(24576) == 0
    monkeypatch.setenv("R2L_FORCE_VARIANT", "coop16")
    assert lib.r2l_variant_for(98304) == 2 and lib.r2l_forward_layout_for(98304, 1) == 16
    # buffer sizes: a stash slot holds 1 KiB + 32 B of mask words per (padded) ray; the streams hold every layout + status words
    assert lib.r2l_padded_rows(33) == 64 and lib.r2l_stash_slot_floats(33) == 64 * 264
    nb = 43
    stages_f, stages_b, pad = 64 + 34 * nb, 34 * nb, 8
    assert lib.r2l_fwd_stream_floats(nb) >= (stages_f + pad) * (24576 + 16384) // 4 + 16
    assert lib.r2l_bwd_stream_floats(nb) >= (stages_b + pad) * (24576 + 16384) // 4 + 16
    # body partials | head partials (64 slices) | tail partials (512 workgroups) | status words: disjoint regions (round 5)
    assert lib.r2l_dw_slab_floats() == 256 * 2 * (256 * 256 + 256) + 64 * 256 * 1024 + 512 * 4 * 256 + 16


def test_options_readme_command(tmp_path):
    from r2l_amd.options import parse_args
    cfg = os.path.join(ROOT, "configs", "lego_noview.txt")
    a = parse_args(["--model_name", "R2L", "--config", cfg, "--n_sample_per_ray", "16", "--netwidth", "256",
                    "--netdepth", "88", "--use_residual", "--trial.ON", "--trial.body_arch", "resmlp", "--N_rand", "20",
                    "--data_mode", "rays", "--hard_ratio", "0.2", "--hard_mul", "20", "--num_worker", "8",
                    "--warmup_lr", "0.0001,200", "--N_iters", "1200000", "--datadir_kd", "data/x:data/y"])
    assert a.netdepth == 88 and a.trial.body_arch == "resmlp" and a.trial.res_scale == 1.0 and a.trial.n_block == -1
    assert a.white_bkgd and a.half_res and not a.use_viewdirs and a.lrate_decay == 500  # from the config file
    assert a.N_rand == 20  # command line overrides the file's 1024
    assert a.hard_ratio == 0.2 and a.hard_mul == 20 and a.num_workers == 8 and a.n_pose_video == 40
    assert not hasattr(a, "trial.ON")
    b = parse_args(["--config", os.path.join(ROOT, "configs", "lego.txt")])
    assert b.use_viewdirs and b.N_importance == 128 and not hasattr(b, "trial")


def test_arithmetic_options_parse_and_resolve(tmp_path, monkeypatch):
    """--r2l_precision / --r2l_dw_mode (this build's own keys, also as config-file lines): parsed next to the reference's flags,
    and engine.arithmetic() resolves AUTO the way the library does (csrc/r2l_common.h r2l_use_fwd3 / r2l_use_fwd2 / r2l_dw_exact)."""
    from r2l_amd import _lib, engine
    from r2l_amd.options import parse_args
    a = parse_args(["--model_name", "R2L"])
    assert (a.r2l_precision, a.r2l_dw_mode) == ("auto", "auto")
    cfg = tmp_path / "c.txt"
    cfg.write_text("r2l_precision = bf16x3\nr2l_dw_mode = exact\n")
    a = parse_args(["--config", str(cfg)])
    assert (a.r2l_precision, a.r2l_dw_mode) == ("bf16x3", "exact")
    a = parse_args(["--config", str(cfg), "--r2l_precision", "fp32_mfma"])  # the command line wins
    assert a.r2l_precision == "fp32_mfma"
    with pytest.raises(SystemExit):
        parse_args(["--r2l_precision", "fp8"])
    for k in ("R2L_NO_FWD3", "R2L_NO_FWD2", "R2L_NO_BWD2", "R2L_NO_DW2", "R2L_DW_EXACT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(engine, "DEFAULT_CONFIG", {})
    assert engine.arithmetic() == {"precision": "fp16x2", "dw_mode": "fp16"}
    assert engine.arithmetic(_lib.make_config(dw_mode="exact")) == {"precision": "fp16x2", "dw_mode": "exact"}
    assert engine.arithmetic(_lib.make_config(precision="fp32_mfma")) == {"precision": "fp32_mfma", "dw_mode": "exact"}
    assert engine.arithmetic(_lib.make_config(precision="bf16x3", dw_mode="fp16")) == {"precision": "bf16x3", "dw_mode": "exact"}
    monkeypatch.setenv("R2L_NO_FWD3", "1")
    assert engine.arithmetic()["precision"] == "fp32_mfma"
    assert engine.arithmetic(_lib.make_config(precision="fp16x2"))["precision"] == "fp16x2"  # an explicit field beats the environment
    monkeypatch.setenv("R2L_NO_FWD3", "0")
    monkeypatch.setenv("R2L_DW_EXACT", "1")
    assert engine.arithmetic() == {"precision": "fp16x2", "dw_mode": "exact"}
    monkeypatch.setattr(engine, "DEFAULT_CONFIG", {"precision": "bf16x3"})
    assert engine.arithmetic()["precision"] == "bf16x3"


def test_unpickle_reference_checkpoint(golden_dir):
    """ckpt_w32d6.tar was written by the REFERENCE's save_ckpt (pickled reference NeRF_v3_2): it must load into our
    classes without running __init__, and the restored module must compute the reference's rgb."""
    from r2l_amd.checkpoint import load_ckpt, save_ckpt
    import model.nerf_raybased as mine
    ckpt = load_ckpt(os.path.join(golden_dir, "ckpt_w32d6.tar"), map_location="cpu")
    assert set(ckpt) >= {"global_step", "best_psnr", "network_fn_state_dict", "optimizer_state_dict", "network_fn"}
    net = ckpt["network_fn"]
    assert type(net) is mine.NeRF_v3_2 and type(net.body[0]) is mine.ResMLP
    g = np.load(os.path.join(golden_dir, "r2l_w32d6.npz"))
    g256 = np.load(os.path.join(golden_dir, "r2l_w256d88.npz"))
    emb = O.positional_embed(O.sample_train(torch.from_numpy(g256["rays_o"]), torch.from_numpy(g256["rays_d"]),
                                            O.z_vals(16, 2., 6.), 0.), 10)
    # the checkpoint holds the weights after 3 Adam steps
    sd3 = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p3/")}
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd3[k]), k
    with torch.no_grad():
        out = net(emb)
    np.testing.assert_allclose(out.numpy(), O.r2l_forward(sd3, emb).numpy(), atol=1e-6)
    assert ckpt["global_step"] == 3 and ckpt["optimizer_state_dict"]["state"][0]["exp_avg"].shape == (32, 1008)
    # and our writer produces a file the same loader (and hence the reference's) reads back identically
    path = save_ckpt(os.path.join(str(golden_dir), "..", "_tmp_ckpt.tar"), 7, net, ckpt["optimizer_state_dict"], 1., 2)
    again = load_ckpt(path, map_location="cpu")
    os.remove(path)
    assert again["global_step"] == 7 and type(again["network_fn"]) is mine.NeRF_v3_2
    assert all(torch.equal(again["network_fn_state_dict"][k], v) for k, v in net.state_dict().items())


def test_lr_schedule_matches_oracle():
    from r2l_amd.train_step import lr_schedule
    for step in (1, 50, 199, 200, 201, 5000, 600000):
        for w in ("", "0.0001,200"):
            assert lr_schedule(step, 5e-4, 500, w) == O.lr_schedule(step, 5e-4, 500, w)


def test_pose_spherical_and_shards(tmp_path, golden_dir):
    from r2l_amd import data
    g = np.load(os.path.join(golden_dir, "sampler.npz"))
    for (th, ph, r), c2w in zip(g["pose_spherical_args"], g["poses"]):
        np.testing.assert_allclose(data.pose_spherical(th, ph, r)[:3, :4].numpy(), c2w, atol=1e-6)
    rows = np.random.RandomState(0).rand(3 * 4096 + 100, 9).astype(np.float32)
    nxt = data.write_ray_shards(rows, str(tmp_path), 5)
    assert nxt == 8 and sorted(os.listdir(tmp_path)) == ["data_5.npy", "data_6.npy", "data_7.npy"]
    assert os.path.getsize(tmp_path / "data_5.npy") == 147584  # NumPy v1 header + 4096*9*4
    files = data.list_ray_shards(str(tmp_path))
    ds = data.BlenderDataset_v2(str(tmp_path), pseudo_ratio=-1)
    o, d, c = ds[0]
    assert o.shape == (4096, 3) and torch.equal(torch.cat([o, d, c], -1), torch.from_numpy(rows[:4096]))
    assert data.shard_for_rank(files, 0, 2) + data.shard_for_rank(files, 1, 2) != files  # interleaved
    assert sorted(data.shard_for_rank(files, 0, 2) + data.shard_for_rank(files, 1, 2)) == sorted(files)
    ld = data.RayShardLoader(files, 2, rank=1, world=2, pin=False)  # rank 1 of 2 owns exactly one of the 3 files
    b = ld.next()
    ld.close()
    assert b.shape == (8192, 9)
    own = np.load(data.shard_for_rank(files, 1, 2)[0])
    assert np.array_equal(b[:4096].numpy(), own) and np.array_equal(b[4096:].numpy(), own)


def test_native_shard_reader(tmp_path):
    """The reader threads of libr2l_hip.so against np.load: bit-identical payloads, every file exactly once per
    permutation (InfiniteSampler semantics, main.py:759-767), seeded order, .npy v2 headers, loud shape errors."""
    from numpy.lib import format as npf
    from r2l_amd import data
    n_files, rows = 7, 64
    files = []
    for k in range(n_files):
        a = np.full((rows, 9), float(k), np.float32) + np.arange(rows * 9, dtype=np.float32).reshape(rows, 9) / 1024
        f = str(tmp_path / ("data_%d.npy" % k))
        if k % 2:
            np.save(f, a)
        else:  # NumPy format 2.0 (4-byte header length)
            with open(f, "wb") as fp:
                npf.write_array(fp, a, version=(2, 0))
        files.append(f)
    seen = []
    ld = data.RayShardLoader(files, 3, seed=5, pin=False, threads=3)
    for _ in range(7):  # 21 shards = 3 full permutations
        b = ld.next().clone()
        assert b.shape == (3 * rows, 9)
        for j in range(3):
            blk = b[j * rows:(j + 1) * rows].numpy()
            k = int(blk[0, 0])
            assert np.array_equal(blk, np.load(files[k]))
            seen.append(k)
    assert ld.files_read() >= 21
    ld.close()
    for e in range(3):
        assert sorted(seen[e * 7:(e + 1) * 7]) == list(range(7))
    assert seen[:7] != seen[7:14] or seen[7:14] != seen[14:21]  # reshuffled between epochs
    ld2 = data.RayShardLoader(files, 3, seed=5, pin=False, threads=1)
    again = []
    for _ in range(2):
        b = ld2.next()
        again += [int(b[j * rows, 0]) for j in range(3)]
    ld2.close()
    assert again == seen[:6]  # the order depends on the seed only, not on thread timing
    np.save(str(tmp_path / "data_9.npy"), np.zeros((rows + 1, 9), np.float32))
    bad = data.RayShardLoader(files[:1] + [str(tmp_path / "data_9.npy")], 2, pin=False)
    with pytest.raises(RuntimeError, match="expected"):
        bad.next()
    bad.close()
    np.save(str(tmp_path / "data_10.npy"), np.zeros((rows, 9), np.float64))
    with pytest.raises(RuntimeError, match="float32"):
        data.RayShardLoader([str(tmp_path / "data_10.npy")], 1, pin=False)


def test_load_blender_synthetic(tmp_path):
    """A tiny Blender-format scene written with PIL: loader returns the documented shapes; half_res = 2x2 mean."""
    import json
    from PIL import Image
    from r2l_amd import data
    rng = np.random.RandomState(0)
    for split, n in (("train", 3), ("val", 1), ("test", 2)):
        os.makedirs(tmp_path / split)
        frames = []
        for i in range(n):
            img = (rng.rand(8, 8, 4) * 255).astype(np.uint8)
            Image.fromarray(img).save(tmp_path / split / ("r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i),
                           "transform_matrix": data.pose_spherical(30. * i, -30., 4.).tolist()})
        with open(tmp_path / ("transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, f)
    imgs, poses, render_poses, hwf, i_split = data.load_blender_data(str(tmp_path), half_res=True, testskip=1)
    assert imgs.shape == (6, 4, 4, 4) and poses.shape == (6, 4, 4) and render_poses.shape == (40, 4, 4)
    assert hwf[0] == 4 and abs(hwf[2] - 0.5 * 8 / np.tan(0.5 * 0.6911112070083618) / 2) < 1e-9
    assert [len(s) for s in i_split] == [3, 1, 2]
    full, *_ = data.load_blender_data(str(tmp_path), half_res=False, testskip=1)
    np.testing.assert_allclose(imgs.numpy(), full.numpy().reshape(6, 4, 2, 4, 2, 4).mean(axis=(2, 4)), atol=1e-7)


def test_half_res_hand_computed_4x4_fixture(tmp_path):
    """--half_res (dataset/load_blender.py:100-112: cv2.resize(img, (H/2, W/2), interpolation=cv2.INTER_AREA) on the float image):
    for an exact factor of 2 INTER_AREA is the mean of each 2x2 block.  cv2 is on neither box, so the resize is pinned here by a
    HAND-COMPUTED 4x4 fixture instead of by cv2's output: every expected value below was worked out on paper
    from the 8-bit pixels, (a + b + c + d) / (4 * 255); float32 rounding of either summation order stays within one ulp of it."""
    import json
    from PIL import Image
    from r2l_amd import data
    # channel R: blocks with exactly representable means; G: mixed values; B: a ramp; A: opaque / half / transparent
    R = np.array([[0, 255, 255, 255], [255, 0, 255, 255], [0, 0, 51, 51], [0, 0, 204, 204]], np.uint8)
    G = np.array([[10, 20, 1, 2], [30, 40, 3, 4], [100, 150, 7, 7], [200, 250, 7, 8]], np.uint8)
    B = np.arange(16, dtype=np.uint8).reshape(4, 4) * 16
    A = np.array([[255, 255, 0, 0], [255, 255, 0, 0], [255, 0, 128, 128], [0, 255, 128, 128]], np.uint8)
    want = np.zeros((2, 2, 4))
    want[..., 0] = [[510 / 1020, 1020 / 1020], [0 / 1020, 510 / 1020]]      # (0+255+255+0), (255*4) | 0, (51+51+204+204)
    want[..., 1] = [[100 / 1020, 10 / 1020], [700 / 1020, 29 / 1020]]        # 10+20+30+40, 1+2+3+4 | 100+150+200+250, 7+7+7+8
    want[..., 2] = [[160 / 1020, 288 / 1020], [672 / 1020, 800 / 1020]]      # 0+16+64+80, 32+48+96+112 | 128+144+192+208, 160+176+224+240
    want[..., 3] = [[1020 / 1020, 0 / 1020], [510 / 1020, 512 / 1020]]
    img = np.stack([R, G, B, A], -1)
    for split in ("train", "val", "test"):
        os.makedirs(tmp_path / split)
        Image.fromarray(img).save(tmp_path / split / "r_0.png")
        with open(tmp_path / ("transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": 0.6911112070083618,
                       "frames": [{"file_path": "./%s/r_0" % split, "transform_matrix": data.pose_spherical(0., -30., 4.).tolist()}]}, f)
    imgs, _, _, hwf, _ = data.load_blender_data(str(tmp_path), half_res=True, testskip=1)
    assert imgs.shape == (3, 2, 2, 4) and imgs.dtype == torch.float32 and hwf[:2] == [2, 2]
    got = imgs[0].numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 2.0 ** -24, np.abs(got - want).max()  # one ulp of a float32 in [0.5, 1)
    assert got[0, 1, 0] == 1.0 and got[1, 0, 0] == 0.0 and got[0, 1, 3] == 0.0 and got[0, 0, 3] == 1.0  # the exact cases
    # and the white-background composite the drivers apply to it (main.py:933-937): rgb * a + (1 - a)
    comp = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    assert torch.equal(comp[0, 0, 1], torch.ones(3))  # transparent block -> white
    assert abs(comp[0, 1, 1, 1].item() - ((29 / 1020) * (512 / 1020) + 1 - 512 / 1020)) < 1e-6


def test_convert_images_to_ray_shards(tmp_path):
    """README step 4 data: every pixel of the train views becomes one [o,d,rgb] row; shards feed BlenderDataset_v2."""
    from tests.test_driver_cpu import make_scene
    from r2l_amd import data
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)  # 2 train views, half_res 64x64 -> 8192 rays -> 2 shards
    savedir, n = data.convert_images_to_ray_shards(scene, ("train",), rng=np.random.RandomState(0))
    assert n == 2 and sorted(os.listdir(savedir)) == ["train_1.npy", "train_2.npy"]
    rows = np.concatenate([np.load(os.path.join(savedir, f)) for f in sorted(os.listdir(savedir))])
    assert rows.shape == (8192, 9) and rows.dtype == np.float32
    imgs, poses, _, hwf, i_split = data.load_blender_data(scene, half_res=True, testskip=1)
    imgs = imgs[..., :3] * imgs[..., -1:] + (1. - imgs[..., -1:])
    # the multiset of rgb values is exactly the composited train images; origins are the two camera centres
    ref = imgs[i_split[0]].reshape(-1, 3).numpy()
    assert np.allclose(np.sort(rows[:, 6:].sum(1)), np.sort(ref.sum(1)), atol=1e-6)
    centres = {tuple(np.round(p[:3, 3].numpy(), 4)) for p in poses[i_split[0]]}
    assert {tuple(np.round(r, 4)) for r in rows[:, :3]} == centres
    files = data.list_ray_shards(savedir, pseudo_ratio=-1)
    assert len(files) == 2  # 'train_*' files count as original data in BlenderDataset_v2's selection rule


def test_all_blender_scene_configs_parse(tmp_path):
    """Teacher and student config files for the 8 NeRF-synthetic scenes at 400x400 and 800x800, as tools/gen_configs.py
    writes them (the repo tracks the lego ones, the scene BASELINE.json names; they must equal the generator's output)."""
    import importlib.util
    from r2l_amd.options import parse_args
    spec = importlib.util.spec_from_file_location("gen_configs", os.path.join(ROOT, "tools", "gen_configs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    scenes = ["chair", "drums", "ficus", "hotdog", "lego", "materials", "mic", "ship"]
    assert gen.SCENES == scenes
    gen.write_configs(scenes, str(tmp_path))
    tracked = sorted(os.listdir(os.path.join(ROOT, "configs")))
    assert tracked == ["lego.txt", "lego_800x800.txt", "lego_noview.txt", "lego_noview_800x800.txt"]
    for name in tracked:
        assert open(os.path.join(ROOT, "configs", name)).read() == open(str(tmp_path / name)).read()
    for scene in scenes:
        for student in (False, True):
            for full in (False, True):
                name = scene + ("_noview" if student else "") + ("_800x800" if full else "") + ".txt"
                a = parse_args(["--config", str(tmp_path / name)])
                assert a.datadir.endswith("nerf_synthetic/" + scene) and a.dataset_type == "blender"
                assert a.half_res == (not full) and a.use_viewdirs == (not student) and a.white_bkgd
                assert (a.N_samples, a.N_importance, a.lrate_decay) == (64, 128, 500)


def _student_args(inact="relu", body_arch="resmlp", netdepth=88):
    import argparse
    trial = argparse.Namespace(ON=True, body_arch=body_arch, inact=inact, outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    return argparse.Namespace(netdepth=netdepth, netwidth=256, layerwise_netwidths="", act="relu", linear_tail=False,
                              use_residual=True, trial=trial)


def test_seeded_construction_coincides_with_reference(golden_dir):
    """torch.manual_seed(s); NeRF_v3_2(args, 1008, 3) must give the reference's parameters: its constructor draws (and
    discards) a plain D-2 layer body before the ResMLP blocks (model/nerf_raybased.py:502-505).  The oracle replays that
    order, and the golden fixture pins the oracle's tensors to the reference's by checksum."""
    from model.nerf_raybased import NeRF_v3_2
    for seed, depth in ((0, 88), (7, 10)):
        torch.manual_seed(seed)
        net = NeRF_v3_2(_student_args(netdepth=depth), 1008, 3)
        ref = O.make_state_dict(n_block=(depth - 2) // 2, seed=seed)
        sd = net.state_dict()
        assert list(sd) == list(ref)
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k
    g = np.load(os.path.join(golden_dir, "r2l_w256d88.npz"))
    torch.manual_seed(0)  # the REFERENCE's own seed-0 tensors, by the checksums gen_golden.py froze
    sd = NeRF_v3_2(_student_args(), 1008, 3).state_dict()
    np.testing.assert_allclose([v.double().sum().item() for v in sd.values()], g["param_sums"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([v.double().abs().sum().item() for v in sd.values()], g["param_abs_sums"], rtol=1e-12)


def test_inner_activation_variants_match_reference_structure():
    """--trial.inact none builds Linear, Linear (state_dict keys body.b.body.{0,1}) as the reference does; only
    --trial.inact relu is on the HIP path — anything else must be reported unsupported, not silently computed with ReLU."""
    from model.nerf_raybased import NeRF_v3_2
    from r2l_amd.engine import supported_reason
    net = NeRF_v3_2(_student_args(inact="none", netdepth=6), 1008, 3)
    assert "body.0.body.1.weight" in net.state_dict() and "body.0.body.2.weight" not in net.state_dict()
    assert supported_reason(net) is not None
    net = NeRF_v3_2(_student_args(inact="lrelu", netdepth=6), 1008, 3)
    assert isinstance(net.body[0].body[1], torch.nn.LeakyReLU) and "inact" in supported_reason(net)
    assert supported_reason(NeRF_v3_2(_student_args(netdepth=6), 1008, 3)) is None
    # plain-MLP body through --trial.body_arch mlp: the reference builds it twice (two sets of RNG draws)
    torch.manual_seed(3)
    a = NeRF_v3_2(_student_args(body_arch="mlp", netdepth=5), 1008, 3)
    torch.manual_seed(3)
    head = torch.nn.Linear(1008, 256)
    for _ in range(3):
        torch.nn.Linear(256, 256)
    first = torch.nn.Linear(256, 256)
    assert torch.equal(a.head[0].weight, head.weight) and torch.equal(a.body[0].weight, first.weight)


def test_product_path_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing that ships (the package, the reference-named modules and CLI around it) may
    import it or read /root/reference; bench.py and __graft_entry__ may — only inside the checker / cpu_baseline functions."""
    import ast
    shipped = []
    for top in ("r2l_amd", "model", "utils", "dataset", "smilelogging"):
        for dirpath, _, names in os.walk(os.path.join(ROOT, top)):
            shipped += [os.path.join(dirpath, n) for n in names if n.endswith(".py")]
    shipped += [os.path.join(ROOT, n) for n in ("main.py", "option.py")]
    assert len(shipped) > 15
    for path in shipped:
        tree = ast.parse(open(path).read())
        docstrings = set()  # (the reference is CITED in docstrings, file:line; it must not appear in code)
        for node in ast.walk(tree):
            if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)) and node.body and \
                    isinstance(node.body[0], ast.Expr) and isinstance(node.body[0].value, ast.Constant):
                docstrings.add(id(node.body[0].value))
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in docstrings:
                assert "/root/reference" not in node.value, path
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), path
    for name, allowed in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"smoke"})):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            own = fn.body if isinstance(fn, ast.Module) else ast.walk(fn)
            for node in own:
                if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                    assert isinstance(fn, ast.FunctionDef) and fn.name in allowed, (name, getattr(fn, "name", "module level"))
    for dirpath, _, names in os.walk(os.path.join(ROOT, "r2l_amd", "csrc")):
        for n in names:
            assert "oracle" not in open(os.path.join(dirpath, n), errors="ignore").read().lower(), n


def test_missing_library_fails_loudly(tmp_path):
    """No non-HIP fallback: with the shared library absent every entry into the hot path raises (it does not compute on the
    CPU, the oracle or eager torch)."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['R2L_LIB_PATH'] = %r\n"
            "from r2l_amd import _lib\n"
            "try:\n    _lib.load()\nexcept RuntimeError as e:\n    print('RAISED', e)\n" % (ROOT, str(tmp_path / "nope.so")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED" in r.stdout and "no non-HIP fallback" in r.stdout, r.stdout + r.stderr


def test_c99_host_compiles_links_and_runs(tmp_path):
    """include/r2l_hip.h is a C header (not only C++): a C99 host (tests/chost/host.c, -pedantic -Werror) compiles against it,
    links to libr2l_hip.so and runs its host-side entry points — sizes, dispatch queries with an r2l_config, argument checks —
    without a GPU."""
    import shutil
    from r2l_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    _lib.load()
    exe = str(tmp_path / "chost")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "chost", "host.c"), "-o", exe, "-L", libdir, "-lr2l_hip", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C99 host ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (dev container only)")
def test_checkpoint_written_here_loads_in_the_reference(tmp_path):
    """"checkpoints written here load in the reference".  save_ckpt's .tar (pickled network_fn + state dict +
    Adam-format optimizer state) is opened in a SEPARATE process whose sys.path holds only the reference: the pickle resolves to
    the reference's own model.nerf_raybased.NeRF_v3_2, the forward matches bit for bit, torch.optim.Adam takes the state."""
    import argparse
    from model.nerf_raybased import NeRF_v3_2
    from r2l_amd.checkpoint import save_ckpt
    trial = argparse.Namespace(ON=True, body_arch="resmlp", inact="relu", outact="none", res_scale=1., n_learnable=2,
                               n_block=-1, near=-1, far=-1)
    args = argparse.Namespace(netdepth=8, netwidth=256, layerwise_netwidths="", act="relu", linear_tail=False,
                              use_residual=True, trial=trial)
    torch.manual_seed(0)
    m = NeRF_v3_2(args, 1008, 3)
    x = torch.randn(5, 1008)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    m(x).sum().backward()
    opt.step()
    # (with the arithmetic record the drivers store, driver.apply_arithmetic: a key the reference's loaders never read)
    save_ckpt(str(tmp_path / "ckpt.tar"), 7, m, opt.state_dict(), 1.0, 3,
              r2l_config={"precision": "fp32_mfma", "dw_mode": "exact", "requested": {"precision": "fp32_mfma", "dw_mode": "auto"}})
    torch.save({"x": x, "y": m(x).detach()}, str(tmp_path / "io.pt"))
    code = ("import sys; sys.path.insert(0, '/root/reference')\n"
            "import torch, model.nerf_raybased as rm\n"
            "assert rm.__file__.startswith('/root/reference')\n"
            "ck = torch.load(%r, weights_only=False, map_location='cpu')\n"
            "m = ck['network_fn']\n"
            "assert type(m) is rm.NeRF_v3_2 and ck['global_step'] == 7 and ck['r2l_config']['precision'] == 'fp32_mfma'\n"
            "m.load_state_dict(ck['network_fn_state_dict'])\n"
            "io = torch.load(%r)\n"
            "assert torch.equal(m(io['x']), io['y'])\n"
            "opt = torch.optim.Adam(m.parameters(), lr=5e-4)\n"
            "opt.load_state_dict(ck['optimizer_state_dict'])\n"
            "assert len(opt.state_dict()['state']) == 16\n"
            "print('REFERENCE LOADED IT')\n" % (str(tmp_path / "ckpt.tar"), str(tmp_path / "io.pt")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert "REFERENCE LOADED IT" in r.stdout, r.stdout + r.stderr


def test_checkpoint_helpers_error_behaviour():
    """helpers:333-382 / main.py:1088-1094: load_weights_v2 insists that model and state dict agree on the DataParallel 'module.'
    prefix (NotImplementedError otherwise, as the reference), undataparallel strips exactly one prefix, parse_expid_iter reads the
    experiment id out of a smilelogging path and answers 'Unknown' elsewhere."""
    from collections import OrderedDict
    from r2l_amd.checkpoint import load_weights_v2, parse_expid_iter, undataparallel
    net = torch.nn.Sequential(torch.nn.Linear(3, 2))
    sd = net.state_dict()
    load_weights_v2(net, {"k": sd}, "k")
    with pytest.raises(NotImplementedError):
        load_weights_v2(net, {"k": OrderedDict(("module." + k, v) for k, v in sd.items())}, "k")
    wrapped = torch.nn.Sequential(OrderedDict(module=net))  # named_modules() now starts with 'module.'
    with pytest.raises(NotImplementedError):
        load_weights_v2(wrapped, {"k": sd}, "k")
    load_weights_v2(wrapped, {"k": OrderedDict(("module." + k, v) for k, v in sd.items())}, "k")
    assert list(undataparallel(OrderedDict(("module." + k, v) for k, v in sd.items()))) == list(sd)
    assert undataparallel(wrapped) is net
    assert parse_expid_iter("Experiments/R2L__lego_SERVER142-20210704-150540/weights/ckpt_200000.tar") == \
        ("SERVER142-20210704-150540", "ckpt_200000")
    assert parse_expid_iter("/data/SERVER3/ckpt.tar") == ("Unknown", "Unknown") == parse_expid_iter("ckpt.tar")


@pytest.mark.parametrize("flags", [["--plucker"], ["--learn_depth", "1"], ["--shuffle_input"], ["--convert_to_onnx"],
                                   ["--given_render_path_rays", "x.pt"], ["--dataset_type", "llff"]])
def test_out_of_scope_variants_fail_loudly(flags):
    """Reference variants outside the accelerated path parse — the reference's command lines keep working up to
    that point — and then raise NotImplementedError; they are never silently ignored."""
    from r2l_amd.options import parse_args, validate_accelerated
    base = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt")]
    validate_accelerated(parse_args(base))
    with pytest.raises(NotImplementedError):
        validate_accelerated(parse_args(base + flags))


def test_logger_layout_round_trips_through_parse_expid_iter(tmp_path, monkeypatch):
    """Experiments/<name>_SERVER<id>-<time>/{weights, gen_img, log} (smilelogging/logger.py:234-288) — the layout the reference's
    `--render_only` reads the experiment id back from (main.py:1088-1094): a checkpoint path under weights/ gives the ExpID."""
    import argparse
    from r2l_amd.checkpoint import parse_expid_iter
    from r2l_amd.logger import Logger
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("R2L_SERVER_ID", "142")
    lg = Logger(argparse.Namespace(experiment_name="R2L__lego", debug=False, trial=argparse.Namespace(ON=True)), rank=0)
    assert re.fullmatch(r"SERVER142-\d{8}-\d{6}", lg.ExpID)
    for d in (lg.weights_path, lg.gen_img_path, lg.log_path):
        assert os.path.isdir(d) and d.startswith(os.path.join("Experiments", "R2L__lego_" + lg.ExpID))
    lg.info("hello")
    assert "hello" in open(os.path.join(lg.log_path, "log.txt")).read()
    assert os.path.exists(os.path.join(lg.log_path, "args.yaml"))
    assert parse_expid_iter(os.path.join(lg.weights_path, "ckpt_1000.tar")) == (lg.ExpID, "ckpt_1000")
    other = Logger(argparse.Namespace(experiment_name="x", debug=True), rank=1)  # other ranks: same attributes, no files
    assert other.exp_path.startswith("Debug_Dir") and not os.path.exists(other.exp_path)


def test_native_png_writer_round_trip(tmp_path):
    """include/r2l_hip.h r2l_png_writer_*: the encoder threads that replace imageio.imwrite in render_path (main.py:337-344).
    RGB, grey and RGBA frames, odd sizes, many jobs in flight, out-of-order completion: every file decodes (PIL) to exactly the
    bytes handed over; wait(job) covers all earlier jobs; a path that cannot be written is reported, not swallowed."""
    import ctypes
    import numpy as np
    from PIL import Image
    from r2l_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.r2l_png_writer_open(0, 1, ctypes.byref(h)) != 0 and b"n_threads" in lib.r2l_last_error()
    _lib.check(lib.r2l_png_writer_open(4, 1, ctypes.byref(h)), "open")
    rng = np.random.default_rng(0)
    jobs = []
    for i, shape in enumerate([(400, 400, 3), (37, 53, 3), (1, 1, 3), (64, 48, 1), (33, 17, 4)] * 6):
        if i % 2:  # smooth content (compresses) and noise (does not)
            arr = rng.integers(0, 256, size=shape, dtype=np.uint8)
        else:
            yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
            arr = np.stack([(xx * 3 + yy * (c + 1) + i) % 256 for c in range(shape[2])], -1).astype(np.uint8)
        arr = np.ascontiguousarray(arr)
        path = str(tmp_path / ("f%03d.png" % i))
        job = ctypes.c_int64()
        _lib.check(lib.r2l_png_writer_submit(h, path.encode(), ctypes.c_void_p(arr.ctypes.data), shape[0], shape[1], shape[2],
                                             None, ctypes.byref(job)), "submit")
        assert job.value == i
        jobs.append((path, arr))
    _lib.check(lib.r2l_png_writer_wait(h, 9), "wait")  # jobs 0 .. 9 are on disk now
    for path, arr in jobs[:10]:
        got = np.asarray(Image.open(path))
        assert np.array_equal(got.reshape(arr.shape), arr), path
    _lib.check(lib.r2l_png_writer_wait(h, -1), "wait all")
    for path, arr in jobs:
        got = np.asarray(Image.open(path))
        assert np.array_equal(got.reshape(arr.shape), arr), path
    bad = np.zeros((2, 2, 3), np.uint8)
    assert lib.r2l_png_writer_submit(h, str(tmp_path / "no_such_dir" / "x.png").encode(), ctypes.c_void_p(bad.ctypes.data), 2, 2, 3,
                                     None, None) == 0
    assert lib.r2l_png_writer_close(h) != 0 and b"cannot open" in lib.r2l_last_error()
    assert lib.r2l_png_writer_submit(None, b"x", None, 1, 1, 3, None, None) != 0


def test_range_words_decoding():
    """_lib.decode_range_words: the telemetry of the fp16 kernels' range control from the 16 status words (layout in
    include/r2l_hip.h); an area the library has not initialised yet (no magic word) reads as scale 1, nothing seen."""
    import struct
    from r2l_amd import _lib
    bits = lambda x: struct.unpack("<i", struct.pack("<f", x))[0]
    w = torch.zeros(16, dtype=torch.int32)
    assert _lib.decode_range_words(w) == {"amax": 0.0, "scale": 1.0, "headroom": float("inf"), "trips": 0, "rescales": 0, "flag": 0}
    w[1], w[2], w[3], w[4], w[5], w[6], w[7] = bits(5000.0), bits(16.0), bits(1 / 16.), _lib.RANGE_MAGIC, 1, bits(123.0), 2
    info = _lib.decode_range_words(w)
    assert info["amax"] == 80000.0 and info["scale"] == 16.0 and abs(info["headroom"] - 32768.0 * 16 / 80000.0) < 1e-9
    assert info["trips"] == 1 and info["rescales"] == 2 and info["flag"] == 0
    w[1] = 0  # nothing seen since the last commit: the previous epoch's peak is reported
    assert _lib.decode_range_words(w)["amax"] == 123.0
