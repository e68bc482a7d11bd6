"""End-to-end on the GPU through the real CLI surface: teacher pseudo-data generation (utils/create_data.py 'rand'),
then R2L distillation training on those shards with hard-ray mining, test-set evaluation, checkpoint save + resume."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O
from tests.test_driver_cpu import ROOT, make_scene

pytestmark = pytest.mark.gpu


def test_create_data_then_train(tmp_path, monkeypatch):
    from r2l_amd import create_data, driver
    from r2l_amd.checkpoint import load_ckpt
    monkeypatch.chdir(tmp_path)
    scene = str(tmp_path / "scene")
    os.makedirs(scene)
    make_scene(scene, size=128)  # half_res -> 64x64 = 4096 rays per pose = one shard per pose
    csd, fsd = O.make_teacher_state_dicts(5, 2, alpha_bias=0.5)
    torch.save({"network_fn_state_dict": csd, "network_fine_state_dict": fsd}, str(tmp_path / "teacher.tar"))
    kd = str(tmp_path / "pseudo")
    out = create_data.main(["--create_data", "rand", "--config", os.path.join(ROOT, "configs", "lego.txt"), "--datadir",
                            scene, "--teacher_ckpt", str(tmp_path / "teacher.tar"), "--n_pose_kd", "3",
                            "--create_data_chunk", "2", "--datadir_kd", scene + ":" + kd, "--experiment_name", "cd"])
    files = sorted(os.listdir(kd))
    assert len(files) == 3 and out["n_rays"] == 3 * 4096
    rows = np.load(os.path.join(kd, files[0]))
    assert rows.shape == (4096, 9) and rows.dtype == np.float32
    assert np.all(np.isfinite(rows)) and rows[:, 6:].min() >= -1e-4 and rows[:, 6:].max() <= 1.0 + 1e-4
    assert np.allclose(np.linalg.norm(rows[:, :3], axis=1), 4.0, atol=1e-4)  # origins on the radius-4 sphere

    common = ["--model_name", "R2L", "--config", os.path.join(ROOT, "configs", "lego_noview.txt"), "--datadir", scene,
              "--n_sample_per_ray", "16", "--netwidth", "256", "--netdepth", "6", "--use_residual", "--trial.ON",
              "--trial.body_arch", "resmlp", "--testskip", "1", "--datadir_kd", kd, "--data_mode", "rays", "--N_rand", "2",
              "--hard_ratio", "0.2", "--hard_mul", "2", "--warmup_lr", "0.0001,200", "--i_print", "2", "--i_testset", "4",
              "--i_weights", "6", "--experiment_name", "train"]
    res = driver.main(common + ["--N_iters", "6"])
    wdir = res["logger"].weights_path
    assert sorted(os.listdir(wdir)) == ["ckpt.tar", "ckpt_best.tar"]
    ck = load_ckpt(os.path.join(wdir, "ckpt.tar"))
    assert ck["global_step"] == 6 and ck["optimizer_state_dict"]["state"][0]["exp_avg"].shape == (256, 1008)
    loss = res["trainer"].loss_out[0].item()
    assert np.isfinite(loss) and 0 < loss < 1
    # resume from it and take two more steps; the step counter and Adam moments carry over
    res2 = driver.main(common + ["--N_iters", "8", "--pretrained_ckpt", os.path.join(wdir, "ckpt.tar"), "--resume"])
    assert res2["trainer"].step_count == 8
    # render_only with the trained checkpoint reports PSNR on the 2 test views
    res3 = driver.main(common + ["--pretrained_ckpt", os.path.join(wdir, "ckpt.tar"), "--render_only", "--render_test"])
    assert res3["rgbs"].shape == (2, 64, 64, 3) and np.isfinite(res3["misc"]["test_psnr"].item())
