"""Small numeric helpers shared by the drivers (reference: utils/run_nerf_raybased_helpers.py:14-20)."""
import numpy as np
import torch


def to_tensor(x, device=None):
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    return x.to(device) if isinstance(x, torch.Tensor) else torch.Tensor(x).to(device)


def to_array(x):
    return x if isinstance(x, np.ndarray) else x.data.cpu().numpy()


def to8b(x):
    return (255 * np.clip(to_array(x), 0, 1)).astype(np.uint8)


def img2mse(x, y):
    return torch.mean((x - y)**2)


def mse2psnr(x):
    return -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))
