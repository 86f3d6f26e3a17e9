import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def det_normal(shape, seed, idx=0):
    """Same deterministic-input recipe as tools/make_golden.py (numpy Philox)."""
    g = np.random.Generator(np.random.Philox(key=[seed, idx]))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float32))


def det_image(shape, seed):
    """Same synthetic clean-image recipe as tools/make_golden.py."""
    x = det_normal(shape, seed, 7)
    k = torch.ones(shape[1], 1, 3, 3) / 9.0
    for _ in range(5):
        x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), k, groups=shape[1])
    lo = x.amin(dim=(1, 2, 3), keepdim=True); hi = x.amax(dim=(1, 2, 3), keepdim=True)
    return ((x - lo) / (hi - lo) * 2 - 1).contiguous()


def checksums(t):
    d = t.double()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], dtype=np.float64)


CFGS = {
    "mnist": dict(input_channels=1, input_height=28, ch=32, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(16,)),
    "tiny4": dict(input_channels=3, input_height=64, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=1, attn_resolutions=(16, 8)),
    "celeba128": dict(input_channels=3, input_height=128, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)),
    "afhq256": dict(input_channels=3, input_height=256, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)),
    # shapes the BASELINE nets never produce: ragged tiles at every level (48 -> 24 -> 12 px), equal widths at two levels,
    # attention with 144 tokens (unfused path) at 64 channels, a 1-channel image with three levels
    "odd48": dict(input_channels=3, input_height=48, ch=32, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(12,)),
    "gray40": dict(input_channels=1, input_height=40, ch=32, ch_mult=(1, 1, 4), num_res_blocks=1, attn_resolutions=(20, 10)),
}


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def thirdparty():
    """Fixtures written by tools/pin_thirdparty.py FROM the third-party packages the reference's metrics / masks live in (ignite, lpips +
    torchvision, cv2).  None of them is in this image: until someone runs the tool where they are, the rows stay unpinned and say so."""
    def load(name, package):
        path = os.path.join(GOLDEN, f"thirdparty_{name}.npz")
        if not os.path.isfile(path):
            pytest.skip(f"parity unpinned: tests/golden/thirdparty_{name}.npz absent - run `python tools/pin_thirdparty.py` where {package} is installed")
        return np.load(path)
    return load


def thirdparty_pair(shape, seed):
    """(clean, degraded) of tools/pin_thirdparty.py."""
    a = det_normal(shape, seed).clamp(-1, 1)
    b = (a + 0.1 * det_normal(shape, seed + 1)).clamp(-1, 1)
    return a, b
