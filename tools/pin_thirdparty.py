#!/usr/bin/env python3
"""One-command pin of the three rows whose reference arithmetic lives in third-party packages that this image does not ship
(SURVEY 8f N2 / N3):

    python tools/pin_thirdparty.py          # run it wherever ignite / lpips + torchvision / cv2 are importable

For every package that imports it writes a fixture under tests/golden/ from THE PACKAGE ITSELF, called the way the reference calls
it; the tests pick the fixtures up when they exist and say "unpinned" in their skip reason when they do not:

  thirdparty_ssim.npz        ignite.metrics.SSIM(data_range=1.0) as pnpflow/utils.py:780-816 uses it (update((rec, clean)); compute())
                             on seeded images of four shapes (square, 1-channel, ragged)                    -> tests: *_thirdparty_ssim*
  thirdparty_lpips.npz       lpips.LPIPS(net='alex')(a, b, normalize=True) as pnpflow/utils.py:677-724 calls it, on seeded images,
                             TOGETHER WITH the published weights it ran on (torchvision AlexNet features + the five linear heads:
                             the engine's loader reads them from the fixture, so the test needs neither package)   -> *_thirdparty_lpips*
  thirdparty_paintbrush.npz  the reference's MaskGenerator stroke recipe (pnpflow/utils.py:904-969: random.seed(42), ten cv2.line strokes
                             per mask) rasterised by cv2.line itself at 64 x 96, 128^2 and 256^2           -> *_thirdparty_paintbrush*

Inputs are stored by recipe (seed + shape: numpy Philox, the generators of tests/conftest.py); outputs are arrays.  No package
source travels.  Nothing under pnpflow_amd/ imports this file.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
SSIM_SHAPES = [(3, 3, 64, 64), (2, 1, 28, 28), (1, 3, 256, 256), (2, 3, 50, 70)]
LPIPS_SHAPES = [(3, 3, 128, 128), (2, 3, 256, 256), (2, 3, 64, 96)]
BRUSH_SHAPES = [(4, 64, 96), (2, 128, 128), (2, 256, 256)]


def det_normal(shape, seed, idx=0):
    g = np.random.Generator(np.random.Philox(key=[seed, idx]))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float32))


def pair(shape, seed):
    """(clean, degraded) in [-1, 1]: the same recipe tests/test_gpu_parity.py uses for its metric tests."""
    a = det_normal(shape, seed).clamp(-1, 1)
    b = (a + 0.1 * det_normal(shape, seed + 1)).clamp(-1, 1)
    return a, b


def pin_ssim():
    from ignite.metrics import SSIM
    rec = {"shapes": np.array([list(s) for s in SSIM_SHAPES]), "seed": np.array(51)}
    for i, shape in enumerate(SSIM_SHAPES):
        a, b = pair(shape, 51)
        clean, recd = (a + 1) / 2, (b + 1) / 2                   # utils.postprocess: [-1, 1] -> [0, 1]
        vals = []
        for k in range(shape[0]):                                # per image (the engine reports per-image values, the reference their batch mean)
            m = SSIM(data_range=1.0); m.update((recd[k:k + 1], clean[k:k + 1])); vals.append(float(m.compute()))
        m = SSIM(data_range=1.0); m.update((recd, clean))
        rec[f"per_image_{i}"] = np.array(vals, dtype=np.float64)
        rec[f"batch_{i}"] = np.array(float(m.compute()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "thirdparty_ssim.npz"), **rec)
    print("pinned SSIM against ignite", rec["batch_0"])


def pin_lpips():
    import lpips
    net = lpips.LPIPS(net="alex").eval()
    rec = {"shapes": np.array([list(s) for s in LPIPS_SHAPES]), "seed": np.array(71)}
    for k, v in net.state_dict().items():
        rec["w::" + k] = v.detach().cpu().numpy()
    with torch.no_grad():
        for i, shape in enumerate(LPIPS_SHAPES):
            a, b = pair(shape, 71)
            # utils.compute_lpips: postprocess to [0, 1], back to [-1, 1], then normalize=True once more
            rec[f"d_{i}"] = net(a, b, normalize=True).reshape(-1).numpy().astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "thirdparty_lpips.npz"), **rec)
    print("pinned LPIPS against lpips + torchvision", rec["d_0"])


def pin_paintbrush():
    import random

    import cv2
    rec = {"shapes": np.array([list(s) for s in BRUSH_SHAPES])}
    for i, (B, H, W) in enumerate(BRUSH_SHAPES):
        random.seed(42)                                          # MaskGenerator(..., rand_seed=42): utils.py:343, :921-922
        masks, strokes = [], []
        size = int((W + H) * 0.08)
        for _ in range(B):
            img = np.zeros((H, W, 1), np.uint8)
            for _ in range(10):                                  # utils.py:933-940
                x1, x2 = random.randint(W // 2 - 30, W // 2 + 30), random.randint(W // 2 - 30, W // 2 + 30)
                y1, y2 = random.randint(H // 2 - 30, H // 2 + 30), random.randint(H // 2 - 30, H // 2 + 30)
                t = random.randint(8, size)
                cv2.line(img, (x1, y1), (x2, y2), (255, 255, 255), t)
                strokes.append([x1, y1, x2, y2, t])
            masks.append(((1 - img).transpose(2, 0, 1)[0].astype(np.int64) - 1 == 0))      # paintbrush_mask: (mask - 1 == 0) keeps the pixel
        rec[f"keep_{i}"] = np.stack(masks).astype(np.uint8)
        rec[f"strokes_{i}"] = np.array(strokes, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "thirdparty_paintbrush.npz"), **rec)
    print("pinned the paintbrush raster against cv2.line")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    done = 0
    for name, fn in (("ignite", pin_ssim), ("lpips", pin_lpips), ("cv2", pin_paintbrush)):
        try:
            fn(); done += 1
        except ImportError as exc:
            print(f"{name}: not importable here ({exc}); its fixture stays absent and the tests keep saying 'unpinned'")
    sys.exit(0 if done else 3)
