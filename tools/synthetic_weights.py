"""Seed-fixed synthetic state_dict for a pnpflow_amd.models.UNet when no checkpoint is
available (product-side helper: does not import oracle/)."""
import math
import zlib

import numpy as np
import torch


def synthetic_state_dict(model, seed=0):
    sd = {}
    for name, shape in model.state_dict_shapes().items():   # names/shapes reported by the engine
        rng = np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))
        u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
        if name.endswith(".W") and len(shape) == 1:          # NCSN++ Fourier frequencies: 16 N(0,1) (layerspp.py:36)
            u = (rng.standard_normal(size=shape) * 16.0).astype(np.float32)
        elif name.endswith(".W"):                             # NCSN++ NIN weights [in][out]
            u *= np.float32(math.sqrt(3.0 / shape[0]))
        elif name.endswith("weight") and len(shape) >= 2:
            u *= np.float32(math.sqrt(3.0 / int(np.prod(shape[1:]))))
        elif name.endswith("weight"):
            u = np.float32(1.0) + np.float32(0.1) * u
        else:
            u *= np.float32(0.05)
        sd[name] = torch.from_numpy(u)
    return sd
