"""Import the *real* reference (read-only at /root/reference) in the build container.

Only used by tools/make_golden.py, i.e. to generate the golden vectors under tests/golden/
(tests/test_oracle_golden.py then validates oracle/ against them).  Nothing under tests/ (gpu or
not), bench.py or __graft_entry__.py imports this: /root/reference does not exist on
the GPU box.

Six third-party modules the reference imports at module top but that are absent
offline (and that do not take part in the hot-path arithmetic) are replaced by empty
stubs: torchvision, torchmetrics, ignite, cv2, deepinv, lpips.
"""
import sys
import types

import torch.nn as nn

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    class _Blk(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.models.inception = _stub("torchvision.models.inception", InceptionA=_Blk, InceptionC=_Blk, InceptionE=_Blk)
    tv.transforms = _stub("torchvision.transforms")
    tm = _stub("torchmetrics")
    tm.functional = _stub("torchmetrics.functional")
    tm.functional.image = _stub("torchmetrics.functional.image", peak_signal_noise_ratio=None)
    ig = _stub("ignite")
    ig.metrics = _stub("ignite.metrics", SSIM=None)
    _stub("cv2")
    _stub("deepinv")
    _stub("lpips")


def import_reference():
    """Returns (models, degradations, utils, pnp_flow) modules of the reference."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import pnpflow.models as models
    import pnpflow.utils as utils
    import pnpflow.degradations as degradations
    import pnpflow.methods.pnp_flow as pnp_flow
    return models, degradations, utils, pnp_flow
