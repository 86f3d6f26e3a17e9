"""create_model / get_model_fn of pnpflow/image_generation/models/utils.py:91-135 for the engine-backed NCSNpp."""
from __future__ import annotations

from .ncsnpp import NCSNpp

_MODELS = {"ncsnpp": NCSNpp}


def get_model(name):
    return _MODELS[name]


def create_model(config, device_index: int = 0):
    """models/utils.py:91-103 (no DataParallel wrapper: one process per GPU; `module.`-prefixed checkpoints still load)."""
    name = config["model"]["name"] if isinstance(config, dict) else config.model.name
    return get_model(name)(config, device_index=device_index)


def get_model_fn(model, train=False):
    """models/utils.py:106-135"""
    if train:
        raise NotImplementedError("inference engine")

    def model_fn(x, labels):
        model.eval()
        return model(x, labels)

    return model_fn
