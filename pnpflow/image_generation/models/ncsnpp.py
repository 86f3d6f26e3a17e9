from pnpflow_amd.image_generation.models.ncsnpp import NCSNpp  # noqa: F401
