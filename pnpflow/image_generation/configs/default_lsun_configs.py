from pnpflow_amd.image_generation.configs.default_lsun_configs import get_default_configs  # noqa: F401
