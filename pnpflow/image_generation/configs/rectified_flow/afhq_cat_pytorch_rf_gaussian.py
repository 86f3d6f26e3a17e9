from pnpflow_amd.image_generation.configs.rectified_flow.afhq_cat_pytorch_rf_gaussian import get_config  # noqa: F401
