from pnpflow_amd.image_generation.configs.rectified_flow.celeba_hq_pytorch_rf_gaussian import get_config  # noqa: F401
