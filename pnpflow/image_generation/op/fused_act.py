from pnpflow_amd.image_generation.op.fused_act import FusedLeakyReLU, fused_bias_act, fused_leaky_relu  # noqa: F401
