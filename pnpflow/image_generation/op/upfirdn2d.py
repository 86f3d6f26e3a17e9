from pnpflow_amd.image_generation.op.upfirdn2d import upfirdn2d, upfirdn2d_xy  # noqa: F401
