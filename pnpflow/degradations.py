from pnpflow_amd.degradations import *  # noqa: F401,F403
from pnpflow_amd.degradations import Degradation, Denoising, BoxInpainting, RandomInpainting, GaussianDeblurring, Superresolution, PaintbrushInpainting  # noqa: F401
