from pnpflow_amd.models import UNet  # noqa: F401
