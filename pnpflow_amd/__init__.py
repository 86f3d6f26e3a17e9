"""pnpflow_amd: MI355X-native PnP-Flow restoration engine behind the reference's
`pnpflow.methods` / `pnpflow.degradations` / `pnpflow.utils` API.

The arithmetic lives in libpnpflow_hip.so (hand-written HIP for gfx950, C ABI in
include/pnpflow_hip.h); this package is the Python host side: it owns no numerics beyond
scalar schedule values, and PyTorch is used only for device memory and streams.
"""
__all__ = ["models", "degradations", "utils", "methods"]
