from pnpflow_amd.methods.pnp_flow import PNP_FLOW  # noqa: F401
