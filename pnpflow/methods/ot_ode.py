from pnpflow_amd.methods.ot_ode import OT_ODE  # noqa: F401
