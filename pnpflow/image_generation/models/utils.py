from pnpflow_amd.image_generation.models.utils import create_model, get_model, get_model_fn  # noqa: F401
