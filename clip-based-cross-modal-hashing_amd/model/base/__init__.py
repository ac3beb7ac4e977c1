from .model import CLIP, build_model, convert_weights  # noqa: F401
