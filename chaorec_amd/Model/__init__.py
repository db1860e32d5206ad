from .LightGCN import LightGCN  # noqa: F401
