from .base_class import BasePolicy  # noqa: F401
