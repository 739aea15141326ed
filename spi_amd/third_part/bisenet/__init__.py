from .bisenet import BiSeNet  # noqa: F401
