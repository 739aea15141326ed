from .api import FaceAlignment, LandmarksType  # noqa: F401
