from .image_projector import ImageProjector, PinholeCamera

__all__ = ["ImageProjector", "PinholeCamera"]
