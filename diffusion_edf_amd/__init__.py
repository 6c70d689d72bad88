"""MI355X-native implementation of the Diffusion-EDF SE(3) score-head hot path."""
from .gnn_data import FeaturedPoints  # noqa: F401

__version__ = "0.1.0"
