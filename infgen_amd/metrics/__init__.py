"""Device-side rollout-metric features - mirrors of infgen/metrics/*_features.py."""
from .interact_features import compute_distance_to_nearest_object

__all__ = ['compute_distance_to_nearest_object']
