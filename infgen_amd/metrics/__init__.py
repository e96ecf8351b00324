"""Device-side rollout-metric features - mirrors of infgen/metrics/*_features.py."""
from .interact_features import compute_distance_to_nearest_object, compute_time_to_collision_with_object_in_front
from .trajectory_features import compute_kinematic_features
from .map_features import compute_distance_to_road_edge, tensorize_polylines
from .placement_features import compute_num_placement, compute_distance_placement
from .compute_metrics import (MetricFeatures, ObjectTrajectories, ScenarioRollouts, compute_metric_features,
                              format_rollouts, get_scenario_id_int_tensor, output_to_rollouts)
from .scores import compute_scenario_metrics, window_log_likelihood
from .long_metric import LongMetric, compute_log_distributions, get_log_distributions

__all__ = ['LongMetric', 'compute_log_distributions', 'get_log_distributions', 'compute_scenario_metrics', 'window_log_likelihood', 'MetricFeatures', 'ObjectTrajectories', 'ScenarioRollouts', 'compute_metric_features', 'format_rollouts',
           'get_scenario_id_int_tensor', 'output_to_rollouts', 'compute_distance_to_nearest_object', 'compute_time_to_collision_with_object_in_front',
           'compute_kinematic_features', 'compute_num_placement', 'compute_distance_placement',
           'compute_distance_to_road_edge', 'tensorize_polylines']
