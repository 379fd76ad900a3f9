from .data_driven import (fit_linear_dynamics, identify_dynamics_from_arrays,
                          identify_dynamics_from_data, transitions_from_episodes)
from .projection import ProjectionMatrixBuilder
from .registry import (DYNAMICS_REGISTRY, STATE_DIM_REGISTRY, double_integrator,
                       get_dynamics_for_env, state_dim_for_env)

__all__ = ["ProjectionMatrixBuilder", "fit_linear_dynamics", "identify_dynamics_from_arrays",
           "identify_dynamics_from_data", "transitions_from_episodes", "get_dynamics_for_env",
           "state_dim_for_env", "double_integrator", "DYNAMICS_REGISTRY", "STATE_DIM_REGISTRY"]
