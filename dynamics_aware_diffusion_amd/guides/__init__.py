from .policies import DynamicsAwarePolicy, GuidedPolicy, MPCPolicy, ValueGuidedPolicy

__all__ = ["GuidedPolicy", "MPCPolicy", "ValueGuidedPolicy", "DynamicsAwarePolicy"]
