"""Callers of the hot path ("next" rows N1/N2 of SURVEY.md 8f): trainers and evaluators with the reference's class
names, constructor signatures and observable behaviour (engine/*.py).  Pure host orchestration; all model arithmetic
stays in the HIP kernels behind ``build_model``."""
from .evaluators import InteractiveEvaluator, RandomPolicyEvaluator  # noqa: F401
from .trainers import DirectSupervisionTrainer, InteractronRandomTrainer, InteractronTrainer  # noqa: F401
