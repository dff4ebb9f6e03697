"""Core equaliser API on plain ndarrays (mirror of ``qampy.core.equalisation``, qampy/core/equalisation/__init__.py)."""
from .equalisation import (DATA_AIDED, DECISION_BASED, NONDECISION_BASED, REAL_VALUED, TRAINING_FCTS,  # noqa: F401
                           apply_filter, dual_mode_equalisation, equalise_signal, equalise_signal_windows,
                           generate_symbols_for_eq)
from . import hip_equalisation  # noqa: F401
