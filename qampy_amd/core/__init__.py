"""Core API (plain ndarrays in/out), mirror of ``qampy.core`` for the equaliser + BPS hot path."""
from . import equalisation, phaserecovery, ber_functions, filter, pilotbased_receiver  # noqa: F401
