"""tsu -- MI355X-native drop-in for the stochastic spin-update hot path of tsu-emulator.

Scope: ``tsu.gibbs`` (GibbsSampler sweep), ``tsu.models.ising`` (IsingGrid /
"IsingModel2D" lattice sweep) and ``tsu.core`` (ThermalSamplingUnit Langevin step), with the reference's names,
signatures and error behaviour, executed by hand-written HIP kernels for gfx950 (``tsu/_lib/libtsu_hip.so``,
C ABI in ``include/tsu_hip.h``) through ctypes.  Of the reference's other sub-packages only the two benchmark suites that
time this path are mirrored (``tsu.benchmarks``: sampling, optimisation, runner); api, ml, visualization, demos and cli are
callers of this path and are not part of this package.
"""

__version__ = "0.1.0"

from .core import (  # noqa: F401
    ConfigurationError,
    ProbabilisticNeuron,
    QuadraticEnergy,
    QuadraticForm,
    SamplingError,
    ThermalSamplingUnit,
    TSUConfig,
    TSUError,
    validate_distribution,
)
from .core import ThermalSamplingUnit as TSU  # noqa: F401
from .gibbs import GibbsConfig, GibbsSampler, HardwareEmulator  # noqa: F401
from .models import (  # noqa: F401
    IsingChain,
    IsingGrid,
    IsingModel,
    IsingModel2D,
    demonstrate_phase_transition,
)

__all__ = [
    "ThermalSamplingUnit", "TSU", "TSUConfig", "ProbabilisticNeuron", "validate_distribution", "TSUError",
    "ConfigurationError", "SamplingError", "QuadraticEnergy", "QuadraticForm",
    "GibbsSampler", "GibbsConfig", "HardwareEmulator",
    "IsingModel", "IsingChain", "IsingGrid", "IsingModel2D", "demonstrate_phase_transition",
]
