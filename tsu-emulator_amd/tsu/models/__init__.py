"""Energy-based models (reference: tsu/models/__init__.py:11-13) plus the README's IsingModel2D facade."""
from .ising import (IsingChain, IsingConfig, IsingGrid, IsingModel, IsingModel2D, demonstrate_phase_transition,
                    temperature_scan)

__all__ = ["IsingModel", "IsingChain", "IsingGrid", "IsingModel2D", "IsingConfig", "demonstrate_phase_transition", "temperature_scan"]
