"""tsu -- MI355X-native drop-in for the stochastic spin-update hot path of tsu-emulator."""
__version__ = "0.1.0"
