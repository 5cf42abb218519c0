"""monohair_amd -- MI355X-native PMVO hot path of MonoHair."""
__version__ = "0.1.0"
