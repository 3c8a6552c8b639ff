"""witch_amd: MI355X-native query-vs-eHMM scoring and alignment for WITCH."""
__version__ = "0.1.0"
