"""modaltune_amd: MI355X-native Modal-Adapter fine-tuning hot path."""
__version__ = "0.1.0"
