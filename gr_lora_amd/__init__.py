"""gr_lora_amd -- MI355X-native LoRa PHY demodulator behind the gr-lora block API.

Importing the package is cheap; the HIP library (liblora_hip.so) is loaded on
first use of a decoder and its absence is a hard error (no CPU fallback).
"""
__all__ = ["synth"]
