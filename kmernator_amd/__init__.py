"""MI355X-native k-mer spectrum builder (Kmernator FilterReads / KmerSpectrum hot path).

Everything that computes lives in csrc/ (hand-written HIP for gfx950 behind the C-ABI of
include/kmernator_amd.h); this package is the thin host-side mirror of the reference's
KmerSpectrum interface plus the one-process-per-GPU owner-partitioned driver.
"""
from ._lib import (KMR_MAP_SINGLETON, KMR_MAP_WEAK, KMR_VALUE_COUNT_DIR, KMR_VALUE_EXT, KmrConfig, default_config, load,
                   record_bytes)
from .spectrum import FilterKnownOddities, KmerSpectrum, KmerSpectrumError, Histogram, ReadSet, synth_reads_device

__all__ = ["KmerSpectrum", "KmerSpectrumError", "ReadSet", "Histogram", "FilterKnownOddities", "synth_reads_device", "KmrConfig", "default_config", "load", "record_bytes",
           "KMR_MAP_WEAK", "KMR_MAP_SINGLETON", "KMR_VALUE_COUNT_DIR", "KMR_VALUE_EXT"]
