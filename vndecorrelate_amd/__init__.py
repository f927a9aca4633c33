"""vndecorrelate_amd - MI355X-native velvet-noise decorrelator.

Drop-in for the velvet-noise path of ckonst/VNDecorrelate: import
``vndecorrelate_amd.decorrelation`` where you imported
``vndecorrelate.decorrelation``.  The tap sum runs in hand-written HIP kernels
for gfx950 behind the C ABI of ``include/vnd_amd.h``; see DESIGN.md.
"""
__version__ = '0.1.0'

from .decorrelation import (  # noqa: F401
    MODE_EXACT,
    MODE_FAST,
    MODE_FMA,
    HaasEffect,
    SignalChain,
    VelvetNoise,
    WhiteNoise,
    convolve_velvet_noise,
    convolve_velvet_noise_bank,
    convolve_velvet_noise_batched,
    decorrelate_bank,
    generate_velvet_noise,
    set_default_mode,
    set_device_epilogue,
)
