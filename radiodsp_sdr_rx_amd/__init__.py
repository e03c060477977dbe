"""MI355X-native many-channel SDR receive chain (per-block IQ path of
gcallipo/RadioDSP_SDR_RX): HIP kernels behind a C-ABI (include/rdsp.h)."""
from ._lib import ChainConfig, RdspError, SynthConfig, load  # noqa: F401
from .config import AGC, ALS, AUDIO_FILTER, DEMOD, K_CONFIGS, make_config, synth_config  # noqa: F401


def __getattr__(name):
    # torch is only needed by the device-side wrappers
    if name in ("Chain", "synth_iq", "calc_cplx_FIR_coeffs", "init_filter_mask"):
        from . import chain
        return getattr(chain, name)
    raise AttributeError(name)
