"""Chain configuration helpers: enums and the BASELINE.json configs K1..K5."""
from ._lib import ChainConfig, SynthConfig

DEMOD = {"IQ": 0, "USB": 1, "LSB": 2, "CW_USB": 3, "CW_LSB": 4, "AM": 5, "SAM": 6}
AGC = {"off": 0, "fast": 1, "medium": 2, "slow": 3}
ALS = {"off": 0, "notch": 1, "peak": 2}
AUDIO_FILTER = {"audioCW": 0, "audio2100": 1, "audio2700": 2, "audio3100": 3, "audioAM": 4,
                "audioWSPR": 5}

DEFAULTS = dict(
    fs_in=96000.0, decim=4, fir_taps=256, fir_cut_hz=10000.0, nco_hz=12000.0,
    fft_l=256, window=1, flo_hz=300.0, fhi_hz=2700.0, filter_on=1, demod="USB",
    spectral_nr=0, spectral_level=0.0, lms_nr=0, als_mode="off", als_strength=20,
    agc_mode="off", input_gain=1.0, output_gain=1.0, iq_balance=1.0, mute=0,
)

# BASELINE.json configs: chains per config
K_CONFIGS = {
    # 1 channel, 96 kHz IQ, 128-sample blocks, USB demod, NR/notch off
    "K1": dict(channels=1, cfg=dict(fft_l=256, demod="USB")),
    # 4096 channels, 256-tap polyphase /4 + USB demod
    "K2": dict(channels=4096, cfg=dict(fft_l=256, demod="USB")),
    # 4096 channels, SSB + 512-pt spectral NR + LMS auto-notch + AGC
    "K3": dict(channels=4096, cfg=dict(fft_l=512, demod="USB", spectral_nr=1, spectral_level=2.0,
                                       als_mode="notch", als_strength=20, agc_mode="medium",
                                       output_gain=0.5)),
    # 8192 channels, CW, 2048-tap overlap-save narrow filter (FFT_L 4096) + AGC
    "K4": dict(channels=8192, cfg=dict(fft_l=4096, demod="CW_USB", flo_hz=450.0, fhi_hz=950.0,
                                       agc_mode="fast", output_gain=0.5), cw=True),
    # 8192 channels per GPU, full chain (scaling curve)
    "K5": dict(channels=8192, cfg=dict(fft_l=512, demod="USB", spectral_nr=1, spectral_level=2.0,
                                       als_mode="notch", als_strength=20, agc_mode="medium",
                                       output_gain=0.5)),
}


def make_config(struct_cls=ChainConfig, **kw):
    d = dict(DEFAULTS)
    d.update(kw)
    if isinstance(d["demod"], str):
        d["demod"] = DEMOD[d["demod"]]
    if isinstance(d["agc_mode"], str):
        d["agc_mode"] = AGC[d["agc_mode"]]
    if isinstance(d["als_mode"], str):
        d["als_mode"] = ALS[d["als_mode"]]
    s = struct_cls()
    for name, _ in struct_cls._fields_:
        setattr(s, name, d[name])
    return s


def synth_config(cw=False, fs=96000.0, f_off=12000.0):
    s = SynthConfig()
    s.fs, s.f_off, s.cw = fs, f_off, 1 if cw else 0
    s.amp_tone, s.amp_carrier, s.sigma = 0.20, 0.30, 0.05
    return s
