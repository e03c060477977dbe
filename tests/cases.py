"""Shared test cases: chain configurations (BASELINE.json K1..K4 at test sizes) and
the golden-vector manifest."""
K1 = dict(fft_l=256, demod="USB")                       # K1/K2 chain
K3 = dict(fft_l=512, demod="USB", spectral_nr=1, spectral_level=2.0, als_mode="notch",
          als_strength=20, agc_mode="medium", output_gain=0.5)
K4 = dict(fft_l=4096, demod="CW_USB", flo_hz=450.0, fhi_hz=950.0, agc_mode="fast", output_gain=0.5)
CONV_LITERAL = dict(fs_in=44100.0, decim=1, nco_hz=0.0, fft_l=256, flo_hz=300.0, fhi_hz=4000.0,
                    demod="IQ")                         # the in-tree CONV stage at its native rate: SAMPLE_RATE =
# (double)AUDIO_SAMPLE_RATE_EXACT (CONV:35) is 44100.0 in the reference's firmware image (tests/golden/firmware_tables.npz
# `sample_rate`; the Teensy 4 cores define 44100.0f -- 44117.64706, which rounds 1-4 assumed, is the Teensy 3 value)

GOLDEN_CASES = {
    "conv_literal_256": dict(channels=2, blocks=16, cfg=CONV_LITERAL),
    "k2_usb_256": dict(channels=3, blocks=32, cfg=K1),
    "k3_full_512": dict(channels=3, blocks=64, cfg=K3),
    "k4_cw_4096": dict(channels=2, blocks=128, cfg=K4, cw=True),
    "nr_lms_30": dict(channels=2, blocks=32, cfg=dict(fft_l=256, demod="USB", lms_nr=30)),
    "am_agc_512": dict(channels=2, blocks=32,
                       cfg=dict(fft_l=512, demod="AM", flo_hz=-3900.0, fhi_hz=3900.0, agc_mode="slow")),
}

# SURVEY 8f row F3 (engine features, build-defined): SAM demodulator; noise blanker + IQ swap on
# an input with ignition-style bursts.  `setup` is applied to the oracle chain and to the GPU chain.
GOLDEN_CASES["sam_agc_512"] = dict(channels=2, blocks=48,
                                   cfg=dict(fft_l=512, demod="SAM", flo_hz=-3900.0, fhi_hz=3900.0,
                                            nco_hz=12950.0, agc_mode="slow"))   # the 13 kHz carrier sits 50 Hz off tune
GOLDEN_CASES["blanker_swap_256"] = dict(channels=2, blocks=32, cfg=K1, impulses=True,
                                        setup=dict(noise_blanker_db=8.0, swap_iq=True))
# the older spectral-NR variant of the A6 row (BK_INO:1586-1630) and a non-default design window
# (CONV:159-179, id 3) on the negative side band with the 1024-point filter
GOLDEN_CASES["spectral_old_256"] = dict(channels=2, blocks=32,
                                        cfg=dict(fft_l=256, demod="USB", spectral_nr=2, agc_mode="fast", output_gain=0.5))
GOLDEN_CASES["window3_lsb_1024"] = dict(channels=2, blocks=64,
                                        cfg=dict(fft_l=1024, demod="LSB", flo_hz=-2700.0, fhi_hz=-300.0, window=3))


def add_impulses(iq, every=1777, burst=3):
    """deterministic full-scale bursts (golden inputs): samples n*every .. +burst of every channel"""
    out = iq.copy()
    for c in range(iq.shape[0]):
        for pos in range(3000 + 97 * c, iq.shape[1] - burst, every):
            out[c, pos:pos + burst, 0] = 30000 if (pos // every) % 2 else -30000
            out[c, pos:pos + burst, 1] = -30000 if (pos // every) % 3 else 30000
    return out


def apply_setup(chain, setup, oracle=False):
    """engine setters that are not part of the config struct, on an OracleChain or a GPU Chain"""
    if not setup:
        return
    if "noise_blanker_db" in setup:
        if oracle:
            chain.set_noise_blanker(True, setup["noise_blanker_db"])
        else:
            chain.enableNoiseBlanker()
            chain.setNoiseBlankerThresholdDb(setup["noise_blanker_db"])
    if setup.get("swap_iq"):
        chain.set_swap_iq(True) if oracle else chain.swapIQ(True)


# feed-forward chains: the north-star tolerance (normwise, per channel)
TOL = 1e-5
