"""ctypes loader for librdsp_hip.so (the C-ABI of include/rdsp.h).

The library is built in-tree by __graft_entry__.build() / csrc/Makefile.  There
is no Python or CPU fallback: if the shared object is missing, loading fails
loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree build, and nothing else: no environment variable can put another library under the tests or
# under a caller.  Measurement harnesses that compare builds call use_library() themselves (bench.py --lib).
LIB_PATH = os.path.join(_HERE, "librdsp_hip.so")

RDSP_OK = 0
ERRORS = {-1: "INVALID", -2: "NO_DEVICE", -3: "HIP", -4: "NOT_READY", -5: "UNSUPPORTED", -6: "NOMEM"}


class RdspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rdsp error {code} ({ERRORS.get(code, '?')}): {msg}")
        self.code = code


class ChainConfig(C.Structure):
    """rdsp_chain_config_t (include/rdsp.h)."""
    _fields_ = [
        ("fs_in", C.c_double),
        ("decim", C.c_int32),
        ("fir_taps", C.c_int32),
        ("fir_cut_hz", C.c_double),
        ("nco_hz", C.c_double),
        ("fft_l", C.c_int32),
        ("window", C.c_int32),
        ("flo_hz", C.c_double),
        ("fhi_hz", C.c_double),
        ("filter_on", C.c_int32),
        ("demod", C.c_int32),
        ("spectral_nr", C.c_int32),
        ("spectral_level", C.c_float),
        ("lms_nr", C.c_int32),
        ("als_mode", C.c_int32),
        ("als_strength", C.c_int32),
        ("agc_mode", C.c_int32),
        ("input_gain", C.c_float),
        ("output_gain", C.c_float),
        ("iq_balance", C.c_float),
        ("mute", C.c_int32),
    ]


class SynthConfig(C.Structure):
    """rdsp_synth_config_t (include/rdsp.h)."""
    _fields_ = [
        ("fs", C.c_double),
        ("f_off", C.c_double),
        ("cw", C.c_int32),
        ("amp_tone", C.c_double),
        ("amp_carrier", C.c_double),
        ("sigma", C.c_double),
    ]


class StreamStats(C.Structure):
    """rdsp_stream_stats_t (include/rdsp.h)."""
    _fields_ = [("blocks", C.c_int64), ("samples_in", C.c_int64), ("samples_out", C.c_int64),
                ("seconds", C.c_double), ("read_seconds", C.c_double), ("write_seconds", C.c_double)]


_lib = None
SOURCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int16), C.c_size_t, C.c_int)  # rdsp_source_fn
SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int16), C.c_size_t, C.c_int)    # rdsp_sink_fn

UPDATE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)  # rdsp_update_fn

# every symbol include/rdsp.h declares: (name, restype, argtypes)
_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_f32p, _f64p, _i16p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int16)
SYMBOLS = [
    ("rdsp_last_error", C.c_char_p, []),
    ("rdsp_version", C.c_char_p, []),
    ("rdsp_experimental_build", _i, []),
    ("rdsp_chain_front_kernel_name", C.c_char_p, [_vp]),
    ("rdsp_device_count", _i, []),
    ("rdsp_calc_cplx_FIR_coeffs", None, [_f64p, _f64p, _i, _d, _d, _d, _i]),
    ("rdsp_init_filter_mask", _i, [_f32p, _f64p, _f64p, _i]),
    ("rdsp_chain_create", _i, [C.POINTER(ChainConfig), _i, _i, _i, C.POINTER(_vp)]),
    ("rdsp_chain_destroy", None, [_vp]),
    ("rdsp_chain_channels", _i, [_vp]),
    ("rdsp_chain_granule_blocks", _i, [_vp]),
    ("rdsp_chain_call_unit_blocks", _i, [_vp]),
    ("rdsp_set_spectral_resynthesis", _i, [_vp, _i]),
    ("rdsp_set_nlms_energy_mode", _i, [_vp, _i]),
    ("rdsp_chain_reset", _i, [_vp, _vp]),
    ("rdsp_doConvolutionalInitialize", _i, [_vp, _vp]),
    ("rdsp_reInitializeFilter", _i, [_vp, _d, _d, _vp]),
    ("rdsp_Init_LMS_NR", _i, [_vp, _i, _vp]),
    ("rdsp_LMS_NoiseReduction", _i, [_vp, _i, _vp, _sz, _vp]),
    ("rdsp_chain_process", _i, [_vp, _vp, _sz, _i, _vp, _sz, _vp, _vp]),
    ("rdsp_doConvolutionalProcessing", _i, [_vp, _f, _i, _d, _d, _vp, _sz, _i, _vp, _sz, _vp]),
    ("rdsp_q15_to_float", _i, [_vp, _vp, _sz, _vp]),
    ("rdsp_float_to_q15", _i, [_vp, _vp, _sz, _vp]),
    ("rdsp_sdr_enableAGC", _i, [_vp]),
    ("rdsp_sdr_disableAGC", _i, [_vp]),
    ("rdsp_sdr_setAGCmode", _i, [_vp, _i]),
    ("rdsp_sdr_enableALSfilter", _i, [_vp]),
    ("rdsp_sdr_disableALSfilter", _i, [_vp]),
    ("rdsp_sdr_setALSfilterNotch", _i, [_vp]),
    ("rdsp_sdr_setALSfilterPeak", _i, [_vp]),
    ("rdsp_sdr_setALSfilterAdaptive", _i, [_vp]),
    ("rdsp_sdr_enableNoiseBlanker", _i, [_vp]),
    ("rdsp_sdr_disableNoiseBlanker", _i, [_vp]),
    ("rdsp_sdr_setNoiseBlankerThresholdDb", _i, [_vp, _f]),
    ("rdsp_pre_swapIQ", _i, [_vp, _i]),
    ("rdsp_pre_startAutoI2SerrorDetection", _i, [_vp]),
    ("rdsp_pre_setIQslip", _i, [_vp, _i]),
    ("rdsp_estimate_iq_slip", _i, [C.POINTER(C.c_int16), C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("rdsp_sdr_setInputGain", _i, [_vp, _f]),
    ("rdsp_sdr_setOutputGain", _i, [_vp, _f]),
    ("rdsp_sdr_setIQgainBalance", _i, [_vp, _f]),
    ("rdsp_sdr_enableAudioFilter", _i, [_vp]),
    ("rdsp_sdr_setAudioFilter", _i, [_vp, _i, _vp]),
    ("rdsp_sdr_setDemodMode", C.c_uint32, [_vp, _i, _vp]),
    ("rdsp_sdr_setMute", _i, [_vp, _i]),
    ("rdsp_sdr_setTuningOffsetHz", _i, [_vp, _d]),
    ("rdsp_set_nr_level", _i, [_vp, _i]),
    ("rdsp_set_spectral_nr", _i, [_vp, _i, _f]),
    ("rdsp_chain_set_pipelined", _i, [_vp, _i]),
    ("rdsp_chain_flush", _i, [_vp, _vp]),
    ("rdsp_chain_set_front_variant", _i, [_vp, _i]),
    ("rdsp_chain_set_sub_batch", _i, [_vp, _i]),
    ("rdsp_chain_set_timing", _i, [_vp, _i]),
    ("rdsp_chain_get_timing", _i, [_vp, _f64p, _f64p, C.POINTER(C.c_int)]),
    ("rdsp_chain_get_timing_span", _i, [_vp, _f64p, C.POINTER(C.c_int)]),
    ("rdsp_chain_get_scalars", _i, [_vp, _f32p, _vp]),
    ("rdsp_chain_device", _i, [_vp]),
    ("rdsp_spectrum_device", _i, [_vp]),
    ("rdsp_chain_state_bytes", C.c_size_t, [_vp, _i]),
    ("rdsp_chain_save_state", _i, [_vp, _i, _i, _vp, C.c_size_t, _vp]),
    ("rdsp_chain_load_state", _i, [_vp, _i, _vp, C.c_size_t, _vp]),
    ("rdsp_chain_get_lms_coeffs", _i, [_vp, _i, _f32p, _vp]),
    ("rdsp_chain_get_status", _i, [_vp, C.POINTER(C.c_uint32), _vp]),
    ("rdsp_chain_reset_nlms_channels", _i, [_vp, _i, _i, _i, _vp]),
    ("rdsp_chain_get_mask", _i, [_vp, _f32p]),
    ("rdsp_iq_reader_open", _i, [C.c_char_p, _i, C.POINTER(_vp)]),
    ("rdsp_iq_reader_sample_rate", _d, [_vp]),
    ("rdsp_iq_reader_frames", C.c_int64, [_vp]),
    ("rdsp_iq_reader_format", _i, [_vp]),
    ("rdsp_iq_reader_read", _sz, [_vp, _i16p, _sz]),
    ("rdsp_iq_reader_close", None, [_vp]),
    ("rdsp_audio_writer_open", _i, [C.c_char_p, _i, _d, C.POINTER(_vp)]),
    ("rdsp_audio_writer_write", _sz, [_vp, _i16p, _sz]),
    ("rdsp_audio_writer_frames", C.c_int64, [_vp]),
    ("rdsp_audio_writer_close", _i, [_vp]),
    ("rdsp_stream_run", _i, [_vp, SOURCE_FN, _vp, SINK_FN, _vp, _i, C.c_int64, C.POINTER(StreamStats)]),
    ("rdsp_stream_run_files", _i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i, C.c_int64, C.POINTER(StreamStats)]),
    ("rdsp_stream_run_memory", _i, [_vp, _i16p, _sz, C.c_int64, _i16p, _sz, _i, C.POINTER(StreamStats)]),
    ("rdsp_spectrum_node_create", _vp, [_vp, _vp]),
    ("rdsp_spectrum_node_available", _i, [_vp]),
    ("rdsp_spectrum_node_output", C.POINTER(C.c_uint16), [_vp]),
    ("rdsp_spectrum_node_status", _i, [_vp]),
    ("rdsp_chain_set_tail_variant", _i, [_vp, _i, _i]),
    ("rdsp_chain_set_fir_variant", _i, [_vp, _i]),
    ("rdsp_chain_set_priorities", _i, [_vp, _i, _i]),
    ("rdsp_chain_set_groups", _i, [_vp, _i, C.POINTER(C.c_uint16)]),
    ("rdsp_chain_groups", _i, [_vp]),
    ("rdsp_group_reInitializeFilter", _i, [_vp, _i, _d, _d, _vp]),
    ("rdsp_group_setAudioFilter", _i, [_vp, _i, _i, _vp]),
    ("rdsp_group_setDemodMode", C.c_uint32, [_vp, _i, _i, _vp]),
    ("rdsp_group_setTuningOffsetHz", _i, [_vp, _i, _d]),
    ("rdsp_group_get_mask", _i, [_vp, _i, _f32p]),
    ("rdsp_pbt_step", _i, [_f64p, _f64p, _i, _i]),
    ("rdsp_group_pbt", _i, [_vp, _i, _i, _i, _vp]),
    ("rdsp_group_tuningMode", C.c_uint32, [_vp, _i, _i, _d, _vp]),
    ("rdsp_chain_get_fir_taps", _i, [_vp, _f32p]),
    ("rdsp_graph_create", _vp, [_i]),
    ("rdsp_graph_destroy", None, [_vp]),
    ("rdsp_graph_channels", _i, [_vp]),
    ("rdsp_memory", _i, [_vp, _i]),
    ("rdsp_memory_usage", _i, [_vp]),
    ("rdsp_memory_usage_max", _i, [_vp]),
    ("rdsp_node_create", _vp, [_vp, _i, UPDATE_FN, _vp]),
    ("rdsp_node_set_destructor", None, [_vp, _vp]),
    ("rdsp_node_user", _vp, [_vp]),
    ("rdsp_node_graph", _vp, [_vp]),
    ("rdsp_connect", _i, [_vp, _i, _vp, _i]),
    ("rdsp_update_all", _i, [_vp]),
    ("rdsp_no_interrupts", None, [_vp]),
    ("rdsp_interrupts", None, [_vp]),
    ("rdsp_allocate", _vp, [_vp]),
    ("rdsp_receive_readonly", _vp, [_vp, _i]),
    ("rdsp_receive_writable", _vp, [_vp, _i]),
    ("rdsp_transmit", None, [_vp, _vp, _i]),
    ("rdsp_release", None, [_vp]),
    ("rdsp_block_data", _i16p, [_vp]),
    ("rdsp_block_refcount", _i, [_vp]),
    ("rdsp_record_queue_create", _vp, [_vp]),
    ("rdsp_record_queue_begin", None, [_vp]),
    ("rdsp_record_queue_end", None, [_vp]),
    ("rdsp_record_queue_available", _i, [_vp]),
    ("rdsp_record_queue_readBuffer", _i16p, [_vp]),
    ("rdsp_record_queue_freeBuffer", None, [_vp]),
    ("rdsp_play_queue_create", _vp, [_vp]),
    ("rdsp_play_queue_getBuffer", _i16p, [_vp]),
    ("rdsp_play_queue_playBuffer", _i, [_vp]),
    ("rdsp_input_node_create", _vp, [_vp]),
    ("rdsp_input_node_push", _i, [_vp, _i16p, _i16p]),
    ("rdsp_sdr_node_create", _vp, [_vp, _vp]),
    ("rdsp_sdr_node_status", _i, [_vp]),
    ("rdsp_chain_decim", _i, [_vp]),
    ("rdsp_window_q15", None, [_i, _i16p]),
    ("rdsp_biquad_design", None, [_i, _d, _d, _d, _f32p]),
    ("rdsp_design_audio_iir", None, [_d, _d, _d, _f32p]),
    ("rdsp_biquad_create", _i, [_i, _i, _d, C.POINTER(_vp)]),
    ("rdsp_biquad_destroy", None, [_vp]),
    ("rdsp_biquad_setCoefficients", _i, [_vp, _i, _f64p]),
    ("rdsp_biquad_setLowpass", _i, [_vp, _i, _f, _f]),
    ("rdsp_biquad_setHighpass", _i, [_vp, _i, _f, _f]),
    ("rdsp_biquad_setBandpass", _i, [_vp, _i, _f, _f]),
    ("rdsp_biquad_setNotch", _i, [_vp, _i, _f, _f]),
    ("rdsp_biquad_get_coeffs", _i, [_vp, _f32p]),
    ("rdsp_biquad_get_definition", _i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    ("rdsp_biquad_setCoefficients_int", _i, [_vp, _i, C.POINTER(C.c_int32)]),
    ("rdsp_teensy_biquad_design", None, [_i, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]),
    ("rdsp_biquad_update", _i, [_vp, _vp, _sz, _i, _i, _vp, _sz, _i, _vp]),
    ("rdsp_biquad_node_create", _vp, [_vp, _vp]),
    ("rdsp_biquad_node_status", _i, [_vp]),
    ("rdsp_sdr_setAudioFilterKind", _i, [_vp, _i, _vp]),
    ("rdsp_chain_get_iir_coeffs", _i, [_vp, _i, _f32p]),
    ("rdsp_group_setAudioIIRCoefficients", _i, [_vp, _i, _f32p]),
    ("rdsp_sdr_setAudioIIRCoefficients", _i, [_vp, _f32p]),
    ("rdsp_window_q15_n", None, [_i, _i, _i16p]),
    ("rdsp_fft1024_create", _i, [_i, _i, _i, C.POINTER(_vp)]),
    ("rdsp_fft1024_destroy", None, [_vp]),
    ("rdsp_fft1024_windowFunction", _i, [_vp, _i]),
    ("rdsp_fft1024_averageTogether", _i, [_vp, _i]),
    ("rdsp_fft1024_outputs_for", _i, [_vp, _i]),
    ("rdsp_fft1024_update", _i, [_vp, _vp, _sz, _i, _i, _vp, _sz, C.POINTER(C.c_int), _vp]),
    ("rdsp_fft1024_node_create", _vp, [_vp, _vp]),
    ("rdsp_fft1024_node_available", _i, [_vp]),
    ("rdsp_fft1024_node_output", C.POINTER(C.c_uint16), [_vp]),
    ("rdsp_fft1024_node_status", _i, [_vp]),
    ("rdsp_spectrum_create", _i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    ("rdsp_spectrum_destroy", None, [_vp]),
    ("rdsp_spectrum_averageTogether", _i, [_vp, _i]),
    ("rdsp_spectrum_windowFunction", _i, [_vp, _i]),
    ("rdsp_spectrum_outputs_for", _i, [_vp, _i]),
    ("rdsp_spectrum_update", _i, [_vp, _vp, _sz, _i, _vp, _sz, C.POINTER(C.c_int), _vp]),
    ("rdsp_spectrum_create_default", _i, [_i, _i, C.POINTER(_vp)]),
    ("rdsp_spectrum_windowFunction_table", _i, [_vp, _i16p]),
    ("rdsp_spectrum_read", _f, [C.POINTER(C.c_uint16), C.c_uint]),
    ("rdsp_spectrum_read_range", _f, [C.POINTER(C.c_uint16), C.c_uint, C.c_uint]),
    ("rdsp_spectrum_node_read", _f, [_vp, _i, C.c_uint]),
    ("rdsp_spectrum_node_read_range", _f, [_vp, _i, C.c_uint, C.c_uint]),
    ("rdsp_sqrt_uint32_approx", C.c_uint32, [C.c_uint32]),
    ("rdsp_fft1024_windowFunction_table", _i, [_vp, _i16p]),
    ("rdsp_fft1024_read", _f, [C.POINTER(C.c_uint16), C.c_uint]),
    ("rdsp_fft1024_read_range", _f, [C.POINTER(C.c_uint16), C.c_uint, C.c_uint]),
    ("rdsp_fft1024_node_read", _f, [_vp, _i, C.c_uint]),
    ("rdsp_fft1024_node_read_range", _f, [_vp, _i, C.c_uint, C.c_uint]),
    ("rdsp_engine_create", _i, [_i, _i, _i, C.POINTER(_vp)]),
    ("rdsp_engine_destroy", None, [_vp]),
    ("rdsp_engine_reset", _i, [_vp, _vp]),
    ("rdsp_engine_load_tables", _i, [_vp, _f32p, _f32p]),
    ("rdsp_engine_enableAGC", _i, [_vp]),
    ("rdsp_engine_setAGCmode", _i, [_vp, _i]),
    ("rdsp_engine_enableALSfilter", _i, [_vp]),
    ("rdsp_engine_disableALSfilter", _i, [_vp]),
    ("rdsp_engine_setALSfilterNotch", _i, [_vp]),
    ("rdsp_engine_setALSfilterPeak", _i, [_vp]),
    ("rdsp_engine_setALSfilterAdaptive", _i, [_vp]),
    ("rdsp_engine_enableNoiseBlanker", _i, [_vp]),
    ("rdsp_engine_disableNoiseBlanker", _i, [_vp]),
    ("rdsp_engine_setInputGain", _i, [_vp, _f]),
    ("rdsp_engine_setOutputGain", _i, [_vp, _f]),
    ("rdsp_engine_setIQgainBalance", _i, [_vp, _f]),
    ("rdsp_engine_enableAudioFilter", _i, [_vp]),
    ("rdsp_engine_setAudioFilter", _i, [_vp, _i]),
    ("rdsp_engine_setDemodMode", _f, [_vp, _i]),
    ("rdsp_engine_setMute", _i, [_vp, _i]),
    ("rdsp_engine_update", _i, [_vp, _vp, _sz, _i, _vp, _sz, _vp]),
    ("rdsp_engine_set_groups", _i, [_vp, _i, C.POINTER(C.c_int)]),
    ("rdsp_engine_groups", _i, [_vp]),
    ("rdsp_engine_select_group", _i, [_vp, _i]),
    ("rdsp_engine_state_bytes", C.c_size_t, [_vp, _i]),
    ("rdsp_engine_save_state", _i, [_vp, _i, _i, _vp, C.c_size_t, _vp]),
    ("rdsp_engine_load_state", _i, [_vp, _i, _vp, C.c_size_t, _vp]),
    ("rdsp_engine_channels", _i, [_vp]),
    ("rdsp_engine_device", _i, [_vp]),
    ("rdsp_engine_max_blocks", _i, [_vp]),
    ("rdsp_engine_get_scalars", _i, [_vp, _f32p, _vp]),
    ("rdsp_engine_agc_curve", _f32p, [_vp]),
    ("rdsp_engine_sine_table", _f32p, [_vp]),
    ("rdsp_preproc_create", _i, [_i, _i, C.POINTER(_vp)]),
    ("rdsp_preproc_destroy", None, [_vp]),
    ("rdsp_preproc_startAutoI2SerrorDetection", _i, [_vp]),
    ("rdsp_preproc_swapIQ", _i, [_vp, _i]),
    ("rdsp_preproc_update", _i, [_vp, _vp, _sz, _i, _vp, _sz, _vp]),
    ("rdsp_preproc_get_state", _i, [_vp, _i16p, _vp]),
    ("rdsp_preproc_channels", _i, [_vp]),
    ("rdsp_preproc_device", _i, [_vp]),
    ("rdsp_preproc_node_create", _vp, [_vp, _vp]),
    ("rdsp_engine_node_create", _vp, [_vp, _vp]),
    ("rdsp_engine_node_status", _i, [_vp]),
    ("rdsp_sdr_set_engine_literal", _i, [_vp, _i]),
    ("rdsp_sdr_load_engine_tables", _i, [_vp, _f32p, _f32p]),
    ("rdsp_chain_engine", _vp, [_vp]),
    ("rdsp_chain_preproc", _vp, [_vp]),
    ("rdsp_synth_iq", None, [_i16p, _i, _i, C.c_uint64, _i, C.POINTER(SynthConfig), _i]),
]


def use_library(path):
    """A/B harness only (bench.py --lib): load another build of the same library instead of the in-tree one.
    Must be called before the first load()."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() after the library was loaded")
    LIB_PATH = os.path.abspath(path)


def load():
    """Load librdsp_hip.so and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: torch ships its own libamdhip64.so.7.  Importing
    # torch first makes the loader hand that same runtime to librdsp_hip.so (same
    # SONAME); two runtimes in one process see no devices in the second one.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure-C / numpy-only users fall back to the system ROCm runtime
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != RDSP_OK:
        raise RdspError(rc, load().rdsp_last_error().decode())
    return rc
