#!/usr/bin/env python3
"""Constant tables of the reference's own shipped firmware, as test fixtures.

Runs in the build container only (the reference tree does not travel).  Reads
/root/reference/pre_compiled/RadioDSP_SDR_RX.ino.hex with a small Intel-HEX reader, cuts out the
read-only tables named below by their offsets in the flat image (first byte of the image = 0) and
writes tests/golden/firmware_tables.npz.  Data only: no code bytes, never the hex itself.

These are the only reference-held numbers that can pin anything in this build (SURVEY Appendix C,
DESIGN.md section 2):

  hann256               AudioWindowHanning256,           FFT.windowFunction(...)       INO:144
  hann1024              AudioWindowHanning1024,          AudioFFT.windowFunction(...)  INO:147
  blackman_nuttall256   AudioWindowBlackmanNuttall256,   constructor default           FFTIQ.h:56
  sqrt_guess            sqrt_integer_guess_table[33]     sqrt_uint32_approx            FFTIQ.cpp:105
  twiddle_q15_4096      CMSIS twiddleCoef_4096_q15 (cos, sin pairs; 3/4 of a turn)     FFTIQ.cpp:82
  bitrev_1024           CMSIS armBitRevTable[1024]                                      FFTIQ.cpp:82
  twiddle_f32_256/_128  CMSIS twiddleCoef_256 / _128 (arm_cfft_f32, CONV:291,309)
  biquad_sets           15 x {b0,b1,b2,a1,a2} x 4: the engine's IIR audio filters       CTL:153-177
  hilbert_half64        64 odd taps of one side of the engine's Hilbert transformer
  sine257               sin(2 pi k / 256), k = 0..256: the engine's oscillator table
  sample_rate           the double next to the 300.0 / 4000.0 filter edges in the initialised data: 44100.0 --
                        SAMPLE_RATE = (double)AUDIO_SAMPLE_RATE_EXACT (CONV:35) as this build had it (Teensy 4
                        cores define 44100.0f; 44117.64706 is the Teensy 3 value)
  lms_epsilon           1.19209289e-7f in the literal pool of the NLMS code (arm_lms_norm_f32's energy floor, NR:73)
  nr_gain               the double 1.1 of CONV:334
  design_constants      the literal pool of calc_cplx_FIR_coeffs (CONV:127-185): pi, 0.01, 2 pi, 4 pi, 6 pi and the
                        cosine-sum window coefficients of ids 2, 1 and the default
  iq_gain_balance       1.020f of SDR.setIQgainBalance (INO:135)
  code_ops_names / code_ops_offsets
                        NOT data tables: where in the image's CODE a handful of Thumb-2 instruction classes occur -- the
                        DSP-extension parallel arithmetic (SHADD16, QADD16, QSUB16, SHSUB16, QASX, QSAX, SHASX, SHSAX), the
                        dual multiplies (SMUAD, SMUADX, SMUSD, SMUSDX), CLZ and UDIV -- as class names and offsets,
                        decoded here from their fixed bit patterns (no operands, no other instructions, no bytes).
                        Their ORDER is what the restatements of arm_radix4_butterfly_q15, of the analysers' update()
                        and of sqrt_uint32_approx are checked against (tests/test_firmware_tables.py): the image
                        cannot be run here, but the sequence of these operations in it can be read.
  code_pack_names / code_pack_offsets
                        the conversion routine behind CONV:346-347 (arm_float_to_q15): VMOV.F32 #0.5 / #-0.5, VMUL,
                        VCMP #0, VADD, VCVT.S32.F32 (toward zero) and SSAT #16 in the code region of the five SSAT #16
                        without a shift -- which of CMSIS' two variants the image holds (with or without
                        ARM_MATH_ROUNDING)
  code_vfp_names / code_vfp_offsets
                        likewise the single-precision VFP arithmetic classes (VMUL, VADD, VSUB, VDIV, VNMUL and the
                        multiply-accumulates VMLA / VMLS / VFMA / VFMS / VFNMA / VFNMS): whether the image's CMSIS
                        routines round products and sums separately (they do) and in which order arm_lms_norm_f32
                        updates its energy, takes the dot product, forms the step and updates the taps.

Each entry was located by its content (a symmetric 256-entry int16 table that starts 0, 5, 20, 45
is a Hann window whatever it is called); the names are those of the libraries' published headers.
"""
import os
import sys

import numpy as np

HEX = "/root/reference/pre_compiled/RadioDSP_SDR_RX.ino.hex"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "firmware_tables.npz")

# name: (image offset, count, dtype)
TABLES = {
    "blackman_nuttall256": (0x1E8F4, 256, "<i2"),
    "hann1024": (0x1EAF4, 1024, "<i2"),
    "hann256": (0x1F2F4, 256, "<i2"),
    "sqrt_guess": (0x1F558, 33, "<u2"),
    "twiddle_f32_256": (0x1F92C, 512, "<f4"),
    "twiddle_q15_4096": (0x2012C, 6144, "<i2"),
    "bitrev_1024": (0x2312C, 1024, "<u2"),
    "twiddle_f32_128": (0x2392C, 256, "<f4"),
    "biquad_sets": (0x1DCBC, 15 * 20, "<f4"),
    "hilbert_half64": (0x1DCBC + 4 * 300, 64, "<f4"),
    "sine257": (0x1DCBC + 4 * 364, 257, "<f4"),
    "sample_rate": (0x24B04, 1, "<f8"),
    "lms_epsilon": (0x14998, 1, "<f4"),
    "nr_gain": (0x9430, 1, "<f8"),
    "design_constants": (0x90B0, 15, "<f8"),
    "iq_gain_balance": (0xB208, 1, "<f4"),
}


def read_ihex(path):
    """Intel HEX -> (lowest address, flat bytes).  Record types 00 data, 01 end, 02/04 base."""
    mem = {}
    base = 0
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line.startswith(":"):
                continue
            rec = bytes.fromhex(line[1:])
            if sum(rec) & 0xFF:
                raise ValueError("checksum: " + line)
            n, addr, typ, data = rec[0], (rec[1] << 8) | rec[2], rec[3], rec[4:-1]
            assert len(data) == n
            if typ == 0:
                for i, b in enumerate(data):
                    mem[base + addr + i] = b
            elif typ == 2:
                base = ((data[0] << 8) | data[1]) << 4
            elif typ == 4:
                base = ((data[0] << 8) | data[1]) << 16
            elif typ == 1:
                break
    lo, hi = min(mem), max(mem)
    img = bytearray(hi - lo + 1)
    for a, b in mem.items():
        img[a - lo] = b
    return lo, bytes(img)


def dsp_opcode_classes(img):
    """(offsets, names) of the instruction classes listed in the module docstring, in address order.  Thumb-2
    32-bit encodings, two little-endian halfwords: parallel add/sub 1111 1010 1ooo nnnn | 1111 dddd 0ppp mmmm
    (ooo: 001 ADD16, 101 SUB16, 010 ASX, 110 SAX; ppp: 001 saturating, 010 halving); SMUAD{X} 1111 1011 0010 nnnn
    | 1111 dddd 000x mmmm; SMUSD{X} ... 0100 ...; CLZ 1111 1010 1011 mmmm | 1111 dddd 1000 mmmm; UDIV 1111 1011 1011
    nnnn | 1111 dddd 1111 mmmm."""
    hw = np.frombuffer(img[:len(img) // 2 * 2], dtype="<u2").astype(np.int64)
    par = {0xFA90: "ADD16", 0xFAD0: "SUB16", 0xFAA0: "ASX", 0xFAE0: "SAX"}
    offs, names = [], []
    for i in range(len(hw) - 1):
        h1, h2 = int(hw[i]), int(hw[i + 1])
        if (h2 & 0xF000) != 0xF000:
            continue
        op, k, name = h1 & 0xFFF0, h2 & 0x00F0, None
        if op in par and k in (0x10, 0x20):
            name = ("Q" if k == 0x10 else "SH") + par[op]
        elif op == 0xFB20 and k in (0x00, 0x10):
            name = "SMUAD" + ("X" if k else "")
        elif op == 0xFB40 and k in (0x00, 0x10):
            name = "SMUSD" + ("X" if k else "")
        elif op == 0xFAB0 and k == 0x80 and (h1 & 0xF) == (h2 & 0xF):
            name = "CLZ"
        elif op == 0xFBB0 and k == 0xF0:
            name = "UDIV"
        if name:
            offs.append(2 * i)
            names.append(name)
    return np.array(offs, np.int64), np.array(names)


def vfp_f32_classes(img):
    """(offsets, names) of the single-precision VFP data-processing classes: 1110 1110 op.. | .... 1010 .o.0 ...."""
    hw = np.frombuffer(img[:len(img) // 2 * 2], dtype="<u2").astype(np.int64)
    table = {(0xEEA0, 0x00): "VFMA", (0xEEA0, 0x40): "VFMS", (0xEE90, 0x00): "VFNMS", (0xEE90, 0x40): "VFNMA",
             (0xEE00, 0x00): "VMLA", (0xEE00, 0x40): "VMLS", (0xEE20, 0x00): "VMUL", (0xEE20, 0x40): "VNMUL",
             (0xEE30, 0x00): "VADD", (0xEE30, 0x40): "VSUB", (0xEE80, 0x00): "VDIV"}
    offs, names = [], []
    for i in range(len(hw) - 1):
        h1, h2 = int(hw[i]), int(hw[i + 1])
        if (h2 & 0x0F00) != 0x0A00:
            continue
        name = table.get((h1 & 0xFFB0, h2 & 0x0050))
        if name:
            offs.append(2 * i)
            names.append(name)
    return np.array(offs, np.int64), np.array(names)


def pack_routine_classes(img):
    """The instruction classes of arm_float_to_q15 around the image's `SSAT Rd, #16, Rn` (no shift) instructions."""
    hw = np.frombuffer(img[:len(img) // 2 * 2], dtype="<u2").astype(np.int64)
    ssat = [2 * i for i in range(len(hw) - 1)
            if (int(hw[i]) & 0xFFF0) == 0xF300 and (int(hw[i + 1]) & 0xF0FF) == 0x000F]      # sat_imm - 1 = 15, imm3:imm2 = 0
    lo, hi = min(ssat) - 0x60, max(ssat) + 4
    offs, names = [], []
    for i in range(lo // 2, hi // 2):
        h1, h2 = int(hw[i]), int(hw[i + 1])
        name = None
        if (h1 & 0xFFB0) == 0xEEB0 and (h2 & 0x0FF0) == 0x0A00 and (h1 & 0xF) in (0x6, 0xE) and (h2 & 0xF) == 0:
            name = "VMOV #0.5" if (h1 & 0xF) == 0x6 else "VMOV #-0.5"
        elif (h1 & 0xFFBF) == 0xEEB5 and (h2 & 0x0F50) == 0x0A40:
            name = "VCMP #0"
        elif (h1 & 0xFFBF) == 0xEEBD and (h2 & 0x0FD0) == 0x0AC0:
            name = "VCVT.S32.F32"
        elif (h1 & 0xFFF0) == 0xF300 and (h2 & 0xF0FF) == 0x000F:
            name = "SSAT #16"
        elif (h1 & 0xFFB0) == 0xEE20 and (h2 & 0x0F50) == 0x0A00:
            name = "VMUL"
        elif (h1 & 0xFFB0) == 0xEE30 and (h2 & 0x0F50) == 0x0A00:
            name = "VADD"
        if name:
            offs.append(2 * i)
            names.append(name)
    return np.array(offs, np.int64), np.array(names)


def fixed_biquad_classes(img):
    """The instruction classes of AudioFilterBiquad::update around the image's only SMLAWT instructions: the 32 x 16
    multiply-accumulates (SMLAWB / SMLAWT), the saturating `SSAT Rd, #16, Rn, ASR #14`, the 14-bit error feedback
    (`UBFX Rd, Rn, #0, #14`) and the PKHBT that packs the two outputs of a loop turn."""
    hw = np.frombuffer(img[:len(img) // 2 * 2], dtype="<u2").astype(np.int64)

    def cls(h1, h2):
        if (h1 & 0xFFF0) == 0xFB30 and (h2 & 0x00E0) == 0 and (h2 & 0xF000) != 0xF000:
            return "SMLAWT" if h2 & 0x10 else "SMLAWB"
        if (h1 & 0xFFD0) == 0xF300 and (h2 & 0x8020) == 0:
            return "SSAT #%d %s #%d" % ((h2 & 0x1F) + 1, "ASR" if h1 & 0x20 else "LSL", ((h2 >> 12) & 7) << 2 | ((h2 >> 6) & 3))
        if (h1 & 0xFFF0) == 0xF3C0 and (h2 & 0x8020) == 0:
            return "UBFX #%d #%d" % (((h2 >> 12) & 7) << 2 | ((h2 >> 6) & 3), (h2 & 0x1F) + 1)
        if (h1 & 0xFFF0) == 0xEAC0 and (h2 & 0x8010) == 0:
            return "PKHTB" if h2 & 0x20 else "PKHBT"
        return None

    top = [2 * i for i in range(len(hw) - 1) if cls(int(hw[i]), int(hw[i + 1])) == "SMLAWT"]
    lo, hi = min(top) - 0x40, max(top) + 0x20
    offs, names = [], []
    for i in range(lo // 2, hi // 2):
        name = cls(int(hw[i]), int(hw[i + 1]))
        if name:
            offs.append(2 * i)
            names.append(name)
    return np.array(offs, np.int64), np.array(names)


def main():
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    lo, img = read_ihex(HEX)
    assert lo == 0x60000000 and len(img) == 206012, (hex(lo), len(img))
    out = {}
    for name, (off, count, dt) in TABLES.items():
        size = np.dtype(dt).itemsize * count
        out[name] = np.frombuffer(img[off:off + size], dtype=dt).copy()
        out[name + "_offset"] = np.int64(off)
    out["biquad_sets"] = out["biquad_sets"].reshape(15, 4, 5)
    out["code_ops_offsets"], out["code_ops_names"] = dsp_opcode_classes(img)
    out["code_vfp_offsets"], out["code_vfp_names"] = vfp_f32_classes(img)
    out["code_pack_offsets"], out["code_pack_names"] = pack_routine_classes(img)
    out["code_fixbq_offsets"], out["code_fixbq_names"] = fixed_biquad_classes(img)
    # what the image does NOT hold: CMSIS' sinTable_f32 (513 entries, arm_sin_f32 / arm_cos_f32 of SPEC:229-232) -- at any
    # 2-byte alignment; the only sine table is the engine's 257-entry oscillator table above
    s513 = np.sin(2 * np.pi * np.arange(1, 4) / 512).astype(np.float32)
    hits = 0
    for off in (0, 2):
        f = np.frombuffer(img[off:off + (len(img) - off) // 4 * 4], dtype="<f4")
        with np.errstate(invalid="ignore"):
            hits += int(np.count_nonzero((np.abs(f[:-2] - s513[0]) < 1e-6) & (np.abs(f[1:-1] - s513[1]) < 1e-6) & (np.abs(f[2:] - s513[2]) < 1e-6)))
    out["cmsis_sin513_occurrences"] = np.int64(hits)
    # sanity: what each table is, so a wrong offset cannot slip through
    i = np.arange(256)
    assert np.array_equal(out["hann256"], np.minimum(32767, np.round(32768 * 0.5 * (1 - np.cos(2 * np.pi * i / 255)))))
    assert out["sqrt_guess"][0] == 55109 and out["sqrt_guess"][32] == 0
    assert out["bitrev_1024"][0] == 0x400 and out["twiddle_q15_4096"][3] == 50
    assert abs(out["hilbert_half64"][63] + 2 / np.pi) < 1e-4 and out["sine257"][64] == 1.0
    assert out["sample_rate"][0] == 44100.0 and out["nr_gain"][0] == 1.1 and out["lms_epsilon"][0] == np.float32(1.19209289e-7)
    assert out["design_constants"][0] == np.pi and out["design_constants"][1] == 0.01 and out["iq_gain_balance"][0] == np.float32(1.02)
    if "--check" in sys.argv[1:]:                      # recompute and compare with the committed fixture
        have = np.load(OUT)
        bad = [k for k in out if k not in have.files or not np.array_equal(np.asarray(out[k]), have[k])] + [k for k in have.files if k not in out]
        print("check against", OUT, ":", "identical (%d arrays)" % len(out) if not bad else "DIFFERENT: %s" % bad)
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
