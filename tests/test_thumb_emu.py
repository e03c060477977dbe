"""The instruction-set interpreter behind tests/golden/firmware_kat.npz, checked on its own: hand-assembled Thumb-2
fragments (encodings from the ARMv7-M Architecture Reference Manual) with known results.  The interpreter's real proof
is elsewhere -- the image's integer routines run under it reproduce independent C restatements bit for bit
(tests/test_firmware_kat.py) -- but these run anywhere, without the reference tree."""
import os
import struct
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "thumb_emu.py")):
    pytest.skip("the interpreter stays in the build container (.gpurunignore)", allow_module_level=True)
from thumb_emu import Cpu, Memory, Unimplemented, bits_f32, f32_bits, fma32, round_fraction  # noqa: E402

BX_LR = 0x4770


def run(halfwords, r=(), s=()):
    mem = Memory()
    code = mem.map(0x1000, 0x1000)
    mem.map(0x20000000, 0x1000)
    prog = list(halfwords) + [BX_LR]
    code[:2 * len(prog)] = struct.pack("<%dH" % len(prog), *prog)
    cpu = Cpu(mem)
    cpu.r[13] = 0x20000800
    for i, v in enumerate(s):
        cpu.sets(i, v)
    cpu.call(0x1000, list(r))
    return cpu


def test_data_processing_flags_and_it_blocks():
    c = run([0x2005, 0x3003, 0x0081])                       # movs r0,#5; adds r0,#3; lsls r1,r0,#2
    assert c.r[0] == 8 and c.r[1] == 32
    c = run([0x1A40], r=[3, 5])                             # subs r0, r0, r1 : 3 - 5
    assert c.r[0] == 0xFFFFFFFE and (c.n, c.z, c.c, c.v) == (1, 0, 0, 0)
    c = run([0x1840], r=[0x7FFFFFFF, 1])                    # adds r0, r0, r1 : signed overflow
    assert c.r[0] == 0x80000000 and (c.n, c.z, c.c, c.v) == (1, 0, 0, 1)
    for x, want in ((0, 1), (7, 2)):
        c = run([0x2800, 0xBF0C, 0x2101, 0x2102], r=[x])    # cmp r0,#0; ite eq; moveq r1,#1; movne r1,#2
        assert c.r[1] == want
    c = run([0xF04F, 0x30FF])                               # mov.w r0, #0xffffffff (modified immediate 0xFF replicated)
    assert c.r[0] == 0xFFFFFFFF
    c = run([0xF64A, 0x30CD, 0xF2C1, 0x2034])               # movw r0,#0xabcd; movt r0,#0x1234
    assert c.r[0] == 0x1234ABCD


def test_saturation_and_dsp_extension():
    c = run([0xF321, 0x308F], r=[0, 0x7FFFFFFF])            # ssat r0, #16, r1, asr #14
    assert c.r[0] == 32767 and c.q == 1
    c = run([0xF321, 0x308F], r=[0, (-5 << 14) & 0xFFFFFFFF])
    assert c.r[0] == 0xFFFFFFFB and c.q == 0
    c = run([0xFB31, 0x3002], r=[0, 0x40000000, 0x00017FFF, 10])   # smlawb r0, r1, r2, r3 : 10 + (2^30 * 32767 >> 16)
    assert c.r[0] == 10 + ((0x40000000 * 32767) >> 16)
    c = run([0xFB31, 0x3012], r=[0, 0x40000000, 0x00017FFF, 10])   # smlawt: top half (1)
    assert c.r[0] == 10 + (0x40000000 >> 16)
    c = run([0xFA91, 0xF012], r=[0, 0x7FFF8000, 0x00010001])   # qadd16 r0, r1, r2 : the top saturates, the bottom does not
    assert c.r[0] == 0x7FFF8001
    c = run([0xFA91, 0xF022], r=[0, 0x7FFF8000, 0x7FFF8000])       # shadd16: halved sums
    assert c.r[0] == 0x7FFF8000
    c = run([0xFB21, 0xF002], r=[0, 0x00030002, 0x00050004])       # smuad r0, r1, r2 : 2*4 + 3*5
    assert c.r[0] == 23
    c = run([0xFBB1, 0xF0F2], r=[0, 100, 7])                       # udiv
    assert c.r[0] == 14
    c = run([0xFAB1, 0xF081], r=[0, 0x00010000])                   # clz
    assert c.r[0] == 15
    c = run([0xF3C1, 0x000D], r=[0, 0xFFFFFFFF])                   # ubfx r0, r1, #0, #14
    assert c.r[0] == 0x3FFF


def test_memory_and_stack():
    c = run([0xB430, 0x2400, 0x2500, 0xBC30], r=[0, 0])            # push {r4,r5}; movs r4,#0; movs r5,#0; pop {r4,r5}
    assert c.r[13] == 0x20000800
    c = run([0x6001, 0x6842, 0x8803], r=[0x20000010, 0xDEADBEEF])  # str r1,[r0]; ldr r2,[r0,#4]; ldrh r3,[r0]
    assert c.mem.read(0x20000010, 4) == 0xDEADBEEF and c.r[2] == 0 and c.r[3] == 0xBEEF
    with pytest.raises(MemoryError):
        run([0x6801], r=[0x40000000])                              # ldr r1,[r0] outside the mapped regions


def test_floating_point_is_ieee_single():
    c = run([0xEE30, 0x0A20], s=[0.1, 0.2])                        # vadd.f32 s0, s0, s1
    assert c.s[0] == f32_bits(bits_f32(f32_bits(0.1)) + bits_f32(f32_bits(0.2)))
    c = run([0xEE20, 0x0A20], s=[1.0 + 2.0 ** -23, 1.0 + 2.0 ** -23])   # vmul.f32: product rounded to even
    assert c.fs(0) == 1.0 + 2.0 ** -22
    # vfma.f32 s0, s1, s2 : one rounding where multiply-then-add has two
    a = 1.0 + 2.0 ** -12
    c = run([0xEEA0, 0x0A81], s=[-1.0, a, a])
    assert c.fs(0) == 2.0 ** -11 + 2.0 ** -24                        # exact: a*a - 1 = 2^-11 + 2^-24 (representable)
    c = run([0xEE20, 0x0A81], s=[0.0, a, a])                        # vmul.f32 s0, s1, s2 : a*a rounds to 1 + 2^-11, losing the 2^-24
    assert c.fs(0) == 1.0 + 2.0 ** -11
    c = run([0xEEFD, 0x0AC0], s=[-2.75])                             # vcvt.s32.f32 s1, s0 (toward zero)
    assert c.s[1] == 0xFFFFFFFE
    c = run([0xEEB1, 0x0AC0], s=[2.0])                               # vsqrt.f32 s0, s0
    assert c.s[0] == f32_bits(2.0 ** 0.5)
    assert fma32(3.0, 5.0, 7.0) == 22.0 and round_fraction(__import__("fractions").Fraction(1, 3), 24, -126, 127) == bits_f32(f32_bits(1 / 3))


def test_unknown_encodings_raise():
    with pytest.raises(Unimplemented):
        run([0xDE00])                                               # udf
