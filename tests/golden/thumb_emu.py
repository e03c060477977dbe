"""A small ARMv7E-M (Thumb-2 + DSP extension + FPv5) interpreter, written for one purpose: to run single routines
of the reference's shipped firmware image (pre_compiled/*.hex, a Teensy 4 / Cortex-M7 build) on inputs of our choosing
and record what they return, so that the restatements in oracle/ can be checked against what the reference's own
compiled code computes (tests/golden/make_firmware_kat.py -> tests/golden/firmware_kat.npz).

Test infrastructure, build container only: it needs the image under /root/reference, which does not travel.  Nothing of
the image is kept here; the interpreter knows the instruction set (ARMv7-M Architecture Reference Manual), not the
program.  No peripherals, no interrupts, no exceptions: flat memory regions and the core's registers.  An encoding
that is not implemented raises Unimplemented with its address, so nothing is ever silently skipped.

Floating point: single- and double-precision arithmetic is IEEE round-to-nearest-even (FPSCR as the Teensy core
leaves it: no flush-to-zero, default NaN off); +, -, x, /, sqrt of floats are computed in double and rounded once
more, which is exact for those operations; fused multiply-adds are evaluated exactly (rationals) and rounded once.
"""
import ctypes
import math
import struct
from fractions import Fraction

M32 = 0xFFFFFFFF


class Unimplemented(Exception):
    pass


def s32(v):
    v &= M32
    return v - (1 << 32) if v & 0x80000000 else v


def s16(v):
    v &= 0xFFFF
    return v - 65536 if v & 0x8000 else v


def s8(v):
    v &= 0xFF
    return v - 256 if v & 0x80 else v


def sext(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


_cf = ctypes.c_float


def f32_round(x):
    """double -> nearest float (as a Python float), overflow to inf"""
    return _cf(x).value


def bits_f32(b):
    return struct.unpack("<f", struct.pack("<I", b & M32))[0]


def f32_bits(x):
    try:
        return struct.unpack("<I", struct.pack("<f", x))[0]
    except OverflowError:
        return 0x7F800000 if x > 0 else 0xFF800000


def bits_f64(b):
    return struct.unpack("<d", struct.pack("<Q", b & 0xFFFFFFFFFFFFFFFF))[0]


def f64_bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def round_fraction(fr, pbits, emin, emax):
    """nearest-even rounding of a rational to a binary format with pbits of precision (incl. the hidden bit), normal
    exponents emin..emax (of the leading bit); returns a Python float (exact for pbits <= 53)"""
    if fr == 0:
        return 0.0
    sign = -1.0 if fr < 0 else 1.0
    a = abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, emin)
    scale = e - (pbits - 1)                                  # value = m * 2^scale
    q = a / (Fraction(2) ** scale)
    m = q.numerator // q.denominator
    rem = q - m
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (m & 1)):
        m += 1
    val = math.ldexp(m, scale) if scale > -1100 else 0.0      # (a carry into the next binade leaves m a power of two: still exact)
    if m.bit_length() + scale - 1 > emax:
        return sign * math.inf
    return sign * val


def fma32(a, b, c):
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return f32_round(a * b + c)
    r = round_fraction(Fraction(a) * Fraction(b) + Fraction(c), 24, -126, 127)
    if r == 0.0:                                              # exact zero: IEEE gives -0 only when product and addend are both -0
        p = a * b
        return -0.0 if (p == 0.0 and c == 0.0 and math.copysign(1, p) < 0 and math.copysign(1, c) < 0) else 0.0
    return r


def fma64(a, b, c):
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c
    return round_fraction(Fraction(a) * Fraction(b) + Fraction(c), 53, -1022, 1023)


class Memory:
    def __init__(self):
        self.regions = []                                    # (base, end, bytearray)

    def map(self, base, size_or_bytes):
        buf = bytearray(size_or_bytes) if isinstance(size_or_bytes, int) else bytearray(size_or_bytes)
        self.regions.append((base, base + len(buf), buf))
        return buf

    def _find(self, addr, n):
        for base, end, buf in self.regions:
            if base <= addr and addr + n <= end:
                return buf, addr - base
        raise MemoryError("access outside the mapped regions: 0x%08x (+%d)" % (addr, n))

    def read(self, addr, n):
        buf, o = self._find(addr, n)
        return int.from_bytes(buf[o:o + n], "little")

    def write(self, addr, n, v):
        buf, o = self._find(addr, n)
        buf[o:o + n] = (v & ((1 << (8 * n)) - 1)).to_bytes(n, "little")

    def read_bytes(self, addr, n):
        buf, o = self._find(addr, n)
        return bytes(buf[o:o + n])

    def write_bytes(self, addr, data):
        buf, o = self._find(addr, len(data))
        buf[o:o + len(data)] = data


RETURN_MAGIC = 0x7FFFFFF0


class Cpu:
    def __init__(self, mem):
        self.mem = mem
        self.r = [0] * 16
        self.n = self.z = self.c = self.v = self.q = 0
        self.ge = 0
        self.s = [0] * 64                                    # S0..S31 as bit patterns; D16..D31 do not exist on FPv5-D16 (kept for safety)
        self.fn = self.fz = self.fc = self.fv = 0             # FPSCR flags
        self.it = 0
        self.hooks = {}                                      # address -> f(cpu): runs INSTEAD of the routine there, then returns to lr
        self.watch = {}                                      # address -> f(cpu): runs when execution reaches the address, then goes on
        self.count = 0
        self.trace = None

    # ---- registers / flags ------------------------------------------------------------------------------------------
    def cond(self, c):
        k = c >> 1
        if k == 0: r = self.z
        elif k == 1: r = self.c
        elif k == 2: r = self.n
        elif k == 3: r = self.v
        elif k == 4: r = self.c and not self.z
        elif k == 5: r = self.n == self.v
        elif k == 6: r = self.n == self.v and not self.z
        else: r = True
        r = bool(r)
        if (c & 1) and c != 15:
            r = not r
        return r

    def setnz(self, v):
        self.n = (v >> 31) & 1
        self.z = 1 if (v & M32) == 0 else 0

    def addc(self, x, y, c, setflags):
        us = (x & M32) + (y & M32) + c
        res = us & M32
        if setflags:
            self.setnz(res)
            self.c = us >> 32
            self.v = 1 if s32(res) != s32(x) + s32(y) + c else 0
        return res

    def getd(self, d):
        return self.s[2 * d] | (self.s[2 * d + 1] << 32)

    def setd(self, d, v):
        self.s[2 * d] = v & M32
        self.s[2 * d + 1] = (v >> 32) & M32

    def fs(self, i):
        return bits_f32(self.s[i])

    def sets(self, i, x):
        self.s[i] = f32_bits(x)

    def fd(self, i):
        return bits_f64(self.getd(i))

    def setfd(self, i, x):
        self.setd(i, f64_bits(x))

    # ---- shifts -----------------------------------------------------------------------------------------------------
    def shift_c(self, v, typ, amount, cin):
        v &= M32
        if amount == 0:
            return v, cin
        if typ == 0:                                          # LSL
            if amount > 32: return 0, 0
            r = v << amount
            return r & M32, (r >> 32) & 1
        if typ == 1:                                          # LSR
            if amount > 32: return 0, 0
            return (v >> amount) & M32, (v >> (amount - 1)) & 1
        if typ == 2:                                          # ASR
            if amount > 32: amount = 32
            sv = s32(v)
            return (sv >> amount) & M32, (sv >> (amount - 1)) & 1
        if typ == 3:                                          # ROR
            a = amount % 32
            r = ((v >> a) | (v << (32 - a))) & M32 if a else v
            return r, (r >> 31) & 1
        if typ == 4:                                          # RRX
            return ((cin << 31) | (v >> 1)) & M32, v & 1
        raise AssertionError

    @staticmethod
    def decode_imm_shift(typ, imm5):
        if typ == 0: return 0, imm5
        if typ == 1: return 1, imm5 or 32
        if typ == 2: return 2, imm5 or 32
        return (4, 1) if imm5 == 0 else (3, imm5)

    def expand_imm_c(self, imm12):
        if (imm12 >> 10) == 0:
            k, b = (imm12 >> 8) & 3, imm12 & 0xFF
            if k == 0: v = b
            elif k == 1: v = (b << 16) | b
            elif k == 2: v = (b << 24) | (b << 8)
            else: v = (b << 24) | (b << 16) | (b << 8) | b
            return v, self.c
        un = 0x80 | (imm12 & 0x7F)
        rot = imm12 >> 7
        v = ((un >> rot) | (un << (32 - rot))) & M32
        return v, (v >> 31) & 1

    # ---- memory helpers ---------------------------------------------------------------------------------------------
    def ld(self, a, n): return self.mem.read(a & M32, n)
    def st(self, a, n, v): self.mem.write(a & M32, n, v)

    def branch(self, target):
        self.r[15] = target & ~1 & M32

    def in_it(self):
        return (self.it & 0xF) != 0

    # ---- run --------------------------------------------------------------------------------------------------------
    def call(self, addr, args=(), sargs=(), stack_args=(), dargs=None, max_steps=200_000_000):
        """AAPCS-VFP call: integer / pointer args in r0-r3 then the stack; float args (Python floats) in s0... (sargs) or
        doubles in the given d registers (dargs = {index: value}); returns r0"""
        for i, a in enumerate(args[:4]):
            self.r[i] = a & M32
        extra = list(args[4:]) + list(stack_args)
        sp0 = self.r[13]
        sp = sp0 & ~7
        sp -= 4 * len(extra)
        if sp & 7:
            sp -= 4
        for i, a in enumerate(extra):
            self.st(sp + 4 * i, 4, a)
        self.r[13] = sp
        for i, x in enumerate(sargs):
            self.sets(i, x)
        for i, x in (dargs or {}).items():
            self.setfd(i, x)
        self.r[14] = RETURN_MAGIC | 1
        self.branch(addr)
        self.it = 0
        start = self.count
        while self.r[15] != RETURN_MAGIC:
            h = self.hooks.get(self.r[15])
            if h is not None:
                h(self)
                self.branch(self.r[14])
                continue
            if self.watch:
                w = self.watch.get(self.r[15])
                if w is not None:
                    w(self)
            self.step()
            if self.count - start > max_steps:
                raise RuntimeError("step limit")
        self.r[13] = sp0
        return self.r[0]

    def step(self):
        pc = self.r[15]
        hw1 = self.mem.read(pc, 2)
        self.count += 1
        wide = (hw1 >> 11) >= 0b11101
        if self.it & 0xF:
            c = self.it >> 4
            ok = self.cond(c)
            self.it = 0 if (self.it & 7) == 0 else ((self.it & 0xE0) | ((self.it << 1) & 0x1F))
            initblock = True
        else:
            ok, initblock = True, False
        if self.trace is not None:
            self.trace(self, pc, hw1)
        if wide:
            hw2 = self.mem.read(pc + 2, 2)
            self.r[15] = pc + 4
            if ok:
                self.exec32(pc, hw1, hw2, initblock)
        else:
            self.r[15] = pc + 2
            if ok:
                self.exec16(pc, hw1, initblock)

    # ---- 16-bit -----------------------------------------------------------------------------------------------------
    def exec16(self, pc, hw, initb):
        r = self.r
        sf = not initb
        top = hw >> 11
        if top < 3:                                           # LSL / LSR / ASR imm
            rd, rm, imm5 = hw & 7, (hw >> 3) & 7, (hw >> 6) & 31
            t, a = self.decode_imm_shift(top, imm5)
            v, c = self.shift_c(r[rm], t, a, self.c)
            r[rd] = v
            if sf:
                self.setnz(v); self.c = c
        elif top == 3:
            rd, rn, x = hw & 7, (hw >> 3) & 7, (hw >> 6) & 7
            y = x if hw & 0x400 else r[x]
            r[rd] = self.addc(r[rn], (~y) & M32, 1, sf) if hw & 0x200 else self.addc(r[rn], y, 0, sf)
        elif top < 8:
            rd, imm = (hw >> 8) & 7, hw & 0xFF
            if top == 4:
                r[rd] = imm
                if sf: self.setnz(imm)
            elif top == 5:
                self.addc(r[rd], (~imm) & M32, 1, True)
            elif top == 6:
                r[rd] = self.addc(r[rd], imm, 0, sf)
            else:
                r[rd] = self.addc(r[rd], (~imm) & M32, 1, sf)
        elif top == 8:
            if not hw & 0x400:
                op, rm, rd = (hw >> 6) & 15, (hw >> 3) & 7, hw & 7
                a, b = r[rd], r[rm]
                if op == 0: v = a & b
                elif op == 1: v = a ^ b
                elif op in (2, 3, 4, 7):
                    t = {2: 0, 3: 1, 4: 2, 7: 3}[op]
                    v, c = self.shift_c(a, t, b & 0xFF, self.c)
                    r[rd] = v
                    if sf: self.setnz(v); self.c = c
                    return
                elif op == 5: r[rd] = self.addc(a, b, self.c, sf); return
                elif op == 6: r[rd] = self.addc(a, (~b) & M32, self.c, sf); return
                elif op == 8: self.setnz(a & b); return
                elif op == 9: r[rd] = self.addc((~b) & M32, 0, 1, sf); return
                elif op == 10: self.addc(a, (~b) & M32, 1, True); return
                elif op == 11: self.addc(a, b, 0, True); return
                elif op == 12: v = a | b
                elif op == 13: v = (a * b) & M32
                elif op == 14: v = a & ~b & M32
                else: v = (~b) & M32
                r[rd] = v
                if sf: self.setnz(v)
            else:
                op = (hw >> 8) & 3
                rm = (hw >> 3) & 15
                rdn = (hw & 7) | ((hw >> 4) & 8)
                vm = (pc + 4) if rm == 15 else r[rm]
                if op == 0:
                    vn = (pc + 4) if rdn == 15 else r[rdn]
                    v = (vn + vm) & M32
                    if rdn == 15: self.branch(v)
                    else: r[rdn] = v
                elif op == 1:
                    self.addc(r[rdn], (~vm) & M32, 1, True)
                elif op == 2:
                    if rdn == 15: self.branch(vm)
                    else: r[rdn] = vm
                else:
                    if hw & 0x80:
                        r[14] = (pc + 2) | 1
                    self.branch(vm)
        elif top == 9:
            rt = (hw >> 8) & 7
            r[rt] = self.ld(((pc + 4) & ~3) + ((hw & 0xFF) << 2), 4)
        elif top in (10, 11):
            op, rm, rn, rt = (hw >> 9) & 7, (hw >> 6) & 7, (hw >> 3) & 7, hw & 7
            a = (r[rn] + r[rm]) & M32
            if op == 0: self.st(a, 4, r[rt])
            elif op == 1: self.st(a, 2, r[rt])
            elif op == 2: self.st(a, 1, r[rt])
            elif op == 3: r[rt] = s8(self.ld(a, 1)) & M32
            elif op == 4: r[rt] = self.ld(a, 4)
            elif op == 5: r[rt] = self.ld(a, 2)
            elif op == 6: r[rt] = self.ld(a, 1)
            else: r[rt] = s16(self.ld(a, 2)) & M32
        elif top in (12, 13, 14, 15, 16, 17):
            imm5, rn, rt = (hw >> 6) & 31, (hw >> 3) & 7, hw & 7
            size = 4 if top < 14 else (1 if top < 16 else 2)
            a = r[rn] + imm5 * size
            if top & 1: r[rt] = self.ld(a, size)
            else: self.st(a, size, r[rt])
        elif top in (18, 19):
            rt, a = (hw >> 8) & 7, r[13] + ((hw & 0xFF) << 2)
            if top & 1: r[rt] = self.ld(a, 4)
            else: self.st(a, 4, r[rt])
        elif top == 20:
            r[(hw >> 8) & 7] = (((pc + 4) & ~3) + ((hw & 0xFF) << 2)) & M32
        elif top == 21:
            r[(hw >> 8) & 7] = (r[13] + ((hw & 0xFF) << 2)) & M32
        elif top in (22, 23):
            self.misc16(pc, hw)
        elif top == 24:
            rn, a = (hw >> 8) & 7, r[(hw >> 8) & 7]
            for i in range(8):
                if hw & (1 << i):
                    self.st(a, 4, r[i]); a += 4
            r[rn] = a & M32
        elif top == 25:
            rn, a = (hw >> 8) & 7, r[(hw >> 8) & 7]
            for i in range(8):
                if hw & (1 << i):
                    r[i] = self.ld(a, 4); a += 4
            if not hw & (1 << rn):
                r[rn] = a & M32
        elif top in (26, 27):
            c = (hw >> 8) & 15
            if c >= 14:
                raise Unimplemented("UDF/SVC at 0x%x" % pc)
            if self.cond(c):
                self.branch(pc + 4 + (s8(hw & 0xFF) << 1))
        elif top == 28:
            self.branch(pc + 4 + (sext(hw & 0x7FF, 11) << 1))
        else:
            raise Unimplemented("16-bit 0x%04x at 0x%x" % (hw, pc))

    def misc16(self, pc, hw):
        r = self.r
        k = hw & 0xFF00
        if k == 0xB000:
            imm = (hw & 0x7F) << 2
            r[13] = (r[13] - imm if hw & 0x80 else r[13] + imm) & M32
        elif k in (0xB100, 0xB300, 0xB900, 0xBB00):
            rn = hw & 7
            off = (((hw >> 9) & 1) << 6) | (((hw >> 3) & 31) << 1)
            if (r[rn] == 0) != bool(hw & 0x800):
                self.branch(pc + 4 + off)
        elif k == 0xB200:
            rm, rd, op = (hw >> 3) & 7, hw & 7, (hw >> 6) & 3
            v = r[rm]
            r[rd] = (s16(v) if op == 0 else s8(v) if op == 1 else v & 0xFFFF if op == 2 else v & 0xFF) & M32
        elif k in (0xB400, 0xB500):
            regs = [i for i in range(8) if hw & (1 << i)] + ([14] if hw & 0x100 else [])
            a = r[13] - 4 * len(regs)
            r[13] = a & M32
            for i in regs:
                self.st(a, 4, r[i]); a += 4
        elif k in (0xBC00, 0xBD00):
            a = r[13]
            for i in range(8):
                if hw & (1 << i):
                    r[i] = self.ld(a, 4); a += 4
            if hw & 0x100:
                t = self.ld(a, 4); a += 4
                r[13] = a & M32
                self.branch(t)
            else:
                r[13] = a & M32
        elif k == 0xBA00:
            rm, rd, op = (hw >> 3) & 7, hw & 7, (hw >> 6) & 3
            v = r[rm]
            if op == 0: r[rd] = int.from_bytes(v.to_bytes(4, "little"), "big")
            elif op == 1: r[rd] = ((v & 0xFF00FF00) >> 8) | ((v & 0x00FF00FF) << 8)
            elif op == 3: r[rd] = s16(((v & 0xFF) << 8) | ((v >> 8) & 0xFF)) & M32
            else: raise Unimplemented("0x%04x at 0x%x" % (hw, pc))
        elif k == 0xBF00:
            if hw & 0xF:
                self.it = hw & 0xFF
            # else: NOP / YIELD / WFE / WFI / SEV hints
        elif k == 0xB600:
            pass                                              # CPSIE / CPSID: no interrupts here
        else:
            raise Unimplemented("16-bit misc 0x%04x at 0x%x" % (hw, pc))

    # ---- 32-bit -----------------------------------------------------------------------------------------------------
    def exec32(self, pc, hw1, hw2, initb):
        r = self.r
        if (hw1 & 0xEC00) == 0xEC00:
            return self.vfp(pc, hw1, hw2)
        op1 = (hw1 >> 11) & 3
        if op1 == 1:
            if (hw1 & 0xFE40) == 0xE800:
                return self.ldm_stm(pc, hw1, hw2)
            if (hw1 & 0xFE40) == 0xE840:
                return self.ldrd_strd(pc, hw1, hw2)
            if (hw1 & 0xFE00) == 0xEA00:
                return self.dp_shifted_reg(pc, hw1, hw2)
        elif op1 == 2:
            if hw2 & 0x8000:
                return self.branches(pc, hw1, hw2)
            if not hw1 & 0x0200:
                return self.dp_mod_imm(pc, hw1, hw2)
            return self.dp_plain_imm(pc, hw1, hw2)
        else:
            if (hw1 & 0xFE00) == 0xF800:
                return self.ldst_single(pc, hw1, hw2)
            if (hw1 & 0xFF00) == 0xFA00:
                return self.dp_reg(pc, hw1, hw2)
            if (hw1 & 0xFF80) == 0xFB00:
                return self.mul32(pc, hw1, hw2)
            if (hw1 & 0xFF80) == 0xFB80:
                return self.mul64(pc, hw1, hw2)
        raise Unimplemented("32-bit 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))

    def ldm_stm(self, pc, hw1, hw2):
        r = self.r
        op, w, l, rn = (hw1 >> 7) & 3, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15
        regs = [i for i in range(16) if hw2 & (1 << i)]
        if op == 1:
            a = r[rn]
            end = a + 4 * len(regs)
        elif op == 2:
            a = r[rn] - 4 * len(regs)
            end = a
        else:
            raise Unimplemented("LDM/STM mode at 0x%x" % pc)
        target = None
        for i in regs:
            if l:
                v = self.ld(a, 4)
                if i == 15: target = v
                else: r[i] = v
            else:
                self.st(a, 4, r[i])
            a += 4
        if w and not (l and rn in regs):
            r[rn] = end & M32
        if target is not None:
            self.branch(target)

    def ldrd_strd(self, pc, hw1, hw2):
        r = self.r
        p, u, w, l, rn = (hw1 >> 8) & 1, (hw1 >> 7) & 1, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15
        if not p and not w:
            if (hw1 & 0xFFF0) == 0xE8D0 and (hw2 & 0xFFE0) == 0xF000:      # TBB / TBH
                base = (pc + 4) if rn == 15 else r[rn]
                rm = hw2 & 15
                if hw2 & 0x10: off = self.ld(base + (r[rm] << 1), 2)
                else: off = self.ld(base + r[rm], 1)
                return self.branch(pc + 4 + 2 * off)
            raise Unimplemented("exclusive access at 0x%x" % pc)
        rt, rt2, imm = (hw2 >> 12) & 15, (hw2 >> 8) & 15, (hw2 & 0xFF) << 2
        base = ((pc + 4) & ~3) if rn == 15 else r[rn]
        off = base + imm if u else base - imm
        a = off if p else base
        if l:
            r[rt] = self.ld(a, 4); r[rt2] = self.ld(a + 4, 4)
        else:
            self.st(a, 4, r[rt]); self.st(a + 4, 4, r[rt2])
        if w:
            r[rn] = off & M32

    def dp_shifted_reg(self, pc, hw1, hw2):
        r = self.r
        op, s, rn = (hw1 >> 5) & 15, (hw1 >> 4) & 1, hw1 & 15
        rd, rm = (hw2 >> 8) & 15, hw2 & 15
        imm5, typ = (((hw2 >> 12) & 7) << 2) | ((hw2 >> 6) & 3), (hw2 >> 4) & 3
        if op == 6:                                           # PKHBT / PKHTB
            if typ & 1:
                raise Unimplemented("PKH encoding at 0x%x" % pc)
            tb = (hw2 >> 5) & 1
            t, a = self.decode_imm_shift(2 if tb else 0, imm5)
            v, _ = self.shift_c(r[rm], t, a, self.c)
            r[rd] = ((r[rn] & 0xFFFF0000) | (v & 0xFFFF)) if tb else ((v & 0xFFFF0000) | (r[rn] & 0xFFFF))
            return
        t, a = self.decode_imm_shift(typ, imm5)
        vm, c = self.shift_c(r[rm], t, a, self.c)
        vn = r[rn]
        logical = None
        if op == 0: logical = vn & vm
        elif op == 1: logical = vn & ~vm & M32
        elif op == 2: logical = vm if rn == 15 else (vn | vm)
        elif op == 3: logical = ((~vm) & M32) if rn == 15 else (vn | (~vm & M32))
        elif op == 4: logical = vn ^ vm
        if logical is not None:
            if s:
                self.setnz(logical); self.c = c
            if not (rd == 15 and s and op in (0, 4)):
                if rd == 15: raise Unimplemented("write to pc at 0x%x" % pc)
                r[rd] = logical
            return
        if op == 8: v = self.addc(vn, vm, 0, s)
        elif op == 10: v = self.addc(vn, vm, self.c, s)
        elif op == 11: v = self.addc(vn, (~vm) & M32, self.c, s)
        elif op == 13: v = self.addc(vn, (~vm) & M32, 1, s)
        elif op == 14: v = self.addc((~vn) & M32, vm, 1, s)
        else: raise Unimplemented("dp shifted reg op %d at 0x%x" % (op, pc))
        if not (rd == 15 and s):
            r[rd] = v

    def dp_mod_imm(self, pc, hw1, hw2):
        r = self.r
        op, s, rn = (hw1 >> 5) & 15, (hw1 >> 4) & 1, hw1 & 15
        rd = (hw2 >> 8) & 15
        imm12 = (((hw1 >> 10) & 1) << 11) | (((hw2 >> 12) & 7) << 8) | (hw2 & 0xFF)
        imm, c = self.expand_imm_c(imm12)
        vn = r[rn]
        logical = None
        if op == 0: logical = vn & imm
        elif op == 1: logical = vn & ~imm & M32
        elif op == 2: logical = imm if rn == 15 else (vn | imm)
        elif op == 3: logical = ((~imm) & M32) if rn == 15 else (vn | (~imm & M32))
        elif op == 4: logical = vn ^ imm
        if logical is not None:
            if s:
                self.setnz(logical); self.c = c
            if not (rd == 15 and s and op in (0, 4)):
                r[rd] = logical
            return
        if op == 8: v = self.addc(vn, imm, 0, s)
        elif op == 10: v = self.addc(vn, imm, self.c, s)
        elif op == 11: v = self.addc(vn, (~imm) & M32, self.c, s)
        elif op == 13: v = self.addc(vn, (~imm) & M32, 1, s)
        elif op == 14: v = self.addc((~vn) & M32, imm, 1, s)
        else: raise Unimplemented("dp modified imm op %d at 0x%x" % (op, pc))
        if not (rd == 15 and s):
            r[rd] = v

    def dp_plain_imm(self, pc, hw1, hw2):
        r = self.r
        op, rn, rd = (hw1 >> 4) & 31, hw1 & 15, (hw2 >> 8) & 15
        i, imm3, imm8 = (hw1 >> 10) & 1, (hw2 >> 12) & 7, hw2 & 0xFF
        imm12 = (i << 11) | (imm3 << 8) | imm8
        if op == 0:
            base = ((pc + 4) & ~3) if rn == 15 else r[rn]
            r[rd] = (base + imm12) & M32
        elif op == 10:
            base = ((pc + 4) & ~3) if rn == 15 else r[rn]
            r[rd] = (base - imm12) & M32
        elif op == 4:
            r[rd] = ((hw1 & 15) << 12) | imm12
        elif op == 12:
            r[rd] = (r[rd] & 0xFFFF) | ((((hw1 & 15) << 12) | imm12) << 16)
        elif op in (16, 18, 24, 26):
            sh = (imm3 << 2) | ((hw2 >> 6) & 3)
            sat = hw2 & 31
            unsigned = op >= 24
            if (op & 2) and sh == 0:                          # SSAT16 / USAT16
                n = (hw2 & 15) + (0 if unsigned else 1)
                out = 0
                for k in (0, 16):
                    v = s16(r[rn] >> k)
                    lo, hi = (0, (1 << n) - 1) if unsigned else (-(1 << (n - 1)), (1 << (n - 1)) - 1)
                    if v < lo: v = lo; self.q = 1
                    elif v > hi: v = hi; self.q = 1
                    out |= (v & 0xFFFF) << k
                r[rd] = out
                return
            v = s32(r[rn])
            v = (v >> sh) if (op & 2) else s32(v << sh)
            n = sat if unsigned else sat + 1
            lo, hi = (0, (1 << n) - 1) if unsigned else (-(1 << (n - 1)), (1 << (n - 1)) - 1)
            if v < lo: v = lo; self.q = 1
            elif v > hi: v = hi; self.q = 1
            r[rd] = v & M32
        elif op in (20, 28):
            lsb = (imm3 << 2) | ((hw2 >> 6) & 3)
            width = (hw2 & 31) + 1
            v = (r[rn] >> lsb) & ((1 << width) - 1)
            r[rd] = (sext(v, width) & M32) if op == 20 else v
        elif op == 22:
            lsb = (imm3 << 2) | ((hw2 >> 6) & 3)
            msb = hw2 & 31
            width = msb - lsb + 1
            mask = ((1 << width) - 1) << lsb
            src = 0 if rn == 15 else r[rn]
            r[rd] = (r[rd] & ~mask & M32) | ((src << lsb) & mask)
        else:
            raise Unimplemented("dp plain imm op %d at 0x%x" % (op, pc))

    def branches(self, pc, hw1, hw2):
        s = (hw1 >> 10) & 1
        j1, j2 = (hw2 >> 13) & 1, (hw2 >> 11) & 1
        if (hw2 & 0x5000) == 0:
            c = (hw1 >> 6) & 15
            if c >= 14:
                # MSR / MRS / hints / barriers: nothing to do without a system
                if (hw1 & 0xFFF0) == 0xF3B0 or (hw1 & 0xFFF0) == 0xF3A0:
                    return
                if (hw1 & 0xFFE0) == 0xF3E0:                  # MRS
                    self.r[(hw2 >> 8) & 15] = 0
                    return
                if (hw1 & 0xFFE0) == 0xF380:                  # MSR
                    return
                raise Unimplemented("misc control 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
            imm = sext((s << 20) | (j2 << 19) | (j1 << 18) | ((hw1 & 63) << 12) | ((hw2 & 0x7FF) << 1), 21)
            if self.cond(c):
                self.branch(pc + 4 + imm)
            return
        i1, i2 = 1 - (j1 ^ s), 1 - (j2 ^ s)
        imm = sext((s << 24) | (i1 << 23) | (i2 << 22) | ((hw1 & 0x3FF) << 12) | ((hw2 & 0x7FF) << 1), 25)
        if hw2 & 0x4000:
            if not hw2 & 0x1000:
                raise Unimplemented("BLX imm at 0x%x" % pc)
            self.r[14] = (pc + 4) | 1
        self.branch(pc + 4 + imm)

    def ldst_single(self, pc, hw1, hw2):
        r = self.r
        size, l, sign, rn, rt = (hw1 >> 5) & 3, (hw1 >> 4) & 1, (hw1 >> 8) & 1, hw1 & 15, (hw2 >> 12) & 15
        n = 1 << size
        if size == 3:
            raise Unimplemented("ldst size at 0x%x" % pc)
        wb = None
        if rn == 15:
            if not l: raise Unimplemented("store literal at 0x%x" % pc)
            imm = hw2 & 0xFFF
            a = ((pc + 4) & ~3) + (imm if hw1 & 0x80 else -imm)
        elif hw1 & 0x80:
            a = r[rn] + (hw2 & 0xFFF)
        elif hw2 & 0x800:
            p, u, w, imm = (hw2 >> 10) & 1, (hw2 >> 9) & 1, (hw2 >> 8) & 1, hw2 & 0xFF
            off = r[rn] + imm if u else r[rn] - imm
            a = off if p else r[rn]
            if w: wb = off & M32
        elif (hw2 & 0x0FC0) == 0:
            a = r[rn] + ((r[hw2 & 15] << ((hw2 >> 4) & 3)) & M32)
        else:
            raise Unimplemented("ldst form 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        if l:
            if rt == 15 and size < 2:
                pass                                          # PLD / PLI: hint
            else:
                v = self.ld(a, n)
                if sign:
                    v = (s8(v) if n == 1 else s16(v)) & M32
                if wb is not None: r[rn] = wb
                if rt == 15: self.branch(v)
                else: r[rt] = v
                return
        else:
            self.st(a, n, r[rt])
        if wb is not None:
            r[rn] = wb

    def dp_reg(self, pc, hw1, hw2):
        r = self.r
        rn, rd, rm = hw1 & 15, (hw2 >> 8) & 15, hw2 & 15
        if (hw2 & 0xF000) != 0xF000:
            raise Unimplemented("dp reg 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        if not hw1 & 0x80:
            if (hw2 & 0xF0) == 0:                             # shift by register
                v, c = self.shift_c(r[rn], (hw1 >> 5) & 3, r[rm] & 0xFF, self.c)
                r[rd] = v
                if hw1 & 0x10:
                    self.setnz(v); self.c = c
                return
            if hw2 & 0x80:                                    # extend (and add)
                op, rot = (hw1 >> 4) & 7, ((hw2 >> 4) & 3) * 8
                v = ((r[rm] >> rot) | (r[rm] << (32 - rot))) & M32 if rot else r[rm]
                base = 0 if rn == 15 else r[rn]
                if op == 0: r[rd] = (base + s16(v)) & M32
                elif op == 1: r[rd] = (base + (v & 0xFFFF)) & M32
                elif op == 4: r[rd] = (base + s8(v)) & M32
                elif op == 5: r[rd] = (base + (v & 0xFF)) & M32
                elif op in (2, 3):
                    f = s8 if op == 2 else (lambda x: x & 0xFF)
                    lo = ((base & 0xFFFF) + f(v)) & 0xFFFF
                    hi = (((base >> 16) & 0xFFFF) + f(v >> 16)) & 0xFFFF
                    r[rd] = (hi << 16) | lo
                else: raise Unimplemented("extend op at 0x%x" % pc)
                return
            raise Unimplemented("dp reg 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        if (hw2 & 0x80) == 0:                                  # parallel add / sub
            return self.parallel(pc, (hw1 >> 4) & 7, (hw2 >> 4) & 7, rn, rd, rm)
        op1, op2 = (hw1 >> 4) & 3, (hw2 >> 4) & 3
        if (hw1 & 0xFFC0) != 0xFA80:
            raise Unimplemented("dp reg misc 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        if op1 == 0:
            a, b = s32(r[rm]), s32(r[rn])                      # QADD Rd, Rm, Rn : sat(Rm + Rn); QDADD: sat(Rm + sat(2 Rn))
            def sat(x):
                if x > 0x7FFFFFFF: self.q = 1; return 0x7FFFFFFF
                if x < -0x80000000: self.q = 1; return -0x80000000
                return x
            if op2 & 1: b = sat(2 * b)
            r[rd] = sat(a - b if op2 & 2 else a + b) & M32
        elif op1 == 1:
            v = r[rm]
            if op2 == 0: r[rd] = int.from_bytes(v.to_bytes(4, "little"), "big")
            elif op2 == 1: r[rd] = ((v & 0xFF00FF00) >> 8) | ((v & 0x00FF00FF) << 8)
            elif op2 == 2: r[rd] = int(format(v, "032b")[::-1], 2)
            else: r[rd] = s16(((v & 0xFF) << 8) | ((v >> 8) & 0xFF)) & M32
        elif op1 == 2 and op2 == 0:                           # SEL
            out = 0
            for k in range(4):
                src = r[rn] if (self.ge >> k) & 1 else r[rm]
                out |= src & (0xFF << (8 * k))
            r[rd] = out
        elif op1 == 3 and op2 == 0:
            r[rd] = 32 - r[rm].bit_length()
        else:
            raise Unimplemented("dp reg misc 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))

    def parallel(self, pc, op1, op2, rn, rd, rm):
        """op1: 1 ADD16, 2 ASX, 6 SAX, 5 SUB16, 0 ADD8, 4 SUB8; op2: 0 S, 1 Q, 2 SH, 4 U, 5 UQ, 6 UH"""
        a, b = self.r[rn], self.r[rm]
        unsigned = bool(op2 & 4)
        kind = op2 & 3
        if kind == 3:
            raise Unimplemented("parallel prefix at 0x%x" % pc)
        if op1 in (0, 4):
            raise Unimplemented("8-bit parallel at 0x%x" % pc)
        ext = (lambda x: x & 0xFFFF) if unsigned else s16
        al, ah, bl, bh = ext(a), ext(a >> 16), ext(b), ext(b >> 16)
        if op1 == 1: lo, hi = al + bl, ah + bh
        elif op1 == 5: lo, hi = al - bl, ah - bh
        elif op1 == 2: lo, hi = al - bh, ah + bl                # ASX: hi = a.hi + b.lo, lo = a.lo - b.hi
        elif op1 == 6: lo, hi = al + bh, ah - bl                # SAX: lo = a.lo + b.hi, hi = a.hi - b.lo
        else: raise Unimplemented("parallel op1 %d at 0x%x" % (op1, pc))
        if kind == 1:
            if unsigned: lo, hi = min(max(lo, 0), 0xFFFF), min(max(hi, 0), 0xFFFF)
            else: lo, hi = min(max(lo, -32768), 32767), min(max(hi, -32768), 32767)
        elif kind == 2:
            lo, hi = lo >> 1, hi >> 1
        else:
            if unsigned:
                raise Unimplemented("unsigned GE-setting parallel op at 0x%x" % pc)
            self.ge = (3 if lo >= 0 else 0) | (12 if hi >= 0 else 0)
        self.r[rd] = ((hi & 0xFFFF) << 16) | (lo & 0xFFFF)

    def mul32(self, pc, hw1, hw2):
        r = self.r
        op1, op2 = (hw1 >> 4) & 7, (hw2 >> 4) & 3
        rn, ra, rd, rm = hw1 & 15, (hw2 >> 12) & 15, (hw2 >> 8) & 15, hw2 & 15
        acc = 0 if ra == 15 else s32(r[ra])
        a, b = r[rn], r[rm]

        def setq(v):
            if v != s32(v): self.q = 1
            return v & M32
        if op1 == 0:
            if op2 == 0: r[rd] = (a * b + (0 if ra == 15 else r[ra])) & M32
            elif op2 == 1: r[rd] = (r[ra] - a * b) & M32
            else: raise Unimplemented("mul at 0x%x" % pc)
        elif op1 == 1:                                        # SMLAxy / SMULxy
            x = s16(a >> 16) if op2 & 2 else s16(a)
            y = s16(b >> 16) if op2 & 1 else s16(b)
            r[rd] = setq(x * y + acc)
        elif op1 in (2, 4):                                   # SMLAD(X) / SMLSD(X)
            if op2 & 2: raise Unimplemented("mul at 0x%x" % pc)
            if op2 & 1: b = ((b >> 16) | (b << 16)) & M32
            p1, p2 = s16(a) * s16(b), s16(a >> 16) * s16(b >> 16)
            r[rd] = setq((p1 + p2 if op1 == 2 else p1 - p2) + acc)
        elif op1 == 3:                                        # SMLAWy / SMULWy
            if op2 & 2: raise Unimplemented("mul at 0x%x" % pc)
            y = s16(b >> 16) if op2 & 1 else s16(b)
            r[rd] = setq(((s32(a) * y) >> 16) + acc)
        elif op1 in (5, 6):                                   # SMMLA(R) / SMMLS(R)
            if op2 & 2: raise Unimplemented("mul at 0x%x" % pc)
            p = s32(a) * s32(b)
            t = ((acc << 32) - p) if op1 == 6 else ((acc << 32) + p)
            if op2 & 1: t += 0x80000000
            r[rd] = (t >> 32) & M32
        else:
            raise Unimplemented("mul op1 %d at 0x%x" % (op1, pc))

    def mul64(self, pc, hw1, hw2):
        r = self.r
        op1, op2 = (hw1 >> 4) & 7, (hw2 >> 4) & 15
        rn, lo, hi, rm = hw1 & 15, (hw2 >> 12) & 15, (hw2 >> 8) & 15, hw2 & 15
        a, b = r[rn], r[rm]
        if op1 == 0 and op2 == 0: v = s32(a) * s32(b)
        elif op1 == 2 and op2 == 0: v = a * b
        elif op1 == 1 and op2 == 15:
            x, y = s32(a), s32(b)
            r[hi] = 0 if y == 0 else (int(abs(x) // abs(y)) * (1 if (x < 0) == (y < 0) else -1)) & M32
            return
        elif op1 == 3 and op2 == 15:
            r[hi] = 0 if b == 0 else a // b
            return
        elif op1 == 4 and op2 == 0: v = s32(a) * s32(b) + sext((r[hi] << 32) | r[lo], 64)
        elif op1 == 6 and op2 == 0: v = a * b + ((r[hi] << 32) | r[lo])
        elif op1 == 6 and op2 == 6: v = a * b + r[hi] + r[lo]
        elif op1 == 4 and (op2 & 0xC) == 8:
            x = s16(a >> 16) if op2 & 2 else s16(a)
            y = s16(b >> 16) if op2 & 1 else s16(b)
            v = x * y + sext((r[hi] << 32) | r[lo], 64)
        elif op1 in (4, 5) and (op2 & 0xE) == 12:
            if op2 & 1: b = ((b >> 16) | (b << 16)) & M32
            p1, p2 = s16(a) * s16(b), s16(a >> 16) * s16(b >> 16)
            v = (p1 + p2 if op1 == 4 else p1 - p2) + sext((r[hi] << 32) | r[lo], 64)
        else:
            raise Unimplemented("long mul 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        r[lo], r[hi] = v & M32, (v >> 32) & M32

    # ---- floating point -----------------------------------------------------------------------------------------------
    def vfp(self, pc, hw1, hw2):
        r = self.r
        cp = (hw2 >> 8) & 15
        if cp not in (10, 11):
            raise Unimplemented("coprocessor %d at 0x%x" % (cp, pc))
        dbl = cp == 11
        D, Vn, Vd = (hw1 >> 6) & 1, hw1 & 15, (hw2 >> 12) & 15
        N, M, Vm = (hw2 >> 7) & 1, (hw2 >> 5) & 1, hw2 & 15
        if dbl: d, n, m = (D << 4) | Vd, (N << 4) | Vn, (M << 4) | Vm
        else: d, n, m = (Vd << 1) | D, (Vn << 1) | N, (Vm << 1) | M
        if (hw1 & 0xFF00) == 0xFE00:
            return self.vfp_v5(pc, hw1, hw2, dbl, d, n, m)
        if (hw1 & 0xEF00) == 0xEE00:
            if hw2 & 0x10:
                return self.vfp_transfer(pc, hw1, hw2)
            get, put = (self.fd, self.setfd) if dbl else (self.fs, self.sets)
            rnd = (lambda x: x) if dbl else f32_round
            fma = fma64 if dbl else fma32
            o1, o2, op = (hw1 >> 7) & 1, (hw1 >> 4) & 3, (hw2 >> 6) & 1
            if o1 == 0 and o2 == 0:                           # VMLA / VMLS: product rounded, then added
                p = rnd(get(n) * get(m))
                put(d, rnd(get(d) - p if op else get(d) + p))
            elif o1 == 0 and o2 == 1:                         # VNMLS (op 0): -d + n m ... ; VNMLA (op 1): -d - n m
                p = rnd(get(n) * get(m))
                put(d, rnd(-get(d) - p if op else -get(d) + p))
            elif o1 == 0 and o2 == 2:
                p = rnd(get(n) * get(m))
                put(d, -p if op else p)
            elif o1 == 0 and o2 == 3:
                put(d, rnd(get(n) - get(m) if op else get(n) + get(m)))
            elif o1 == 1 and o2 == 0 and op == 0:
                a, b = get(n), get(m)
                if b == 0.0:
                    q = math.nan if (a == 0.0 or a != a) else math.copysign(math.inf, a) * math.copysign(1.0, b)
                elif math.isinf(a) and math.isinf(b):
                    q = math.nan
                else:
                    q = a / b if dbl else round_div32(a, b)
                put(d, q)
            elif o1 == 1 and o2 == 1:                         # VFNMS (op 0): -d + n m ; VFNMA (op 1): -d - n m   (fused)
                put(d, fma(-get(n) if op else get(n), get(m), -get(d)))
            elif o1 == 1 and o2 == 2:                         # VFMA (op 0) / VFMS (op 1)
                put(d, fma(-get(n) if op else get(n), get(m), get(d)))
            elif o1 == 1 and o2 == 3:
                if op == 0:                                   # VMOV immediate
                    imm8 = ((hw1 & 15) << 4) | (hw2 & 15)
                    a, b, rest = (imm8 >> 7) & 1, (imm8 >> 6) & 1, imm8 & 0x3F
                    if dbl:
                        bits = (a << 63) | ((1 - b) << 62) | ((0xFF if b else 0) << 54) | (rest << 48)
                        self.setd(d, bits)
                    else:
                        self.s[d] = (a << 31) | ((1 - b) << 30) | ((0x1F if b else 0) << 25) | (rest << 19)
                    return
                opc2, b7 = hw1 & 15, (hw2 >> 7) & 1
                if opc2 == 0:
                    if b7: put(d, abs(get(m)))
                    elif dbl: self.setd(d, self.getd(m))
                    else: self.s[d] = self.s[m]
                elif opc2 == 1:
                    if b7:
                        x = get(m)
                        put(d, math.nan if x < 0 else (rnd(math.sqrt(x)) if not dbl else math.sqrt(x)))
                    elif dbl: self.setd(d, self.getd(m) ^ (1 << 63))
                    else: self.s[d] = self.s[m] ^ 0x80000000
                elif opc2 in (4, 5):
                    a = get(d)
                    b = 0.0 if opc2 == 5 else get(m)
                    if a != a or b != b: f = (0, 0, 1, 1)
                    elif a == b: f = (0, 1, 1, 0)
                    elif a < b: f = (1, 0, 0, 0)
                    else: f = (0, 0, 1, 0)
                    self.fn, self.fz, self.fc, self.fv = f
                elif opc2 == 7 and b7:                        # VCVT double <-> single
                    if dbl:                                   # source is double (sz = 1): Sd = Dm
                        sd = (Vd << 1) | D
                        self.sets(sd, f32_round(self.fd(m)))
                    else:
                        dd = (D << 4) | Vd
                        self.setfd(dd, self.fs(m))
                elif opc2 == 8:                               # VCVT from integer (source is always an S register)
                    sm = (Vm << 1) | M
                    iv = s32(self.s[sm]) if b7 else self.s[sm]
                    put(d, float(iv) if dbl else f32_round(float(iv)))
                elif opc2 in (12, 13):                        # VCVT to integer (destination is always an S register)
                    sd = (Vd << 1) | D
                    x = get(m)
                    signed = opc2 == 13
                    if x != x: iv = 0
                    else:
                        if b7: iv = math.trunc(x) if math.isfinite(x) else (1 << 40) * (1 if x > 0 else -1)
                        else:
                            if not math.isfinite(x): iv = (1 << 40) * (1 if x > 0 else -1)
                            else:
                                fl = math.floor(x); diff = x - fl
                                iv = fl + (1 if diff > 0.5 or (diff == 0.5 and (fl & 1)) else 0)
                        lo, hi = (-(1 << 31), (1 << 31) - 1) if signed else (0, (1 << 32) - 1)
                        iv = min(max(iv, lo), hi)
                    self.s[sd] = iv & M32
                elif opc2 in (10, 11, 14, 15):                # fixed point <-> float, in place
                    sx = (hw2 >> 7) & 1
                    size = 32 if sx else 16
                    fbits = size - (((hw2 & 15) << 1) | ((hw2 >> 5) & 1))
                    unsigned = opc2 & 1
                    if opc2 & 4:                              # to fixed (round toward zero)
                        x = get(d) * (2.0 ** fbits)
                        iv = 0 if x != x else (math.trunc(x) if math.isfinite(x) else (1 << 40) * (1 if x > 0 else -1))
                        lo, hi = (0, (1 << size) - 1) if unsigned else (-(1 << (size - 1)), (1 << (size - 1)) - 1)
                        iv = min(max(iv, lo), hi)
                        if dbl: raise Unimplemented("VCVT double to fixed at 0x%x" % pc)
                        self.s[d] = (sext(iv, size) & M32) if not unsigned else iv
                    else:
                        raw = (self.getd(d) if dbl else self.s[d]) & ((1 << size) - 1)
                        iv = raw if unsigned else sext(raw, size)
                        x = iv / (2.0 ** fbits)
                        put(d, x if dbl else f32_round(x))
                else:
                    raise Unimplemented("VFP other op opc2=%d b7=%d at 0x%x" % (opc2, b7, pc))
            else:
                raise Unimplemented("VFP dp 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
            return
        # ---- load / store / 64-bit transfers
        if (hw1 & 0xFE00) != 0xEC00:
            raise Unimplemented("VFP 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
        p, u, w, l, rn = (hw1 >> 8) & 1, (hw1 >> 7) & 1, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15
        imm8 = hw2 & 0xFF
        if p == 0 and u == 0 and w == 0:
            if not D:
                raise Unimplemented("VFP ldst 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))
            rt, rt2 = (hw2 >> 12) & 15, hw1 & 15
            if dbl:
                if l: r[rt], r[rt2] = self.s[2 * m], self.s[2 * m + 1]
                else: self.s[2 * m], self.s[2 * m + 1] = r[rt], r[rt2]
            else:
                if l: r[rt], r[rt2] = self.s[m], self.s[m + 1]
                else: self.s[m], self.s[m + 1] = r[rt], r[rt2]
            return
        if p == 1 and w == 0:                                 # VLDR / VSTR
            base = ((pc + 4) & ~3) if rn == 15 else r[rn]
            a = base + (imm8 << 2) if u else base - (imm8 << 2)
            if dbl:
                if l: self.setd(d, self.ld(a, 4) | (self.ld(a + 4, 4) << 32))
                else:
                    v = self.getd(d); self.st(a, 4, v & M32); self.st(a + 4, 4, v >> 32)
            else:
                if l: self.s[d] = self.ld(a, 4)
                else: self.st(a, 4, self.s[d])
            return
        if (p, u) in ((0, 1), (1, 0)):                        # VLDM / VSTM / VPUSH / VPOP
            nregs = imm8 // 2 if dbl else imm8
            nbytes = imm8 * 4
            a = r[rn] if u else r[rn] - nbytes
            if w: r[rn] = (r[rn] + nbytes if u else r[rn] - nbytes) & M32
            first = 2 * d if dbl else d
            for k in range(nregs * (2 if dbl else 1)):
                if l: self.s[first + k] = self.ld(a + 4 * k, 4)
                else: self.st(a + 4 * k, 4, self.s[first + k])
            return
        raise Unimplemented("VFP ldst 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))

    def vfp_transfer(self, pc, hw1, hw2):
        r = self.r
        rt = (hw2 >> 12) & 15
        l = (hw1 >> 4) & 1
        cp = (hw2 >> 8) & 15
        a = (hw1 >> 5) & 7
        if cp == 10 and a == 0:                               # VMOV Sn <-> Rt
            n = ((hw1 & 15) << 1) | ((hw2 >> 7) & 1)
            if l: r[rt] = self.s[n]
            else: self.s[n] = r[rt]
            return
        if cp == 10 and a == 7:                               # VMRS / VMSR
            if l:
                if rt == 15: self.n, self.z, self.c, self.v = self.fn, self.fz, self.fc, self.fv
                else: r[rt] = (self.fn << 31) | (self.fz << 30) | (self.fc << 29) | (self.fv << 28)
            return
        if cp == 11 and (a & 6) == 0 and (hw2 & 0x60) == 0:   # VMOV.32 Dd[x] <-> Rt
            dd = (((hw2 >> 7) & 1) << 4) | (hw1 & 15)
            x = (hw1 >> 5) & 1
            if l: r[rt] = self.s[2 * dd + x]
            else: self.s[2 * dd + x] = r[rt]
            return
        raise Unimplemented("VFP transfer 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))

    def vfp_v5(self, pc, hw1, hw2, dbl, d, n, m):
        get, put = (self.fd, self.setfd) if dbl else (self.fs, self.sets)
        if (hw1 & 0xFF80) == 0xFE00 and (hw2 & 0x50) == 0x00:  # VSEL
            cc = (hw1 >> 4) & 3
            cond = {0: self.z, 1: self.v, 2: self.n == self.v, 3: (not self.z) and self.n == self.v}[cc]
            src = n if cond else m
            if dbl: self.setd(d, self.getd(src))
            else: self.s[d] = self.s[src]
            return
        if (hw1 & 0xFFB0) == 0xFE80 and (hw2 & 0x10) == 0:    # VMAXNM / VMINNM
            a, b = get(n), get(m)
            if a != a: v = b
            elif b != b: v = a
            else: v = min(a, b) if hw2 & 0x40 else max(a, b)
            put(d, v)
            return
        if (hw1 & 0xFFBC) == 0xFEBC and (hw2 & 0x50) == 0x40:  # VCVTA / N / P / M to integer
            rm_ = hw1 & 3
            x = get(m)
            signed = (hw2 >> 7) & 1
            if x != x: iv = 0
            elif not math.isfinite(x): iv = (1 << 40) * (1 if x > 0 else -1)
            elif rm_ == 0: iv = math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)
            elif rm_ == 1:
                fl = math.floor(x); diff = x - fl
                iv = fl + (1 if diff > 0.5 or (diff == 0.5 and (fl & 1)) else 0)
            elif rm_ == 2: iv = math.ceil(x)
            else: iv = math.floor(x)
            lo, hi = (-(1 << 31), (1 << 31) - 1) if signed else (0, (1 << 32) - 1)
            sd = (((hw2 >> 12) & 15) << 1) | ((hw1 >> 6) & 1)
            self.s[sd] = min(max(iv, lo), hi) & M32
            return
        if (hw1 & 0xFFBC) == 0xFEB8 and (hw2 & 0x50) == 0x40:  # VRINTA / N / P / M
            rm_ = hw1 & 3
            x = get(m)
            if x != x or not math.isfinite(x): v = x
            elif rm_ == 0: v = math.copysign(math.floor(abs(x) + 0.5), x)
            elif rm_ == 1:
                fl = math.floor(x); diff = x - fl
                v = math.copysign(float(fl + (1 if diff > 0.5 or (diff == 0.5 and (int(fl) & 1)) else 0)), x)
            elif rm_ == 2: v = math.copysign(float(math.ceil(x)), x)
            else: v = math.copysign(float(math.floor(x)), x)
            put(d, v)
            return
        raise Unimplemented("FPv5 0x%04x 0x%04x at 0x%x" % (hw1, hw2, pc))


def round_div32(a, b):
    """float / float rounded once to float (the quotient in double and then to float is exact for 24-bit operands)"""
    return f32_round(a / b)
