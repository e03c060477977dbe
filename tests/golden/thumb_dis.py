"""A reading aid for tests/golden/make_firmware_kat.py: one line per Thumb-2 instruction with the fields that matter
for finding a routine's arguments and the layout of the objects it touches (loads / stores with their offsets,
calls, branches, compares, immediates).  Anything else prints as its raw halfwords.  Not a complete disassembler."""
import struct


def _sx(v, b):
    v &= (1 << b) - 1
    return v - (1 << b) if v >> (b - 1) else v


def dis(code, a):
    """-> (length, text) for the instruction at byte offset a of `code` (ITCM image: offset = address)"""
    hw = struct.unpack_from("<H", code, a)[0]
    top = hw >> 11
    if top >= 0b11101:
        hw2 = struct.unpack_from("<H", code, a + 2)[0]
        return 4, _dis32(a, hw, hw2)
    R = lambda n: "r%d" % n
    if top < 3:
        return 2, "%s %s, %s, #%d" % (("lsls", "lsrs", "asrs")[top], R(hw & 7), R((hw >> 3) & 7), (hw >> 6) & 31)
    if top == 3:
        x = (hw >> 6) & 7
        return 2, "%s %s, %s, %s" % ("subs" if hw & 0x200 else "adds", R(hw & 7), R((hw >> 3) & 7), ("#%d" % x) if hw & 0x400 else R(x))
    if top < 8:
        return 2, "%s %s, #%d" % (("movs", "cmp", "adds", "subs")[top - 4], R((hw >> 8) & 7), hw & 0xFF)
    if top == 8:
        if not hw & 0x400:
            n = ("ands", "eors", "lsls", "lsrs", "asrs", "adcs", "sbcs", "rors", "tst", "rsbs", "cmp", "cmn", "orrs", "muls", "bics", "mvns")[(hw >> 6) & 15]
            return 2, "%s %s, %s" % (n, R(hw & 7), R((hw >> 3) & 7))
        op, rm, rd = (hw >> 8) & 3, (hw >> 3) & 15, (hw & 7) | ((hw >> 4) & 8)
        if op == 3:
            return 2, "%s %s" % ("blx" if hw & 0x80 else "bx", R(rm))
        return 2, "%s %s, %s" % (("add", "cmp", "mov")[op], R(rd), R(rm))
    if top == 9:
        t = ((a + 4) & ~3) + ((hw & 0xFF) << 2)
        return 2, "ldr %s, [pc -> 0x%x] = 0x%08x" % (R((hw >> 8) & 7), t, struct.unpack_from("<I", code, t)[0])
    if top in (10, 11):
        n = ("str", "strh", "strb", "ldrsb", "ldr", "ldrh", "ldrb", "ldrsh")[(hw >> 9) & 7]
        return 2, "%s %s, [%s, %s]" % (n, R(hw & 7), R((hw >> 3) & 7), R((hw >> 6) & 7))
    if 12 <= top <= 17:
        size = 4 if top < 14 else (1 if top < 16 else 2)
        n = ("str", "ldr", "strb", "ldrb", "strh", "ldrh")[top - 12]
        return 2, "%s %s, [%s, #%d]" % (n, R(hw & 7), R((hw >> 3) & 7), ((hw >> 6) & 31) * size)
    if top in (18, 19):
        return 2, "%s %s, [sp, #%d]" % ("ldr" if top & 1 else "str", R((hw >> 8) & 7), (hw & 0xFF) << 2)
    if top == 20:
        return 2, "adr %s, 0x%x" % (R((hw >> 8) & 7), ((a + 4) & ~3) + ((hw & 0xFF) << 2))
    if top == 21:
        return 2, "add %s, sp, #%d" % (R((hw >> 8) & 7), (hw & 0xFF) << 2)
    if top in (22, 23):
        k = hw & 0xFF00
        if k == 0xB000:
            return 2, "%s sp, #%d" % ("sub" if hw & 0x80 else "add", (hw & 0x7F) << 2)
        if k in (0xB100, 0xB300, 0xB900, 0xBB00):
            off = (((hw >> 9) & 1) << 6) | (((hw >> 3) & 31) << 1)
            return 2, "%s %s, 0x%x" % ("cbnz" if hw & 0x800 else "cbz", R(hw & 7), a + 4 + off)
        if k in (0xB400, 0xB500):
            return 2, "push {%s%s}" % (",".join(R(i) for i in range(8) if hw & (1 << i)), ",lr" if hw & 0x100 else "")
        if k in (0xBC00, 0xBD00):
            return 2, "pop {%s%s}" % (",".join(R(i) for i in range(8) if hw & (1 << i)), ",pc" if hw & 0x100 else "")
        if k == 0xB200:
            return 2, "%s %s, %s" % (("sxth", "sxtb", "uxth", "uxtb")[(hw >> 6) & 3], R(hw & 7), R((hw >> 3) & 7))
        if k == 0xBF00:
            return 2, ("it 0x%02x" % (hw & 0xFF)) if hw & 0xF else "nop"
        return 2, "misc16 0x%04x" % hw
    if top in (24, 25):
        return 2, "%s %s!, {%s}" % ("ldmia" if top & 1 else "stmia", R((hw >> 8) & 7), ",".join(R(i) for i in range(8) if hw & (1 << i)))
    if top in (26, 27):
        c = (hw >> 8) & 15
        cc = ("eq", "ne", "cs", "cc", "mi", "pl", "vs", "vc", "hi", "ls", "ge", "lt", "gt", "le", "udf", "svc")[c]
        return 2, "b%s 0x%x" % (cc, a + 4 + (_sx(hw & 0xFF, 8) << 1))
    if top == 28:
        return 2, "b 0x%x" % (a + 4 + (_sx(hw & 0x7FF, 11) << 1))
    return 2, "?16 0x%04x" % hw


def _dis32(a, hw1, hw2):
    R = lambda n: ("sp", "lr", "pc")[n - 13] if n >= 13 else "r%d" % n
    if (hw1 & 0xF800) == 0xF000 and (hw2 & 0x8000):
        s, j1, j2 = (hw1 >> 10) & 1, (hw2 >> 13) & 1, (hw2 >> 11) & 1
        if (hw2 & 0x5000) == 0:
            c = (hw1 >> 6) & 15
            if c < 14:
                imm = _sx((s << 20) | (j2 << 19) | (j1 << 18) | ((hw1 & 63) << 12) | ((hw2 & 0x7FF) << 1), 21)
                cc = ("eq", "ne", "cs", "cc", "mi", "pl", "vs", "vc", "hi", "ls", "ge", "lt", "gt", "le")[c]
                return "b%s.w 0x%x" % (cc, a + 4 + imm)
            return "sys 0x%04x 0x%04x" % (hw1, hw2)
        i1, i2 = 1 - (j1 ^ s), 1 - (j2 ^ s)
        imm = _sx((s << 24) | (i1 << 23) | (i2 << 22) | ((hw1 & 0x3FF) << 12) | ((hw2 & 0x7FF) << 1), 25)
        return "%s 0x%x" % ("bl" if hw2 & 0x4000 else "b.w", a + 4 + imm)
    if (hw1 & 0xFE00) == 0xF800:
        size, l, sign, rn, rt = (hw1 >> 5) & 3, (hw1 >> 4) & 1, (hw1 >> 8) & 1, hw1 & 15, (hw2 >> 12) & 15
        n = ("ldr" if l else "str") + ("s" if sign else "") + ("b", "h", "", "?")[size]
        if rn == 15:
            if _CODE is not None and size == 2 and l:
                t = ((a + 4) & ~3) + (hw2 & 0xFFF) * (1 if hw1 & 0x80 else -1)
                return "%s %s, [pc -> 0x%x] = 0x%08x" % (n, R(rt), t, struct.unpack_from("<I", _CODE, t)[0])
            return "%s %s, [pc, #%s%d]" % (n, R(rt), "" if hw1 & 0x80 else "-", hw2 & 0xFFF)
        if hw1 & 0x80:
            return "%s.w %s, [%s, #%d]" % (n, R(rt), R(rn), hw2 & 0xFFF)
        if hw2 & 0x800:
            p, u, w, imm = (hw2 >> 10) & 1, (hw2 >> 9) & 1, (hw2 >> 8) & 1, hw2 & 0xFF
            sg = "" if u else "-"
            if p and not w: return "%s %s, [%s, #%s%d]" % (n, R(rt), R(rn), sg, imm)
            if p: return "%s %s, [%s, #%s%d]!" % (n, R(rt), R(rn), sg, imm)
            return "%s %s, [%s], #%s%d" % (n, R(rt), R(rn), sg, imm)
        return "%s %s, [%s, %s, lsl #%d]" % (n, R(rt), R(rn), R(hw2 & 15), (hw2 >> 4) & 3)
    if (hw1 & 0xFE40) == 0xE800:
        op, w, l, rn = (hw1 >> 7) & 3, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15
        regs = ",".join(R(i) for i in range(16) if hw2 & (1 << i))
        return "%s%s %s%s, {%s}" % ("ldm" if l else "stm", ("?", "ia", "db", "?")[op], R(rn), "!" if w else "", regs)
    if (hw1 & 0xFE40) == 0xE840:
        p, u, w, l, rn = (hw1 >> 8) & 1, (hw1 >> 7) & 1, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15
        if (hw1 & 0xFFF0) == 0xE8D0 and (hw2 & 0xFFE0) == 0xF000:
            return "tb%s [%s, %s]" % ("h" if hw2 & 0x10 else "b", R(rn), R(hw2 & 15))
        return "%s %s, %s, [%s, #%s%d]%s" % ("ldrd" if l else "strd", R((hw2 >> 12) & 15), R((hw2 >> 8) & 15), R(rn), "" if u else "-", (hw2 & 0xFF) << 2,
                                             "" if p and not w else ("!" if p else " (post)"))
    if (hw1 & 0xFBE0) == 0xF240 or (hw1 & 0xFBF0) == 0xF2C0:
        imm = ((hw1 & 15) << 12) | (((hw1 >> 10) & 1) << 11) | (((hw2 >> 12) & 7) << 8) | (hw2 & 0xFF)
        return "%s %s, #0x%x" % ("movt" if hw1 & 0x80 else "movw", R((hw2 >> 8) & 15), imm)
    if (hw1 & 0xEC00) == 0xEC00:
        cp = (hw2 >> 8) & 15
        if (hw1 & 0xEF00) == 0xED00 and not (hw1 & 0x20):
            l, u, rn = (hw1 >> 4) & 1, (hw1 >> 7) & 1, hw1 & 15
            D, Vd = (hw1 >> 6) & 1, (hw2 >> 12) & 15
            reg = ("d%d" % ((D << 4) | Vd)) if cp == 11 else ("s%d" % ((Vd << 1) | D))
            if rn == 15 and _CODE is not None:
                t = ((a + 4) & ~3) + ((hw2 & 0xFF) << 2) * (1 if u else -1)
                v = struct.unpack_from("<d" if cp == 11 else "<f", _CODE, t)[0]
                return "vldr %s, [pc -> 0x%x] = %.17g" % (reg, t, v)
            return "%s %s, [%s, #%s%d]" % ("vldr" if l else "vstr", reg, R(rn), "" if u else "-", (hw2 & 0xFF) << 2)
        return _vfp(a, hw1, hw2)
    if (hw1 & 0xFA00) == 0xF000 and not (hw2 & 0x8000):
        op, s, rn, rd = (hw1 >> 5) & 15, (hw1 >> 4) & 1, hw1 & 15, (hw2 >> 8) & 15
        imm12 = (((hw1 >> 10) & 1) << 11) | (((hw2 >> 12) & 7) << 8) | (hw2 & 0xFF)
        if (imm12 >> 10) == 0:
            k, b = (imm12 >> 8) & 3, imm12 & 0xFF
            v = (b, (b << 16) | b, (b << 24) | (b << 8), (b << 24) | (b << 16) | (b << 8) | b)[k]
        else:
            un, rot = 0x80 | (imm12 & 0x7F), imm12 >> 7
            v = ((un >> rot) | (un << (32 - rot))) & 0xFFFFFFFF
        n = {0: "and", 1: "bic", 2: "orr", 3: "orn", 4: "eor", 8: "add", 10: "adc", 11: "sbc", 13: "sub", 14: "rsb"}.get(op, "dp%d" % op)
        if rd == 15 and s: n = {0: "tst", 4: "teq", 8: "cmn", 13: "cmp"}.get(op, n)
        if rn == 15 and op == 2: return "mov%s %s, #0x%x" % ("s" if s else "", R(rd), v)
        if rn == 15 and op == 3: return "mvn%s %s, #0x%x" % ("s" if s else "", R(rd), v)
        if rd == 15 and s: return "%s %s, #0x%x" % (n, R(rn), v)
        return "%s%s %s, %s, #0x%x" % (n, "s" if s else "", R(rd), R(rn), v)
    if (hw1 & 0xFB50) in (0xF200, 0xF2A0 & 0xFB50) and not (hw2 & 0x8000) and ((hw1 >> 4) & 31) in (0, 10):
        imm12 = (((hw1 >> 10) & 1) << 11) | (((hw2 >> 12) & 7) << 8) | (hw2 & 0xFF)
        return "%s %s, %s, #%d" % ("subw" if (hw1 >> 4) & 31 == 10 else "addw", R((hw2 >> 8) & 15), R(hw1 & 15), imm12)
    if (hw1 & 0xFE00) == 0xEA00:
        op, s, rn, rd, rm = (hw1 >> 5) & 15, (hw1 >> 4) & 1, hw1 & 15, (hw2 >> 8) & 15, hw2 & 15
        imm5, typ = (((hw2 >> 12) & 7) << 2) | ((hw2 >> 6) & 3), (hw2 >> 4) & 3
        n = {0: "and", 1: "bic", 2: "orr", 3: "orn", 4: "eor", 6: "pkh", 8: "add", 10: "adc", 11: "sbc", 13: "sub", 14: "rsb"}.get(op, "dp%d" % op)
        sh = ", %s #%d" % (("lsl", "lsr", "asr", "ror")[typ], imm5) if imm5 or typ else ""
        if rn == 15 and op == 2: return "mov%s.w %s, %s%s" % ("s" if s else "", R(rd), R(rm), sh)
        if rd == 15 and s: return "%s.w %s, %s%s" % ({0: "tst", 4: "teq", 8: "cmn", 13: "cmp"}.get(op, n), R(rn), R(rm), sh)
        return "%s%s.w %s, %s, %s%s" % (n, "s" if s else "", R(rd), R(rn), R(rm), sh)
    if (hw1 & 0xFFD0) in (0xF3C0, 0xF340) and not (hw2 & 0x8000):
        lsb, w = (((hw2 >> 12) & 7) << 2) | ((hw2 >> 6) & 3), (hw2 & 31) + 1
        return "%s %s, %s, #%d, #%d" % ("ubfx" if hw1 & 0x80 else "sbfx", R((hw2 >> 8) & 15), R(hw1 & 15), lsb, w)
    if (hw1 & 0xFFF0) == 0xFAB0 and (hw2 & 0xF0F0) == 0xF080:
        return "clz %s, %s" % (R((hw2 >> 8) & 15), R(hw2 & 15))
    if (hw1 & 0xFF80) == 0xFA00 and (hw2 & 0xF0F0) == 0xF000:
        return "%s%s.w %s, %s, %s" % (("lsl", "lsr", "asr", "ror")[(hw1 >> 5) & 3], "s" if hw1 & 0x10 else "", R((hw2 >> 8) & 15), R(hw1 & 15), R(hw2 & 15))
    if (hw1 & 0xFF80) == 0xFB00:
        op1, op2, ra = (hw1 >> 4) & 7, (hw2 >> 4) & 15, (hw2 >> 12) & 15
        if op1 == 0:
            n = "mls" if op2 == 1 else ("mul" if ra == 15 else "mla")
            return "%s %s, %s, %s%s" % (n, R((hw2 >> 8) & 15), R(hw1 & 15), R(hw2 & 15), "" if ra == 15 else ", " + R(ra))
        return "mul32 op1=%d op2=%d %s, %s, %s, ra=%s" % (op1, op2, R((hw2 >> 8) & 15), R(hw1 & 15), R(hw2 & 15), R(ra))
    if (hw1 & 0xFF80) == 0xFB80:
        n = {0: "smull", 1: "sdiv", 2: "umull", 3: "udiv", 4: "smlal", 6: "umlal"}.get((hw1 >> 4) & 7, "mul64?")
        return "%s %s, %s, %s, %s" % (n, R((hw2 >> 12) & 15), R((hw2 >> 8) & 15), R(hw1 & 15), R(hw2 & 15))
    return "?32 0x%04x 0x%04x" % (hw1, hw2)


def _vfp(a, hw1, hw2):
    """the VFP / FPv5 encodings the image uses (data processing, transfers, multiple loads and stores)"""
    R = lambda n: ("sp", "lr", "pc")[n - 13] if n >= 13 else "r%d" % n
    cp = (hw2 >> 8) & 15
    dbl = cp == 11
    D, Vn, Vd = (hw1 >> 6) & 1, hw1 & 15, (hw2 >> 12) & 15
    N, M, Vm = (hw2 >> 7) & 1, (hw2 >> 5) & 1, hw2 & 15
    if dbl: d, n, m = "d%d" % ((D << 4) | Vd), "d%d" % ((N << 4) | Vn), "d%d" % ((M << 4) | Vm)
    else: d, n, m = "s%d" % ((Vd << 1) | D), "s%d" % ((Vn << 1) | N), "s%d" % ((Vm << 1) | M)
    sd, sm = "s%d" % ((Vd << 1) | D), "s%d" % ((Vm << 1) | M)
    sz = ".f64" if dbl else ".f32"
    if (hw1 & 0xFF00) == 0xFE00:
        if (hw1 & 0xFF80) == 0xFE00 and (hw2 & 0x50) == 0:
            return "vsel%s%s %s, %s, %s" % (("eq", "vs", "ge", "gt")[(hw1 >> 4) & 3], sz, d, n, m)
        if (hw1 & 0xFFB0) == 0xFE80 and (hw2 & 0x10) == 0:
            return "%s%s %s, %s, %s" % ("vminnm" if hw2 & 0x40 else "vmaxnm", sz, d, n, m)
        if (hw1 & 0xFFBC) == 0xFEBC and (hw2 & 0x50) == 0x40:
            return "vcvt%s.%s32%s %s, %s" % ("anpm"[hw1 & 3], "s" if hw2 & 0x80 else "u", sz, sd, m)
        if (hw1 & 0xFFBC) == 0xFEB8 and (hw2 & 0x50) == 0x40:
            return "vrint%s%s %s, %s" % ("anpm"[hw1 & 3], sz, d, m)
        return "fpv5 0x%04x 0x%04x" % (hw1, hw2)
    if (hw1 & 0xEF00) == 0xEE00:
        if hw2 & 0x10:
            rt, l, k = (hw2 >> 12) & 15, (hw1 >> 4) & 1, (hw1 >> 5) & 7
            if cp == 10 and k == 0:
                return ("vmov %s, s%d" if l else "vmov s%d, %s") % ((R(rt), (Vn << 1) | N) if l else ((Vn << 1) | N, R(rt)))
            if cp == 10 and k == 7:
                return ("vmrs %s, fpscr" % ("APSR_nzcv" if rt == 15 else R(rt))) if l else "vmsr fpscr, %s" % R(rt)
            if cp == 11:
                dd, x = (N << 4) | Vn, (hw1 >> 5) & 1
                return ("vmov %s, d%d[%d]" % (R(rt), dd, x)) if l else ("vmov d%d[%d], %s" % (dd, x, R(rt)))
            return "vfp-xfer 0x%04x 0x%04x" % (hw1, hw2)
        o1, o2, op = (hw1 >> 7) & 1, (hw1 >> 4) & 3, (hw2 >> 6) & 1
        if (o1, o2) != (1, 3):
            nm = {(0, 0): ("vmla", "vmls"), (0, 1): ("vnmls", "vnmla"), (0, 2): ("vmul", "vnmul"), (0, 3): ("vadd", "vsub"),
                  (1, 0): ("vdiv", "vdiv?"), (1, 1): ("vfnms", "vfnma"), (1, 2): ("vfma", "vfms")}[(o1, o2)][op]
            return "%s%s %s, %s, %s" % (nm, sz, d, n, m)
        if op == 0:
            imm8 = ((hw1 & 15) << 4) | (hw2 & 15)
            sg, b, rest = (imm8 >> 7) & 1, (imm8 >> 6) & 1, imm8 & 0x3F
            bits = (sg << 31) | ((1 - b) << 30) | ((0x1F if b else 0) << 25) | (rest << 19)
            return "vmov%s %s, #%g" % (sz, d, struct.unpack("<f", struct.pack("<I", bits))[0])
        opc2, b7 = hw1 & 15, (hw2 >> 7) & 1
        if opc2 == 0: return "%s%s %s, %s" % ("vabs" if b7 else "vmov", sz, d, m)
        if opc2 == 1: return "%s%s %s, %s" % ("vsqrt" if b7 else "vneg", sz, d, m)
        if opc2 in (4, 5): return "vcmp%s%s %s, %s" % ("e" if b7 else "", sz, d, "#0" if opc2 == 5 else m)
        if opc2 == 7 and b7:
            return ("vcvt.f32.f64 %s, %s" % (sd, m)) if dbl else ("vcvt.f64.f32 d%d, %s" % ((D << 4) | Vd, m))
        if opc2 == 8: return "vcvt%s.%s32 %s, %s" % (sz, "s" if b7 else "u", d, sm)
        if opc2 in (12, 13): return "vcvt%s.%s32%s %s, %s" % ("" if b7 else "r", "s" if opc2 == 13 else "u", sz, sd, m)
        if opc2 in (10, 11, 14, 15):
            size = 32 if b7 else 16
            fb = size - (((hw2 & 15) << 1) | ((hw2 >> 5) & 1))
            t = "%s%d" % ("u" if opc2 & 1 else "s", size)
            return ("vcvt.%s%s %s, %s, #%d" % (t, sz, d, d, fb)) if opc2 & 4 else ("vcvt%s.%s %s, %s, #%d" % (sz, t, d, d, fb))
        return "vfp-other 0x%04x 0x%04x" % (hw1, hw2)
    p, u, w, l, rn, imm8 = (hw1 >> 8) & 1, (hw1 >> 7) & 1, (hw1 >> 5) & 1, (hw1 >> 4) & 1, hw1 & 15, hw2 & 0xFF
    if p == 0 and u == 0 and w == 0:
        rt, rt2 = (hw2 >> 12) & 15, hw1 & 15
        regs = ("d%d" % ((M << 4) | Vm)) if dbl else ("s%d, s%d" % ((Vm << 1) | M, ((Vm << 1) | M) + 1))
        return ("vmov %s, %s, %s" % (R(rt), R(rt2), regs)) if l else ("vmov %s, %s, %s" % (regs, R(rt), R(rt2)))
    nregs = imm8 // 2 if dbl else imm8
    first = ((D << 4) | Vd) if dbl else ((Vd << 1) | D)
    pre = "d" if dbl else "s"
    lst = "{%s%d-%s%d}" % (pre, first, pre, first + nregs - 1)
    if rn == 13 and w and ((l and u) or (not l and not u)):
        return "%s %s" % ("vpop" if l else "vpush", lst)
    return "%s%s %s%s, %s" % ("vldm" if l else "vstm", "ia" if u else "db", R(rn), "!" if w else "", lst)


_CODE = None


def listing(code, start, end):
    global _CODE
    _CODE = code
    a = start
    out = []
    while a < end:
        n, t = dis(code, a)
        out.append("%6x: %s" % (a, t))
        a += n
    return "\n".join(out)
