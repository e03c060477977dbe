"""The reference's firmware image as a Teensy 4 lays it out at run time: the flash contents at 0x60000000, the code
the start-up routine copies to ITCM (address 0) and the initialised data it copies to DTCM (0x20000000), zeroed .bss,
a stack.  The three (destination, source, end) triples are read out of the reset handler's own literal pool.
Build container only (reads /root/reference)."""
import struct

from make_firmware_tables import HEX, read_ihex
from thumb_emu import Cpu, Memory

FLASH = 0x60000000
SCRATCH = 0x20200000            # OCRAM2 ("DMAMEM"): test buffers live here
SCRATCH_SIZE = 0x80000


class Image:
    def __init__(self):
        lo, img = read_ihex(HEX)
        assert lo == FLASH
        self.img = bytes(img)
        w = lambda o: struct.unpack_from("<I", self.img, o)[0]
        assert w(0x1000) == 0x402000D1                          # the boot ROM's image vector table
        entry = (w(0x1004) & ~1) - FLASH
        # ResetHandler: memory_copy(&_stext, &_stextload, &_etext); memory_copy(&_sdata, &_sdataload, &_edata); memory_clear(&_sbss, &_ebss)
        pool = [w(o) for o in range(entry, entry + 0x400, 4)]
        k = next(i for i in range(len(pool) - 8) if pool[i] == 0 and (pool[i + 1] >> 20) == 0x600 and pool[i + 3] == 0x20000000)
        self.stext, self.stextload, self.etext, self.sdata, self.sdataload, self.edata, self.sbss, self.ebss = pool[k:k + 8]
        assert self.sdataload - FLASH + (self.edata - self.sdata) == len(self.img)
        self.itcm_off = self.stextload - FLASH
        self.dtcm_off = self.sdataload - FLASH

    def itcm_of_offset(self, off):
        """hex-file offset (as tests/golden/make_firmware_tables.py counts them) -> ITCM address"""
        assert self.itcm_off <= off < self.itcm_off + self.etext
        return off - self.itcm_off

    def dtcm_of_offset(self, off):
        assert off >= self.dtcm_off
        return self.sdata + off - self.dtcm_off

    def machine(self):
        mem = Memory()
        mem.map(FLASH, self.img)
        itcm = mem.map(0, 0x80000)
        itcm[:self.etext] = self.img[self.itcm_off:self.itcm_off + self.etext]
        dtcm = mem.map(0x20000000, 0x80000)
        n = self.edata - self.sdata
        dtcm[:n] = self.img[self.dtcm_off:self.dtcm_off + n]
        mem.map(SCRATCH, SCRATCH_SIZE)
        cpu = Cpu(mem)
        cpu.r[13] = 0x20000000 + 0x80000 - 0x100                # top of DTCM, as the linker script puts it
        return cpu

    def bl_targets(self):
        """every BL in the ITCM code: {target address: number of call sites}"""
        code = self.img[self.itcm_off:self.itcm_off + self.etext]
        out = {}
        for o in range(0, len(code) - 4, 2):
            hw1, hw2 = struct.unpack_from("<HH", code, o)
            if (hw1 & 0xF800) == 0xF000 and (hw2 & 0xD000) == 0xD000:
                s, j1, j2 = (hw1 >> 10) & 1, (hw2 >> 13) & 1, (hw2 >> 11) & 1
                i1, i2 = 1 - (j1 ^ s), 1 - (j2 ^ s)
                imm = (s << 24) | (i1 << 23) | (i2 << 22) | ((hw1 & 0x3FF) << 12) | ((hw2 & 0x7FF) << 1)
                if imm & (1 << 24):
                    imm -= 1 << 25
                t = o + 4 + imm
                if 0 <= t < len(code):
                    out[t] = out.get(t, 0) + 1
        return out
