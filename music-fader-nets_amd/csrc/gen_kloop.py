#!/usr/bin/env python3
"""Generates kloop_asm.h: the K loops of the 128/64-row weight-stationary GRU scans (gru_persist.hip) at H = 512 as ONE hand-placed
asm statement each.

Why: with one wavefront per SIMD nothing hides an instruction that sits between two MFMAs unless it is ALONE there - hipcc gathers the
operand loads, LDS fragment reads and their address arithmetic at the chunk boundaries (6 ds_read_b128 + 14 scalar ops in a row), and
the in-kernel stamps put the forward K loop at 34.8-37 cycles per v_mfma_f32_16x16x4_f32 (32 is the pipe's rate) and the backward at 42.8.
Here every memory instruction gets its own MFMA shadow, addresses are immediates off a scalar base that advances twice per group, and
nothing is requested beyond the end of K.

Pipeline unit = 16 K values of the wave's 2 row tiles ("half chunk"): 2 operand loads (1 KB each) from the exchange slab, NB fragment
reads from the LDS-resident weight slice, 24 (forward: 2 row tiles x 3 gate tiles x 4 k-steps) or 8 (backward: 2 x 1 x 4) MFMAs.
RU - 1 units are in flight (a ring of RU x 2 x 4 registers); during unit u the registers of unit u-1 are re-filled with unit u+RU-1 and
the weight fragments of unit u+1 are read (double buffer).  Everything lives in AGPRs (MFMA, VMEM and DS instructions of gfx950 take them
directly), so the compiler's registers are untouched; the accumulation order of every accumulator equals the C++ loop's: bit-identical.

Register maps (a = AGPR):
  forward : acc[m][n] a[4(3m+n)..] (a0-23), ring[slot][m] a[24+4(2slot+m)..] (a24-87, 8 slots), wfrag[bs][n] a[88+4(3bs+n)..] (a88-111)
  backward: acc[m][par] a[4(2m+par)..] (a0-15), ring[slot][m] a[16+4(2slot+m)..] (a16-143, 16 slots), wfrag[bs] a[144+4bs..] (a144-151)
Scalars: s[SB:SB+1] = operand base (advanced in 4 KB steps; 12-bit immediates reach the rest), s[SB+2] = group counter.
"""
import sys

SB = 84          # scalar temporaries s84..s86 (clobbered)


class Gen:
    def __init__(self, name, fwd):
        self.name, self.fwd = name, fwd
        self.RU = 8 if fwd else 16
        self.NB = 3 if fwd else 1
        self.acc0 = 0
        self.ring0 = 24 if fwd else 16
        self.w0 = self.ring0 + self.RU * 8
        self.nagpr = self.w0 + 2 * self.NB * 4
        self.lines = []
        self.s_rel = 0           # what s[SB:SB+1] points at, relative to the first unit of the current group (bytes)

    def e(self, s):
        self.lines.append(s)

    def ring(self, slot, m):
        return self.ring0 + 4 * (2 * slot + m)

    def wf(self, bs, n):
        return self.w0 + 4 * (self.NB * bs + n)

    def acc(self, m, x):
        return 4 * ((3 if self.fwd else 2) * m + x)

    def mfmas(self, slot, bs):
        out = []
        for jj in range(4):
            for m in range(2):
                for x in (range(3) if self.fwd else [jj & 1]):
                    a = self.ring(slot, m) + jj
                    b = (self.wf(bs, x) if self.fwd else self.wf(bs, 0)) + jj
                    c = self.acc(m, x)
                    out.append("v_mfma_f32_16x16x4_f32 a[%d:%d], a%d, a%d, a[%d:%d]" % (c, c + 3, a, b, c, c + 3))
        return out

    def advance_to(self, need_rel):
        """scalar adds so that need_rel - s_rel fits a non-negative 12-bit immediate"""
        out = []
        while need_rel - self.s_rel > 3072:
            out += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            self.s_rel += 4096
        assert 0 <= need_rel - self.s_rel <= 3072, (need_rel, self.s_rel)
        return out

    def gload(self, slot, m, need_rel):
        imm = need_rel - self.s_rel
        assert 0 <= imm <= 4095
        r = self.ring(slot, m)
        return "global_load_dwordx4 a[%d:%d], %%[vo%d], s[%d:%d] offset:%d sc1" % (r, r + 3, m, SB, SB + 1, imm)

    def wread(self, bs, n, unit_in_group):
        """weight fragment of unit (group start + unit_in_group), gate tile n"""
        r = self.wf(bs, n)
        if self.fwd:      # wl[3][nk][2][64][4] floats, gate stride 32 KB at H = 512: gates 0, 1 off %[lp], gate 2 off %[lq] = lp + 64 KB
            ptr, off = ("%[lp]", n * 32768 + unit_in_group * 1024) if n < 2 else ("%[lq]", unit_in_group * 1024)
        else:
            ptr, off = "%[lp]", unit_in_group * 1024
        assert off < 65536
        return "ds_read_b128 a[%d:%d], %s offset:%d" % (r, r + 3, ptr, off)

    def unit_fill(self, k, refill, wnext):
        """memory instructions of pipeline unit k (= slot = unit index inside the group): MFMA index -> instructions issued right behind it"""
        RU = self.RU
        fill = {}
        if refill:
            need = (k + RU - 1) * 1024
            assert not self.advance_to(need), "the scalar base must have been advanced in the previous unit"
            fill.setdefault(0, []).append(self.gload((k - 1) % RU, 0, need))
            fill.setdefault(6 if self.fwd else 4, []).append(self.gload((k - 1) % RU, 1, need))
        if wnext:
            slots = [3, 9, 12] if self.fwd else [2]
            for n in range(self.NB):
                fill.setdefault(slots[n], []).append(self.wread((k + 1) & 1, n, k + 1))
        return fill

    def unit(self, k, vm, fill, tail_ops):
        mf = self.mfmas(k, k & 1)
        t_tail = 15 if self.fwd else 6
        ops = list(tail_ops)
        # one scalar / vector bookkeeping instruction per MFMA gap from t_tail on
        self.e("s_waitcnt vmcnt(%d)" % vm)
        self.e("s_waitcnt lgkmcnt(0)")
        for t, ins in enumerate(mf):
            self.e(ins)
            for x in fill.get(t, []):
                self.e(x)
            if t >= t_tail and ops:
                # keep s_add / s_addc pairs together (SCC)
                self.e(ops.pop(0))
                if ops and ops[0].startswith("s_addc"):
                    self.e(ops.pop(0))
        for x in ops:
            self.e(x)

    def group(self, final):
        RU = self.RU
        start_rel = (RU - 1) * 1024 - 3072
        assert self.s_rel == start_rel
        for k in range(RU):
            refill = (not final) or k == 0
            wnext = (not final) or k < RU - 1
            if final:
                newer = min(RU - 2, RU - 1 - k)
            else:
                newer = RU - 2
            fill = self.unit_fill(k, refill, wnext)
            tail = []
            # prepare the scalar base for the next unit's re-fill
            nxt_refill = (not final) and k + 1 < RU
            if nxt_refill:
                tail += self.advance_to((k + 1 + RU - 1) * 1024)
            if k == RU - 1 and not final:
                # end of group: base to (next group start) + start_rel, LDS pointers to the next group, counter
                tail += self.advance_to(RU * 1024 + start_rel + 3072)     # forces s_rel >= RU*1024 + start_rel
                assert self.s_rel == RU * 1024 + start_rel, (self.s_rel, RU * 1024 + start_rel)
                tail.append("v_add_u32 %%[lp], 0x%x, %%[lp]" % (RU * 1024))
                if self.fwd:
                    tail.append("v_add_u32 %%[lq], 0x%x, %%[lq]" % (RU * 1024))
            self.unit(k, 2 * newer, fill, tail)
        if not final:
            self.s_rel -= RU * 1024

    def build(self):
        RU, e = self.RU, self.e
        e("s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1))
        e("s_mov_b32 s%d, %%[ngrp]" % (SB + 2))
        # prologue: units 0 .. RU-2 of the operand (unit RU-1 is the "re-fill" of unit 0), weight fragments of unit 0
        self.s_rel = 0
        for n in range(self.NB):
            e(self.wread(0, n, 0))
        for u in range(RU - 1):
            for x in self.advance_to(u * 1024):
                e(x)
            for m in range(2):
                e(self.gload(u, m, u * 1024))
        start_rel = (RU - 1) * 1024 - 3072
        while self.s_rel < start_rel:
            e("s_add_u32 s%d, s%d, 0x1000" % (SB, SB))
            e("s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1))
            self.s_rel += 4096
        assert self.s_rel == start_rel
        for c in range(0, (24 if self.fwd else 16), 4):
            # accumulators = 0 (v_accvgpr_write takes an inline constant)
            for i in range(4):
                e("v_accvgpr_write_b32 a%d, 0" % (c + i))
        e("s_sub_u32 s%d, s%d, 1" % (SB + 2, SB + 2))
        e("s_cmp_eq_u32 s%d, 0" % (SB + 2))
        e("s_cbranch_scc1 .Lfinal_%s_%%=" % self.name)
        e(".Lloop_%s_%%=:" % self.name)
        self.group(final=False)
        e("s_sub_u32 s%d, s%d, 1" % (SB + 2, SB + 2))
        e("s_cmp_eq_u32 s%d, 0" % (SB + 2))
        e("s_cbranch_scc0 .Lloop_%s_%%=" % self.name)
        e(".Lfinal_%s_%%=:" % self.name)
        self.group(final=True)
        # accumulators -> LDS (padded MFMA C layout, see gru_persist.hip (d)); MFMA results need 18 wait states before a DS read of them
        e("s_nop 15")
        e("s_nop 7")
        if self.fwd:
            for m in range(2):
                for n in range(3):
                    c = self.acc(m, n)
                    e("ds_write_b128 %%[red], a[%d:%d] offset:%d" % (c, c + 3, (m * 3 + n) * 1088))
        else:
            # the two k-parity accumulators of a row tile go to two planes of 8 tiles (WK x EM = 8 in both tilings); the epilogue adds them (= acc[m][0] + acc[m][1])
            for m in range(2):
                for par in range(2):
                    c = self.acc(m, par)
                    e("ds_write_b128 %%[red], a[%d:%d] offset:%d" % (c, c + 3, m * 1088 + par * 8 * 1088))
        e("s_waitcnt lgkmcnt(0)")

    def emit(self):
        self.build()
        body = "\n".join('        "%s\\n\\t"' % l for l in self.lines)
        clob = ", ".join('"a%d"' % i for i in range(self.nagpr))
        lq = ', [lq] "+v"(lq)' if self.fwd else ""
        lqarg = ", unsigned lq" if self.fwd else ""
        return """
// %s K loop of one time step: xin = this wave's first operand unit (uniform), vo0 / vo1 = byte offsets of its two row tiles (+ lane * 16),
// lp%s = LDS byte address of its first weight-fragment unit (+ lane * 16), ngrp = K units / %d (>= 1), red = LDS byte address of its
// accumulator tiles.  %d MFMAs per unit; %d AGPRs.
FN_DEVINL void %s(const float* xin_, unsigned vo0, unsigned vo1, unsigned lp%s, int ngrp_, unsigned red) {
    // wave-uniform by construction (the K range depends on the wave id only); the compiler cannot see that
    const unsigned long long xa = (unsigned long long)(uintptr_t)xin_;
    const float* xin = reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xa >> 32)) << 32) |
                                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(xa & 0xffffffffull))));
    const int ngrp = __builtin_amdgcn_readfirstlane(ngrp_);
    asm volatile(
%s
        : [lp] "+v"(lp)%s
        : [xin] "s"(xin), [vo0] "v"(vo0), [vo1] "v"(vo1), [ngrp] "s"(ngrp), [red] "v"(red)
        : "memory", "scc", "s%d", "s%d", "s%d", %s);
}
""" % ("forward" if self.fwd else "backward", " / lq" if self.fwd else "", self.RU, 24 if self.fwd else 8, self.nagpr, self.name, lqarg,
       body, lq, SB, SB + 1, SB + 2, clob)


def main(path):
    out = ["// GENERATED by gen_kloop.py - do not edit.  Hand-placed K loops of the weight-stationary GRU scans (H = 512, 2 row tiles per wave).",
           "#pragma once", '#include "mma_core.h"', ""]
    out.append(Gen("fn_kloop_fwd_h512", True).emit())
    out.append(Gen("fn_kloop_bwd_h512", False).emit())
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kloop_asm.h")
