#!/usr/bin/env python3
"""Generates kloop_asm.h: the K loops of the 128/64-row weight-stationary GRU scans (gru_persist.hip) at H = 512 as ONE hand-placed
asm statement per time step.

Why: with one wavefront per SIMD nothing hides an instruction that sits between two MFMAs unless it is ALONE there - hipcc gathers the
operand loads, LDS fragment reads and their address arithmetic at the chunk boundaries (6 ds_read_b128 + 14 scalar ops in a row), and
the in-kernel stamps put the forward K loop at 34.8-37 cycles per v_mfma_f32_16x16x4_f32 (32 is the pipe's rate) and the backward at 42.8.
Here every memory instruction gets its own MFMA shadow, addresses are immediates off a scalar base that advances in 4 KB steps, and
nothing is requested beyond the end of K.  The step's exchange-independent epilogue operands (token-table / projection rows in the
forward, saved gates + previous state + external gradient in the backward: HBM reads) are requested INSIDE the loop as well, behind the
first MFMAs: their latency disappears behind 25 k cycles of MFMAs instead of sitting between hand-over and first operand load (vmcnt
retires in order), and their address arithmetic no longer sits between a workgroup's arrival and its next poll.

Pipeline unit = 16 K values of the wave's 2 row tiles ("half chunk"): 2 operand loads (1 KB each) from the exchange slab, NB fragment
reads from the LDS-resident weight slice, 24 (forward: 2 row tiles x 3 gate tiles x 4 k-steps) or 8 (backward: 2 x 1 x 4) MFMAs.
RU - 1 units are in flight (a ring of RU x 2 x 4 registers); during unit u the registers of unit u-1 are re-filled with unit u+RU-1 and
the weight fragments of unit u+1 are read (double buffer).  Ring, fragments and accumulators live in AGPRs (MFMA, VMEM and DS instructions
of gfx950 take them directly), so the compiler's registers are untouched; the accumulation order of every accumulator equals the C++
loop's: bit-identical results.

s_waitcnt vmcnt(N) values come from a simulation of the issue order (N = memory instructions issued behind the awaited one); the
code is straight-line (one function per K length): the epilogue operands are spread over the whole loop.

Register maps (a = AGPR):
  forward : acc[m][n] a[4(3m+n)..] (a0-23), ring[slot][m] a[24+4(2slot+m)..] (a24-87, 8 slots), wfrag[bs][n] a[88+4(3bs+n)..] (a88-111)
  backward: acc[m][par] a[4(2m+par)..] (a0-15), ring[slot][m] a[16+4(2slot+m)..] (a16-143, 16 slots), wfrag[bs] a[144+4bs..] (a144-151)
Scalars: s[SB:SB+1] = operand base.
"""
import sys

SB = 84          # scalar temporaries s84..s85 (clobbered)


class Gen:
    def __init__(self, name, fwd, G):
        self.name, self.fwd, self.G = name, fwd, G
        self.RU = 8 if fwd else 16
        self.NB = 3 if fwd else 1
        self.ring0 = 24 if fwd else 16
        self.w0 = self.ring0 + self.RU * 8
        self.nagpr = self.w0 + 2 * self.NB * 4
        self.nmf = 24 if fwd else 8
        # epilogue operands requested inside the loop: instruction texts + the MFMA gaps (per unit of group 0) they go into
        if fwd:      # 2 items x 3 gate rows off one base each (base = middle gate: +-2048 bytes)
            self.extras = ["global_load_dwordx4 %%[ex%d%d], %%[xa%d], off offset:%d" % (i, q, i, (q - 1) * 2048) for i in range(2) for q in range(3)]
            step = 3 if G >= 4 else 2
        else:        # 2 items x (r, z, n, hn saved gates off one base; previous state; external gradient)
            self.extras = []
            for i in range(2):
                self.extras += ["global_load_dwordx4 %%[gt%d%d], %%[ga%d], off offset:%d" % (i, q, i, q * 1024) for q in range(4)]
                self.extras += ["global_load_dwordx4 %%[hp%d], %%[ha%d], off" % (i, i), "global_load_dwordx4 %%[xt%d], %%[xa%d], off" % (i, i)]
            step = 5 if G >= 6 else 3
        # ONE of them every `step` units from unit 1 on: all workgroups of the chip run in step, so the 24 / 48 KB per workgroup must not
        # become one HBM burst (12 MB: ~5 k cycles, during which no younger ring load can retire: measured +4 k cycles per backward step
        # when they were issued back to back), and each must be ring-depth units old before the first ring load behind it is awaited
        # all of them at the head of the FINAL group: no ring load is issued behind them, so nothing is blocked (vmcnt retires in order),
        # and 4-6 k cycles of MFMAs are left for them to land
        u0 = (self.G - 1) * self.RU
        if fwd:
            self.extra_units = {u0: [15, 18, 21], u0 + 1: [15, 18, 21]}
        else:
            self.extra_units = {u0 + j: [1, 6] for j in range(len(self.extras) // 2)}

    # ---- register maps
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

    # ---- stream state: s_rel = what s[SB:SB+1] points at relative to the first unit of the current group; vmops = issue order
    def advance_to(self, need_rel):
        out = []
        while need_rel - self.s_rel > 3072:
            out += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            self.s_rel += 4096
        assert 0 <= need_rel - self.s_rel <= 3072, (need_rel, self.s_rel)
        return out

    def gload(self, unit_abs, slot, m, need_rel):
        imm = need_rel - self.s_rel
        assert 0 <= imm <= 4095
        r = self.ring(slot, m)
        self.vmops.append(("ring", unit_abs))
        return "global_load_dwordx4 a[%d:%d], %%[vo%d], s[%d:%d] offset:%d sc1" % (r, r + 3, m, SB, SB + 1, imm)

    def wait_unit(self, unit_abs):
        last = max(i for i, o in enumerate(self.vmops) if o == ("ring", unit_abs))
        n = len(self.vmops) - 1 - last
        assert n < 64
        return "s_waitcnt vmcnt(%d)" % n

    def wread(self, bs, n, unit_in_group):
        r = self.wf(bs, n)
        if self.fwd:      # wl[3][nk][2][64][4] floats, gate stride 32 KB at H = 512: gates 0, 1 off %[lp], gate 2 off %[lq] = lp + 64 KB
            ptr, off = ("%[lp]", n * 32768 + unit_in_group * 1024) if n < 2 else ("%[lq]", unit_in_group * 1024)
        else:
            ptr, off = "%[lp]", unit_in_group * 1024
        assert off < 65536
        return "ds_read_b128 a[%d:%d], %s offset:%d" % (r, r + 3, ptr, off)

    def group(self, g, final):
        """text of group g (units g*RU .. g*RU+RU-1); final: nothing behind it"""
        RU = self.RU
        out = []
        start_rel = (RU - 1) * 1024 - 3072
        assert self.s_rel == start_rel
        wslots = [3, 9, 12] if self.fwd else [2]
        gslots = [0, 6] if self.fwd else [0, 4]
        t_tail = 15 if self.fwd else 6
        for k in range(RU):
            u = g * RU + k
            refill = (not final) or k == 0
            wnext = (not final) or k < RU - 1
            if refill:
                assert not self.advance_to((k + RU - 1) * 1024), "the scalar base must have been advanced in the previous unit"
            extra_here = self.extra_units.get(u, [])
            out.append(self.wait_unit(u))
            out.append("s_waitcnt lgkmcnt(0)")
            companions = []                   # per MFMA: the instructions issued right behind it
            for t in range(self.nmf):
                c = []
                if refill and t in gslots:
                    c.append(self.gload(u + RU - 1, (k - 1) % RU, gslots.index(t), (k + RU - 1) * 1024))
                if wnext and t in wslots:
                    c.append(self.wread((k + 1) & 1, wslots.index(t), k + 1))
                if t in extra_here and self.pending_extras:
                    c.append(self.pending_extras.pop(0))
                    self.vmops.append(("extra", 0))
                    if t == self.nmf - 2: c.append("s_nop 0")
                companions.append(c)
            # bookkeeping for the next unit / group: one instruction (or one s_add / s_addc pair) per free MFMA gap from t_tail on
            tail = []
            if (not final) and k + 1 < RU:
                tail += self.advance_to((k + 1 + RU - 1) * 1024)
            if k == RU - 1 and not final:
                tail += self.advance_to(RU * 1024 + start_rel + 3072)
                assert self.s_rel == RU * 1024 + start_rel
                tail.append("v_add_u32 %%[lp], 0x%x, %%[lp]" % (RU * 1024))
                if self.fwd:
                    tail.append("v_add_u32 %%[lq], 0x%x, %%[lq]" % (RU * 1024))
            for t, ins in enumerate(self.mfmas(k, k & 1)):
                out.append(ins)
                out += companions[t]
                if t >= t_tail and tail and (not companions[t] or t == self.nmf - 1):
                    out.append(tail.pop(0))
                    if tail and tail[0].startswith("s_addc"):
                        out.append(tail.pop(0))
            out += tail
        if not final:
            self.s_rel -= RU * 1024
        return out

    def stream(self, G):
        """(prologue, [group texts]) of a step with G groups"""
        RU = self.RU
        self.vmops, self.s_rel = [], 0
        self.pending_extras = list(self.extras)
        pro = ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1), "s_nop 4"]
        for n in range(self.NB):
            pro.append(self.wread(0, n, 0))
        for u in range(RU - 1):       # units 0 .. RU-2 (unit RU-1 is the "re-fill" of unit 0)
            adv = self.advance_to(u * 1024)
            pro += adv + (["s_nop 4"] if adv else [])        # SALU write of the base -> VMEM read of it
            for m in range(2):
                pro.append(self.gload(u, u, m, u * 1024))
        start_rel = (RU - 1) * 1024 - 3072
        while self.s_rel < start_rel:
            pro += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            self.s_rel += 4096
        assert self.s_rel == start_rel
        for c in range(24 if self.fwd else 16):
            pro.append("v_accvgpr_write_b32 a%d, 0" % c)
        groups = [self.group(g, g == G - 1) for g in range(G)]
        assert not self.pending_extras
        return pro, groups

    def build(self):
        pro, groups = self.stream(self.G)     # straight-line: the epilogue operands are spread over the whole loop
        L = list(pro)
        for g in groups:
            L += g
        # accumulators -> LDS (padded MFMA C layout, see gru_persist.hip (d)); an MFMA result needs 12 wait states before anything but an
        # accumulating MFMA reads it.  The final vmcnt(0) of the ring also covers the epilogue operands (issued long before).
        L += ["s_nop 15", "s_nop 7"]
        if self.fwd:
            for m in range(2):
                for n in range(3):
                    c = self.acc(m, n)
                    L.append("ds_write_b128 %%[red], a[%d:%d] offset:%d" % (c, c + 3, (m * 3 + n) * 1088))
        else:
            # the two k-parity accumulators of a row tile go to two planes of 8 tiles (WK x EM = 8 in both tilings); the epilogue adds them
            for m in range(2):
                for par in range(2):
                    c = self.acc(m, par)
                    L.append("ds_write_b128 %%[red], a[%d:%d] offset:%d" % (c, c + 3, m * 1088 + par * 8 * 1088))
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return L

    def emit(self):
        L = self.build()
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join('"a%d"' % i for i in range(self.nagpr))
        if self.fwd:
            sig = ("const float* xin_, unsigned vo0, unsigned vo1, unsigned lp, unsigned lq, unsigned red,\n"
                   "                                 const float* xa0, const float* xa1, f32x4 (&ex)[2][3]")
            outs = ", ".join('[ex%d%d] "=&v"(ex[%d][%d])' % (i, q, i, q) for i in range(2) for q in range(3))
            ins = '[xa0] "v"(xa0), [xa1] "v"(xa1)'
            io = '[lp] "+v"(lp), [lq] "+v"(lq)'
            doc = ("xa0 / xa1 = per-lane address of the MIDDLE gate's 4 input-projection values of epilogue item 0 / 1 (gates 2 KB apart);\n"
                   "// ex[i][q] receives them")
        else:
            sig = ("const float* xin_, unsigned vo0, unsigned vo1, unsigned lp, unsigned red,\n"
                   "                                 const float* ga0, const float* ga1, const float* ha0, const float* ha1, const float* xa0, const float* xa1,\n"
                   "                                 f32x4 (&gt)[2][4], f32x4 (&hp)[2], f32x4 (&xt)[2]")
            outs = ", ".join(['[gt%d%d] "=&v"(gt[%d][%d])' % (i, q, i, q) for i in range(2) for q in range(4)] +
                             ['[hp%d] "=&v"(hp[%d])' % (i, i) for i in range(2)] + ['[xt%d] "=&v"(xt[%d])' % (i, i) for i in range(2)])
            ins = ", ".join('[%s%d] "v"(%s%d)' % (n, i, n, i) for n in ("ga", "ha", "xa") for i in range(2))
            io = '[lp] "+v"(lp)'
            doc = ("ga / ha / xa = per-lane addresses of epilogue item 0 / 1: saved gates (r; z, n, hn follow 1 KB apart), previous state, external\n"
                   "// gradient; gt / hp / xt receive them")
        return """
// %s K loop of one time step: xin = this wave's first operand unit (uniform), vo0 / vo1 = byte offsets of its two row tiles (+ lane * 16),
// lp%s = LDS byte address of its first weight-fragment unit (+ lane * 16), %d K units (groups of %d), red = LDS byte address of its
// accumulator tiles; %s.
// %d MFMAs per unit; %d AGPRs.  Everything requested here has landed when the statement ends.
FN_DEVINL void %s(%s) {
    // wave-uniform by construction (the K range depends on the wave id only); the compiler cannot see that
    const unsigned long long xq = (unsigned long long)(uintptr_t)xin_;
    const float* xin = reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xq >> 32)) << 32) |
                                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(xq & 0xffffffffull))));
    asm volatile(
%s
        : %s, %s
        : [xin] "s"(xin), [vo0] "v"(vo0), [vo1] "v"(vo1), [red] "v"(red), %s
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % ("forward" if self.fwd else "backward", " / lq" if self.fwd else "", self.G * self.RU, self.RU, doc, self.nmf, self.nagpr, self.name, sig,
       body, io, outs, ins, SB, SB + 1, clob)


def main(path):
    out = ["// GENERATED by gen_kloop.py - do not edit.  Hand-placed K loops of the weight-stationary GRU scans (H = 512, 2 row tiles per wave).",
           "#pragma once", '#include "mma_core.h"', ""]
    out.append(Gen("fn_kloop_fwd_h512_k512", True, 4).emit())        # one wave over all of K (128-row tiling)
    out.append(Gen("fn_kloop_fwd_h512_k256", True, 2).emit())        # half of K per wave (64-row tiling)
    out.append(Gen("fn_kloop_bwd_h512_k1536", False, 6).emit())
    out.append(Gen("fn_kloop_bwd_h512_k768", False, 3).emit())
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kloop_asm.h")
