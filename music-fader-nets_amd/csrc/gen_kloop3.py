#!/usr/bin/env python3
"""Generates kloop3_asm.h: the K loops of the PING-PONG weight-stationary forward scans on the bf16 MFMA with exact bf16 triple splits
("bf16 x 6", gru_persist.hip: gru_fwd_x6pp_kernel) at H = 512.

Round 5.  Same protocol as kloop2_asm.h (gen_kloop2.py has the why): a workgroup owns two halves of its row group and alternates between
them, phase f = (half X, step p): K loop of X -> accumulators to LDS -> gate epilogue of X (arithmetic only); the STORES of an epilogue and
the arrival for them are issued inside the NEXT phase's K loop, the counter of the other half is read there too.  What differs:

  * pipeline unit = one K block of 32 values of ONE row tile: 3 operand loads (the bf16 pieces hi / mid / lo of the recurrent state,
    1 KB each, 3 KB contiguous on the exchange slab), 9 weight-fragment reads (3 gates x 3 pieces), 18 v_mfma_f32_16x16x32_bf16
    (6 of the 9 partial products per gate, smallest first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; the three gates take turns so that
    an accumulator is touched every third MFMA).  Per accumulator: K blocks in order, the six products in this order - the
    accumulation order of gru_fwd_x6_kernel, bit-identical sums where the K range is not split;
  * a 16-cycle MFMA phase is 2.7 x shorter than the fp32 one while store / counter / L2 latencies are what they are, so the counter of the
    other half is LOADED near the end of the loop and LOOKED AT behind its last MFMA; the ring of the next phase is then requested by a
    separate statement (`*_pro`, 3 (RU - 1) loads back to back) in front of the epilogue - it lands while the epilogue computes.  One code
    path: no request duplicated into the loop's tail;
  * the scalar operand base advances by one unit (3 KB) per unit; every refill is base + {0, 1, 2} KB;
  * the epilogue operands (HBM reads) are requested behind the LAST refill of the loop: vmcnt retires in order, a slow load in front of a ring
    load would stall the unit that waits for the latter;
  * the s_barrier of the arrival block is executed in EVERY phase (arrival or not): the accumulator tiles in LDS hold one half only
    (144 KB of weight triples leave 13 KB), and the barrier is what keeps a wave's accumulator writes behind the other waves' epilogue reads.

Register map (a = AGPR): acc[q] a[4 q ..], ring[slot][piece] a[12 + 12 slot + 4 piece ..], wfrag[bs][q][piece] a[w0 + 36 bs + 12 q + 4 piece ..].
Scalars s84..s87.
"""
import sys

SB = 84
UB = 3072            # bytes of one unit on the exchange slab / in the LDS weight image (3 pieces x 1 KB)
GS = 49152           # gate stride of the LDS weight image: 16 K blocks x 3 KB
PROD = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]      # (piece of A = state, piece of B = weights), smallest product first; 0 = hi, 1 = mid, 2 = lo


class GenX6:
    def __init__(self, name, units, RU, stores, masked, u_arr, poll_unit, extra_unit):
        self.name, self.units, self.RU, self.stores, self.masked = name, units, RU, stores, masked
        self.u_arr, self.poll_unit, self.extra_unit = u_arr, poll_unit, extra_unit
        assert RU <= units and RU % 2 == 0
        self.ring0 = 12
        self.w0 = self.ring0 + 12 * RU
        self.nagpr = self.w0 + 72
        assert self.nagpr <= 256

    def ring(self, slot, pc):
        return self.ring0 + 12 * (slot % self.RU) + 4 * pc

    def wf(self, bs, q, pc):
        return self.w0 + 36 * bs + 12 * q + 4 * pc

    def wread(self, bs, q, pc, unit):
        r = self.wf(bs, q, pc)
        off = unit * UB + pc * 1024
        assert 0 <= off < 65536
        return "ds_read_b128 a[%d:%d], %%[lp%d] offset:%d" % (r, r + 3, q, off)

    def mfmas(self, slot, bs):
        out = []
        for pa, pb in PROD:
            for q in range(3):
                a, b, c = self.ring(slot, pa), self.wf(bs, q, pb), 4 * q
                out.append("v_mfma_f32_16x16x32_bf16 a[%d:%d], a[%d:%d], a[%d:%d], a[%d:%d]" % (c, c + 3, a, a + 3, b, b + 3, c, c + 3))
        return out

    def wait_unit(self, u):
        last = max(i for i, o in enumerate(self.vmops) if o == ("ring", u))
        n = len(self.vmops) - 1 - last
        assert n < 60, n
        return "s_waitcnt vmcnt(%d)" % n

    def masked_ins(self, ins):
        if not self.masked:
            return ins
        return ["s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 0xffffffff"] + ins + ["s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3)]

    def arrive_block(self):
        """the slab stores of the previous epilogue have completed in every wave (barrier: also the fence between the previous epilogue's
        reads of the accumulator tiles and this phase's writes) -> one arrival (arr: 0 = none due, 1 = due, 2 = due and this wave issues it)"""
        slab = [i for i, o in enumerate(self.vmops) if o == ("store", "slab")]
        # no stores in this statement: the previous epilogue's were issued in front of it (flush_stores) - everything older than the statement
        n = (len(self.vmops) - 1 - max(slab)) if slab else (len(self.vmops) - self.n0)
        # (the atomic is NOT entered into the bookkeeping: only one wave issues it, and an operation the count does not know makes a later
        # wait more patient, one it wrongly knows would end a wait too early)
        return ["s_waitcnt vmcnt(%d)" % n, "s_barrier", "s_cmp_lt_u32 %[arr], 2", "s_cbranch_scc1 .Lnoarr_%=",
                "s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 1", "v_mov_b32 %[pv], 1",
                "global_atomic_add %[acnt], %[pv], off", "s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3), ".Lnoarr_%=:"]

    def body(self):
        RU, units = self.RU, self.units
        self.vmops = [("ring", u) for u in range(RU - 1) for _ in range(3)]      # in flight when the statement starts: the ring request and nothing else
        self.n0 = len(self.vmops)
        L = ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1)]
        # running base = first refill of unit 0's turn = unit RU - 1
        adv0 = (RU - 1) * UB
        L += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, adv0), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
        for c in range(12):
            L.append("v_accvgpr_write_b32 a%d, 0" % c)
        if self.stores:
            # previous epilogue: exchange slab (write-through, the three pieces 1 KB apart), h_all, the four saved-gate vectors (1 KB apart)
            slab = [("slab", "global_store_dwordx2 %%[sa0], %%[t%d], off offset:%d sc1" % (pc, pc * 1024)) for pc in range(3)]
            outs = [("out", "global_store_dwordx4 %[sa1], %[d0], off")]
            outs += [("out", "global_store_dwordx4 %%[sa2], %%[d%d], off offset:%d" % (q + 1, q * 1024)) for q in range(4)]
        else:
            slab, outs = [], []
        extras = ["global_load_dwordx4 %%[ex%d], %%[xa], off offset:%d" % (q, (q - 1) * 2048) for q in range(3)] + ["global_load_dword %[tokn], %[ta], off"]
        done_arr = False
        for u in range(units):
            bs = u & 1
            comp = [[] for _ in range(18)]
            vm = [[] for _ in range(18)]                   # bookkeeping entries in issue order
            refill = u + RU - 1 < units
            if refill:
                for pc in range(3):
                    r = self.ring(u + RU - 1, pc)
                    comp[pc].append("global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d sc1" % (r, r + 3, SB, SB + 1, pc * 1024))
                    vm[pc].append(("ring", u + RU - 1))
                comp[3] += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            if u + 1 < units:
                n = 0
                for q in range(3):
                    for pc in range(3):
                        comp[2 + n].append(self.wread(bs ^ 1, q, pc, u + 1))
                        n += 1
            if u == 0 and slab:
                for i, (kind, ins) in enumerate(slab):
                    comp[11 + 2 * i] += self.masked_ins([ins])
                    vm[11 + 2 * i].append(("store", kind))
            elif outs and u >= 1 and not (u == self.extra_unit):
                kind, ins = outs.pop(0)
                comp[12] += self.masked_ins([ins])
                vm[12].append(("store", kind))
            if u == self.extra_unit:
                for i, ins in enumerate(extras):
                    comp[11 + i].append(ins)
                    vm[11 + i].append(("extra", 0))
            if u == self.poll_unit:
                comp[16].append("global_load_dword %[pv], %[pcnt], off sc1")
                vm[16].append(("poll", 0))
            L.append(self.wait_unit(u))
            L.append("s_waitcnt lgkmcnt(0)")
            for t, ins in enumerate(self.mfmas(u, bs)):
                L.append(ins)
                L += comp[t]
                self.vmops += vm[t]
            if u == self.u_arr:
                L += self.arrive_block()
                done_arr = True
        assert done_arr and not outs, (done_arr, outs)
        # everything requested by this statement has landed (epilogue operands, the counter value; the stores have completed)
        L += ["s_waitcnt vmcnt(0)", "s_nop 7"]
        L += ["ds_write_b128 %%[red], a[%d:%d] offset:%d" % (4 * q, 4 * q + 3, q * 1088) for q in range(3)]
        L.append("s_waitcnt lgkmcnt(0)")
        return L

    def emit_main(self):
        L = self.body()
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join('"a%d"' % i for i in range(self.nagpr))
        sig = ("const void* xin_, unsigned vo, unsigned lp0, unsigned lp1, unsigned lp2, unsigned red,\n"
               "        int arr, u32* acnt, const u32* pcnt, const float* xa, const int* ta")
        ins = '[xa] "v"(xa), [ta] "v"(ta)'
        if self.stores:
            sig += (",\n        void* sa0, float* sa1, float* sa2, const u32x2& t0, const u32x2& t1, const u32x2& t2,\n"
                    "        const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3, const f32x4& d4")
            ins += (', [sa0] "v"(sa0), [sa1] "v"(sa1), [sa2] "v"(sa2), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), '
                    '[d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [d4] "v"(d4)')
        sig += ",\n        f32x4 (&ex)[3], int& tokn, unsigned& pv"
        outs = ", ".join(['[ex%d] "=&v"(ex[%d])' % (q, q) for q in range(3)] + ['[tokn] "=&v"(tokn)', '[pv] "=&v"(pv)'])
        what = ("\n// sa0 + t0..t2: this lane's 8 bytes per piece on the exchange slab and the new state as bf16 triples; sa1 / sa2 + d0..d4: h_all and "
                "saved-gates addresses, new state, r, z, n, W_hn h + b_hn") if self.stores else ""
        return """
// %s: K loop of one phase on the bf16 MFMA (%d units of 32 K values x 3 pieces, ring of %d, 18 MFMAs per unit, %d AGPRs; %s%s).
// xin = this wave's first operand unit of THIS phase (uniform), vo = byte offset of its row tile (+ lane * 16); the first %d units are already in
// flight (`*_pro`).  lp0..2 = LDS byte addresses of this wave's first weight unit of gates r, z, n.  arr / acnt: arrival for the previous phase's
// epilogue (0 none, 1 due, 2 due and this wave issues it / its counter).  pcnt: counter of the next phase's half, returned in pv (loaded %d unit(s)
// before the end).  xa / ta: this phase's epilogue operands (middle gate row of the input pre-activations; token of the next step).%s
FN_DEVINL void %s(%s) {
    const void* xin = fn_uniform_ptr(reinterpret_cast<const float*>(xin_));
    arr = __builtin_amdgcn_readfirstlane(arr);       // wave-uniform by construction; an "s" operand the compiler believes divergent would be handed over in a VGPR
    asm volatile(
%s
        : %s
        : [xin] "s"(xin), [vo] "v"(vo), [red] "v"(red), [lp0] "v"(lp0), [lp1] "v"(lp1), [lp2] "v"(lp2), [arr] "s"(arr),
          [acnt] "v"(acnt), [pcnt] "v"(pcnt), %s
        : "memory", "scc", "vcc", "s%d", "s%d", "s%d", "s%d", %s);
}
""" % (self.name, self.units, self.RU, self.nagpr, "issues the previous epilogue's stores" if self.stores else "no stores to issue",
       ", lanes 0-31 store" if self.masked and self.stores else "", self.RU - 1, self.units - self.poll_unit, what, self.name, sig, body, outs, ins,
       SB, SB + 1, SB + 2, SB + 3, clob)

    def emit_pro(self, name):
        L = ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1), "s_nop 4"]
        for u in range(self.RU - 1):
            for pc in range(3):
                r = self.ring(u, pc)
                L.append("global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d sc1" % (r, r + 3, SB, SB + 1, pc * 1024))
            if u + 1 < self.RU - 1:
                L += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
        for q in range(3):
            for pc in range(3):
                L.append(self.wread(0, q, pc, 0))
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        regs = list(range(self.ring0, self.ring0 + (self.RU - 1) * 12)) + list(range(self.w0, self.w0 + 36))
        clob = ", ".join('"a%d"' % i for i in regs)
        return """
// ring request of a phase (units 0 .. %d, three pieces each) + the weight fragments of unit 0
FN_DEVINL void %s(const void* xin_, unsigned vo, unsigned lp0, unsigned lp1, unsigned lp2) {
    const void* xin = fn_uniform_ptr(reinterpret_cast<const float*>(xin_));
    asm volatile(
%s
        :
        : [xin] "s"(xin), [vo] "v"(vo), [lp0] "v"(lp0), [lp1] "v"(lp1), [lp2] "v"(lp2)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (self.RU - 2, name, body, SB, SB + 1, clob)


HEAD = """// GENERATED by gen_kloop3.py - do not edit.  K loops of the ping-pong forward scans on the bf16 MFMA with exact bf16 triple splits (H = 512).
#pragma once
#include "kloop2_asm.h"
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
"""


# (units, tag, masked, RU, unit of the arrival, unit whose last MFMAs load the other half's counter, unit that requests the epilogue operands)
# ring depth: measured in one session (scratch/r5_call4.sh: encoder launch / 32-step decoder launch) RU = 8: 2.39-2.43 ms / 260-264 us, RU = 6: 2.34-2.38 ms /
# 244-250 us, RU = 4: 2.42-2.48 ms / 284-291 us - the 3 (RU - 1) loads of `*_pro` reach the CU's address unit (64 B per clock, four waves at once) as
# one burst in front of the epilogue, a shallow ring exposes L2 latency inside the loop
CONFIG = {"k512": (16, False, 6, 4, 13, 11), "k256": (8, True, 6, 4, 6, 3)}


def main(path, overrides=()):
    out = [HEAD]
    cfg = dict(CONFIG)
    for o in overrides:                               # tuning builds: k256=RU,u_arr,poll,extra
        tag, vals = o.split("=")
        cfg[tag] = cfg[tag][:2] + tuple(int(v) for v in vals.split(","))
    # k512: one wave over all of K (128-row groups, every lane has an epilogue item); k256: K split over two wave pairs (64-row groups, lanes
    # 0-31 of every wave have one)
    for tag in ("k512", "k256"):
        units, masked, RU, u_arr, poll, extra = cfg[tag]
        for stores in (1, 0):
            out.append(GenX6("fn_x6_fwd_%s_%s" % (tag, "main" if stores else "first"), units, RU, stores, masked, u_arr, poll, extra).emit_main())
        out.append(GenX6("x", units, RU, 0, masked, u_arr, poll, extra).emit_pro("fn_x6_fwd_%s_pro" % tag))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kloop3_asm.h", sys.argv[2:])
