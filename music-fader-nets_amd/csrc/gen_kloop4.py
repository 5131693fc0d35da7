#!/usr/bin/env python3
"""Generates kloop4_asm.h: the K loops of the PING-PONG, register-stationary BACKWARD scan on the bf16 MFMA with exact bf16 triple splits
("bf16 x 6", gru_persist.hip: gru_bwd_x6_kernel) at H = 512.

Round 5.  Decomposition of gru_bwd_rs_kernel (gen_kloop2.py, GenRS): a workgroup owns 32 dh columns (16 slices) of a 32 TH-row group, two halves of
TH row tiles; every wave multiplies the current half over ITS K quarter (12 blocks of 32 gate-gradient columns) against both column tiles; the four
partial sums meet in LDS.  What the triple arithmetic changes:

  * the exchanged operand - the gate gradients [dr' | dz' | dn' r] of a row, 3H columns - travels as bf16 TRIPLES (6 instead of 4 bytes per value:
    [row tile][48 K blocks][piece][64 lanes][8 bf16]); pipeline unit = one K block of ONE row tile: 3 operand loads (1 KB each), 12
    v_mfma_f32_16x16x32_bf16 (6 partial products x 2 column tiles, smallest first, the column tiles take turns).  One 1 KB load feeds 4 MFMAs of
    16 cycles: the CU's address path (64 B per clock, four waves) is as busy as the MFMA pipes - this loop is bound by the operand stream;
  * W_hh^T slice as triples: column tile 0 of the wave's K quarter in AGPRs (12 blocks x 3 pieces x 4 = 144 registers, loaded once), column tile 1
    in LDS (blocks 0..10 of every quarter: 132 KB) with block 11 in AGPRs too (an LDS image of all 12 would leave no room for the accumulator tiles);
  * register budget (256 AGPRs): W 144 + 12, ring 6 x 12, the epilogue operands of the NEXT phase 24.  Accumulators and the LDS weight fragments
    live in VGPRs that only exist inside a statement (v208..v255: clobbered, nothing crosses a statement in them);
  * the epilogue operands (saved gates, previous state, external gradient: 6 HBM reads per item) are requested ONE PHASE AHEAD into AGPRs
    behind the last refill of the loop and are NOT waited for at its end (vmcnt retires in order: they stand in front of nothing but the next
    phase's ring, which is needed a whole epilogue later); the NEXT K loop statement hands them to the compiler: behind its arrival barrier the
    wave's own accumulator tiles in LDS are free until the end of the loop, so the 128-bit values bounce through them (ds_write_b128 from the
    AGPRs, ds_read_b128 into the output operands - inline asm cannot name a component of a 128-bit operand, v_accvgpr_read would need 24
    scalar outputs and the statement is limited to 30 operands);
  * the exchange-slab stores of an epilogue (9 x 8 bytes per item, write-through: ~30 cycles of the CU's address path each - 1.2 k cycles per phase
    when the four waves issue them back to back, measured) ride in the NEXT K loop, three per unit in front of the arrival; the four 16-byte
    stores nobody waits for (dgx, dghn) are a small statement of their own right behind the epilogue (`*_out`: counted, so that they enter the
    vmcnt arithmetic) - with both store sets the K loop statement would need 33 operands (inline asm takes 30);
  * counter of the other half: loaded near the end of the loop, looked at behind it; the ring request is a statement of its own (as kloop3_asm.h).

State that crosses statements: W (a0..a155), the ring (RU - 1 units in flight), the next phase's epilogue operands (in flight), the slab stores,
and the vmcnt arithmetic: between two statements the compiler issues NO vector-memory instruction (scratch/check_pp_isa.py).

AGPR map: W0[blk][piece] a[12 blk + 4 piece ..] (blk 0..11), W1[11][piece] a[144 + 4 piece ..], ring[slot][piece] a[156 + 12 slot + 4 piece ..],
ext[j] a[228 + 4 j ..] (j = 0..3 saved gates r, z, n, hn; 4 previous state; 5 external gradient).
VGPR temporaries: acc[m][ct] v[208 + 4 (2 m + ct) ..], wfrag[bs][piece] v[224 + 12 bs + 4 piece ..], v255 = offset of the second row tile.  Scalars s84..s87.
"""
import os
import sys

# Experiment switches (KLOOP4_* environment variables, used by scratch/ variant scripts only) change wait counts / hints of the generated statements: a
# production build must not pick up a stray one.  They are honoured only when KLOOP_EXPERIMENT=1 is set with them; otherwise the generator refuses.
_stray = sorted(k for k in os.environ if k.startswith("KLOOP4_"))
if _stray and os.environ.get("KLOOP_EXPERIMENT") != "1":
    sys.exit("gen_kloop4.py: experiment switches %s are set without KLOOP_EXPERIMENT=1 - refusing to generate a production header" % ", ".join(_stray))

NT = " nt" if os.environ.get("KLOOP4_NT") == "1" else ""      # experiment: streaming hint for read-once / write-once traffic
SB = 84
UB = 3072
TILE = 48 * UB       # bytes of one row tile on the exchange slab
PROD = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]      # (piece of A = gate gradients, piece of B = weights), smallest product first
W0, W1X, RING0, EXT0 = 0, 144, 156, 228
VACC, VWF, VO2 = 208, 224, 255
NBLK = 12
NEXT = 6             # epilogue-operand loads per phase
NSLAB = 9            # exchange-slab stores per item (3 gates x 3 pieces)
NOUT = 4             # dgx r / z / n and dghn stores per item


class GenX6B:
    def __init__(self, name, TH, RU, stores, u_arr, poll_unit):
        self.name, self.TH, self.RU, self.stores = name, TH, RU, stores
        self.masked = TH == 1
        self.units = NBLK * TH
        self.u_arr, self.poll_unit = u_arr, poll_unit
        assert RING0 + 12 * RU <= EXT0 and poll_unit + 1 < self.units and poll_unit > self.units - RU
        self.nacc = 2 * TH

    def ring(self, u, pc):
        return RING0 + 12 * (u % self.RU) + 4 * pc

    def blk_m(self, u):
        return u // self.TH, u % self.TH

    def vo(self, m):
        return "%[vo]" if m == 0 else "v%d" % VO2

    def wfrag(self, blk, pc):
        """B operand of column tile 1 for K block blk: LDS-fed VGPR double buffer, or the resident block 11"""
        if blk == NBLK - 1:
            return "a[%d:%d]" % (W1X + 4 * pc, W1X + 4 * pc + 3)
        r = VWF + 12 * (blk & 1) + 4 * pc
        return "v[%d:%d]" % (r, r + 3)

    def wread(self, blk):
        assert blk < NBLK - 1
        return ["ds_read_b128 v[%d:%d], %%[lp] offset:%d" % (VWF + 12 * (blk & 1) + 4 * pc, VWF + 12 * (blk & 1) + 4 * pc + 3, blk * UB + pc * 1024) for pc in range(3)]

    def mfmas(self, u):
        blk, m = self.blk_m(u)
        out = []
        for pa, pb in PROD:
            a = self.ring(u, pa)
            for ct in range(2):
                c = VACC + 4 * (2 * m + ct)
                b = "a[%d:%d]" % (W0 + 12 * blk + 4 * pb, W0 + 12 * blk + 4 * pb + 3) if ct == 0 else self.wfrag(blk, pb)
                out.append("v_mfma_f32_16x16x32_bf16 v[%d:%d], a[%d:%d], %s, v[%d:%d]" % (c, c + 3, a, a + 3, b, c, c + 3))
        return out

    def load_unit(self, u):
        """the three operand loads of unit u off the running base (which must point at unit u's K block)"""
        blk, m = self.blk_m(u)
        return ["global_load_dwordx4 a[%d:%d], %s, s[%d:%d] offset:%d sc1" % (self.ring(u, pc), self.ring(u, pc) + 3, self.vo(m), SB, SB + 1, pc * 1024)
                for pc in range(3)]

    def wait_for(self, key):
        last = max(i for i, o in enumerate(self.vmops) if o == key)
        n = len(self.vmops) - 1 - last
        assert n < 62, n
        return "s_waitcnt vmcnt(%d)" % n

    def masked_ins(self, ins):
        if not self.masked:
            return ins
        return ["s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 0xffffffff"] + ins + ["s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3)]

    def arrive_block(self):
        """every wave's exchange-slab stores (issued by this statement's first units) have completed -> barrier (also the fence between the previous epilogue's reads of the accumulator tiles and this phase's writes) -> one
        arrival (arr: 0 = none due, 1 = due, 2 = due and this wave issues it).  `_first`: the stores it would wait for were drained by the caller."""
        slab = [i for i, o in enumerate(self.vmops) if o == ("store", "slab")]
        L = ["s_waitcnt vmcnt(%d)" % (len(self.vmops) - 1 - max(slab))] if slab else []
        return L + ["s_barrier", "s_cmp_lt_u32 %[arr], 2", "s_cbranch_scc1 .Lnoarr_%=",
                    "s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 1", "v_mov_b32 %[pv], 1",
                    "global_atomic_add %[pcnt], %[pv], off", "s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3), ".Lnoarr_%=:"]

    def body(self):
        RU, units, TH = self.RU, self.units, self.TH
        # in flight when the statement starts, oldest first: this phase's epilogue operands (requested by the previous statement, or `_ext`), the ring
        # request (`_pro`), the previous epilogue's dgx / dghn stores (`_out`)
        self.vmops = [("ext", 0)] * NEXT + [("ring", u) for u in range(RU - 1) for _ in range(3)] + ([("store", "out")] * NOUT if self.stores else [])
        self.n0 = len(self.vmops)
        L = []
        if TH == 2:
            L.append("v_add_u32 v%d, 0x%x, %%[vo]" % (VO2, TILE))
        base_blk = (RU - 1) // TH
        L += ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1), "s_add_u32 s%d, s%d, 0x%x" % (SB, SB, base_blk * UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
        L += self.wread(0)
        for c in range(4 * self.nacc):
            L.append("v_mov_b32 v%d, 0" % (VACC + c))
        slabs = []
        if self.stores:      # previous epilogue's gate gradients as triples: gate g at byte offset 0xC000 g (16 K blocks), piece p 1 KB further
            for g in range(3):
                for pc in range(3):
                    pre = ["v_add_u32 v%d, 0x%x, %%[so0]" % (VO2 - 1, 0xC000 * g)] if (g and pc == 0) else []
                    slabs.append(pre + ["global_store_dwordx2 %s, %%[t%d%d], %%[sbase] offset:%d sc1" % ("%[so0]" if g == 0 else "v%d" % (VO2 - 1), g, pc, pc * 1024)])
        ext_out = ["gt0", "gt1", "gt2", "gt3", "hp", "xt"]
        ext_in = ["global_load_dwordx4 a[%d:%d], %%[ga], off offset:%d" % (EXT0 + 4 * q, EXT0 + 4 * q + 3, q * 1024) + NT for q in range(4)]
        ext_in += ["global_load_dwordx4 a[%d:%d], %%[ha], off" % (EXT0 + 16, EXT0 + 19) + NT, "global_load_dwordx4 a[%d:%d], %%[xa], off" % (EXT0 + 20, EXT0 + 23) + NT]
        done_arr = False
        for u in range(units):
            blk, m = self.blk_m(u)
            comp = [[] for _ in range(12)]
            vm = [[] for _ in range(12)]
            v = u + RU - 1
            if v < units:
                nb = v // TH
                if nb != base_blk:
                    L += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, (nb - base_blk) * UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
                    base_blk = nb
                for pc, ins in enumerate(self.load_unit(v)):
                    comp[pc].append(ins)
                    vm[pc].append(("ring", v))
            if m == TH - 1 and blk + 1 < NBLK - 1:       # last unit of a K block: the LDS half of the next block's weights
                for i, ins in enumerate(self.wread(blk + 1)):
                    comp[4 + i].append(ins)
            # hand-over of this phase's epilogue operands (they landed long ago: older than unit 0's ring loads): AGPRs -> own accumulator tiles in LDS ->
            # output operands, nacc vectors per unit, in the units behind the arrival barrier (a wave's LDS operations execute in order, every lane
            # reads back its own 16 bytes: no wait in between)
            rounds = [list(range(i, min(i + self.nacc, NEXT))) for i in range(0, NEXT, self.nacc)]
            if self.u_arr < u <= self.u_arr + len(rounds):
                js = rounds[u - self.u_arr - 1]
                for i, j in enumerate(js):
                    comp[i].append("ds_write_b128 %%[red], a[%d:%d] offset:%d" % (EXT0 + 4 * j, EXT0 + 4 * j + 3, i * 1088))
                for i, j in enumerate(js):
                    comp[6 + i].append("ds_read_b128 %%[%s], %%[red] offset:%d" % (ext_out[j], i * 1088))
            if slabs and u < 3:                          # three slab stores per unit in units 0..2 (gate u)
                for i in range(3):
                    comp[5 + 2 * i] += self.masked_ins(slabs[3 * u + i])
                    vm[5 + 2 * i].append(("store", "slab"))
            if u == self.poll_unit:
                comp[8].append("global_load_dword %[pv], %[pcnt], off sc1")
                vm[8].append(("poll", 0))
            if u == self.poll_unit + 1:                  # the NEXT phase's epilogue operands: behind the counter load, they stay in flight past the end
                for i, ins in enumerate(ext_in):
                    comp[3 + i].append(ins)
                    vm[3 + i].append(("next_ext", 0))
            L.append(self.wait_for(("ring", u)))
            if m == 0 and blk < NBLK - 1:
                L.append("s_waitcnt lgkmcnt(0)")
            for t, ins in enumerate(self.mfmas(u)):
                L.append(ins)
                L += comp[t]
                self.vmops += vm[t]
            if u == self.u_arr:
                L += self.arrive_block()
                done_arr = True
        assert done_arr
        # the counter value and everything older have landed; the next phase's epilogue operands may still be in flight
        assert self.vmops[-NEXT:] == [("next_ext", 0)] * NEXT
        L += ["s_waitcnt vmcnt(%d)" % NEXT, "s_waitcnt lgkmcnt(0)", "s_nop 7"]
        L += ["ds_write_b128 %%[red], v[%d:%d] offset:%d" % (VACC + 4 * j, VACC + 4 * j + 3, j * 1088) for j in range(self.nacc)]
        L.append("s_waitcnt lgkmcnt(0)")
        return L

    def clobbers(self, ring_only=False):
        regs = ['"a%d"' % i for i in range(RING0, EXT0 + 24)]
        regs += ['"v%d"' % i for i in range(VACC, VACC + 4 * self.nacc)] + ['"v%d"' % i for i in range(VWF, VWF + 24)] + ['"v%d"' % VO2, '"v%d"' % (VO2 - 1)]
        return ", ".join(regs)

    def emit_main(self):
        L = self.body()
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        sig = ("const void* xin_, unsigned vo, unsigned lp, unsigned red, int arr, u32* pcnt,\n"
               "        const float* ga, const float* ha, const float* xa")
        ins = '[ga] "v"(ga), [ha] "v"(ha), [xa] "v"(xa)'
        decl = post = ""
        if self.stores:
            sig += ",\n        void* sbase_, unsigned so0, " + ", ".join("const u32x2& t%d%d" % (g, pc) for g in range(3) for pc in range(3))
            ins += ', [sbase] "s"(sbase), [so0] "v"(so0), ' + ", ".join('[t%d%d] "v"(t%d%d)' % (g, pc, g, pc) for g in range(3) for pc in range(3))
            decl = "    void* sbase = const_cast<float*>(fn_uniform_ptr(reinterpret_cast<const float*>(sbase_)));\n"
        sig += ",\n        f32x4 (&gt)[4], f32x4& hp, f32x4& xt, unsigned& pv"
        outs = ", ".join(['[gt%d] "=&v"(gt[%d])' % (q, q) for q in range(4)] + ['[hp] "=&v"(hp)', '[xt] "=&v"(xt)', '[pv] "=&v"(pv)'])
        return """
// %s: K loop of one backward phase on the bf16 MFMA (%d row tile(s) x 2 column tiles, %d units = this wave's K quarter, ring of %d, 12 MFMAs per unit; %s).
// xin = this wave's first K block of THIS phase's operand slab (uniform), vo = byte offset of the half's first row tile (+ lane * 16); lp = LDS byte
// address of this wave's first weight block of column tile 1.  In flight at the start, oldest first: this phase's epilogue operands, the ring (`_pro`)%s.
// arr / pcnt: arrival for the previous phase's epilogue (0 none, 1 due, 2 due and this wave issues it) - its half is the half of the NEXT phase, whose
// counter pcnt also is: the value the loop loads near its end is returned in pv.  ga / ha / xa: epilogue operands of the NEXT phase (saved gates, previous
// state, external gradient): requested here into a[%d:%d], still in flight at the end, handed over by the NEXT statement; gt / hp / xt: those of THIS phase.%s
FN_DEVINL void %s(%s) {
    const void* xin = fn_uniform_ptr(reinterpret_cast<const float*>(xin_));
    arr = __builtin_amdgcn_readfirstlane(arr);
%s    asm volatile(
%s
        : %s
        : [xin] "s"(xin), [vo] "v"(vo), [red] "v"(red), [lp] "v"(lp), [arr] "s"(arr), [pcnt] "v"(pcnt), %s
        : "memory", "scc", "vcc", "s%d", "s%d", "s%d", "s%d", %s);
%s}
""" % (self.name, self.TH, self.units, self.RU, "issues the previous epilogue's exchange-slab stores" if self.stores else "no stores to issue",
       ", the previous epilogue's dgx / dghn stores (`_out`)" if self.stores else "",
       EXT0, EXT0 + 23,
       "\n// sbase + so0: exchange slab of the previous epilogue (uniform) and this lane's byte offset of gate 0 in it; t<gate><piece>: its gate gradients dr', dz', dn' r as bf16 triples" if self.stores else "",
       self.name, sig, decl, body, outs, ins, SB, SB + 1, SB + 2, SB + 3, self.clobbers(), post)

    def emit_pro(self, name):
        L = (["v_add_u32 v%d, 0x%x, %%[vo]" % (VO2, TILE)] if self.TH == 2 else []) + ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1), "s_nop 4"]
        base_blk = 0
        for u in range(self.RU - 1):
            nb = u // self.TH
            if nb != base_blk:
                L += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, (nb - base_blk) * UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
                base_blk = nb
            L += self.load_unit(u)
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join(['"a%d"' % i for i in range(RING0, RING0 + 12 * (self.RU - 1))] + ['"v%d"' % VO2])
        return """
// ring request of a backward phase (units 0 .. %d, three pieces each, %d row tile(s))
FN_DEVINL void %s(const void* xin_, unsigned vo) {
    const void* xin = fn_uniform_ptr(reinterpret_cast<const float*>(xin_));
    asm volatile(
%s
        :
        : [xin] "s"(xin), [vo] "v"(vo)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (self.RU - 2, self.TH, name, body, SB, SB + 1, clob)


def emit_ext(name):
    L = ["global_load_dwordx4 a[%d:%d], %%[ga], off offset:%d" % (EXT0 + 4 * q, EXT0 + 4 * q + 3, q * 1024) + NT for q in range(4)]
    L += ["global_load_dwordx4 a[%d:%d], %%[ha], off" % (EXT0 + 16, EXT0 + 19) + NT, "global_load_dwordx4 a[%d:%d], %%[xa], off" % (EXT0 + 20, EXT0 + 23) + NT]
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    clob = ", ".join('"a%d"' % i for i in range(EXT0, EXT0 + 24))
    return """
// the epilogue operands of the FIRST K phase of a launch (every later phase's are requested by the statement in front of it): saved gates r, z, n, hn
// (1 KB apart), previous state, external gradient -> a[%d:%d], left in flight
FN_DEVINL void %s(const float* ga, const float* ha, const float* xa) {
    asm volatile(
%s
        :
        : [ga] "v"(ga), [ha] "v"(ha), [xa] "v"(xa)
        : "memory", %s);
}
""" % (EXT0, EXT0 + 23, name, body, clob)


def emit_out(name, masked):
    L = ["global_store_dwordx4 %[sg], %[d0], off offset:-2048" + NT, "global_store_dwordx4 %[sg], %[d1], off" + NT,
         "global_store_dwordx4 %[sg], %[d2], off offset:2048" + NT, "global_store_dwordx4 %[sg], %[d3], off".replace("%[sg], %[d3]", "%[sn], %[d3]") + NT]
    if masked:
        L = ["s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 0xffffffff"] + L + ["s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3)]
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    return """
// the stores of an epilogue item%s nobody in the launch waits for: dgx r / z / n (sg = the middle gate, 2 KB apart) and dghn (sn); d0..d3 = dr', dz', dn', dn' r.
// Counted by the next K loop statement (four operations behind the ring request).
FN_DEVINL void %s(float* sg, float* sn, const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3) {
    asm volatile(
%s
        :
        : [sg] "v"(sg), [sn] "v"(sn), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3)
        : "memory", "s%d", "s%d");
}
""" % (" (lanes 0-31)" if masked else "", name, body, SB + 2, SB + 3)


def emit_pub(name, masked):
    """the 9 exchange-slab stores of an epilogue item: gate g at byte offset 0xC000 g (16 K blocks), piece p 1 KB further"""
    # sbase reaches the statement through v_readfirstlane: a VALU-written SGPR needs 5 wait states before a vector-memory instruction may use it as its
    # scalar base, and the compiler inserts none in front of inline asm (found the hard way: the first store went to {high word, 0} + offset)
    L = ["s_nop 4"]
    for g in range(3):
        if g:
            L.append("v_add_u32 v%d, 0x%x, %%[so0]" % (VO2, 0xC000 * g))
        for pc in range(3):
            L.append("global_store_dwordx2 %s, %%[t%d%d], %%[sbase] offset:%d sc1" % ("%[so0]" if g == 0 else "v%d" % VO2, g, pc, pc * 1024))
    if masked:
        L = ["s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 0xffffffff"] + L + ["s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3)]
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    sig = ", ".join("const u32x2& t%d%d" % (g, pc) for g in range(3) for pc in range(3))
    ins = ", ".join('[t%d%d] "v"(t%d%d)' % (g, pc, g, pc) for g in range(3) for pc in range(3))
    return """
// publication of an epilogue item%s outside of a K loop (iteration 0 of a launch): its gate gradients dr', dz', dn' r as bf16 triples (t<gate><piece>, four
// packed values each) on the exchange slab sbase (uniform) at this lane's byte offset so0 - write-through
FN_DEVINL void %s(void* sbase_, unsigned so0, %s) {
    void* sbase = const_cast<float*>(fn_uniform_ptr(reinterpret_cast<const float*>(sbase_)));
    asm volatile(
%s
        :
        : [sbase] "s"(sbase), [so0] "v"(so0), %s
        : "memory", "s%d", "s%d", "v%d");
}
""" % (" (lanes 0-31)" if masked else "", name, sig, body, ins, SB + 2, SB + 3, VO2)


def emit_wload(name):
    """one-time: this wave's K quarter of column tile 0 (12 blocks x 3 KB, contiguous in the fn_frag3_pack image) -> a[0:143] and block 11 of column
    tile 1 -> a[144:155]"""
    L = ["s_mov_b64 s[%d:%d], %%[src0]" % (SB, SB + 1), "s_nop 4"]
    for blk in range(NBLK):
        if blk:
            L += ["s_add_u32 s%d, s%d, 0x%x" % (SB, SB, UB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
        for pc in range(3):
            r = W0 + 12 * blk + 4 * pc
            L.append("global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d" % (r, r + 3, SB, SB + 1, pc * 1024))
    L += ["s_mov_b64 s[%d:%d], %%[src1]" % (SB, SB + 1), "s_nop 4"]
    for pc in range(3):
        L.append("global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d" % (W1X + 4 * pc, W1X + 4 * pc + 3, SB, SB + 1, pc * 1024))
    L.append("s_waitcnt vmcnt(0)")
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    clob = ", ".join('"a%d"' % i for i in range(W0, W1X + 12))
    return """
// one-time: the register-stationary part of the W_hh^T slice.  src0 = this wave's first K block of column tile 0 in the triple image, src1 = block 11 of
// its quarter of column tile 1 (both uniform), vo = lane * 16.  a[0:155] are read by every K loop statement and never written again.
FN_DEVINL void %s(const void* src0_, const void* src1_, unsigned vo) {
    const void* src0 = fn_uniform_ptr(reinterpret_cast<const float*>(src0_));
    const void* src1 = fn_uniform_ptr(reinterpret_cast<const float*>(src1_));
    asm volatile(
%s
        :
        : [src0] "s"(src0), [src1] "s"(src1), [vo] "v"(vo)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (name, body, SB, SB + 1, clob)


HEAD = """// GENERATED by gen_kloop4.py - do not edit.  K loops of the ping-pong, register-stationary backward scan on the bf16 MFMA with exact bf16 triple splits (H = 512).
#pragma once
#include "kloop3_asm.h"
"""

# TH -> (ring depth, unit of the arrival, unit whose MFMAs load the other half's counter)
CONFIG = {"t2": (2, 6, 6, 21), "t1": (1, 6, 6, 10)}


def main(path, overrides=()):
    out = [HEAD]
    cfg = dict(CONFIG)
    for o in overrides:                               # tuning builds: t1=RU,u_arr,poll
        tag, vals = o.split("=")
        cfg[tag] = cfg[tag][:1] + tuple(int(v) for v in vals.split(","))
    for tag in ("t2", "t1"):
        TH, RU, u_arr, poll = cfg[tag]
        for stores in (1, 0):
            out.append(GenX6B("fn_x6_bwd_%s_%s" % (tag, "main" if stores else "first"), TH, RU, stores, u_arr, poll).emit_main())
        out.append(GenX6B("x", TH, RU, 0, u_arr, poll).emit_pro("fn_x6_bwd_%s_pro" % tag))
        out.append(emit_pub("fn_x6_bwd_%s_pub" % tag, TH == 1))
        out.append(emit_out("fn_x6_bwd_%s_out" % tag, TH == 1))
    out.append(emit_ext("fn_x6_bwd_ext"))
    out.append(emit_wload("fn_x6_bwd_wload"))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kloop4_asm.h", sys.argv[2:])
