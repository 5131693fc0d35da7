#!/usr/bin/env python3
"""Generates kloop2_asm.h: the K loops of the PING-PONG weight-stationary GRU scans (gru_persist.hip, gru_*_pp_kernel) at H = 512.

Round 4.  What the in-kernel stamps of the round-3 scans say (profiles/r04_scan_stamps.txt): per time step the MFMA pipe idles for
  poll of the arrival counter 0.5-2 k cycles | first operand chunks from L2 ~2 k | gate epilogue 1.4-1.9 k | store drain 0.6-1.5 k | arrive 0.2 k
and scratch/mfma_fill.hip says that a v_mfma_f32_16x16x4_f32 stream does NOT hide VALU work of its own wave (one v_add behind every MFMA:
32.3 -> 45.5 cycles per MFMA) while LDS reads are free up to two per MFMA and vector-memory instructions cost their issue slot: the
epilogue's arithmetic cannot disappear, the LATENCIES and the STORE ISSUE can.  So a workgroup owns two halves of its row group (every wave:
one 16-row tile of each) and runs them alternately:

    phase f = (half X, step p):   K loop of X  ->  accumulators to LDS  ->  gate epilogue of X (arithmetic only)

  * the operand ring of phase f+1 (the other half: its inputs were published a whole K loop ago) is requested at the END of phase f's K loop,
    BEFORE the epilogue of f: it lands while the epilogue computes, phase f+1 starts multiplying at once;
  * the STORES of phase f's epilogue (exchange slab first, then the outputs nobody waits for) are issued INSIDE phase f+1's K loop, one per
    K unit, each in the shadow of an MFMA (24 store instructions per CU and phase cost ~1 k cycles of issue when they follow each other);
  * the arrival for those stores is made there as well (counted s_waitcnt behind the slab stores, s_barrier, one atomic by wave 0): nobody
    waits for a store drain;
  * the counter of the other half is read inside the K loop too (an ordinary load, ~2.5 k cycles before the end); only when it is short
    does the wave fall back to a polling loop (and requests the ring itself, `*_pro`).

One asm statement per phase (`*_main`; `*_first` = the variant with no stores to issue: first K phase of a launch).  State that crosses
statements: the ring (RU - 1 units in flight) and the first weight fragments, in fixed AGPRs that the compiler never touches, and the vmcnt
arithmetic: between two statements the compiler issues NO vector-memory instruction (scratch/check_pp_isa.py checks the ISA).

Pipeline unit = 16 K values of ONE row tile: 1 operand load (1 KB), NB weight-fragment reads, 12 (forward: 3 gate tiles x 4 k-steps) or 4
(backward) MFMAs.  The accumulation order of every accumulator equals the round-3 loops': bit-identical sums.

Register maps (a = AGPR):  acc a[0..nacc), ring[slot] a[nacc + 4 slot ..], wfrag[bs][n] a[w0 + 4 (NB bs + n) ..].  Scalars s84..s87,
temporaries v200, v201 (backward).
"""
import os
import sys

# Experiment switches (KLOOP2_* environment variables, used by scratch/ variant scripts only) change wait counts / hints of the generated statements: a
# production build must not pick up a stray one.  They are honoured only when KLOOP_EXPERIMENT=1 is set with them; otherwise the generator refuses.
_stray = sorted(k for k in os.environ if k.startswith("KLOOP2_"))
if _stray and os.environ.get("KLOOP_EXPERIMENT") != "1":
    sys.exit("gen_kloop2.py: experiment switches %s are set without KLOOP_EXPERIMENT=1 - refusing to generate a production header" % ", ".join(_stray))

SB = 84          # s84:85 = running operand base, s86:87 = saved exec / scratch
TV = 200         # v200, v201: address temporaries of the backward slab stores


class Gen:
    def __init__(self, name, fwd, units, RU, stores, masked):
        self.name, self.fwd, self.units, self.RU, self.stores, self.masked = name, fwd, units, RU, stores, masked
        assert units % RU == 0 and RU % 2 == 0
        self.G = units // RU
        self.NB = 3 if fwd else 1
        self.nmf = 12 if fwd else 4
        self.nacc = 12 if fwd else 8
        self.ring0 = self.nacc
        self.w0 = self.ring0 + RU * 4
        self.nagpr = self.w0 + 2 * self.NB * 4
        self.wslots = [2, 5, 8] if fwd else [1]
        self.t_store, self.t_extra, self.t_tail = (6, 10, 9) if fwd else (3, 2, 2)
        self.KC = 2 if fwd else 7                  # the counter is looked at behind unit u0 + KC (~1.1 k cycles after its load)
        self.u_arr = 3 if fwd else 10              # ~1.2 k cycles behind the (last) slab store
        self.next_ts = [0, 3, 6, 9, 1, 4, 7, 10] if fwd else [0, 2, 1, 3]
        self.extra_lead = 14 if fwd else 40        # units between the first epilogue-operand load and the counter check
        if fwd:      # 3 gate rows of the epilogue item off one base (middle gate: +-2048 bytes) + the token of the next step
            self.extras = ["global_load_dwordx4 %%[ex%d], %%[xa], off offset:%d" % (q, (q - 1) * 2048) for q in range(3)]
            self.extras.append("global_load_dword %[tokn], %[ta], off")
            # stores of the previous epilogue: exchange slab (write-through), h_all, the four saved-gate vectors (1 KB apart)
            self.st = [("slab", ["global_store_dwordx4 %[sa0], %[d0], off sc1"]), ("out", ["global_store_dwordx4 %[sa1], %[d0], off"])]
            self.st += [("out", ["global_store_dwordx4 %%[sa2], %%[d%d], off offset:%d" % (q + 1, q * 1024)]) for q in range(4)]
        else:        # saved gates r, z, n, hn (1 KB apart), previous state, external gradient
            self.extras = ["global_load_dwordx4 %%[gt%d], %%[ga], off offset:%d" % (q, q * 1024) for q in range(4)]
            self.extras += ["global_load_dwordx4 %[hp], %[ha], off", "global_load_dwordx4 %[xt], %[xa], off"]
            # exchange slab: dr', dz', dn' r at column offsets 0, H, 2H of the [rows][3H] fragment image = 32 KB apart; dgx r / z / n (2 KB apart
            # around the middle one), dghn
            self.st = [("slab", ["global_store_dwordx4 %[so0], %[d0], %[sbase] sc1"]),
                       ("slab", ["v_add_u32 v%d, 0x8000, %%[so0]" % TV, "global_store_dwordx4 v%d, %%[d1], %%[sbase] sc1" % TV]),
                       ("slab", ["v_add_u32 v%d, 0x10000, %%[so0]" % (TV + 1), "global_store_dwordx4 v%d, %%[d3], %%[sbase] sc1" % (TV + 1)]),
                       ("out", ["global_store_dwordx4 %[sg], %[d0], off offset:-2048"]), ("out", ["global_store_dwordx4 %[sg], %[d1], off"]),
                       ("out", ["global_store_dwordx4 %[sg], %[d2], off offset:2048"]), ("out", ["global_store_dwordx4 %[sn], %[d3], off"])]

    def ring(self, slot):
        return self.ring0 + 4 * slot

    def refill_ins(self, slot, imm):
        """(MFMA gap, instruction) pairs that re-fill ring slot `slot` from byte offset imm off the running base"""
        r = self.ring(slot)
        return [(0, "global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d sc1" % (r, r + 3, SB, SB + 1, imm))]

    def next_ring_ins(self, slot, off):
        r = self.ring(slot)
        return ["global_load_dwordx4 a[%d:%d], %%[voy], s[%d:%d] offset:%d sc1" % (r, r + 3, SB, SB + 1, off)]

    def wf(self, bs, n):
        return self.w0 + 4 * (self.NB * bs + n)

    def mfmas_u(self, slot, bs, u):
        return self.mfmas(slot, bs)

    def mfmas(self, slot, bs):
        out = []
        for jj in range(4):
            for x in (range(3) if self.fwd else [jj & 1]):
                a = self.ring(slot) + jj
                b = (self.wf(bs, x) if self.fwd else self.wf(bs, 0)) + jj
                c = 4 * x
                out.append("v_mfma_f32_16x16x4_f32 a[%d:%d], a%d, a%d, a[%d:%d]" % (c, c + 3, a, b, c, c + 3))
        return out

    # ---- ring request of a phase: units 0 .. RU-2 into slots 0 .. RU-2 off the scalar base in s[SB:SB+1] (clobbered), + weight fragments of unit 0
    def request(self, xin, vo):
        L = ["s_mov_b64 s[%d:%d], %s" % (SB, SB + 1, xin), "s_nop 4"]
        rel = 0
        for u in range(self.RU - 1):
            while u * 1024 - rel > 4095:
                L += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
                rel += 4096
            r = self.ring(u)
            L.append("global_load_dwordx4 a[%d:%d], %s, s[%d:%d] offset:%d sc1" % (r, r + 3, vo, SB, SB + 1, u * 1024 - rel))
        for n in range(self.NB):
            L.append(self.wread_at(0, n, 0))
        return L

    def wread_at(self, bs, n, unit):
        """weight fragments of K unit `unit` (absolute inside this wave's K range); no pointer is ever advanced"""
        r = self.wf(bs, n)
        if self.fwd:      # wl[3][nk][2][64][4] floats, gate stride 32 KB at H = 512: gates 0, 1 off lp, gate 2 off lq = lp + 64 KB
            ptr, off = ("%[lp]", n * 32768 + unit * 1024) if n < 2 else ("%[lq]", unit * 1024)
        else:             # wl[nk3][2][64][4]: units 0-47 off lp, 48-95 off lq = lp + 48 KB
            ptr, off = ("%[lp]", unit * 1024) if unit < 48 else ("%[lq]", (unit - 48) * 1024)
        assert 0 <= off < 65536
        return "ds_read_b128 a[%d:%d], %s offset:%d" % (r, r + 3, ptr, off)

    def advance_to(self, need_rel):
        out = []
        while need_rel - self.s_rel > 3072:
            out += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            self.s_rel += 4096
        assert 0 <= need_rel - self.s_rel <= 3072, (need_rel, self.s_rel)
        return out

    def wait_unit(self, unit_abs):
        last = max(i for i, o in enumerate(self.vmops) if o == ("ring", unit_abs))
        n = len(self.vmops) - 1 - last
        assert n < 60, n
        if os.environ.get("KLOOP2_RING0") == "1":        # experiment (round 5): every ring wait drains the counter
            n = 0
        if os.environ.get("KLOOP2_RING0_RANGE"):         # experiment: only the waits of units lo..hi (inside this wave's K range) drain it
            lo, hi = (int(x) for x in os.environ["KLOOP2_RING0_RANGE"].split("-"))
            if lo <= unit_abs <= hi:
                n = 0
        if os.environ.get("KLOOP2_NOSTORE") == "1":      # experiment (round 5): younger STORES are not assumed to stay behind the ring load (loads only)
            n = sum(1 for o in self.vmops[last + 1:] if o[0] != "store")
        if os.environ.get("KLOOP2_NOEXTRA") == "1":      # experiment (round 5): neither are the plain (non-sc1) loads of the epilogue operands
            n = sum(1 for o in self.vmops[last + 1:] if o[0] != "extra")
        return "s_waitcnt vmcnt(%d)" % n

    def arrive_block(self):
        """the slab stores of the previous epilogue (or, without stores, everything older than this statement) have completed in every
        wave -> one arrival (arr: 0 = none due, 1 = due, 2 = due and this wave issues it)"""
        slab = [i for i, o in enumerate(self.vmops) if o == ("store", "slab")]
        n = (len(self.vmops) - 1 - max(slab)) if slab else (len(self.vmops) - self.n0)
        if os.environ.get("KLOOP2_ARR0") == "1":         # experiment (round 5): the arrival waits for everything in flight
            n = 0
        return ["s_cmp_eq_u32 %[arr], 0", "s_cbranch_scc1 .Lnoarr_%=", "s_waitcnt vmcnt(%d)" % n, "s_barrier",
                "s_cmp_lt_u32 %[arr], 2", "s_cbranch_scc1 .Lnoarr_%=",
                "s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 1", "v_mov_b32 %[pv], 1",
                "global_atomic_add %[acnt], %[pv], off", "s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3), ".Lnoarr_%=:"]

    def store_ins(self, ins):
        if not self.masked:
            return ins
        return ["s_mov_b64 s[%d:%d], exec" % (SB + 2, SB + 3), "s_mov_b64 exec, 0xffffffff"] + ins + ["s_mov_b64 exec, s[%d:%d]" % (SB + 2, SB + 3)]

    def body(self):
        RU, G = self.RU, self.G
        assert G >= 2
        self.vmops = [("ring", u) for u in range(RU - 1) for _ in range(getattr(self, "TH", 1))]      # in flight when the statement starts: the ring request and nothing else
        self.n0 = len(self.vmops)
        self.s_rel = 0
        L = ["s_mov_b64 s[%d:%d], %%[xin]" % (SB, SB + 1)]
        start_rel = (RU - 1) * 1024 - 3072
        while self.s_rel < start_rel:
            L += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)]
            self.s_rel += 4096
        assert self.s_rel == start_rel
        for c in range(self.nacc):
            L.append("v_accvgpr_write_b32 a%d, 0" % c)
        pending = list(self.extras)
        stores = list(self.st) if self.stores else []
        nslab = sum(1 for k, _ in stores if k == "slab")
        u_arr = self.u_arr if self.stores else 0                       # no stores: at once
        assert u_arr >= nslab and u_arr < self.units - RU
        u0 = (G - 1) * RU
        # the epilogue operands of THIS phase (HBM reads): early enough to have retired (in order) when the counter is looked at (its check
        # waits for vmcnt(0)); the ring loads issued behind them are not needed before RU - 1 units later
        ne = len(self.extras)
        e0 = max(1, min(u0 - ne, u0 + self.KC - self.extra_lead))       # >= 2.3 k cycles (HBM latency under load) in front of the counter check
        extra_units = {e0 + j: j for j in range(ne)}
        wslots, t_store, t_extra, KC = self.wslots, self.t_store, self.t_extra, self.KC

        def unit(g, k, path):
            """instructions of unit (g, k); path: None = common part, "R" = the next phase's ring is being requested, "N" = it is not"""
            out = []
            final = g == G - 1
            u = g * RU + k
            refill = (not final) or k == 0
            wnext = (not final) or k < RU - 1
            if path is None:
                out.append(self.wait_unit(u))
                if os.environ.get("KLOOP2_NOP"):             # experiment (round 5): wait states between the counted wait and the first MFMA that reads the ring slot
                    out.append("s_nop %d" % (int(os.environ["KLOOP2_NOP"]) - 1))
            out.append("s_waitcnt lgkmcnt(0)")
            comp = [[] for _ in range(self.nmf)]
            issued = []            # (MFMA slot, order inside the slot, what) of this unit's memory operations: self.vmops must list them in ISSUE order -
            # an epilogue-operand load sits in an earlier slot than the store of the same unit, and the arrival's wait counts the operations BEHIND
            # the last slab store (round 5: listed in append order, that wait of fn_rs_bwd_t1_main / fn_pp_bwd_k768_main was one too lenient - the
            # last slab store could still be in flight when the arrival was posted)
            if refill:
                imm = (k + RU - 1) * 1024 - self.s_rel
                assert 0 <= imm <= 4095
                for t_, ins_ in self.refill_ins((k - 1) % RU, imm):
                    comp[t_].append(ins_)
                    issued.append((t_, len(issued), ("ring", u + RU - 1)))
            if final and k == 0:
                # the last refill is out: the counter of the NEXT phase's half, looked at KC units later
                comp[1].append("global_load_dword %[pv], %[pcnt], off sc1")
                issued.append((1, len(issued), ("poll", 0)))
            if path == "R":
                # ring of the next phase: slot s is free once unit u0 + s has been multiplied.  Unit KC + 1 requests slots 0 .. KC, every later one its predecessor's
                slots = list(range(0, KC + 1)) if k == KC + 1 else [k - 1]
                ts = self.next_ts
                n_ = 0
                for sl in slots:
                    if sl > RU - 2:
                        continue
                    assert -4096 <= sl * 1024 - self.y_rel <= 4095
                    for ins_ in self.next_ring_ins(sl, sl * 1024 - self.y_rel):
                        comp[ts[n_ % len(ts)]].append(ins_)
                        n_ += 1
            if stores and path is None and not final:                 # one store of the previous epilogue per unit, the exchange slab first
                kind, ins = stores.pop(0)
                comp[t_store] += self.store_ins(ins)
                issued.append((t_store, len(issued), ("store", kind)))
            if wnext:
                for n, t in enumerate(wslots):
                    comp[t].append(self.wread_at((k + 1) & 1, n, u + 1))
            if u in extra_units:
                comp[t_extra].append(pending.pop(0))
                issued.append((t_extra, len(issued), ("extra", 0)))
            tail = []
            if (not final) and k + 1 < RU:
                tail += self.advance_to((k + 1 + RU - 1) * 1024)
            if k == RU - 1 and not final:
                tail += self.advance_to(RU * 1024 + start_rel + 3072)
                assert self.s_rel == RU * 1024 + start_rel
            t_tail = self.t_tail
            for t, ins in enumerate(self.mfmas_u(k, k & 1, u)):
                out.append(ins)
                out += comp[t]
                if t >= t_tail and tail and (not comp[t] or t == self.nmf - 1):
                    out.append(tail.pop(0))
                    if tail and tail[0].startswith("s_addc"):
                        out.append(tail.pop(0))
            out += tail
            self.vmops += [e for _, _, e in sorted(issued)]
            if u == u_arr:
                out += self.arrive_block()
            return out

        for g in range(G - 1):
            assert self.s_rel == start_rel
            for k in range(RU):
                L += unit(g, k, None)
            self.s_rel -= RU * 1024
        assert not pending and not stores
        for k in range(KC + 1):
            L += unit(G - 1, k, None)
        # ---- the counter: everything requested so far has landed (own ring, epilogue operands, the counter value)
        L += ["s_waitcnt vmcnt(0)", "v_readfirstlane_b32 s%d, %%[pv]" % (SB + 2), "s_cmp_ge_u32 s%d, %%[ptgt]" % (SB + 2), "s_cbranch_scc0 .Lnoreq_%="]
        # path R: the remaining units with the next phase's ring request between their MFMAs
        L += ["s_mov_b64 s[%d:%d], %%[xiny]" % (SB, SB + 1)]
        self.y_rel = 0
        for k in range(KC + 1, RU):
            # scalar base of the next phase's operands: advance in 4 KB steps so that slot offsets stay below 4096
            need = (k - 1) * 1024
            if need - self.y_rel > 3072:
                L += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
                self.y_rel += 4096
                # slots requested by unit KC + 1 below the new base were issued before the advance (they are at offsets < 4096 of the old base)
            L += unit(G - 1, k, "R")
        L += ["s_branch .Ldone_%=", ".Lnoreq_%=:"]
        for k in range(KC + 1, RU):
            L += unit(G - 1, k, "N")
        L += [".Ldone_%=:"]
        for n in range(self.NB):          # weight fragments of the next phase's unit 0 (the same slice for both halves)
            L.append(self.wread_at(0, n, 0))
        L += ["s_nop 15"] * int(os.environ.get("KLOOP2_ENDNOP", "1"))      # (experiment switch, round 5: more wait states in front of the accumulator hand-over)
        # accumulators -> LDS (padded MFMA C layout); an MFMA result needs 12 wait states before anything but an accumulating MFMA reads it
        L += self.acc_writes()
        L.append("s_waitcnt lgkmcnt(0)")
        return L

    def acc_writes(self):
        if self.fwd:
            return ["ds_write_b128 %%[red], a[%d:%d] offset:%d" % (4 * n, 4 * n + 3, n * 1088) for n in range(3)]
        return ["ds_write_b128 %%[red], a[%d:%d] offset:%d" % (4 * par, 4 * par + 3, par * 8 * 1088) for par in range(2)]

    def emit_main(self):
        L = self.body()
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join(['"a%d"' % i for i in range(self.nagpr)] + ([] if self.fwd or not self.stores else ['"v%d"' % TV, '"v%d"' % (TV + 1)]))
        sig = ("const float* xin_, unsigned vo, unsigned lp, unsigned lq, unsigned red,\n"
               "        int arr, u32* acnt, const u32* pcnt, unsigned ptgt, const float* xiny_, unsigned voy,\n")
        pre = ""
        if self.fwd:
            sig += "        const float* xa, const int* ta"
            if self.stores:
                sig += ",\n        float* sa0, float* sa1, float* sa2, const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3, const f32x4& d4"
            sig += ",\n        f32x4 (&ex)[3], int& tokn, unsigned& pv"
            outs = ", ".join(['[ex%d] "=&v"(ex[%d])' % (q, q) for q in range(3)] + ['[tokn] "=&v"(tokn)', '[pv] "=&v"(pv)'])
            ins = '[xa] "v"(xa), [ta] "v"(ta)'
            if self.stores:
                ins += ', [sa0] "v"(sa0), [sa1] "v"(sa1), [sa2] "v"(sa2), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [d4] "v"(d4)'
        else:
            sig += "        const float* ga, const float* ha, const float* xa"
            if self.stores:
                sig += ",\n        float* sbase_, unsigned so0, float* sg, float* sn, const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3"
                pre = "    float* sbase = const_cast<float*>(fn_uniform_ptr(sbase_));\n"
            sig += ",\n        f32x4 (&gt)[4], f32x4& hp, f32x4& xt, unsigned& pv"
            outs = ", ".join(['[gt%d] "=&v"(gt[%d])' % (q, q) for q in range(4)] + ['[hp] "=&v"(hp)', '[xt] "=&v"(xt)', '[pv] "=&v"(pv)'])
            ins = '[ga] "v"(ga), [ha] "v"(ha), [xa] "v"(xa)'
            if self.stores:
                ins += ', [sbase] "s"(sbase), [so0] "v"(so0), [sg] "v"(sg), [sn] "v"(sn), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3)'
        if self.fwd:
            what = "sa0 / sa1 / sa2 + d0..d4: exchange-slab, h_all and saved-gates addresses and new state, r, z, n, W_hn h + b_hn" if self.stores else ""
        else:
            what = ("sbase + so0: exchange slab of the previous epilogue and this lane's byte offset in it, sg / sn: dgx (middle gate) and dghn addresses, "
                    "d0..d3 = dr', dz', dn', dn' r") if self.stores else ""
        return """
// %s: K loop of one phase (%d units of 16 K values, ring of %d, %d MFMAs per unit, %d AGPRs; %s%s).
// xin = this wave's first operand unit of THIS phase (uniform), vo = byte offset of its row tile (+ lane * 16); the first %d units are already
// in flight.  arr / acnt: arrival for the previous phase's epilogue (0 none, 1 due, 2 due and this wave issues it / its counter).
// pcnt / ptgt: counter and target of the next phase's half; xiny / voy: its operand base and tile offset - its ring is requested when
// pv >= ptgt at the end of this loop (pv is returned: otherwise the caller polls and requests it with the *_pro statement).%s
FN_DEVINL void %s(%s) {
    const float* xin = fn_uniform_ptr(xin_);
    const float* xiny = fn_uniform_ptr(xiny_);
    // wave-uniform by construction; an "s" operand the compiler believes divergent would be handed over in a VGPR
    arr = __builtin_amdgcn_readfirstlane(arr);
    ptgt = (unsigned)__builtin_amdgcn_readfirstlane((int)ptgt);
%s    asm volatile(
%s
        : %s
        : [xin] "s"(xin), [xiny] "s"(xiny), [vo] "v"(vo), [voy] "v"(voy), [red] "v"(red), [lp] "v"(lp), [lq] "v"(lq), [arr] "s"(arr),
          [acnt] "v"(acnt), [pcnt] "v"(pcnt), [ptgt] "s"(ptgt), %s
        : "memory", "scc", "vcc", "s%d", "s%d", "s%d", "s%d", %s);
}
""" % (self.name, self.units, self.RU, self.nmf, self.nagpr, "issues the previous epilogue's stores" if self.stores else "no stores to issue",
       ", lanes 0-31 store" if self.masked and self.stores else "", self.RU - 1, ("\n// " + what) if what else "", self.name, sig, pre, body, outs, ins,
       SB, SB + 1, SB + 2, SB + 3, clob)

    def emit_pro(self, name):
        L = self.request("%[xin]", "%[vo]")
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join('"a%d"' % i for i in list(range(self.ring0, self.ring0 + (self.RU - 1) * 4)) + list(range(self.w0, self.w0 + self.NB * 4)))
        return """
// ring request of a phase (units 0 .. %d) + the weight fragments of unit 0: what the previous phase's statement does at its end when the
// counter was already there
FN_DEVINL void %s(const float* xin_, unsigned vo, unsigned lp, unsigned lq) {
    const float* xin = fn_uniform_ptr(xin_);
    asm volatile(
%s
        :
        : [xin] "s"(xin), [vo] "v"(vo), [lp] "v"(lp), [lq] "v"(lq)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (self.RU - 2, name, body, SB, SB + 1, clob)


class GenRS(Gen):
    """backward K loop with HALF of the workgroup's W_hh^T slice REGISTER-stationary (round 4).

    The backward product has one third of the forward's operand reuse (a workgroup of the kernels above owns 16 dh columns x all 3H gate-gradient
    columns: every 1 KB operand load feeds 4 MFMAs, 768 KB per workgroup and step stream from L2 - the K loop is bound by that stream and by
    the issue slots of its 192 loads).  Here a workgroup owns 32 dh columns of HALF as many rows: the slice of the first 16 columns lives in
    AGPRs (each of the 4 waves holds its K quarter: 24 units x 4 registers = 96), the slice of the other 16 in LDS as before; every wave
    multiplies ALL row tiles of the current half over its K quarter, the four partial sums meet in LDS (the epilogue adds them in wave order).
    One operand load now feeds 8 MFMAs, the operand stream per workgroup halves (384 KB per step for 64 rows), weight-fragment LDS reads halve.
    TH = row tiles per half (2: 64-row groups, 1: 32-row groups).

    Register map (a = AGPR): acc[m][ct] a[4 (2 m + ct) ..], ring[slot][m] a[8 TH + 4 (TH slot + m) ..], wfrag[bs] (LDS half), W[u] a[wr0 + 4 u ..] (never
    written after the preload).  v200-v203: address temporaries.
    """

    def __init__(self, name, TH, RU, stores):
        Gen.__init__(self, name, False, 24, RU, stores, TH == 1)
        self.TH = TH
        self.NB = 1
        self.nmf = 8 * TH
        self.nacc = 8 * TH
        self.ring0 = self.nacc
        self.w0 = self.ring0 + RU * TH * 4
        self.wr0 = self.w0 + 8
        self.nagpr = self.wr0 + 96
        self.wslots = [2]
        self.t_store, self.t_extra, self.t_tail = (6, 10, 12) if TH == 2 else (5, 3, 6)
        self.KC = 2 if TH == 2 else 4
        self.u_arr = 4 if TH == 2 else 7
        self.next_ts = [0, 2, 4, 6, 8, 10, 12, 14] if TH == 2 else [0, 1, 2, 3, 4, 5, 6, 7]
        self.extra_lead = 12 if TH == 2 else 14

    def ring(self, slot, m=0):
        return self.ring0 + 4 * (self.TH * slot + m)

    def vreg(self, m, y=False):
        return ("%[voy]" if y else "%[vo]") if m == 0 else "v%d" % (TV + 3 if y else TV + 2)

    def refill_ins(self, slot, imm):
        out = []
        for m in range(self.TH):
            r = self.ring(slot, m)
            out.append((4 * m, "global_load_dwordx4 a[%d:%d], %s, s[%d:%d] offset:%d sc1" % (r, r + 3, self.vreg(m), SB, SB + 1, imm)))
        return out

    def next_ring_ins(self, slot, off):
        return ["global_load_dwordx4 a[%d:%d], %s, s[%d:%d] offset:%d sc1" % (self.ring(slot, m), self.ring(slot, m) + 3, self.vreg(m, True), SB, SB + 1, off)
                for m in range(self.TH)]

    def request(self, xin, vo):
        y = vo == "%[voy]"
        L = ["s_mov_b64 s[%d:%d], %s" % (SB, SB + 1, xin), "s_nop 4"]
        rel = 0
        for u in range(self.RU - 1):
            while u * 1024 - rel > 4095:
                L += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
                rel += 4096
            for m in range(self.TH):
                r = self.ring(u, m)
                L.append("global_load_dwordx4 a[%d:%d], %s, s[%d:%d] offset:%d sc1" % (r, r + 3, self.vreg(m, y), SB, SB + 1, u * 1024 - rel))
        L.append(self.wread_at(0, 0, 0))
        return L

    def mfmas_u(self, slot, bs, u):
        out = []
        for jj in range(4):
            for m in range(self.TH):
                a = self.ring(slot, m) + jj
                for ct, b in ((0, self.wr0 + 4 * u + jj), (1, self.wf(bs, 0) + jj)):
                    c = 4 * (2 * m + ct)
                    out.append("v_mfma_f32_16x16x4_f32 a[%d:%d], a%d, a%d, a[%d:%d]" % (c, c + 3, a, b, c, c + 3))
        return out

    def acc_writes(self):
        return ["ds_write_b128 %%[red], a[%d:%d] offset:%d" % (4 * t, 4 * t + 3, t * 1088) for t in range(2 * self.TH)]

    def body(self):
        # the second row tile's offsets (one row tile = nk3 * 2 KB further on) live in v202 / v203 for the whole statement
        pre = ["v_add_u32 v%d, 0x18000, %%[vo]" % (TV + 2), "v_add_u32 v%d, 0x18000, %%[voy]" % (TV + 3)] if self.TH == 2 else []
        # the ring holds TH loads per unit: the base class counts one vmop per ring load, which refill_ins / the initial state provide
        L = Gen.body(self)
        return pre + L

    def emit_main(self):
        # initial ring state: TH loads per unit (Gen.body starts from one per unit: patch the bookkeeping through a subclass hook)
        L = self.body()
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join(['"a%d"' % i for i in range(self.wr0)] + ['"v%d"' % (TV + i) for i in range(4)])
        sig = ("const float* xin_, unsigned vo, unsigned lp, unsigned red,\n"
               "        int arr, u32* acnt, const u32* pcnt, unsigned ptgt, const float* xiny_, unsigned voy,\n"
               "        const float* ga, const float* ha, const float* xa")
        ins = '[ga] "v"(ga), [ha] "v"(ha), [xa] "v"(xa)'
        pre = ""
        if self.stores:
            sig += ",\n        float* sbase_, unsigned so0, float* sg, float* sn, const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3"
            pre = "    float* sbase = const_cast<float*>(fn_uniform_ptr(sbase_));\n"
            ins += ', [sbase] "s"(sbase), [so0] "v"(so0), [sg] "v"(sg), [sn] "v"(sn), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3)'
        sig += ",\n        f32x4 (&gt)[4], f32x4& hp, f32x4& xt, unsigned& pv"
        outs = ", ".join(['[gt%d] "=&v"(gt[%d])' % (q, q) for q in range(4)] + ['[hp] "=&v"(hp)', '[xt] "=&v"(xt)', '[pv] "=&v"(pv)'])
        return """
// %s: register-stationary backward K loop of one phase (%d row tile(s) x 2 column tiles, 24 units of 16 K values = this wave's K quarter, ring of %d,
// %d MFMAs per unit; %s).  Operands as in the fn_pp_bwd_* statements; vo / voy = byte offset of the FIRST row tile of the half (the second one
// follows nk3 * 2 KB further on); lp = LDS byte address of this wave's first weight-fragment unit of the LDS half of the slice.
FN_DEVINL void %s(%s) {
    const float* xin = fn_uniform_ptr(xin_);
    const float* xiny = fn_uniform_ptr(xiny_);
    arr = __builtin_amdgcn_readfirstlane(arr);
    ptgt = (unsigned)__builtin_amdgcn_readfirstlane((int)ptgt);
%s    asm volatile(
%s
        : %s
        : [xin] "s"(xin), [xiny] "s"(xiny), [vo] "v"(vo), [voy] "v"(voy), [red] "v"(red), [lp] "v"(lp), [arr] "s"(arr),
          [acnt] "v"(acnt), [pcnt] "v"(pcnt), [ptgt] "s"(ptgt), %s
        : "memory", "scc", "vcc", "s%d", "s%d", "s%d", "s%d", %s);
}
""" % (self.name, self.TH, self.RU, self.nmf, "issues the previous epilogue's stores" if self.stores else "no stores to issue", self.name, sig, pre, body, outs, ins,
       SB, SB + 1, SB + 2, SB + 3, clob)

    def emit_pro(self, name):
        L = (["v_add_u32 v%d, 0x18000, %%[vo]" % (TV + 2)] if self.TH == 2 else []) + self.request("%[xin]", "%[vo]")
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        regs = list(range(self.ring0, self.ring0 + (self.RU - 1) * self.TH * 4)) + list(range(self.w0, self.w0 + 4))
        clob = ", ".join(['"a%d"' % i for i in regs] + ['"v%d"' % (TV + 2)])
        return """
// ring request of a phase (units 0 .. %d, %d row tile(s)) + the LDS-half weight fragments of unit 0
FN_DEVINL void %s(const float* xin_, unsigned vo, unsigned lp) {
    const float* xin = fn_uniform_ptr(xin_);
    asm volatile(
%s
        :
        : [xin] "s"(xin), [vo] "v"(vo), [lp] "v"(lp)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (self.RU - 2, self.TH, name, body, SB, SB + 1, clob)

    def emit_wload(self, name):
        """the register-stationary half of the slice: this wave's 24 units of column tile 0 -> a[wr0 .. wr0 + 96)"""
        L = ["s_mov_b64 s[%d:%d], %%[src]" % (SB, SB + 1), "s_nop 4"]
        rel = 0
        for u in range(24):
            while u * 1024 - rel > 4095:
                L += ["s_add_u32 s%d, s%d, 0x1000" % (SB, SB), "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1), "s_nop 4"]
                rel += 4096
            L.append("global_load_dwordx4 a[%d:%d], %%[vo], s[%d:%d] offset:%d" % (self.wr0 + 4 * u, self.wr0 + 4 * u + 3, SB, SB + 1, u * 1024 - rel))
        L.append("s_waitcnt vmcnt(0)")
        body = "\n".join('        "%s\\n\\t"' % l for l in L)
        clob = ", ".join('"a%d"' % i for i in range(self.wr0, self.wr0 + 96))
        return """
// one-time: the register-stationary half of the W_hh^T slice.  src = this wave's first unit of column tile 0 in the fragment image (uniform),
// vo = lane * 16.  a[%d:%d] are read by every %s statement and never written again.
FN_DEVINL void %s(const float* src_, unsigned vo) {
    const float* src = fn_uniform_ptr(src_);
    asm volatile(
%s
        :
        : [src] "s"(src), [vo] "v"(vo)
        : "memory", "scc", "s%d", "s%d", %s);
}
""" % (self.wr0, self.wr0 + 95, self.name, name, body, SB, SB + 1, clob)


HEAD = """// GENERATED by gen_kloop2.py - do not edit.  K loops of the ping-pong weight-stationary GRU scans (H = 512, one row tile per wave and phase).
#pragma once
#include "mma_core.h"
typedef unsigned int u32;

// wave-uniform by construction (depends on the wave id only); the compiler cannot see that
FN_DEVINL const float* fn_uniform_ptr(const float* p) {
    const unsigned long long q = (unsigned long long)(uintptr_t)p;
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(q & 0xffffffffull))));
}
"""


def main(path):
    out = [HEAD]
    # k512 / k1536: one wave over all of K (128-row groups, every lane has an epilogue item); k256 / k768: K split in two (64-row groups,
    # lanes 0-31 of every wave have one)
    for units, tag, masked, RUf in ((32, "k512", False, 8), (16, "k256", True, 8)):
        for stores in (1, 0):
            out.append(Gen("fn_pp_fwd_%s_%s" % (tag, "main" if stores else "first"), True, units, RUf, stores, masked).emit_main())
        out.append(Gen("x", True, units, RUf, 0, masked).emit_pro("fn_pp_fwd_%s_pro" % tag))
    for units, tag, masked in ((96, "k1536", False), (48, "k768", True)):
        for stores in (1, 0):
            out.append(Gen("fn_pp_bwd_%s_%s" % (tag, "main" if stores else "first"), False, units, 16, stores, masked).emit_main())
        out.append(Gen("x", False, units, 16, 0, masked).emit_pro("fn_pp_bwd_%s_pro" % tag))
    for TH, tag, RU in ((2, "t2", 8), (1, "t1", 12)):
        for stores in (1, 0):
            out.append(GenRS("fn_rs_bwd_%s_%s" % (tag, "main" if stores else "first"), TH, RU, stores).emit_main())
        out.append(GenRS("fn_rs_bwd_%s" % tag, TH, RU, 0).emit_pro("fn_rs_bwd_%s_pro" % tag))
        out.append(GenRS("fn_rs_bwd_%s" % tag, TH, RU, 0).emit_wload("fn_rs_bwd_%s_wload" % tag))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "kloop2_asm.h")
