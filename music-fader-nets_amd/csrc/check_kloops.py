#!/usr/bin/env python3
"""Static check of the generated K-loop statements (kloop2_asm.h, kloop3_asm.h, kloop4_asm.h): the hand-counted `s_waitcnt vmcnt(n)` against the ISSUE order of
the statement's memory operations (vector memory operations retire in order on this counter).

  1. arrival: the wait in front of an `s_barrier` must leave at most as many operations outstanding as were issued BEHIND the last exchange-slab store
     (an `sc1` store) of the statement - else the arrival can be posted while that store is still in flight.
  2. operands: when an MFMA (or a ds_write / store that reads registers) is issued, no load that is still allowed to be outstanding may have one of its source
     registers as destination.  The loads in flight when a `*_main` / `*_first` statement starts are those of the family's `*_pro` statement (the
     ring request), in that order - for the bf16 x 6 backward followed by the `*_out` stores of the last epilogue; both paths behind the counter check are walked.

Round 5: check 1 found fn_rs_bwd_t1_main and fn_pp_bwd_k768_main one operation too lenient (the generator listed a unit's operations in the order it
appended them, not in the order of their MFMA slots).  A real defect, fixed in the generator - but NOT the cause of the rare wrong 16 x 32 patch of eager
training steps that was being hunted then (the rate was unchanged with the fix: profiles/r05_eager_nondeterminism.txt; that one is handled by
gru_bwd_rs_kernel claiming the whole register file, gru_persist.hip).

usage: check_kloops.py kloop2_asm.h [...]     exit status 1 on a finding
"""
import re
import sys


def statements(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'FN_DEVINL void (\w+)\(', s):
        j = s.find('FN_DEVINL', m.end())
        body = s[m.end(): j if j > 0 else len(s)]
        lines = [l.strip() for l in re.findall(r'"([^"\\]*)\\n\\t"', body)]
        if lines:
            out[m.group(1)] = lines
    return out


def regs(tok):
    """'a[8:11]' -> {('a', 8) ..}; 'v200' -> {('v', 200)}; operands like %[gt0] are opaque names"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r'([av])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([av])(\d+)', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r'%\[(\w+)\]', tok)
    if m:
        return {("op", m.group(1))}
    return set()


VM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store")


def check_arrival(name, lines):
    bad = []
    for bi, l in enumerate(lines):
        if not l.startswith("s_barrier"):
            continue
        wi = max(i for i in range(bi) if lines[i].startswith("s_waitcnt vmcnt"))
        n = int(re.search(r'vmcnt\((\d+)\)', lines[wi]).group(1))
        slab = [i for i in range(wi) if lines[i].startswith("global_store") and " sc1" in lines[i]]
        if not slab:
            continue                                       # *_first: everything older than the statement; checked by the operand walk's start state
        younger = sum(1 for i in range(slab[-1] + 1, wi) if lines[i].startswith(VM))
        if n > younger:
            bad.append("%s: arrival waits vmcnt(%d) with %d operations issued behind the last slab store" % (name, n, younger))
    return bad


def walk(name, lines, start_fifo):
    """in-order retirement model: fifo = outstanding operations (oldest first), each (destination registers or None)"""
    bad = []
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}

    def run(pc, fifo, depth):
        fifo = list(fifo)
        while pc < len(lines):
            l = lines[pc]
            pc += 1
            if l.endswith(":"):
                continue
            op = l.split()[0]
            args = [a.strip() for a in l[len(op):].split(",")]
            if op == "s_waitcnt":
                m = re.search(r'vmcnt\((\d+)\)', l)
                if m:
                    n = int(m.group(1))
                    if len(fifo) > n:
                        fifo = fifo[len(fifo) - n:] if n else []
                continue
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = args[-1].split()[-1]
                if "noarr" in tgt:                         # the wave that does not post the arrival: fewer operations outstanding = the stricter case
                    continue
                if op == "s_branch":
                    pc = labels[tgt]
                    continue
                if depth < 4:
                    run(labels[tgt], fifo, depth + 1)      # taken
                continue                                   # and not taken
            if op.startswith(VM):
                if op.startswith("global_load") or op.startswith("buffer_load"):
                    fifo.append(regs(args[0]))
                else:
                    src = set()
                    for a in args[:2]:
                        src |= regs(a.split()[0]) if a else set()
                    for d in fifo:
                        if d and d & src:
                            bad.append("%s: '%s' reads %s while a load into it may be outstanding" % (name, l, sorted(d & src)[:2]))
                    fifo.append(None)
                continue
            if op.startswith("v_mfma") or op.startswith("ds_write") or op.startswith("v_accvgpr_read"):
                src = set()
                for a in (args[1:] if op.startswith("v_mfma") or op.startswith("v_accvgpr") else args):
                    src |= regs(a.split()[0]) if a else set()
                for d in fifo:
                    if d and d & src:
                        bad.append("%s: '%s' reads %s while a load into it may be outstanding" % (name, l, sorted(d & src)[:2]))
                        break
        return

    run(0, start_fifo, 0)
    return sorted(set(bad))


def main(paths):
    findings = []
    for p in paths:
        st = statements(p)
        for name, lines in st.items():
            findings += check_arrival(name, lines)
            fam = re.sub(r'_(main|first)$', '', name)
            if name.endswith(("_main", "_first")) and fam + "_pro" in st:
                start = [regs(l.split()[1]) for l in st[fam + "_pro"] if l.startswith("global_load")]
                if name.endswith("_main") and fam + "_out" in st:        # kloop4: the dgx / dghn stores of the last epilogue leave behind the ring request
                    start += [None for l in st[fam + "_out"] if l.startswith("global_store")]
                findings += walk(name, lines, start)
    for f in findings:
        print(f)
    print("%d statements of %d files checked, %d findings" % (sum(len(statements(p)) for p in paths), len(paths), len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
