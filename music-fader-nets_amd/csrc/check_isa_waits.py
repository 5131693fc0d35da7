#!/usr/bin/env python3
"""Static check of the COMPILED kernels (hipcc --cuda-device-only -S): every vector-memory load whose completion the source counts by hand - the
asm-statement loads of mma_core.h (fn_gld4_asm ...) and of the generated K loops, invisible to the compiler's own s_waitcnt bookkeeping - against
the `s_waitcnt vmcnt(n)` that are really in the instruction stream.

Model (the one the hand-written waits assume, measured in round 5: scratch/vmcnt_order.hip): a wave's vector-memory operations - loads, stores,
atomics, whoever issued them - retire IN ORDER on vmcnt; `s_waitcnt vmcnt(n)` returns when at most n of them are outstanding.  A load's destination
registers hold garbage until a wait has covered it.  Finding = an instruction that reads or writes a register that a load may still be writing.

The check walks the control-flow graph of every function that contains an asm-issued load: state = the ordered list of outstanding operations with
their destination registers; a block is visited once per distinct state that reaches it (loops included), so software-pipelined rings - loads
requested in one trip and waited for in a later one - are followed across the back-edge.  It covers what check_kloops.py cannot: the hand-written
fill / ring code of gru.hip, gemm.hip, decode_persist.hip (the decode race of round 5 lived there) and the C++ around the generated statements.

usage: check_isa_waits.py file.s [...]     exit status 1 on a finding
"""
import re
import sys

VM_LOAD = ("global_load", "buffer_load", "scratch_load", "flat_load")
VM_OTHER = ("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic", "flat_atomic", "buffer_wbl2", "buffer_inv",
            "global_wb", "global_inv")
REG = re.compile(r'\b([av])(?:(\d+)|\[(\d+):(\d+)\])')
MAX_STATES = 4000


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("op", "args", "asm", "line", "touch", "dest", "kind", "wait", "target")

    def __init__(self, op, args, asm, line):
        self.op, self.args, self.asm, self.line = op, args, asm, line
        self.kind, self.wait, self.target, self.dest = None, None, None, frozenset()
        if op.startswith(VM_LOAD):
            self.kind = "load"
            first = args.split(",")[0]
            self.dest = frozenset() if " lds" in (" " + args) else frozenset(regs_of(first))      # LDS-DMA loads write no register
            self.touch = regs_of(args)             # (address registers are read at issue: a pending load of theirs would be a finding)
        elif op.startswith(VM_OTHER):
            self.kind = "vm"
            rtn = ("sc0" in args.split() or "glc" in args.split()) and "atomic" in op
            self.dest = frozenset(regs_of(args.split(",")[0])) if rtn else frozenset()
            self.touch = regs_of(args)
        else:
            self.touch = regs_of(args)
        if op == "s_waitcnt":
            m = re.search(r'vmcnt\((\d+)\)', args)
            if m:
                self.wait = int(m.group(1))
            elif re.fullmatch(r'\s*(0x[0-9a-fA-F]+|\d+)\s*', args):      # raw immediate: gfx9 vmcnt = bits [3:0] | [15:14] << 4
                v = int(args.strip(), 0)
                self.wait = (v & 0xf) | (((v >> 14) & 3) << 4)
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            self.target = args.strip()


def parse(path):
    """-> {function: [Ins or label str]}"""
    funcs, cur, name, in_asm = {}, None, None, False
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s:
            continue
        m = re.fullmatch(r'([A-Za-z_.$][\w.$]*):', s)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                name, cur = lab, []
                funcs[name] = cur
            elif cur is not None:
                cur.append(lab)
            continue
        if s.startswith(".") or cur is None:
            if s.startswith(".Lfunc_end") or s.startswith(".size"):
                pass
            continue
        parts = s.split(None, 1)
        cur.append(Ins(parts[0], parts[1] if len(parts) > 1 else "", in_asm, ln))
    return funcs


def blocks_of(items):
    """basic blocks: list of (label or None, [Ins]); successors by index"""
    blocks, cur, lab = [], [], None
    for it in items:
        if isinstance(it, str):
            if cur or lab is not None:
                blocks.append((lab, cur))
            cur, lab = [], it
            continue
        cur.append(it)
        if it.target is not None or it.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            blocks.append((lab, cur))
            cur, lab = [], None
    if cur or lab is not None:
        blocks.append((lab, cur))
    index = {b[0]: i for i, b in enumerate(blocks) if b[0] is not None}
    succ = []
    for i, (lab, ins) in enumerate(blocks):
        s = []
        last = ins[-1] if ins else None
        if last is not None and last.target is not None:
            if last.target in index:
                s.append(index[last.target])
            if last.op != "s_branch" and i + 1 < len(blocks):
                s.append(i + 1)
        elif last is not None and last.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    return blocks, succ


def check_function(name, items, path):
    if not any(isinstance(i, Ins) and i.asm and i.kind == "load" for i in items):
        return 0, 0
    blocks, succ = blocks_of(items)
    seen = [set() for _ in blocks]
    work = [(0, ())]
    seen[0].add(())
    findings, reported, total = 0, set(), 0
    while work:
        b, state = work.pop()
        pend = list(state)                         # ordered: oldest first; entries = (dest regs, is_asm, line)
        for ins in blocks[b][1]:
            if ins.wait is not None and len(pend) > ins.wait:
                pend = pend[len(pend) - ins.wait:] if ins.wait > 0 else []
            if ins.touch:
                # a LOAD that overwrites the destination of an older pending load is ordered behind it (in-order return): only its ADDRESS registers count
                touch = ins.touch - ins.dest if ins.kind == "load" else ins.touch
                for dest, is_asm, ln in pend:
                    if dest and not dest.isdisjoint(touch) and (ins.line, ln) not in reported:
                        reported.add((ins.line, ln))
                        findings += 1
                        who = "asm-issued" if is_asm else "compiler-issued"
                        print("%s:%d: %s: `%s %s` touches %s while the %s load of line %d may still be in flight (%d operations outstanding)" % (
                            path, ins.line, name[:60], ins.op, ins.args[:60], sorted(dest & touch)[:4], who, ln, len(pend)))
            if ins.kind == "load":
                pend.append((ins.dest, ins.asm, ins.line))
            elif ins.kind == "vm":
                pend.append((ins.dest, ins.asm, ins.line))
            if len(pend) > 63:                     # the counter saturates at 63: nothing beyond can be told apart (never the case in these kernels)
                pend = pend[-63:]
        out = tuple(pend)
        key = tuple((d, a) for d, a, _ in pend)
        for s in succ[b]:
            # a state that is a SUFFIX of one already seen at this block is covered by it (the longer list has the same youngest operations plus older
            # ones: every wait leaves it a superset)
            if any(len(k) >= len(key) and k[len(k) - len(key):] == key for k in seen[s]):
                continue
            if len(seen[s]) >= MAX_STATES:
                print("%s: %s: more than %d distinct states reach one block - analysis gave up" % (path, name[:60], MAX_STATES))
                return findings + 1, total
            seen[s].add(key)
            work.append((s, out))
        total += 1
    return findings, total


# Kernel families the model cannot follow: the number of loads in flight depends on run-time conditions that the later waits repeat (prologue loads
# guarded by the trip count, a ring that is requested either inside a statement or by the fall-back behind it), and a path-insensitive walk pairs
# the request of one case with the wait of the other.  They are listed, not silently skipped; the generated statements among them are covered by
# check_kloops.py, which knows the protocol.
UNVERIFIABLE = ("gru_cell_wlds_kernel", "gru_fwd_step_kernel", "gru_fwd_persist_kernel", "gru_bwd_x6_kernel", "decode_greedy_kernel", "out_head_kernel")


def family(name):
    m = re.match(r'_Z(?:N12_GLOBAL__N_1)?(\d+)', name)
    return name[m.end(): m.end() + int(m.group(1))] if m else name


def _one(job):
    name, items, p = job
    return check_function(name, items, p)


def main(paths, jobs=None):
    import multiprocessing
    import os
    bad = nfun = 0
    skipped = {}
    todo = []
    for p in paths:
        for name, items in parse(p).items():
            if not any(isinstance(i, Ins) and i.asm and i.kind == "load" for i in items):
                continue
            fam = family(name)
            if fam in UNVERIFIABLE:
                skipped[fam] = skipped.get(fam, 0) + 1
                continue
            todo.append((name, items, p))
    jobs = jobs or min(8, os.cpu_count() or 1)
    if jobs > 1 and len(todo) > 8:
        with multiprocessing.Pool(jobs) as pool:
            results = pool.map(_one, todo, chunksize=1)
    else:
        results = [_one(j) for j in todo]
    for f, visited in results:
        bad += f
        nfun += 1 if visited else 0
    print("%d kernels with hand-counted loads in %d files checked, %d findings; not verifiable by this model (listed in UNVERIFIABLE): %s" % (
        nfun, len(paths), bad, ", ".join("%s x %d" % kv for kv in sorted(skipped.items())) or "none"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
