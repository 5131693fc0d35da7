#!/usr/bin/env python3
"""Build-time check of the ping-pong scan kernels' ISA (round 4): between the hand-written asm statements the compiler must issue
NO vector-memory instruction of its own inside the time loop, must not wait on vmcnt there and must never touch an AGPR - the asm
statements keep loads in flight across statement boundaries and count every outstanding operation.
usage: check_pp_isa.py file.s kernel-substring   (file.s = hipcc -S --cuda-device-only of gru_persist.hip)"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") or (key in l and "; @" in l and l.startswith("_ZN"))]
for s0 in start:
    name = lines[s0].split(":")[0]
    end = next(i for i in range(s0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    inasm, out, nasm = False, [], 0
    for i in range(s0, end):
        l = lines[i].strip()
        if l.startswith(";;#ASMSTART"):
            inasm = True; nasm += 1; continue
        if l.startswith(";;#ASMEND"):
            inasm = False; continue
        if inasm or not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        if re.match(r"(global_|buffer_|flat_|scratch_)", op) or "vmcnt" in l or "accvgpr" in op or re.search(r"\ba\[?\d", l.split(";")[0]):
            out.append((i + 1, nasm, l.split(";")[0].strip()))
    print("== %s: %d asm statements, %d compiler-issued memory / vmcnt / AGPR instructions outside them" % (name[:60], nasm, len(out)))
    for o in out:
        print("   line %d (after asm #%d): %s" % o)
