"""derived per-dispatch figures from the four per-group files of scratch/pmc_step.sh (gpurun_out/pmc_step/*.txt):
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 halves FETCH_SIZE for wide coalesced reads, MI355X_MICROARCH.md),
MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), wait shares of the wave cycles."""
import sys, os, re, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_step"
vals = collections.defaultdict(dict)
for fn in sorted(os.listdir(d)):
    if not fn.endswith(".txt"):
        continue
    k = None
    for line in open(os.path.join(d, fn)):
        if line.startswith("== "):
            k = line[3:].strip()
        else:
            m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([\d.]+)\s+\(n=(\d+)\)", line)
            if m and k:
                vals[k][m.group(1)] = float(m.group(2)); vals[k]["n"] = int(m.group(3))
for k, v in vals.items():
    if "FETCH_SIZE" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    f, w = v["FETCH_SIZE"] / 1e6, v.get("WRITE_SIZE", 0.0) / 1e6
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    print("  %-58s n=%-3d HBM = 2 x %.3f + %.3f = %.3f GB/dispatch;  MFMA busy %.3g / %.3g cycles = %.1f %%;  waves on s_waitcnt %.1f %%, issue-stalled %.1f %% of wave cycles"
          % (k[:58], v["n"], f, w, 2 * f + w, busy, cyc, 100 * busy / cyc if cyc else 0, 100 * v.get("SQ_WAIT_ANY", 0) / wc if wc else 0,
             100 * v.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0))
