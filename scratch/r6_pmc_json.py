"""round 6 (r5 script with the round-6 kernel names): per-kernel averages of the rocprofv3 --pmc passes of scratch/r6_pmc_step.sh -> a readable table (stdout) and the JSON bench.py reads
(profiles/r06_pmc_traffic.json).  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 reports half of the bytes of wide coalesced reads,
MI355X_MICROARCH.md), MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs).
usage: r5_pmc_json.py <dir with <group>.csv> <out.json>"""
import csv, json, os, re, sys, collections
d, outp = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))      # (symbol, instance, grid) -> counter -> [sum, n]
for fn in sorted(os.listdir(d)):
    if not fn.endswith(".csv"):
        continue
    for r in csv.DictReader(open(os.path.join(d, fn))):
        name = r.get("Kernel_Name", "")
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
        if not m:
            continue
        key = (m.group(1), m.group(2) or "", r.get("Grid_Size", ""))
        a = rows[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1


def derived(v):
    g = lambda k: (v[k][0] / v[k][1]) if k in v and v[k][1] else None
    f, w, cyc, busy, wc = g("FETCH_SIZE"), g("WRITE_SIZE"), g("GRBM_GUI_ACTIVE"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_WAVE_CYCLES")
    out = dict(n=max(x[1] for x in v.values()))
    if f is not None and w is not None:
        out["hbm_bytes"] = (2 * f + w) * 1e3
        out["fetch_kb"], out["write_kb"] = f, w
    if cyc and busy is not None:
        out["mfma_busy"] = (busy / 1024) / (cyc / 8)
    if wc:
        for k, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "issue_stalled")):
            if g(k) is not None:
                out[nm] = g(k) / wc
    return out


def merge(keys):
    v = collections.defaultdict(lambda: [0.0, 0])
    for k in keys:
        for c, (s, n) in rows[k].items():
            v[c][0] += s; v[c][1] += n
    return v


heavy = ("gru_fwd_x6pp_kernel", "gru_fwd_pp_kernel", "gru_bwd_x6_kernel", "gru_bwd_rs_kernel", "gemm_tn_x6v_kernel", "gemm_tn_x6w_kernel", "gemm_nt_x6w_kernel", "gemm_tn_x6_kernel", "gemm_tn_kernel", "gemm_nt_direct_kernel", "gemm_kernel",
         "out_head_kernel", "eg_piece_kernel", "eg_final_kernel")
by_symbol, table = {}, []
for sym in heavy:
    keys = [k for k in rows if k[0] == sym]
    if not keys:
        continue
    dv = derived(merge(keys))
    if "hbm_bytes" in dv:
        by_symbol[sym] = dv["hbm_bytes"]
    table.append(("%s (all %d launch shapes merged)" % (sym, len(keys)), dv))
    for k in sorted(keys, key=lambda k: -max(x[1] for x in rows[k].values())):
        table.append(("    %s%s grid %s" % k, derived(rows[k])))
for name, dv in table:
    print("%-72s n=%-4d %s%s%s" % (name[:72], dv["n"], ("HBM = 2 x %.0f + %.0f KB = %.3f GB/launch; " % (dv["fetch_kb"], dv["write_kb"], dv["hbm_bytes"] / 1e9)) if "hbm_bytes" in dv else "",
                                  ("MFMA busy %.1f %%; " % (100 * dv["mfma_busy"])) if "mfma_busy" in dv else "",
                                  ("waves on s_waitcnt %.1f %%, issue-stalled %.1f %%" % (100 * dv.get("wait_any", 0), 100 * dv.get("issue_stalled", 0))) if "wait_any" in dv else ""))


def pick(sym, inst):      # traffic of one launch shape (template instance with the most launches of that instance)
    keys = [k for k in rows if k[0] == sym and (inst is None or k[1].startswith(inst))]
    if not keys:
        return None
    k = max(keys, key=lambda k: derived(rows[k]).get("hbm_bytes", 0))
    return derived(rows[k]).get("hbm_bytes")


by_row = {}
for row, cands in (("enc_fwd_scan", (("gru_fwd_x6pp_kernel", "<1"), ("gru_fwd_pp_kernel", "<1"))), ("dec_fwd_scan_chunk", (("gru_fwd_x6pp_kernel", "<2"), ("gru_fwd_pp_kernel", "<2"))),
                   ("enc_bwd_scan", (("gru_bwd_x6_kernel", "<2"), ("gru_bwd_rs_kernel", "<2"))), ("dec_bwd_scan_chunk", (("gru_bwd_x6_kernel", "<1"), ("gru_bwd_rs_kernel", "<1"))),
                   ("dwhh_gemm_tn", (("gemm_tn_x6v_kernel", None), ("gemm_tn_x6_kernel", None), ("gemm_tn_kernel", None))), ("dwhh_gemm_tn_attr", (("gemm_tn_x6w_kernel", None),)), ("gemm_nt", (("gemm_nt_x6w_kernel", None),)), ("out_head", (("out_head_kernel", None),))):
    for sym, inst in cands:
        v = pick(sym, inst)
        if v:
            by_row[row] = v
            break
json.dump(dict(how="rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE; bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction); per launch: mean over the launches of "
                   "two eager forward+backward passes (scratch/r6_pmc_step.sh)", by_symbol=by_symbol, by_row=by_row), open(outp, "w"), indent=1)
