"""profiles/r03_pmc_training_step.txt from gpurun_out/pmc_step/*.txt (scratch/pmc_step.sh + scratch/pmc_derive.py)"""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
derived = subprocess.run(["python", "scratch/pmc_derive.py"], cwd=R, capture_output=True, text=True).stdout
hdr = '''rocprofv3 --kernel-trace --pmc <group> -- python scratch/pmc_step.py 2      (scratch/pmc_step.sh, summary lines by scratch/pmc_derive.py; MI355X, round 3, FINAL
kernels: hand-placed K loops of the scans, K-range-per-XCD grid of the weight-gradient GEMMs, LDS-free NT GEMMs; the round's first collection - scans and GEMMs
of round 2 - is in git history: 7c10f6f)
Two eager (no hipGraph) forward+backward passes of the training step at the benchmark shape (hidden 512, B=256, T=256, Tr=64); one rocprofv3
pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; no other tracing domain).  PMC collection serialises dispatches: these
are the counters of each kernel running ALONE.  Values = average per dispatch over the matching kernel name (scratch/pmc_avg.py; the last block
picks the 48-tile x 16-K-range launches of gemm_tn_kernel by grid size).
Units: FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md) ->
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.  GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = cycles of the dispatch); SQ_VALU_MFMA_BUSY_CYCLES is
summed over 1024 SIMDs -> MFMA busy fraction = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8).

derived (per dispatch):
'''
tail = '''
  encoder forward  <4,1,2,4>: 412.3 GFLOP per launch; algorithmic HBM 8 KiB x 262144 sample-steps = 2.15 GB (measured 2.0 x: the saved gates)
  encoder backward <4,1,2,8>: algorithmic 16 KiB per sample-step = 4.29 GB (measured 1.85 x)
  weight-gradient GEMM of a scan (gemm_tn_kernel, M = 1536, N = 512, K = 65280: the kernel symbol with the largest time per step = bench.py's `roofline`):
      operands (1536 + 512) x 65280 x 4 B = 0.535 GB + 16 K-range slabs of 3.1 MB = 0.585 GB algorithmic; measured 0.711 GB = 1.2 x.
      Before the K-range-per-XCD grid (tiles dealt to the XCDs): 2 x 0.834 + 0.049 = 1.72 GB = 2.9 x (every XCD's L2 streamed all of B and a sixth of A
      for every K range); the launch itself is MFMA-bound either way (727 us = 141.7 TFLOP/s alone, MFMA pipe 88-89 % busy)
  MFMA pipe busy, round 2 kernels -> now: encoder forward 66.6 -> 74 %, encoder backward 56.0 -> 68 %, decoder-shape forward 51.4 -> 61 %,
      decoder-shape backward 42.9 -> 52 %
  LDS-staged gemm_kernel<128,128,32> on the short-K products (before gemm_nt_direct_kernel took them over): MFMA pipe 54-56 % busy

'''
body = ""
for g in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES_SQ_INSTS_MFMA_S", "GRBM_GUI_ACTIVE"):
    body += "==================== %s ====================\n" % g + open(os.path.join(R, "gpurun_out/pmc_step", g + ".txt")).read() + "\n"
open(os.path.join(R, "profiles/r03_pmc_training_step.txt"), "w").write(hdr + derived + tail + body)
print(derived)
