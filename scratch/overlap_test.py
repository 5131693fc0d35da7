"""Does a weight-gradient GEMM make progress beside a weight-stationary scan (one persistent workgroup per CU)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
def mk(n, B, T):
    fw = []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        fw.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 h0=torch.randn(B, H, device=dev) * 0.1, gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
    return fw
side = torch.cuda.Stream()
K = 65536
A = torch.randn(K, 1536, device=dev) * 0.01; Bm = torch.randn(K, 512, device=dev) * 0.01; Cm = torch.zeros(1536, 512, device=dev)
def gemm():
    ops.lane = "side/"
    ops.gemm(A, Bm, Cm, a_k=False, b_k=False, splitk=16)
    ops.lane = ""
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for name, (n, B, T) in (("4 scans x B128 (MT=1 kernel, 64 rows/WG)", (4, 128, 128)), ("4 scans x B256 (MT=2 kernel, 128 rows/WG)", (4, 256, 128))):
    fw = mk(n, B, T)
    def scan(): ops.gru_seq_fwd(fw)
    def both():
        side.wait_stream(torch.cuda.current_stream())
        scan()
        with torch.cuda.stream(side):
            torch.cuda._sleep(200000)      # let the scan's workgroups take their CUs first
            gemm()
        torch.cuda.current_stream().wait_stream(side)
    ts, tg, tb = t(scan), t(gemm), t(both)
    # where does the GEMM run inside the concurrent case?
    ea, eb, ec, ed = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    ea.record(); scan(); ed.record()
    with torch.cuda.stream(side):
        torch.cuda._sleep(200000)
        eb.record(side); gemm(); ec.record(side)
    torch.cuda.synchronize()
    print("   scan [0, %.3f] ms ; gemm [%.3f, %.3f] ms" % (ea.elapsed_time(ed), ea.elapsed_time(eb), ea.elapsed_time(ec)))
    print("%s: scan %.3f ms, gemm %.3f ms, both concurrently %.3f ms (sum %.3f)" % (name, ts, tg, tb, ts + tg), flush=True)
