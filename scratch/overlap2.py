"""Weight-gradient GEMM beside the ENCODER scans (one 336/374-register wavefront per SIMD): normal (194 regs) vs lean (128 regs) instance."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
n, Bn, T = 4, 256, 128
fw, bw = [], []
for s in range(n):
    w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
    d = dict(B=Bn, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
             gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (Bn, T), dtype=torch.int32, device=dev),
             h_all=torch.zeros(T, Bn, H, device=dev), gates=torch.zeros(T, ops.gates_floats(Bn, H), device=dev))
    fw.append(d)
    bw.append(dict(B=Bn, T=T, H=H, w_hh_t_frag=wtf, h0=None, h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, Bn, H, device=dev) * 0.01,
                   dgx_all=torch.zeros(T, Bn, 3*H, device=dev), dghn_all=torch.zeros(T, Bn, H, device=dev), scratch=torch.zeros(Bn, H, device=dev),
                   dgx_rowsum=torch.zeros(Bn, 3*H, device=dev), dghn_rowsum=torch.zeros(Bn, H, device=dev)))
ops.gru_seq_fwd(fw)
side = torch.cuda.Stream()
K = 65536
A = torch.randn(K, 1536, device=dev) * 0.01; Bm = torch.randn(K, 512, device=dev) * 0.01; Cm = torch.zeros(1536, 512, device=dev)
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 2
DBG = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def gemm(lean, sk):
    ops.lane = "side/"
    for _ in range(NG):
        ops.gemm(A, Bm, Cm, a_k=False, b_k=False, splitk=sk | (DBG << 17), lean=lean)
    ops.lane = ""
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for sname, scan in (("enc fwd", lambda: ops.gru_seq_fwd(fw)), ("enc bwd", lambda: ops.gru_seq_bwd(bw))):
    ts = t(scan)
    for lean in (True,):
        for sk in (16,):
            tg = t(lambda: gemm(lean, sk))
            def both(order):
                side.wait_stream(torch.cuda.current_stream())
                if order == 0: scan()
                with torch.cuda.stream(side):
                    if order == 0: torch.cuda._sleep(100000)
                    gemm(lean, sk)
                if order == 1: scan()
                torch.cuda.current_stream().wait_stream(side)
            tb0, tb1 = t(lambda: both(0)), t(lambda: both(1))
            ea, eb, ec, ed = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            ea.record(); scan(); ed.record()
            with torch.cuda.stream(side):
                torch.cuda._sleep(100000)
                eb.record(side); gemm(lean, sk); ec.record(side)
            torch.cuda.synchronize()
            print("%s %.3f ms | %d x dW GEMM %s splitk %2d alone %.3f ms (%.0f TF/s) | both: scan first %.3f, gemm first %.3f (serial %.3f) | scan [0, %.3f] gemm [%.3f, %.3f]"
                  % (sname, ts, NG, "lean" if lean else "full", sk, tg, NG * 2.0 * 1536 * 512 * K / tg / 1e9, tb0, tb1, ts + tg,
                     ea.elapsed_time(ed), ea.elapsed_time(eb), ea.elapsed_time(ec)), flush=True)
