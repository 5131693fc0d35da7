"""Encoder scans (4 scans x B=256, H=512): one-wavefront-per-SIMD kernels (variant 0x400) vs the wave-per-row-tile kernels, stagger sweep,
bit equality of all outputs."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, H, V = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 512, 342
torch.manual_seed(0)
fw, bw = [], []
for s in range(4):
    w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
    d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, b_ih=torch.randn(3*H, device=dev) * 0.1,
             gx_table=torch.randn(V, 3*H, device=dev) * 0.3, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
             h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
    fw.append(d)
    bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=None, h_all=d["h_all"], gates=d["gates"], dh_last=torch.randn(B, H, device=dev) * 0.1,
                   dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev),
                   dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev)))
def t(fn, reps=4):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
ref = {}
for name, v in [("one wavefront per SIMD <4,1,2>", 0x400), ("wave, s_nop 6", 0), ("wave, s_nop 7", 0x1000), ("wave, s_nop 5", 0x2000), ("wave, no pad", 0x3000), ("wave, s_nop 6, interleaved", 0x4000), ("wave, s_nop 7, interleaved", 0x5000), ("wave, s_nop 5, interleaved", 0x6000), ("wave, s_nop 6 again", 0), ("wave, s_nop 6, stagger 0", 255 << 16)]:
    tf = t(lambda: ops.gru_seq_fwd(fw, variant=v))
    outs = [x.clone() for d in fw for x in (d["h_all"], d["gates"])]
    for b in bw:
        b["dgx_rowsum"].zero_(); b["dghn_rowsum"].zero_()
    ops.gru_seq_bwd(bw, variant=0x400); torch.cuda.synchronize()
    outs += [x.clone() for d in bw for x in (d["dgx_all"], d["dghn_all"], d["dgx_rowsum"], d["dghn_rowsum"])]
    tb = t(lambda: ops.gru_seq_bwd(bw, variant=0x400))
    if not ref: ref["o"] = outs
    same = all(torch.equal(a, b) for a, b in zip(ref["o"], outs))
    print("%-36s fwd %.3f ms (%.2f us/step, %.1f TF/s)  bwd %.3f ms (%.2f us/step)  outputs bit-equal to the first: %s  err %d"
          % (name, tf, tf * 1e3 / T, 4 * T * B * 2.0 * H * 3 * H / tf / 1e9, tb, tb * 1e3 / T, same, ops.gru_sync_error(False)), flush=True)
