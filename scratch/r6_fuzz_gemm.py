"""round 6: randomised shapes through the three bf16 x 6 weight-gradient kernels (per-wave / 128 x 128 / 128 x 256 tiles) and the Linear-forward kernel: ragged M, N, K tails,
every split depth, column-offset views with padded leading dimensions, the two-source form; bit-identity where the K ranges are whole 32-k blocks, float64 error bound everywhere.
usage: python scratch/r6_fuzz_gemm.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ops.dw_x6 = True
bad = 0
for case in range(N_CASES):
    M = int(rng.choice([rng.randint(1, 1700), 128 * rng.randint(1, 13), 1536, 342]))
    N = int(rng.choice([rng.randint(1, 700), 128 * rng.randint(1, 5), 512]))
    K = int(rng.choice([rng.randint(1024, 9000), 32 * rng.randint(32, 300), 1024 + 32 * rng.randint(0, 6)]))
    sk = int(rng.choice([1, 2, 3, 5, 8, 16, 24, 32, 40]))
    lda, ldb = (M + 3) // 4 * 4 + 4 * rng.randint(0, 3), (N + 3) // 4 * 4 + 4 * rng.randint(0, 3)
    Af = torch.randn(K, lda, device=dev) * float(rng.choice([1.0, 1e-3, 30.0])); Bf = torch.randn(K, ldb, device=dev) * 0.3
    A, B = Af[:, :M], Bf[:, :N]
    ref = A.double().t() @ B.double()
    scale = float((A.double().abs().t() @ B.double().abs()).max()) + 1e-300
    out = {}
    for name, wide, pw in (("perwave", False, True), ("tile128", False, False), ("tile256", "force", False)):
        ops.x6_wide, ops.x6_perwave = wide, pw
        C = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(A, B, C, a_k=False, b_k=False, splitk=sk)
        out[name] = C
    klen = K if sk <= 1 else ((K + sk - 1) // sk + 31) // 32 * 32
    whole = all(min(K, k0 + klen) % 32 == 0 for k0 in range(0, K, klen))
    errs = {n: float((c.double() - ref).abs().max()) / scale for n, c in out.items()}
    ok = all(e < 2e-6 for e in errs.values()) and torch.equal(out["tile128"], out["tile256"]) and (not whole or torch.equal(out["perwave"], out["tile128"]))
    if not ok:
        bad += 1
        print("TN MISMATCH M %d N %d K %d splitk %d lda %d ldb %d whole %s errs %s eq128/256 %s eq pw %s" % (M, N, K, sk, lda, ldb, whole, errs, torch.equal(out["tile128"], out["tile256"]),
                                                                                                      torch.equal(out["perwave"], out["tile128"])), flush=True)
    if M >= 512 and M % 3 == 0 and N % 4 == 0 and (M // 3) == N and ldb == N:      # the two-source [dgx | dghn] form wants M = 3 N, dense B
        pass
ops.x6_wide, ops.x6_perwave = True, False
print("TN: %d cases, %d mismatches" % (N_CASES, bad), flush=True)
# the two-source form of fn_gru_dwhh_f32 at H = 512 / 256
bad2 = 0
for case in range(N_CASES // 4):
    H = int(rng.choice([512, 256, 128]))
    rows = int(rng.choice([rng.randint(1024, 9000), 32 * rng.randint(32, 300)]))
    sk = int(rng.choice([1, 4, 8, 16, 24]))
    dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
    want = torch.cat([dgx[:, :2 * H], dghn], 1).double().t() @ hp.double()
    sc = float((torch.cat([dgx[:, :2 * H], dghn], 1).double().abs().t() @ hp.double().abs()).max())
    res = {}
    for name, wide, pw in (("perwave", False, True), ("tile128", False, False), ("tile256", "force", False)):
        ops.x6_wide, ops.x6_perwave = wide, pw
        dW = torch.full((3 * H, H), float("nan"), device=dev)
        ops.gru_dwhh(dgx, dghn, hp, dW, splitk=sk)
        res[name] = dW
    if not (all(float((c.double() - want).abs().max()) / sc < 2e-6 for c in res.values()) and torch.equal(res["tile128"], res["tile256"])):
        bad2 += 1
        print("DWHH MISMATCH H %d rows %d splitk %d" % (H, rows, sk), {n: float((c.double() - want).abs().max()) / sc for n, c in res.items()}, flush=True)
ops.x6_wide, ops.x6_perwave = True, False
print("dW_hh two-source form: %d cases, %d mismatches" % (N_CASES // 4, bad2), flush=True)
bad3 = 0
for case in range(N_CASES // 2):
    M, N, K = 128 * int(rng.randint(8, 40)), 128 * int(rng.randint(1, 13)), 32 * int(rng.randint(4, 50))
    if (M // 128) * (N // 128) < 128:
        M = 128 * (128 // (N // 128) + 1)
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.2
    bias = torch.randn(N, device=dev) if rng.randint(2) else None
    alpha, beta = float(rng.choice([1.0, 0.5])), float(rng.choice([0.0, 2.0]))
    C0 = torch.randn(M, N, device=dev)
    want = alpha * (A.double() @ W.double().t()) + beta * C0.double() + (bias.double() if bias is not None else 0)
    scale = float((A.double().abs() @ W.double().abs().t()).max()) + 2 * float(C0.abs().max())
    ops.nt_x6 = True
    C = C0.clone(); ops.gemm(A, W, C, a_k=True, b_k=True, alpha=alpha, beta=beta, bias=bias)
    ops.nt_x6 = False
    Cf = C0.clone(); ops.gemm(A, W, Cf, a_k=True, b_k=True, alpha=alpha, beta=beta, bias=bias)
    e6, e32 = float((C.double() - want).abs().max()) / scale, float((Cf.double() - want).abs().max()) / scale
    if not (e6 < 2e-6 and e6 <= 2 * e32 + 1e-9):
        bad3 += 1
        print("NT MISMATCH M %d N %d K %d alpha %g beta %g bias %s: x6 %.3e fp32 %.3e" % (M, N, K, alpha, beta, bias is not None, e6, e32), flush=True)
ops.nt_x6 = True
print("NT: %d cases, %d mismatches" % (N_CASES // 2, bad3), flush=True)
sys.exit(1 if bad + bad2 + bad3 else 0)
