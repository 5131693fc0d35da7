"""Arithmetic of the deep MFMA products of the path (weight-gradient GEMMs, recurrent products of the scans).

The reference computes in fp32 (torch CPU kernels).  Both arithmetics below are fp32-class: operands keep all 24 significand bits and
sums are accumulated in fp32.

  "f32"     v_mfma_f32_16x16x4_f32 chains (157 TFLOP/s dense peak on MI355X)
  "bf16x6"  every fp32 operand value is cut EXACTLY into three bf16 pieces (hi + mid + lo == x), six of the nine partial products
            (all but lo*lo, lo*mid, mid*lo: <= 2^-24 |a b|) are accumulated smallest first in the fp32 accumulator of
            v_mfma_f32_16x16x32_bf16 - measured against float64 as accurate as the fp32 chain (tests/test_gpu_parity.py::
            test_bf16x6_adversarial_operands_vs_float64, scratch/mfma_bf16x9.hip), 2.4 x its rate for pre-split operands.

A model carries its choice (``model.set_arith``); None = the package default below.  Kernels without a bf16 x 6 form, and shapes the
bf16 x 6 kernels do not take, run on the fp32 MFMA whatever the choice.
"""
F32, BF16X6 = "f32", "bf16x6"
NAMES = (F32, BF16X6)
_default = BF16X6      # since round 5 (the conditions of the round-4 review hold: DESIGN.md section 3 / 11); "f32" stays one call away


def check(name):
    if name not in NAMES:
        raise ValueError("arith must be one of %s, got %r" % (NAMES, name))
    return name


def default():
    return _default


def set_default(name):
    """package-wide default for models that have not chosen (returns the previous one)"""
    global _default
    prev, _default = _default, check(name)
    return prev


def resolve(name):
    return _default if name is None else check(name)


def describe(name):
    return {F32: "fp32 MFMA (v_mfma_f32_16x16x4_f32)",
            BF16X6: "bf16x3 exact split, 6 products, f32 accumulate (v_mfma_f32_16x16x32_bf16)"}[resolve(name)]
