"""Numerical check behind DESIGN.md section 11, item 1 (not part of the product): a 256-deep fp32 dot product evaluated as split-bf16
products accumulated in fp32 (blocks of 16 contraction indices, as v_mfma_f32_32x32x16_bf16 would) against the k-ordered fp32 fma
chain the kernels run today, both measured against float64.  Post-ReLU N(0, 1) activations, N(0, 0.06) weights, K = 256.

    fp32 chain   max err / sum|ab| 2.5e-07   mean 2.0e-08
    bf16 x 3     3.3e-06 / 4.9e-07      (hi*hi, hi*mid, mid*hi: too coarse for north_star's 1e-5 on losses and bit-exact indices)
    bf16 x 6     1.4e-07 / 6.3e-09      (+ hi*lo, mid*mid, lo*hi: fp32 class)
    bf16 x 9     1.4e-07 / 6.3e-09      (the three dropped terms change nothing)
"""
import torch as th

th.manual_seed(0)
M, K, N = 512, 256, 256
A = th.relu(th.randn(M, K))
B = th.randn(K, N) * 0.06
ref = A.double() @ B.double()
scale = A.abs().double() @ B.abs().double()


def chain_fp32(a, b):
    acc = th.zeros(a.shape[0], b.shape[1])
    for k in range(a.shape[1]):
        acc = th.addcmul(acc, a[:, k:k + 1], b[k:k + 1, :])
    return acc


def split3(x):
    hi = x.bfloat16().float()
    mid = (x - hi).bfloat16().float()
    lo = (x - hi - mid).bfloat16().float()
    return hi, mid, lo


def mm_blocks(a, b):            # products of bf16 values are exact in fp32; one fp32 rounding per 16-deep block
    acc = th.zeros(a.shape[0], b.shape[1])
    for k0 in range(0, a.shape[1], 16):
        acc = acc + (a[:, k0:k0 + 16].double() @ b[k0:k0 + 16, :].double()).float()
    return acc


ah, am, al = split3(A)
bh, bm, bl = split3(B)
runs = {"fp32 chain": chain_fp32(A, B),
        "bf16 x 3": sum(mm_blocks(x, y) for x, y in ((am, bh), (ah, bm), (ah, bh))),
        "bf16 x 6": sum(mm_blocks(x, y) for x, y in ((al, bh), (am, bm), (ah, bl), (am, bh), (ah, bm), (ah, bh))),
        "bf16 x 9": sum(mm_blocks(x, y) for x, y in ((al, bl), (al, bm), (am, bl), (al, bh), (am, bm), (ah, bl), (am, bh), (ah, bm), (ah, bh)))}
for name, c in runs.items():
    e = (c.double() - ref).abs() / scale
    print(f"{name:12s} max err / sum|ab| {float(e.max()):.2e}   mean {float(e.mean()):.2e}")
