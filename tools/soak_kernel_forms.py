"""Soak check of kernel forms that must be bit-identical: N consecutive Envelope steps from fixed seeds, one digest over the final
parameters, Adam state and the last step's outputs -- run once per environment (legs given as KEY=VALUE[,KEY=VALUE] arguments; '-' is
the default environment) in fresh interpreters; the digests must agree.  A race in a hand-over (a weight stage read before it landed,
a buffer reused too early) shows up as a different digest within a few hundred steps.
    python tools/soak_kernel_forms.py --steps 2000 --shape 256,64 - MORL_BF_PW=0
    python tools/soak_kernel_forms.py --steps 2000 --shape 256,32 - MORL_BF_PW=0 MORL_BF_PW=5"""
import hashlib, os, subprocess, sys

SNIPPET = r"""
import hashlib, os, sys
import torch as th
ROOT = sys.argv[1]
sys.path[:0] = [ROOT]
import morl_baselines_amd.ops as ops
from morl_baselines_amd.native import load_library
steps, B, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib, dev = load_library(), th.device("cuda:0")
D, R, A, arch = 32, 3, 6, (256, 256, 256, 256)
g = th.Generator().manual_seed(7)
ctx = ops.QNetContext(D, R, A, arch, B, W, lib=lib)
P = ctx.n_params
po = (th.randn(P, generator=g) * 0.05).to(dev); pt = po.clone()
grads, m, v = th.zeros(P, device=dev), th.zeros(P, device=dev), th.zeros(P, device=dev)
NB = 8                                      # a few batches, cycled
obs = [th.randn(B, D, generator=g).to(dev) for _ in range(NB)]; nobs = [th.randn(B, D, generator=g).to(dev) for _ in range(NB)]
act = [th.randint(0, A, (B,), generator=g).to(th.int32).to(dev) for _ in range(NB)]
rew = [th.randn(B, R, generator=g).to(dev) for _ in range(NB)]; done = [(th.rand(B, generator=g) < 0.1).float().to(dev) for _ in range(NB)]
ws = []
for _ in range(NB):
    w = th.rand(W, R, generator=g); ws.append((w / w.sum(1, keepdim=True)).to(dev))
h = hashlib.sha256()
for t in range(steps):
    k = t % NB
    out = ops.envelope_update(ctx, po, pt, grads, m, v, obs[k], nobs[k], act[k], rew[k], done[k], ws[k], gamma=0.98, lr=3e-4,
                              adam_step=t + 1, max_grad_norm=1.0, homotopy_lambda=0.3)
    if t % 200 == 199:
        pt.copy_(po)
        h.update(out["loss"].cpu().numpy().tobytes()); h.update(po.cpu().numpy().tobytes())
th.cuda.synchronize()
for x in (po, m, v, grads, out["loss"], out["priority"]):
    h.update(x.cpu().numpy().tobytes())
print("SOAK_DIGEST", h.hexdigest(), "bits", ctx.last_step_bf16(), "loss", float(out["loss"]))
"""


def main():
    args = sys.argv[1:]
    steps, shape = 2000, "256,64"
    legs = []
    while args:
        a = args.pop(0)
        if a == "--steps": steps = int(args.pop(0))
        elif a == "--shape": shape = args.pop(0)
        else: legs.append(a)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B, W = shape.split(",")
    digests = {}
    for leg in legs or ["-"]:
        env = dict(os.environ)
        if leg != "-":
            env.update(dict(kv.split("=", 1) for kv in leg.split(",")))
        r = subprocess.run([sys.executable, "-c", SNIPPET, root, str(steps), B, W], capture_output=True, text=True, env=env, cwd=root, timeout=1500)
        if r.returncode != 0 or "SOAK_DIGEST" not in r.stdout:
            print(f"leg {leg}: FAILED\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
            sys.exit(1)
        line = r.stdout.split("SOAK_DIGEST")[1].strip()
        digests[leg] = line.split()[0]
        print(f"shape {shape} steps {steps} leg {leg}: {line}")
    ok = len(set(digests.values())) == 1
    print("AGREE" if ok else "DIFFERENT")
    sys.exit(0 if ok else 2)


if __name__ == "__main__":
    main()
