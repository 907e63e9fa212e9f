"""Diagnostic: the batch-axis rank steps of a world-N job run one after the other in ONE process (staged path, collectives
intercepted), their [gradient | loss | priorities] buffers summed on the host and compared with the unsharded step."""
import os, sys, numpy as np, torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/tests', ROOT + '/oracle', ROOT + '/tests/golden']
import morl_baselines_amd.native as native
import test_distributed as td
from morl_baselines_amd.distributed import shard_envelope_agent
B, W, arch, world = int(sys.argv[1]), int(sys.argv[2]), tuple(int(x) for x in sys.argv[3].split(',')), int(sys.argv[4])
gpu = th.cuda.is_available()
if gpu:
    lib, dev = native.load_library(), th.device("cuda:0")
else:
    import simlib
    lib, dev = simlib.load_sim(), th.device("cpu")
    native.use_library(lib)
ag0 = td._make_agent(lib, False, False, dev=dev, arch=arch, B=B, W=W)
P = ag0.q_net.ctx.n_params
g0 = th.zeros(P, device=dev)
ag0._grads.zero_()
ag0.update()
print("unsharded loss", ag0.last_loss())
tot = None
for r in range(world):
    ag = td._make_agent(lib, False, False, dev=dev, arch=arch, B=B, W=W)
    got = []
    class D(td._OneRank):
        def all_reduce(self, t, op=None, group=None):
            got.append(t.clone())
            raise RuntimeError("stop")
    shard_envelope_agent(ag, D(), emulate=(world, r), axis="batch")
    try:
        ag.update()
    except RuntimeError as e:
        assert "stop" in str(e)
    g = got[0].double().cpu()
    print(" rank", r, "loss share", float(g[P]), "|g|", float(g[:P].norm()))
    tot = g if tot is None else tot + g
print("summed loss", float(tot[P]), " unsharded", ag0.last_loss())
