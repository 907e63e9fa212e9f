"""Diagnostic: N ranks on ONE GPU (gloo), per-step losses of the staged and the one-call rank step against the unsharded agent."""
import os, sys, numpy as np, torch as th, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/tests', ROOT + '/oracle', ROOT + '/tests/golden']

def worker(rank, world, port, axis, transport, per):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import morl_baselines_amd.native as native
    import test_distributed as td
    from morl_baselines_amd.distributed import shard_envelope_agent
    th.cuda.set_device(0)
    dev = th.device("cuda:0")
    lib = native.load_library()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ag = td._make_agent(lib, per, True, dev=dev, arch=(256, 256, 256), B=64, W=16)
    shard_envelope_agent(ag, dist, axis=axis, transport=transport)
    out = []
    nosync = os.environ.get("DIAG_NOSYNC", "0") == "1"
    for _ in range(6):
        ag.update(); ag.global_step += 1
        if not nosync:
            th.cuda.synchronize()
            out.append(round(ag.last_loss(), 7))
    if nosync:
        th.cuda.synchronize()
        out.append(round(ag.last_loss(), 7))
    if rank == 0:
        print(f"  {axis:8s} {str(transport):7s} per={per} world {world}: {out}  |p|={float(ag.q_net.flat.double().norm()):.9f}", flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    import morl_baselines_amd.native as native
    import test_distributed as td
    world = int(sys.argv[1])
    lib = native.load_library(); dev = th.device("cuda:0")
    for per in (False,):
        ref = td._make_agent(lib, per, True, dev=dev, arch=(256, 256, 256), B=64, W=16)
        out = []
        for _ in range(6):
            ref.update(); ref.global_step += 1
            out.append(round(ref.last_loss(), 7))
        print(f"unsharded per={per}: {out}  |p|={float(ref.q_net.flat.double().norm()):.9f}", flush=True)
        k = 0
        for axis in ("batch", "weights"):
            for transport in ((None,) if os.environ.get("DIAG_ONECALL", "0") == "1" else ("staged", None)):
                k += 1
                mp.spawn(worker, args=(world, 29971 + k, axis, transport, per), nprocs=world, join=True)
