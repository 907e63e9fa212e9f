"""Diagnostic: gloo all_reduce on device tensors with N ranks sharing one GPU (the transport of the shared-GPU rank-step test)."""
import os, sys, torch as th, torch.multiprocessing as mp

def worker(rank, world, port, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    th.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = th.device("cuda:0")
    bad = 0
    for it in range(20):
        n = 135431 + it
        x = th.full((n,), float(rank + 1), device=dev) * (it + 1)
        y = th.randn(2048, 2048, device=dev)
        z = y @ y                                   # work in flight on the stream before the collective
        if mode == "ext":
            with th.cuda.stream(th.cuda.ExternalStream(th._C._cuda_getCurrentRawStream(0), device=dev)):
                dist.all_reduce(x)
        elif mode == "host":
            h = x.cpu(); dist.all_reduce(h); x.copy_(h)
        else:
            dist.all_reduce(x)
        w = x * 2.0                                  # consumer on the stream right behind it
        th.cuda.synchronize()
        want = 2.0 * (it + 1) * sum(range(1, world + 1))
        if not bool((w == want).all()):
            bad += 1
            print(f"rank {rank} it {it} mode {mode}: mismatch, got {w[:3].tolist()} .. {w[-3:].tolist()} want {want}", flush=True)
    print(f"rank {rank} mode {mode}: {bad} bad of 20", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1])
    for k, mode in enumerate(("plain", "ext", "host")):
        mp.spawn(worker, args=(world, 29871 + k, mode), nprocs=world, join=True)
