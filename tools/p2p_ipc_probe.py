"""Can N processes that share ONE GPU map each other's device buffers (hipIpc through torch's storage sharing) and see each
other's device-side stores?  Feasibility probe for the P2P exchange engine.  python tools/p2p_ipc_probe.py [world]"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    buf = torch.full((1 << 20,), float(rank), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    info = buf.untyped_storage()._share_cuda_()
    infos = [None] * world
    dist.all_gather_object(infos, info)
    peers = []
    for j in range(world):
        if j == rank:
            peers.append(buf)
            continue
        st = torch.UntypedStorage._new_shared_cuda(*infos[j])
        peers.append(torch.empty(0, dtype=torch.float32, device=dev).set_(st, 0, (1 << 20,)))
    torch.cuda.synchronize()
    dist.barrier()
    seen = [float(p[12345].item()) for p in peers]
    dist.barrier()
    # every rank writes its id + 100 into element (rank) of every peer's buffer
    for j in range(world):
        peers[j][rank] = 100.0 + rank
    torch.cuda.synchronize()
    dist.barrier()
    mine = buf[:world].tolist()
    ret[rank] = (seen, mine)
    dist.barrier()
    del peers
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ret = mp.Manager().dict()
    t0 = time.time()
    mp.spawn(worker, args=(world, 29811, ret), nprocs=world, join=True)
    for r in range(world):
        print(r, ret[r])
    ok = all(ret[r][0] == [float(j) for j in range(world)] and ret[r][1] == [100.0 + j for j in range(world)] for r in range(world))
    print("IPC peer mapping on one GPU:", "OK" if ok else "FAILED", f"({time.time() - t0:.1f} s)")
