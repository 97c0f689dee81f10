"""The RCCL code paths on ONE GPU (the test box has one): a process group of world size 1 on the ``nccl`` backend (= RCCL)
and a 1-rank ``ncclComm_t`` for the two C-ABI wrappers.  Nothing crosses a link, but everything else of the exchanges
runs for real: RCCL symbol resolution, bf16 datatype codes, split lists, the async handles of
``all_to_all_single(async_op=True)`` on the communicator's own stream and the compute stream's wait on them,
``all_gather_into_tensor`` on device tensors -- the branches of ``parallel.py`` / ``engine.py`` that the gloo tests
(host-staged) never reach.  ``parallel.FORCE_COLLECTIVES`` makes the engine take the sharded path with one rank; the
result must equal the plain single-GPU step BIT FOR BIT (same kernels, same row counts, copies in between).

Each case runs in a spawned child process: a failed RCCL initialisation must not poison the pytest process."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine_worker(rank, port, ret, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["BYA_SP_TRANSPORT"] = "torch"             # this file covers the torch.distributed / RCCL transport
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel, parallel
        from bind_your_avatar_implementation_amd.parallel import shard_sequence
        from bind_your_avatar_implementation_amd.synth import synth_inputs
        from test_forward_gpu import SMALL_KW, to_dev
        n_id = 3 if mode == "three_ids" else 2
        model = BindyouravatarTransformer3DModel(**SMALL_KW, device=dev).init_synthetic(seed=1, fast=True)
        inp = to_dev(synth_inputs(batch=1, frames=3, height=16, width=24, seed=3, n_id=n_id), dev)
        full = model(**inp)[0].clone()
        if mode == "allgather":
            os.environ["BYA_SP_ALLGATHER"] = "1"          # exchange A in its K/V all-gather form
            os.environ["BYA_ROUTER_REPLICATED"] = "1"     # exchange B as one all-gather of the router rows
        parallel.FORCE_COLLECTIVES = True
        shard_sequence(model, dist.group.WORLD)
        assert dist.get_backend(model._seq_group) == "nccl"
        part = model(**inp)[0]
        again = model(**inp)[0]                           # second step: every exchange buffer is reused
        torch.cuda.synchronize()
        ret["equal"] = bool(torch.equal(part, full)) and bool(torch.equal(again, full))
        if mode == "graph":
            # use_hip_graph on a SHARDED model: RCCL collectives cannot be captured on this stack (even a lone
            # all_to_all_single under torch.cuda.graph never returns: tools/rccl_graph_probe.py), so the model must notice
            # and run the sharded step eagerly instead of hanging in a capture
            model.use_hip_graph = True
            assert not model._graph_capturable()
            g1 = model(**inp)[0].clone()
            torch.cuda.synchronize()
            ret["graph_equal"] = bool(torch.equal(g1, full))
            ret["graphs"] = len(model._graphs)
        ret["maxdiff"] = float((part.float() - full.float()).abs().max())
        ret["counters"] = dict(parallel.COLLECTIVE_CALLS)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["head_parallel", "allgather", "three_ids", "graph"])
def test_engine_collective_paths_on_a_one_rank_rccl_group(dev, mode):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_engine_worker, args=(31500 + os.getpid() % 1000 + len(mode), ret, mode), nprocs=1, join=True)
    print(mode, dict(ret))
    assert ret["equal"], dict(ret)
    c = ret["counters"]
    if mode == "allgather":
        assert c.get("all_gather", 0) > 0
    else:
        # every exchange runs on the communicator's stream (async_op): 2 layers x (3 in + 1 out) of the joint attention
        # + the 7 repartitions of the one routing layer's four blocks; the logits and the output are all-gathered
        assert c.get("all_to_all_async", 0) >= 15 and c.get("all_to_all", 0) == 0 and c.get("all_gather", 0) > 0
    if mode == "graph":
        assert ret["graph_equal"] and ret["graphs"] == 0, dict(ret)


def _load_rccl():
    """The RCCL copy already mapped into this process (torch links it), found the way csrc/comm.hip finds it."""
    import torch  # noqa: F401  (maps librccl)
    for name in ("librccl.so", "librccl.so.1"):
        try:
            return ctypes.CDLL(name, mode=os.RTLD_NOW | os.RTLD_NOLOAD)
        except OSError:
            continue
    return ctypes.CDLL("librccl.so", mode=os.RTLD_NOW | os.RTLD_GLOBAL)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _abi_worker(rank, ret):
    from bind_your_avatar_implementation_amd import _hip
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    rccl = _load_rccl()
    uid = _UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0 and comm.value
    lib = _hip.load()
    side = torch.cuda.Stream()                              # "give them a stream of their own"
    g = torch.Generator().manual_seed(0)
    rows, elems = 333, 3072
    k = torch.randn(rows, elems, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(rows, elems, generator=g).to(torch.bfloat16).to(dev)
    kf, vf = torch.zeros_like(k), torch.zeros_like(v)
    torch.cuda.synchronize()
    rc = lib.bya_allgather_kv(k.data_ptr(), v.data_ptr(), kf.data_ptr(), vf.data_ptr(), rows, elems, comm, side.cuda_stream)
    side.synchronize()
    ret["allgather_rc"], ret["allgather_ok"] = rc, bool(torch.equal(kf, k) and torch.equal(vf, v))
    n = rows * elems
    recv = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()
    cnt = (ctypes.c_int64 * 1)(n)
    rc = lib.bya_alltoall_router(k.data_ptr(), recv.data_ptr(), cnt, cnt, 1, comm, side.cuda_stream)
    side.synchronize()
    ret["alltoall_rc"], ret["alltoall_ok"] = rc, bool(torch.equal(recv.view(rows, elems), k))
    cnt2 = (ctypes.c_int64 * 2)(n // 2, n // 2)
    ret["wrong_world_rc"] = lib.bya_alltoall_router(k.data_ptr(), recv.data_ptr(), cnt2, cnt2, 2, comm, side.cuda_stream)
    neg = (ctypes.c_int64 * 1)(-5)
    ret["negative_count_rc"] = lib.bya_alltoall_router(k.data_ptr(), recv.data_ptr(), neg, neg, 1, comm, side.cuda_stream)
    side.synchronize()
    rccl.ncclCommDestroy(comm)


def test_c_abi_rccl_wrappers_with_a_one_rank_communicator(dev):
    """bya_allgather_kv / bya_alltoall_router (include/bya.h) against a real ncclComm_t: the results equal the inputs, a
    world that does not match the communicator and a negative count are refused."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_abi_worker, args=(ret,), nprocs=1, join=True)
    print(dict(ret))
    assert ret["allgather_rc"] == 0 and ret["allgather_ok"]
    assert ret["alltoall_rc"] == 0 and ret["alltoall_ok"]
    assert ret["wrong_world_rc"] == -1 and ret["negative_count_rc"] != 0
