"""CPU, world_size 2 over gloo: the sequence-sharding data movement of bind_your_avatar_implementation_amd.parallel
(the N > 1 path of the engine).  Compute inside the checks is plain torch (the HIP kernels need a GPU); what is
verified is that shard -> all-gather -> local work -> gather reproduces the unsharded result exactly."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _worker(rank, world, port, S, Tt, per_frame, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bind_your_avatar_implementation_amd.parallel import SeqShard
        torch.manual_seed(0)                      # same "global" tensors on every rank
        H, D = 4, 16
        sh = SeqShard(rank, world, S, Tt, dist.group.WORLD)
        N = S - Tt
        assert sh.r1 - sh.r0 in (S // world, S // world + 1) and sh.N_loc == sh.v1 - sh.v0
        assert (sh.Tt_loc == Tt) == (rank == 0)
        # exchange A: K/V all-gather + local queries == full attention on the local rows
        q, k, v = (torch.randn(S, H * D) for _ in range(3))
        kf, vf = sh.gather_rows(k[sh.r0:sh.r1].clone()), sh.gather_rows(v[sh.r0:sh.r1].clone())
        assert torch.equal(kf, k) and torch.equal(vf, v)
        hd = lambda t: t.view(-1, H, D).transpose(0, 1)
        full = F.scaled_dot_product_attention(hd(q)[None], hd(k)[None], hd(v)[None])[0].transpose(0, 1).reshape(S, -1)
        loc = F.scaled_dot_product_attention(hd(q[sh.r0:sh.r1])[None], hd(kf)[None], hd(vf)[None])[0]
        loc = loc.transpose(0, 1).reshape(sh.S_loc, -1)
        assert torch.allclose(loc, full[sh.r0:sh.r1], atol=1e-5)
        # exchange A, head-parallel form: rows -> heads all-to-all, attention on whole sequences of the local heads,
        # heads -> rows all-to-all; must equal the full attention restricted to this rank's rows
        Hl = H // world
        Dl = Hl * D
        def blocks(t):      # [world, S_loc, Dl]: local rows, column block j = heads of rank j
            return t[sh.r0:sh.r1].view(sh.S_loc, world, Dl).transpose(0, 1).contiguous()
        qh, kh, vh = (sh.rows_to_heads(blocks(t)) for t in (q, k, v))
        assert torch.equal(qh, q[:, rank * Dl:(rank + 1) * Dl])
        hl = lambda t: t.view(S, Hl, D).transpose(0, 1)
        oh = F.scaled_dot_product_attention(hl(qh)[None], hl(kh)[None], hl(vh)[None])[0].transpose(0, 1).reshape(S, Dl)
        mine = sh.heads_to_rows(oh)
        assert torch.allclose(mine, full[sh.r0:sh.r1], atol=1e-5)
        # exchange B: video-row gather (rank 0 owns the text rows, hence fewer video rows)
        feats = torch.randn(2, 3, N, 8)           # e.g. [B, n_id, N, F]
        got = sh.gather_video_rows(feats[:, :, sh.v0:sh.v1].contiguous())
        assert got.shape == feats.shape and torch.equal(got, feats)
        # per-frame segments of the local rows tile [v0, v1) exactly and never cross a frame
        segs = sh.frame_segments(per_frame)
        pos = sh.v0
        for f, start, length in segs:
            assert start == pos - sh.v0 and length > 0
            assert (pos // per_frame) == f == ((pos + length - 1) // per_frame)
            pos += length
        assert pos == sh.v1
        # output gather: every rank ends with the whole prediction
        y = torch.randn(1, N, 6)
        assert torch.equal(sh.gather_video_rows(y[:, sh.v0:sh.v1].contiguous()), y)
        # router repartition: frame-major <-> location-major all-to-all is a permutation (round trip = identity)
        from bind_your_avatar_implementation_amd.parallel import RouterPartition
        pairs, Fd = 6, 4
        rp = RouterPartition(rank, world, pairs, per_frame, dist.group.WORLD)
        glob = torch.arange(pairs * per_frame * Fd, dtype=torch.float32).view(pairs, per_frame, Fd)
        xa = glob[rp.pa0:rp.pa1].clone()
        xb = rp.a_to_b(xa)
        assert torch.equal(xb, glob[:, rp.lb0:rp.lb1])
        assert torch.equal(rp.b_to_a(xb), xa)
        assert torch.equal(rp.gather_b_rows(glob[:, rp.lb0:rp.lb1, :2].contiguous()), glob[:, :, :2])
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S,Tt,per_frame", [(64, 10, 9), (17776, 226, 1350), (65, 10, 11), (47027, 227, 3600)])
def test_sequence_shard_world2(S, Tt, per_frame):
    world = 2
    port = 29500 + (os.getpid() % 2000) + S % 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, S, Tt, per_frame, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_shard_geometry_single_process():
    from bind_your_avatar_implementation_amd.parallel import SeqShard
    for world in (1, 2, 4, 8):
        covered = 0
        for r in range(world):
            sh = SeqShard(r, world, 17776, 226)
            assert sh.S_loc * world == 17776
            covered += sh.N_loc
            assert sh.Tt_loc == (226 if r == 0 else 0)
        assert covered == 17550
    # uneven geometry (49 x 720 x 1280: 46800 tokens + 226 text rows over 8 ranks): sizes differ by at most one row
    sizes = [SeqShard(r, 8, 47026, 226).S_loc for r in range(8)]
    assert sum(sizes) == 47026 and max(sizes) - min(sizes) == 1
    assert sum(SeqShard(r, 8, 47026, 226).N_loc for r in range(8)) == 46800
    with pytest.raises(ValueError):
        SeqShard(0, 128, 17776, 226)              # the text rows no longer fit into rank 0's shard


def _cfg_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bind_your_avatar_implementation_amd.parallel import CfgSplit, SeqShard
        cs = CfgSplit(dist.group.WORLD)
        assert cs.half == rank // (world // 2) and cs.half_size == world // 2
        assert dist.get_world_size(cs.seq_group) == world // 2 and dist.get_world_size(cs.pair_group) == 2
        torch.manual_seed(0)
        batch = torch.randn(2, 3, 5)
        h = slice(cs.half, cs.half + 1)
        # the engine's argument tuple, sliced BY POSITION: tensors that merely happen to have a leading 2 (a 2-row
        # RoPE table, a [1 or 2, N, 2] forcing tensor, an unbatched [2, 2] af_matrix) must stay whole
        hid, enc, ts = torch.randn(2, 3, 4, 2, 2), torch.randn(2, 5, 8), torch.tensor([999, 999])
        rope = (torch.randn(2, 64), torch.randn(2, 64))
        idc = [torch.randn(2, 1280), torch.randn(2, 1280)]
        vit = [[torch.randn(2, 7, 16) for _ in range(5)] for _ in range(2)]
        audio, af, forcing = torch.randn(2, 2, 13, 12, 8), torch.eye(2)[None].repeat(2, 1, 1), torch.randn(2, 6, 2)
        mine = cs.take((hid, enc, ts, rope, idc, vit, audio, af, forcing))
        assert torch.equal(mine[0], hid[h]) and torch.equal(mine[1], enc[h]) and torch.equal(mine[2], ts[h])
        assert mine[3][0] is rope[0] and mine[8] is forcing
        assert torch.equal(mine[4][1], idc[1][h]) and torch.equal(mine[5][1][4], vit[1][4][h])
        assert torch.equal(mine[6], audio[h]) and torch.equal(mine[7], af[h])
        shared = cs.take((hid, enc, torch.tensor(999), None, None, None, None, torch.eye(2), None))
        assert shared[2].dim() == 0 and shared[7].shape == (2, 2) and shared[3] is None
        with pytest.raises(ValueError):
            cs.take((torch.randn(3, 3, 4, 2, 2), enc, ts, rope, idc, vit, audio, af, forcing))
        mine = [batch[h]]
        # each half computes f(sample) on its own (sequence-sharded inside the half when it has 2 ranks) ...
        f = lambda x: x * 2 + 1
        local = mine[0]
        if cs.half_size > 1:
            sh = SeqShard(dist.get_rank(cs.seq_group), cs.half_size, 4, 0, cs.seq_group)
            rows = torch.randn(2, 4, 6)[cs.half]                  # same on every rank (seeded)
            assert torch.equal(sh.gather_rows(rows[sh.r0:sh.r1].clone()), rows)
        # ... and the pair exchange restores the [uncond, cond] batch on every rank
        assert torch.equal(cs.join(f(local)), f(batch))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg_split(world):
    """CFG-batch split: 2 ranks = one sample each; 4 ranks = 2 x 2 with sequence sharding inside."""
    port = 31500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cfg_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def _ladder_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from bind_your_avatar_implementation_amd.parallel import shard_sequence

        class Model(torch.nn.Module):             # what shard_sequence touches of the transformer
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.zeros(1))
                self.invalidated = 0

            def invalidate_engine(self):
                self.invalidated += 1
        m = Model()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            shard_sequence(m, dist.group.WORLD, transport="p2p")
        # no GPU here: both P2P rungs fail at set-up ON EVERY RANK, the ranks agree and land on the collectives together
        assert m._seq_transport == "torch" and m._seq_p2p is None and m._seq_world == world and m._seq_rank == rank
        assert len(m._seq_transport_notes) == 2 and m._seq_transport_notes[0].startswith("p2p:")
        assert m._seq_transport_notes[1].startswith("p2p-fine:") and m.invalidated == 1
        assert any("running on 'torch'" in str(w.message) for w in caught)
        # the collectives still work afterwards (no rank is left inside a half-finished set-up collective)
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        assert t.item() == world * (world + 1) / 2
        shard_sequence(m, dist.group.WORLD, transport="torch")
        assert m._seq_transport == "torch" and m._seq_transport_notes == [] and m.invalidated == 2
        with pytest.raises(ValueError):
            shard_sequence(m, dist.group.WORLD, transport="carrier pigeon")
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_transport_ladder_walks_down_together_without_a_gpu():
    """shard_sequence's ladder p2p -> p2p-fine -> torch (bench.py --gpus N relies on it): where the P2P
    engine cannot be set up -- here: no GPU at all -- every rank records why, all ranks agree, and the model runs on
    torch.distributed collectives."""
    world, port = 2, 33100 + (os.getpid() % 1500)
    ret = mp.Manager().dict()
    mp.spawn(_ladder_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def _slip_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bind_your_avatar_implementation_amd.p2p import PhaseMismatch, gather_tagged
        g = dist.group.WORLD
        log = []
        # a rung of the transport ladder: set-up collectives ("handles", "agree"), then the ladder's own gather.  Rank 1
        # raises LOCALLY before the first set-up collective (a launch error, a refused mapping) and goes straight to the ladder.
        err = None
        try:
            if rank == 1:
                raise RuntimeError("local failure inside the set-up")
            gather_tagged(g, world, "handles", rank)
            gather_tagged(g, world, ("agree", "self-test"), None)
        except Exception as e:                       # noqa: BLE001  (shard_sequence catches everything a rung may raise)
            err = repr(e)
            log.append(type(e).__name__)
        for _ in range(3):
            try:
                ok = gather_tagged(g, world, ("ladder", "p2p"), err is None)
                break
            except PhaseMismatch as e:
                err = err or repr(e)
                log.append("retry")
        # in step again: the next collective pairs up and carries what it should on every rank
        nxt = gather_tagged(g, world, ("ladder", "torch"), rank * 10)
        ret[rank] = (ok, nxt, log)
    finally:
        dist.destroy_process_group()


def test_set_up_collectives_get_back_in_step_after_a_local_failure():
    """The advisor's round-5 finding: a rank that raises locally inside the P2P set-up reaches the ladder's object collective
    one collective early and from then on every rank reads somebody else's answers.  With tagged gathers every rank sees
    the slip in the same collective: the peers raise out of their set-up, the early rank repeats its gather, all agree that
    the rung failed, and the following collective is in step."""
    ret = mp.Manager().dict()
    mp.spawn(_slip_worker, args=(2, 29650 + os.getpid() % 300, ret), nprocs=2, join=True)
    for r in (0, 1):
        ok, nxt, log = ret[r]
        assert ok == [False, False] and nxt == [0, 10], ret[r]
    assert ret[0][2] == ["PhaseMismatch"] and ret[1][2] == ["RuntimeError", "retry"]
