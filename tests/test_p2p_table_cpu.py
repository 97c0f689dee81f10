"""CPU: the copy tables of the P2P exchange engine (p2p.P2PGroup._build_table) against a Python restatement of the push kernel's
chunk walk (csrc/comm.hip, p2p_copy_chunks): 2-D pieces -- strided sources, pitched destinations, rows longer and shorter than a
64-KiB chunk, odd sizes -- must copy every byte exactly once to the right place, whatever the number of workgroups."""
import numpy as np
import pytest
import torch

from bind_your_avatar_implementation_amd import p2p

CHUNK = p2p.CHUNK


class _FakeGroup:
    """Just enough of a P2PGroup for _build_table: named destination buffers that live in host memory."""
    _rows_of = staticmethod(p2p.P2PGroup._rows_of)
    _build_table = p2p.P2PGroup._build_table

    def __init__(self, buffers):
        self.dev = torch.device("cpu")
        self.ctrl = torch.zeros(4, dtype=torch.int32)
        self._bufs = buffers

    def peers(self, name):
        return [p2p._Peer(t.data_ptr(), t.shape, t.dtype, t) for t in self._bufs[name]]


def run_table(table, n_rows, total_chunks, grid, mem, world=1, rank=0):
    """p2p_copy_chunks: workgroup b takes iterations b, b + grid, ... of the walk down the columns of a [world x share]
    arrangement of the chunk indices (consecutive iterations = different peers' shares; rank r opens on peer r + 1);
    `mem` maps an address to (numpy byte array, base address)."""
    def view(addr, n):
        for arr, base in mem:
            if base <= addr and addr + n <= base + arr.size:
                return arr[addr - base:addr - base + n]
        raise AssertionError(f"address {addr:#x} + {n} outside every buffer")
    rows = table.tolist()
    share = -(-total_chunks // world)
    seen = []
    for b in range(grid):
        for it in range(b, share * world, grid):
            c = ((it + rank + 1) % world) * share + it // world
            if c >= total_chunks:
                continue
            seen.append(c)
            i = 0
            while i + 1 < n_rows and rows[i + 1][6] <= c:
                i += 1
            src, dst, row_bytes, nrow, sp, dp, chunk0 = rows[i]
            if row_bytes <= 0 or nrow <= 0:
                continue
            ci = c - chunk0
            if row_bytes > CHUNK:
                cpr = -(-row_bytes // CHUNK)
                r0, col0 = divmod(ci, cpr)
                col0 *= CHUNK
                nr, n = 1, min(CHUNK, row_bytes - col0)
            else:
                rpc = CHUNK // row_bytes
                r0, col0 = ci * rpc, 0
                nr, n = min(rpc, nrow - r0), row_bytes
            assert nr >= 1 and n >= 1
            for r in range(nr):
                d = view(dst + (r0 + r) * dp + col0, n)
                d += 1                                                  # count the writes of every byte ...
                view(dst + (r0 + r) * dp + col0 + (1 << 40), n)[:] = view(src + (r0 + r) * sp + col0, n)     # ... and copy (shadow)
    assert sorted(seen) == list(range(total_chunks))                    # the walk is a permutation of the chunks
    return seen


def test_chunk_walk_spreads_concurrent_workgroups_over_the_peers():
    """What the 64 workgroups of a push work on at the same time must not all go to one peer (xGMI links are point to point):
    with pieces listed peer by peer -- the layout of every table the engine builds -- the first `world` iterations of the
    walk land in `world` different peers' shares, starting with peer rank + 1."""
    world, per_peer = 8, 81                                  # the packed q|k|v exchange of an 8-rank step: 81 chunks per peer
    total = world * per_peer
    for rank in (0, 3, 7):
        share = -(-total // world)
        first = [(((it + rank + 1) % world) * share + it // world) // per_peer for it in range(64)]
        assert first[:world] == [(rank + 1 + j) % world for j in range(world)]
        assert all(first.count(pr) == 8 for pr in range(world))          # 64 workgroups: 8 on every link
    # ragged totals: still a permutation
    for total, world in ((1, 8), (5, 8), (17, 4), (640, 8), (641, 8), (100, 3)):
        share = -(-total // world)
        cs = [((it + 2) % world) * share + it // world for it in range(share * world)]
        assert sorted(c for c in cs if c < total) == list(range(total))


@pytest.mark.parametrize("grid,world,rank", [(1, 1, 0), (7, 2, 1), (64, 8, 5), (64, 2, 0)])
def test_copy_table_covers_every_byte_once(grid, world, rank):
    torch.manual_seed(0)
    P, L, F = 5, 37, 96                                    # pairs x locations x features: row = 96 bf16 = 192 bytes
    xa = torch.randn(P, L, F).to(torch.bfloat16)
    big = torch.randn(3, 50000).to(torch.bfloat16)         # rows of 100000 bytes: longer than a chunk
    small = torch.randn(7, 3).to(torch.bfloat16)           # 6-byte rows: the 2-byte path
    dst_a = [torch.zeros(P, 11, F, dtype=torch.bfloat16), torch.zeros(P, 26, F, dtype=torch.bfloat16)]       # per peer: its own shape
    dst_b = [torch.zeros(3, 60000, dtype=torch.bfloat16)] * 2
    dst_c = [torch.zeros(7, 40, dtype=torch.bfloat16)] * 2
    g = _FakeGroup({"a": dst_a, "b": dst_b, "c": dst_c})
    pieces = [(xa[:, 0:11], 0, "a", 0),                    # strided source (rows = pairs), dense destination
              (xa[:, 11:37], 1, "a", 0),
              (big, 1, "b", 5000, 60000),                  # contiguous source scattered to pitched destination rows
              (small, 0, "c", 17, 40),
              (torch.zeros(0, dtype=torch.bfloat16), 0, "c", 0),        # an empty piece is skipped
              (xa[2, 5, :20], 0, "c", 6 * 40 + 20)]         # a contiguous piece: one row, behind the last row's 2-D piece
    table, n, total_chunks, _ = g._build_table(pieces)
    assert n == 5 and table.shape == (5, 7)
    # memory model: the real destination tensors count writes in a byte view; a shadow space 2^40 above them takes the data
    mem, shadows = [], []
    for t in (xa, big, small):
        mem.append((t.view(torch.uint8).reshape(-1).numpy(), t.data_ptr()))
    for t in dst_a + dst_b[:1] + dst_c[:1]:
        counts = np.zeros(t.numel() * 2, dtype=np.uint8)
        shadow = np.zeros(t.numel() * 2, dtype=np.uint8)
        mem.append((counts, t.data_ptr()))
        mem.append((shadow, t.data_ptr() + (1 << 40)))
        shadows.append((t, counts, shadow))
    run_table(table, n, total_chunks, grid, mem, world, rank)

    def got(t):
        for tt, counts, shadow in shadows:
            if tt is t:
                return counts, torch.from_numpy(shadow.copy()).view(torch.bfloat16).view(tt.shape)
    c0, a0 = got(dst_a[0])
    assert torch.equal(a0, xa[:, 0:11]) and c0.min() == 1 and c0.max() == 1
    c1, a1 = got(dst_a[1])
    assert torch.equal(a1, xa[:, 11:37]) and c1.min() == 1 and c1.max() == 1
    cb, b_ = got(dst_b[0])
    assert torch.equal(b_[:, 5000:55000], big) and cb.reshape(3, -1)[:, 10000:110000].min() == 1 and cb.sum() == big.numel() * 2
    cc, c_ = got(dst_c[0])
    assert torch.equal(c_[:, 17:20], small) and torch.equal(c_[6, 20:40], xa[2, 5, :20])
    assert cc.max() == 1 and cc.sum() == (small.numel() + 20) * 2


def test_copy_table_rejects_what_the_kernel_cannot_copy():
    g = _FakeGroup({"a": [torch.zeros(4, 8, dtype=torch.bfloat16)]})
    with pytest.raises(ValueError, match="does not fit"):
        g._build_table([(torch.zeros(5, 8, dtype=torch.bfloat16), 0, "a", 0)])
    with pytest.raises(ValueError, match="does not fit"):
        g._build_table([(torch.zeros(2, 8, dtype=torch.bfloat16), 0, "a", 0, 4)])         # destination pitch shorter than a row
    with pytest.raises(ValueError, match="dense rows"):
        g._build_table([(torch.zeros(4, 8, 2, dtype=torch.bfloat16)[:, :, 0], 0, "a", 0)])
    with pytest.raises(ValueError, match="does not fit"):
        g._build_table([(torch.zeros(4, 8, dtype=torch.float32), 0, "a", 0)])               # dtype of the buffer
