"""P2P exchange engine, host side: the exchanges of the sharded denoise step as PUSH kernels that store straight into
the peers' HBM over xGMI (``bya_p2p_push`` / ``bya_p2p_wait``, csrc/comm.hip, include/bya.h).

The reference has no inference parallelism; BASELINE.json asks for it.  Round 3 issued every
exchange through ``torch.distributed`` (RCCL): ~360 collectives per rank-step at ~20 us each, none of which can be
captured in a hipGraph on this stack.  Here an exchange is ONE ordinary kernel launch whatever its scatter/gather list,
and the whole sharded step replays as a graph.

``P2PGroup(group, device)``        one per process group; collective (every rank constructs it at the same point)
``.symmetric(name, shape, dtype)``  a buffer every rank allocates under the same name: returns the local tensor; peers'
                                    copies are mapped into this process through hipIpc (torch's CUDA storage sharing),
                                    handles traded ONCE over the process group (any backend: gloo works)
``.channel(key, pieces)``          the channel (control block, sequence counters) of the logical exchange ``key``, bound to
                                    the copy table of ``pieces`` = [(src, peer, name, offset)], "copy the contiguous local
                                    tensor ``src`` to element ``offset`` of buffer ``name`` on rank ``peer``"; one table per
                                    set of source addresses (workspaces change, captured graphs own theirs), built once
``Channel.exchange(side=False)``   push + wait on the current stream (``side``: the push runs on the group's side stream
                                    behind the work already enqueued; call ``.wait()`` where the data is needed)

All ranks of the group must issue the same sequence of ``symmetric`` calls and, per channel, the same sequence of
exchanges (sequence numbers live in device memory, one pair per channel).  One process per GPU in production; the tests
run several processes on ONE GPU -- hipIpc maps a peer's buffer of the same device just the same.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _hip

MAX_CHANNELS = 64
CTRL_WORDS = 64
CHUNK = 64 * 1024


class Channel:
    """One exchange of the step: a control block (flags + device-resident sequence counters) and, per set of source
    pointers it has been used with, a device-resident copy table.  The engine re-allocates its workspace when the geometry
    changes and gives every captured hipGraph a workspace of its own: the SAME logical exchange then comes with other
    source addresses -- it keeps its channel (and its sequence numbers, which all ranks advance in step) and gets one more
    table; tables are never freed (a captured graph may hold their address; they are ~32 bytes per piece)."""

    def __init__(self, grp, index):
        self.grp, self.index = grp, index
        self.ctrl_ptr = grp.ctrl[index].data_ptr()
        self.peer_ctrl = torch.tensor([grp.ctrl_peers[j][index].data_ptr() for j in range(grp.world)], dtype=torch.int64,
                                      device=grp.dev)
        self.tables = {}                 # source-pointer signature -> (table tensor, n_copies, total_chunks, sources)
        self.cur = None
        self._join = None

    def bind(self, pieces):
        sig = tuple((p[0].data_ptr(), p[1], p[2], p[3], p[0].numel()) for p in pieces)
        ent = self.tables.get(sig)
        if ent is None:
            ent = self.tables[sig] = self.grp._build_table(pieces)
        self.cur = ent
        return self

    def push(self, side=False):
        g = self.grp
        lib = _hip.load()
        table, n, total_chunks, _ = self.cur
        cur = torch.cuda.current_stream(g.dev)
        stream = cur
        if side and g.world > 1:
            stream = g.side_stream
            ev = torch.cuda.Event()
            ev.record(cur)
            stream.wait_event(ev)
        _hip.check(lib.bya_p2p_push(table.data_ptr(), n, total_chunks, self.peer_ctrl.data_ptr(), g.world, g.rank,
                                    self.ctrl_ptr, stream.cuda_stream), "bya_p2p_push")
        self._join = None
        if stream is not cur:
            self._join = torch.cuda.Event()
            self._join.record(stream)
        g.pushes += 1
        return self

    def wait(self):
        g = self.grp
        cur = torch.cuda.current_stream(g.dev)
        if self._join is not None:
            cur.wait_event(self._join)              # (also rejoins the side stream into a graph capture)
            self._join = None
        _hip.check(_hip.load().bya_p2p_wait(self.ctrl_ptr, g.world, cur.cuda_stream), "bya_p2p_wait")
        return self

    def exchange(self):
        return self.push().wait()


class P2PGroup:
    def __init__(self, group=None, device=None):
        self.group = group if group is not None else dist.group.WORLD
        self.world, self.rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if self.world > 32:
            raise ValueError("P2P exchange engine: at most 32 ranks (one node)")
        self.dev = torch.device(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        self._named = {}
        self._keep = []
        self._channels = {}
        self._p2p_enabled = set()
        self.pushes = 0
        self.side_stream = torch.cuda.Stream(self.dev)
        self.ctrl = self.symmetric("__ctrl__", (MAX_CHANNELS, CTRL_WORDS), torch.int32, zero=True)
        self.ctrl_peers = self._named["__ctrl__"][1]

    # ---- symmetric buffers -------------------------------------------------------------------------------------------
    def symmetric(self, name, shape, dtype=torch.bfloat16, zero=False):
        """COLLECTIVE on first use of ``name`` (every rank, same order).  -> local tensor; ``peers(name)[j]`` = rank j's
        tensor mapped here, in rank j's OWN shape (shapes may differ between ranks: uneven shards)."""
        ent = self._named.get(name)
        if ent is not None:
            loc = ent[0]
            if tuple(loc.shape) != tuple(shape) or loc.dtype != dtype:
                raise ValueError(f"symmetric buffer {name!r} exists with shape {tuple(loc.shape)} {loc.dtype}")
            return loc
        with torch.cuda.device(self.dev):
            # the WHOLE allocation is what a peer maps, and it must stay alive for as long as any peer may store into it
            # (= the life of this object)
            local = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
            torch.cuda.synchronize(self.dev)
            peers = [None] * self.world
            peers[self.rank] = local
            if self.world > 1:
                info = local.untyped_storage()._share_cuda_()
                infos = [None] * self.world
                dist.all_gather_object(infos, (info, local.storage_offset(), tuple(shape)), group=self.group)
                for j, (inf, off, shp) in enumerate(infos):
                    if j == self.rank:
                        continue
                    st = torch.UntypedStorage._new_shared_cuda(*inf)
                    self._keep.append(st)
                    # (the mapped storage carries the OWNER's device index; only its address is used, by kernels of this GPU)
                    peers[j] = torch.empty(0, dtype=dtype, device=st.device).set_(st, off, shp)
                    if st.device != self.dev and (self.dev.index, st.device.index) not in self._p2p_enabled:
                        # one process sees several GPUs (torchrun without per-rank visibility): a kernel of THIS GPU may only
                        # dereference the peer's memory once peer access is enabled; torch does that on the first P2P copy
                        probe = torch.empty(1, dtype=dtype, device=self.dev)
                        probe.copy_(peers[j].reshape(-1)[:1])
                        self._p2p_enabled.add((self.dev.index, st.device.index))
                dist.barrier(group=self.group)          # nobody pushes before everybody has mapped
        self._named[name] = (local, peers)
        return local

    def peers(self, name):
        return self._named[name][1]

    def has(self, name):
        return name in self._named

    # ---- channels ------------------------------------------------------------------------------------------------------
    def channel(self, key, pieces):
        """The channel of the logical exchange ``key`` (created on first use: every rank, same order), bound to ``pieces`` =
        [(contiguous local tensor, peer rank, receive-buffer name, element offset there)]."""
        ch = self._channels.get(key)
        if ch is None:
            if len(self._channels) >= MAX_CHANNELS:
                raise RuntimeError("P2P exchange engine: out of channels")
            ch = self._channels[key] = Channel(self, len(self._channels))
        return ch.bind(pieces)

    def _build_table(self, pieces):
        rows, chunk0 = [], 0
        for src, peer, name, offset in pieces:
            nbytes = src.numel() * src.element_size()
            if nbytes == 0:
                continue
            dst_t = self.peers(name)[peer]
            if not src.is_contiguous():
                raise ValueError("P2P piece sources must be contiguous")
            if dst_t.dtype != src.dtype or offset < 0 or offset + src.numel() > dst_t.numel():
                raise ValueError(f"P2P piece does not fit buffer {name!r} on rank {peer}")
            dst = dst_t.data_ptr() + offset * src.element_size()
            if (src.data_ptr() | dst | nbytes) & 1:
                raise ValueError("P2P pieces must be 2-byte aligned in address and size")
            rows.append((src.data_ptr(), dst, nbytes, chunk0))
            chunk0 += (nbytes + CHUNK - 1) // CHUNK
        if not rows:                                     # nothing to send: still takes part in the flag protocol
            rows, chunk0 = [(self.ctrl.data_ptr(), self.ctrl.data_ptr(), 0, 0)], 1
        table = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return table, len(rows), chunk0, [p[0] for p in pieces]      # (the sources stay alive with the table)

    def self_test(self):
        """One tiny all-to-all through the engine, checked: every rank must see every peer's value (raises otherwise).
        Run once at set-up, so that a platform where peer stores do not arrive is noticed before the first step."""
        n = 64
        recv = self.symmetric("__selftest__", (self.world, n), torch.float32, zero=True)
        send = [torch.full((n,), 1000.0 + self.rank * 32 + j, dtype=torch.float32, device=self.dev) for j in range(self.world)]
        self.channel("__selftest__", [(send[j], j, "__selftest__", self.rank * n) for j in range(self.world)]).exchange()
        want = torch.tensor([1000.0 + r * 32 + self.rank for r in range(self.world)], device=self.dev)[:, None].expand(-1, n)
        torch.cuda.synchronize(self.dev)
        if self.timeouts() or not torch.equal(recv, want):
            raise RuntimeError("P2P exchange self-test failed: peer stores did not arrive")
        if self.world > 1:
            dist.barrier(group=self.group)

    def timeouts(self):
        """Waits that gave up (0 on a healthy run); synchronises."""
        torch.cuda.synchronize(self.dev)
        return int(self.ctrl[:, 35].sum().item())


