"""Token-axis (sequence) sharding of one denoise step across the GPUs of a node: one process per GPU,
``torch.distributed`` with the ``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` in the CPU tests.

The reference has no inference parallelism at all; this is new capability required by
BASELINE.json.  Partition: the joint sequence ``[text (Tt) | video (N)]`` of S = Tt + N rows is cut into ``world``
equal contiguous row ranges (S = 17776 divides by 2, 4 and 8).  Everything on the path is row-local EXCEPT:

  * joint self-attention: every rank needs all keys/values -> one all-gather of K and one of V per layer
    (``gather_rows``), queries stay local;
  * the Embedding Router's spatial / temporal attentions, which mix tokens of a frame / of a location: the
    512-wide router feature rows are all-gathered once per routing layer (36 MB) and the four small
    SpatialTemporalAttentionBlocks run replicated (``gather_video_rows``); each rank keeps its rows of the logits;
  * the final unpatchify, which needs every token's 64 output channels (2 MB all-gather).

Weights are replicated (17 GB of 288 GB).  There is no reduce: no GEMM is K-split.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class SeqShard:
    """Row range of this rank.  Global rows [r0, r1); text rows are global rows < Tt."""
    rank: int
    world: int
    S: int
    Tt: int
    group: object = None

    def __post_init__(self):
        if self.S % self.world:
            raise ValueError(f"sequence length {self.S} is not divisible by {self.world} ranks")
        self.S_loc = self.S // self.world
        self.r0, self.r1 = self.rank * self.S_loc, (self.rank + 1) * self.S_loc
        if self.world > 1 and self.Tt > self.S_loc:
            raise ValueError("text rows must fit inside the first rank's shard")
        self.Tt_loc = max(0, min(self.Tt, self.r1) - self.r0)          # local text rows (rank 0 only)
        self.v0 = max(self.r0, self.Tt) - self.Tt                      # first local video token (global index)
        self.v1 = self.r1 - self.Tt
        self.N_loc = self.v1 - self.v0
        self.N = self.S - self.Tt

    # ---- collectives ------------------------------------------------------------------------------------
    def _all_gather(self, out, local):
        """RCCL all-gather; with the gloo backend (single-GPU functional tests, CPU tests) device tensors are
        staged through host memory because gloo has no device all-gather."""
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host_out, local.cpu(), group=self.group)
            out.copy_(host_out)
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)

    def gather_rows(self, local, out=None):
        """[S_loc, F] (every rank the same shape) -> [S, F], rank-major == global row order."""
        if self.world == 1:
            return local
        if out is None:
            out = torch.empty(self.S, *local.shape[1:], dtype=local.dtype, device=local.device)
        self._all_gather(out, local.contiguous())
        return out

    def gather_video_rows(self, local_video, scratch=None):
        """[..., N_loc, F] per rank (rank 0 owns fewer video rows: its shard starts with the text rows)
        -> [..., N, F].  Implemented as an equal-size row all-gather of Tt_loc junk rows + the video rows."""
        if self.world == 1:
            return local_video
        lead = local_video.shape[:-2]
        F = local_video.shape[-1]
        flat = local_video.reshape(-1, self.N_loc, F)
        C = flat.shape[0]
        pad = torch.empty(C, self.S_loc, F, dtype=flat.dtype, device=flat.device) if scratch is None else scratch
        pad[:, self.Tt_loc:] = flat
        if self.Tt_loc:
            pad[:, :self.Tt_loc] = 0
        full = torch.empty(self.world * C, self.S_loc, F, dtype=flat.dtype, device=flat.device)
        self._all_gather(full, pad)                                       # rank-major concatenation along dim 0
        full = full.view(self.world, C, self.S_loc, F).permute(1, 0, 2, 3).reshape(C, self.S, F)[:, self.Tt:]
        return full.reshape(*lead, self.N, F).contiguous()

    def frame_segments(self, per_frame):
        """Local video rows split at frame boundaries: [(frame, local_start, length)]."""
        out, v = [], self.v0
        while v < self.v1:
            f = v // per_frame
            end = min(self.v1, (f + 1) * per_frame)
            out.append((f, v - self.v0, end - v))
            v = end
        return out


def shard_sequence(model, group=None):
    """Switch ``model`` (BindyouravatarTransformer3DModel) to sequence-parallel execution over ``group``."""
    model._seq_group = group if group is not None else dist.group.WORLD
    model._seq_world = dist.get_world_size(model._seq_group)
    model._seq_rank = dist.get_rank(model._seq_group)
    model.invalidate_engine()
    return model
