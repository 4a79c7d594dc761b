"""torch.distributed helpers of the multi-process bench and the gloo tests (control plane / CPU stand-in for the
RCCL all-reduce): max-reduction of packed scores and merge of per-rank mappings."""
import numpy as np

from nhd_amd.sharding import from_ordered_int64, to_ordered_int64


def allreduce_max_scores(score: np.ndarray, dist=None) -> np.ndarray:
    """Max-reduce packed scores over the default torch.distributed group (any backend)."""
    if dist is None:
        import torch.distributed as dist
    import torch
    t = torch.from_numpy(to_ordered_int64(score).copy())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return from_ordered_int64(t.numpy())


def merge_mappings(maps: np.ndarray, dist=None) -> np.ndarray:
    """Each pod's mapping is valid on exactly one rank (the winner's owner) and all-zero elsewhere."""
    if dist is None:
        import torch.distributed as dist
    import torch
    t = torch.from_numpy(maps.view(np.int8).astype(np.int32).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().astype(np.int8).view(maps.dtype).reshape(maps.shape)


class TorchTransport:
    """nhd_amd.sharding's transport interface over torch.distributed (gloo on CPU): what the product does with ncclSend / ncclRecv /
    ncclAllReduce behind the C-ABI (nhd_amd.sharding.RcclTransport), for the multi-process CPU tests and the bench's dry run."""

    def __init__(self, dist=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def sendrecv(self, send, dst, recv, src):
        import torch
        if self.world == 1 or (dst == self.rank and src == self.rank):
            recv[...] = send
            return
        out = torch.from_numpy(np.ascontiguousarray(send))
        inp = torch.from_numpy(recv)
        req = self.dist.isend(out, dst)
        self.dist.recv(inp, src)
        req.wait()

    def allreduce_sum_u8(self, buf):
        import torch
        t = torch.from_numpy(buf)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
