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
