"""Single-node multi-GPU context for the row-sharded embedding path (SURVEY.md §8e — new
functionality; the reference has no multi-GPU training, CHANGELOG.md:5).

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).  Per step:
  all_to_all  ids  -> owners          (int32, fixed capacity per peer, one fused exchange per id matrix)
  all_to_all  rows -> requesters      (fp32 [cap, D] per peer)
  all_to_all  row gradients -> owners (backward)
  all_reduce  dense gradients (one flat buffer) + one scalar (table part of the clip norm)
The gloo path (CPU tensors, or GPU tensors staged through the host) exists so that the routing logic
is testable with 2 processes in a GPU-less container and on a 1-GPU box.
"""
import torch
import torch.distributed as dist


class DistContext(object):
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("shard='row' needs torch.distributed to be initialised "
                               "(launch with torch.distributed.run)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def _stage(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_to_all(self, send):
        """send: [world * k, ...] -> recv of the same shape (chunk i goes to rank i)."""
        if self._stage(send):
            s = send.cpu()
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=self.group)
            return r.to(send.device)
        if self.backend == "gloo":
            # gloo implements all_to_all_single for CPU tensors via send/recv pairs
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send.contiguous(), group=self.group)
            return recv
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send.contiguous(), group=self.group)
        return recv

    def all_reduce_sum(self, t):
        if self._stage(t):
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast(self, t, src=0):
        if self._stage(t):
            c = t.cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)
            return t
        dist.broadcast(t, src, group=self.group)
        return t
