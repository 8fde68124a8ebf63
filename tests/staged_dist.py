"""TEST INFRASTRUCTURE: a torch.distributed look-alike whose collectives stage device tensors through host memory, so
several ranks that share ONE GPU can talk over gloo (RCCL refuses two ranks on the same device; the gpurun box has one
GPU).  It lets dist.ShardedOps run world_size 2 with the real HIP stage implementation."""
import torch
import torch.distributed as dist


class StagedDist:
    ReduceOp = dist.ReduceOp

    @staticmethod
    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)

    @staticmethod
    def all_gather(outs, t, group=None):
        c = t.cpu()
        co = [torch.empty_like(c) for _ in outs]
        dist.all_gather(co, c, group=group)
        for o, x in zip(outs, co):
            o.copy_(x)

    @staticmethod
    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        ci = inp.cpu().contiguous()
        co = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(co, ci, output_split_sizes, input_split_sizes, group=group)
        out.copy_(co)

    @staticmethod
    def all_gather_object(objs, obj, group=None):
        dist.all_gather_object(objs, obj, group=group)

    @staticmethod
    def barrier(group=None):
        dist.barrier(group=group)

    # rank groups (dist.ShardedOps.session_groups) and the pairwise swap of their results
    new_group = staticmethod(dist.new_group)
    get_rank = staticmethod(dist.get_rank)

    @staticmethod
    def send(t, dst):
        dist.send(t.cpu().contiguous(), dst)

    @staticmethod
    def recv(t, src):
        c = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(c, src)
        t.copy_(c)

