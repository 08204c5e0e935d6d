"""Multi-GPU: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU tests).

Two ways the path shards:
  * K-fold level -- independent units, zero data-path exchange: fold k runs on rank k mod world; only the per-fold
    result records are gathered at the end (`gather_fold_results`).  Reproduces single-GPU results exactly.
  * patient/batch level (DDP): each rank steps on its shard of the global batch; the flat gradient buffer is
    all-reduced between backward and the clip+Adam kernels: one blocking collective (`allreduce_mean_`), or -- the default of
    the staged step -- SUM in 5 buckets (`gradient_buckets`: heads, then the DenseNet121 backward stages 3..0), each launched
    asynchronously as soon as its stage has been enqueued (`allreduce_ranges_async`) so that it overlaps the rest of the backward; the
    update waits for all of them and divides by the world size.  (The RCCL branch has not run on hardware yet: 1-GPU leases; the gloo
    CPU tests take the same async code path.)
    Cox risk set: rank-local (default), or GLOBAL over the world*B patients of the step (`train_step(..., ddp_world=N,
    global_cox=True)`).  BatchNorm: rank-local, or `sync_bn=True` = exact global-batch semantics (engine._ddp_step_syncbn).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """-> (world, rank, local device index).  Backend: argument, else $MMS_DIST_BACKEND, else nccl (= RCCL) when a GPU is
    present.  `gloo` lets several ranks share one card (rehearsals on a 1-GPU box); the local index wraps accordingly."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("MMS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)
    return world, rank, local


def folds_of_rank(n_folds, world, rank):
    """fold k -> rank k mod world."""
    return [k for k in range(n_folds) if k % world == rank]


def gather_fold_results(local_results, world):
    """local_results: list of dicts (each with a 'fold' key) -> all folds' records, ordered by fold, on every rank."""
    if world <= 1:
        return sorted(local_results, key=lambda r: r["fold"])
    out = [None] * world
    dist.all_gather_object(out, local_results)
    return sorted([r for part in out for r in part], key=lambda r: r["fold"])


def allreduce_sum_(flat, world):
    """In-place sum over ranks (gradients of a loss that is already global: each rank holds its samples' share)."""
    if world > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_gather_into(out, local, world):
    """out[r * n:(r + 1) * n] = rank r's `local` (n elements, same on every rank): the (hazard, time, event, valid) exchange
    of the global Cox risk set -- 4 collectives of world * B floats."""
    if world <= 1:
        out.copy_(local.reshape(-1))
        return out
    if local.is_cuda and dist.get_backend() == "gloo":
        parts = [torch.empty(local.numel()) for _ in range(world)]
        dist.all_gather(parts, local.detach().reshape(-1).cpu())
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, local.detach().reshape(-1).contiguous())
    return out


def allreduce_mean_(flat, world):
    """In-place mean of a flat gradient buffer over ranks (one bucket: the buffer is already contiguous)."""
    if world > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":      # rehearsal path: gloo reduces through host memory
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
    return flat


def allreduce_ranges_async(flat, ranges, world):
    """SUM over ranks of the slices flat[a:a+n] of one gradient bucket, launched asynchronously: RCCL runs the collective on its
    own stream behind the work already enqueued on the current stream, so it overlaps whatever the caller enqueues next.
    -> a callable that makes the CURRENT stream wait for the bucket.  (gloo rehearsal path: synchronous, through host memory.)"""
    if world <= 1:
        return lambda: None
    if flat.is_cuda and dist.get_backend() == "gloo":
        for a, n in ranges:
            h = flat[a:a + n].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat[a:a + n].copy_(h)
        return lambda: None
    works = [dist.all_reduce(flat[a:a + n], op=dist.ReduceOp.SUM, async_op=True) for a, n in ranges]

    def wait():
        for w in works:
            w.wait()
    return wait


ENC_STAGE_CUTS = (0, 39, 114, 261, 364)     # csrc/dn_net.hip Idx: parameter-table index where dense block b's backward stage begins (b = 0..3)


def gradient_buckets(params, encoder_params, staged):
    """Gradient buckets of the data-parallel step in the order the backward finalises them: the heads, then (staged = DenseNet121
    encoder) its backward stages 3, 2, 1, 0 -- each a list of contiguous (offset, length) ranges of the flat gradient buffer, in which
    the parameters lie in `params` order.  -> [[(offset, length), ...], ...]; together the ranges cover the buffer exactly once."""
    offs, o = {}, 0
    for q in params:
        offs[id(q)] = (o, q.numel())
        o += q.numel()

    def ranges(ids):
        out = []
        for a, n in sorted(offs[i] for i in ids):
            if out and out[-1][0] + out[-1][1] == a:
                out[-1] = (out[-1][0], out[-1][1] + n)
            else:
                out.append((a, n))
        return out
    eids = [id(q) for q in encoder_params]
    eset = set(eids)
    hids = [id(q) for q in params if id(q) not in eset]
    if staged:
        c = ENC_STAGE_CUTS
        if len(eids) != c[-1]:
            raise ValueError("staged buckets are defined for the DenseNet121 encoder (%d parameter tensors), got %d" % (c[-1], len(eids)))
        return [ranges(hids)] + [ranges(eids[c[b]:c[b + 1]]) for b in (3, 2, 1, 0)]
    return [ranges(hids + eids)]


def max_over_ranks(x, device):
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
