"""Sample partition of a render over several GPUs (one process per GPU).

Every (pixel, frame) pair owns an independent random stream and frames meet only in the accumulator sum
(pathtracer.cu:560,583-586), so ranks render disjoint frame ranges of the same scene and one reduce of
the fp32 accumulators onto rank 0 is the only exchange. The sum order differs from a single GPU's
frame order, so the result agrees with a one-GPU render to fp32 rounding (1e-4 relative), not bit for bit.
"""


def split_frames(total_frames, world, rank):
    """Strong split: rank r of `world` renders frames [r*total/world, (r+1)*total/world)."""
    if world < 1 or not 0 <= rank < world or total_frames < 0:
        raise ValueError('bad partition: %d frames, rank %d of %d' % (total_frames, rank, world))
    return rank * total_frames // world, (rank + 1) * total_frames // world


def step_frames(step, world, rank, frames_per_rank):
    """Weak split used by the benchmark: every rank renders `frames_per_rank` new frames each step."""
    if world < 1 or not 0 <= rank < world or frames_per_rank < 0 or step < 0:
        raise ValueError('bad partition')
    f0 = (step * world + rank) * frames_per_rank
    return f0, f0 + frames_per_rank


def reduce_accum(accum, dst=0):
    """Sums the per-rank accumulators onto rank `dst` (RCCL on GPUs, gloo in CPU tests); no-op for one rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
