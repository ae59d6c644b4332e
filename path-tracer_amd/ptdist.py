"""Multi-GPU partition of the render: one process per GPU, the scene and its tree replicated,
frames dealt out in blocks, the linear accumulation sums merged by ONE all-reduce at the end.

The path has no other exchange step: every (pixel, frame) sample depends only on the frame index
and that frame's seed, so ranks never talk while tracing (weak scaling: per-GPU work is fixed).
"""

FRAMES_PER_BLOCK = 128    # four full cycles of the shader's frame-index stratification
                          # (framecount & 1, (framecount >> 1) & 3, (framecount >> 3) & 3: period 32);
                          # one launch per block: the longer the launch, the smaller the share of its tail


def block_frames(step, rank, world_size, frames_per_block=FRAMES_PER_BLOCK):
    """First frame and count of the block that `rank` traces in its step number `step`.
    Blocks are dealt round-robin: block b = step * world_size + rank covers frames
    [b * frames_per_block, (b + 1) * frames_per_block)."""
    b = step * world_size + rank
    return b * frames_per_block, frames_per_block


def all_frames(steps, world_size, frames_per_block=FRAMES_PER_BLOCK):
    """The set of frames the whole job traces (for checks): 0 .. steps*world_size*fpb - 1."""
    return steps * world_size * frames_per_block


def reduce_linear_sum(tensor, dist=None):
    """Sum the per-rank linear accumulation images (float RGBA, alpha = sample count) in place.
    `dist` is torch.distributed (RCCL when the tensors are on GPUs, gloo in CPU tests)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
