"""Multi-GPU: independent frames shard embarrassingly (the reference just loops over frames,
smplifyx/main.py:207).  One process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in CPU tests); rank r fits a contiguous block of frames with no
collective in the data path, then ONE all_gather of fixed-size per-frame result records
(fitted parameters + camera + final loss + evaluation count, ~0.8 KB/frame) -- SURVEY.md 8e.
"""
import numpy as np
import torch

RECORD_FIELDS = (("cam_translation", 3), ("global_orient", 3), ("betas", 10), ("left_hand_pose", 12),
                 ("right_hand_pose", 12), ("expression", 10), ("jaw_pose", 3), ("leye_pose", 3),
                 ("reye_pose", 3), ("body_pose", 63), ("final_loss", 1), ("evals", 1), ("frame", 1))
RECORD_LEN = sum(n for _, n in RECORD_FIELDS)


def shard_range(n_frames, rank, world):
    """Contiguous block of frame indices of `rank`: sizes differ by at most one."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_records(res, first_frame):
    """dict of [B,.] arrays (driver.fit_frames) -> float32 [B, RECORD_LEN]."""
    B = res["cam_translation"].shape[0]
    cols = []
    for name, n in RECORD_FIELDS:
        if name == "evals":
            v = res["stage_evals"].sum(1, keepdims=True)
        elif name == "frame":
            v = (first_frame + np.arange(B))[:, None]
        elif name == "final_loss":
            v = res["final_loss"][:, None]
        else:
            v = res[name]
        cols.append(np.asarray(v, np.float32).reshape(B, n))
    return np.concatenate(cols, 1)


def unpack_records(rec):
    out, o = {}, 0
    for name, n in RECORD_FIELDS:
        out[name] = rec[:, o:o + n]
        o += n
    return out


def gather_records(rec, n_frames, device=None):
    """All ranks call with their [B_r, RECORD_LEN] block; every rank gets [n_frames, RECORD_LEN]
    in frame order.  Blocks are padded to the largest shard so one all_gather suffices."""
    import torch.distributed as dist
    import os
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size() == 1 and os.environ.get("SFX_FORCE_COLLECTIVE") != "1"):
        return np.asarray(rec, np.float32)         # (SFX_FORCE_COLLECTIVE=1: a 1-GPU box walks the RCCL call as well)
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        device = None           # gloo gathers host tensors (CPU tests, single-GPU rehearsal of the N > 1 path)
    bmax = max(shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world))
    t = torch.zeros([bmax, RECORD_LEN], dtype=torch.float32, device=device)
    t[:rec.shape[0]] = torch.as_tensor(np.asarray(rec, np.float32), device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    parts = []
    for r in range(world):
        a, b = shard_range(n_frames, r, world)
        parts.append(outs[r][:b - a].cpu().numpy())
    return np.concatenate(parts, 0)
