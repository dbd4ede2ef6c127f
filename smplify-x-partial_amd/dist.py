"""Multi-GPU: independent frames shard embarrassingly (the reference just loops over frames,
smplifyx/main.py:207).  One process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in CPU tests); rank r fits a contiguous block of frames with no
collective in the data path, then ONE all_gather of fixed-size per-frame result records
(fitted parameters + camera + final loss + evaluation count, ~1 KB/frame) -- SURVEY.md 8e.

Records are float64 rows: the frame index and the evaluation count are integers and stay exact
(float32 would stop at 2^24 frames), the fp32 parameters are widened losslessly.  Field widths follow
the result dict (num_betas, num_pca_comps, num_expression_coeffs are configuration), not constants.
"""
import numpy as np
import torch

FIELD_ORDER = ("cam_translation", "global_orient", "betas", "left_hand_pose", "right_hand_pose", "expression",
               "jaw_pose", "leye_pose", "reye_pose", "body_pose", "final_loss", "evals", "frame")
# widths of the default configuration (10 betas, 12 hand PCA components, 10 expression coefficients)
RECORD_FIELDS = (("cam_translation", 3), ("global_orient", 3), ("betas", 10), ("left_hand_pose", 12),
                 ("right_hand_pose", 12), ("expression", 10), ("jaw_pose", 3), ("leye_pose", 3),
                 ("reye_pose", 3), ("body_pose", 63), ("final_loss", 1), ("evals", 1), ("frame", 1))
RECORD_LEN = sum(n for _, n in RECORD_FIELDS)


def record_fields(res):
    """(name, width) of every record column for the result dict `res` of driver.fit_frames."""
    out = []
    for name in FIELD_ORDER:
        if name in ("final_loss", "evals", "frame"):
            out.append((name, 1))
        else:
            a = np.asarray(res[name])
            out.append((name, int(a.reshape(a.shape[0], -1).shape[1])))
    return tuple(out)


def shard_range(n_frames, rank, world):
    """Contiguous block of frame indices of `rank`: sizes differ by at most one."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_by_cost(cost, rank, world):
    """Frame indices of `rank` when the frames KNOWN to be long (driver.predicted_cost: side views are fitted twice, frames with
    an under-determined camera take 2-3 x the evaluations) are dealt round-robin instead of landing on one GPU with their
    contiguous block: frames in order of predicted cost (longest first, stable), dealt to the ranks back and forth (0 .. world-1,
    world-1 .. 0, ...) so that the rank that got the longest frame of a pass gets the shortest of the next.  Sizes differ by at
    most one; returned ascending, so a rank still walks its frames in input order."""
    order = np.argsort(-np.asarray(cost, np.float64), kind="stable")
    pos = np.arange(len(order))
    q = pos % (2 * world)
    owner = np.where(q < world, q, 2 * world - 1 - q)
    return np.sort(order[owner == rank])


def pack_records(res, first_frame, fields=None):
    """dict of [B,.] arrays (driver.fit_frames) -> float64 [B, record length]."""
    fields = record_fields(res) if fields is None else fields
    B = np.asarray(res["cam_translation"]).shape[0]
    cols = []
    for name, n in fields:
        if name == "evals":
            v = np.asarray(res["stage_evals"]).sum(1, keepdims=True)
        elif name == "frame":
            v = (first_frame + np.arange(B))[:, None]
        elif name == "final_loss":
            v = np.asarray(res["final_loss"])[:, None]
        else:
            v = res[name]
        cols.append(np.asarray(v, np.float64).reshape(B, n))
    return np.concatenate(cols, 1)


def unpack_records(rec, fields=RECORD_FIELDS):
    out, o = {}, 0
    for name, n in fields:
        out[name] = rec[:, o:o + n]
        o += n
    assert o == rec.shape[1], "record length %d does not match the field list (%d)" % (rec.shape[1], o)
    return out


def gather_records(rec, n_frames, device=None):
    """All ranks call with their [B_r, L] block; every rank gets [n_frames, L] in frame order.
    Blocks are padded to the largest shard so one all_gather suffices."""
    import torch.distributed as dist
    import os
    rec = np.asarray(rec, np.float64)
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size() == 1 and os.environ.get("SFX_FORCE_COLLECTIVE") != "1"):
        return rec                               # (SFX_FORCE_COLLECTIVE=1: a 1-GPU box walks the RCCL call as well)
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        device = None           # gloo gathers host tensors (CPU tests, single-GPU rehearsal of the N > 1 path)
    bmax = max(shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world))
    t = torch.zeros([bmax, rec.shape[1]], dtype=torch.float64, device=device)
    t[:rec.shape[0]] = torch.as_tensor(rec, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    parts = []
    for r in range(world):
        a, b = shard_range(n_frames, r, world)
        parts.append(outs[r][:b - a].cpu().numpy())
    return np.concatenate(parts, 0)
