"""CPU test of the N>1 path: world_size-2 gloo processes shard frames, build result records and
all_gather them; the gathered table must be identical to the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _fake_fit(first, last):
    """Stand-in for driver.fit_frames on CPU: deterministic per-frame 'results'."""
    idx = np.arange(first, last)
    rng = lambda i, n: np.random.RandomState(1000 + i).normal(size=n).astype(np.float32)
    res = dict(cam_translation=np.stack([rng(i, 3) for i in idx]), global_orient=np.stack([rng(i + 1, 3) for i in idx]),
               betas=np.stack([rng(i + 2, 10) for i in idx]), left_hand_pose=np.zeros((len(idx), 12), np.float32),
               right_hand_pose=np.zeros((len(idx), 12), np.float32), expression=np.zeros((len(idx), 10), np.float32),
               jaw_pose=np.zeros((len(idx), 3), np.float32), leye_pose=np.zeros((len(idx), 3), np.float32),
               reye_pose=np.zeros((len(idx), 3), np.float32), body_pose=np.stack([rng(i + 3, 63) for i in idx]),
               final_loss=idx.astype(np.float32) * 1.5, stage_evals=np.tile(idx[:, None], (1, 4)).astype(np.int32))
    return res


def _worker(rank, world, port, n_frames, out_dir):
    import torch.distributed as dist
    from smplifyx_amd import dist as sd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sd.shard_range(n_frames, rank, world)
    rec = sd.pack_records(_fake_fit(a, b), a)
    full = sd.gather_records(rec, n_frames)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    from smplifyx_amd import dist as sd
    n = 7                                   # ragged: 4 + 3
    assert [sd.shard_range(n, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [sd.shard_range(8192, r, 8) for r in range(8)][3] == (3072, 4096)
    # the cost-ordered deal: every frame to exactly one rank, sizes within one, the long frames spread evenly
    cost = np.ones(37); cost[[3, 4, 5, 6, 30, 31]] = 2.5; cost[[10, 11]] = 5.0
    shares = [sd.shard_by_cost(cost, r, 4) for r in range(4)]
    assert sorted(np.concatenate(shares).tolist()) == list(range(37))
    assert max(len(s_) for s_ in shares) - min(len(s_) for s_ in shares) <= 1 and all(np.all(np.diff(s_) > 0) for s_ in shares)
    loads = [cost[s_].sum() for s_ in shares]
    contiguous = [cost[slice(*sd.shard_range(37, r, 4))].sum() for r in range(4)]
    assert max(loads) < max(contiguous) and max(loads) - min(loads) < max(contiguous) - min(contiguous), (loads, contiguous)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    want = sd.pack_records(_fake_fit(0, n), 0)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert got.shape == (n, sd.RECORD_LEN)
        assert np.array_equal(got, want)
    u = sd.unpack_records(want)
    assert np.array_equal(u["frame"][:, 0], np.arange(n)) and u["body_pose"].shape == (n, 63)


def test_main_deals_the_long_frames_over_the_ranks():
    """main.rank_share: every frame to exactly one rank, input order kept within a rank, and the side views (fitted twice) of a
    sequence that holds them in one stretch spread over the ranks instead of landing on one."""
    from smplifyx_amd import main as M
    cfg = dict(format="coco25", confidence_threshold=0.2, init_joints_idxs=[9, 12, 2, 5], side_view_thsh=25.0,
               left_shoulder_idx=2, right_shoulder_idx=5)
    rng = np.random.RandomState(0)
    items = []
    for i in range(40):
        kp = np.concatenate([rng.uniform(100, 500, size=(25, 2)), rng.uniform(0.5, 1.0, size=(25, 1))], 1).astype(np.float32)
        if 8 <= i < 16:
            kp[5, :2] = kp[2, :2] + 3.0                    # shoulders 3 px apart: a side view
        items.append(dict(fn="f%02d" % i, keypoints=kp))
    jw = np.ones(25, np.float32)
    shares = [M.rank_share(items, cfg, jw, r, 4) for r in range(4)]
    names = [[it["fn"] for it in s_] for s_ in shares]
    assert sorted(sum(names, [])) == sorted(it["fn"] for it in items)
    assert all(n == sorted(n) for n in names) and max(map(len, names)) - min(map(len, names)) <= 1
    side = [sum(1 for n in s_ if 8 <= int(n[1:]) < 16) for s_ in names]
    assert side == [2, 2, 2, 2], side                      # (contiguous blocks of ten: [2, 6, 0, 0])
    assert M.rank_share(items, cfg, jw, 0, 1) is items
