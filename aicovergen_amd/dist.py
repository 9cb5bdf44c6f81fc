"""Chunk sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm and runs over xGMI).  The reference has no distributed code: both hot paths are embarrassingly
parallel over fixed-size work items, so the only collective is the join:

  * MDX: the flattened (segment, window) list is cut into `world` contiguous slices; each rank separates its slice
    and one all_gather of equal-size (per, 2, gen) blocks rebuilds the window list on every rank (30-min stereo
    track: 635 MB total, 79 MB per peer link -- bandwidth-trivial on 7 x 153 GB/s xGMI links);
  * RVC: `vc()` chunks are round-robined over ranks and joined by a length exchange + padded all_gather.  The cut search is
    computed redundantly on every rank (deterministic, a few milliseconds).  The f0 estimators are cut where they can be: CREPE's
    per-frame network over frame slices (crepe.predict), RMVPE's U-Net over time segments with an exact context margin
    (rmvpe.E2E.features_sharded), each joined by one all_gather; RMVPE's BiGRU is one sequential recurrence over the track that no
    rank can shorten, so every rank runs it in full (a broadcast from one rank would leave the others idle for the same time).

With world_size == 1 (or no initialised process group) every function degenerates to the single-GPU path, so the
same code is exercised by the 1-GPU tests.  `force_collectives` (AICG_FORCE_COLLECTIVES=1) removes that shortcut for a process group
of ONE rank: every join then goes through the real torch.distributed calls -- on a one-GPU box that is the only way to execute the
RCCL path (init with device_id, all_gather on device tensors) before an 8-GPU node exists (tests/test_rccl_one_rank.py,
tools/rccl_one_rank.py).  Same values either way: an all_gather over one rank is a copy.
"""
import os
import time

import torch
import torch.distributed as td


def world(group=None):
    """(rank, world size) of `group` (None = the default group).  The string "single" means: behave as a one-rank job whatever
    torch.distributed state the process has (a stand-alone f0 call outside VC.pipeline, see VC._f0_group)."""
    if isinstance(group, str):
        return 0, 1
    if td.is_available() and td.is_initialized():
        return td.get_rank(group), td.get_world_size(group)
    return 0, 1


force_collectives = os.environ.get("AICG_FORCE_COLLECTIVES") == "1"


def single(ws, group=None):
    """True when a join over `ws` ranks may skip its collective: one rank, unless `force_collectives` asks a real (initialised)
    one-rank group to run it anyway."""
    if ws != 1:
        return False
    return not (force_collectives and not isinstance(group, str) and td.is_available() and td.is_initialized())


# bench.py's instrumented step sets this: time every keyed collective with the device drained on both sides.  Off (the default, and in
# bench.py's TIMED steps): collectives are only counted -- a production step carries no device-wide drain for bookkeeping.
profile_joins = False


def _timed(key, fn, device_tensor):
    """Run a data-path collective.  With `profile_joins` its wall time (device drained on both sides) is added to last_join[key] for
    bench.py's per-rank split; otherwise it is only counted.  Only the multi-rank / forced paths come here.  key None: never timed --
    the f0 branch's joins are queued on a side stream under the HuBERT pass, and draining the device there would serialise the two
    branches."""
    last_join["collectives"] = last_join.get("collectives", 0) + 1
    if key is None or not profile_joins:
        return fn()
    cuda = device_tensor.is_cuda
    if cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    if cuda:
        torch.cuda.synchronize()
    last_join[key] = last_join.get(key, 0.0) + time.perf_counter() - t0
    return out


def all_gather_equal(block, group=None, key=None):
    """Every rank contributes an equal-shape block; returns the concatenation along dim 0 in rank order."""
    rank, ws = world(group)
    if single(ws, group):
        return block
    outs = [torch.empty_like(block) for _ in range(ws)]
    _timed(key, lambda: td.all_gather(outs, block.contiguous(), group=group), block)
    return torch.cat(outs, dim=0)


def mdx_separate(mdx_sess, wave, denoise, m_threads=2, group=None):
    """wave: (2, N) device tensor (normalised).  Returns the separated (2, N) device tensor on every rank."""
    rank, ws = world(group)
    part, meta = mdx_sess.separate(wave, denoise, m_threads, shard=(rank, ws))
    per = meta["per"]
    if part.shape[0] < per:  # the last rank may own fewer windows: pad to the common block size
        pad = torch.zeros((per - part.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        part = torch.cat([part, pad], 0)
    # bench.py's per-rank split: where a multi-GPU step's time goes (own windows done -> join done) is last_join["mdx_allgather_s"]
    allw = all_gather_equal(part, group, key="mdx_allgather_s")[: len(meta["jobs"])]
    return mdx_sess.join_windows(allw, meta)


# seconds this rank spent in the data-path collectives since the caller last cleared it, per join (bench.py reports them per rank:
# the collective alone, as opposed to waiting for the slowest rank, which shows up in the stage walls around it)
last_join = {}


def gather_pieces(pieces, total, device, group=None):
    """RVC join: `pieces` maps chunk index -> 1-D float32 device tensor for the chunks this rank converted
    (round-robin ownership: chunk i belongs to rank i % world).  Every rank gets all `total` pieces: lengths are
    exchanged first, then one all_gather of equal-size padded blocks."""
    rank, ws = world(group)
    if single(ws, group):
        return pieces
    rounds = (total + ws - 1) // ws
    lens = torch.zeros(rounds, dtype=torch.int64, device=device)
    for r in range(rounds):
        ci = r * ws + rank
        if ci in pieces:
            lens[r] = pieces[ci].numel()
    all_lens = [torch.empty_like(lens) for _ in range(ws)]
    _timed("rvc_lengths_allgather_s", lambda: td.all_gather(all_lens, lens, group=group), lens)
    maxlen = int(torch.stack(all_lens).max().item())
    block = torch.zeros((rounds, maxlen), dtype=torch.float32, device=device)
    for r in range(rounds):
        ci = r * ws + rank
        if ci in pieces:
            block[r, : pieces[ci].numel()] = pieces[ci]
    blocks = [torch.empty_like(block) for _ in range(ws)]
    _timed("rvc_pieces_allgather_s", lambda: td.all_gather(blocks, block, group=group), block)
    out = {}
    for rk in range(ws):
        ln = all_lens[rk].cpu().tolist()
        for r in range(rounds):
            ci = r * ws + rk
            if ci < total:
                out[ci] = blocks[rk][r, : int(ln[r])]
    return out
