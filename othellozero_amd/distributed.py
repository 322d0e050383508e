"""Multi-GPU self-play: shard games over ranks, pool move records with ONE all-gather.

Replaces the reference's only "parallelism": WorkerManager splitting episodes over workers and
concatenating their result lists (workers.py:168-184,298-303) and the ssh/pickle return path
(workers.py:147-159).  Games are independent, so the data path has no collective; the only exchange
is the all-gather of compact 48-byte move records at the end of a self-play batch (RCCL over xGMI
with backend "nccl", gloo on CPU).  The 8-fold symmetry expansion happens AFTER the gather
(training.expand_examples), so ~18x fewer bytes cross the links.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import RECORD_DTYPE

RECORD_BYTES = RECORD_DTYPE.itemsize


def shard_games(total_games, rank, world_size):
    """contiguous block of global game ids for `rank` (first ranks take the remainder): (first_id, count)"""
    base, rem = divmod(total_games, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def records_to_tensor(records, device="cpu"):
    rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    t = torch.from_numpy(rec.view(np.uint8).reshape(-1, RECORD_BYTES).copy())
    return t.to(device)


def tensor_to_records(t):
    a = t.detach().cpu().contiguous().numpy().reshape(-1)
    return a.view(RECORD_DTYPE).copy()


def gather_records(local, group=None, single_rank_collective=False):
    """all-gather variable-length record tensors (uint8 [R_i, W]; W = 48 for move records) -> uint8 [sum R_i, W] on every rank,
    ordered by rank.  Counts are gathered first, payloads are padded to the per-rank maximum.
    A one-rank group returns `local` without a collective unless single_rank_collective is set (the one-GPU test of
    the RCCL calls themselves)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size(group) == 1 and not single_rank_collective:
        return local
    world = dist.get_world_size(group)
    dev = local.device
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if mx == 0:
        return local
    width = int(local.shape[1])
    padded = torch.zeros((mx, width), dtype=torch.uint8, device=dev)
    padded[: local.shape[0]] = local
    out = torch.empty((world * mx, width), dtype=torch.uint8, device=dev)
    if dev.type == "cuda" and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(out, padded, group=group)
    else:
        _all_gather_list(out, padded, world, group)
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)


def _all_gather_list(out, padded, world, group):
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    for r, p in enumerate(parts):
        out[r * padded.shape[0]: (r + 1) * padded.shape[0]] = p


def engine_records_tensor(engine, device):
    """move records of the engine's completed games as a uint8 [R, 48] tensor on `device`, copied device to
    device from the engine's HBM buffer (no host round trip)."""
    total = engine.stats()["records"]
    t = torch.empty((max(total, 1), RECORD_BYTES), dtype=torch.uint8, device=device)
    written = engine.records_to_device(t.data_ptr(), total) if total else 0
    return t[:written]


def pooled_selfplay_records(engine, device, group=None):
    """records of every rank's completed games, identical on all ranks, sorted by (game_id, ply)"""
    allrec = tensor_to_records(gather_records(engine_records_tensor(engine, device), group))
    return allrec[np.lexsort((allrec["ply"], allrec["game_id"]))]


# ---------------------------------------------------------------- arena games sharded over the ranks (all-gather of (winner, points))
ARENA_RESULT_DTYPE = np.dtype([("game_id", "<i4"), ("points", "<i4"), ("n_moves", "<i4"), ("winner", "i1"), ("pad", "u1", 3)])


def arena_sharded(net_a, net_b, board_size=8, total_games=512, num_simulations=800, degree_exploration=1.0, seed=0, device="cpu",
                  group=None, **arena_kwargs):
    """agents.arena_batch with the games sharded over the ranks (contiguous blocks of global game ids, shard_games) and ONE all-gather of
    the per-game results -- 16 bytes per game: id, winner (+1 = BLACK's agent), points, moves -- so that every rank holds the whole match,
    sorted by game id.  Replaces the reference's fan-out of duels over workers and the concatenation of their result lists
    (workers.py:168-184, main.py:196-214).  A game's moves depend on its global id only (RNG streams are keyed by it), never on the
    rank that played it: the pooled result equals one arena_batch of all the games.  `device`: where the gathered tensor lives
    ("cuda" for the nccl = RCCL backend, "cpu" for gloo); extra keyword arguments go to arena_batch."""
    from .agents import arena_batch
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    first, count = shard_games(total_games, rank, world)
    rows = np.zeros(count, dtype=ARENA_RESULT_DTYPE)
    failed = None
    if count:
        try:
            res = arena_batch(net_a, net_b, board_size, count, num_simulations, degree_exploration, seed=seed, first_game_id=first, **arena_kwargs)
            rows["game_id"] = first + np.arange(count)
            rows["winner"], rows["points"], rows["n_moves"] = res["winner"], res["points"], res["n_moves"]
        except Exception as e:                                   # still join the collectives below: the peers must not wait for a rank that left
            failed = e
    if world > 1:
        # one status word first (max over ranks): a rank whose games raised ends the match on EVERY rank, before anybody enters the gather
        status = torch.tensor([1 if failed is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
        if int(status.item()):
            raise failed if failed is not None else RuntimeError("arena_sharded: another rank's games raised; the match is abandoned on every rank")
    elif failed is not None:
        raise failed
    t = torch.from_numpy(rows.view(np.uint8).reshape(-1, ARENA_RESULT_DTYPE.itemsize).copy()).to(device)
    pooled = gather_records(t, group).detach().cpu().contiguous().numpy().reshape(-1).view(ARENA_RESULT_DTYPE)
    pooled = pooled[np.argsort(pooled["game_id"], kind="stable")]
    return dict(game_id=pooled["game_id"].copy(), winner=pooled["winner"].copy(), points=pooled["points"].copy(), n_moves=pooled["n_moves"].copy())


# ---------------------------------------------------------------- the same exchange step behind the C ABI (no torch on the data path)
class Comm:
    """RCCL communicator owned by the library (oz_comm_*): what a non-Python host uses.  Rank 0 makes the 128-byte id, the host hands it
    to every rank (here: any callable `share(id_bytes_or_None) -> id_bytes`, e.g. a torch.distributed broadcast), every rank creates
    its communicator on its own device -- the device of the CALLING THREAD (oz_set_device / the HIP runtime's current device are
    per host thread: a helper thread must select the rank's GPU before it creates the communicator)."""

    def __init__(self, rank, world, share=None):
        import ctypes as C
        from . import _lib
        lib = _lib.require_gpu()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(lib.oz_comm_unique_id(ident))
        raw = bytes(ident)
        if world > 1:
            assert share is not None, "a multi-rank communicator needs a way to hand rank 0's id to the other ranks"
            raw = share(raw if rank == 0 else None)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        _lib.check(lib.oz_comm_create(C.byref(self._h), ident, rank, world))

    def gather_records(self, engine, first_record=0, max_records=None):
        """COLLECTIVE: the records [first_record, ...) of every rank's engine in rank order -> (records, per-rank counts).
        Two collective calls: the counts alone (out = NULL), then the records into a buffer sized from the pooled count -- the same
        number on every rank, so no rank can come up short while its peers wait in the payload all-gather.  max_records, if
        given, is this rank's room: the library fails the call on every rank together when it is too small on any."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        written, per = C.c_int64(), np.zeros(self.world, np.int64)
        if max_records is None:
            _lib.check(lib.oz_selfplay_gather_records(engine._h, self._h, int(first_record), None, 0, C.byref(written), _lib.p_i64(per)))
            cap = max(int(written.value), 1)
        else:
            cap = int(max_records)
        out = np.zeros(cap, dtype=RECORD_DTYPE)
        _lib.check(lib.oz_selfplay_gather_records(engine._h, self._h, int(first_record), out.ctypes.data_as(C.c_void_p), cap,
                                                  C.byref(written), _lib.p_i64(per)))
        return out[:written.value], per

    def close(self):
        if self._h:
            from . import _lib
            _lib.load().oz_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_share(device):
    """`share` for Comm over an initialised torch.distributed group: broadcast of the 128 id bytes from rank 0"""
    def share(raw):
        t = torch.zeros(128, dtype=torch.uint8, device=device if dist.get_backend() != "gloo" else "cpu")
        if raw is not None:
            t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return share


# ---------------------------------------------------------------- data-parallel training
class GradientAllReduce:
    """Averages a Trainer's gradient arena across ranks between backward and apply: ONE all-reduce per optimiser step
    over the whole flat arena (16 M floats = 64 MB at 8x8 / 512 channels -- a single bucket keeps the xGMI ring at its
    per-link bandwidth instead of paying per-tensor latency 28 times).  The arena is a torch tensor OWNED HERE and handed
    to the library as `external_grads`, so RCCL reduces the memory the backward kernels wrote, with no copy.

        ar = GradientAllReduce(board_size, channels, in_channels, device)     # before the Trainer
        tr = Trainer(..., external_grads_ptr=ar.ptr)
        fit(tr, ..., allreduce=ar)

    With the gloo backend (CPU rehearsal of the control flow) the arena is staged through host memory."""

    def __init__(self, board_size, channels, in_channels=2, device="cuda", group=None, single_rank_collective=False):
        from .trainer import Trainer
        self.group = group
        self.single_rank_collective = single_rank_collective    # run the all-reduce on a one-rank group too (one-GPU RCCL test)
        # one element more than the library's arena: the STATUS word of the step (0 = this rank's step is valid, 1 = it failed) rides in the
        # same all-reduce -- its sum is the number of failed ranks -- instead of overloading a gradient element (a NaN in
        # element 0 that a genuinely diverged step produced read as "another rank failed")
        self.size = Trainer.arena_size(board_size, channels, in_channels)
        self.flat = torch.zeros(self.size + 1, dtype=torch.float32, device=device)
        self.ptr = self.flat.data_ptr()

    def __call__(self, trainer, failed=None):
        """failed: the exception this rank's forward / backward raised (None: the step is valid).  A failed rank still takes part in the
        all-reduce, with its status word set; afterwards every rank reads the number of failed ranks and raises if it is not zero, so a
        step that is invalid on one rank (the f16x2 range guard, a bad batch, an out-of-memory) ends the job on ALL ranks instead of leaving
        the others blocked in the collective."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.group) == 1 and not self.single_rank_collective):
            if failed is not None:
                raise failed
            return
        world = dist.get_world_size(self.group)
        if failed is None:
            trainer.sync()                               # the library's stream wrote the arena
            self.flat[self.size] = 0.0
        else:
            self.flat[: self.size].zero_()               # whatever the failed step left there must not reach the peers' sums as NaN / Inf
            self.flat[self.size] = 1.0
        if dist.get_backend(self.group) == "gloo" and self.flat.is_cuda:
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        bad = int(round(float(self.flat[self.size].item())))
        self.flat[: self.size].div_(world)
        if self.flat.is_cuda:
            torch.cuda.synchronize(self.flat.device)     # torch's stream -> before the library's Adam kernel reads it
        if bad:
            from ._lib import OZ_ERR_STATE, OzError
            raise failed if failed is not None else OzError(OZ_ERR_STATE, f"the training step was invalid on {bad} other rank(s) (their forward / backward "
                                                                          "raised): the step is not applied on any rank")


def average_moving_statistics(weights, group=None):
    """BN moving statistics are per-replica under data parallelism (each rank normalises with its own batch); average
    them at the end of a fit so that every rank holds the same inference network.  `weights` = the 40 get_weights()
    arrays; returns the list with indices 6l+4, 6l+5 averaged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return weights
    world = dist.get_world_size(group)
    idx = [i for i in range(36) if i % 6 in (4, 5)]
    flat = torch.from_numpy(np.concatenate([np.asarray(weights[i], dtype=np.float32).ravel() for i in idx]))
    if dist.get_backend(group) != "gloo":
        flat = flat.cuda()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = (flat / world).cpu().numpy()
    out, pos = list(weights), 0
    for i in idx:
        n = np.asarray(weights[i]).size
        out[i] = flat[pos:pos + n].reshape(np.asarray(weights[i]).shape).copy()
        pos += n
    return out


def any_rank(flag, group=None):
    """True on every rank when `flag` is true on at least one (a MAX all-reduce of one integer; `flag` itself without a process group)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if dist.get_backend(group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))
