"""Multitemporal sequence driver: the data-parallel replacement of the reference's sequential epoch loop
(`main_dev.py:60`: `for ep in cfg.proc.epoch_to_process`, with a fresh matcher per epoch `main_dev.py:115-132`,
i.e. no state is carried between epochs).

Epochs (stereo pairs) are sharded over ranks, one process per GPU; every rank matches its own pairs with its own
`Engine` and writes one fixed-size *match-table record* per pair into a device tensor. The only communication of
the whole job is one all-gather of the per-rank tables at the end (RCCL over xGMI when the backend is "nccl";
"gloo" on CPU for the tests), after which every rank holds the complete per-epoch match table.

Record layout (int32 words):
    [0] epoch  [1] n0  [2] n1  [3] n_matches (-1 = pair failed)  [4] stop layer  [5..7] reserved
    [8 : 8+K]       matches0 (int32, -1 = unmatched)
    [8+K : 8+2K]    matching_scores0 (float32 bit patterns)
    [8+2K : 8+6K]   optional payload (`with_keypoints`): keypoints0, keypoints1 as float32 [K][2] (x, y) bit patterns - the 98 KB
                    record of SURVEY §8d config 4, with which a rank that holds the gathered table can rebuild matched point pairs
                    of every epoch without the extracting rank's buffers
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

HEADER = 8
_DEBUG_GUARDS = __import__("os").environ.get("IM_DEBUG_GUARDS") == "1"
_FORCE_COLLECTIVE = __import__("os").environ.get("IM_BENCH_FORCE_DIST") == "1"   # rehearsal: run the all-gather at world size 1 too


def record_words(max_kpts: int, with_keypoints: bool = False) -> int:
    return HEADER + (6 if with_keypoints else 2) * max_kpts


def shard_epochs(n_epochs: int, rank: int, world: int) -> List[int]:
    """Epochs owned by `rank`: e = rank (mod world). Round-robin keeps the ranks' work equal to within one
    pair and every rank's share spread over the whole sequence."""
    return list(range(rank, n_epochs, world))


def new_table(n_rows: int, max_kpts: int, device, with_keypoints: bool = False) -> torch.Tensor:
    t = torch.full((n_rows, record_words(max_kpts, with_keypoints)), -1, dtype=torch.int32, device=device)
    t[:, HEADER + max_kpts:] = 0
    return t


def mark_failed(table: torch.Tensor, row: int, epoch: int, max_kpts: int) -> None:
    """The record of a pair that could not be matched: epoch, n0 = n1 = 0, n_matches = -1, no matches (`pending_epochs` hands such
    epochs to a resumed run). The reference logs the error and goes on with the next epoch (`matchers.py:199-207`,
    `main_dev.py:270-274, 295-301`); in a sharded run the rank must in addition reach the table all-gather, or every other rank
    waits in it. Fill kernels only: enqueued on the current stream like the record of a pair that worked."""
    r = table[row]
    r[0:1].fill_(int(epoch))
    r[1:HEADER].zero_()
    r[3:4].fill_(-1)
    r[HEADER:HEADER + max_kpts].fill_(-1)
    r[HEADER + max_kpts:].zero_()


def write_records(table: torch.Tensor, row: int, first_epoch: int, n_pairs: int, engine) -> None:
    """Rows row .. row + n_pairs - 1 of `table` from the outputs of the engine's last `lightglue(n_pairs=...)`: one kernel."""
    from ._lib import ptr
    K = engine.max_kpts
    if table.shape[1] not in (HEADER + 2 * K, HEADER + 6 * K):
        raise ValueError(f"match table rows are {table.shape[1]} words wide; records of this engine (K = {K}) are {HEADER + 2 * K} "
                         f"or, with keypoints, {HEADER + 6 * K}")
    assert table.is_contiguous() and row + n_pairs <= table.shape[0]
    kpts = ptr(engine.kpts) if table.shape[1] == HEADER + 6 * K else None     # keypoint payload: engine.kpts [2 P][K][2]
    engine.ctx.call("im_pack_records", int(n_pairs), ptr(engine.n), ptr(engine.matches), ptr(engine.mscores), ptr(engine.info),
                    int(first_epoch), table[row].data_ptr(), kpts, engine.stream_ptr())


def write_record(table: torch.Tensor, row: int, epoch: int, n: torch.Tensor, matches0: torch.Tensor,
                 mscores0: torch.Tensor, info: torch.Tensor, engine=None) -> None:
    """Device-side (no host sync): fill one row from the engine's output buffers. With an engine the row is packed
    by one library kernel (`im_pack_record`); the torch indexing path is the CPU twin used by the gloo tests."""
    K = matches0.shape[0]
    if engine is not None and table.is_cuda:
        from ._lib import ptr
        assert table.shape[1] == HEADER + 2 * K and table.is_contiguous()
        engine.ctx.call("im_pack_record", ptr(n), ptr(matches0), ptr(mscores0), ptr(info), int(epoch),
                        table[row].data_ptr(), engine.stream_ptr())
        return
    r = table[row]
    r[0:1].fill_(epoch)   # a fill kernel (a scalar assignment would be a host->device copy: not capturable)
    r[1:3] = n[:2]
    r[3] = (matches0 > -1).sum().to(torch.int32)
    r[4] = info.reshape(-1)[0]
    r[HEADER:HEADER + K] = matches0
    r[HEADER + K:HEADER + 2 * K] = mscores0.view(torch.int32)


def all_gather_tables(local: torch.Tensor, group=None) -> torch.Tensor:
    """One all-gather of the fixed-size per-rank tables; returns the global table sorted by epoch
    (rows of ranks that own fewer epochs are padded with epoch = -1 and dropped)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVE):
        out = local
    else:
        world = dist.get_world_size(group)
        rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_rows = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(all_rows, rows, group=group)
        max_rows = int(max(int(r.item()) for r in all_rows))
        if local.shape[0] < max_rows:
            pad = torch.full((max_rows - local.shape[0], local.shape[1]), -1, dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], 0)
        out = torch.empty((world * max_rows, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    out = out[out[:, 0] >= 0]
    return out[torch.argsort(out[:, 0])]


def decode_record(rec: np.ndarray, max_kpts: int) -> dict:
    rec = np.ascontiguousarray(rec)
    K = max_kpts
    n0, n1 = max(int(rec[1]), 0), max(int(rec[2]), 0)
    out = dict(epoch=int(rec[0]), n0=int(rec[1]), n1=int(rec[2]), n_matches=int(rec[3]), stop=int(rec[4]),
               matches0=rec[HEADER:HEADER + K][:n0].astype(np.int64),
               matching_scores0=rec[HEADER + K:HEADER + 2 * K].view(np.float32)[:n0])
    if len(rec) == HEADER + 6 * K:
        kp = rec[HEADER + 2 * K:].view(np.float32).reshape(2, K, 2)
        out["keypoints0"], out["keypoints1"] = kp[0, :n0], kp[1, :n1]
    return out


def save_table(path: str, table: torch.Tensor, max_kpts: int) -> None:
    """Checkpoint of a (partial) match table: the resumable artefact of a sharded run (the reference pickles the whole
    `Epoch` per epoch, `core/epoch.py:455-473`, driven by `cfg.proc.load_existing_results`, `main_dev.py:70-92`)."""
    t = table.detach().cpu().numpy()
    t = t[t[:, 0] >= 0]
    np.savez_compressed(path, records=t[np.argsort(t[:, 0], kind="stable")], max_kpts=np.int64(max_kpts))


def load_table(path: str):
    """-> (records int32 [n, 8 + 2K] sorted by epoch, K). `pending_epochs` tells a resumed run what is left."""
    z = np.load(path)
    return z["records"], int(z["max_kpts"])


def pending_epochs(n_epochs: int, done_records: Optional[np.ndarray]) -> List[int]:
    """Epochs without a successful record (n_matches >= 0) yet."""
    done = set()
    if done_records is not None and len(done_records):
        done = {int(r[0]) for r in done_records if r[3] >= 0}
    return [e for e in range(n_epochs) if e not in done]


def records_to_features(rec: np.ndarray, max_kpts: int, kpts0: np.ndarray, kpts1: np.ndarray):
    """Matched point arrays (mkpts0, mkpts1, mconf) of one record, the inputs of
    `core.Features.append_features_from_numpy` / `sfm.RelativeOrientation` (`main_dev.py:160-172, 220-226`)."""
    r = decode_record(rec, max_kpts)
    v = r["matches0"] > -1
    return kpts0[: r["n0"]][v], kpts1[r["matches0"][v]], r["matching_scores0"][v]


class SequenceMatcher:
    """SuperPoint + LightGlue (or SuperGlue, `matcher="superglue"`) over a list of stereo pairs on one GPU, results kept on
    the device.

    A pair is ~200 kernel launches, none of which needs the host (counts, early stop and pruning are device state),
    so the whole pair is captured once into a HIP graph and replayed per epoch: the input pair is copied into a
    static device buffer, the graph runs, the record row is copied out. `use_graph=False` enqueues the launches
    directly (same results bit for bit).

    `pairs_per_launch = P > 1` (LightGlue): P pairs share every launch (a batch dimension over pairs inside the kernels,
    `im_lightglue_forward_pairs`): `match_pair` collects pairs and runs the group when it is full; `flush()` runs a partial
    group. Records are bit-identical to P = 1."""

    def __init__(self, engine, height: int, width: int, max_keypoints: int = 4096, nms_radius: int = 4,
                 detection_threshold: float = 0.0005, remove_borders: int = 4, depth_confidence: float = 0.95,
                 width_confidence: float = 0.99, filter_threshold: float = 0.1, use_graph: bool = True,
                 matcher: str = "lightglue", sinkhorn_iterations: int = 20, match_threshold: float = 0.3,
                 pruning_min_kpts: int = -1, channels: int = 1, pairs_per_launch: int = 1, with_keypoints: bool = False):
        self.e = engine
        self.h, self.w, self.k = height, width, max_keypoints
        self.matcher = matcher
        if matcher == "superglue":   # icepy4d's SuperGlue defaults (`matchers.py:854-867`)
            nms_radius = 3 if nms_radius == 4 else nms_radius
            detection_threshold = 0.001 if detection_threshold == 0.0005 else detection_threshold
            if pairs_per_launch != 1:
                raise NotImplementedError("pairs_per_launch > 1 is implemented for the LightGlue matcher")
        elif matcher != "lightglue":
            raise ValueError(f"unknown matcher {matcher!r}")
        self.P = int(pairs_per_launch)
        self.sp = (nms_radius, detection_threshold, remove_borders)
        self.lg = dict(depth_confidence=depth_confidence, width_confidence=width_confidence, filter_threshold=filter_threshold,
                       pruning_min_kpts=pruning_min_kpts)
        self.sg = dict(sinkhorn_iterations=sinkhorn_iterations, match_threshold=match_threshold)
        engine.reserve(height, width, 2 * self.P, max_keypoints)
        self.use_graph = use_graph
        self._graph = None
        shape = (2 * self.P, height, width) if channels == 1 else (2 * self.P, height, width, 3)
        self._inp = torch.zeros(shape, dtype=torch.uint8, device=engine.device)
        self._rec = new_table(self.P, engine.max_kpts, engine.device, with_keypoints)
        self._pending = []           # (epoch, table, row) of the pairs waiting in self._inp
        self.failed = []             # (epoch, message) of the pairs whose record says n_matches = -1
        self._pinned = None          # host feeder: ring of page-locked staging buffers (match_host_pair)
        self._pin_i = 0

    def _enqueue(self, pairs_u8: torch.Tensor) -> None:
        e = self.e
        if self.matcher == "superglue":
            e.superpoint(pairs_u8, self.sp[0], self.sp[1], self.sp[2], self.k, flavour=1)
            e.superglue((self.h, self.w), (self.h, self.w), **self.sg)
            return
        e.superpoint(pairs_u8, self.sp[0], self.sp[1], self.sp[2], self.k)
        e.lightglue((self.w, self.h), (self.w, self.h), n_pairs=pairs_u8.shape[0] // 2, **self.lg)

    def _record(self, table: torch.Tensor, row: int, epoch: int, n_pairs: int) -> None:
        write_records(table, row, epoch, n_pairs, self.e)

    def _capture(self) -> None:
        dev = self.e.device
        with torch.cuda.device(dev):       # the capture stream torch creates belongs to the CURRENT device: make that the engine's
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # warm-up outside capture: lazy kernel attributes, allocator pools
                for _ in range(2):
                    self._enqueue(self._inp)
                    self._record(self._rec, 0, 0, self.P)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue(self._inp)
                self._record(self._rec, 0, 0, self.P)
        self._graph = g

    def _fail(self, epoch: int, table: torch.Tensor, row: int, exc: BaseException) -> None:
        """Failure isolation (the reference logs and continues, `matchers.py:199-207`, `main_dev.py:270-274`): the pair gets
        the record `mark_failed` writes and the sequence goes on. If even that cannot be enqueued the device is gone: the error
        propagates to the caller, who owns the process group (bench.py: abort, exit non-zero)."""
        import logging
        logging.getLogger("icepy4d_amd").error("epoch %d failed: %s: %s", epoch, type(exc).__name__, exc)
        self.failed.append((int(epoch), f"{type(exc).__name__}: {exc}"))
        mark_failed(table, row, epoch, self.e.max_kpts)

    def _run_group(self) -> None:
        pend, self._pending = self._pending, []
        if not pend:
            return
        for epoch, table, row in pend:     # a caller's programming error, not a pair's failure: raised, before anything runs
            if table.shape[1] != self._rec.shape[1]:
                raise ValueError(f"match table rows are {table.shape[1]} words wide, this matcher writes records of {self._rec.shape[1]} "
                                 f"(with_keypoints={self._rec.shape[1] == record_words(self.e.max_kpts, True)}): build the table with "
                                 "new_table(..., with_keypoints=) to match the matcher's")
        try:
            if not self.use_graph or len(pend) < self.P:
                # direct launches; also for a partially filled group (the tail of a sequence whose length is not a multiple of the
                # pairs per launch): only the parked pairs are computed, the captured graph always runs all P
                self._enqueue(self._inp[:2 * len(pend)] if self.P > 1 else self._inp)
                self._record(self._rec, 0, 0, len(pend))
            else:
                if self._graph is None:
                    self._capture()
                self._graph.replay()
                if _DEBUG_GUARDS:       # forwards inside a graph carry no guard check of their own (csrc/weights.hip): from the host
                    self.e.ctx.call("im_debug_guards_check", self.e.stream_ptr())
        except Exception as group_exc:      # noqa: BLE001 - any error of a launch group is isolated to the pairs that cause it
            # the group as a whole could not be enqueued: its pairs one by one with direct launches (their inputs are still parked in
            # self._inp), so that only the pairs that fail by themselves are marked. The group's own error is logged whatever the retries
            # give (it vanished when they succeeded); a failed graph capture is not tried again for every later group; and an
            # error that is not a pair's (a sticky HIP error, out of device memory: the retries would fail the same way, or worse, half work)
            # goes to the caller, who owns the process group.
            import logging
            log = logging.getLogger("icepy4d_amd")
            log.error("launch group of %d pair(s) (epochs %s) failed as a whole: %s: %s - retrying its pairs one by one", len(pend),
                      [int(e) for e, _, _ in pend], type(group_exc).__name__, group_exc)
            if self.use_graph and self._graph is None:
                self.use_graph = False
                log.error("HIP graph capture failed: this matcher goes on with direct launches")
            msg = str(group_exc).lower()
            if isinstance(group_exc, (MemoryError, torch.cuda.OutOfMemoryError)) or "out of memory" in msg or "out of device memory" in msg \
                    or "hiperror" in msg.replace(" ", "") or "illegal" in msg or "device-side" in msg:
                raise
            for j, (epoch, table, row) in enumerate(pend):
                try:
                    if len(pend) == 1:
                        raise group_exc
                    self._enqueue(self._inp[2 * j:2 * j + 2])
                    self._record(self._rec, 0, 0, 1)
                    table[row].copy_(self._rec[0], non_blocking=True)
                    table[row, 0:1].fill_(epoch)
                except Exception as exc:    # noqa: BLE001
                    self._fail(epoch, table, row, exc)
            return
        for j, (epoch, table, row) in enumerate(pend):
            table[row].copy_(self._rec[j], non_blocking=True)
            table[row, 0:1].fill_(epoch)

    def match_pair(self, pair_u8: torch.Tensor, epoch: int, table: torch.Tensor, row: int) -> None:
        """pair_u8: device uint8 [2, H, W]. Enqueues the pair (or, with pairs_per_launch > 1, parks it until its group is
        full) and its record; never synchronises."""
        if self.P == 1 and not self.use_graph:
            try:
                self._enqueue(pair_u8)
                self._record(table, row, epoch, 1)
            except Exception as exc:        # noqa: BLE001
                self._fail(epoch, table, row, exc)
            return
        if self.use_graph and self._graph is None:
            self._capture()
        j = len(self._pending)
        try:
            self._inp[2 * j:2 * j + 2].copy_(pair_u8, non_blocking=True)
        except Exception as exc:            # noqa: BLE001 - e.g. an image of another shape or type than the sequence's: this pair only
            self._fail(epoch, table, row, exc)
            return
        self._pending.append((epoch, table, row))
        if len(self._pending) == self.P:
            self._run_group()

    def match_host_pair(self, pair_u8: np.ndarray, epoch: int, table: torch.Tensor, row: int) -> None:
        """The same for a pair in HOST memory (what the reference's loop has after `cv2.imread`, `core/images.py:44-93` ->
        `main_dev.py:115-132`): numpy uint8 [2, H, W] (or [2, H, W, 3]). The pair is copied into a page-locked staging buffer
        (a ring of 2 P + 2 buffers, so that the host never rewrites a buffer whose upload may still be in flight) and uploaded
        with an asynchronous copy on the CURRENT stream, in front of the launches that read it: no host synchronisation. On ONE
        stream the copy is ordered behind the launches already enqueued there (it does not overlap them); overlap of uploads and
        kernels comes from `PairPipeline`'s other slots, whose streams run kernels while this slot's copy is in flight. Before a
        staging buffer is reused its last upload is waited for through an event - by then two launch groups old."""
        if self._pinned is None:
            shape = tuple(self._inp.shape[1:])
            self._pinned = [[torch.empty((2,) + shape, dtype=torch.uint8).pin_memory(), None] for _ in range(2 * self.P + 2)]
        slot = self._pinned[self._pin_i]
        self._pin_i = (self._pin_i + 1) % len(self._pinned)
        if slot[1] is not None:
            slot[1].synchronize()
        try:
            np.copyto(slot[0].numpy(), pair_u8)       # an unreadable / wrongly shaped image fails here: this pair only
        except Exception as exc:            # noqa: BLE001
            self._fail(epoch, table, row, exc)
            return
        if self.use_graph and self._graph is None:
            self._capture()
        j = len(self._pending) if (self.P > 1 or self.use_graph) else 0
        self._inp[2 * j:2 * j + 2].copy_(slot[0], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.e.device))
        if self.P == 1 and not self.use_graph:
            try:
                self._enqueue(self._inp)
                self._record(table, row, epoch, 1)
            except Exception as exc:        # noqa: BLE001
                self._fail(epoch, table, row, exc)
            return
        self._pending.append((epoch, table, row))
        if len(self._pending) == self.P:
            self._run_group()

    def flush(self) -> None:
        """Run a partially filled group (direct launches for just the parked pairs)."""
        self._run_group()

    def run(self, pairs: Sequence[torch.Tensor], epochs: Sequence[int], table: Optional[torch.Tensor] = None) -> torch.Tensor:
        if table is None:
            table = new_table(len(epochs), self.e.max_kpts, self.e.device)
        for row, (p, ep) in enumerate(zip(pairs, epochs)):
            self.match_pair(p, ep, table, row)
        self.flush()
        return table


class PairPipeline:
    """`n_streams` independent (engine, HIP stream, captured graph) slots on one GPU; pairs are dealt round-robin.

    One 1080p / 4096-keypoint pair does not fill the chip at every stage (the attention launches are exactly one
    4-wave block per CU, the small GEMMs and the selection kernels much less), so two pairs in flight on separate
    streams overlap one pair's latency-bound stages with the other's MFMA-bound ones. Each slot owns its context and
    workspace; results are written to disjoint rows of the caller's table. Launches are made on side streams, never
    on the legacy null stream (graph launches there serialise against every other stream)."""

    def __init__(self, make_engine, height: int, width: int, max_keypoints: int = 4096, n_streams: int = 2,
                 use_graph: bool = True, pairs_per_launch: int = 1, **matcher_conf):
        self.slots = []
        for _ in range(max(1, n_streams)):
            eng = make_engine()
            stream = torch.cuda.Stream(device=eng.device)
            with torch.cuda.stream(stream):
                sm = SequenceMatcher(eng, height, width, max_keypoints, use_graph=use_graph, pairs_per_launch=pairs_per_launch,
                                     **matcher_conf)
                if use_graph:
                    sm._capture()  # capture now, while nothing else is running on the device
            stream.synchronize()
            self.slots.append((eng, stream, sm))
        self._next = 0
        self.device = self.slots[0][0].device
        self.max_kpts = self.slots[0][0].max_kpts

    def match_pair(self, pair_u8: torch.Tensor, epoch: int, table: torch.Tensor, row: int) -> None:
        eng, stream, sm = self.slots[self._next]
        with torch.cuda.stream(stream):
            sm.match_pair(pair_u8, epoch, table, row)
        if not sm._pending:                       # the slot's launch group went out: the next pair goes to the next slot
            self._next = (self._next + 1) % len(self.slots)

    def match_host_pair(self, pair_u8: np.ndarray, epoch: int, table: torch.Tensor, row: int) -> None:
        """A pair in host memory (numpy uint8): staged through page-locked buffers and uploaded asynchronously on the slot's
        stream (`SequenceMatcher.match_host_pair`); never synchronises the device."""
        eng, stream, sm = self.slots[self._next]
        with torch.cuda.stream(stream):
            sm.match_host_pair(pair_u8, epoch, table, row)
        if not sm._pending:
            self._next = (self._next + 1) % len(self.slots)

    @property
    def failed(self):
        """(epoch, message) of every pair that got a `mark_failed` record, over all slots."""
        return sorted(f for _, _, sm in self.slots for f in sm.failed)

    def flush(self) -> None:
        """Enqueue whatever is still waiting for a full launch group (nothing with one pair per launch)."""
        for eng, stream, sm in self.slots:
            with torch.cuda.stream(stream):
                sm.flush()

    def synchronize(self) -> None:
        for _, stream, _ in self.slots:
            stream.synchronize()

    def close(self) -> None:
        for eng, _, _ in self.slots:
            eng.close()
