"""The step loop bench.py times, as an importable object: a ring of device feature slots, SuperPoint on one HIP
stream, `len(pms)` PointMatching handles on their own streams taking batches in turn, the host one step ahead of
the GPU, and -- for N ranks -- the slot exchange of include/urf.h (urf_comm_*).

This is the caller-side shape of Tracking::ExtractFeatureAndMatch (src/tracking.cc:338-377) for a stream that
arrives in batches: step b = SuperPoint on the rank's `batch` frames of global batch b, then the pairs
(frame g - 1, frame g) whose second frame the rank owns.  tests/test_gpu_fullsize.py runs this very loop
against the CPU oracle; bench.py times it.

Exchange (world > 1): ONE communicator and ONE stream per rank, the same order on every rank:

    step b:   gather(b - M)  ->  all-gather(b)            M = len(pms)

* all-gather(b): waits (event) for SuperPoint(b); matcher stream b % M waits for it.  Runs beside SuperPoint(b+1)
  and the other matcher.
* gather(b - M): the match lists of the batch this rank's host fetched one step ago -- final by then (a resident
  Sinkhorn give-up or a near-tie redo rewrites the device lists inside urf_pm_fetch) and still in the buffers of
  matcher b % M, which is about to be reused: the matcher stream waits for the gather before match(b) starts.
  Rank 0 therefore holds every rank's lists of batch b - M after step b (`finish()` ships the last M batches).
"""
import numpy as np
import torch

from . import dist as D
from . import frontend as F
from . import _lib


MAX_BEGUN = 2      # begun batches a PointMatching handle holds (include/urf.h: urf_pm_fetch_begin)


class SlotRingPipeline:
    def __init__(self, sp, pms, d_frames, batch, H, W, *, device, rank=0, world=1, comm=None, gloo=False, overlap=2,
                 outlier_rejection=True, keep_gathered=False, sp_ahead=2, defer=3):
        """sp / pms: built SuperPoint / PointMatching handles on `device` (max_batch = max_pairs = batch).
        d_frames: u8 tensor [NB * batch, H, W] on the device, this rank's frames of NB consecutive global batches
        (cycled); NB >= len(pms) + 1 + sp_ahead.  comm: this rank's D.Comm (RCCL, world-of-one RCCL, or loopback) -- with it
        the slots go through the exchange even in a world of one; gloo=True: host-staged torch.distributed rig.
        sp_ahead: how many batches SuperPoint is enqueued ahead of the matcher (step b enqueues match(b) and SP(b + sp_ahead)).
        2 (default): the host waits for the lists of batch b - len(pms) + 1 with TWO batches of SuperPoint work queued, so a
        matcher that takes long over one batch (the strict mode's exact redo of a flagged pair: ~5 ms of dependent launches)
        does not drain SuperPoint's stream -- the critical path -- while the host waits; 1 = the round-3 loop.
        defer: how many steps the hand-out of a batch whose flagged pairs are being redone may lag (0: the host waits for
        the redo in the step that finds it).  3 since round 5 (2 and 3 measure the same with an engine per handle; a shared,
        merging engine -- urf_sg_config.redo_shared_engine / redo_merge -- starts a batch's redo one step later).  With the
        exchange the lists of batch g are shipped in step g + len(pms), which bounds the lag at 1."""
        self.sp, self.pms = sp, list(pms)
        self.B, self.H, self.W = int(batch), int(H), int(W)
        self.dev, self.rank, self.world = device, int(rank), int(world)
        self.d_frames = d_frames
        self.NB = d_frames.shape[0] // self.B
        self.overlap = overlap if len(self.pms) == 1 else 2
        self.ahead = int(sp_ahead) if self.overlap == 2 else 1
        self.defer = max(0, min(int(defer), 4))
        # ring slot k is refilled by SuperPoint(b + NB), enqueued in step b + NB - ahead; the last slot of batch b is read by
        # match(b + 1), which the host has fetched by step b + M: NB >= M + 1 + ahead
        assert self.NB * self.B == d_frames.shape[0] and self.NB >= len(self.pms) + 1 + self.ahead, "ring too short for the matchers"
        assert 1 <= self.ahead <= 3
        self.outlier = bool(outlier_rejection)
        self.comm, self.gloo = comm, bool(gloo)
        self.exchange = comm is not None or self.gloo
        sf = _lib.lib().urf_slot_bytes() // 4
        self.ring = torch.zeros((self.NB, self.B, sf), dtype=torch.float32, device=device)
        self.gathered = [None] * self.NB
        self.sp_calls = 0
        self.pending = []                 # [batch, matcher, fetch begun?] enqueued and not handed out yet, oldest first
        self._late = []                   # batches handed out outside step_compute (exchange: before the gather that ships them)
        self._record = None
        self.on_collect = None            # optional callback(batch index, matcher, results)
        self.keep_gathered = keep_gathered
        self.gather_log = {}              # rank 0, keep_gathered: batch -> (counts [world, B], matches [world, B, 1024] struct)
        self.gathered_matches_last = None
        if self.overlap == 0:
            for m in self.pms:
                m.share_stream(sp)
        if comm is not None:
            M = len(self.pms)
            self.sp_ext = torch.cuda.ExternalStream(sp.result_stream_ptr(), device=device)    # where SuperPoint's slots become final
            self.pm_ext = [torch.cuda.ExternalStream(m.stream_ptr(), device=device) for m in self.pms]
            self.cs = torch.cuda.Stream(device=device)
            self.gathered_buf = torch.zeros((self.NB, self.world * self.B, sf), dtype=torch.float32, device=device)
            # the root receives every rank's counts and 12-byte matches; one buffer set per matcher (batch b - M has left it
            # before batch b arrives: the gathers are in order on one stream)
            self.all_counts = [torch.zeros((self.world, self.B), dtype=torch.int32, device=device) for _ in range(M)]
            self.all_matches = [torch.zeros((self.world, self.B * 1024 * 3), dtype=torch.int32, device=device) for _ in range(M)]
            self.gathered_upto = -1       # last batch whose lists went to the root
            self.sp_ev = [None] * self.NB
        torch.cuda.synchronize(device)

    # ------------------------------------------------------------------ enqueue
    def sp_step(self, b):
        self.sp_calls = b + 1
        k = b % self.NB
        self.sp.infer_device(self.d_frames[k * self.B].data_ptr(), self.B, self.H, self.W, self.ring[k].data_ptr())
        # what waits for SP(b) is told so HERE, before any later batch is enqueued behind it on SuperPoint's stream: the
        # all-gather of batch b (exchange), or the matcher that will take batch b (its stream is in order: the wait sits
        # behind the batch it is still working on and in front of match(b))
        if self.comm is not None:
            ev = torch.cuda.Event()
            ev.record(self.sp_ext)
            self.sp_ev[k] = ev
        elif self.overlap == 2 and not self.gloo:
            self.pms[b % len(self.pms)].wait_for_sp(self.sp)

    def slots_of(self, b):
        k = b % self.NB
        return (self.ring[k], 0) if not self.exchange else (self.gathered[k], self.rank * self.B)

    def pair_slots(self, b):
        """(first, second) slot tensors of the pairs this rank owns in global batch b (urf_comm_plan_pairs)"""
        cur, base = self.slots_of(b)
        prev_all = self.slots_of(b - 1)[0] if b > 0 else None
        first, second = D.plan_pairs(self.world, self.rank, self.B)
        s0, s1 = [], []
        for f, s in zip(first, second):
            if f >= 0:
                s0.append(cur[f])
            elif prev_all is not None:
                s0.append(prev_all[-1])        # globally last frame of the previous batch (the carried slot)
            else:
                s0.append(cur[s])              # very first frame of the stream: matched with itself
            s1.append(cur[s])
        return s0, s1

    def pm_step(self, b, matcher):
        s0, s1 = self.pair_slots(b)
        matcher.match_device_async([t.data_ptr() for t in s0], [t.data_ptr() for t in s1], self.outlier)

    def _gather(self, g):
        """ship the (fetched, final) match lists of batch g to rank 0, on the exchange stream"""
        M = len(self.pms)
        mi = g % M
        d_m, d_n = self.pms[mi].device_results()
        self.comm.gather(d_n, self.B * 4, self.all_counts[mi].data_ptr(), 0, self.cs.cuda_stream)
        self.comm.gather(d_m, self.B * 1024 * 12, self.all_matches[mi].data_ptr(), 0, self.cs.cuda_stream)
        self.gathered_upto = g
        self._gather_done = (g, mi)

    def _after_gather(self):
        """bookkeeping once every rank has made the gather calls (loopback: the copies exist only then)"""
        if getattr(self, "_gather_done", None) is None:
            return
        g, mi = self._gather_done
        self._gather_done = None
        ev = torch.cuda.Event()
        ev.record(self.cs)
        self.pm_ext[mi].wait_event(ev)         # match(g + M) overwrites the buffers the gather reads
        if self.keep_gathered and self.rank == 0:
            self.cs.synchronize()
            cnt = self.all_counts[mi].cpu().numpy().copy()
            mt = self.all_matches[mi].cpu().numpy().view(F.MATCH_DTYPE).reshape(self.world, self.B, 1024).copy()
            self.gather_log[g] = (cnt, mt)

    def step_exchange(self, b):
        """the collectives of step b (nothing without an exchange).  All ranks make these calls in the same order."""
        if self.comm is not None:
            M = len(self.pms)
            if b - M > self.gathered_upto:
                while self.pending and self.pending[0][0] <= b - M and self.pending[0][2]:
                    self._late.append(self._hand_out(self._record))        # its redo has had a step's time: finish it, then ship
                assert all(e[0] != b - M for e in self.pending), "gather of a batch that was not fetched yet"
                self._gather(b - M)
            k = b % self.NB
            self.cs.wait_event(self.sp_ev[k])                           # SP(b), not whatever was enqueued behind it
            self.comm.allgather_slots(self.ring[k].data_ptr(), self.B, self.gathered_buf[k].data_ptr(), self.cs.cuda_stream)
            self.gathered[k] = self.gathered_buf[k]
        elif self.gloo:                                                 # test rig: host-staged, synchronous
            self.sp.sync()
            self.gathered[b % self.NB] = D.all_gather_slots(self.ring[b % self.NB], self.world)
            torch.cuda.synchronize(self.dev)

    def step_compute(self, b, record=None):
        """enqueue match(b) and SuperPoint(b + 1); fetch the oldest batch once len(pms) are in flight"""
        mt = self.pms[b % len(self.pms)]
        if self.comm is not None:
            self._after_gather()
            ev_ag = torch.cuda.Event()
            ev_ag.record(self.cs)
            self.pm_ext[b % len(self.pms)].wait_event(ev_ag)
        if self.overlap == 1 or (self.overlap == 2 and self.gloo):
            mt.wait_for_sp(self.sp)                 # match(b) needs SP(b)  (three-stream mode: told in sp_step)
        self.pm_step(b, mt)
        if self.overlap == 1:
            mt.let_sp_overlap_sinkhorn(self.sp)     # SP(b+1) starts when match(b) reaches Sinkhorn
        self.sp_step(b + self.ahead)
        self.pending.append([b, mt, False])
        self._record = record
        out, self._late = self._late, []
        # The host stays len(pms) - 1 batches behind the enqueue: every older batch has its fetch BEGUN (the host waits for
        # that batch's fast pass only; strict parity: the exact redo of its flagged pairs starts on the handle's redo engine,
        # beside the handle's next batches).  Batches are handed out in order, as soon as the head is final -- a head whose
        # redo still runs is waited for only once `defer` younger batches have queued up behind the len(pms) - 1 (a handle
        # holds at most two begun batches: MAX_BEGUN).
        M = len(self.pms)
        for e in self.pending[:len(self.pending) - (M - 1)]:
            if not e[2]:
                while sum(1 for q in self.pending if q[2] and q[1] is e[1]) >= MAX_BEGUN:
                    out.append(self._hand_out(record))
                e[1].fetch_begin(self.B)
                e[2] = True
        while self.pending and self.pending[0][2] and (len(self.pending) >= M + self.defer or self.pending[0][1].fetch_ready()):
            out.append(self._hand_out(record))
        return out

    def one_step(self, b, record=None):
        self.step_exchange(b)
        return self.step_compute(b, record)

    def _hand_out(self, record=None):
        """the oldest batch, whose fetch has begun: wait for its redo (if one is running) and deliver"""
        b, mt, begun = self.pending.pop(0)
        assert begun
        res = mt.fetch_end(self.B, as_arrays=True)
        if record is not None:
            record(b, mt, res)
        return b, res

    def collect(self, record=None):
        head = self.pending[0]
        if not head[2]:
            head[1].fetch_begin(self.B)             # waits (event) for that batch's fast pass only; starts the redo of flagged pairs
            head[2] = True
        return self._hand_out(record)

    def drain(self, record=None):
        out, self._late = self._late, []
        while self.pending:
            out.append(self.collect(record))
        return out

    def finish_exchange(self, last_b):
        """after drain(): ship the lists of the last len(pms) batches to rank 0 (call on every rank; lockstep callers use
        finish_gather / finish_after per batch instead)"""
        if self.comm is None:
            return
        for g in range(max(self.gathered_upto + 1, 0), last_b + 1):
            self._gather(g)
            self._after_gather()
        self.cs.synchronize()
        if self.rank == 0:
            self.gathered_matches_last = int(self.all_counts[last_b % len(self.pms)].sum().item())

    # ------------------------------------------------------------------ driving loops
    def prologue(self):
        for b in range(self.ahead):      # so that the loop body is exactly one match + one SuperPoint per step
            self.sp_step(b)

    def run(self, b0, steps, record=None):
        out = []
        for b in range(b0, b0 + steps):
            out += self.one_step(b, record)
        return out


def run_lockstep(pipes, b0, steps, record=None):
    """N logical ranks of ONE process (loopback communicators): every rank's collectives of a step first, then every rank's
    compute.  record(rank, b, matcher, results)."""
    out = [[] for _ in pipes]
    for b in range(b0, b0 + steps):
        for p in pipes:
            p.step_exchange(b)
        for r, p in enumerate(pipes):
            rec = (lambda bb, mt, res, r=r: record(r, bb, mt, res)) if record else None
            out[r] += p.step_compute(b, rec)
    return out


def finish_lockstep(pipes, last_b, record=None):
    out = [[] for _ in pipes]
    for r, p in enumerate(pipes):
        rec = (lambda bb, mt, res, r=r: record(r, bb, mt, res)) if record else None
        out[r] += p.drain(rec)
    g0 = max(pipes[0].gathered_upto + 1, 0)
    for g in range(g0, last_b + 1):
        for p in pipes:
            p._gather(g)
        for p in pipes:
            p._after_gather()
    for p in pipes:
        p.cs.synchronize()
    return out


def match_coords(res, feats_prev, feats_cur):
    """a match list as a set of pixel correspondences (x0, y0, x1, y1): the keypoint ORDER may differ between precision
    modes (score-sorted, near-equal scores swap), coordinates do not"""
    return {(feats_prev[q, 1], feats_prev[q, 2], feats_cur[t, 1], feats_cur[t, 2])
            for q, t in zip(res["queryIdx"], res["trainIdx"])}
