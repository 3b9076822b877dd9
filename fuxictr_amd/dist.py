"""Single-node multi-GPU context for the row-sharded embedding path (SURVEY.md §8e — new
functionality; the reference has no multi-GPU training, CHANGELOG.md:5).

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).  Per step:
  all_to_all  ids  -> owners          (int32, fixed capacity per peer, one fused exchange per id matrix)
  all_to_all  rows -> requesters      (fp32 [cap, D] per peer)
  all_to_all  row gradients -> owners (backward)
  all_reduce  dense gradients (one flat buffer) + one scalar (table part of the clip norm)
The gloo path (CPU tensors, or GPU tensors staged through the host) exists so that the routing logic
is testable with 2 processes in a GPU-less container and on a 1-GPU box.
"""
import os
import time

import torch
import torch.distributed as dist


class GraphSegments(object):
    """A training step with collectives, as hipGraph segments with the collectives launched eagerly
    in between: [graph 0] a2a ids [graph 1] a2a rows ... [graph k] all-reduce [graph k+1].
    Nothing about RCCL is captured; every buffer a collective touches was allocated inside the
    capture (one memory pool shared by all segments), so its address is fixed across replays.

    WHILE RECORDING, the eager collectives run on their own `comm` stream, never on the stream that
    is being captured: ProcessGroupNCCL's watchdog thread polls the completion event of every
    collective it has not reaped yet (every ~100 ms), and HIP refuses a query of an event whose
    stream is capturing (hipErrorCapturedEvent) — that kills the watchdog (and with it the process)
    and invalidates the capture.  For the same reason `settle()` lets the watchdog reap everything
    launched so far before a capture begins.  Replays launch the collectives on the replay stream
    itself (nothing is capturing then; the two stream hops per collective cost ~25 us each)."""

    def __init__(self, comm_stream=None, settle_s=0.0):
        self.pool = torch.cuda.graph_pool_handle()
        self.items = []          # CUDAGraph | callable, in program order
        self._cur = None
        self.comm = comm_stream
        self.settle_s = settle_s

    def settle(self):
        if self.settle_s > 0:
            torch.cuda.synchronize()
            time.sleep(self.settle_s)

    def _eager(self, fn):
        if self.comm is None:
            fn()
            return
        cur = torch.cuda.current_stream()
        self.comm.wait_stream(cur)
        with torch.cuda.stream(self.comm):
            fn()
        cur.wait_stream(self.comm)

    def begin(self):
        self.settle()
        g = torch.cuda.CUDAGraph()
        # thread_local: API calls of OTHER threads must not invalidate the capture; the autograd
        # thread's launches still land in it because capture is a property of the stream
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._cur = g
        delay = float(os.environ.get("FX_SEG_CAPTURE_DELAY", "0"))     # debug: widen the window
        if delay > 0:
            time.sleep(delay)

    def cut(self, fn):
        """End the current segment, run `fn` eagerly (now, and at this point of every replay),
        start the next segment."""
        self._cur.capture_end()
        self.items.append(self._cur)
        self._eager(fn)
        self.items.append(fn)
        self.begin()

    def finish(self):
        self._cur.capture_end()
        self.items.append(self._cur)
        self._cur = None

    def abort(self):
        """End a capture that failed half way (best effort) and drop what was recorded."""
        if self._cur is not None:
            try:
                self._cur.capture_end()
            except Exception:   # noqa: BLE001 — the capture is already invalid
                pass
            self._cur = None
        self.items = []

    def replay(self):
        if os.environ.get("FX_SEG_DEBUG"):
            for i, it in enumerate(self.items):
                print("[seg %d] %s" % (i, type(it).__name__), flush=True)
                it.replay() if isinstance(it, torch.cuda.CUDAGraph) else it()
                torch.cuda.synchronize()
            return
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


class DistContext(object):
    TEARDOWN_EXIT = 75   # exit status of shutdown()'s watchdog (EX_TEMPFAIL): the teardown hung
    recorder = None      # a GraphSegments while a sharded step is being captured
    graph_mode = None    # how the captured step launches its collectives (set by the capture)

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("shard='row' needs torch.distributed to be initialised "
                               "(launch with torch.distributed.run)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        # RCCL: the collectives are recorded INTO the step's hipGraph — one graph launch per step, no
        # launch gaps around the four collectives (round 4 default; FX_GRAPH_COLLECTIVES=0 or a failed
        # capture falls back to hipGraph segments with the collectives launched eagerly between them).
        # A communicator must outlive no graph that recorded its kernels: see release_graphs().
        self.capture_collectives = (self.backend == "nccl"
                                    and os.environ.get("FX_GRAPH_COLLECTIVES", "1") != "0")

    def _stage(self, t):
        return self.backend == "gloo" and t.is_cuda

    def all_to_all(self, send, recv=None):
        """send: [world * k, ...] -> recv of the same shape (chunk i goes to rank i); `recv`: a
        contiguous buffer to receive into (e.g. the leading rows of a larger block)."""
        send = send.contiguous()
        if recv is None:
            recv = torch.empty_like(send)
        assert recv.is_contiguous() and recv.shape == send.shape
        if (self.recorder is not None and self.capture_collectives
                and os.environ.get("FX_TEST_CAPTURE_FAIL") == "1"):
            # test hook: a stack that cannot record RCCL kernels (tests/test_gpu_dist.py checks that the
            # step then falls back to hipGraph segments by itself)
            raise RuntimeError("FX_TEST_CAPTURE_FAIL: simulated failure to record a collective")
        if self.recorder is not None and not self.capture_collectives:
            self.recorder.cut(lambda: self._a2a_into(recv, send))
        else:
            self._a2a_into(recv, send)
        return recv

    def _a2a_into(self, recv, send):
        if self._stage(send):      # gloo on GPU tensors: through the host (test / debug only)
            s = send.cpu()
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=self.group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, group=self.group)

    def all_reduce_sum(self, t):
        if self.recorder is not None and not self.capture_collectives:
            self.recorder.cut(lambda: self._all_reduce_into(t))
        else:
            self._all_reduce_into(t)
        return t

    def _all_reduce_into(self, t):
        if self._stage(t):
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather_cat(self, t):
        """1-D tensors of ANY length per rank -> their concatenation in rank order, on every rank
        (evaluation: every rank scores the global validation set, so every rank takes the same
        lr-decay / early-stop decision)."""
        t = t.reshape(-1).contiguous()
        stage = self._stage(t)
        src = t.cpu() if stage else t
        if self.backend == "nccl" and not src.is_cuda:      # host-side metric vectors over RCCL
            return self.all_gather_cat(src.cuda()).cpu()
        n = torch.tensor([src.numel()], dtype=torch.int64, device=src.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(k.item()) for k in sizes]
        pad = torch.zeros(max(max(sizes), 1), dtype=src.dtype, device=src.device)
        pad[:src.numel()] = src
        outs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(outs, pad, group=self.group)
        out = torch.cat([o[:k] for o, k in zip(outs, sizes)])
        return out.to(t.device) if stage else out

    def require_same(self, value, what):
        """Every forward of a row-sharded model issues collectives, so all ranks must run the same
        number of them: raise on every rank (instead of deadlocking) when `value` differs."""
        dev = "cuda" if self.backend == "nccl" else "cpu"
        v = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        vals = [torch.zeros_like(v) for _ in range(self.world)]
        dist.all_gather(vals, v, group=self.group)
        vals = [int(k.item()) for k in vals]
        if len(set(vals)) != 1:
            raise RuntimeError("row-sharded training needs the same %s on every rank, got %s "
                               "(per rank)" % (what, vals))

    @staticmethod
    def shutdown(models=(), timeout_s=30.0):
        """Leave the process group cleanly after collectives were recorded into hipGraphs: RCCL's
        communicator teardown waits until every graph that captured its kernels is destroyed
        (ncclCommDestroy polls the graphs' references), so the captured steps go first — then
        destroy_process_group().  A watchdog ends the process if the teardown still does not return
        (the line a benchmark printed is already out)."""
        import gc
        import sys
        import threading
        for m in models:
            if hasattr(m, "release_graphs"):
                m.release_graphs()
        from . import layers
        layers._SHARD_PEERS.clear()
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if not dist.is_initialized():
            return

        def _bail():
            # a stuck communicator teardown is a failure, and it must read as one: distinct non-zero exit
            # status, after saying so on stderr
            sys.stderr.write("[fuxictr_amd.dist] destroy_process_group() did not return within %.0f s: "
                             "leaving the process with exit status %d\n" % (timeout_s, DistContext.TEARDOWN_EXIT))
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(DistContext.TEARDOWN_EXIT)
        timer = threading.Timer(timeout_s, _bail)
        timer.daemon = True
        timer.start()
        dist.destroy_process_group()
        timer.cancel()

    def broadcast(self, t, src=0):
        if self._stage(t):
            c = t.cpu()
            dist.broadcast(c, src, group=self.group)
            t.copy_(c)
            return t
        dist.broadcast(t, src, group=self.group)
        return t
