"""Decode-sized calls served from captured HIP graphs — transparently.

The reference's decode table and layer benchmark are EAGER (README.md:300-310, benchmarks/layer_benchmark.py:131-143): a decoder layer's
one-token step is ~13 library launches of 3-30 us each, and from Python every launch costs 10-25 us of allocation, argument marshalling and
module plumbing — the step is host-bound (tools/bench_decode.py: 271 us eager against 73 us of kernels). The answer on this platform is a HIP
graph; this module supplies it WITHOUT caller-side graph code:

    deploy.fuse(model, capture=True)        # every decoder layer's forward is wrapped in a GraphedDecode

``GraphedDecode(fn)`` behaves like ``fn``. A call whose tensor arguments are decode-sized (<= ``max_rows`` rows) and which runs without
autograd is, per argument signature (shapes, dtypes, the other arguments by value or identity):
  1. run eagerly ``warmup`` times (default 2) — the library's caches (operand images, workspaces, launch plans) fill; the last of these runs
     on the wrapper's capture stream and RECORDS which host steps the paged KV caches among the arguments took (cache.update advances a length
     and rewrites its index tensors in place on the host: kv_cache.py ``_host_step``);
  2. captured: the tensor arguments are copied into static buffers, the caches' host steps are applied OUTSIDE the capture
     (``replay_host``), the callable runs once under ``torch.cuda.graph`` with the caches told to leave their host side alone;
  3. replayed from then on: copy the inputs in, apply the host steps, ``graph.replay()``, return the graph's output tensors.
Everything a step reads that changes between steps therefore sits in static device memory the replay sees: the inputs (copied), the cache's
pages (appended to in place by the captured launches through the index tensors), the index tensors (rewritten in place by the host step).
What makes an entry stale and sends the call back to the eager path (then re-captured): the cache's storage moved (page growth:
``cache.generation``), the library's caches were invalidated (``ops.cache_epoch``), a prefill-sized or masked (ragged) call, autograd on.

The returned tensors are the graph's OUTPUT BUFFERS: valid until the next call of the same wrapper with the same signature — a decoder
layer's output is consumed by the next layer within the step, which is the reference's usage; ``clone_outputs=True`` copies them out."""
import torch
from torch.utils import _pytree

from .. import ops

DECODE_ROWS = 128


def _is_cache(a) -> bool:
    return hasattr(a, "replay_host") and hasattr(a, "generation") and hasattr(a, "_needs_init")


def _walk_out(o, fn):
    if isinstance(o, torch.Tensor):
        return fn(o)
    if isinstance(o, (list, tuple)):
        return type(o)(_walk_out(v, fn) for v in o)
    if isinstance(o, dict):
        return {k: _walk_out(v, fn) for k, v in o.items()}
    return o


class _Entry:
    __slots__ = ("graph", "static_in", "out", "logs", "gens", "epoch", "pool_keepalive")


class GraphedDecode:
    MAX_SIGNATURES = 64     # captured graphs + signatures still in warm-up kept per wrapper (least recently created first out)

    def __init__(self, fn, max_rows: int = DECODE_ROWS, warmup: int = 2, clone_outputs: bool = False):
        self.fn, self.max_rows, self.warmup, self.clone_outputs = fn, max_rows, max(1, warmup), clone_outputs
        self._entries, self._seen, self._logs, self._blocked = {}, {}, {}, set()
        self._stream = None
        self.replays = self.captures = self.eager_calls = 0     # (counters: tests, tools/bench_decode.py)

    # -- argument handling ------------------------------------------------------------------------------------------------
    def _split(self, args, kwargs):
        """-> (leaves, spec, tensor positions, tensors, caches, signature) or None when the call is not one to capture. The arguments are
        flattened as a pytree, so tensors inside tuples / lists / dicts (HF's ``position_embeddings=(cos, sin)``) are graph inputs too."""
        leaves, spec = _pytree.tree_flatten((args, kwargs))
        pos, tensors, caches, sig = [], [], [], [spec]
        for i, v in enumerate(leaves):
            if isinstance(v, torch.Tensor):
                if not v.is_cuda or v.requires_grad:
                    return None
                pos.append(i), tensors.append(v)
                sig.append((tuple(v.shape), v.dtype, v.device.index))
            elif _is_cache(v):
                caches.append(v)
                sig.append(("cache", id(v)))
            elif v is None or isinstance(v, (bool, int, float, str)):
                sig.append(v)
            else:
                sig.append(("obj", id(v)))     # (a module, a config object, ...: by identity — its CONTENT must not change between steps)
        if not tensors:
            return None
        rows = 1
        for d in tensors[0].shape[:-1]:
            rows *= d
        if rows > self.max_rows or rows == 0:
            return None
        return leaves, spec, pos, tensors, caches, tuple(sig)

    @staticmethod
    def _subst(leaves, spec, pos, tensors):
        leaves = list(leaves)
        for i, t in zip(pos, tensors):
            leaves[i] = t
        return _pytree.tree_unflatten(leaves, spec)

    def _remember(self, sig, n):
        """warm-up counts per signature, bounded: a caller whose signature changes on every call (a growing attention mask, a fresh
        object per step) must not grow this table without limit — it simply stays eager."""
        if sig not in self._seen and len(self._seen) >= self.MAX_SIGNATURES:
            self._seen.pop(next(iter(self._seen)))
        self._seen[sig] = n

    # -- the call -----------------------------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        if torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return self.fn(*args, **kwargs)
        sp = self._split(args, kwargs)
        if sp is None:
            return self.fn(*args, **kwargs)
        leaves, spec, pos, tensors, caches, sig = sp
        if sig in self._blocked or any(any(c._needs_init) for c in caches):   # (capture failed once | a prompt has not been through this cache
            return self.fn(*args, **kwargs)                                    #  yet: the prefill branch of update())
        ent = self._entries.get(sig)
        if ent is not None:
            stale = ent.epoch != ops.cache_epoch() or any(c.generation != g for c, g in zip(caches, ent.gens)) \
                or any(c.would_grow(sum(a for _, a in log)) for c, log in zip(caches, ent.logs))
            if stale:
                del self._entries[sig]
                self._remember(sig, self.warmup - 1)         # one eager call (it re-allocates / re-warms), then a new capture
                ent = None
        if ent is None:
            n = self._seen.get(sig, 0)
            if n + 1 < self.warmup:
                self._remember(sig, n + 1)
                self.eager_calls += 1
                return self.fn(*args, **kwargs)
            if n + 1 == self.warmup:                          # the last warm-up: on the capture stream, recording the caches' host steps
                self._remember(sig, n + 1)
                self.eager_calls += 1
                return self._recorded_eager(args, kwargs, caches, sig)
            ent = self._capture(leaves, spec, pos, tensors, caches, sig)
            if isinstance(ent, tuple):                        # (the capture failed; the step ran eagerly inside _capture)
                self.eager_calls += 1
                return ent[1]
            if ent is None:
                self.eager_calls += 1
                return self.fn(*args, **kwargs)
            if len(self._entries) >= self.MAX_SIGNATURES:
                self._entries.pop(next(iter(self._entries)))
            self._entries[sig] = ent
        else:
            for s, t in zip(ent.static_in, tensors):
                s.copy_(t)
            for c, log in zip(caches, ent.logs):
                c.replay_host(log)
        ent.graph.replay()
        self.replays += 1
        return _walk_out(ent.out, lambda t: t.clone()) if self.clone_outputs else ent.out

    def _recorded_eager(self, args, kwargs, caches, sig):
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        for c in caches:
            c._log = []
        try:
            with torch.cuda.stream(self._stream):
                out = self.fn(*args, **kwargs)
        finally:
            logs = [c._log for c in caches]
            for c in caches:
                c._log = None
        cur.wait_stream(self._stream)
        if sig not in self._logs and len(self._logs) >= self.MAX_SIGNATURES:
            self._logs.pop(next(iter(self._logs)))
        self._logs[sig] = logs
        return out

    def _capture(self, leaves, spec, pos, tensors, caches, sig):
        logs = self._logs.pop(sig, None)
        if logs is None or any(step is None for log in logs for step in log):
            return None                                       # (a masked / prefill update: not a replayable step)
        if any(c.would_grow(sum(a for _, a in log)) for c, log in zip(caches, logs)):
            return None
        ent = _Entry()
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            ent.static_in = [t.clone() for t in tensors]
        sargs, skwargs = self._subst(leaves, spec, pos, ent.static_in)
        for c, log in zip(caches, logs):
            c.replay_host(log)                                # the host side of THIS step, outside the capture (stream-ordered in front of it)
        self._stream.wait_stream(cur)
        ops.images_ready()                                    # fragment images prepared by the warm-up are complete: shared, not re-prepared in the graph
        ent.graph = torch.cuda.CUDAGraph()
        for c in caches:
            c._skip_host = True
        try:
            with torch.cuda.graph(ent.graph, stream=self._stream):
                ent.out = self.fn(*sargs, **skwargs)
        except Exception:
            # The callable cannot be captured (an op that synchronises, allocates on the host, ...). The caches' host steps of THIS call have
            # been applied: run the step eagerly with the caches still told to leave their host side alone, never try this signature again.
            self._blocked.add(sig)
            torch.cuda.synchronize()
            return ("eager", self.fn(*sargs, **skwargs))
        finally:
            for c in caches:
                c._skip_host = False
        cur.wait_stream(self._stream)
        ent.logs, ent.gens, ent.epoch = logs, [c.generation for c in caches], ops.cache_epoch()
        self.captures += 1
        return ent

    def release(self):
        """Drop every captured graph (and its private memory pool)."""
        self._entries.clear(), self._seen.clear(), self._logs.clear(), self._blocked.clear()
