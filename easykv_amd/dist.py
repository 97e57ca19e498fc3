"""One process per GPU: layer-block ownership and the pipeline hand-off.

Eviction state and decisions are per (layer, KV head), so a layer's K/V rows, slot map and score rows stay on
the GPU that owns the layer; the only thing that ever crosses xGMI is the stage output ``[1, q, hidden]`` fp16
(8 KB per decode token for Llama2-7B) between neighbouring stages — a point-to-point send/recv over one link,
no all-reduce / all-gather anywhere on the path.  ``torch.distributed`` backend ``nccl`` is RCCL on ROCm; the
same code runs on ``gloo`` for the CPU tests.
"""
from __future__ import annotations

import os
import collections
import time
import weakref
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class LayerShard:
    rank: int
    world: int
    n_layers: int

    @property
    def begin(self) -> int:
        return (self.n_layers * self.rank) // self.world

    @property
    def end(self) -> int:
        return (self.n_layers * (self.rank + 1)) // self.world

    @property
    def count(self) -> int:
        return self.end - self.begin

    @property
    def next_rank(self) -> int:
        return (self.rank + 1) % self.world

    @property
    def prev_rank(self) -> int:
        return (self.rank - 1) % self.world


def init(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run). Returns (rank, local, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def ring_handoff(send: torch.Tensor, recv: torch.Tensor, shard: LayerShard):
    """Stage output to the next stage, stage input from the previous one (one P2P pair per step)."""
    if shard.world == 1:
        recv.copy_(send)
        return
    ops = [dist.P2POp(dist.isend, send, shard.next_rank), dist.P2POp(dist.irecv, recv, shard.prev_rank)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def ring_handoff_async(send: torch.Tensor, recv: torch.Tensor, shard: LayerShard, pending=None):
    """Pipelined hand-off: first make the current stream wait for the PREVIOUS step's transfer (``pending``), then post
    this step's send/recv and return its requests without waiting — the transfer runs on the communicator's stream while
    the next step's kernels execute, as in a pipeline whose stages work on different tokens.  ``send``/``recv`` may be
    reused every step: the previous transfer has been waited for before the new one is posted."""
    for req in pending or ():
        req.wait()
    if shard.world == 1:
        recv.copy_(send)
        return []
    ops = [dist.P2POp(dist.isend, send, shard.next_rank), dist.P2POp(dist.irecv, recv, shard.prev_rank)]
    return list(dist.batch_isend_irecv(ops))


def _needs_host_staging(t: torch.Tensor) -> bool:
    """gloo moves device tensors for collectives but has no device-side send/recv: stage point-to-point payloads through
    the host (tests on a 1-GPU box; RCCL — backend 'nccl' — sends device buffers over xGMI directly)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def send(t: torch.Tensor, dst: int):
    dist.send(t.cpu() if _needs_host_staging(t) else t, dst)


def recv(t: torch.Tensor, src: int):
    if _needs_host_staging(t):
        buf = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(buf, src)
        t.copy_(buf)
    else:
        dist.recv(t, src)
    return t


def broadcast(t: torch.Tensor, src: int):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return t


def broadcast_object(obj, src: int):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src)
    return box[0]


_STAGES = weakref.WeakSet()     # the live PipelineStages of this process (drain_stages); a stage dies with its model


def drain_stages():
    """Wait for every stage output this process has posted but whose transfer has not completed (end of a generate)."""
    for st in list(_STAGES):
        st.drain()


class PipelineStage:
    """The hand-off of a layer-sharded model: stage r receives the hidden state ``[1, n, hidden]`` of the
    forward in flight from stage r-1, runs its own layer block, and sends the result to stage r+1 — one point-to-point
    transfer per stage boundary and forward, nothing else crosses xGMI (K/V, slot maps and score rows stay with the layer).
    The reference gets the same partition from accelerate's ``device_map='auto'`` hooks (test_passkey.py:25-35), which copy
    the hidden state between devices — and every layer's probability matrix to one device (llama_patch.py:244-246).

    **The strided prefill is a pipeline.**  Chunk i+1's input is the prompt, not chunk i's logits (easykv/easykv.py:426-433),
    and eviction state is per layer, so stage r may start chunk i+1 while stage r+1 still works on chunk i.  The stage
    output is therefore POSTED (``isend``) and the stage carries on: up to ``depth`` outputs may be in flight before a stage
    waits for its successor (one is enough to keep every stage busy; two absorb jitter).  Decode steps resynchronise by
    themselves — the next token comes from the last stage (easykv_amd.api.generate broadcasts it).  ``run_ahead`` records
    how many forwards this stage was ahead of the completion of its oldest posted output (evidence for the tests / bench)."""

    def __init__(self, shard: LayerShard, depth: int = 2):
        self.shard = shard
        self.depth = max(1, depth)
        self._posted = []          # (request, buffer kept alive) of stage outputs in flight, oldest first
        self.n_sent = 0
        # evidence for the tests / bench, bounded (a long-running process must not grow by an entry per forward):
        self.run_ahead = collections.deque(maxlen=4096)   # per send: outputs still in flight right after posting (0 = the successor keeps up)
        self.t_recv, self.t_send = collections.deque(maxlen=4096), collections.deque(maxlen=4096)   # host clock when forward i's input had arrived / its output was posted
        _STAGES.add(self)

    @property
    def first(self) -> bool:
        return self.shard.rank == 0

    @property
    def last(self) -> bool:
        return self.shard.rank == self.shard.world - 1

    def recv_hidden(self, like: torch.Tensor) -> torch.Tensor:
        """Stage input: ``like`` itself on the first stage, else the previous stage's output (same shape / dtype)."""
        if self.shard.world == 1 or self.first:
            self.t_recv.append(time.time())
            return like
        got = recv(torch.empty_like(like), self.shard.rank - 1)
        self.t_recv.append(time.time())
        return got

    def send_hidden(self, hidden: torch.Tensor):
        if self.shard.world == 1 or self.last:
            self.t_send.append(time.time())
            return
        buf = hidden.contiguous()
        if _needs_host_staging(buf):
            buf = buf.cpu()            # (gloo: waits for this stage's kernels; RCCL sends the device buffer from the stream)
        self._posted = [(r, b) for (r, b) in self._posted if not r.is_completed()]
        while len(self._posted) >= self.depth:      # the successor is `depth` forwards behind: wait for the oldest transfer
            self._posted.pop(0)[0].wait()
        self._posted.append((dist.isend(buf, self.shard.rank + 1), buf))
        self.n_sent += 1
        self.run_ahead.append(len(self._posted) - 1)
        self.t_send.append(time.time())

    def drain(self):
        for req, _ in self._posted:
            req.wait()
        self._posted = []


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_gather_floats(value: float, device=None):
    """One float of every rank, in rank order (a one-element all-gather over the job's backend)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
