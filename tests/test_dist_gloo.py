"""world_size-2 gloo test of the N > 1 host path (easykv_amd/dist.py): layer-block ownership, the ring hand-off of
the stage output and the max-over-ranks timing reduction that bench.py uses.  The stage compute is the CPU oracle
(test infrastructure) so the pipelined result can be checked against a single-process run."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stage(hidden, layer):
    # stand-in for one decoder layer's path: deterministic, depends on the stage input and the layer id
    return torch.tanh(hidden * (1.0 + 0.1 * layer) + layer)


def _worker(rank, world, port, n_layers, out_q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from easykv_amd import dist as D
    r, _, w = D.init("gloo")
    shard = D.LayerShard(r, w, n_layers)
    # pipeline: rank 0 starts from the embedding, every rank applies its own layer block, hands off to the next
    hidden = torch.arange(8, dtype=torch.float32).view(1, 8) / 8.0
    recv = torch.zeros_like(hidden)
    for stage in range(w):
        if stage == r:
            x = hidden if r == 0 else recv
            for l in range(shard.begin, shard.end):
                x = _stage(x, l)
            hidden = x
        D.ring_handoff(hidden, recv, shard)     # every rank takes part in every exchange (ring)
        D.barrier()
    t = D.max_over_ranks(float(r + 1))
    # plain lists, not tensors: a tensor crosses a torch.multiprocessing queue as a shared-memory handle, and the parent fails
    # with EOFError if this process has exited before the handle is opened (seen once under load)
    out_q.put((r, shard.begin, shard.end, hidden.flatten().tolist(), recv.flatten().tolist(), t))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_layer_shard_partition():
    from easykv_amd.dist import LayerShard
    for world in (1, 2, 3, 4, 8):
        for n_layers in (8, 32, 40):
            owned = []
            for r in range(world):
                s = LayerShard(r, world, n_layers)
                owned += list(range(s.begin, s.end))
                assert s.next_rank == (r + 1) % world and s.prev_rank == (r - 1) % world
            assert owned == list(range(n_layers))


def test_two_rank_pipeline_handoff():
    world, n_layers = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layers, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.arange(8, dtype=torch.float32).view(1, 8) / 8.0
    for l in range(n_layers):
        ref = _stage(ref, l)
    (r0, b0, e0, h0, recv0, t0), (r1, b1, e1, h1, recv1, t1) = res
    h1, recv0 = torch.tensor(h1).view(1, 8), torch.tensor(recv0).view(1, 8)
    assert (b0, e0, b1, e1) == (0, 3, 3, 6)
    assert torch.allclose(h1, ref)            # the last stage holds the full-depth result
    assert torch.allclose(recv0, h1)          # ...and the ring returns it to stage 0 (next token's input)
    assert t0 == t1 == 2.0                    # max over ranks


def _worker_async(rank, world, port, steps, out_q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from easykv_amd import dist as D
    r, _, w = D.init("gloo")
    shard = D.LayerShard(r, w, 4 * w)
    send, recv = torch.zeros(1, 8), torch.zeros(1, 8)
    got, pending = [], []
    for i in range(steps):
        # what bench.py does: this step's "kernels", then wait for the previous transfer and post this step's
        if i > 0:
            for req in pending:
                req.wait()
            got.append(recv.clone())                 # the value posted by the previous rank at step i-1
            pending = []
        send.fill_(float(100 * r + i))
        pending = D.ring_handoff_async(send, recv, shard, pending)
    for req in pending:
        req.wait()
    got.append(recv.clone())
    out_q.put((r, [float(g[0, 0]) for g in got]))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_pipelined_async_handoff():
    """bench.py's overlapped hand-off: the transfer of step i is waited for at step i+1; every step's value arrives, in order."""
    world, steps = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == [100.0 + i for i in range(steps)]     # rank 0 receives rank 1's values
    assert res[1] == [0.0 + i for i in range(steps)]
