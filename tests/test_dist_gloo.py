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
    seen, per_rank = D.sum_over_ranks(1.0), D.all_gather_floats(10.0 * r + 1.0)      # bench.py: ranks_seen, per-rank us_per_step
    # plain lists, not tensors: a tensor crosses a torch.multiprocessing queue as a shared-memory handle, and the parent fails
    # with EOFError if this process has exited before the handle is opened (seen once under load)
    out_q.put((r, shard.begin, shard.end, hidden.flatten().tolist(), recv.flatten().tolist(), (t, seen, per_rank)))
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
    assert t0 == t1 == (2.0, 2.0, [1.0, 11.0])       # max over ranks; ranks seen by an all-reduce of ones; one float per rank, in rank order


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


def _worker_sharded_oracle(rank, world, port, out_q):
    """The host logic of the layer-sharded driver (LayerShard + PipelineStage + token broadcast) with the CPU oracle as the
    per-layer engine (test infrastructure; the product's engine is the HIP KVBank, see tests/test_hip_sharded_generate.py)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from easykv_amd import dist as D
    from oracle import easykv_oracle as O
    from tests.golden_util import load_golden
    r, _, w = D.init("gloo")
    g = load_golden("dec_roco")
    m = g["meta"]
    qs, ks, vs = (x.float() for x in g["streams"])
    L, H = m["dims"]["L"], m["dims"]["H"]
    P, budget = m["length"], m["config"]["budget"]
    shard = D.LayerShard(r, w, L)
    stage = D.PipelineStage(shard)
    states = {}
    for l in range(shard.begin, shard.end):
        st = O.LayerState(k=ks[l][:, :P].unsqueeze(0), v=vs[l][:, :P].unsqueeze(0))
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        states[l] = st
    ids_log, hid_log, toks = [], [], []
    for step in range(m["config"]["max_new_tokens"]):
        t = P + step
        hidden = stage.recv_hidden(torch.zeros(1, 1, qs.shape[1] * qs.shape[3]))
        for l in range(shard.begin, shard.end):
            st = states[l]
            gen = st.k.shape[2] + 1 - P
            plan = O.StepPlan(policy="roco", phase="decode", evict=gen > budget, score_off=P, budget=budget)
            o, ids = O.layer_step(st, qs[l][:, t:t + 1].unsqueeze(0), ks[l][:, t:t + 1].unsqueeze(0), vs[l][:, t:t + 1].unsqueeze(0), plan)
            hidden = hidden + o[0].transpose(0, 1).reshape(1, 1, -1)
            if ids is not None:
                ids_log.append((step, l, (ids + P).flatten().tolist()))
        stage.send_hidden(hidden)
        tok = torch.tensor([[step % 7]]) if stage.last else torch.zeros(1, 1, dtype=torch.long)   # "sampled" on the last stage only
        toks.append(int(D.broadcast(tok, w - 1)[0, 0]))
        hid_log.append(float(hidden.sum()))
    out_q.put((r, shard.begin, shard.end, ids_log, hid_log, toks))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_layer_sharded_driver_logic_matches_the_reference_golden():
    """Two gloo ranks, one layer block each: per-rank layer state, hidden-state hand-off, token broadcast from the last stage.
    The evicted ids of every layer equal the reference's golden vector; the last stage sees the full-depth hidden state."""
    import numpy as np
    from tests.golden_util import load_golden, split_ids
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_oracle, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("dec_roco")
    ref = split_ids(g)                       # per evicting step: [L, H, 1]
    (r0, b0, e0, ids0, hid0, tok0), (r1, b1, e1, ids1, hid1, tok1) = res
    assert (b0, e0, b1, e1) == (0, 1, 1, 2)
    first = min(s for s, _, _ in ids0)
    for log in (ids0, ids1):
        for step, l, ids in log:
            assert np.array_equal(np.asarray(ids, dtype=np.int32), ref[step - first][l].flatten()), (step, l)
    assert len(ids0) == len(ids1) == len(ref)
    assert tok0 == tok1 == [s % 7 for s in range(len(tok0))]     # every rank continues with the last stage's token
    assert all(abs(a) > 0 for a in hid1) and hid0 != hid1         # rank 1 continued rank 0's running sum
