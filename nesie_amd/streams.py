"""Fork/join of independent device work onto side streams.

The loss of a VoteNet step is eight small, mutually independent chains of tiny launches; on one
stream they run back to back at ~4 us each, in a captured hipGraph as one long dependency
chain.  ``fork_join`` runs such chains on side streams forked from -- and joined back into --
the current stream, so a graph captured around it has parallel branches (autograd replays each
branch's backward on the stream its forward ran on, so the backward forks the same way).
Outside a GPU context it simply calls the functions in order.
"""
import torch

_pools = {}
_in_use = {}    # device -> number of pool streams handed out to fork_join calls still running


def _streams(device, first, n):
    """Streams [first, first + n) of the device's pool (created on demand)."""
    pool = _pools.setdefault(device, [])
    while len(pool) < first + n:
        pool.append(torch.cuda.Stream(device))
    return pool[first:first + n]


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


def fork_join(fns, like=None, enabled=True):
    """Run the callables ``fns`` (each returns tensors / nested containers of tensors) as
    parallel branches; returns their results in order.  ``like``: a tensor on the device."""
    if not enabled or like is None or not like.is_cuda or len(fns) < 2:
        return [f() for f in fns]
    cur = torch.cuda.current_stream(like.device)
    # a branch that forks again gets streams of its own (an offset into the pool), so nested
    # calls neither serialise against their parent's branches nor tie them together in a capture
    first = _in_use.get(like.device, 0)
    streams = _streams(like.device, first, len(fns) - 1)
    _in_use[like.device] = first + len(streams)
    outs = [None] * len(fns)
    try:
        for i, f in enumerate(fns[1:]):
            s = streams[i]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i + 1] = f()
        outs[0] = fns[0]()                   # the first branch stays on the current stream
    finally:
        _in_use[like.device] = first
    # under capture the tensors live in the graph's private pool: record_stream has nothing to do
    capturing = torch.cuda.is_current_stream_capturing()
    for i, s in enumerate(streams):
        cur.wait_stream(s)
        if not capturing:
            for t in _tensors(outs[i + 1]):
                t.record_stream(cur)
    return outs
