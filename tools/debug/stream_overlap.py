#!/usr/bin/env python3
"""Which pairs of streams run CONCURRENTLY?  HIP multiplexes streams onto a few hardware queues; two streams that share a queue run
their work back to back.  For the current stream and each of N freshly created torch streams: a 300-us one-wave delay kernel on both,
wall time of the pair (300 us = concurrent, 600 us = serialized)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402

lib = _lib.lib()
torch.zeros(1, device="cuda")
cur = torch.cuda.current_stream()


def pair_us(a, b, us=300):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(a)
    b.wait_event(e0)
    _lib.check(lib.mh_stream_delay(us, a.cuda_stream), "delay")
    _lib.check(lib.mh_stream_delay(us, b.cuda_stream), "delay")
    eb = torch.cuda.Event()
    eb.record(b)
    a.wait_event(eb)
    e1.record(a)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


pair_us(cur, torch.cuda.Stream())
streams = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12)]
print("current stream vs new stream k:", " ".join("%d:%.0f" % (k, pair_us(cur, s)) for k, s in enumerate(streams)))
print("new stream 0 vs new stream k:  ", " ".join("%d:%.0f" % (k, pair_us(streams[0], s)) for k, s in enumerate(streams) if k))
hp = [torch.cuda.Stream(priority=-1) for _ in range(4)]
print("current vs high-priority k:    ", " ".join("%d:%.0f" % (k, pair_us(cur, s)) for k, s in enumerate(hp)))
