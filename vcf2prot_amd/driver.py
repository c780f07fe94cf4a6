"""Cohorts larger than one device image: the role of the reference's sample-level driver (parts/exec.rs:23-42) for the gpu engine.

The reference hands samples to a Rayon pool one by one; here haplotypes are cut into contiguous ranges whose result bytes fit a
budget (shard.shard_by_bytes's prefix-sum logic) and the ranges stream through v2p_pipeline_* -- H2D of slice k+1 and D2H of slice
k-1 overlap the build + execute of slice k.  Results arrive in haplotype order.

run_streamed (round 6) feeds SLICES OF THE TRANSCRIPT STREAM -- the per-transcript GIRs of step 4b, nothing packed on the host
(v2p_pipeline_submit_stream); run_batched feeds host-packed images (v2p_pipeline_submit).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, List, Sequence, Tuple

import numpy as np

from .engine import Context, Pipeline


def cut_by_bytes(result_bytes: Sequence[int], budget: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) ranges whose result bytes stay within `budget` (a haplotype larger than the budget gets a range
    of its own)."""
    cuts, begin, acc = [], 0, 0
    for h, n in enumerate(result_bytes):
        n = int(n)
        if h > begin and acc + n > budget:
            cuts.append((begin, h))
            begin, acc = h, 0
        acc += n
    if len(result_bytes) > begin:
        cuts.append((begin, len(result_bytes)))
    return cuts


@dataclass
class BatchResult:
    h_begin: int                 # haplotype range of this image
    h_end: int
    out: np.ndarray              # pinned host view of the image's result arena (valid until the next batch is yielded)
    hap_out_begin: np.ndarray    # [h_end - h_begin + 1] offsets into `out`
    digests: object = None       # run_streamed(digests=True): the device digests of the slice's haplotypes

    def haplotype(self, h: int) -> np.ndarray:
        i = h - self.h_begin
        return self.out[int(self.hap_out_begin[i]):int(self.hap_out_begin[i + 1])]


def run_batched(ctx: Context, pack: Callable[[int, int], object], result_bytes: Sequence[int], budget_bytes: int,
                h0: int = 0, slots: int = 3) -> Iterator[BatchResult]:
    """Execute haplotypes h0 .. h0 + len(result_bytes) in HBM-sized images.  `pack(begin, end)` returns a packed image (an object with
    desc, chunks, payload, hap_out_begin, out_bytes: cohort.Cohort.pack, or any builder of the device image format); the resident
    reference must already be uploaded to `ctx`.  Yields one BatchResult per image, in order."""
    ranges = [(h0 + a, h0 + b) for a, b in cut_by_bytes(result_bytes, budget_bytes)]
    pipe = Pipeline(ctx, slots)
    try:
        inflight: List[Tuple[int, int, int, np.ndarray]] = []
        nxt = 0
        while nxt < len(ranges) or inflight:
            while nxt < len(ranges) and len(inflight) < slots:
                a, b = ranges[nxt]
                img = pack(a, b)
                t = pipe.submit(img.desc, img.chunks, img.payload, img.out_bytes)
                inflight.append((t, a, b, np.asarray(img.hap_out_begin).copy()))
                nxt += 1
            t, a, b, hb = inflight.pop(0)
            out = pipe.wait(t)
            yield BatchResult(a, b, out, hb)
            pipe.release(t)
    finally:
        pipe.close()


def run_streamed(ctx: Context, make_stream: Callable[[int, int], object], result_bytes: Sequence[int], budget_bytes: int,
                 h0: int = 0, slots: int = 3, kernel: int = 0, digests: bool = False, copy_threads: int = 0,
                 reserve: Tuple[int, int] = (0, 0)) -> Iterator[BatchResult]:
    """Execute haplotypes h0 .. h0 + len(result_bytes) slice by slice from their TRANSCRIPT STREAM.  `make_stream(begin, end)` returns the
    v2p_txstream of that range (cohort.Cohort.txstream, txstream.TxStreamBuilder.finish: an object with `.struct`, optionally `.close()`);
    the resident reference must already be uploaded to `ctx`.  Yields one BatchResult per slice, in order; `out` is the slot's pinned
    result buffer (FASTA text when the stream carries record headers) and `digests` the device digests when asked for."""
    ranges = [(h0 + a, h0 + b) for a, b in cut_by_bytes(result_bytes, budget_bytes)]
    pipe = Pipeline(ctx, slots)
    try:
        if any(reserve) or copy_threads:
            pipe.reserve(reserve[0], reserve[1], copy_threads)
        inflight: List[Tuple[int, int, int]] = []
        nxt = 0
        while nxt < len(ranges) or inflight:
            while nxt < len(ranges) and len(inflight) < slots:
                a, b = ranges[nxt]
                st = make_stream(a, b)
                t = pipe.submit_stream(st, kernel, digests)
                if hasattr(st, "close"):
                    st.close()                               # staged: the host copy may go
                inflight.append((t, a, b))
                nxt += 1
            t, a, b = inflight.pop(0)
            out = pipe.wait(t)
            info = pipe.result_info(t)
            r = BatchResult(a, b, out, info["hap_out_begin"])
            r.digests = info["digests"]
            yield r
            pipe.release(t)
    finally:
        pipe.close()
