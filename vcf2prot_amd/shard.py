"""Multi-GPU sharding of the SIR executor: one process per GPU, haplotypes are the unit.

A haplotype's Tasks read only the shared proteome and that haplotype's alt bytes, and write
only that haplotype's result range (haplotype_instruction.rs:94-133); the reference already
treats samples as independent jobs (parts/exec.rs:36).  So ranks take contiguous haplotype
ranges, the proteome is replicated, and no payload ever crosses xGMI.  The only exchange
is one all-gather of every rank's {haplotypes, result bytes} so each rank (and the writer)
knows its global haplotype index and its byte offset in the cohort-wide result/FASTA stream.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple


def shard_by_count(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `n_units` for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_units, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_bytes(result_bytes: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous haplotype ranges with balanced result bytes: cut the prefix sum at k/world
    quantiles.  Returns one [begin, end) per rank."""
    total = int(sum(int(x) for x in result_bytes))
    cuts, acc, h = [0], 0, 0
    n = len(result_bytes)
    for k in range(1, world):
        target = total * k / world
        while h < n and acc + int(result_bytes[h]) / 2 < target:
            acc += int(result_bytes[h])
            h += 1
        cuts.append(h)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


@dataclass
class GlobalLayout:
    rank: int
    world: int
    n_haps: List[int]          # per rank
    out_bytes: List[int]       # per rank
    hap_offset: int            # global index of this rank's first haplotype
    byte_offset: int           # offset of this rank's arena in the cohort-wide result stream

    @property
    def total_haps(self) -> int:
        return sum(self.n_haps)

    @property
    def total_bytes(self) -> int:
        return sum(self.out_bytes)


def layout_from_sizes(rank: int, n_haps: Sequence[int], out_bytes: Sequence[int]) -> GlobalLayout:
    """What the size all-gather tells rank `rank`: its first global haplotype index and its byte offset in the cohort-wide stream
    (exclusive prefix sums over the ranks before it)."""
    nh, nb = [int(x) for x in n_haps], [int(x) for x in out_bytes]
    return GlobalLayout(rank, len(nh), nh, nb, sum(nh[:rank]), sum(nb[:rank]))


def exchange_sizes(n_haps: int, out_bytes: int, device=None, group=None) -> GlobalLayout:
    """All-gather {haplotypes, result bytes} (16 bytes per rank; RCCL when the tensors live on
    a GPU, gloo on CPU).  Without an initialised process group this is the 1-rank layout."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return GlobalLayout(0, 1, [n_haps], [out_bytes], 0, 0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([n_haps, out_bytes], dtype=torch.int64, device=device)
    everyone = torch.zeros(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    flat = everyone.cpu().tolist()
    nh, nb = flat[0::2], flat[1::2]
    return layout_from_sizes(rank, nh, nb)
