"""CPU suite: the N>1 path (haplotype sharding + the size all-gather) over gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_by_count_partitions_everything():
    from vcf2prot_amd.shard import shard_by_count
    for n in (0, 1, 7, 8, 2000, 20001):
        for w in (1, 2, 3, 8):
            r = [shard_by_count(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_bytes_balances_result_bytes():
    from vcf2prot_amd.shard import shard_by_bytes
    rng = np.random.default_rng(3)
    sizes = rng.integers(0, 9_000_000, size=500)
    sizes[::17] = 0                                    # empty haplotypes
    r = shard_by_bytes(sizes, 8)
    assert r[0][0] == 0 and r[-1][1] == 500 and all(r[i][1] == r[i + 1][0] for i in range(7))
    per = [int(sizes[b:e].sum()) for b, e in r]
    assert max(per) - min(per) <= 2 * int(sizes.max())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, preset, n_samples, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import exchange_sizes, shard_by_count
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = Cohort.preset(preset, n_samples=n_samples)
        h0, h1 = shard_by_count(c.n_haplotypes, rank, world)
        img = c.pack(h0, h1, n_threads=2)
        lay = exchange_sizes(h1 - h0, img.out_bytes)
        dig = int(np.bitwise_xor.reduce(img.desc)) if img.desc.size else 0
        q.put((rank, h0, h1, img.out_bytes, lay.hap_offset, lay.byte_offset, lay.total_haps, lay.total_bytes,
               [int(x) for x in img.hap_out_begin], dig, img.n_tasks, img.n_copy_bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("preset,n_samples", [("C1", 4), ("C3", 3)])
def test_two_ranks_over_gloo_tile_the_cohort(built, preset, n_samples):
    """Two processes shard the haplotypes, all-gather their sizes, and together reproduce the
    single-process image's result layout (no data-path collective, only 16 bytes per rank)."""
    import torch.multiprocessing as mp
    from vcf2prot_amd.cohort import Cohort
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, preset, n_samples, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)

    c = Cohort.preset(preset, n_samples=n_samples)
    whole = c.pack(0, c.n_haplotypes, n_threads=1)
    (r0, a0, b0, ob0, ho0, bo0, th0, tb0, hb0, _, nt0, nc0), (r1, a1, b1, ob1, ho1, bo1, th1, tb1, hb1, _, nt1, nc1) = res
    assert (r0, r1) == (0, 1) and a0 == 0 and b0 == a1 and b1 == c.n_haplotypes
    assert (ho0, bo0) == (0, 0) and (ho1, bo1) == (b0 - a0, ob0)
    assert th0 == th1 == c.n_haplotypes and tb0 == tb1 == whole.out_bytes == ob0 + ob1
    glob = hb0[:-1] + [x + bo1 for x in hb1]
    assert glob == [int(x) for x in whole.hap_out_begin]
    assert nt0 + nt1 == whole.n_tasks and nc0 + nc1 == whole.n_copy_bytes


def test_cpp_host_cuts_the_ranges_the_python_rule_cuts(built):
    """ppgg::shard_by_bytes (csrc/host/ppgg_gpu.hpp: the single-process multi-device host mode, `v2p_harness sharded`) against
    shard.shard_by_bytes on random result sizes incl. empty haplotypes, more ranks than haplotypes, one rank."""
    import subprocess
    from vcf2prot_amd import build
    from vcf2prot_amd.shard import shard_by_bytes
    harness = build.build_harness()
    rng = np.random.default_rng(11)
    for trial in range(25):
        n = int(rng.integers(0, 300))
        sizes = rng.integers(0, 5_000_000, size=n)
        if n:
            sizes[rng.integers(0, n, size=n // 7)] = 0
        world = int(rng.choice([1, 2, 3, 5, 8, 16]))
        p = subprocess.run([harness, "shard", str(world)] + [str(int(x)) for x in sizes], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        got = [tuple(int(v) for v in ln.split()) for ln in p.stdout.strip().split("\n") if ln]
        assert got == shard_by_bytes(sizes.tolist(), world), (trial, n, world)
