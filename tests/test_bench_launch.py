"""bench.py --gpus N must launch itself (one rank per GPU) from a plain shell, propagate the ranks' exit code and print
ONE JSON line on rank 0.  Here on CPU: --dry-run runs the whole N-rank path (spawn, rendezvous on 127.0.0.1, sharding,
the all-gather of {haplotypes, result bytes} -- once per image, outside the step loop) over gloo without launching a kernel."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_strong_scaling_two_ranks_self_launch(built):
    """one cohort cut into contiguous haplotype ranges of equal result bytes (shard_by_bytes)."""
    one = run_bench("--gpus", "1", "--workload", "C3", "--scaling", "strong", "--samples", "40")
    two = run_bench("--gpus", "2", "--workload", "C3", "--scaling", "strong", "--samples", "40")
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and one["n_gpus"] == 1
    ranks = two["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert sum(r["haplotypes"] for r in ranks) == 80 == one["config"]["haplotypes_rank0"]
    assert sum(r["aa"] for r in ranks) == one["config"]["aa_rank0"]                 # same cohort, same residues
    a, b = (r["result_bytes"] for r in ranks)
    assert abs(a - b) < 0.1 * (a + b)                                               # balanced by bytes


def test_weak_scaling_two_ranks_self_launch(built):
    two = run_bench("--gpus", "2", "--workload", "C2", "--scaling", "weak", "--samples", "3")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert [r["haplotypes"] for r in two["per_rank"]] == [6, 6]
    assert "cpu_baseline" not in two                                                # rank 0 at N=1 only


def test_several_gpus_default_to_the_north_star_run(built):
    """--gpus N > 1 without --workload / --scaling is the north star's run: the C3 cohort, strong scaling (one cohort cut by result
    bytes), with the per-rank table and the world size the collective saw; one GPU stays on BASELINE's configs[1] (C2)."""
    two = run_bench("--gpus", "2", "--samples", "30")
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["workload"].startswith("C3")
    assert two["world_size_seen_by_rccl"] == 2 and len(two["per_rank"]) == 2
    assert sum(r["haplotypes"] for r in two["per_rank"]) == 60
    one = run_bench("--samples", "2")
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["config"]["workload"].startswith("C3")    # the same cohort at every N: the whole of it on one GPU


def test_eight_ranks_cut_the_north_star_cohort_evenly(built):
    """World size 8 over gloo (no GPU): the C3 cohort cut by result bytes -- shares within 1 % of each other for a cohort of this size,
    the ranges contiguous and complete, every rank's offsets those of the single arena (what the size all-gather tells it)."""
    one = run_bench("--gpus", "1", "--samples", "800")
    eight = run_bench("--gpus", "8", "--samples", "800")
    ranks = eight["per_rank"]
    assert eight["world_size_seen_by_rccl"] == 8 and [r["rank"] for r in ranks] == list(range(8))
    assert sum(r["haplotypes"] for r in ranks) == 1600 == one["config"]["haplotypes_rank0"]
    assert sum(r["aa"] for r in ranks) == one["config"]["aa_rank0"]
    share = [r["result_bytes"] for r in ranks]
    assert max(share) - min(share) < 0.01 * (sum(share) / 8)
    first, off = 0, 0
    for r in ranks:                                                                  # exclusive prefix sums = offsets inside ONE arena
        assert r["first_haplotype"] == first and r["byte_offset"] == off
        first += r["haplotypes"]; off += r["result_bytes"]
    assert off == sum(share) and "allgather_us" in eight


def test_child_failure_is_propagated(built):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2", "--workload", "C9"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
