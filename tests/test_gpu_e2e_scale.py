"""GPU suite: the 1 GB VCF of the BASELINE.md section 2 cohort (200 samples x 2 000 transcripts) through `v2p_harness vcf`;
every proband's FASTA must hash to what the reference binary wrote (tests/golden/e2e_200x2000_digests.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_gigabyte_vcf_end_to_end(built):
    env = dict(os.environ, V2P_E2E_RUNS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_vcf_bench.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().split("\n")[-1])
    assert "all 200 probands equals the reference binary's" in out["verified"]
    assert out["this_engine"]["records"] == 778045 and out["this_engine"]["fasta_bytes"] == 337886000


def test_full_alteration_mix_end_to_end(built):
    """100 samples x 1 000 transcripts with the C3 mix (missense, insertions, deletions, frameshifts, stop gained / lost):
    110 000 VCF records, 60 000 altered transcript-haplotypes, every proband's FASTA against the reference binary's digest."""
    env = dict(os.environ, V2P_E2E_RUNS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_vcf_bench.py"), "e2e_mix_100x1000"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().split("\n")[-1])
    assert "all 100 probands equals the reference binary's" in out["verified"]
    assert out["this_engine"]["records"] == 109477
