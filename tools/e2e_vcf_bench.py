#!/usr/bin/env python3
"""End to end on the GPU box: the 200-sample x 2 000-transcript cohort of BASELINE.md section 2 as a 1 GB VCF through
`v2p_harness vcf` (C++ above the C ABI, no Rust), every proband's FASTA checked against the digest of what the reference
binary wrote for the same file (tests/golden/e2e_200x2000_digests.json, made by oracle/make_e2e_digests.py in the build
container, where the binary needed 126 s on 8 cores).  Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_cohort_vcf as E  # noqa: E402
from vcf2prot_amd import build  # noqa: E402


def main():
    stem = sys.argv[1] if len(sys.argv) > 1 else "e2e_200x2000"
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", stem + "_digests.json")))
    build.build_all()
    harness = build.build_harness()
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        info = E.write_cohort(gold["samples"], gold["transcripts"], os.path.join(tmp, "cohort"), gold.get("preset", "C2"), **gold.get("overrides", {}))
        assert info["vcf_bytes"] == gold["vcf_bytes"] and info["records"] == gold["records"], "the generator is not reproducing the fixture's VCF"
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        runs = []
        for _ in range(int(os.environ.get("V2P_E2E_RUNS", "2"))):    # the second run has the files in the page cache, like the reference's run had
            t0 = time.time()
            p = subprocess.run([harness, "vcf", os.path.join(tmp, "cohort.vcf"), os.path.join(tmp, "cohort_reference.fasta"), out, "--no-test"],
                               capture_output=True, text=True)
            wall = time.time() - t0
            assert p.returncode == 0, p.stdout + p.stderr
            line = json.loads(p.stdout.strip().split("\n")[-1])
            line["wall_seconds_incl_process_start"] = wall
            runs.append(line)
        bad = [s for s in info["samples"] if E.sample_digest(os.path.join(out, s + ".fasta")) != gold["digests"][s]]
        assert not bad, f"{len(bad)} probands differ from the reference binary: {bad[:5]}"
    print(json.dumps({"workload": f"{gold['samples']} samples x {gold['transcripts']} transcripts, {gold['records']} VCF records, "
                                  f"{gold['vcf_bytes'] / 1e9:.2f} GB of VCF -> {gold['fasta_bytes'] / 1e6:.0f} MB of FASTA",
                      "verified": f"sha256 of the sorted records of all {gold['samples']} probands equals the reference binary's",
                      "host_cores": os.cpu_count(), "this_engine": runs[-1], "first_run": runs[0], "reference_binary": gold["reference"]}))


if __name__ == "__main__":
    main()
