#!/usr/bin/env python3
"""Build container only: run the reference binary (-g mt -v) on the 200-sample x 2 000-transcript cohort VCF of
tools/e2e_cohort_vcf.py (the cohort of BASELINE.md section 2) and keep, per proband, the sha256 of its sorted FASTA records
plus the binary's stage times.  -> tests/golden/e2e_200x2000_digests.json (data only).

usage: python oracle/make_e2e_digests.py [--samples 200 --transcripts 2000]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_cohort_vcf as E  # noqa: E402

BIN = "/root/reference/bins/Linux/vcf2prot"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--transcripts", type=int, default=2000)
    ap.add_argument("--preset", default="C2")
    ap.add_argument("--altered-per-hap", type=int, default=0)
    ap.add_argument("--stem", default="")
    a = ap.parse_args()
    overrides = {"altered_per_hap": a.altered_per_hap} if a.altered_per_hap else {}
    if not os.path.exists(BIN):
        sys.exit("reference binary not found: this script only runs in the build container")
    with tempfile.TemporaryDirectory() as tmp:
        info = E.write_cohort(a.samples, a.transcripts, os.path.join(tmp, "cohort"), a.preset, **overrides)
        out = os.path.join(tmp, "out")
        os.makedirs(out)
        env = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CPU_EXEC", "INSPECT_TXP", "INSPECT_INS_GEN", "PANIC_INSPECT_ERR", "DEBUG_TXP", "DEBUG_GPU")}
        t0 = time.time()
        p = subprocess.run([BIN, "-f", os.path.join(tmp, "cohort.vcf"), "-r", os.path.join(tmp, "cohort_reference.fasta"), "-o", out, "-g", "mt", "-v"],
                           env=env, capture_output=True, text=True)
        wall = time.time() - t0
        if p.returncode != 0:
            sys.exit(p.stdout[-2000:] + p.stderr[-2000:])
        stamps = re.findall(r"(\d\d):(\d\d):(\d\d\.\d+) UTC", p.stdout)
        secs = [int(h) * 3600 + int(m) * 60 + float(s) for h, m, s in stamps]
        digests = {s: E.sample_digest(os.path.join(out, s + ".fasta")) for s in info["samples"]}
        fasta_bytes = sum(os.path.getsize(os.path.join(out, s + ".fasta")) for s in info["samples"])
    res = dict(generator="oracle/make_e2e_digests.py", preset=a.preset, overrides=overrides, samples=a.samples, transcripts=a.transcripts, records=info["records"],
               alterations=info["alterations"], vcf_bytes=info["vcf_bytes"], fasta_bytes=fasta_bytes, digests=digests,
               reference=dict(binary="vcf2prot 0.1.2 (bins/Linux), -g mt -v", host_cores=os.cpu_count(), wall_seconds=wall,
                              stage_seconds=dict(parse_vcf=secs[1] - secs[0], fasta_and_steps_4_to_6=secs[3] - secs[1], write=secs[5] - secs[4]) if len(secs) >= 6 else None))
    with open(os.path.join(ROOT, "tests", "golden", (a.stem or f"e2e_{a.samples}x{a.transcripts}") + "_digests.json"), "w") as f:
        json.dump(res, f, indent=0)
    print({k: v for k, v in res.items() if k != "digests"})


if __name__ == "__main__":
    main()
