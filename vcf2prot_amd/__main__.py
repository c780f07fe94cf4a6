"""`python -m vcf2prot_amd -f in.vcf -r reference.fasta -o outdir [-g gpu] [-a] [--no-test]`: the reference's command line
(parts/cli.rs:70-140: -f/--vcf_file, -r/--fasta_ref, -o/--output_path, -g/--engine, -a/--write_all_proteins, -c/--write_compressed) on top of
`v2p_harness vcf`, i.e. the whole program without Rust.  Only the gpu engine exists here: `-g st|mt` is the reference's own
CPU code and is refused."""
import argparse
import os
import subprocess
import sys


def main() -> int:
    ap = argparse.ArgumentParser(prog="python -m vcf2prot_amd")
    ap.add_argument("-f", "--vcf_file", required=True)
    ap.add_argument("-r", "--fasta_ref", required=True)
    ap.add_argument("-o", "--output_path", required=True)
    ap.add_argument("-g", "--engine", default="gpu")
    ap.add_argument("-a", "--write_all_proteins", action="store_true")
    ap.add_argument("-c", "--write_compressed", action="store_true")
    ap.add_argument("--no-test", action="store_true", help="like exporting NO_TEST=1 (cli.rs:275-335): no INSPECT_* checks")
    a = ap.parse_args()
    from .engine import Engine
    if Engine.from_str(a.engine) is not Engine.GPU:                      # engines.rs:17-29
        sys.exit("only -g gpu is implemented here; st / mt are the reference's CPU engines")
    from . import build
    build.build_all()
    os.makedirs(a.output_path, exist_ok=True)
    cmd = [build.build_harness(), "vcf", a.vcf_file, a.fasta_ref, a.output_path]
    if a.no_test or "NO_TEST" in os.environ:
        cmd.append("--no-test")
    if a.write_all_proteins:
        cmd.append("-a")
    if a.write_compressed:
        cmd.append("-c")
    return subprocess.run(cmd).returncode


if __name__ == "__main__":
    sys.exit(main())
