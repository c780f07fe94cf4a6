python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for t in 1 4 16; do vcf2prot_amd/lib/v2p_harness run C2 64 $t | cut -c1-260; done
