for v in old new two old new two; do
cp tools/variants/$v.hip vcf2prot_amd/csrc/stitch_kernels.hip; python -m vcf2prot_amd.build > /dev/null 2>&1
echo "== $v"; python tools/ab.py --workload C3 --samples 2000 --rounds 12 "dbg=0" | tail -1; python tools/ab.py --rounds 8 "dbg=0" | tail -1
done
