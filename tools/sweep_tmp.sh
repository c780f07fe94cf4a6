for v in old new old new; do
cp tools/variants/$v.hip vcf2prot_amd/csrc/stitch_kernels.hip; python -m vcf2prot_amd.build > /dev/null 2>&1
echo "== $v"; python tools/ab.py --rounds 9 "dbg=0" | tail -1
done
