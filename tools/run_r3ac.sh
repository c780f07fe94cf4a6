mkdir -p gpurun_out/r3ac
H=vcf2prot_amd/lib/v2p_harness
for cfg in "16 32 8" "16 16 8" "32 32 8"; do
set -- $cfg
echo "threads $1 cap $2 MB batches $3"
V2P_COALESCE_MB=$2 V2P_COALESCE_BATCHES=$3 V2P_COALESCE_PROFILE=1 timeout 300 $H run C2 1024 $1 --shared 2>gpurun_out/r3ac/p.txt | cut -c230-290; cat gpurun_out/r3ac/p.txt
done
timeout 300 python -m pytest tests/test_gpu_gir_shared.py -q 2>&1 | tail -2
