timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gir_shared.py tests/test_gpu_harness.py tests/test_random_kats.py -m gpu -q 2>&1 | tail -2
H=vcf2prot_amd/lib/v2p_harness
timeout 300 $H run C2 1024 16 | cut -c150-300
timeout 300 $H run C2 1024 16 --shared | cut -c150-330
timeout 300 $H run C2 256 16 --shared | cut -c150-330
timeout 300 $H run C2 256 4 | cut -c150-300
timeout 300 $H run C2 256 1 | cut -c150-300
