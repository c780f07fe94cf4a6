mkdir -p gpurun_out/r3ab
timeout 900 python -m pytest tests/test_gpu_gir_shared.py tests/test_gpu_parity.py tests/test_gpu_harness.py -x -q > gpurun_out/r3ab/pytest.txt 2>&1; tail -15 gpurun_out/r3ab/pytest.txt
H=vcf2prot_amd/lib/v2p_harness
for t in 16 32; do
timeout 300 $H run C2 256 $t | cut -c1-330 > gpurun_out/r3ab/run_c2_$t.json; cat gpurun_out/r3ab/run_c2_$t.json | cut -c1-300
timeout 300 $H run C2 256 $t --shared | cut -c1-330 > gpurun_out/r3ab/run_c2_${t}_shared.json; cat gpurun_out/r3ab/run_c2_${t}_shared.json | cut -c1-300
done
nproc
