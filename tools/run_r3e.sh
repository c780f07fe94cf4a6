bash tools/pmc_sq.sh r3_c3w --workload C3 --kernel 4 > gpurun_out/pmc_r3_c3w.txt 2>&1
bash tools/pmc_sq.sh r3_c2w --workload C2 --kernel 4 > gpurun_out/pmc_r3_c2w.txt 2>&1
bash tools/pmc_sq.sh r3_c3pb --workload C3 > gpurun_out/pmc_r3_c3pb.txt 2>&1
bash tools/pmc_mem.sh r3m_c3w --workload C3 --kernel 4 > gpurun_out/pmcm_r3_c3w.txt 2>&1
tail -n 40 gpurun_out/pmc_r3_c3w.txt gpurun_out/pmc_r3_c2w.txt gpurun_out/pmc_r3_c3pb.txt gpurun_out/pmcm_r3_c3w.txt
