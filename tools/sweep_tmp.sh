python tools/e2e_vcf_bench.py > gpurun_out/e2e_vcf_bench.json 2> gpurun_out/e2e_vcf_bench.err; tail -c 1800 gpurun_out/e2e_vcf_bench.json; tail -3 gpurun_out/e2e_vcf_bench.err
