export TMPDIR=/tmp
mkdir -p gpurun_out/prof_decode
python tools/decode_bench.py > gpurun_out/prof_decode/bench_min.json 2> gpurun_out/prof_decode/bench_min.err
python tools/decode_bench.py --format rich --records 50000 > gpurun_out/prof_decode/bench_rich.json 2> gpurun_out/prof_decode/bench_rich.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_decode/trace -o decode -- python3 tools/decode_bench.py --no-cpu-baseline > gpurun_out/prof_decode/bench_trace.json 2>&1
tail -c 1500 gpurun_out/prof_decode/bench_min.json
