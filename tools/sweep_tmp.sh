python tools/ab.py --workload C5 --samples 10000 --rounds 8 "chunk_tasks=1024" "chunk_tasks=2048,tpt=8" "chunk_tasks=1536,tpt=8" | tail -3
python tools/ab.py --workload C3 --samples 2000 --rounds 6 "chunk_tasks=1024,tpt=4" "chunk_tasks=2048,tpt=8" | tail -2
