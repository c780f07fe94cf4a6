python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/pcie_pipeline.py --samples 500 --images 10 --slots 3
python tools/pcie_pipeline.py --samples 500 --images 10 --slots 3 --fasta
