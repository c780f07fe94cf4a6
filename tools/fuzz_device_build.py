#!/usr/bin/env python3
"""Many seeds of tests/test_gpu_device_build_fuzz.py's random streams (device builder + all three kernels) in one process.
    python tools/fuzz_device_build.py [n_seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from test_gpu_device_build_fuzz import random_stream  # noqa: E402
from vcf2prot_amd import engine  # noqa: E402
from vcf2prot_amd._native import V2PError  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = engine.Context(0, development=True)
    bad = refused = 0
    for seed in range(1000, 1000 + n):
        rng = np.random.default_rng(seed)
        shape = ("snv", "mix", "long")[seed % 3]
        window = int(rng.choice([4096, 8192, 12288, 16384, 28672, 32768, 61440]))
        kernel = int(rng.integers(1, 4))
        if kernel == 3:
            window = int(rng.choice([4096, 8192, 12288]))       # (a dense image's window is its 12 KiB LDS image at most)
        proteome, stream, want = random_stream(rng, n_haps=int(rng.integers(1, 80)), n_ref_tx=int(rng.integers(1, 40)), shape=shape, window=window)
        ctx.upload_proteome(proteome)
        b = ctx.batch()
        try:
            b.build_on_device(stream, window, kernel)
        except V2PError as e:
            refused += 1                      # e.g. a window denser than the kernel's descriptor limit: refused, never mis-built
            b.close()
            continue
        b.execute()
        b.sync()
        for h, w in enumerate(want):
            got = b.download_hap(h)
            if got.size != w.size or not np.array_equal(got, w):
                bad += 1
                print("MISMATCH", seed, shape, window, kernel, h, got.size, w.size, int(np.argmax(got != w)) if got.size == w.size else -1, flush=True)
                break
        b.close()
    print(f"seeds {n} mismatches {bad} refused {refused}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
