import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vcf2prot_amd import build
build.build_hip(); build.build_cohort()
from vcf2prot_amd.engine import Context
from stream_util import random_stream
seed = int(sys.argv[1]); kernel = int(sys.argv[2]); variant = int(sys.argv[3]); reps = int(sys.argv[4])
def P(*a): print(*a, file=sys.stderr, flush=True)
with Context(0) as ctx:
    rng = np.random.default_rng(seed)
    shape = ("snv", "mix", "long")[seed % 3]
    n_haps = int(rng.integers(1, 700)); n_ref = int(rng.integers(1, 40)); window = int(rng.choice([1024, 4096, 8192]))
    proteome, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=n_ref, shape=shape, window=window)
    P("stream", stream.struct.n_haps, stream.struct.n_tx, stream.struct.n_tasks, stream.struct.n_alt, "proteome", proteome.size)
    ctx.upload_proteome(proteome)
    rs = ctx.upload_stream(stream)
    ctx.set_launch_opts(variant=variant)
    b = ctx.batch()
    P("call"); b.build_and_execute(rs, kernel, 0); b.sync(); P("called", b.oneshot_info(), b.image_form(), b.counts())
    for rep in range(reps):
        ok = all(np.array_equal(b.download_hap(h), w) for h, w in enumerate(want)); P("rep", rep, "ok", ok, b.image_form())
        P("execute"); b.execute(); b.sync(); P("executed", b.image_form())
    b.close(); rs.close()
P("done")
