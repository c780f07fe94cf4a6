"""The fuzz drivers of tools/  with a handful of seeds each: they must stay runnable, and their seeds widen what the
fixed-seed tests cover -- random irregular streams through every builder, image form and error path of the one call; random VCF text and
malformed columns through the bitmask decode; random Task vectors through the GIR-faithful arm from one and from sixteen threads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,first,count,extra", [("fuzz_one_call.py", 31, 5, []), ("fuzz_one_call.py", 41, 3, ["--development"]), ("fuzz_decode.py", 31, 12, []), ("fuzz_gir.py", 31, 24, []), ("fuzz_pipeline.py", 31, 12, [])],
                         ids=["one_call(product)", "one_call(development library)", "decode", "gir", "stream pipeline"])
def test_fuzz_tool_runs_clean(built, gpu_ctx, tool, first, count, extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(first), str(count)] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stderr[-2000:]
    out = json.loads(lines[-1])
    assert p.returncode == 0 and out["n_failures"] == 0 and (out.get("runs", 0) > 0 or out.get("slices", 0) > 0), (out, p.stderr[-1500:])
