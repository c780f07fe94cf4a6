"""CPU suite: the C-ABI library builds, loads, and exports what include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v2p_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol(built):
    from vcf2prot_amd import _native as N
    lib = ctypes.CDLL(N.HIP_LIB_PATH)
    names = _declared("vcf2prot_hip.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vcf2prot_hip.h but not exported"
    assert set(names) == set(N.HIP_API), "python binding table out of sync with the header"


def test_host_library_exports_every_declared_symbol(built):
    """include/v2p_cohort.h and include/v2p_step4b.h live in libv2p_cohort.so (plain C++, no HIP)."""
    from vcf2prot_amd import _native as N
    from vcf2prot_amd._cohort_api import COHORT_API
    lib = ctypes.CDLL(N.COHORT_LIB_PATH)
    names = _declared("v2p_cohort.h") + _declared("v2p_step4b.h")
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared but not exported"
    assert set(names) == set(COHORT_API), "python binding table out of sync with the headers"


def test_frontend_header_symbols_are_exported(built):
    """include/v2p_frontend.h: the decode lives in the HIP library, the record index and the grouping in the host library."""
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.frontend import DECODE_API, HOST_API
    names = _declared("v2p_frontend.h")
    hip, host = ctypes.CDLL(N.HIP_LIB_PATH), ctypes.CDLL(N.COHORT_LIB_PATH)
    assert len(names) >= 30
    for n in names:
        lib = hip if n.startswith("v2p_decode_") else host
        assert hasattr(lib, n), f"{n} declared in include/v2p_frontend.h but not exported"
    assert set(names) == set(DECODE_API) | set(HOST_API), "python binding table out of sync with the header"


def test_step4a_header_symbols_are_exported(built):
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.step4a import STEP4A_API
    names = _declared("v2p_step4a.h")
    host = ctypes.CDLL(N.COHORT_LIB_PATH)
    for n in names:
        assert hasattr(host, n), f"{n} declared in include/v2p_step4a.h but not exported"
    assert set(names) == set(STEP4A_API)


def test_engine_from_str_is_engines_rs(built):
    # engines.rs:17-29
    from vcf2prot_amd.engine import Engine
    assert Engine.from_str("st") is Engine.ST and Engine.from_str("ST") is Engine.ST
    assert Engine.from_str("mt") is Engine.MT and Engine.from_str("MT") is Engine.MT
    assert Engine.from_str("gpu") is Engine.GPU and Engine.from_str("GPU") is Engine.GPU
    for bad in ("Gpu", "cuda", "", "hip"):
        with pytest.raises(ValueError):
            Engine.from_str(bad)


def test_no_gpu_means_loud_failure(built):
    """Without a device the engine must refuse to start -- it never falls back to a CPU path."""
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.engine import Context
    if N.hip_lib().v2p_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(N.V2PError) as e:
        Context(0)
    assert e.value.code == N.V2P_ERR_HIP


def test_product_does_not_reach_into_oracle():
    """Nothing under vcf2prot_amd/ may import, link or open anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vcf2prot_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"sir_oracle|oracle/|from oracle|import oracle", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_product_library_exports_only_what_the_headers_declare(built):
    """Round 6: libvcf2prot_hip.so is built with -fvisibility=hidden -- its dynamic symbol table holds the C ABI of include/vcf2prot_hip.h and
    include/v2p_frontend.h (the decode) and nothing else: no C++ internals, none of the development library's entry points (PATCH images,
    the packed-flag launcher, the A/B switch setter), nothing of the grid builders."""
    import subprocess
    from vcf2prot_amd import _native as N
    out = subprocess.run(["nm", "-D", "--defined-only", N.HIP_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    declared = set(_declared("vcf2prot_hip.h")) | {n for n in _declared("v2p_frontend.h") if n.startswith("v2p_decode_")}
    assert exported <= declared, sorted(exported - declared)
    assert declared <= exported, sorted(declared - exported)
    for dev_only in ("v2p_batch_download_patch_image", "v2p_stitch_launch", "v2p_bench_set_variant"):
        assert dev_only not in exported
    text = open(os.path.join(ROOT, "include", "vcf2prot_hip.h")).read()
    assert "variant" not in text.split("typedef struct {\n    uint32_t nontemporal;")[1].split("} v2p_launch_opts;")[0].replace("`variant`", "")
