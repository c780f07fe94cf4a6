"""GPU suite: the C++ host mirror (vcf2prot_amd/csrc/host/ppgg_gpu.hpp) and harness (the role of
the reference's exec::execute, parts/exec.rs:23-42) on top of the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(built):
    from vcf2prot_amd import build
    return build.build_harness()


def test_cpp_mirror_known_answers(harness):
    # task.rs:118-144, gir.rs:172-196, bounds panic, Engine::from_str -- through GIR::execute(Engine::GPU) in C++
    p = subprocess.run([harness, "kat"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "kat: ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.parametrize("preset,n,threads", [("C1", 8, 3), ("C3", 6, 4)])
def test_harness_thread_pool_matches_oracle(harness, coracle, preset, n, threads):
    """Haplotype GIRs executed concurrently from several host threads (one ctx each), as Rayon
    workers enter GIR::execute; every result digest must equal the oracle's."""
    from vcf2prot_amd.cohort import Cohort
    p = subprocess.run([harness, "run", preset, str(n), str(threads)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().split("\n")[-1])
    c = Cohort.preset(preset)
    assert out["haplotypes"] == n
    for h in range(n):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute(t, c.ref_tape_u32(h), hap.alt.astype(np.uint32), np.full(hap.n_res, ord("."), dtype=np.uint32))
        assert out["digests"][h] == coracle.digest_u32(want), (preset, h)


@pytest.mark.parametrize("how", [[], ["--slice-kb", "1"], ["--host-build"]], ids=["one-slice", "slice-per-proband", "host-build"])
@pytest.mark.parametrize("stem", ["c1_example", "e2e_dense", "e2e_long"])
def test_harness_vcf_mode_writes_the_reference_files(harness, tmp_path, stem, how):
    """`v2p_harness vcf in.vcf ref.fasta outdir --no-test`: the reference's command line with no Rust behind it; the files it
    writes hold the same records as the reference binary's -- whether the probands travel through the stream-fed pipeline as one slice,
    as a slice each (--slice-kb 1: every proband flushes one), or through the host builder."""
    golden = os.path.join(ROOT, "tests", "golden")
    want = json.load(open(os.path.join(golden, stem + ".json")))["fasta"]
    p = subprocess.run([harness, "vcf", os.path.join(golden, stem + ".vcf"), os.path.join(golden, stem + "_reference.fasta"), str(tmp_path), "--no-test"] + how,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    if how == ["--slice-kb", "1"]:
        assert json.loads(p.stdout.strip().split("\n")[-1])["slices"] >= min(2, len(want))
    for sample, recs in want.items():
        lines = open(os.path.join(tmp_path, sample + ".fasta")).read().split("\n")[:-1]
        got = sorted([lines[i][1:], lines[i + 1]] for i in range(0, len(lines), 2))
        assert got == sorted(recs), sample


def test_harness_vcf_mode_aborts_like_the_reference(harness, tmp_path):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "decode_cases.json")))["cases"]
    n = 0
    for c in cases:
        if not c["panics"]:
            continue
        vcf, fa = tmp_path / "in.vcf", tmp_path / "ref.fasta"
        vcf.write_text(c["vcf"])
        fa.write_text(c["reference_fasta"])
        p = subprocess.run([harness, "vcf", str(vcf), str(fa), str(tmp_path), "--no-test"], capture_output=True, text=True, timeout=120)
        assert p.returncode == 101 and "panicked" in p.stderr, (c["name"], p.stdout, p.stderr)    # a Rust panic exits with 101
        n += 1
    assert n >= 9


def test_harness_vcf_mode_write_all(harness, tmp_path):
    """-a / --write_all_proteins: every transcript of the reference per haplotype (personalized_genome.rs:118-204)."""
    golden = os.path.join(ROOT, "tests", "golden")
    for stem in ("c1_example", "e2e_long"):
        want = json.load(open(os.path.join(golden, stem + ".json")))["fasta_write_all"]
        out = tmp_path / stem
        out.mkdir()
        p = subprocess.run([harness, "vcf", os.path.join(golden, stem + ".vcf"), os.path.join(golden, stem + "_reference.fasta"), str(out), "--no-test", "-a"],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        for sample, recs in want.items():
            lines = open(os.path.join(out, sample + ".fasta")).read().split("\n")[:-1]
            got = sorted([lines[i][1:], lines[i + 1]] for i in range(0, len(lines), 2))
            assert got == sorted(recs), (stem, sample)


def test_module_command_line(built, tmp_path):
    """python -m vcf2prot_amd -f .. -r .. -o .. -g gpu --no-test"""
    import sys
    golden = os.path.join(ROOT, "tests", "golden")
    p = subprocess.run([sys.executable, "-m", "vcf2prot_amd", "-f", os.path.join(golden, "c1_example.vcf"), "-r", os.path.join(golden, "c1_example_reference.fasta"),
                        "-o", str(tmp_path), "-g", "gpu", "--no-test"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    want = json.load(open(os.path.join(golden, "c1_example.json")))["fasta"]
    assert sorted(f[:-6] for f in os.listdir(tmp_path)) == sorted(want)
    p = subprocess.run([sys.executable, "-m", "vcf2prot_amd", "-f", "x", "-r", "y", "-o", str(tmp_path), "-g", "mt"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode != 0


def test_harness_and_python_pipeline_agree_on_random_vcfs(harness, gpu_ctx, tmp_path):
    """Two hosts over the same C ABI (C++ harness, Python pipeline) on random VCFs with every consequence kind, with and
    without -a, checks on and off: same files, or both abort."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from frontend_util import random_vcf
    from vcf2prot_amd import _native as N
    from vcf2prot_amd import step4a
    from vcf2prot_amd.pipeline import vcf_to_fasta
    rng = random.Random(3)
    n_ok = n_abort = 0
    for trial in range(12):
        text = random_vcf(500 + trial, 40 + 10 * (trial % 3), 3 + trial % 4, max_csq=6, n_tx=25)
        aa = "ACDEFGHIKLMNPQRSTVWY"
        ref = "".join(f">ENST{i:011d}\n{'M' + ''.join(rng.choice(aa) for _ in range(700))}\n" for i in range(0, 25, 1 if trial % 2 else 2))
        vcf, fa = tmp_path / f"t{trial}.vcf", tmp_path / f"t{trial}.fasta"
        vcf.write_text(text)
        fa.write_text(ref)
        for write_all in (False, True):
            for no_test in (True, False):
                out = tmp_path / f"o{trial}_{int(write_all)}_{int(no_test)}"
                out.mkdir()
                cmd = [harness, "vcf", str(vcf), str(fa), str(out)] + (["--no-test"] if no_test else []) + (["-a"] if write_all else []) + (["--slice-kb", "8"] if trial % 2 else [])
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
                try:
                    want = vcf_to_fasta(gpu_ctx, text.encode(), ref, flags=0 if no_test else step4a.DEFAULT_FLAGS, write_all=write_all,
                                        slice_bytes=(256 << 20) if trial % 3 else 4096)
                except N.V2PError:
                    assert p.returncode == 101, (trial, write_all, no_test, p.stdout, p.stderr)
                    n_abort += 1
                    continue
                assert p.returncode == 0, (trial, write_all, no_test, p.stdout, p.stderr)
                for sample, data in want.items():
                    assert open(out / (sample + ".fasta"), "rb").read() == data, (trial, sample, write_all, no_test)
                n_ok += 1
    assert n_ok >= 10 and n_abort >= 4


def test_harness_vcf_mode_write_compressed(harness, tmp_path):
    """-c / --write_compressed: <proband>.fasta.gz (personalized_genome.rs:76-108), same records after gunzip."""
    import gzip
    golden = os.path.join(ROOT, "tests", "golden")
    want = json.load(open(os.path.join(golden, "c1_example.json")))["fasta"]
    p = subprocess.run([harness, "vcf", os.path.join(golden, "c1_example.vcf"), os.path.join(golden, "c1_example_reference.fasta"), str(tmp_path), "--no-test", "-c"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    for sample, recs in want.items():
        lines = gzip.open(os.path.join(tmp_path, sample + ".fasta.gz"), "rt").read().split("\n")[:-1]
        assert sorted([lines[i][1:], lines[i + 1]] for i in range(0, len(lines), 2)) == sorted(recs), sample


@pytest.mark.parametrize("preset,samples,devices", [("C3", 40, 1), ("C3", 40, 3), ("C5", 300, 2), ("C2", 6, 4)])
def test_sharded_host_mode_in_one_process(harness, gpu_ctx, coracle, preset, samples, devices):
    """`v2p_harness sharded <preset> <samples> --devices N` (ppgg::execute_sharded: parts/exec.rs:34-40 across the devices of a node,
    in ONE process -- N contexts, N worker threads, ranges of equal result bytes, one v2p_batch_build_and_execute each; on a
    one-GPU box the shards share the device): the ranges are shard_by_bytes', the byte offsets the prefix sum of the cohort's result
    sizes, and every haplotype's digest the oracle's."""
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import shard_by_bytes
    p = subprocess.run([harness, "sharded", preset, str(samples), "--devices", str(devices), "--oversubscribe", "--threads", "8"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().split("\n")[-1])
    c = Cohort.preset(preset, n_samples=samples)
    n = c.n_haplotypes
    sizes = c.result_sizes(0, n)
    assert out["haplotypes"] == n and out["result_bytes"] == int(sizes.sum()) and len(out["shards"]) == devices
    ranges = shard_by_bytes(sizes.tolist(), devices)
    off = 0
    for r, s in enumerate(out["shards"]):
        assert (s["h0"], s["h1"]) == ranges[r] and s["byte_offset"] == off and s["bytes"] == int(sizes[s["h0"]:s["h1"]].sum())
        off += s["bytes"]
    for h in range(0, n, max(1, n // 64)):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        assert out["digests"][h] == coracle.digest_u8(want), (preset, h)


@pytest.mark.parametrize("preset,samples,devices,slice_mb", [("C3", 40, 1, 8), ("C3", 40, 3, 4), ("C5", 300, 2, 1), ("C2", 6, 2, 32)])
def test_streamed_host_mode_in_one_process(harness, gpu_ctx, coracle, preset, samples, devices, slice_mb):
    """`v2p_harness sharded ... --streamed` (ppgg::execute_streamed): one v2p_pipeline per device context, every shard's Task vectors in
    slices through v2p_pipeline_submit_stream, the results in HOST memory -- the digests printed are computed by the harness from the
    host bytes (and compared there with the device's digests of the arena); here with the oracle's."""
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import shard_by_bytes
    p = subprocess.run([harness, "sharded", preset, str(samples), "--devices", str(devices), "--oversubscribe", "--threads", "8", "--streamed", "--slice-mb", str(slice_mb)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().split("\n")[-1])
    c = Cohort.preset(preset, n_samples=samples)
    n = c.n_haplotypes
    sizes = c.result_sizes(0, n)
    assert out["streamed"] and out["haplotypes"] == n and out["result_bytes"] == int(sizes.sum()) == out["host_bytes"] and len(out["shards"]) == devices
    assert out["slices"] >= devices and out["slices"] >= int(sizes.sum()) // (slice_mb << 20)
    ranges = shard_by_bytes(sizes.tolist(), devices)
    for r, s in enumerate(out["shards"]):
        assert (s["h0"], s["h1"]) == ranges[r]
    for h in range(0, n, max(1, n // 64)):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        assert out["digests"][h] == coracle.digest_u8(want), (preset, h)
