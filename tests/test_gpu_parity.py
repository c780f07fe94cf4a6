"""GPU suite: the HIP engine (through the C ABI) against the oracle, bit for bit."""
import numpy as np
import pytest

from gen_util import oracle_run, random_gir, random_tape
from sir_oracle import str_to_u32, u32_to_str

pytestmark = pytest.mark.gpu


def _engine():
    from vcf2prot_amd.engine import GIR, Engine, Task
    return GIR, Engine, Task


def test_task_rs_test_execute(gpu_ctx):
    # task.rs:118-144 -- descending result offsets, untouched cells keep 'x' => ordered path
    GIR, Engine, Task = _engine()
    ref = list("ABCFEFGH")
    g = GIR([Task(0, 1, 1, 8), Task(0, 4, 1, 4), Task(0, 6, 2, 6)], {}, list(reversed(ref)), ref, ["x"] * 10)
    res, _ = g.execute(Engine.from_str("gpu"), gpu_ctx)
    assert "".join(res) == "xxxxExGHBx"


def test_gir_rs_doc_example(gpu_ctx):
    # gir.rs:172-196
    GIR, Engine, Task = _engine()
    g = GIR([Task(0, 0, 4, 0), Task(1, 0, 1, 4)], {"Seq_1": (0, 5)}, ["G"], list("TEST"), ["."] * 5)
    res, ann = g.execute(Engine.GPU, gpu_ctx)
    assert "".join(res) == "TESTG" and ann == {"Seq_1": (0, 5)}


def test_golden_transcripts(gpu_ctx, golden):
    # Task vectors + sequences produced by the reference binary; asserts of transcript_instructions.rs:884-1594
    GIR, Engine, Task = _engine()
    for c in golden["cases"]:
        g = GIR([Task(*t) for t in c["tasks"]], {c["transcript"]: (0, c["res_len"])}, list(c["alt"]), list(c["ref"]),
                ["."] * c["res_len"])
        res, _ = g.execute(Engine.GPU, gpu_ctx)
        assert "".join(res) == c["expected"], c["name"]


def test_non_ascii_chars_survive(gpu_ctx):
    # the GIR path moves Rust chars (u32), not bytes
    ref = np.array([0x41, 0x3A9, 0x1F600, 0x42], dtype=np.uint32)
    alt = np.array([0x10FFFF], dtype=np.uint32)
    res = np.full(5, ord("."), dtype=np.uint32)
    gpu_ctx.execute_gir([0, 1], [0, 0], [4, 1], [0, 4], ref, alt, res)
    assert res.tolist() == [0x41, 0x3A9, 0x1F600, 0x42, 0x10FFFF]


@pytest.mark.parametrize("seed,n_tasks,mean_len,p_gap", [
    (1, 1, 5, 0.0), (2, 7, 3, 0.0), (3, 255, 40, 0.0), (4, 256, 40, 0.0), (5, 257, 40, 0.0),
    (6, 5000, 6, 0.0), (7, 5000, 200, 0.0), (8, 3000, 30, 0.2), (9, 20000, 133, 0.0), (10, 600, 2000, 0.01),
])
def test_random_canonical_gir(gpu_ctx, coracle, seed, n_tasks, mean_len, p_gap):
    rng = np.random.default_rng(seed)
    ref, alt = random_tape(rng, 50000), random_tape(rng, 4000)
    g = random_gir(rng, n_tasks, ref.size, alt.size, mean_len=mean_len, p_gap=p_gap)
    want = oracle_run(coracle, g, ref, alt)
    res = np.full(g["n_res"], ord("."), dtype=np.uint32)
    gpu_ctx.execute_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
    assert np.array_equal(res, want)


def test_uncovered_cells_keep_caller_content(gpu_ctx, coracle):
    rng = np.random.default_rng(21)
    ref, alt = random_tape(rng, 3000), random_tape(rng, 300)
    g = random_gir(rng, 500, ref.size, alt.size, mean_len=9, p_gap=0.3)
    want = oracle_run(coracle, g, ref, alt, fill=ord("x"))
    res = np.full(g["n_res"], ord("x"), dtype=np.uint32)
    gpu_ctx.execute_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
    assert np.array_equal(res, want)


def test_long_tasks_are_split(gpu_ctx, coracle):
    # single tasks longer than a chunk (64 KiB) and than the 22-bit descriptor length
    rng = np.random.default_rng(22)
    ref, alt = random_tape(rng, 3_000_000), random_tape(rng, 10)
    code = np.array([0, 1, 0, 0], dtype=np.uint8)
    sp = np.array([5, 3, 100_003, 1], dtype=np.uint64)
    ln = np.array([1_200_001, 2, 70_000, 0], dtype=np.uint64)
    sr = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    g = dict(code=code, start_pos=sp, length=ln, start_pos_res=sr, n_res=int(ln.sum()))
    want = oracle_run(coracle, g, ref, alt)
    res = np.full(g["n_res"], ord("."), dtype=np.uint32)
    gpu_ctx.execute_gir(code, sp, ln, sr, ref, alt, res)
    assert np.array_equal(res, want)


def test_overlapping_tasks_later_wins(gpu_ctx, coracle):
    # not something step 5 emits, but GIR::execute defines it: tasks run in order (gir.rs:233)
    rng = np.random.default_rng(23)
    ref, alt = random_tape(rng, 2000), random_tape(rng, 200)
    n = 300
    code = (rng.random(n) < 0.5).astype(np.uint8)
    ln = rng.integers(0, 40, size=n).astype(np.uint64)
    n_src = np.where(code == 0, ref.size, alt.size)
    sp = (rng.random(n) * (n_src - ln)).astype(np.uint64)
    n_res = 1500
    sr = (rng.random(n) * (n_res - ln.astype(np.int64))).astype(np.uint64)
    g = dict(code=code, start_pos=sp, length=ln, start_pos_res=sr, n_res=n_res)
    want = oracle_run(coracle, g, ref, alt, fill=ord("#"))
    res = np.full(n_res, ord("#"), dtype=np.uint32)
    gpu_ctx.execute_gir(code, sp, ln, sr, ref, alt, res)
    assert np.array_equal(res, want)


def test_out_of_bounds_is_an_error_not_a_write(gpu_ctx):
    from vcf2prot_amd import _native as N
    ref, alt = str_to_u32("ABCDE"), str_to_u32("XY")
    res = np.full(8, ord("."), dtype=np.uint32)
    with pytest.raises(N.V2PError) as e:      # source beyond the tape: task.rs:43 panics
        gpu_ctx.execute_gir([0, 0], [0, 3], [2, 5], [0, 2], ref, alt, res)
    assert e.value.code == N.V2P_ERR_SRC_OOB and e.value.index == 1
    with pytest.raises(N.V2PError) as e:      # result beyond the tape
        gpu_ctx.execute_gir([1], [0], [2], [7], ref, alt, res)
    assert e.value.code == N.V2P_ERR_RES_OOB and e.value.index == 0
    assert u32_to_str(res) == "........"


def test_validate_gir_debug_gpu(gpu_ctx, coracle):
    from vcf2prot_amd import _native as N
    rng = np.random.default_rng(31)
    g = random_gir(rng, 10000, 50000, 4000, mean_len=20)
    assert gpu_ctx.validate_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], 50000, 4000, g["n_res"]) == (-1, 0)
    # break contiguity at two rows: the first one is reported (gir.rs:208-226)
    sr = g["start_pos_res"].copy()
    sr[7000:] += 3
    sr[1234:] += 1
    t = coracle.pack_tasks(g["code"], g["start_pos"], g["length"], sr)
    bad, reason = gpu_ctx.validate_gir(g["code"], g["start_pos"], g["length"], sr, 50000, 4000, g["n_res"] + 4)
    assert bad == coracle.validate(t) == 1234 and reason == N.V2P_ERR_NOT_CONTIGUOUS
    code = g["code"].copy()
    code[99] = 2          # phi code must never reach the executor (haplotype_instruction.rs:154)
    assert gpu_ctx.validate_gir(code, g["start_pos"], g["length"], g["start_pos_res"], 50000, 4000, g["n_res"]) == (99, N.V2P_ERR_BAD_CODE)
    assert gpu_ctx.validate_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], 50000, 4000, g["n_res"] - 1)[1] == N.V2P_ERR_RES_OOB
    sp = g["start_pos"].copy()
    i = int(np.nonzero(g["code"] == 1)[0][5])
    sp[i] = 4000
    ln = g["length"].copy()
    ln[i] = max(1, ln[i])
    sr2 = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    assert gpu_ctx.validate_gir(g["code"], sp, ln, sr2, 50000, 4000, int(ln.sum())) == (i, N.V2P_ERR_SRC_OOB)


def test_debug_gpu_flag_refuses_non_contiguous(built):
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.engine import Context
    with Context(0, debug_gpu=True) as ctx:
        ref, alt = str_to_u32("ABCDEFGH"), str_to_u32("X")
        res = np.full(8, ord("."), dtype=np.uint32)
        with pytest.raises(N.V2PError) as e:
            ctx.execute_gir([0, 1], [0, 0], [4, 1], [0, 5], ref, alt, res)
        assert e.value.code == N.V2P_ERR_NOT_CONTIGUOUS and e.value.index == 1
        ctx.execute_gir([0, 1], [0, 0], [4, 1], [0, 4], ref, alt, res)
        assert u32_to_str(res) == "ABCDX..."


def _batch_case(rng, n_haps, with_empty=True):
    haps = []
    for h in range(n_haps):
        if with_empty and h % 5 == 3:
            haps.append(None)
            continue
        ref, alt = random_tape(rng, int(rng.integers(200, 30000))), random_tape(rng, int(rng.integers(1, 2000)))
        g = random_gir(rng, int(rng.integers(1, 3000)), ref.size, alt.size, mean_len=float(rng.choice([3, 30, 150])),
                       p_gap=float(rng.choice([0.0, 0.1])))
        haps.append((g, ref, alt))
    return haps


def test_batch_add_gir_matches_oracle(gpu_ctx, coracle):
    rng = np.random.default_rng(41)
    haps = _batch_case(rng, 23)
    b = gpu_ctx.batch()
    empty = np.zeros(0, dtype=np.uint64)
    for h in haps:
        if h is None:       # haplotype without any altered transcript: empty GIR
            b.add_gir(np.zeros(0, np.uint8), empty, empty, empty, np.zeros(0, np.uint32), np.zeros(0, np.uint32), 0)
        else:
            g, ref, alt = h
            b.add_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, g["n_res"])
    b.finalize()
    b.execute()
    b.sync()
    dig = b.digests()
    assert b.counts()["n_haps"] == len(haps)
    for i, h in enumerate(haps):
        got = b.download_hap(i)
        if h is None:
            assert got.size == 0 and dig[i] == 0
            continue
        g, ref, alt = h
        want = oracle_run(coracle, g, ref.astype(np.uint8), alt.astype(np.uint8))
        assert np.array_equal(got, want), i
        assert int(dig[i]) == coracle.digest_u8(want), i
    b.close()


def test_batch_resident_proteome(gpu_ctx, coracle):
    """Code-0 tasks rebased from the haplotype's private ref tape onto the resident proteome
    (haplotype_instruction.rs:118,130 builds that tape by concatenating transcripts)."""
    rng = np.random.default_rng(42)
    tx_len = rng.integers(50, 900, size=300)
    tx_off = np.concatenate([[0], np.cumsum(tx_len)]).astype(np.uint64)
    proteome = random_tape(rng, int(tx_off[-1]), dtype=np.uint8)
    gpu_ctx.upload_proteome(proteome)
    b = gpu_ctx.batch()
    wants = []
    for h in range(9):
        pick = np.sort(rng.choice(300, size=int(rng.integers(1, 120)), replace=False))
        seg_len = tx_len[pick]
        seg_begin = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.uint64)
        ref = np.concatenate([proteome[int(tx_off[t]):int(tx_off[t + 1])] for t in pick])
        alt = random_tape(rng, 500, dtype=np.uint8)
        # tasks that never cross a transcript boundary of the private tape
        code, sp, ln = [], [], []
        for s in range(len(pick)):
            a, e = int(seg_begin[s]), int(seg_begin[s + 1])
            cut = int(rng.integers(a, e + 1))
            code += [0, 1, 0]
            sp += [a, int(rng.integers(0, 490)), cut]
            ln += [cut - a, int(rng.integers(0, 6)), e - cut]
        ln = np.array(ln, dtype=np.uint64)
        sr = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
        g = dict(code=np.array(code, np.uint8), start_pos=np.array(sp, np.uint64), length=ln, start_pos_res=sr, n_res=int(ln.sum()))
        wants.append(oracle_run(coracle, g, ref, alt))
        b.add_haplotype(g["code"], g["start_pos"], g["length"], g["start_pos_res"], seg_begin, tx_off[pick], alt, g["n_res"])
    b.finalize()
    b.execute()
    b.sync()
    for i, w in enumerate(wants):
        assert np.array_equal(b.download_hap(i), w), i
    b.close()


def test_batch_rejects_what_the_reference_panics_on(gpu_ctx):
    from vcf2prot_amd import _native as N
    b = gpu_ctx.batch()
    ref, alt = str_to_u32("ABCDE"), str_to_u32("XY")
    with pytest.raises(N.V2PError) as e:
        b.add_gir([2], [0], [1], [0], ref, alt, 4)             # haplotype_instruction.rs:154
    assert e.value.code == N.V2P_ERR_BAD_CODE
    with pytest.raises(N.V2PError) as e:
        b.add_gir([0], [3], [4], [0], ref, alt, 8)
    assert e.value.code == N.V2P_ERR_SRC_OOB
    with pytest.raises(N.V2PError) as e:
        b.add_gir([0, 0], [0, 0], [3, 2], [0, 2], ref, alt, 8)   # overlap: only the ordered GIR path takes it
    assert e.value.code == N.V2P_ERR_NOT_CANONICAL
    assert b.counts()["n_haps"] == 0
    b.close()


def test_two_contexts_from_two_threads(built, coracle):
    """GIR::execute is entered concurrently from Rayon workers (parts/exec.rs:36-39): one ctx per thread."""
    import threading
    from vcf2prot_amd.engine import Context
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            with Context(0) as ctx:
                for _ in range(5):
                    ref, alt = random_tape(rng, 20000), random_tape(rng, 900)
                    g = random_gir(rng, 2000, ref.size, alt.size, mean_len=25, p_gap=0.05)
                    want = oracle_run(coracle, g, ref, alt)
                    res = np.full(g["n_res"], ord("."), dtype=np.uint32)
                    ctx.execute_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
                    if not np.array_equal(res, want):
                        errors.append(seed)
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))
    ts = [threading.Thread(target=worker, args=(s,)) for s in (101, 102, 103, 104)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_streamed_pipeline_matches_oracle(built, gpu_ctx, coracle):
    """v2p_pipeline_*: several images in flight (H2D / kernel / D2H overlap), results byte-exact."""
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Pipeline
    c = Cohort.preset("C3", n_samples=6)
    gpu_ctx.upload_proteome(c.proteome())
    imgs = [c.pack(h, h + 3, n_threads=2) for h in range(0, 12, 3)]
    pipe = Pipeline(gpu_ctx, 2)
    tickets, got = [], []
    for img in imgs:
        if len(tickets) == 2:
            t = tickets.pop(0)
            got.append(pipe.wait(t).copy())
            pipe.release(t)
        tickets.append(pipe.submit(img.desc, img.chunks, img.payload, img.out_bytes))
    for t in tickets:
        got.append(pipe.wait(t).copy())
        pipe.release(t)
    pipe.close()
    for k, img in enumerate(imgs):
        for i in range(3):
            h = 3 * k + i
            hap = c.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            a, b = int(img.hap_out_begin[i]), int(img.hap_out_begin[i + 1])
            assert np.array_equal(got[k][a:b], want), h


def _adversarial_gir(rng, kind):
    """Task vectors that stress the chunker and the block map."""
    n_ref, n_alt = 70000, 5000
    if kind == "tiny":            # thousands of 1-3 residue tasks: >256 tasks without reaching a 16-byte cut (ragged cuts)
        ln = rng.integers(1, 4, size=6000)
    elif kind == "zero_runs":     # long runs of zero-length tasks, whole chunks of them
        ln = np.where(rng.random(5000) < 0.9, 0, rng.integers(1, 300, size=5000))
        ln[1000:1700] = 0
    elif kind == "mixed":         # bimodal: 1-residue alt tasks between long reference runs, some > 64 KiB/4 chars
        ln = np.where(rng.random(3000) < 0.5, 1, rng.integers(100, 900, size=3000))
        ln[::500] = 20000
    elif kind == "block_edges":   # lengths that are multiples of 4 chars (16 bytes) and off-by-one around them
        ln = rng.choice([3, 4, 5, 8, 12, 16, 1024, 1023, 1025], size=4000)
    else:                         # "all_zero"
        ln = np.zeros(700, dtype=np.int64)
    ln = ln.astype(np.uint64)
    n = ln.size
    code = (rng.random(n) < 0.5).astype(np.uint8)
    n_src = np.where(code == 0, n_ref, n_alt).astype(np.uint64)
    ln = np.minimum(ln, n_src)
    sp = (rng.random(n) * (n_src - ln + 1)).astype(np.uint64)
    sp = np.minimum(sp, n_src - ln)
    sr = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    return dict(code=code, start_pos=sp, length=ln, start_pos_res=sr, n_res=int(ln.sum())), n_ref, n_alt


@pytest.mark.parametrize("kind", ["tiny", "zero_runs", "mixed", "block_edges", "all_zero"])
def test_adversarial_task_vectors(gpu_ctx, coracle, kind):
    rng = np.random.default_rng({"tiny": 51, "zero_runs": 52, "mixed": 53, "block_edges": 54, "all_zero": 55}[kind])
    g, n_ref, n_alt = _adversarial_gir(rng, kind)
    ref, alt = random_tape(rng, n_ref), random_tape(rng, n_alt)
    want = oracle_run(coracle, g, ref, alt)
    res = np.full(g["n_res"], ord("."), dtype=np.uint32)
    gpu_ctx.execute_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)      # u32 tapes: offsets x4
    assert np.array_equal(res, want)
    b = gpu_ctx.batch()                                                                                # u8 image: odd byte offsets
    for _ in range(3):
        b.add_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, g["n_res"])
    b.finalize()
    b.execute()
    b.sync()
    want8 = want.astype(np.uint8)
    for h in range(3):
        assert np.array_equal(b.download_hap(h), want8), (kind, h)
    b.close()


def test_empty_batch_and_empty_haplotypes(gpu_ctx):
    b = gpu_ctx.batch()
    b.finalize()
    b.execute()
    b.sync()
    assert b.counts()["n_haps"] == 0 and b.counts()["out_bytes"] == 0
    b.close()
    b = gpu_ctx.batch()
    e8, e64, e32 = np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    for _ in range(5):
        b.add_gir(e8, e64, e64, e64, e32, e32, 0)
    b.finalize()
    b.execute()
    b.sync()
    assert b.counts()["n_haps"] == 5 and b.digests().tolist() == [0] * 5
    b.close()
    res = np.zeros(0, dtype=np.uint32)
    gpu_ctx.execute_gir(e8, e64, e64, e64, e32, e32, res)      # empty GIR (start-lost only haplotype)
    z = np.zeros(3, dtype=np.uint64)                          # only zero-length tasks, empty result tape
    gpu_ctx.execute_gir(np.array([0, 1, 0], np.uint8), z, z, z, e32, e32, res)
