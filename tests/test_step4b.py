"""step 4b restated in C++ (vcf2prot_amd/csrc/host/transcript_tasks.cpp),
pinned against what the reference binary printed for the golden transcripts: its Instruction lists
go in, its Vec<Task> dumps (DEBUG_TXP, transcript_instructions.rs:372-382) must come out, together
with the alt tape and the result length of the FASTA record it wrote.  CPU only."""
import numpy as np
import pytest


def test_instruction_lists_give_the_reference_task_vectors(built, golden):
    from vcf2prot_amd.step4b import transcript_g_rep
    codes = set()
    for c in golden["cases"]:
        rc, tasks, alt, res_len = transcript_g_rep(c["instructions"], len(c["ref"]))
        assert rc == 0, c["name"]
        assert tasks.tolist() == c["tasks"], c["name"]
        assert alt.decode() == c["alt"], c["name"]
        assert res_len == c["res_len"], c["name"]
        codes |= {i["code"] for i in c["instructions"]}
    # instruction kinds the golden set exercises
    assert {"M", "N", "I", "D", "F", "R", "G", "L", "K", "A", "B", "P", "T", "2", "0"} <= codes


def test_reference_unit_test_task_tuples(built):
    """transcript_instructions.rs:806-882: expected Task tuples of single instructions."""
    from vcf2prot_amd.step4b import transcript_g_rep
    # :823-840 stop_gained 40VGLHFWTM*>40* : phi task, result = 39 residues
    rc, tasks, alt, res_len = transcript_g_rep([dict(code="G", s_state=False, pos_ref=39, pos_res=39, len=0, data="")], 48)
    assert rc == 0 and tasks.tolist() == [[0, 0, 39, 0]] and alt == b"" and res_len == 39
    # :842-860 stop_lost 489*>489S on a 488-residue reference: Task(1,0,1,488), alt 'S'
    rc, tasks, alt, res_len = transcript_g_rep([dict(code="L", s_state=False, pos_ref=488, pos_res=488, len=1, data="S")], 488)
    assert rc == 0 and tasks.tolist() == [[0, 0, 488, 0], [1, 0, 1, 488]] and alt == b"S" and res_len == 489


def test_expected_result_array_length_kat(built):
    """transcript_instructions.rs:790-804: frameshift 40VGLHFWTM*>40VDSTFGQC on a 50-residue reference -> 47."""
    from vcf2prot_amd.step4b import transcript_g_rep
    rc, tasks, alt, res_len = transcript_g_rep([dict(code="F", s_state=False, pos_ref=39, pos_res=39, len=8, data="VDSTFGQC")], 50)
    assert rc == 0 and res_len == 47 and alt == b"VDSTFGQC"
    assert tasks.tolist() == [[0, 0, 39, 0], [1, 0, 8, 39]]      # :806-822: the frameshift task copies all 8 payload residues


def test_error_paths(built):
    from vcf2prot_amd.step4b import transcript_g_rep
    fs = dict(code="F", s_state=False, pos_ref=9, pos_res=9, len=3, data="VAB")
    ms = dict(code="M", s_state=False, pos_ref=20, pos_res=20, len=1, data="K")
    assert transcript_g_rep([fs, ms], 38)[0] == 1          # :499 "must be the last mutation in a transcript"
    assert transcript_g_rep([dict(code="?", s_state=False, pos_ref=1, pos_res=1, len=1, data="A")], 38)[0] == 2
    rc, tasks, alt, res_len = transcript_g_rep([dict(code="0", s_state=False, pos_ref=0, pos_res=0, len=0, data="")], 38)
    assert rc == 0 and tasks.shape == (0, 4) and res_len == 0      # start lost: empty GIR (:338-343)


@pytest.mark.gpu
def test_instructions_to_fasta_without_rust(built, gpu_ctx, golden):
    """Instruction lists -> step 4b (C++) -> builder step 5 -> stitch kernel -> FASTA text."""
    from vcf2prot_amd.step4b import transcript_g_rep
    cases = golden["cases"]
    refs, off = {}, 0
    for c in cases:
        if c["ref"] not in refs:
            refs[c["ref"]] = off
            off += len(c["ref"])
    headers = "\n" + "".join(f">{c['name']}_1\n" for c in cases)
    gpu_ctx.upload_reference(np.frombuffer("".join(refs).encode(), dtype=np.uint8), np.frombuffer(headers.encode(), dtype=np.uint8))
    b = gpu_ctx.batch()
    b.begin_haplotype()
    o = 1
    for c in cases:
        rc, t, alt, res_len = transcript_g_rep(c["instructions"], len(c["ref"]))
        assert rc == 0
        b.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], refs[c["ref"]], len(c["ref"]),
                         np.frombuffer(alt, dtype=np.uint8), res_len, o, len(c["name"]) + 4)
        o += len(c["name"]) + 4
    b.end_haplotype()
    b.finalize()
    b.execute()
    b.sync()
    assert b.download_hap(0).tobytes().decode() == "".join(f">{c['name']}_1\n{c['expected']}\n" for c in cases)
    b.close()
