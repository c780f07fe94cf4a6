"""step 5 (haplotype_instruction.rs:75-158) folded into the image builder.

Per-transcript GIRs -- exactly what TranscriptInstruction::get_g_rep returns, offsets relative to
the transcript -- go straight into the device image; the host neither concatenates reference
tapes nor keeps ref/alt/res counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reference_transcript_girs_straight_from_the_binary(gpu_ctx, golden):
    """The 36 harvested transcript GIRs (the reference binary's own Vec<Task> dumps, un-rebased) as ONE
    haplotype; every record must equal the sequence the reference wrote, incl. the empty start-lost
    record and the '.' cell of test_correct_translation_20.  Also with FASTA emit."""
    cases = golden["cases"]
    refs, off = {}, 0
    for c in cases:
        if c["ref"] not in refs:
            refs[c["ref"]] = off
            off += len(c["ref"])
    proteome = np.frombuffer("".join(refs).encode(), dtype=np.uint8)
    headers = "\n" + "".join(f">{c['name']}_1\n" for c in cases)
    hdr_off, o = [], 1
    for c in cases:
        hdr_off.append(o)
        o += len(c["name"]) + 4
    gpu_ctx.upload_reference(proteome, np.frombuffer(headers.encode(), dtype=np.uint8))
    for fasta in (False, True):
        b = gpu_ctx.batch()
        b.begin_haplotype()
        for i, c in enumerate(cases):
            t = np.array(c["tasks"], dtype=np.uint64).reshape(-1, 4)
            b.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], refs[c["ref"]], len(c["ref"]),
                             np.frombuffer(c["alt"].encode(), dtype=np.uint8), c["res_len"],
                             hdr_off[i] if fasta else 0, len(c["name"]) + 4 if fasta else 0)
        b.end_haplotype()
        b.finalize()
        b.execute()
        b.sync()
        text = b.download_hap(0).tobytes().decode()
        if fasta:
            assert text == "".join(f">{c['name']}_1\n{c['expected']}\n" for c in cases)
        else:
            assert text == "".join(c["expected"] for c in cases)
        b.close()


def test_builder_step5_equals_host_step5(built, gpu_ctx, coracle):
    """add_transcript (device-side layout) == add_haplotype (host-rebased GIR) == oracle, on C3 haplotypes."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3", n_samples=3)
    gpu_ctx.upload_proteome(c.proteome())
    tx_off = c.tx_offsets()
    b1, b2 = gpu_ctx.batch(), gpu_ctx.batch()
    wants = []
    for h in range(c.n_haplotypes):
        hap = c.haplotype(h)
        b1.add_haplotype(hap.code, hap.start_pos, hap.length, hap.start_pos_res, hap.seg_ref_begin, hap.seg_proteome_off, hap.alt, hap.n_res)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        wants.append(coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8)))
        # undo step 5: split the haplotype GIR back into transcript GIRs (subtract the running counters)
        b2.begin_haplotype()
        seg = task_i = alt_counter = 0
        for tx, ra, rb in zip(hap.tx_id, hap.tx_res_begin, hap.tx_res_end):
            ra, rb = int(ra), int(rb)
            R = int(tx_off[int(tx) + 1] - tx_off[int(tx)])
            if ra == rb:                                   # start-lost: empty GIR (transcript_instructions.rs:338-343)
                b2.add_transcript(np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint64),
                                  int(tx_off[int(tx)]), R, np.zeros(0, np.uint8), 0)
                continue
            ref_counter = int(hap.seg_ref_begin[seg])
            j = task_i                                     # zero-length tasks sitting on the boundary stay with this transcript
            while j < hap.n_tasks and (int(hap.start_pos_res[j]) < rb or (int(hap.length[j]) == 0 and int(hap.start_pos_res[j]) == rb)):
                j += 1
            idx = slice(task_i, j)
            code, sp, ln = hap.code[idx], hap.start_pos[idx].astype(np.int64), hap.length[idx].astype(np.int64)
            is_alt = code == 1
            alt_len = int((sp[is_alt] + ln[is_alt]).max()) - alt_counter if is_alt.any() else 0
            sp = np.where(is_alt, sp - alt_counter, sp - ref_counter)
            b2.add_transcript(code, sp.astype(np.uint64), hap.length[idx], hap.start_pos_res[idx] - np.uint64(ra),
                              int(tx_off[int(tx)]), R, hap.alt[alt_counter:alt_counter + alt_len], rb - ra)
            alt_counter += alt_len
            task_i = j
            seg += 1
        assert task_i == hap.n_tasks and alt_counter == hap.alt.size
        b2.end_haplotype()
    for b in (b1, b2):
        b.finalize()
        b.execute()
        b.sync()
    for h, w in enumerate(wants):
        assert np.array_equal(b1.download_hap(h), w), h
        assert np.array_equal(b2.download_hap(h), w), h
    b1.close()
    b2.close()
