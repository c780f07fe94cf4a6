"""VCF text + reference FASTA -> personalized FASTA text per proband, without the Rust host: the record index, the GPU
bitmask decode, the grouping (include/v2p_frontend.h), step 4a and 4b restated in C++ (include/v2p_step4a.h,
v2p_step4b.h), step 5 in the image builder and step 6 + FASTA emit on the GPU (include/vcf2prot_hip.h).

This is the whole of `vcf2prot -f in.vcf -r ref.fasta -g gpu` (main.rs:10-61) as far as the bytes written are concerned;
records come out in transcript order per haplotype where the reference iterates a HashMap.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import _native as N
from . import step4a
from .frontend import VcfIndex, decode_bitmasks, group_per_transcript
from .step4b import inspect_transcript_tasks, transcript_g_rep


def read_fasta(text: str) -> Dict[str, str]:
    """readers.rs:37-76: header = the whole line after '>', sequence = the following lines joined."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    records, header, seq, started = {}, "", [], False
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        if line.startswith(">"):
            if header == "" and not started:
                header = line[1:]
            else:
                records[header] = "".join(seq)
                header, seq = line[1:], []
            started = True
        else:
            seq.append(line)
    records[header] = "".join(seq)
    return records


def vcf_to_fasta(ctx, vcf: bytes, reference_fasta: str, flags: int = step4a.DEFAULT_FLAGS, write_all: bool = False,
                 device_build: bool = True, slice_bytes: int = 256 << 20) -> Dict[str, bytes]:
    """{proband: text of <proband>.fasta}: the altered transcripts (personalized_genome.rs:72-117) or, with write_all
    (-a / --write_all_proteins, :118-204), every transcript of the reference per haplotype, unaltered ones as they are.
    device_build (default): the per-transcript GIRs of whole probands are gathered into SLICES of about `slice_bytes` of FASTA text and
    every slice goes through the stream-fed pipeline as soon as it is complete (v2p_pipeline_submit_stream: step 5, the image, step 6 and
    the record text on the device, the text back in pinned host memory) while steps 4a / 4b of the next probands run here;
    False: the host builder (v2p_batch_add_transcript), one image -- same bytes."""
    ref = read_fasta(reference_fasta)
    idx = VcfIndex(vcf)
    lists = decode_bitmasks(ctx, idx)
    groups = group_per_transcript(idx, lists)
    names = [groups.transcript_name(r) for r in range(groups.n_transcripts)]
    if write_all:
        names = sorted(set(names) | set(ref), key=lambda x: x.encode())
    # resident reference: the transcripts the file touches, and their two record headers each
    off, pieces, hdr, hdr_off = {}, [], ["\n"], {}
    pos, hpos = 0, 1
    for nm in names:
        if nm in ref:
            off[nm] = pos
            pieces.append(ref[nm])
            pos += len(ref[nm])
            for h in (1, 2):
                text = f">{nm}_{h}\n"
                hdr_off[(nm, h)] = (hpos, len(text))
                hdr.append(text)
                hpos += len(text)
    proteome = np.frombuffer("".join(pieces).encode(), dtype=np.uint8) if pieces else np.zeros(0, np.uint8)
    ctx.upload_reference(proteome, np.frombuffer("".join(hdr).encode(), dtype=np.uint8))
    sample_names = idx.sample_names()
    out: Dict[str, bytes] = {}
    b = ctx.batch() if not device_build else None
    pipe = None
    sink = b
    inflight = []                                                      # (ticket, first proband, one past the last, the slice's host stream)
    slots = 3
    if device_build:
        from .engine import Pipeline
        from .txstream import TxStreamBuilder, build_on_device_auto
        pipe = Pipeline(ctx, slots)
        sink = TxStreamBuilder(fasta=True)

    def collect(job):
        t, s0, s1, stream = job
        try:
            text = pipe.wait(t)
            hob = pipe.result_info(t)["hap_out_begin"]
            for s in range(s0, s1):
                k = 2 * (s - s0)
                out[sample_names[s]] = text[int(hob[k]):int(hob[k + 2])].tobytes()
            pipe.release(t)
        except N.V2PError as e:
            if e.code != -9:
                raise
            # a slice even the dense rows image refuses (a 1 KiB row with more than 1 024 descriptors): the host builder takes any stream
            pipe.release(t)
            fb = ctx.batch()
            try:
                build_on_device_auto(fb, stream)
                fb.execute()
                fb.sync()
                for s in range(s0, s1):
                    k = 2 * (s - s0)
                    out[sample_names[s]] = fb.download_hap(k).tobytes() + fb.download_hap(k + 1).tobytes()
            finally:
                fb.close()
        stream.close()

    def flush(s1, s0_box=[0]):
        nonlocal sink
        if s1 == s0_box[0]:
            return
        if len(inflight) == slots:
            collect(inflight.pop(0))
        stream = sink.finish()
        inflight.append((pipe.submit_stream(stream, 0, False), s0_box[0], s1, stream))
        s0_box[0] = s1
        sink = TxStreamBuilder(fasta=True)

    try:
        for hap in range(lists.n_haplotypes):
            if not device_build:
                b.begin_haplotype()
            altered = dict(groups.of(hap))
            todo = [(tx, altered.get(tx)) for tx in names if tx in ref] if write_all else list(altered.items())
            for tx, members in todo:
                if tx not in ref:
                    continue                                           # transcript_instructions.rs:37-41: Err -> skipped
                def reference_copy():                                  # -a: an unaltered transcript is one copy of its reference
                    ho, hl = hdr_off[(tx, 1 + hap % 2)]
                    n = len(ref[tx])
                    sink.add_transcript(np.zeros(1, np.uint8), np.zeros(1, np.uint64), np.array([n], np.uint64), np.zeros(1, np.uint64),
                                        off[tx], n, np.zeros(0, np.uint8), n, ho, hl)
                if members is None:
                    reference_copy()
                    continue
                rc, ins = step4a.group_instructions(groups, members, flags)
                if rc == step4a.SKIP:
                    if write_all:
                        reference_copy()                               # not in the haplotype's annotation -> written as reference (:176-183)
                    continue
                if rc != step4a.OK:
                    raise N.V2PError(-28, f"instruction generation aborts for transcript {tx} (haplotype list {hap})", hap)
                rc, t, alt, res_len = transcript_g_rep(ins, len(ref[tx]))
                if rc == 1:
                    if write_all:
                        reference_copy()
                    continue                                           # haplotype_instruction.rs:100-104: Err -> skipped
                if rc != 0:
                    raise N.V2PError(-28, f"task generation aborts for transcript {tx} (status {rc})", hap)
                if flags & step4a.INSPECT_INS_GEN:                   # the QC switches travel together (cli.rs:337-368): INSPECT_TXP
                    bad, at = inspect_transcript_tasks(t, res_len)
                    if bad:
                        raise N.V2PError(-28, f"INSPECT_TXP fails for transcript {tx} (status {bad}, task {at})", hap)
                ho, hl = hdr_off[(tx, 1 + hap % 2)]
                sink.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], off[tx], len(ref[tx]),
                                    np.frombuffer(alt, dtype=np.uint8), res_len, ho, hl)
            sink.end_haplotype()
            if device_build and hap % 2 == 1 and (sink.result_bytes() >= slice_bytes or hap + 1 == lists.n_haplotypes):
                flush(hap // 2 + 1)                                    # a proband is complete; the slice is full (or the last one)
        if device_build:
            while inflight:
                collect(inflight.pop(0))
        else:
            b.finalize()
            b.execute()
            b.sync()
            for s, name in enumerate(sample_names):
                out[name] = b.download_hap(2 * s).tobytes() + b.download_hap(2 * s + 1).tobytes()
        return out
    finally:
        if pipe is not None:
            pipe.close()
        if b is not None:
            b.close()
