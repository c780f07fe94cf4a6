#!/usr/bin/env python3
"""Harvest golden vectors for the step-6 SIR executor from the REAL reference.

Runs only in the build container (needs /root/reference): it drives the
reference's prebuilt CPU binary ``bins/Linux/vcf2prot`` (v0.1.2, `-g st`) on
small handcrafted VCF + FASTA inputs and records, per case,

* the Instruction list and the ``Vec<Task>`` the reference generated
  (printed by the binary under ``DEBUG_TXP=<transcript>``,
  transcript_instructions.rs:157-167,372-382),
* the alt tape implied by those instructions (transcript_instructions.rs:654-780),
* the personalized sequence the reference wrote to FASTA,
* the assertions of the reference's own unit test for that case
  (transcript_instructions.rs:884-1594), when the case comes from one.

Everything written to tests/golden/ is data (inputs + expected outputs); no
reference source text is stored.  The GPU box never runs this script.

usage: python oracle/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

REF_ROOT = "/root/reference"
BIN = os.path.join(REF_ROOT, "bins/Linux/vcf2prot")
TI_RS = os.path.join(REF_ROOT, "src/data_structures/InternalRep/transcript_instructions.rs")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sir_oracle import Task, gir_execute  # noqa: E402

VCF_HEADER = "##fileformat=VCFv4.2\n##INFO=<ID=BCSQ,Number=.,Type=String,Description=\"synthetic\">\n"


def write_vcf(path, samples, records):
    """records: list of (csq_string, [mask per sample]); one consequence per record so the
    BCSQ bitmask only uses bits 0 (haplotype 1) and 1 (haplotype 2): MaskDecoder.rs:95-121."""
    with open(path, "w") as f:
        f.write(VCF_HEADER)
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for i, (csq, masks) in enumerate(records):
            gts = "\t".join(f"{'1' if m & 1 else '0'}|{'1' if m & 2 else '0'}:{m}" for m in masks)
            f.write(f"7\t{1000 + i}\tv{i}\tC\tT\t100\tPASS\tAC=1;BCSQ={csq}\tGT:BCSQ\t{gts}\n")


def write_fasta(path, seqs):
    with open(path, "w") as f:
        for k, v in seqs.items():
            f.write(f">{k}\n{v}\n")


def read_fasta_records(path):
    recs, name = [], None
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith(">"):
            recs.append((lines[i][1:], lines[i + 1] if i + 1 < len(lines) else ""))
            i += 2
        else:
            i += 1
    return sorted(recs)


INS_RE = re.compile(r"Instruction \{ code: '(.)', s_state: (true|false), pos_ref: (\d+), pos_res: (\d+), len: (\d+), data: \[(.*?)\] \}")
TASK_RE = re.compile(r"Task \{\s*exe_code: (\d+),\s*start_pos: (\d+),\s*length: (\d+),\s*start_pos_res: (\d+),\s*\}")


def run_reference(vcf, fasta, outdir, engine="st", debug_txp=None, extra=()):
    env = dict(os.environ)
    for k in ("DEBUG_CPU_EXEC", "INSPECT_TXP", "INSPECT_INS_GEN", "PANIC_INSPECT_ERR", "DEBUG_TXP", "DEBUG_GPU"):
        env.pop(k, None)
    if debug_txp:
        env["DEBUG_TXP"] = debug_txp
    p = subprocess.run([BIN, "-f", vcf, "-r", fasta, "-o", outdir, "-g", engine, *extra],
                       env=env, capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout + p.stderr


def alt_from_instructions(instructions):
    """Alt-tape pushes of step 4b (transcript_instructions.rs:654-780): missense pushes its
    payload twice (:659-660); frameshift/stop_lost/insertion/deletion/'2'/'3' once; phi kinds nothing."""
    alt = []
    for ins in instructions:
        c, data = ins["code"], ins["data"]
        if c in "MN":
            alt += list(data) + list(data)
        elif c in "FRKBY" or c in "LW" or c in "IJ" or c in "DC" or c in "23":
            alt += list(data)
        elif c in "GXATQZP":
            pass
        else:
            raise ValueError(f"instruction code {c!r} not handled")
    return "".join(alt)


def harvest_single_transcript(name, ref_seq, csqs, tmp):
    """One sample, haplotype 1 carries every consequence."""
    vcf, fa, out = os.path.join(tmp, "in.vcf"), os.path.join(tmp, "ref.fasta"), os.path.join(tmp, "out")
    os.makedirs(out, exist_ok=True)
    for f in os.listdir(out):
        os.remove(os.path.join(out, f))
    write_fasta(fa, {name: ref_seq, "DUMMYTX": "MAAAAAAAAAK"})
    write_vcf(vcf, ["S1"], [(c, [1]) for c in csqs])
    rc, log = run_reference(vcf, fa, out, "st", debug_txp=name)
    result = {"rc": rc}
    m = re.search(r"mutations are: \[(.*?)\] and computed results is: (\d+)", log)
    instructions = []
    if m:
        for im in INS_RE.finditer(m.group(1)):
            data = re.findall(r"'(.)'", im.group(6))
            instructions.append({"code": im.group(1), "s_state": im.group(2) == "true", "pos_ref": int(im.group(3)),
                                 "pos_res": int(im.group(4)), "len": int(im.group(5)), "data": "".join(data)})
        result["computed_size"] = int(m.group(2))
    result["instructions"] = instructions
    tasks = []
    tm = re.search(r"Vector of tasks is: \[(.*?)\n\]", log, re.S)
    if tm:
        tasks = [[int(x) for x in t] for t in TASK_RE.findall(tm.group(1))]
    result["tasks"] = tasks
    fpath = os.path.join(out, "S1.fasta")
    result["fasta"] = read_fasta_records(fpath) if os.path.exists(fpath) else None
    if rc != 0:
        result["log_tail"] = log[-600:]
        result["panic_line"] = next((ln for ln in log.split("\n") if "panicked at" in ln), "")
    return result


def parse_reference_kats():
    """Pull inputs and assertions of test_correct_translation_1..30 out of the reference's test module."""
    src = open(TI_RS).read().split("\n")
    kats = []
    i = 0
    while i < len(src):
        m = re.search(r"fn (test_correct_translation_(\d+))\(\)", src[i])
        if not m:
            i += 1
            continue
        start = i
        j = i + 1
        while j < len(src) and not re.search(r"fn test_", src[j]):
            j += 1
        body = src[start:j]
        muts, ref_seq, asserts = [], None, {"residues": {}}
        for k, line in enumerate(body):
            s = line.strip()
            if s.startswith("//"):
                continue
            mm = re.search(r'"([^"]*\|[^"]*\|[^"]*\|protein_coding\|[^"]*)"', s)
            if mm:
                muts.append(mm.group(1))
            rm = re.search(r'reference\.insert\("([A-Z0-9]+)"\.to_string\(\),"([A-Z]+)"', s)
            if rm:
                ref_name, ref_seq = rm.group(1), rm.group(2)
            am = re.search(r"assert_eq!\((\d+) as usize, ?res_string\.len\(\)\)", s)
            if am:
                asserts["len"] = int(am.group(1))
            am = re.search(r"assert_eq!\(ref_string\.len\(\)\+ ?(\d+) as usize, ?res_string\.len\(\)\)", s)
            if am:
                asserts["len_delta"] = int(am.group(1))
            if re.search(r"assert_eq!\(&ref_string\.len\(\),&res_string\.len\(\)\)", s):
                asserts["len_delta"] = 0
            am = re.search(r"assert_eq!\(res_array\[(\d+)\],'(.)'\)", s)
            if am:
                asserts["residues"][am.group(1)] = am.group(2)
            am = re.search(r"test_equal_expect\(&ref_string,&res_string,vec!\[(.*?)\]\)", s)
            if am:
                inner = am.group(1)
                rep = re.match(r"(\d+);(\d+)", inner)
                idx = [int(rep.group(1))] * int(rep.group(2)) if rep else [int(x) for x in inner.split(",") if x.strip()]
                asserts["equal_except"] = idx
        kats.append({"name": m.group(1), "source": f"transcript_instructions.rs:{start + 1}",
                     "transcript": ref_name, "ref": ref_seq, "mutations": muts, "asserts": asserts})
        i = j
    return kats


def retarget(csq, transcript):
    """The unit tests hand every mutation to one AltTranscript regardless of the id inside the
    csq string (vcf_ds::AltTranscript::new); through a VCF the id groups them, so rewrite it."""
    f = csq.split("|")
    f[2] = transcript
    return "|".join(f)


APPENDIX_REF = "MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK"
APPENDIX_CASES = [
    ("appendix_zero_length_tasks", ["missense|G|TX|protein_coding|+|1M>1V|1A>T", "missense|G|TX|protein_coding|+|2E>2K|1A>T",
                                    "inframe_insertion|G|TX|protein_coding|+|10V>10VWW|1A>T",
                                    "inframe_deletion|G|TX|protein_coding|+|20FISQ>20F|1A>T",
                                    "missense|G|TX|protein_coding|+|60K>60R|1A>T"]),
    ("appendix_frameshift", ["missense|G|TX|protein_coding|+|5G>5A|1A>T",
                             "frameshift|G|TX|protein_coding|+|30GLEELERGGAKLMNPQRSTVWYACDEFGHIK*>30GYYYY*|1A>T"]),
    ("appendix_stop_lost", ["missense|G|TX|protein_coding|+|7N>7D|1A>T", "stop_lost|G|TX|protein_coding|+|61*>61QQQ|1A>T"]),
    ("appendix_start_lost", ["start_lost|G|TX|protein_coding|+|1M>1K|1A>T"]),
    ("appendix_stop_gained", ["missense|G|TX|protein_coding|+|3D>3E|1A>T", "stop_gained|G|TX|protein_coding|+|41L>41*|1A>T"]),
    ("appendix_adjacent_missense", ["missense|G|TX|protein_coding|+|11L>11I|1A>T", "missense|G|TX|protein_coding|+|12S>12T|1A>T",
                                    "missense|G|TX|protein_coding|+|13T>13S|1A>T"]),
    ("appendix_insertion_then_deletion", ["inframe_insertion|G|TX|protein_coding|+|4L>4LPPPP|1A>T",
                                          "inframe_deletion|G|TX|protein_coding|+|30GLEEL>30G|1A>T",
                                          "missense|G|TX|protein_coding|+|50W>50F|1A>T"]),
]


def check_with_oracle(case):
    """Execute the harvested tasks with the restated executor; must reproduce the reference's FASTA."""
    tasks = [Task(*t) for t in case["tasks"]]
    res = ["."] * case["res_len"]
    gir_execute(tasks, list(case["ref"]), list(case["alt"]), res, "st", debug_cpu_exec=True)
    return "".join(res)


def harvest_cohort_example(out_dir, preset="C1", stem="c1_example", **overrides):
    """BASELINE.json config 1 stand-in: the synthetic cohort written as example.vcf +
    reference_sequences.fasta, run through the reference binary with -g mt and -g st."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    from vcf2prot_amd import build
    build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset, **overrides)
    n_samples = c.n_haplotypes // 2
    samples = [f"SAMPLE{s:04d}" for s in range(n_samples)]
    prot, off = c.proteome(), c.tx_offsets()
    seqs = {c.tx_name(t): prot[int(off[t]):int(off[t + 1])].tobytes().decode() for t in range(c.n_transcripts)}
    records = {}     # csq string -> mask per sample
    order = []
    for h in range(c.n_haplotypes):
        for t, kind, aa in c.describe(h):
            csq = f"{kind}|GENE{t}|{c.tx_name(t)}|protein_coding|+|{aa}|{1000 + t}A>T"
            if csq not in records:
                records[csq] = [0] * n_samples
                order.append(csq)
            records[csq][h // 2] |= 1 << (h % 2)
    vcf_path, fa_path = os.path.join(out_dir, stem + ".vcf"), os.path.join(out_dir, stem + "_reference.fasta")
    write_vcf(vcf_path, samples, [(csq, records[csq]) for csq in order])
    write_fasta(fa_path, seqs)
    result = {"generator": "oracle/make_golden.py", "preset": preset, "overrides": overrides, "samples": samples, "n_records": len(order),
              "oracle_binary": "vcf2prot 0.1.2 (bins/Linux)", "fasta": {}}
    with tempfile.TemporaryDirectory() as tmp:
        per_engine = {}
        for engine in ("st", "mt"):
            out = os.path.join(tmp, engine)
            os.makedirs(out)
            rc, log = run_reference(vcf_path, fa_path, out, engine)
            if rc != 0:
                sys.exit(f"reference binary failed on the cohort example with -g {engine}:\n{log[-800:]}")
            per_engine[engine] = {s: read_fasta_records(os.path.join(out, s + ".fasta")) for s in samples}
        if per_engine["st"] != per_engine["mt"]:
            sys.exit("reference -g st and -g mt disagree on the cohort example")
        result["fasta"] = per_engine["mt"]
    with tempfile.TemporaryDirectory() as tmp:                # -a / --write_all_proteins (personalized_genome.rs:118-204)
        out = os.path.join(tmp, "all")
        os.makedirs(out)
        rc, log = run_reference(vcf_path, fa_path, out, "st", extra=("-a",))
        if rc != 0:
            sys.exit(f"reference binary failed on the cohort example with -a:\n{log[-800:]}")
        result["fasta_write_all"] = {s: read_fasta_records(os.path.join(out, s + ".fasta")) for s in samples}
    with open(os.path.join(out_dir, stem + ".json"), "w") as f:
        json.dump(result, f, indent=1)
    n = sum(len(v) for v in result["fasta"].values())
    print(f"cohort example {preset}: {len(order)} VCF records, {n} FASTA records from the reference (-g st == -g mt)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if not os.path.exists(BIN):
        sys.exit("reference binary not found: this script only runs in the build container")

    cases, skipped = [], []
    with tempfile.TemporaryDirectory() as tmp:
        todo = []
        for k in parse_reference_kats():
            todo.append((k["name"], k["source"], k["transcript"], k["ref"],
                         [retarget(c, k["transcript"]) for c in k["mutations"]], k["asserts"], k["mutations"]))
        for name, muts in APPENDIX_CASES:
            todo.append((name, "SURVEY.md Appendix A (shapes from transcript_instructions.rs:508-780)", "TX", APPENDIX_REF, muts, {}, muts))
        for name, source, tx, ref, muts, asserts, orig in todo:
            r = harvest_single_transcript(tx, ref, muts, tmp)
            if r["rc"] != 0 or r["fasta"] is None:
                skipped.append({"name": name, "reason": "reference binary failed", "detail": r.get("log_tail", "")[-300:]})
                continue
            seq = dict(r["fasta"]).get(f"{tx}_1")
            if seq is None:
                skipped.append({"name": name, "reason": "no record for haplotype 1"})
                continue
            case = {"name": name, "source": source, "transcript": tx, "ref": ref, "mutations": orig,
                    "instructions": r["instructions"], "tasks": r["tasks"],
                    "alt": alt_from_instructions(r["instructions"]) if r["tasks"] else "",
                    "res_len": len(seq), "expected": seq, "asserts": asserts,
                    "oracle_binary": "vcf2prot 0.1.2 (bins/Linux), -g st"}
            got = check_with_oracle(case)
            if got != seq:
                skipped.append({"name": name, "reason": "restated executor disagrees with reference FASTA",
                                "got": got, "expected": seq})
                continue
            # cross-check with the source's own unit-test assertions (version skew guard)
            a, ok = asserts, True
            if "len" in a and a["len"] != len(seq):
                ok = False
            if "len_delta" in a and len(ref) + a["len_delta"] != len(seq):
                ok = False
            for idx, ch in a.get("residues", {}).items():
                if seq[int(idx)] != ch:
                    ok = False
            if "equal_except" in a and len(seq) == len(ref):
                for p in range(len(ref)):
                    if p not in a["equal_except"] and seq[p] != ref[p]:
                        ok = False
            case["matches_source_unit_test"] = ok
            cases.append(case)
    with open(os.path.join(args.out, "kat_transcripts.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py", "cases": cases, "skipped": skipped}, f, indent=1)
    print(f"{len(cases)} cases harvested, {len(skipped)} skipped")
    for s in skipped:
        print("  skipped:", s["name"], "-", s["reason"])
    bad = [c["name"] for c in cases if not c["matches_source_unit_test"]]
    if bad:
        print("  binary(0.1.2) vs source(0.1.5) unit-test skew on:", bad)
    harvest_cohort_example(args.out)
    # wider mixes for the VCF -> FASTA test of the whole stack (tests/test_gpu_vcf_to_fasta.py)
    harvest_cohort_example(args.out, "C1", "e2e_dense", seed_cohort=21, n_samples=6, n_transcripts=48, altered_per_hap=30, alts_poisson=2.5)
    harvest_cohort_example(args.out, "C1", "e2e_long", seed_cohort=22, n_samples=5, n_transcripts=24, fixed_len=0, mean_len=320.0,
                           altered_per_hap=16, alts_poisson=5.0, max_fs_tail=40, max_sl_ext=25, p_empty_hap=0.1)


if __name__ == "__main__":
    main()
