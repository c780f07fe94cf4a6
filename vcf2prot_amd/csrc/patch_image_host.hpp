// patch_image_host.hpp -- PATCH images on the HOST (format: patch_format.hpp; what they are for: patch_image.h): a sequential restatement of
// the builder's rules (patch_image.hip: patch_build_kernel) and an interpreter of the image -- the device kernels' second voice in the tests.
//
// The restatement walks the transcripts in order and applies the rules Task by Task: an alt Task of one residue between two reference
// copies, the second going on one residue later in the reference and in the result, is a PATCH and does not end the copy's segment;
// any other Task starts a segment; cells no Task covers are '.' segments (haplotype_instruction.rs:78); FASTA headers and line feeds are
// segments reading the resident header table (personalized_genome.rs:90-113); everything is clipped to the 8 KiB grid.  It differs from
// the device builder in ONE respect that does not change a byte of the result: the device cuts a chain of continuations where its 64-lane
// window ends (the chain goes on as a new segment), the restatement does not.  The tests therefore compare EXECUTED images (device
// kernel, this interpreter, the oracle), and the images' invariants, not their words.
#pragma once
#include <vector>
#include "patch_format.hpp"
#include "rows_image.hpp"            // TxStreamView, the status codes of the rows builder (the same panics)

namespace v2p {

struct PatchImage {
    std::vector<uint64_t> seg;                 // [n_chunks * PATCH_SEG_CAP]
    std::vector<uint32_t> patch;               // [n_chunks * PATCH_PATCH_CAP]
    std::vector<Chunk> chunks;                 // arena order
    std::vector<uint32_t> n_seg, n_patch;      // per chunk (also in the chunk records)
    std::vector<uint64_t> hap_out_begin;
    uint64_t out_bytes = 0;
    uint64_t status = ~0ull;                   // min over offending tasks of (task << 8 | reason), as the device reports it; reason 9: declined
    uint64_t total_seg = 0, total_patch = 0;
};

inline void patch_reference(const TxStreamView& s, uint64_t proteome_len, PatchImage& im)
{
    im = PatchImage();
    // positions: res_counter of haplotype_instruction.rs:90,132
    std::vector<uint64_t> base(s.n_tx + 1, 0);
    for (uint64_t t = 0; t < s.n_tx; ++t) base[t + 1] = base[t] + rows_arena_len(s, t);
    im.out_bytes = base[s.n_tx];
    im.hap_out_begin.assign(s.n_haps + 1, 0);
    for (uint64_t h = 0; h <= s.n_haps; ++h) im.hap_out_begin[h] = base[h < s.n_haps ? s.hap_tx_begin[h] : s.n_tx];
    const uint64_t n_chunks = (im.out_bytes + PATCH_G - 1) / PATCH_G;
    im.seg.assign(n_chunks * PATCH_SEG_CAP, 0); im.patch.assign(n_chunks * PATCH_PATCH_CAP, 0);
    im.n_seg.assign(n_chunks, 0); im.n_patch.assign(n_chunks, 0);
    auto report = [&](uint64_t index, uint32_t reason) { const uint64_t v = (index << 8) | reason; if (v < im.status) im.status = v; };
    // a run of arena bytes [s0, e0) from one source, clipped to every chunk it touches
    auto emit = [&](uint64_t s0, uint64_t e0, uint64_t src, unsigned space) {
        while (s0 < e0) {
            const uint64_t c = s0 / PATCH_G, hi = (c + 1) * PATCH_G, e1 = e0 < hi ? e0 : hi;
            if (im.n_seg[c] < PATCH_SEG_CAP && src <= PATCH_SRC_MAX) im.seg[c * PATCH_SEG_CAP + im.n_seg[c]] = patch_seg(src, uint32_t(s0 - c * PATCH_G), uint32_t(e1 - s0), space);
            else report(c, STATUS_PATCH_DECLINED);
            ++im.n_seg[c];
            if (space == SPACE_IMM) src >>= 8 * (e1 - s0); else if (space != SPACE_FILL) src += e1 - s0;
            s0 = e1;
        }
    };
    auto emit_patch = [&](uint64_t pos, uint8_t byte) {
        const uint64_t c = pos / PATCH_G;
        if (im.n_patch[c] < PATCH_PATCH_CAP) im.patch[c * PATCH_PATCH_CAP + im.n_patch[c]] = patch_word(uint32_t(pos - c * PATCH_G), byte);
        else report(c, STATUS_PATCH_DECLINED);
        ++im.n_patch[c];
    };
    for (uint64_t t = 0; t < s.n_tx; ++t) {
        const uint32_t hl = s.tx_header_len ? s.tx_header_len[t] : 0u;
        const uint64_t hsrc = hl ? proteome_len + s.tx_header_off[t] : 0ull;
        const uint64_t rb = base[t], b0 = rb + hl;
        const uint64_t poff = s.tx_proteome_off[t], alt0 = s.tx_alt_begin[t], n_alt = s.tx_alt_begin[t + 1] - alt0;
        const uint32_t ref_len = s.tx_ref_len[t], res_len = s.tx_res_len[t];
        const uint64_t i0 = s.tx_task_begin[t], i1 = s.tx_task_begin[t + 1];
        if (poff + ref_len > proteome_len) report(i0, ROWS_SRC_OOB);
        if (hl) emit(rb, rb + hl, hsrc, SPACE_PROTEOME);
        // the open reference segment: [seg_s, cur) of the result from poff + seg_src on
        bool open = false;
        uint64_t seg_s = 0, seg_src = 0, cur = 0;
        auto close = [&]() { if (open && cur > seg_s) emit(b0 + seg_s, b0 + cur, poff + seg_src, SPACE_PROTEOME); open = false; };
        bool last_refc = false;                      // the Task just before is a non-empty reference copy (the open segment's last one)
        uint64_t last_srcend = 0;                    // ... and the reference index behind it
        for (uint64_t i = i0; i < i1; ++i) {
            const uint32_t code = s.code[i];
            const uint64_t sp = s.start_pos[i], ln = s.length[i], sr = s.start_pos_res[i];
            uint32_t why = 0;
            if (code > 1u) why = ROWS_BAD_CODE;
            else if (ln > res_len || sr > res_len - ln) why = ROWS_RES_OOB;
            else if (ln > (code == 0 ? uint64_t(ref_len) : n_alt) || sp > (code == 0 ? uint64_t(ref_len) : n_alt) - ln) why = ROWS_SRC_OOB;
            else if (i != i0 && sr < cur) why = ROWS_NOT_CONTIGUOUS;
            if (why) { report(i, why); return; }
            if (sr > cur) { close(); emit(b0 + cur, b0 + sr, 0, SPACE_FILL); }
            if (code == 0) {
                if (ln == 0) { close(); cur = sr; last_refc = false; continue; }       // (an empty copy: nothing to write, and nothing goes on through it)
                close();
                open = true; seg_s = sr; seg_src = sp; cur = sr + ln; last_refc = true; last_srcend = sp + ln;
                continue;
            }
            // an alt Task of ONE residue followed directly by a valid non-empty reference copy (result contiguous, not at the reference's very
            // first residue) is a PATCH, and that copy's run begins one cell early, under it
            bool absorbed = false;
            if (ln == 1 && i + 1 < i1 && s.code[i + 1] == 0) {
                const uint64_t nsp = s.start_pos[i + 1], nln = s.length[i + 1], nsr = s.start_pos_res[i + 1];
                absorbed = nln >= 1 && nsr == sr + 1 && nsp >= 1 && nln <= res_len && nsr <= res_len - nln && nln <= ref_len && nsp <= ref_len - nln;
                if (absorbed) {
                    emit_patch(b0 + sr, s.alt[alt0 + sp]);
                    const bool goes_on = open && last_refc && cur == sr && last_srcend + 1 == nsp;     // a missense between two halves of one run
                    if (!goes_on) { close(); open = true; seg_s = sr; seg_src = nsp - 1; }
                    cur = nsr + nln; last_refc = true; last_srcend = nsp + nln;
                    ++i;
                    continue;
                }
            }
            close();
            last_refc = false;
            if (ln >= 1) {
                if (ln <= PATCH_IMM_MAX) { uint64_t lit = 0; for (uint64_t k = 0; k < ln; ++k) lit |= uint64_t(s.alt[alt0 + sp + k]) << (8 * k); emit(b0 + sr, b0 + sr + ln, lit, SPACE_IMM); }
                else emit(b0 + sr, b0 + sr + ln, alt0 + sp, SPACE_PAYLOAD);
            }
            cur = sr + ln;
        }
        close();
        if (cur < res_len) emit(b0 + cur, b0 + res_len, 0, SPACE_FILL);
        if (hl) emit(b0 + res_len, b0 + res_len + 1, hsrc + hl - 1u, SPACE_PROTEOME);
    }
    im.chunks.resize(n_chunks);
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const bool over = im.n_seg[c] > PATCH_SEG_CAP || im.n_patch[c] > PATCH_PATCH_CAP;
        im.chunks[c] = Chunk{(c * PATCH_SEG_CAP) | (uint64_t(over ? 0u : im.n_patch[c]) << TB_IDX_BITS), (c * PATCH_G) | (uint64_t(over ? 0u : im.n_seg[c]) << 48) | CHUNK_PATCH};
        im.total_seg += im.n_seg[c]; im.total_patch += im.n_patch[c];
    }
}

// The image executed on the host, chunk by chunk, the way stitch_patch_kernel does it: segments first, patches over them.  Returns false
// (and leaves `out` partly written) when a chunk is malformed: a source or result range out of bounds, a cell written twice or never.
inline bool patch_interpret(const uint64_t* seg, const uint32_t* patch, const Chunk* chunks, uint64_t n_chunks, const uint8_t* src0, uint64_t src0_len,
                            const uint8_t* src1, uint64_t src1_len, uint8_t* out, uint64_t out_len)
{
    std::vector<uint8_t> seen(PATCH_G);
    for (uint64_t k = 0; k < n_chunks; ++k) {
        const uint64_t tb = chunks[k].task_begin, dn = chunks[k].dst_n;
        if ((dn & CHUNK_PATCH) != CHUNK_PATCH) return false;
        const uint64_t dst = dn & DST_MASK, c = dst / PATCH_G;
        const uint32_t ns = uint32_t(dn >> 48) & CHUNK_N_MASK, np = patch_chunk_patches(tb);
        if (dst % PATCH_G || dst >= out_len || (tb & TB_IDX_MASK) != c * PATCH_SEG_CAP || ns > PATCH_SEG_CAP || np > PATCH_PATCH_CAP) return false;
        const uint64_t span = out_len - dst < PATCH_G ? out_len - dst : PATCH_G;
        std::fill(seen.begin(), seen.end(), 0);
        for (uint32_t i = 0; i < ns; ++i) {
            const uint64_t w = seg[c * PATCH_SEG_CAP + i], src = patch_seg_src(w);
            const uint32_t st = patch_seg_start(w), ln = patch_seg_len(w);
            const unsigned sp = patch_seg_space(w);
            if (ln == 0 || st + ln > span) return false;
            if (sp == SPACE_PROTEOME && src + ln > src0_len) return false;
            if (sp == SPACE_PAYLOAD && src + ln > src1_len) return false;
            if (sp == SPACE_IMM && ln > PATCH_IMM_MAX) return false;
            for (uint32_t q = 0; q < ln; ++q) {
                if (seen[st + q]) return false;
                seen[st + q] = uint8_t(1 + sp);
                out[dst + st + q] = sp == SPACE_PROTEOME ? src0[src + q] : (sp == SPACE_PAYLOAD ? src1[src + q] : (sp == SPACE_IMM ? uint8_t(src >> (8 * q)) : uint8_t('.')));
            }
        }
        for (uint64_t q = 0; q < span; ++q) if (!seen[q]) return false;
        for (uint32_t i = 0; i < np; ++i) {
            const uint32_t pw = patch[c * PATCH_PATCH_CAP + i], pos = pw & 0x3FFFu;
            if (pos >= span || seen[pos] != 1 + SPACE_PROTEOME) return false;      // a patch sits on a reference cell, once
            seen[pos] = 0xFF;
            out[dst + pos] = uint8_t(pw >> 16);
        }
    }
    return true;
}

}  // namespace v2p
