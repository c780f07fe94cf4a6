// vcf_index.cpp -- include/v2p_frontend.h part (1): which record lines the engine decodes, where their sample columns
// are, and the file-wide consequence table.  One linear pass over the text; nothing is copied out of it.
//
// Restates, for the GPU engine: readers.rs:8-33 (read_vcf), :96-150 (read_file / get_probands_names),
// :151-231 (get_records / return_if_supported / is_supported_csq), vcf_ds.rs:67-87 (get_consequences_vector),
// and the column arithmetic of vcf_ds.rs:126-190 (the first nine columns are dropped).
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <vector>

#include "../../../include/v2p_frontend.h"
#include "frontend_common.hpp"

struct v2p_vcf_index {
    std::string error;
    std::vector<uint64_t> sample_begin, sample_len;
    std::vector<uint64_t> row_begin, row_end;
    std::vector<uint32_t> csq_begin;
    std::vector<uint8_t> csq_supported;
    std::vector<uint64_t> csq_text_begin;
    std::vector<uint32_t> csq_text_len;
};

namespace {

using v2p_frontend::sup_type_index;

// readers.rs:211-231
bool is_supported_csq(std::string_view s)
{
    size_t pipes = 0;
    for (char c : s) pipes += (c == '|');
    if (pipes != 6) return false;
    return sup_type_index(s.substr(0, s.find('|'))) >= 0;
}

// readers.rs:185-210; info = column 8 of the record
bool record_supported(std::string_view info)
{
    size_t pos = 0;
    while (pos <= info.size()) {
        size_t e = info.find(';', pos);
        if (e == std::string_view::npos) e = info.size();
        std::string_view item = info.substr(pos, e - pos);
        if (item.substr(0, 5) == "BCSQ=") {
            // BCSQ_field[0].split('=')[1]: between the first and the second '='
            std::string_view v = item.substr(5);
            const size_t eq = v.find('=');
            if (eq != std::string_view::npos) v = v.substr(0, eq);
            size_t p = 0;
            while (p <= v.size()) {
                size_t c = v.find(',', p);
                if (c == std::string_view::npos) c = v.size();
                if (is_supported_csq(v.substr(p, c - p))) return true;
                p = c + 1;
            }
            return false;
        }
        pos = e + 1;
    }
    return false;
}

int fail(v2p_vcf_index* x, const std::string& msg)
{
    x->error = msg;
    return V2P_ERR_VCF_FORMAT;
}

}  // namespace

extern "C" {

int v2p_vcf_index_build(const uint8_t* text_u8, uint64_t n, v2p_vcf_index** out)
{
    if (!out) return -1;
    *out = nullptr;
    v2p_vcf_index* x = new (std::nothrow) v2p_vcf_index();
    if (!x) return -1;
    *out = x;                                        // kept on failure so that the message can be read
    if (!text_u8 || !n) return fail(x, "the provided file is empty");                     // readers.rs:109-112
    const char* text = reinterpret_cast<const char*>(text_u8);
    bool have_header = false;
    x->csq_begin.push_back(0);
    uint64_t pos = 0;
    while (pos < n) {
        const char* nl = static_cast<const char*>(memchr(text + pos, '\n', n - pos));
        uint64_t end = nl ? uint64_t(nl - text) : n;
        const uint64_t next = end + 1;
        if (end > pos && text[end - 1] == '\r') --end;                                       // str::lines
        std::string_view line(text + pos, end - pos);
        const uint64_t line0 = pos;
        pos = next;
        if (!line.empty() && line[0] == '#') {
            if (!have_header && line.substr(0, 6) == "#CHROM") {                             // readers.rs:116-127
                have_header = true;
                if (!line.empty() && line.back() == '\t') line.remove_suffix(1);             // readers.rs:128-131
                std::vector<std::pair<uint64_t, uint64_t>> cols;
                size_t p = 0;
                while (p <= line.size()) {
                    size_t t = line.find('\t', p);
                    if (t == std::string_view::npos) t = line.size();
                    cols.emplace_back(line0 + p, t - p);
                    p = t + 1;
                }
                if (cols.size() < 9) return fail(x, "The provided file does not contain the minimum number of columns");   // readers.rs:138-143 (+ drain(0..9))
                for (size_t i = 9; i < cols.size(); ++i) { x->sample_begin.push_back(cols[i].first); x->sample_len.push_back(cols[i].second); }
                if (x->sample_begin.empty()) return fail(x, "The file does not contain any patients!!, after removing the mandatory columns");
            }
            continue;
        }
        // a record line: find its first nine tabs
        uint64_t tabs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int nt = 0;
        for (uint64_t p = 0; p < line.size() && nt < 9;) {
            const char* t = static_cast<const char*>(memchr(line.data() + p, '\t', line.size() - p));
            if (!t) break;
            tabs[nt++] = uint64_t(t - line.data());
            p = tabs[nt - 1] + 1;
        }
        if (nt < 7) return fail(x, "record line with fewer than 8 columns (readers.rs:187 would abort)");
        const uint64_t info_b = tabs[6] + 1, info_e = nt >= 8 ? tabs[7] : line.size();
        std::string_view info = line.substr(info_b, info_e - info_b);
        if (!record_supported(info)) continue;
        if (nt < 9) return fail(x, "supported record without sample columns (vcf_ds.rs:148 would abort)");
        x->row_begin.push_back(line0 + tabs[8] + 1);
        x->row_end.push_back(line0 + line.size());
        // vcf_ds.rs:78: rec.split("BCSQ=")[1] -- from the first "BCSQ=" to the next one or the end of the column
        size_t b = info.find("BCSQ=");
        std::string_view v = info.substr(b + 5);
        const size_t again = v.find("BCSQ=");
        if (again != std::string_view::npos) v = v.substr(0, again);
        const uint64_t v0 = line0 + info_b + b + 5;
        size_t p = 0;
        while (p <= v.size()) {
            size_t c = v.find(',', p);
            if (c == std::string_view::npos) c = v.size();
            std::string_view csq = v.substr(p, c - p);
            x->csq_text_begin.push_back(v0 + p);
            x->csq_text_len.push_back(uint32_t(csq.size()));
            x->csq_supported.push_back(sup_type_index(csq.substr(0, csq.find('|'))) >= 0 ? 1 : 0);   // text_parser::get_type + SUP_TYPE
            p = c + 1;
        }
        if (x->csq_supported.size() >= 0xFFFFFFF0ull) return fail(x, "more than 2^32 consequences");
        x->csq_begin.push_back(uint32_t(x->csq_supported.size()));
    }
    if (!have_header) return fail(x, "Could not find a header line");                        // readers.rs:122-125
    if (x->row_begin.empty()) return fail(x, "Could not extract any records from the provided file!!");   // readers.rs:175-178
    return 0;
}

void v2p_vcf_index_destroy(v2p_vcf_index* x) { delete x; }
const char* v2p_vcf_index_error(const v2p_vcf_index* x) { return x ? x->error.c_str() : ""; }
uint64_t v2p_vcf_index_n_samples(const v2p_vcf_index* x) { return x ? x->sample_begin.size() : 0; }
uint64_t v2p_vcf_index_n_records(const v2p_vcf_index* x) { return x ? x->row_begin.size() : 0; }
uint64_t v2p_vcf_index_n_consequences(const v2p_vcf_index* x) { return x ? x->csq_supported.size() : 0; }
int v2p_vcf_index_sample(const v2p_vcf_index* x, uint64_t i, uint64_t* begin, uint64_t* len)
{
    if (!x || i >= x->sample_begin.size() || !begin || !len) return -1;
    *begin = x->sample_begin[i];
    *len = x->sample_len[i];
    return 0;
}
const uint64_t* v2p_vcf_index_row_begin(const v2p_vcf_index* x) { return x ? x->row_begin.data() : nullptr; }
const uint64_t* v2p_vcf_index_row_end(const v2p_vcf_index* x) { return x ? x->row_end.data() : nullptr; }
const uint32_t* v2p_vcf_index_csq_begin(const v2p_vcf_index* x) { return x ? x->csq_begin.data() : nullptr; }
const uint8_t* v2p_vcf_index_csq_supported(const v2p_vcf_index* x) { return x ? x->csq_supported.data() : nullptr; }
const uint64_t* v2p_vcf_index_csq_text_begin(const v2p_vcf_index* x) { return x ? x->csq_text_begin.data() : nullptr; }
const uint32_t* v2p_vcf_index_csq_text_len(const v2p_vcf_index* x) { return x ? x->csq_text_len.data() : nullptr; }

}  // extern "C"
