// v2p_decode_api.hip -- C ABI of the BCSQ bitmask decode (include/v2p_frontend.h, part 2) on the gfx950 kernels of
// decode_kernels.hip.  Replaces the Engine::GPU arm of VCFRecords::get_csq_per_patient (vcf_ds.rs:192-211).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vcf2prot_hip.h"
#include "../../include/v2p_frontend.h"
#include "decode_kernels.h"
#include "v2p_ctx_internal.h"

using namespace v2p;

struct v2p_decode {
    v2p_ctx* ctx = nullptr;
    uint64_t n_samples = 0, n_records = 0, n_ids = 0;
    uint8_t* d_text = nullptr;        // [16 pad | text | 16 pad]
    uint64_t* d_rows = nullptr;       // row_begin | row_end
    uint32_t* d_csq = nullptr;        // csq_begin | sup_pairs | sup_bits
    uint8_t* d_work = nullptr;
    uint64_t* d_hap_begin = nullptr;
    uint32_t* d_ids = nullptr;
    uint64_t* d_status = nullptr;
    std::vector<uint64_t> hap_begin;
    float ms[4] = {0, 0, 0, 0};
    void release() {
        for (void* p : {(void*)d_text, (void*)d_rows, (void*)d_csq, (void*)d_work, (void*)d_hap_begin, (void*)d_ids, (void*)d_status})
            if (p) (void)hipFree(p);
        d_text = nullptr; d_rows = nullptr; d_csq = nullptr; d_work = nullptr; d_hap_begin = nullptr; d_ids = nullptr; d_status = nullptr;
    }
};

namespace {

int reason_to_code(uint32_t r)
{
    switch (r) {
        case DEC_MASK_NEGATIVE: return V2P_ERR_MASK_NEGATIVE;
        case DEC_MASK_PARSE: return V2P_ERR_MASK_PARSE;
        case DEC_MASK_INDEX: return V2P_ERR_MASK_INDEX;
        case DEC_COLUMNS: return V2P_ERR_COLUMNS;
        case DEC_FIELD_TOO_LONG: return V2P_ERR_FIELD_TOO_LONG;
        case DEC_CAPACITY: return V2P_ERR_CAPACITY;
        default: return V2P_ERR_INVALID_ARG;
    }
}

const char* reason_text(uint32_t r)
{
    switch (r) {
        case DEC_MASK_NEGATIVE: return "An invalid bit mask was encountered (negative; text_parser.rs:210,244)";
        case DEC_MASK_PARSE: return "bit mask word is not a u32 (MaskDecoder.rs:41,47)";
        case DEC_MASK_INDEX: return "bit mask selects a consequence the record does not have (vcf_ds.rs:321)";
        case DEC_COLUMNS: return "record does not have one column per proband (vcf_ds.rs:148)";
        case DEC_FIELD_TOO_LONG: return "sample column longer than the 4 KiB window after its last ':'";
        case DEC_CAPACITY: return "multi-word / id capacity exceeded";
        default: return "decode error";
    }
}

struct Guard {
    v2p_ctx* c;
    explicit Guard(v2p_ctx* c_) : c(c_) { ctx_lock(c); }
    ~Guard() { ctx_unlock(c); }
};

void fill_args(DecodeArgs& a, const uint8_t* d_text, uint64_t n_text, const uint64_t* d_row_begin, const uint64_t* d_row_end,
               uint64_t n_records, uint64_t n_samples, const uint32_t* d_csq_begin, const uint32_t* d_sup_pairs,
               const uint32_t* d_sup_bits, uint8_t* d_work, uint64_t ovf_words, uint64_t* d_hap_begin, uint32_t* d_ids,
               uint64_t ids_capacity, uint64_t* d_status)
{
    const DecodeLayout L = decode_layout(n_records, n_samples, ovf_words);
    a.text = d_text; a.n_text = n_text;
    a.row_begin = d_row_begin; a.row_end = d_row_end;
    a.n_rows = uint32_t(n_records); a.n_samples = uint32_t(n_samples);
    a.csq_begin = d_csq_begin; a.sup_pairs = d_sup_pairs; a.sup_bits = d_sup_bits;
    a.carriers = reinterpret_cast<DecCarrier*>(d_work + L.carriers_off);
    a.row_nnz = reinterpret_cast<uint32_t*>(d_work + L.nnz_off);
    a.cnt = reinterpret_cast<uint32_t*>(d_work + L.cnt_off);
    a.group_tot = reinterpret_cast<uint32_t*>(d_work + L.group_off);
    a.blk_total = reinterpret_cast<uint32_t*>(d_work + L.blk_off);
    a.ovf = reinterpret_cast<uint32_t*>(d_work + L.ovf_off);
    a.ovf_capacity = ovf_words;
    a.ovf_used = reinterpret_cast<unsigned long long*>(d_work + L.ovf_used_off);
    a.hap_begin = d_hap_begin; a.ids = d_ids; a.ids_capacity = ids_capacity;
    a.status = reinterpret_cast<unsigned long long*>(d_status);
}

bool sizes_ok(uint64_t n_records, uint64_t n_samples, uint64_t ovf_words)
{
    return n_records >= 1 && n_samples >= 1 && n_records < (1ull << 31) && n_samples < (1ull << 30) && ovf_words < (1ull << 31);
}

}  // namespace

#define DTRY(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { d->release(); delete d; \
    return ctx_fail(ctx, V2P_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e__), -1); } } while (0)

// hipMalloc, and under V2P_DEBUG_POISON=1 (vcf2prot_hip.h: a debugging switch) the allocation filled with 0xA5: no result may depend on
// what fresh or recycled device memory held
static hipError_t dmalloc(void** p, size_t n)
{
    static const bool poison = [] { const char* e = getenv("V2P_DEBUG_POISON"); return e && e[0] == '1'; }();
    const hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess && poison && n) { (void)hipDeviceSynchronize(); (void)hipMemset(*p, 0xA5, n); (void)hipDeviceSynchronize(); }
    return e;
}

extern "C" {

uint64_t v2p_decode_workspace_bytes(uint64_t n_records, uint64_t n_samples, uint64_t ovf_words)
{
    return decode_layout(n_records, n_samples, ovf_words).total;
}

int v2p_decode_launch(void* hip_stream, const uint8_t* d_text, uint64_t n_text,
                      const uint64_t* d_row_begin, const uint64_t* d_row_end, uint64_t n_records, uint64_t n_samples,
                      const uint32_t* d_csq_begin, const uint32_t* d_sup_pairs, const uint32_t* d_sup_bits,
                      uint8_t* d_workspace, uint64_t ovf_words, uint64_t* d_hap_begin, uint32_t* d_ids, uint64_t ids_capacity,
                      uint64_t* d_status, unsigned phases)
{
    if (!d_text || !d_row_begin || !d_row_end || !d_csq_begin || !d_sup_pairs || !d_sup_bits || !d_workspace || !d_hap_begin || !d_status)
        return V2P_ERR_INVALID_ARG;
    if (!sizes_ok(n_records, n_samples, ovf_words) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) return V2P_ERR_INVALID_ARG;
    DecodeArgs a{};
    fill_args(a, d_text, n_text, d_row_begin, d_row_end, n_records, n_samples, d_csq_begin, d_sup_pairs, d_sup_bits,
              d_workspace, ovf_words, d_hap_begin, d_ids, d_ids ? ids_capacity : 0, d_status);
    return launch_decode(a, reinterpret_cast<hipStream_t>(hip_stream), phases) == hipSuccess ? V2P_OK : V2P_ERR_HIP;
}

int v2p_decode_run(v2p_ctx* ctx, const uint8_t* text, uint64_t n_text,
                   const uint64_t* row_begin, const uint64_t* row_end, uint64_t n_records, uint64_t n_samples,
                   const uint32_t* csq_begin, const uint8_t* csq_supported, v2p_decode** out)
{
    if (!ctx) return V2P_ERR_INVALID_ARG;
    Guard g(ctx);
    if (!out || !text || !row_begin || !row_end || !csq_begin || !csq_supported)
        return ctx_fail(ctx, V2P_ERR_INVALID_ARG, "v2p_decode_run: null argument", -1);
    *out = nullptr;
    if (!sizes_ok(n_records, n_samples, 0)) return ctx_fail(ctx, V2P_ERR_INVALID_ARG, "v2p_decode_run: needs at least one record and one sample", -1);
    for (uint64_t r = 0; r < n_records; ++r) {
        if (row_begin[r] > row_end[r] || row_end[r] > n_text || row_end[r] - row_begin[r] >= (1ull << 31))
            return ctx_fail(ctx, V2P_ERR_INVALID_ARG, "v2p_decode_run: record range outside the text", int64_t(r));
        if (csq_begin[r + 1] < csq_begin[r]) return ctx_fail(ctx, V2P_ERR_INVALID_ARG, "v2p_decode_run: csq_begin must ascend", int64_t(r));
    }
    const uint64_t n_csq = csq_begin[n_records];
    // first-word pair masks and the supported bitset (Constants::SUP_TYPE filter of decode_back, vcf_ds.rs:272)
    std::vector<uint32_t> csq(n_records + 1 + n_records + (n_csq + 31) / 32 + 1, 0u);
    uint32_t* sup_pairs = csq.data() + n_records + 1;
    uint32_t* sup_bits = sup_pairs + n_records;
    memcpy(csq.data(), csq_begin, (n_records + 1) * sizeof(uint32_t));
    for (uint64_t i = 0; i < n_csq; ++i) if (csq_supported[i]) sup_bits[i >> 5] |= 1u << (i & 31);
    for (uint64_t r = 0; r < n_records; ++r) {
        uint32_t m = 0;
        const uint32_t b = csq_begin[r], n = csq_begin[r + 1] - b;
        for (uint32_t j = 0; j < n && j < 16; ++j) if (csq_supported[b + j]) m |= 3u << (2 * j);
        sup_pairs[r] = m;
    }

    (void)hipSetDevice(ctx_device(ctx));
    hipStream_t st = ctx_stream(ctx);
    v2p_decode* d = new (std::nothrow) v2p_decode();
    if (!d) return ctx_fail(ctx, V2P_ERR_HIP, "out of host memory", -1);
    d->ctx = ctx; d->n_samples = n_samples; d->n_records = n_records;
    const uint64_t n_haps = 2 * n_samples;
    DTRY(dmalloc(reinterpret_cast<void**>(&d->d_text), n_text + 512), "hipMalloc(text)");
    DTRY(dmalloc(reinterpret_cast<void**>(&d->d_rows), 2 * n_records * sizeof(uint64_t)), "hipMalloc(rows)");
    DTRY(dmalloc(reinterpret_cast<void**>(&d->d_csq), csq.size() * sizeof(uint32_t)), "hipMalloc(csq)");
    DTRY(dmalloc(reinterpret_cast<void**>(&d->d_hap_begin), (n_haps + 1) * sizeof(uint64_t)), "hipMalloc(hap_begin)");
    DTRY(dmalloc(reinterpret_cast<void**>(&d->d_status), 2 * sizeof(uint64_t)), "hipMalloc(status)");
    uint8_t* d_text = d->d_text + 256;
    DTRY(hipMemcpyAsync(d_text, text, n_text, hipMemcpyHostToDevice, st), "H2D(text)");
    DTRY(hipMemcpyAsync(d->d_rows, row_begin, n_records * sizeof(uint64_t), hipMemcpyHostToDevice, st), "H2D(row_begin)");
    DTRY(hipMemcpyAsync(d->d_rows + n_records, row_end, n_records * sizeof(uint64_t), hipMemcpyHostToDevice, st), "H2D(row_end)");
    DTRY(hipMemcpyAsync(d->d_csq, csq.data(), csq.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D(csq)");

    struct Events {                                     // destroyed on every exit path
        hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    } evs;
    hipEvent_t* ev = evs.e;
    for (int k = 0; k < 5; ++k) DTRY(hipEventCreate(&ev[k]), "hipEventCreate");
    // multi-word masks are rare; start with room for one field in 16 and retry with the exact need if that was short
    uint64_t ovf_words = n_records * n_samples / 4 + (1u << 16);
    if (ovf_words >= (1ull << 31)) ovf_words = (1ull << 31) - 1;
    uint64_t row_bytes = 0;
    for (uint64_t r = 0; r < n_records; ++r) row_bytes += row_end[r] - row_begin[r];
    const uint64_t avg_row = row_bytes / n_records;
    uint32_t parse_threads = avg_row <= 1536 ? 64u : (avg_row <= 3072 ? 128u : 256u);     // a tile = 16 bytes per thread
    int rc = V2P_OK;
    bool done = false;                                  // set only after the emit pass: a retry that runs out of attempts is an error
    std::string last_reason = "decode: retries exhausted";
    for (int attempt = 0; attempt < 4 && !done; ++attempt) {
        if (d->d_work) { (void)hipFree(d->d_work); d->d_work = nullptr; }
        const DecodeLayout L = decode_layout(n_records, n_samples, ovf_words);
        DTRY(dmalloc(reinterpret_cast<void**>(&d->d_work), L.total), "hipMalloc(decode workspace)");
        DTRY(hipMemsetAsync(d->d_status, 0xFF, sizeof(uint64_t), st), "hipMemset(status)");
        DecodeArgs a{};
        fill_args(a, d_text, n_text, d->d_rows, d->d_rows + n_records, n_records, n_samples, d->d_csq, d->d_csq + n_records + 1,
                  d->d_csq + 2 * n_records + 1, d->d_work, ovf_words, d->d_hap_begin, nullptr, ~0ull, d->d_status);
        a.parse_threads = parse_threads;
        DTRY(hipEventRecord(ev[0], st), "hipEventRecord");
        DTRY(launch_decode(a, st, 1u), "parse_rows_kernel");
        DTRY(hipEventRecord(ev[1], st), "hipEventRecord");
        DTRY(launch_decode(a, st, 2u), "count_kernel");
        DTRY(hipEventRecord(ev[2], st), "hipEventRecord");
        DTRY(launch_decode(a, st, 4u), "scan kernels");
        DTRY(hipEventRecord(ev[3], st), "hipEventRecord");
        uint64_t status[2] = {~0ull, 0};
        d->hap_begin.assign(n_haps + 1, 0);
        DTRY(hipMemcpyAsync(status, d->d_status, sizeof(status), hipMemcpyDeviceToHost, st), "D2H(status)");
        DTRY(hipMemcpyAsync(d->hap_begin.data(), d->d_hap_begin, (n_haps + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st), "D2H(hap_begin)");
        DTRY(hipStreamSynchronize(st), "hipStreamSynchronize");
        if (status[0] != ~0ull) {
            const uint32_t reason = uint32_t(status[0] & 0xFF);
            last_reason = std::string("decode: ") + reason_text(reason) + " (after a retry)";
            if (reason == DEC_CAPACITY && status[1] > ovf_words && status[1] < (1ull << 31)) { ovf_words = status[1]; continue; }
            if (reason == DEC_FIELD_TOO_LONG && parse_threads != 256u) { parse_threads = 256u; continue; }     // the narrow kernels look back 1-2 KiB only
            rc = ctx_fail(ctx, reason_to_code(reason), std::string("decode: ") + reason_text(reason) + " at record " +
                          std::to_string((status[0] >> 8) / n_samples) + ", sample " + std::to_string((status[0] >> 8) % n_samples),
                          int64_t(status[0] >> 8));
            break;
        }
        d->n_ids = d->hap_begin[n_haps];
        DTRY(dmalloc(reinterpret_cast<void**>(&d->d_ids), (d->n_ids + 64) * sizeof(uint32_t)), "hipMalloc(ids)");
        a.ids = d->d_ids; a.ids_capacity = d->n_ids;
        DTRY(launch_decode(a, st, 8u), "emit_kernel");
        DTRY(hipEventRecord(ev[4], st), "hipEventRecord");
        DTRY(hipStreamSynchronize(st), "hipStreamSynchronize");
        for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&d->ms[k], ev[k], ev[k + 1]);
        done = true;
    }
    if (rc == V2P_OK && !done) rc = ctx_fail(ctx, V2P_ERR_UNSUPPORTED, last_reason, -1);     // never hand back a decode whose emit pass did not run
    if (rc != V2P_OK) { d->release(); delete d; return rc; }
    *out = d;
    return V2P_OK;
}

int v2p_decode_counts(const v2p_decode* d, uint64_t* hap_begin)
{
    if (!d || !hap_begin) return V2P_ERR_INVALID_ARG;
    memcpy(hap_begin, d->hap_begin.data(), d->hap_begin.size() * sizeof(uint64_t));
    return V2P_OK;
}

int v2p_decode_download(v2p_decode* d, uint32_t* ids)
{
    if (!d) return V2P_ERR_INVALID_ARG;
    if (!d->n_ids) return V2P_OK;
    if (!ids) return V2P_ERR_INVALID_ARG;
    Guard g(d->ctx);
    (void)hipSetDevice(ctx_device(d->ctx));
    hipStream_t st = ctx_stream(d->ctx);
    hipError_t e = hipMemcpyAsync(ids, d->d_ids, d->n_ids * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e == hipSuccess ? V2P_OK : ctx_fail(d->ctx, V2P_ERR_HIP, std::string("D2H(ids): ") + hipGetErrorString(e), -1);
}

int v2p_decode_device(const v2p_decode* d, const uint64_t** d_hap_begin, const uint32_t** d_ids)
{
    if (!d || !d_hap_begin || !d_ids) return V2P_ERR_INVALID_ARG;
    *d_hap_begin = d->d_hap_begin;
    *d_ids = d->d_ids;
    return V2P_OK;
}

int v2p_decode_timing(const v2p_decode* d, float* ms_parse, float* ms_count, float* ms_scan, float* ms_emit)
{
    if (!d) return V2P_ERR_INVALID_ARG;
    if (ms_parse) *ms_parse = d->ms[0];
    if (ms_count) *ms_count = d->ms[1];
    if (ms_scan) *ms_scan = d->ms[2];
    if (ms_emit) *ms_emit = d->ms[3];
    return V2P_OK;
}

void v2p_decode_destroy(v2p_decode* d)
{
    if (!d) return;
    (void)hipSetDevice(ctx_device(d->ctx));
    d->release();
    delete d;
}

}  // extern "C"
