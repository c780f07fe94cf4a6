// v2p_api.hip -- the C ABI of include/vcf2prot_hip.h on top of the gfx950 kernels.
// Host side of the engine: packs reference-shaped Task vectors (gir.rs:283-299) into
// the device image (sir_pack.hpp), owns device memory, launches, maps device status
// words back to the reference's panic conditions.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vcf2prot_hip.h"
#include "sir_pack.hpp"
#include "stitch_kernels.h"
#include "build_kernels.h"
#include "build_rows.h"
#include "dense_pieces.h"
#ifdef V2P_BENCH_VARIANTS
#include "patch_image.h"
#include "bench/v2p_bench.h"
#endif
#include "v2p_ctx_internal.h"

using namespace v2p;

namespace {

thread_local std::string g_init_error;

// V2P_DEBUG_POISON=1 (a debugging aid like the reference's DEBUG_* switches; read once): every device buffer is filled with 0xA5 whenever a
// call (re)sizes it -- also when the allocation is reused -- so that nothing can lean on what fresh or recycled memory happens to hold
// (tools/fuzz_*.py and the GPU suite run clean under it; one bug of that kind was found without it)
static bool debug_poison()
{
    static const bool on = [] { const char* e = getenv("V2P_DEBUG_POISON"); return e && e[0] == '1'; }();
    return on;
}

struct DevBuf {
    uint8_t* base = nullptr;
    size_t cap = 0;
    void poison() const { if (debug_poison() && base) { (void)hipDeviceSynchronize(); (void)hipMemset(base, 0xA5, cap); (void)hipDeviceSynchronize(); } }
    // n usable bytes at ptr(), with at least PAD_BYTES readable before and after.  The front pad is a
    // whole 256 bytes so ptr() keeps hipMalloc's alignment: result arenas must start on a cache line
    // (a 1 KiB wave store that straddles lines turns into partial-line writes).
    static constexpr size_t FRONT = 256;
    hipError_t ensure(size_t n) {
        const size_t need = n + FRONT + PAD_BYTES;
        if (need <= cap) { poison(); return hipSuccess; }
        if (base) { (void)hipFree(base); base = nullptr; cap = 0; }
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), want);
        if (e != hipSuccess) { want = need; e = hipMalloc(reinterpret_cast<void**>(&base), want); }
        if (e == hipSuccess) { cap = want; poison(); }
        return e;
    }
    // scratch that lives for one call: exactly n bytes (ensure()'s 25 % slack is for buffers that grow from call to call)
    hipError_t ensure_exact(size_t n) {
        const size_t need = n + FRONT + PAD_BYTES;
        if (need <= cap) { poison(); return hipSuccess; }
        if (base) { (void)hipFree(base); base = nullptr; cap = 0; }
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), need);
        if (e == hipSuccess) { cap = need; poison(); }
        return e;
    }
    hipError_t ensure_os(bool grow, size_t n) { return grow ? ensure(n) : ensure_exact(n); }
    uint8_t* ptr() const { return base ? base + FRONT : nullptr; }
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; }
};

struct PinnedBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), n + n / 4 + 64, hipHostMallocDefault);
        if (e == hipSuccess) cap = n + n / 4 + 64;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

int reason_to_err(uint32_t reason)
{
    switch (reason) {
        case STATUS_BAD_CODE: return V2P_ERR_BAD_CODE;
        case STATUS_RES_OOB: return V2P_ERR_RES_OOB;
        case STATUS_SRC_OOB: return V2P_ERR_SRC_OOB;
        case STATUS_NOT_CONTIGUOUS: return V2P_ERR_NOT_CONTIGUOUS;
        case STATUS_TOO_MANY: return V2P_ERR_UNSUPPORTED;
        default: return V2P_ERR_INVALID_ARG;
    }
}

int pack_to_err(int ps)
{
    switch (ps) {
        case PACK_OK: return V2P_OK;
        case PACK_BAD_CODE: return V2P_ERR_BAD_CODE;
        case PACK_RES_OOB: return V2P_ERR_RES_OOB;
        case PACK_SRC_OOB: return V2P_ERR_SRC_OOB;
        case PACK_NOT_CANONICAL: return V2P_ERR_NOT_CANONICAL;
        default: return V2P_ERR_UNSUPPORTED;
    }
}

const char* err_name(int e)
{
    switch (e) {
        case V2P_ERR_BAD_CODE: return "unsupported stream code (exe_code not in {0,1})";
        case V2P_ERR_RES_OOB: return "task writes beyond the result tape";
        case V2P_ERR_SRC_OOB: return "task reads beyond its source tape";
        case V2P_ERR_NOT_CONTIGUOUS: return "start_pos_res does not equal the previous start_pos_res + length";
        case V2P_ERR_NOT_CANONICAL: return "result ranges overlap or are not ascending";
        case V2P_ERR_NON_BYTE_CHAR: return "char above 0xFF cannot enter the 1-byte image";
        default: return "error";
    }
}

}  // namespace

struct v2p_ctx {
    int device = 0;
    unsigned flags = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t exec_aux = nullptr;                    // the second launch stream of dual-stream phases
    hipEvent_t ev_exec[2] = {nullptr, nullptr};
    hipStream_t build_stream = nullptr, aux_stream = nullptr;   // v2p_batch_build_and_execute: the image is built here (aux: its compaction, next to the cutter) while the context's stream stitches
    mutable std::mutex mu;
    std::string err;
    int64_t err_index = -1;
    DevBuf proteome; uint64_t proteome_len = 0;      // resident reference: [proteome | FASTA record headers]
    uint64_t headers_len = 0;
    std::vector<uint8_t> headers_host;               // host copy of the header table (a few MB at most)
    // GIR-mode scratch (grow-only)
    DevBuf d_ref, d_alt, d_res, d_desc, d_chunks, d_soa, d_status;
    PinnedBuf h_stage, h_in;                         // pinned staging: results coming back / narrowed tapes going out
    ImageBuilder gir_img;                            // reused across v2p_execute_gir calls (its vectors keep their capacity)
    struct GirQueue* queue = nullptr;                // v2p_execute_gir_shared: batches of concurrent callers (created at the first call)
    v2p_launch_opts launch_opts{1u, 0u, 0ull, 0u, -1, 0u, 0u};   // v2p_set_launch_opts: phase size / threshold / store policy of this context's batches (A/B runs, tests)
    uint32_t variant = 0;                              // libv2p_bench.so only (v2p_bench_set_variant: csrc/bench/v2p_bench.h): the A/B switches of the builders and launchers

    int fail(int code, const std::string& msg, int64_t index = -1) { err = msg; err_index = index; return code; }
    int hip_fail(hipError_t e, const char* what) {
        return fail(V2P_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
};

constexpr uint32_t V2P_MAX_SLICES = 32;
struct v2p_batch {
    v2p_ctx* ctx = nullptr;
    ImageBuilder img;
    bool finalized = false;
    bool uses_proteome = false;
    // step-5-on-the-fly state (v2p_batch_begin_haplotype / add_transcript / end_haplotype)
    bool hap_open = false;
    uint64_t hap_res = 0;          // res_counter of haplotype_instruction.rs:90,132
    uint64_t hap_records = 0;
    uint64_t last_hdr_src = 0; uint32_t last_hdr_len = 0;
    DevBuf d_desc, d_chunks, d_payload, d_out, d_hap, d_digest, d_status;
    DevBuf d_build;                // the transcript stream and the builder's scratch (v2p_batch_build_on_device)
    DevBuf d_patch;                // PATCH images (patch_image.h): the substituted residues, PATCH_PATCH_CAP slots per chunk (d_desc holds the segments)
    bool is_patch = false;
    uint64_t patch_segs = 0, patch_patches = 0;
    // a TILE image (dense_pieces.h; round 6: what the one call builds for deep Task vectors): d_desc holds the tiles' piece slots, the tile
    // tables (count per tile in d_tiles; res_counter per tile there too, or the resident stream's) are the executor's work list
    bool is_tiles = false;
    uint32_t tile_slots = 0;
    uint64_t tiles_n = 0;
    const uint32_t* tiles_count = nullptr;
    const uint64_t* tiles_res_base = nullptr;
    const uint8_t* payload_dev = nullptr;   // the alt bytes the image's payload descriptors read: d_payload, or a resident v2p_stream's
    // ... in which case the batch is REGISTERED with that stream (gir.rs:197: GIR::execute(self) owns its tapes by move -- a dangling tape
    // cannot exist there; here the tape is the stream's): v2p_stream_destroy orphans the batches built from it, and an orphan's image is
    // never executed again (V2P_ERR_STATE) -- its arena, the batch's own memory, stays readable (download, digests)
    struct v2p_stream* from_stream = nullptr;
    bool orphaned = false;
    uint64_t pieces_src0_len = 0, pieces_src1_len = 0;   // the source lengths to_pieces validated the piece image against
    // v2p_batch_build_and_execute: tables that outlive the call so that a batch that is rebuilt recycles them, the slices' chunk ranges
    // and the events of the last call (read by v2p_batch_oneshot_info after a sync)
    DevBuf d_tiles, d_cover, d_pad, d_order;
    DevBuf d_pieces, d_chunks2;    // PIECE image (dense_pieces.h): what a dense rows image is executed from when it is executed AGAIN
    bool executed = false;         // the image has been executed at least once (the one call, or a v2p_batch_execute): the NEXT execute is a re-execute
    uint64_t n_pieces = 0;
    int pieces_state = 0;          // 0: not tried; 1: built (d_pieces / d_chunks2 describe the batch's image); -1: the image is not converted (kept on the dense kernel)
    PinnedBuf h_sum;               // the one call: the 64 bytes the host reads per slice (counts, flags, status)
    DevBuf d_stage;                // STAGED descriptors (stitch_kernels.h): two buffers of a phase's chunks x 64 slots, filled by the read-ahead
    // a PADDED wave image (sir_pack.hpp; what the one call leaves behind): d_desc holds ROWS_TILE_SLOTS slots per tile, the chunk records
    // address its slots; desc_slots = its size (the kernels' bound), n_desc the descriptors it holds; pad_tdbase = the scan of the tiles'
    // counts (in d_tiles), with which a download hands out the dense form
    bool pad_image = false;
    bool desc_swapped = false;     // densify() left the image in d_pad's allocation (swapped into d_desc): v2p_batch_reset swaps back
    uint64_t desc_slots = 0, pad_n_tiles = 0;
    uint32_t pad_K = 0;
    const uint64_t* pad_tdbase = nullptr;
    uint32_t n_slices = 0;
    uint64_t slice_chunk0[V2P_MAX_SLICES + 1] = {};
    uint64_t slice_desc[V2P_MAX_SLICES] = {}, slice_bytes[V2P_MAX_SLICES] = {};
    hipEvent_t ev_os[2 + 2 * V2P_MAX_SLICES] = {};
    hipEvent_t ev_aux[2] = {};
    hipEvent_t ev_par[V2P_MAX_SLICES] = {};   // a slice's parse and the scan of its tiles' counts are done (the slices' parses run ahead on a stream of their own)
    bool os_ahead = false;         // the last one call built its slices that way: build time = first parse .. last slice's chunk table
    float os_build_ms = 0.f;
    double os_wall_ms = 0.0;
    int os_kernel = 0;
    uint64_t n_desc = 0, n_chunks = 0, n_payload = 0, out_bytes = 0, n_haps = 0;
    uint32_t max_chunk_tasks = 0;
    int launch_hint = 0;           // stitch_launch_bits() of the chunk table
    bool grow = false;             // the one call sizes its buffers with slack and never shrinks them (a pipeline slot's batch: slice after slice of
                                   // about one size -- an exact-size buffer would be freed and allocated again, and hipFree waits for the device)
};

namespace v2p {
hipStream_t ctx_stream(v2p_ctx* c) { return c->stream; }
int ctx_device(v2p_ctx* c) { return c->device; }
int ctx_fail(v2p_ctx* c, int code, const std::string& msg, int64_t index) { return c->fail(code, msg, index); }
void ctx_lock(v2p_ctx* c) { c->mu.lock(); }
void ctx_unlock(v2p_ctx* c) { c->mu.unlock(); }
}  // namespace v2p

// The A/B switches (16 .. 29) of rounds 4-5 exist in the development library only; the product's rules are not switchable.
static inline uint32_t ctx_variant(const v2p_ctx* c)
{
#ifdef V2P_BENCH_VARIANTS
    return c->variant;
#else
    (void)c;
    return 0u;
#endif
}

#define HIP_TRY(ctx, expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return (ctx)->hip_fail(e__, what); } while (0)

struct GirQueue;
static void queue_destroy(v2p_ctx* c);
static void stream_attach(v2p_batch* b, const struct v2p_stream* st);
static void stream_detach(v2p_batch* b);
static void sync_ctx_streams(v2p_ctx* c);
namespace {
uint32_t narrow_chars(const uint32_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n);   // (AVX2 when the CPU has it: defined with the coalescing queue)
void widen_chars(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n);
}

extern "C" {

const char* v2p_version(void) { return "vcf2prot-hip 0.1.0 (gfx950)"; }

int v2p_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return V2P_ERR_HIP;
    return n;
}

int v2p_engine_from_str(const char* name, int* engine)
{
    if (!name || !engine) return V2P_ERR_INVALID_ARG;
    if (!strcmp(name, "st") || !strcmp(name, "ST")) { *engine = V2P_ENGINE_ST; return V2P_OK; }
    if (!strcmp(name, "mt") || !strcmp(name, "MT")) { *engine = V2P_ENGINE_MT; return V2P_OK; }
    if (!strcmp(name, "gpu") || !strcmp(name, "GPU")) { *engine = V2P_ENGINE_GPU; return V2P_OK; }
    g_init_error = std::string(name) + " is not a supported engine";   // engines.rs:27
    return V2P_ERR_INVALID_ARG;
}

int v2p_init(int device_ordinal, unsigned flags, v2p_ctx** out)
{
    if (!out) return V2P_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_init_error = "no HIP device available: the gpu engine needs an MI355X (there is no CPU fallback)";
        return V2P_ERR_HIP;
    }
    if (device_ordinal < 0 || device_ordinal >= n) { g_init_error = "device ordinal out of range"; return V2P_ERR_INVALID_ARG; }
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) { g_init_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return V2P_ERR_HIP; }
    v2p_ctx* c = new (std::nothrow) v2p_ctx();
    if (!c) return V2P_ERR_HIP;
    c->device = device_ordinal;
    c->flags = flags;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_init_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return V2P_ERR_HIP; }
    c->stream = c->own_stream;
    // The kernels' code objects now, not under the first batch's clock -- once per process and device, on the context's own stream
    // (a device-wide wait here would stall the batches other contexts have in flight on this GPU).
    {
        static std::mutex mu;
        static bool loaded[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        if (device_ordinal < 64 && !loaded[device_ordinal]) {
            hipError_t le = preload_stitch_kernels(c->own_stream);
            if (le == hipSuccess) le = preload_stitch_wave(c->own_stream);
            if (le == hipSuccess) le = preload_build_kernels(c->own_stream);
            if (le == hipSuccess) le = preload_build_rows(c->own_stream);
#ifdef V2P_BENCH_VARIANTS
            if (le == hipSuccess) le = preload_patch_image(c->own_stream);
#endif
            if (le == hipSuccess) le = preload_dense_pieces(c->own_stream);
            if (le == hipSuccess) le = hipStreamSynchronize(c->own_stream);
            if (le != hipSuccess) {
                g_init_error = std::string("loading the kernels' code objects: ") + hipGetErrorString(le);
                (void)hipStreamDestroy(c->own_stream);
                delete c;
                return V2P_ERR_HIP;
            }
            loaded[device_ordinal] = true;
        }
    }
    *out = c;
    return V2P_OK;
}

void v2p_destroy(v2p_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->proteome.release(); c->d_ref.release(); c->d_alt.release(); c->d_res.release();
    c->d_desc.release(); c->d_chunks.release(); c->d_soa.release(); c->d_status.release();
    c->h_stage.release(); c->h_in.release();
    queue_destroy(c);
    if (c->exec_aux) { (void)hipStreamSynchronize(c->exec_aux); (void)hipStreamDestroy(c->exec_aux); }
    for (hipEvent_t e : c->ev_exec) if (e) (void)hipEventDestroy(e);
    if (c->build_stream) (void)hipStreamDestroy(c->build_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* v2p_last_error(const v2p_ctx* c) { return c ? c->err.c_str() : g_init_error.c_str(); }
int64_t v2p_last_error_index(const v2p_ctx* c) { return c ? c->err_index : -1; }

int v2p_set_stream(v2p_ctx* c, void* hip_stream)
{
    if (!c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return V2P_OK;
}

int v2p_upload_reference(v2p_ctx* c, const uint8_t* aa, uint64_t n, const uint8_t* headers, uint64_t n_headers)
{
    if (!c || (!aa && n) || (!headers && n_headers)) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    HIP_TRY(c, c->proteome.ensure(n + n_headers), "hipMalloc(reference)");
    HIP_TRY(c, hipMemsetAsync(c->proteome.base, 0, c->proteome.cap, c->stream), "hipMemset(reference)");
    if (n) HIP_TRY(c, hipMemcpyAsync(c->proteome.ptr(), aa, n, hipMemcpyHostToDevice, c->stream), "H2D(proteome)");
    if (n_headers) HIP_TRY(c, hipMemcpyAsync(c->proteome.ptr() + n, headers, n_headers, hipMemcpyHostToDevice, c->stream), "H2D(headers)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    c->proteome_len = n;
    c->headers_len = n_headers;
    c->headers_host.assign(headers, headers + n_headers);
    return V2P_OK;
}

int v2p_upload_proteome(v2p_ctx* c, const uint8_t* aa, uint64_t n) { return v2p_upload_reference(c, aa, n, nullptr, 0); }

// reads the device status word after a sync; returns V2P_OK or the mapped task error
static int collect_status(v2p_ctx* c, DevBuf& d_status)
{
    unsigned long long st = STATUS_CLEAN;
    HIP_TRY(c, hipMemcpyAsync(&st, d_status.ptr(), sizeof(st), hipMemcpyDeviceToHost, c->stream), "D2H(status)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st == STATUS_CLEAN) return V2P_OK;
    const int code = reason_to_err(uint32_t(st & 0xFFu));
    return c->fail(code, std::string("device: ") + err_name(code) + " at descriptor " + std::to_string(st >> 8), int64_t(st >> 8));
}

static int init_status(v2p_ctx* c, DevBuf& d_status)
{
    HIP_TRY(c, d_status.ensure(sizeof(unsigned long long)), "hipMalloc(status)");
    HIP_TRY(c, hipMemsetAsync(d_status.ptr(), 0xFF, sizeof(unsigned long long), c->stream), "hipMemset(status)");
    return V2P_OK;
}

int v2p_validate_gir(v2p_ctx* c,
                     const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                     const uint64_t* start_pos_res, uint64_t n_tasks,
                     uint64_t n_ref, uint64_t n_alt, uint64_t n_res,
                     int64_t* first_bad, int* reason)
{
    if (!c || !first_bad || !reason || (n_tasks && (!code || !start_pos || !length || !start_pos_res))) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    *first_bad = -1; *reason = 0;
    if (n_tasks == 0) return V2P_OK;
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    const size_t n8 = size_t(n_tasks) * 8, ncode = (size_t(n_tasks) + 7) & ~size_t(7);
    HIP_TRY(c, c->d_soa.ensure(3 * n8 + ncode), "hipMalloc(soa)");
    uint8_t* d = c->d_soa.ptr();
    uint64_t* d_sp = reinterpret_cast<uint64_t*>(d);
    uint64_t* d_ln = reinterpret_cast<uint64_t*>(d + n8);
    uint64_t* d_sr = reinterpret_cast<uint64_t*>(d + 2 * n8);
    uint8_t* d_code = d + 3 * n8;
    HIP_TRY(c, hipMemcpyAsync(d_sp, start_pos, n8, hipMemcpyHostToDevice, c->stream), "H2D(start_pos)");
    HIP_TRY(c, hipMemcpyAsync(d_ln, length, n8, hipMemcpyHostToDevice, c->stream), "H2D(length)");
    HIP_TRY(c, hipMemcpyAsync(d_sr, start_pos_res, n8, hipMemcpyHostToDevice, c->stream), "H2D(start_pos_res)");
    HIP_TRY(c, hipMemcpyAsync(d_code, code, n_tasks, hipMemcpyHostToDevice, c->stream), "H2D(code)");
    int rc = init_status(c, c->d_status);
    if (rc) return rc;
    ValidateArgs a{d_code, d_sp, d_ln, d_sr, n_tasks, n_ref, n_alt, n_res,
                   reinterpret_cast<unsigned long long*>(c->d_status.ptr())};
    HIP_TRY(c, launch_validate(a, c->stream), "launch(validate)");
    unsigned long long st = STATUS_CLEAN;
    HIP_TRY(c, hipMemcpyAsync(&st, c->d_status.ptr(), sizeof(st), hipMemcpyDeviceToHost, c->stream), "D2H(status)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st != STATUS_CLEAN) { *first_bad = int64_t(st >> 8); *reason = reason_to_err(uint32_t(st & 0xFFu)); }
    return V2P_OK;
}

int v2p_execute_gir(v2p_ctx* c,
                    const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                    const uint64_t* start_pos_res, uint64_t n_tasks,
                    const uint32_t* ref, uint64_t n_ref,
                    const uint32_t* alt, uint64_t n_alt,
                    uint32_t* res, uint64_t n_res)
{
    if (!c) return V2P_ERR_INVALID_ARG;
    if ((n_tasks && (!code || !start_pos || !length || !start_pos_res)) || (n_ref && !ref) || (n_alt && !alt) || (n_res && !res))
        return c->fail(V2P_ERR_INVALID_ARG, "null argument");
    if (c->flags & V2P_FLAG_DEBUG_GPU) {          // gir.rs:203-229 / README.md:156-157
        int64_t bad = -1; int reason = 0;
        int rc = v2p_validate_gir(c, code, start_pos, length, start_pos_res, n_tasks, n_ref, n_alt, n_res, &bad, &reason);
        if (rc) return rc;
        if (bad >= 0) {
            std::lock_guard<std::mutex> lk(c->mu);
            return c->fail(reason, std::string("DEBUG_GPU: ") + err_name(reason) + " at row " + std::to_string(bad), bad);
        }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (n_tasks == 0) return V2P_OK;
    // bounds of every task first: the reference would panic, nothing may be written out of range
    bool canonical = true;
    uint64_t cursor = 0;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        const uint64_t n_src = code[i] == 0 ? n_ref : n_alt;      // task.rs:42-49
        if (start_pos_res[i] + length[i] > n_res || start_pos_res[i] + length[i] < length[i])
            return c->fail(V2P_ERR_RES_OOB, std::string(err_name(V2P_ERR_RES_OOB)) + " at row " + std::to_string(i), int64_t(i));
        if (start_pos[i] + length[i] > n_src || start_pos[i] + length[i] < length[i])
            return c->fail(V2P_ERR_SRC_OOB, std::string(err_name(V2P_ERR_SRC_OOB)) + " at row " + std::to_string(i), int64_t(i));
        if (start_pos_res[i] < cursor) canonical = false;
        cursor = start_pos_res[i] + length[i];
    }
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    // Char width: the boundary speaks Rust chars (u32), the device 1 byte per residue.  Amino-acid alphabets are
    // ASCII, so the tapes are narrowed on the host straight into pinned staging (4x fewer bytes over PCIe, no pageable bounce)
    // and the result is widened on the way back; a tape holding a char above 0xFF takes the 4-byte path below.
    uint64_t E = 1;
    {
        HIP_TRY(c, c->h_in.ensure(((n_ref + 15) & ~uint64_t(15)) + n_alt + 64), "hipHostMalloc(in)");
        uint8_t* const h8 = c->h_in.p;
        uint8_t* const a8 = h8 + ((n_ref + 15) & ~uint64_t(15));
        const uint32_t seen = narrow_chars(ref, h8, n_ref) | narrow_chars(alt, a8, n_alt);
        if (seen > 0xFFu) E = sizeof(uint32_t);
    }
    HIP_TRY(c, c->d_ref.ensure(n_ref * E), "hipMalloc(ref)");
    HIP_TRY(c, c->d_alt.ensure(n_alt * E), "hipMalloc(alt)");
    HIP_TRY(c, c->d_res.ensure(n_res * E), "hipMalloc(res)");
    if (E == 1) {
        const uint8_t* h8 = c->h_in.p;
        if (n_ref) HIP_TRY(c, hipMemcpyAsync(c->d_ref.ptr(), h8, n_ref, hipMemcpyHostToDevice, c->stream), "H2D(ref)");
        if (n_alt) HIP_TRY(c, hipMemcpyAsync(c->d_alt.ptr(), h8 + ((n_ref + 15) & ~uint64_t(15)), n_alt, hipMemcpyHostToDevice, c->stream), "H2D(alt)");
    } else {
        if (n_ref) HIP_TRY(c, hipMemcpyAsync(c->d_ref.ptr(), ref, n_ref * E, hipMemcpyHostToDevice, c->stream), "H2D(ref)");
        if (n_alt) HIP_TRY(c, hipMemcpyAsync(c->d_alt.ptr(), alt, n_alt * E, hipMemcpyHostToDevice, c->stream), "H2D(alt)");
    }
    int rc = init_status(c, c->d_status);
    if (rc) return rc;
    HIP_TRY(c, c->h_stage.ensure(n_res * E + 16), "hipHostMalloc(stage)");

    if (canonical) {
        ImageBuilder& img = c->gir_img;
        img.reset();
        img.inline_payload = false;                 // the tapes are this call's own device buffers: plain descriptors
        img.fuse_snv = false;
        img.desc.reserve(n_tasks + 16);
        for (uint64_t i = 0; i < n_tasks; ++i) {
            const int ps = img.add_task(code[i] == 0 ? SPACE_PROTEOME : SPACE_PAYLOAD, start_pos[i] * E, length[i] * E,
                                        start_pos_res[i] * E, n_res * E);
            if (ps != PACK_OK) return c->fail(pack_to_err(ps), "pack failed", int64_t(i));
        }
        img.end_haplotype(n_res * E);
        img.finish();
        HIP_TRY(c, c->d_desc.ensure(img.desc.size() * 8), "hipMalloc(desc)");
        HIP_TRY(c, c->d_chunks.ensure(img.chunks.size() * sizeof(Chunk)), "hipMalloc(chunks)");
        HIP_TRY(c, hipMemcpyAsync(c->d_desc.ptr(), img.desc.data(), img.desc.size() * 8, hipMemcpyHostToDevice, c->stream), "H2D(desc)");
        HIP_TRY(c, hipMemcpyAsync(c->d_chunks.ptr(), img.chunks.data(), img.chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice, c->stream), "H2D(chunks)");
        StitchArgs a{reinterpret_cast<const uint64_t*>(c->d_desc.ptr()), img.desc.size(), reinterpret_cast<const Chunk*>(c->d_chunks.ptr()),
                     uint32_t(img.chunks.size()), c->d_ref.ptr(), n_ref * E, c->d_alt.ptr(), n_alt * E,
                     c->d_res.ptr(), n_res * E, reinterpret_cast<unsigned long long*>(c->d_status.ptr())};
        HIP_TRY(c, launch_stitch(a, c->stream, int(!(c->flags & V2P_FLAG_TEMPORAL)) | stitch_launch_bits(img.chunks.data(), img.chunks.size()), 0), "launch(stitch)");
    } else {
        // ordered path: overlapping / descending result ranges, executed in task order
        const size_t n8 = size_t(n_tasks) * 8, ncode = (size_t(n_tasks) + 7) & ~size_t(7);
        HIP_TRY(c, c->d_soa.ensure(3 * n8 + ncode), "hipMalloc(soa)");
        uint8_t* d = c->d_soa.ptr();
        HIP_TRY(c, hipMemcpyAsync(d, start_pos, n8, hipMemcpyHostToDevice, c->stream), "H2D(start_pos)");
        HIP_TRY(c, hipMemcpyAsync(d + n8, length, n8, hipMemcpyHostToDevice, c->stream), "H2D(length)");
        HIP_TRY(c, hipMemcpyAsync(d + 2 * n8, start_pos_res, n8, hipMemcpyHostToDevice, c->stream), "H2D(start_pos_res)");
        HIP_TRY(c, hipMemcpyAsync(d + 3 * n8, code, n_tasks, hipMemcpyHostToDevice, c->stream), "H2D(code)");
        OrderedArgs oa{d + 3 * n8, reinterpret_cast<const uint64_t*>(d), reinterpret_cast<const uint64_t*>(d + n8),
                       reinterpret_cast<const uint64_t*>(d + 2 * n8), n_tasks, c->d_ref.ptr(), c->d_alt.ptr(), c->d_res.ptr(), E};
        HIP_TRY(c, launch_ordered(oa, c->stream), "launch(ordered)");
    }
    // the tape comes back to pinned staging; only cells some task covers go to the caller (the others keep the caller's content:
    // haplotype_instruction.rs:78 filled them with '.')
    if (n_res) HIP_TRY(c, hipMemcpyAsync(c->h_stage.p, c->d_res.ptr(), n_res * E, hipMemcpyDeviceToHost, c->stream), "D2H(res)");
    rc = collect_status(c, c->d_status);
    if (rc) return rc;
    if (E == 1) {
        const uint8_t* st = c->h_stage.p;
        for (uint64_t i = 0; i < n_tasks;) {                               // runs of tasks that tile the tape are widened in one sweep
            const uint64_t b = start_pos_res[i];
            uint64_t e = b + length[i];
            for (++i; i < n_tasks && start_pos_res[i] == e; ++i) e += length[i];
            widen_chars(st + b, res + b, e - b);
        }
    } else {
        const uint32_t* st = reinterpret_cast<const uint32_t*>(c->h_stage.p);
        for (uint64_t i = 0; i < n_tasks; ++i)
            memcpy(res + start_pos_res[i], st + start_pos_res[i], size_t(length[i]) * sizeof(uint32_t));
    }
    return V2P_OK;
}

// ---- concurrent GIR::execute callers on ONE context: coalesced batches ------------------------------------------------
// The reference enters GIR::execute from every Rayon worker (parts/exec.rs:36-39, personalized_genome.rs:64-65).  One context per
// worker (v2p_execute_gir) gives every haplotype its own three PCIe transfers, launch and synchronisation.  Here the workers share a
// context: a call joins the open batch -- it reserves its ranges in the batch's pinned staging buffers, narrows its own tapes and
// writes its own descriptors and chunks there, all on the calling thread, in parallel with the other callers -- and the first caller of
// a batch (its leader) waits a short window for company, then issues ONE upload, ONE launch over the concatenated image and ONE
// download for all of them; every caller widens its own result back.  Up to three batches are alive, each on its own stream, so the
// upload of one overlaps the download of the one before.
namespace {

// The char loops are the host cost of a GIR (4 bytes per residue in, 4 out): AVX2 when the CPU has it (resolved once), streaming
// stores on the way back so that the 32-bit result tape is written without being read first.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2")))
static uint32_t narrow_chars_avx2(const uint32_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    __m256i acc = _mm256_setzero_si256();
    const __m256i fix = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(in + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(in + i + 8));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(in + i + 16)), d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(in + i + 24));
        acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)));
        const __m256i ab = _mm256_packus_epi32(a, b), cd = _mm256_packus_epi32(c, d);           // (lanes interleaved: fixed by the permute)
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(out + i), _mm256_permutevar8x32_epi32(_mm256_packus_epi16(ab, cd), fix));
    }
    alignas(32) uint32_t t[8];
    _mm256_store_si256(reinterpret_cast<__m256i*>(t), acc);
    uint32_t seen = t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7];
    for (; i < n; ++i) { seen |= in[i]; out[i] = uint8_t(in[i]); }
    return seen;
}
__attribute__((target("avx2")))
static void widen_chars_avx2(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n)
{
    uint64_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(out + i) & 31u); ++i) out[i] = in[i];
    for (; i + 32 <= n; i += 32) {
        for (int k = 0; k < 4; ++k)
            _mm256_stream_si256(reinterpret_cast<__m256i*>(out + i + 8 * k), _mm256_cvtepu8_epi32(_mm_loadl_epi64(reinterpret_cast<const __m128i*>(in + i + 8 * k))));
    }
    for (; i < n; ++i) out[i] = in[i];
    _mm_sfence();
}
static const bool g_avx2 = __builtin_cpu_supports("avx2");
#else
static const bool g_avx2 = false;
static uint32_t narrow_chars_avx2(const uint32_t*, uint8_t*, uint64_t) { return 0; }
static void widen_chars_avx2(const uint8_t*, uint32_t*, uint64_t) {}
#endif
uint32_t narrow_chars(const uint32_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n)
{
    if (g_avx2) return narrow_chars_avx2(in, out, n);
    uint32_t seen = 0;
    for (uint64_t i = 0; i < n; ++i) { seen |= in[i]; out[i] = uint8_t(in[i]); }
    return seen;
}
void widen_chars(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n)
{
    if (g_avx2) { widen_chars_avx2(in, out, n); return; }
    for (uint64_t i = 0; i < n; ++i) out[i] = in[i];
}

struct GirBatch {
    enum State { FREE, OPEN, CLOSED, DONE };
    State state = FREE;
    uint32_t n_reqs = 0, n_ready = 0, n_left = 0;
    uint64_t in_bytes = 0, res_bytes = 0, n_desc = 0, n_chunks = 0;
    std::chrono::steady_clock::time_point opened;
    int rc = V2P_OK;
    std::string err;
    PinnedBuf h_in, h_desc, h_chunks, h_out;
    DevBuf d_in, d_desc, d_chunks, d_out, d_status;
    hipStream_t stream = nullptr;
};

}  // namespace

struct GirQueue {
    static constexpr int N_BATCH = 8;
    int n_batch = 8;                                 // batches alive at once (V2P_COALESCE_BATCHES, <= 8): one filling, the others on the GPU or being read back
    std::mutex mu;
    std::condition_variable cv;
    GirBatch batch[N_BATCH];
    GirBatch* open = nullptr;
    uint64_t cap_bytes = 16ull << 20;                // tape bytes (and result bytes) one batch takes: V2P_COALESCE_MB (small batches, many in flight: 16 workers on C2
                                                     // reach 1.0e10 aa/s with 128 MB x 4, 1.75e10 with 32 MB x 8, 2.1e10 with 16 MB x 8)
    uint32_t window_us = 100;                        // how long a leader waits for company: V2P_COALESCE_US
    uint64_t n_batches = 0, n_joined = 0;            // statistics (v2p_coalesce_stats)
    std::atomic<uint64_t> ns_pack{0}, ns_join{0}, ns_stage{0}, ns_wait{0}, ns_widen{0}, ns_gpu{0};   // V2P_COALESCE_PROFILE: where the callers' time goes
    std::atomic<uint64_t> ns_b_setup{0}, ns_b_h2d{0}, ns_b_launch{0}, ns_b_d2h{0}, ns_b_sync{0}, ns_b_window{0}, ns_b_strag{0};   // ... and the leader's
    std::vector<std::thread> runners;                // one per batch slot: closes its batch when the gathering window ends, uploads, launches, downloads, waits
    bool stop = false;                               // (nobody's GIR::execute call does that any more: v2p_gir_submit returns once its share is staged)
    uint64_t desc_cap() const { return cap_bytes / 16; }        // descriptors (8 B each: half the tape bytes at 16 result bytes per task)
    uint64_t chunk_cap() const { return cap_bytes / 256; }      // chunk records
};

static void queue_destroy(v2p_ctx* c)
{
    GirQueue* q = c->queue;
    if (!q) return;
    { std::lock_guard<std::mutex> lk(q->mu); q->stop = true; }
    q->cv.notify_all();
    for (auto& t : q->runners) if (t.joinable()) t.join();
    if (getenv("V2P_COALESCE_PROFILE"))
        fprintf(stderr, "coalesce: %llu calls in %llu batches; per call ms: pack %.3f join %.3f stage %.3f wait %.3f widen %.3f; gpu per batch %.3f\n",
                (unsigned long long)q->n_joined, (unsigned long long)q->n_batches, q->ns_pack / 1e6 / double(q->n_joined ? q->n_joined : 1),
                q->ns_join / 1e6 / double(q->n_joined ? q->n_joined : 1), q->ns_stage / 1e6 / double(q->n_joined ? q->n_joined : 1),
                q->ns_wait / 1e6 / double(q->n_joined ? q->n_joined : 1), q->ns_widen / 1e6 / double(q->n_joined ? q->n_joined : 1),
                q->ns_gpu / 1e6 / double(q->n_batches ? q->n_batches : 1));
    if (getenv("V2P_COALESCE_PROFILE")) {
        const double nb = double(q->n_batches ? q->n_batches : 1) * 1e6;
        fprintf(stderr, "coalesce leader per batch ms: window %.3f stragglers %.3f | setup %.3f h2d-enqueue %.3f launch %.3f d2h-enqueue %.3f sync %.3f\n",
                q->ns_b_window / nb, q->ns_b_strag / nb, q->ns_b_setup / nb, q->ns_b_h2d / nb, q->ns_b_launch / nb, q->ns_b_d2h / nb, q->ns_b_sync / nb);
    }
    for (auto& b : q->batch) {
        if (b.stream) { (void)hipStreamSynchronize(b.stream); (void)hipStreamDestroy(b.stream); }
        b.h_in.release(); b.h_desc.release(); b.h_chunks.release(); b.h_out.release();
        b.d_in.release(); b.d_desc.release(); b.d_chunks.release(); b.d_out.release(); b.d_status.release();
    }
    delete q;
    c->queue = nullptr;
}

// the leader's part: one upload, one launch, one download for the whole batch (no lock held)
static void batch_run(v2p_ctx* c, GirBatch& b)
{
    auto hip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && b.rc == V2P_OK) { b.rc = V2P_ERR_HIP; b.err = std::string(what) + ": " + hipGetErrorString(e); }
        return e == hipSuccess;
    };
    using clk = std::chrono::steady_clock;
    GirQueue& qq = *c->queue;
    clk::time_point tp = clk::now();
    auto lap = [&](std::atomic<uint64_t>& acc) { const clk::time_point n = clk::now(); acc += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(n - tp).count()); tp = n; };
    // (hipSetDevice once per host thread and device: with sixteen leaders in the runtime at once the call cost 0.3 ms of every batch)
    thread_local int device_set = -1;
    if (device_set != c->device) { if (!hip(hipSetDevice(c->device), "hipSetDevice")) return; device_set = c->device; }
    if (!b.stream && !hip(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking), "hipStreamCreate")) return;
    // (device buffers at the batch's full capacity, once: growing them batch by batch is a hipFree + hipMalloc -- a device-wide stall -- each time)
    const GirQueue& q = *c->queue;
    if (!hip(b.d_in.ensure(q.cap_bytes), "hipMalloc(in)") || !hip(b.d_desc.ensure(q.desc_cap() * 8 + 64), "hipMalloc(desc)") ||
        !hip(b.d_chunks.ensure(q.chunk_cap() * sizeof(Chunk)), "hipMalloc(chunks)") || !hip(b.d_out.ensure(q.cap_bytes), "hipMalloc(out)") ||
        !hip(b.d_status.ensure(sizeof(unsigned long long)), "hipMalloc(status)")) return;
    lap(qq.ns_b_setup);
    if (!hip(hipMemsetAsync(b.d_status.ptr(), 0xFF, sizeof(unsigned long long), b.stream), "hipMemset(status)")) return;
    if (b.in_bytes && !hip(hipMemcpyAsync(b.d_in.ptr(), b.h_in.p, b.in_bytes, hipMemcpyHostToDevice, b.stream), "H2D(tapes)")) return;
    if (b.n_desc && !hip(hipMemcpyAsync(b.d_desc.ptr(), b.h_desc.p, b.n_desc * 8, hipMemcpyHostToDevice, b.stream), "H2D(desc)")) return;
    if (b.n_chunks && !hip(hipMemcpyAsync(b.d_chunks.ptr(), b.h_chunks.p, b.n_chunks * sizeof(Chunk), hipMemcpyHostToDevice, b.stream), "H2D(chunks)")) return;
    lap(qq.ns_b_h2d);
    if (b.n_chunks) {
        const Chunk* hc = reinterpret_cast<const Chunk*>(b.h_chunks.p);
        StitchArgs a{reinterpret_cast<const uint64_t*>(b.d_desc.ptr()), b.n_desc, reinterpret_cast<const Chunk*>(b.d_chunks.ptr()), uint32_t(b.n_chunks),
                     b.d_in.ptr(), b.in_bytes, b.d_in.ptr(), b.in_bytes,          // both source spaces are the one staging blob
                     b.d_out.ptr(), b.res_bytes, reinterpret_cast<unsigned long long*>(b.d_status.ptr())};
        if (!hip(launch_stitch(a, b.stream, int(!(c->flags & V2P_FLAG_TEMPORAL)) | stitch_launch_bits(hc, b.n_chunks), 0), "launch(stitch)")) return;
    }
    lap(qq.ns_b_launch);
    if (b.res_bytes && !hip(hipMemcpyAsync(b.h_out.p, b.d_out.ptr(), b.res_bytes, hipMemcpyDeviceToHost, b.stream), "D2H(results)")) return;
    // (the status word comes back into the batch's pinned result staging, behind the results: a copy to pageable memory is a blocking one)
    unsigned long long* const h_st = reinterpret_cast<unsigned long long*>(b.h_out.p + ((b.res_bytes + 15) & ~15ull));
    if (!hip(hipMemcpyAsync(h_st, b.d_status.ptr(), sizeof *h_st, hipMemcpyDeviceToHost, b.stream), "D2H(status)")) return;
    lap(qq.ns_b_d2h);
    if (!hip(hipStreamSynchronize(b.stream), "hipStreamSynchronize")) return;
    lap(qq.ns_b_sync);
    const unsigned long long st = *h_st;
    if (st != STATUS_CLEAN) {                            // (every task was bounds-checked by its caller: this is an engine fault, not an input error)
        b.rc = reason_to_err(uint32_t(st & 0xFFu));
        b.err = std::string("device: ") + err_name(b.rc) + " at descriptor " + std::to_string(st >> 8) + " of a coalesced batch";
    }
}

// A call in flight between v2p_gir_submit and v2p_gir_collect
struct v2p_gir_ticket {
    GirBatch* b = nullptr;                       // nullptr: executed inside submit (ordered / oversized GIRs) -- rc is final
    int rc = V2P_OK;
    int64_t row = -1;
    std::string err;
    bool wide = false;
    uint64_t res_off = 0;
    // the caller's arrays (they stay the caller's, valid until collect)
    const uint64_t* code = nullptr; const uint64_t* start_pos = nullptr; const uint64_t* length = nullptr; const uint64_t* start_pos_res = nullptr;
    uint64_t n_tasks = 0;
    const uint32_t* ref = nullptr; uint64_t n_ref = 0; const uint32_t* alt = nullptr; uint64_t n_alt = 0;
    uint32_t* res = nullptr; uint64_t n_res = 0;
};

// the runner of batch slot k: everything a batch needs after its members staged their shares
static void queue_runner(v2p_ctx* c, GirQueue* q, int k)
{
    using clk = std::chrono::steady_clock;
    auto ns_since = [](clk::time_point t) { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t).count()); };
    GirBatch* b = &q->batch[k];
    std::unique_lock<std::mutex> lk(q->mu);
    for (;;) {
        q->cv.wait(lk, [&] { return q->stop || b->state == GirBatch::OPEN; });
        if (q->stop) return;
        const auto deadline = b->opened + std::chrono::microseconds(q->window_us);
        const clk::time_point tw = clk::now();
        while (!q->stop && q->open == b && clk::now() < deadline) q->cv.wait_until(lk, deadline);
        if (q->open == b) q->open = nullptr;
        b->state = GirBatch::CLOSED;
        q->ns_b_window += ns_since(tw);
        const clk::time_point ts = clk::now();
        while (b->n_ready < b->n_reqs) q->cv.wait(lk);
        q->ns_b_strag += ns_since(ts);
        lk.unlock();
        const clk::time_point tg = clk::now();
        if (b->rc == V2P_OK) batch_run(c, *b);
        q->ns_gpu += ns_since(tg);
        lk.lock();
        b->state = GirBatch::DONE;
        q->cv.notify_all();
    }
}

static int gir_solo(v2p_ctx* c, v2p_gir_ticket& t)
{
    std::vector<uint8_t> c8(t.n_tasks);
    for (uint64_t i = 0; i < t.n_tasks; ++i) c8[i] = uint8_t(t.code[i]);
    const int rc = v2p_execute_gir(c, c8.data(), t.start_pos, t.length, t.start_pos_res, t.n_tasks, t.ref, t.n_ref, t.alt, t.n_alt, t.res, t.n_res);
    if (rc != V2P_OK) t.row = v2p_last_error_index(c);
    return rc;
}

// may_wait: block until a batch slot is free (only a caller that holds no uncollected ticket may: a slot is freed by the collects of
// its members); else return V2P_BUSY
static int gir_submit_impl(v2p_ctx* c,
                           const uint64_t* code, const uint64_t* start_pos, const uint64_t* length,
                           const uint64_t* start_pos_res, uint64_t n_tasks,
                           const uint32_t* ref, uint64_t n_ref, const uint32_t* alt, uint64_t n_alt,
                           uint32_t* res, uint64_t n_res, v2p_gir_ticket** ticket, bool may_wait)
{
    if (!c || !ticket) return V2P_ERR_INVALID_ARG;
    *ticket = nullptr;
    auto fail = [&](int code_, const std::string& msg, int64_t row) {
        std::lock_guard<std::mutex> lk(c->mu);
        return c->fail(code_, msg, row);
    };
    if ((n_tasks && (!code || !start_pos || !length || !start_pos_res)) || (n_ref && !ref) || (n_alt && !alt) || (n_res && !res))
        return fail(V2P_ERR_INVALID_ARG, "null argument", -1);
    if (!may_wait) {
        // v2p_gir_submit: when every batch of the queue is in flight the answer is V2P_BUSY -- BEFORE this call validates, marshals and
        // packs its GIR (round 4 found out after all of that and threw the work away; a worker that resubmits in a loop redid it every
        // time, exactly when the queue was saturated).  A hint: the join below decides.
        GirQueue* q0 = nullptr;
        { std::lock_guard<std::mutex> lk(c->mu); q0 = c->queue; }
        if (q0) {
            std::lock_guard<std::mutex> lk(q0->mu);
            bool room = q0->open != nullptr;
            for (int k = 0; k < q0->n_batch && !room; ++k) room = q0->batch[k].state == GirBatch::FREE;
            if (!room) return V2P_BUSY;
        }
    }
    std::unique_ptr<v2p_gir_ticket> t(new (std::nothrow) v2p_gir_ticket());
    if (!t) return fail(V2P_ERR_HIP, "out of host memory", -1);
    t->code = code; t->start_pos = start_pos; t->length = length; t->start_pos_res = start_pos_res; t->n_tasks = n_tasks;
    t->ref = ref; t->n_ref = n_ref; t->alt = alt; t->n_alt = n_alt; t->res = res; t->n_res = n_res;
    if (n_tasks == 0) { *ticket = t.release(); return V2P_OK; }
    // every task's bounds first, on the calling thread: the reference would panic (haplotype_instruction.rs:154, task.rs:42-49)
    bool canonical = true;
    uint64_t cursor = 0;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        int bad = 0;
        const uint64_t n_src = code[i] == 0 ? n_ref : n_alt;
        if (code[i] > 1) bad = V2P_ERR_BAD_CODE;
        else if (start_pos_res[i] + length[i] > n_res || start_pos_res[i] + length[i] < length[i]) bad = V2P_ERR_RES_OOB;
        else if (start_pos[i] + length[i] > n_src || start_pos[i] + length[i] < length[i]) bad = V2P_ERR_SRC_OOB;
        if (bad) { t->rc = bad; t->row = int64_t(i); t->err = std::string(err_name(bad)) + " at row " + std::to_string(i); *ticket = t.release(); return V2P_OK; }   // (reported by collect, like a panic inside GIR::execute)
        if (start_pos_res[i] < cursor) canonical = false;
        cursor = start_pos_res[i] + length[i];
    }
    GirQueue* q;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->queue) {
            c->queue = new (std::nothrow) GirQueue();
            if (c->queue) {
                if (const char* e = getenv("V2P_COALESCE_MB")) { const uint64_t mb = strtoull(e, nullptr, 10); if (mb >= 1 && mb <= 16384) c->queue->cap_bytes = mb << 20; }
                if (const char* e = getenv("V2P_COALESCE_US")) c->queue->window_us = uint32_t(strtoul(e, nullptr, 10));
                if (const char* e = getenv("V2P_COALESCE_BATCHES")) { const int k = atoi(e); if (k >= 1 && k <= GirQueue::N_BATCH) c->queue->n_batch = k; }
                // every batch's staging, device buffers, stream and runner now, by the first caller: a batch that allocates at its first use
                // does so inside somebody's call (pinned staging alone is tens of milliseconds per batch)
                GirQueue& q0 = *c->queue;
                hipError_t e = hipSetDevice(c->device);
                for (int k = 0; k < q0.n_batch && e == hipSuccess; ++k) {
                    GirBatch& bb = q0.batch[k];
                    if (e == hipSuccess) e = bb.h_in.ensure(q0.cap_bytes);
                    if (e == hipSuccess) e = bb.h_out.ensure(q0.cap_bytes + 64);
                    if (e == hipSuccess) e = bb.h_desc.ensure(q0.desc_cap() * 8);
                    if (e == hipSuccess) e = bb.h_chunks.ensure(q0.chunk_cap() * sizeof(Chunk));
                    if (e == hipSuccess) e = bb.d_in.ensure(q0.cap_bytes);
                    if (e == hipSuccess) e = bb.d_out.ensure(q0.cap_bytes);
                    if (e == hipSuccess) e = bb.d_desc.ensure(q0.desc_cap() * 8 + 64);
                    if (e == hipSuccess) e = bb.d_chunks.ensure(q0.chunk_cap() * sizeof(Chunk));
                    if (e == hipSuccess) e = bb.d_status.ensure(sizeof(unsigned long long));
                    if (e == hipSuccess) e = hipStreamCreateWithFlags(&bb.stream, hipStreamNonBlocking);
                }
                if (e != hipSuccess) { queue_destroy(c); return c->fail(V2P_ERR_HIP, std::string("coalescing queue: ") + hipGetErrorString(e)); }
                for (int k = 0; k < q0.n_batch; ++k) q0.runners.emplace_back(queue_runner, c, c->queue, k);
            }
        }
        q = c->queue;
    }
    if (!q) return fail(V2P_ERR_HIP, "out of host memory", -1);
    const uint64_t ref_room = (n_ref + 15) & ~15ull, in_need = ref_room + ((n_alt + 15) & ~15ull), res_need = (n_res + 4095) & ~4095ull;
    // ordered / oversized GIRs: the one-haplotype path, serialised on the context, inside this call
    if (!canonical || in_need > q->cap_bytes || res_need > q->cap_bytes) { t->rc = gir_solo(c, *t); if (t->rc) t->err = v2p_last_error(c); *ticket = t.release(); return V2P_OK; }

    using clk = std::chrono::steady_clock;
    auto ns_since = [](clk::time_point tt) { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tt).count()); };
    clk::time_point tp = clk::now();
    // the caller's own image, cut as if its result began the arena: batch offsets are multiples of 4 KiB, so the cuts keep their alignment
    thread_local ImageBuilder img;
    img.reset();
    img.inline_payload = false;                          // the tapes travel with the batch: plain descriptors
    img.fuse_snv = false;
    img.desc.reserve(n_tasks + n_tasks / 4 + 64);
    for (uint64_t i = 0; i < n_tasks; ++i) {
        const int ps = img.add_task(code[i] == 0 ? SPACE_PROTEOME : SPACE_PAYLOAD, start_pos[i], length[i], start_pos_res[i], n_res);
        if (ps != PACK_OK) { t->rc = pack_to_err(ps); t->row = int64_t(i); t->err = "pack failed"; *ticket = t.release(); return V2P_OK; }
    }
    img.end_haplotype(n_res);
    img.finish();
    const uint64_t nd = img.desc.size(), nc = img.chunks.size();
    if (nd > q->desc_cap() || nc > q->chunk_cap()) { t->rc = gir_solo(c, *t); if (t->rc) t->err = v2p_last_error(c); *ticket = t.release(); return V2P_OK; }

    q->ns_pack += ns_since(tp); tp = clk::now();
    // ---- join a batch ----
    GirBatch* b = nullptr;
    uint64_t in_off = 0, res_off = 0, desc_off = 0, chunk_off = 0;
    {
        std::unique_lock<std::mutex> lk(q->mu);
        for (;;) {
            if (q->open && (q->open->in_bytes + in_need > q->cap_bytes || q->open->res_bytes + res_need > q->cap_bytes ||
                            q->open->n_desc + nd > q->desc_cap() || q->open->n_chunks + nc > q->chunk_cap())) {
                q->open = nullptr;                       // full: its runner closes it; this call opens the next one
                q->cv.notify_all();
            }
            if (!q->open) {
                b = nullptr;
                for (int k = 0; k < q->n_batch; ++k) if (q->batch[k].state == GirBatch::FREE) { b = &q->batch[k]; break; }
                if (!b) {                                // every batch is in flight
                    if (!may_wait) return V2P_BUSY;      // (nothing of this call has been staged yet)
                    q->cv.wait(lk);
                    continue;
                }
                b->n_reqs = b->n_ready = b->n_left = 0;
                b->in_bytes = b->res_bytes = b->n_desc = b->n_chunks = 0; b->rc = V2P_OK; b->err.clear();
                b->opened = std::chrono::steady_clock::now();
                b->state = GirBatch::OPEN;
                q->open = b; ++q->n_batches;
                q->cv.notify_all();                      // (its runner starts the gathering window)
            }
            b = q->open;
            break;
        }
        in_off = b->in_bytes; res_off = b->res_bytes; desc_off = b->n_desc; chunk_off = b->n_chunks;
        b->in_bytes += in_need; b->res_bytes += res_need; b->n_desc += nd; b->n_chunks += nc;
        ++b->n_reqs; ++b->n_left; ++q->n_joined;
    }
    q->ns_join += ns_since(tp); tp = clk::now();
    // ---- this caller's share of the staging, on its own thread ----
    {
        uint32_t seen = narrow_chars(ref, b->h_in.p + in_off, n_ref);
        seen |= narrow_chars(alt, b->h_in.p + in_off + ref_room, n_alt);
        t->wide = seen > 0xFFu;                          // a char above 0xFF: this GIR takes the 4-byte path by itself (its chunks run on garbage nobody reads)
        uint64_t* hd = reinterpret_cast<uint64_t*>(b->h_desc.p) + desc_off;
        for (uint64_t i = 0; i < nd; ++i) {
            const uint64_t d = img.desc[i];
            const unsigned sp = unsigned(d >> 62);
            hd[i] = sp == SPACE_PROTEOME ? d + in_off : (sp == SPACE_PAYLOAD ? d + in_off + ref_room : d);     // (source offset: the low 40 bits)
        }
        Chunk* hc = reinterpret_cast<Chunk*>(b->h_chunks.p) + chunk_off;
        for (uint64_t i = 0; i < nc; ++i) hc[i] = Chunk{img.chunks[i].task_begin + desc_off, img.chunks[i].dst_n + res_off};
    }
    q->ns_stage += ns_since(tp);
    {
        std::lock_guard<std::mutex> lk(q->mu);
        ++b->n_ready;
    }
    q->cv.notify_all();                                  // (the batch's runner may be waiting for this caller's share)
    t->b = b; t->res_off = res_off;
    *ticket = t.release();
    return V2P_OK;
}

extern "C" int v2p_gir_submit(v2p_ctx* c,
                              const uint64_t* code, const uint64_t* start_pos, const uint64_t* length,
                              const uint64_t* start_pos_res, uint64_t n_tasks,
                              const uint32_t* ref, uint64_t n_ref, const uint32_t* alt, uint64_t n_alt,
                              uint32_t* res, uint64_t n_res, v2p_gir_ticket** ticket)
{
    return gir_submit_impl(c, code, start_pos, length, start_pos_res, n_tasks, ref, n_ref, alt, n_alt, res, n_res, ticket, false);
}

extern "C" int v2p_gir_collect(v2p_ctx* c, v2p_gir_ticket* tk, int64_t* err_row)
{
    if (err_row) *err_row = -1;
    if (!c || !tk) return V2P_ERR_INVALID_ARG;
    std::unique_ptr<v2p_gir_ticket> t(tk);
    auto fail = [&](int code_, const std::string& msg, int64_t row) {
        if (err_row) *err_row = row;
        std::lock_guard<std::mutex> lk(c->mu);
        return c->fail(code_, msg, row);
    };
    GirBatch* b = t->b;
    if (!b) return t->rc == V2P_OK ? V2P_OK : fail(t->rc, t->err, t->row);
    GirQueue* q = c->queue;
    using clk = std::chrono::steady_clock;
    auto ns_since = [](clk::time_point tt) { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tt).count()); };
    clk::time_point tp = clk::now();
    {
        std::unique_lock<std::mutex> lk(q->mu);
        while (b->state != GirBatch::DONE) q->cv.wait(lk);
    }
    q->ns_wait += ns_since(tp); tp = clk::now();
    const int rc = b->rc;
    const std::string err = rc != V2P_OK ? b->err : std::string();
    if (rc == V2P_OK && !t->wide) {
        // only cells some task covers go to the caller (the others keep the caller's content: haplotype_instruction.rs:78 filled them with '.')
        const uint8_t* st = b->h_out.p + t->res_off;
        for (uint64_t i = 0; i < t->n_tasks;) {
            const uint64_t lo = t->start_pos_res[i];
            uint64_t hi = lo + t->length[i];
            for (++i; i < t->n_tasks && t->start_pos_res[i] == hi; ++i) hi += t->length[i];
            widen_chars(st + lo, t->res + lo, hi - lo);
        }
    }
    q->ns_widen += ns_since(tp);
    {
        std::lock_guard<std::mutex> lk(q->mu);
        if (--b->n_left == 0) { b->state = GirBatch::FREE; q->cv.notify_all(); }
    }
    if (rc != V2P_OK) return fail(rc, err, -1);
    if (t->wide) { const int r2 = gir_solo(c, *t); if (r2 != V2P_OK && err_row) *err_row = t->row; return r2; }
    return V2P_OK;
}

// the blocking form: GIR::execute as the reference calls it
extern "C" int v2p_execute_gir_shared(v2p_ctx* c,
                                      const uint64_t* code, const uint64_t* start_pos, const uint64_t* length,
                                      const uint64_t* start_pos_res, uint64_t n_tasks,
                                      const uint32_t* ref, uint64_t n_ref, const uint32_t* alt, uint64_t n_alt,
                                      uint32_t* res, uint64_t n_res, int64_t* err_row)
{
    if (err_row) *err_row = -1;
    v2p_gir_ticket* t = nullptr;
    const int rc = gir_submit_impl(c, code, start_pos, length, start_pos_res, n_tasks, ref, n_ref, alt, n_alt, res, n_res, &t, true);
    if (rc != V2P_OK) return rc;
    return v2p_gir_collect(c, t, err_row);
}

extern "C" int v2p_coalesce_stats(v2p_ctx* c, uint64_t* n_batches, uint64_t* n_calls)
{
    if (!c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    GirQueue* q = c->queue;
    uint64_t nb = 0, nj = 0;
    if (q) { std::lock_guard<std::mutex> lq(q->mu); nb = q->n_batches; nj = q->n_joined; }
    if (n_batches) *n_batches = nb;
    if (n_calls) *n_calls = nj;
    return V2P_OK;
}

// ---- batch -------------------------------------------------------------------

// Tables of a caller-packed image: every chunk's descriptors inside desc[0, n_desc), its result offset inside the
// arena, haplotype ranges ascending (digest_kernel walks hap_out_begin[h + 1]).  `out_bytes` = hap_out_begin[n_haps]
// unless given.  The result length of a chunk is only known on the device (in-chunk scan), which re-checks it.
static int check_packed(v2p_ctx* c, uint64_t n_desc, const Chunk* chunks, uint64_t n_chunks,
                        const uint64_t* hap_out_begin, uint64_t n_haps, uint64_t out_bytes = ~0ull)
{
    if (hap_out_begin) {
        for (uint64_t h = 0; h < n_haps; ++h)
            if (hap_out_begin[h + 1] < hap_out_begin[h]) return c->fail(V2P_ERR_INVALID_ARG, "hap_out_begin is not ascending", int64_t(h));
        if (out_bytes == ~0ull) out_bytes = hap_out_begin[n_haps];
    }
    for (uint64_t i = 0; i < n_chunks; ++i) {
        const uint64_t n = chunk_n(chunks[i].dst_n), dst = chunks[i].dst_n & DST_MASK;
        if (chunks[i].task_begin > n_desc || n > n_desc - chunks[i].task_begin)
            return c->fail(V2P_ERR_INVALID_ARG, "chunk " + std::to_string(i) + " points outside the descriptor array", int64_t(i));
        if (n > CHUNK_TASKS_DEEP) return c->fail(V2P_ERR_INVALID_ARG, "chunk " + std::to_string(i) + " holds more than 1024 descriptors", int64_t(i));
        if (out_bytes != ~0ull && dst > out_bytes) return c->fail(V2P_ERR_INVALID_ARG, "chunk " + std::to_string(i) + " starts outside the result arena", int64_t(i));
    }
    return V2P_OK;
}

int v2p_batch_create(v2p_ctx* c, v2p_batch** out)
{
    if (!c || !out) return V2P_ERR_INVALID_ARG;
    v2p_batch* b = new (std::nothrow) v2p_batch();
    if (!b) return c->fail(V2P_ERR_HIP, "out of host memory");
    b->ctx = c;
    *out = b;
    return V2P_OK;
}

void v2p_batch_destroy(v2p_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    {
        std::lock_guard<std::mutex> lk(b->ctx->mu);
        stream_detach(b);
        sync_ctx_streams(b->ctx);
    }
    b->d_desc.release(); b->d_chunks.release(); b->d_payload.release(); b->d_out.release();
    b->d_hap.release(); b->d_digest.release(); b->d_status.release(); b->d_build.release();
    b->d_tiles.release(); b->d_cover.release(); b->d_pad.release(); b->d_order.release(); b->d_patch.release(); b->d_stage.release(); b->d_pieces.release(); b->d_chunks2.release(); b->h_sum.release();
    for (hipEvent_t e : b->ev_os) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ev_aux) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ev_par) if (e) (void)hipEventDestroy(e);
    delete b;
}

// shared pre-check of one haplotype's tasks; returns V2P_OK or the error with its row
static int precheck(v2p_ctx* c, const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                    const uint64_t* start_pos_res, uint64_t n_tasks, uint64_t n_ref, uint64_t n_alt, uint64_t n_res)
{
    uint64_t cursor = 0;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        if (code[i] > 1) return c->fail(V2P_ERR_BAD_CODE, std::string(err_name(V2P_ERR_BAD_CODE)) + " at row " + std::to_string(i), int64_t(i));
        const uint64_t n_src = code[i] == 0 ? n_ref : n_alt;
        if (start_pos_res[i] + length[i] > n_res || start_pos_res[i] + length[i] < length[i])
            return c->fail(V2P_ERR_RES_OOB, std::string(err_name(V2P_ERR_RES_OOB)) + " at row " + std::to_string(i), int64_t(i));
        if (start_pos[i] + length[i] > n_src || start_pos[i] + length[i] < length[i])
            return c->fail(V2P_ERR_SRC_OOB, std::string(err_name(V2P_ERR_SRC_OOB)) + " at row " + std::to_string(i), int64_t(i));
        if (start_pos_res[i] < cursor)
            return c->fail(V2P_ERR_NOT_CANONICAL, std::string(err_name(V2P_ERR_NOT_CANONICAL)) + " at row " + std::to_string(i), int64_t(i));
        cursor = start_pos_res[i] + length[i];
    }
    return V2P_OK;
}

int v2p_batch_add_gir(v2p_batch* b,
                      const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                      const uint64_t* start_pos_res, uint64_t n_tasks,
                      const uint32_t* ref, uint64_t n_ref,
                      const uint32_t* alt, uint64_t n_alt,
                      uint64_t n_res)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if ((n_tasks && (!code || !start_pos || !length || !start_pos_res)) || (n_ref && !ref) || (n_alt && !alt))
        return c->fail(V2P_ERR_INVALID_ARG, "null argument");
    int rc = precheck(c, code, start_pos, length, start_pos_res, n_tasks, n_ref, n_alt, n_res);
    if (rc) return rc;
    for (uint64_t i = 0; i < n_ref; ++i) if (ref[i] > 0xFFu) return c->fail(V2P_ERR_NON_BYTE_CHAR, err_name(V2P_ERR_NON_BYTE_CHAR), int64_t(i));
    for (uint64_t i = 0; i < n_alt; ++i) if (alt[i] > 0xFFu) return c->fail(V2P_ERR_NON_BYTE_CHAR, err_name(V2P_ERR_NON_BYTE_CHAR), int64_t(i));
    const uint64_t off_ref = b->img.payload_alloc(n_ref);
    for (uint64_t i = 0; i < n_ref; ++i) b->img.payload[off_ref + i] = uint8_t(ref[i]);
    const uint64_t off_alt = b->img.payload_alloc(n_alt);
    for (uint64_t i = 0; i < n_alt; ++i) b->img.payload[off_alt + i] = uint8_t(alt[i]);
    for (uint64_t i = 0; i < n_tasks; ++i)
        (void)b->img.add_task(SPACE_PAYLOAD, (code[i] == 0 ? off_ref : off_alt) + start_pos[i], length[i], start_pos_res[i], n_res);
    b->img.end_haplotype(n_res);
    return V2P_OK;
}

static int add_haplotype_impl(v2p_batch* b,
                              const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                              const uint64_t* start_pos_res, uint64_t n_tasks,
                              const uint64_t* seg_ref_begin, const uint64_t* seg_proteome_off, uint64_t n_seg,
                              const uint8_t* alt, uint64_t n_alt, uint64_t n_res,
                              const uint64_t* rec_res_end, const uint64_t* rec_header_off, const uint32_t* rec_header_len, uint64_t n_rec,
                              bool fasta)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if ((n_tasks && (!code || !start_pos || !length || !start_pos_res)) || (n_seg && (!seg_ref_begin || !seg_proteome_off)) || (n_alt && !alt)
        || (fasta && n_rec && (!rec_res_end || !rec_header_off || !rec_header_len)))
        return c->fail(V2P_ERR_INVALID_ARG, "null argument");
    const uint64_t n_ref = n_seg ? seg_ref_begin[n_seg] : 0;
    int rc = precheck(c, code, start_pos, length, start_pos_res, n_tasks, n_ref, n_alt, n_res);
    if (rc) return rc;
    RefSegments segs{seg_ref_begin, seg_proteome_off, n_seg};
    // map every reference task before touching the image so a failure leaves the batch unchanged
    std::vector<uint64_t> mapped(n_tasks);
    for (uint64_t i = 0; i < n_tasks; ++i) {
        if (code[i] != 0) continue;
        if (length[i] == 0) { mapped[i] = 0; continue; }
        if (!segs.map(start_pos[i], length[i], &mapped[i]))
            return c->fail(V2P_ERR_SRC_OOB, "reference task crosses a transcript boundary at row " + std::to_string(i), int64_t(i));
        if (mapped[i] + length[i] > c->proteome_len)
            return c->fail(V2P_ERR_SRC_OOB, "reference task beyond the resident proteome at row " + std::to_string(i), int64_t(i));
    }
    std::vector<uint64_t> hdr_src;
    bool lf_before = fasta;
    if (fasta) {
        uint64_t prev = 0, ti = 0;
        hdr_src.resize(n_rec);
        for (uint64_t r = 0; r < n_rec; ++r) {                  // records must tile the result tape, headers must be resident
            if (rec_res_end[r] < prev || rec_res_end[r] > n_res) return c->fail(V2P_ERR_INVALID_ARG, "records are not ascending inside the result tape", int64_t(r));
            if (rec_header_len[r] == 0 || rec_header_off[r] + rec_header_len[r] > c->headers_len)
                return c->fail(V2P_ERR_SRC_OOB, "record header outside the resident header table", int64_t(r));
            while (ti < n_tasks && start_pos_res[ti] < rec_res_end[r]) {
                if (start_pos_res[ti] + length[ti] > rec_res_end[r]) return c->fail(V2P_ERR_INVALID_ARG, "task straddles a record boundary", int64_t(ti));
                ++ti;
            }
            prev = rec_res_end[r];
            hdr_src[r] = c->proteome_len + rec_header_off[r];
            if (c->headers_host[rec_header_off[r] + rec_header_len[r] - 1] != '\n')
                return c->fail(V2P_ERR_INVALID_ARG, "a record header must end in a line feed", int64_t(r));
            if (r > 0 && (rec_header_off[r] == 0 || c->headers_host[rec_header_off[r] - 1] != '\n')) lf_before = false;
        }
        if ((n_rec ? rec_res_end[n_rec - 1] : 0) != n_res) return c->fail(V2P_ERR_INVALID_ARG, "records do not cover the result tape");
    }
    const uint64_t off_alt = b->img.payload_alloc(n_alt);
    if (n_alt) memcpy(&b->img.payload[off_alt], alt, n_alt);
    auto emit_task = [&](uint64_t i) {
        return code[i] == 0 ? b->img.add_task(SPACE_PROTEOME, mapped[i], length[i], start_pos_res[i], n_res)
                            : b->img.add_task(SPACE_PAYLOAD, off_alt + start_pos[i], length[i], start_pos_res[i], n_res);
    };
    if (fasta) {
        (void)interleave_fasta(b->img, start_pos_res, length, n_tasks, rec_res_end, hdr_src.data(), rec_header_len, n_rec, SPACE_PROTEOME, lf_before, emit_task);
    } else {
        for (uint64_t i = 0; i < n_tasks; ++i) (void)emit_task(i);
    }
    b->img.end_haplotype(n_res);
    b->uses_proteome = true;
    return V2P_OK;
}

int v2p_batch_add_haplotype(v2p_batch* b,
                            const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                            const uint64_t* start_pos_res, uint64_t n_tasks,
                            const uint64_t* seg_ref_begin, const uint64_t* seg_proteome_off, uint64_t n_seg,
                            const uint8_t* alt, uint64_t n_alt,
                            uint64_t n_res)
{
    return add_haplotype_impl(b, code, start_pos, length, start_pos_res, n_tasks, seg_ref_begin, seg_proteome_off, n_seg,
                              alt, n_alt, n_res, nullptr, nullptr, nullptr, 0, false);
}

int v2p_batch_add_haplotype_fasta(v2p_batch* b,
                                  const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                                  const uint64_t* start_pos_res, uint64_t n_tasks,
                                  const uint64_t* seg_ref_begin, const uint64_t* seg_proteome_off, uint64_t n_seg,
                                  const uint8_t* alt, uint64_t n_alt, uint64_t n_res,
                                  const uint64_t* rec_res_end, const uint64_t* rec_header_off, const uint32_t* rec_header_len,
                                  uint64_t n_rec)
{
    return add_haplotype_impl(b, code, start_pos, length, start_pos_res, n_tasks, seg_ref_begin, seg_proteome_off, n_seg,
                              alt, n_alt, n_res, rec_res_end, rec_header_off, rec_header_len, n_rec, true);
}

// ---- step 5 folded into the image builder ------------------------------------------------
// The reference concatenates per-transcript GIRs into a haplotype GIR on the host
// (haplotype_instruction.rs:94-133): it copies every transcript's reference into a private tape
// and rebases start_pos by ref_counter / alt_counter and start_pos_res by res_counter.  Here a
// transcript GIR goes straight into the image: reference tasks are rebased onto the resident
// proteome (no tape copy, no ref_counter), alt tasks onto the payload arena, and result
// offsets are implicit (device prefix sum), so the three running sums disappear.

int v2p_batch_begin_haplotype(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if (b->hap_open) return c->fail(V2P_ERR_STATE, "previous haplotype not ended");
    b->hap_open = true; b->hap_res = 0; b->hap_records = 0;
    return V2P_OK;
}

int v2p_batch_add_transcript(v2p_batch* b,
                             const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                             const uint64_t* start_pos_res, uint64_t n_tasks,
                             uint64_t tx_proteome_off, uint64_t tx_ref_len,
                             const uint8_t* alt, uint64_t n_alt, uint64_t res_len,
                             uint64_t header_off, uint32_t header_len)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized || !b->hap_open) return c->fail(V2P_ERR_STATE, "no open haplotype");
    if ((n_tasks && (!code || !start_pos || !length || !start_pos_res)) || (n_alt && !alt)) return c->fail(V2P_ERR_INVALID_ARG, "null argument");
    if (tx_proteome_off + tx_ref_len > c->proteome_len) return c->fail(V2P_ERR_SRC_OOB, "transcript outside the resident proteome");
    int rc = precheck(c, code, start_pos, length, start_pos_res, n_tasks, tx_ref_len, n_alt, res_len);   // update_task :140-158 + task.rs bounds
    if (rc) return rc;
    const bool fasta = header_len != 0;
    if (fasta) {
        if (header_off + header_len > c->headers_len) return c->fail(V2P_ERR_SRC_OOB, "record header outside the resident header table");
        if (c->headers_host[header_off + header_len - 1] != '\n') return c->fail(V2P_ERR_INVALID_ARG, "a record header must end in a line feed");
        const uint64_t src = c->proteome_len + header_off;
        const bool merge = b->hap_records > 0 && header_off > 0 && c->headers_host[header_off - 1] == '\n';
        if (b->hap_records > 0 && !merge) b->img.add_literal(SPACE_PROTEOME, b->last_hdr_src + b->last_hdr_len - 1, 1);
        if (merge) b->img.add_literal(SPACE_PROTEOME, src - 1, header_len + 1);     // previous record's line feed + this header
        else b->img.add_literal(SPACE_PROTEOME, src, header_len);
        b->last_hdr_src = src; b->last_hdr_len = header_len;
    }
    const uint64_t off_alt = b->img.payload_alloc(n_alt);
    if (n_alt) memcpy(&b->img.payload[off_alt], alt, n_alt);
    const uint64_t base = b->hap_res, n_res = b->hap_res + res_len;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        if (code[i] == 0) (void)b->img.add_task(SPACE_PROTEOME, tx_proteome_off + start_pos[i], length[i], base + start_pos_res[i], n_res);
        else              (void)b->img.add_task(SPACE_PAYLOAD, off_alt + start_pos[i], length[i], base + start_pos_res[i], n_res);
    }
    b->img.fill_to(n_res);                 // cells of this transcript no task covers keep '.'
    b->hap_res = n_res;
    ++b->hap_records;
    b->uses_proteome = true;
    return V2P_OK;
}

int v2p_batch_end_haplotype(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized || !b->hap_open) return c->fail(V2P_ERR_STATE, "no open haplotype");
    if (b->hap_records > 0 && b->last_hdr_len != 0)
        b->img.add_literal(SPACE_PROTEOME, b->last_hdr_src + b->last_hdr_len - 1, 1);     // line feed of the last record
    b->img.end_haplotype(b->hap_res);
    b->hap_open = false; b->last_hdr_len = 0;
    return V2P_OK;
}

// ---- ROWS images: the image built in ONE pass (build_rows.hip; format: rows_image.hpp) -----------------------------------------
// A transcript stream as it sits on the device: either uploaded for one build (v2p_batch_build_on_device) or resident (v2p_stream).
struct DevStreamView {
    uint64_t n_haps = 0, n_tx = 0, n_tasks = 0, n_alt = 0;
    const uint64_t* hap_tx_begin = nullptr; const uint64_t* tx_proteome_off = nullptr;
    const uint32_t* tx_ref_len = nullptr; const uint32_t* tx_res_len = nullptr;
    const uint64_t* tx_task_begin = nullptr; const uint64_t* tx_alt_begin = nullptr;
    const uint8_t* code = nullptr; const uint32_t* start_pos = nullptr; const uint32_t* length = nullptr; const uint32_t* start_pos_res = nullptr;
    const uint8_t* alt = nullptr;
    const uint64_t* tx_header_off = nullptr; const uint32_t* tx_header_len = nullptr;      // FASTA emit (nullptr: plain tapes)
    bool fasta = false;
    double items_mean = 1.0, items_var = 0.0;          // items (Tasks; a transcript without Tasks is one) per transcript: rows_pick_k
    double desc_mean = 1.0, desc_var = 0.0;            // ... and an upper estimate of the descriptors per transcript
    double len_mean = 0.0, len_var = 0.0;              // ... and its arena bytes (rows_pick_k_tiles)
};

#ifdef V2P_BENCH_VARIANTS
static int build_patch_image(v2p_batch* b, const DevStreamView& v, float* build_ms, bool own_stream_copy, uint64_t known_out_bytes);
#endif

// mean and spread of the items per transcript from a sample of the (host) stream -- and of an upper estimate of its DESCRIPTORS: a
// one-residue alt Task between two reference copies fuses with both (one descriptor for three Tasks; transcript_instructions.rs:654-663
// is where a missense becomes that triple), everything else is a descriptor of its own, plus a '.' tail
static void stream_item_stats(const v2p_txstream* s, DevStreamView& v)
{
    v.items_mean = 1.0; v.items_var = 0.0; v.desc_mean = 1.0; v.desc_var = 0.0; v.len_mean = 0.0; v.len_var = 0.0;
    const uint64_t n_tx = s->n_tx;
    if (n_tx == 0) return;
    const uint64_t step = n_tx > 65536 ? n_tx / 65536 : 1;
    const bool fasta = s->tx_header_off && s->tx_header_len;
    double sum = 0, sq = 0, dsum = 0, dsq = 0, lsum = 0, lsq = 0, cnt = 0;
    for (uint64_t u = 0; u < n_tx; u += step) {
        const uint64_t t0 = s->tx_task_begin[u], t1 = s->tx_task_begin[u + 1], nt = t1 - t0;
        const double x = nt ? double(nt) : 1.0;                      // (a transcript without tasks is one item)
        uint64_t nf = 0;
        for (uint64_t i = t0 + 1; i + 1 < t1; ++i) nf += (s->code[i] == 1 && s->length[i] == 1 && s->code[i - 1] == 0 && s->code[i + 1] == 0) ? 1u : 0u;
        const double d = (nt > 2 * nf ? double(nt - 2 * nf) : 1.0) + 1.0;
        const uint32_t hl = fasta ? s->tx_header_len[u] : 0u;
        const double l = double(s->tx_res_len[u]) + (hl ? double(hl) + 1.0 : 0.0);       // arena bytes of the transcript
        sum += x; sq += x * x; dsum += d; dsq += d * d; lsum += l; lsq += l * l; cnt += 1;
    }
    const double m = sum / cnt, dm = dsum / cnt, lm = lsum / cnt;
    v.items_mean = m; v.items_var = sq / cnt - m * m > 0 ? sq / cnt - m * m : 0.0;
    v.desc_mean = dm; v.desc_var = dsq / cnt - dm * dm > 0 ? dsq / cnt - dm * dm : 0.0;
    v.len_mean = lm; v.len_var = lsq / cnt - lm * lm > 0 ? lsq / cnt - lm * lm : 0.0;
}
static void copy_stats(const DevStreamView& from, DevStreamView& to)
{
    to.items_mean = from.items_mean; to.items_var = from.items_var; to.desc_mean = from.desc_mean; to.desc_var = from.desc_var; to.len_mean = from.len_mean; to.len_var = from.len_var;
}

// TILE images (dense_pieces.h): the tile is the executor's work item -- K consecutive transcripts whose result fits its LDS image
// (TILE_SPAN_MAX bytes) and whose pieces fit the tile's slots (<= TILE_SLOTS_MAX), with SIX standard deviations to spare on either (a
// tile that does not fit sends the whole build to the dense rows image: it must not happen by chance).  The largest such K; *slots: a
// power of two from 256 up that holds the tile's pieces with the same margin.  0: no K does (transcripts of a dozen KiB) -- no tile image.
static uint32_t rows_pick_k_tiles(const DevStreamView& v, uint32_t* slots)
{
    *slots = 256;
    if (v.n_tx == 0) return 1;
    const double z = 6.0;
    const double lm = v.len_mean > 1.0 ? v.len_mean : 1.0, lsd = sqrt(v.len_var);
    const double pm = v.desc_mean + (v.fasta ? 2.0 : 0.0) + lm / 16.0, psd = sqrt(v.desc_var) + lsd / 16.0;     // pieces <= descriptors + result bytes / 16
    auto k_for = [&](double m, double sd, double cap) { const double x = (-z * sd + sqrt(z * z * sd * sd + 4.0 * m * cap)) / (2.0 * m); return floor(x * x); };
    double k = k_for(lm, lsd, double(TILE_SPAN_MAX) - 64.0);
    const double kp = k_for(pm, psd, double(TILE_SLOTS_MAX) - 16.0);
    if (kp < k) k = kp;
    if (k > 64.0) k = 64.0;
    if (k < 1.0) return 0;
    const double need = k * pm + z * psd * sqrt(k) + 16.0;
    uint32_t sl = 256;
    while (sl < TILE_SLOTS_MAX && double(sl) < need) sl <<= 1;
    *slots = sl;
    return uint32_t(k);
}

// Tiles of K consecutive transcripts, one wave each (K <= 64: the kernel's prologue gives every transcript of the tile a lane).  A
// wave takes 64 items in its first window and ADV more in every further one, so K is picked to FILL the tile's last window: among
// the window counts w whose capacity stays below ~360 items (a tile's descriptors must fit its 256 slots of the padded array; one
// that does not sends the build to its two-pass form), the K <= 64 with the least work per transcript (windows + a tile's fixed
// cost) -- mean and spread of the items per transcript from a sample of the stream.
static uint32_t rows_pick_k(const DevStreamView& v, int mode)
{
    if (mode == ROWS_TILES) { uint32_t slots; const uint32_t k = rows_pick_k_tiles(v, &slots); return k ? k : 1u; }
    if (v.n_tx == 0) return 1;
    const double m = v.items_mean, var = v.items_var, sd = sqrt(var);
    const double adv = mode == ROWS_DENSE ? 58.0 : 60.0, z = 1.3;
    const double k_cap = v.fasta ? 240.0 / (m + 2.0) : 64.0;         // FASTA: a header and a line feed per transcript on top
    // ... and the tile's DESCRIPTORS must fit its 256 slots: where few Tasks fuse (runs of insertions, deletions, long payloads) the items
    // are the descriptors, and a tile sized by its windows alone overflows -- the build then ran a second time in its two-pass form
    double k_desc = 64.0;
    {
        const double dm = v.desc_mean + (v.fasta ? 2.0 : 0.0), dsd = sqrt(v.desc_var);
        const double x = (-z * dsd + sqrt(z * z * v.desc_var + 4.0 * dm * 244.0)) / (2.0 * dm);
        k_desc = floor(x * x);
        if (k_desc < 1.0) k_desc = 1.0;
    }
    uint32_t best_k = 1;
    double best_cost = 1e30;
    for (uint32_t w = 1; w <= 8; ++w) {
        const double cap = 64.0 + (w - 1) * adv;
        if (w > 1 && cap > 360.0) break;
        // K m + z sd sqrt(K) <= cap
        const double x = (-z * sd + sqrt(z * z * var + 4.0 * m * cap)) / (2.0 * m);
        double k = floor(x * x);
        if (k > 64.0) k = 64.0;
        if (k > k_cap) k = floor(k_cap);
        if (k > k_desc) k = k_desc;
        if (k < 1.0) k = 1.0;
        const double windows = 1.0 + (k * m + z * sd * sqrt(k) > 64.0 ? ceil((k * m + z * sd * sqrt(k) - 64.0) / adv) : 0.0);
        const double cost = (windows + 4.0) / k;                        // (a tile's fixed cost -- its dependent loads before the first window -- is worth about four windows)
        if (cost < best_cost - 1e-12 || (cost < best_cost + 1e-12 && k > best_k)) { best_cost = cost; best_k = uint32_t(k); }
    }
    return best_k;
}

// Where the stream's arrays sit in ONE allocation (device, and -- the streamed pipeline -- its pinned mirror on the host): offsets of
// 16-byte aligned arrays, the Task arrays with the slack the parse reads past a transcript's last task.  The alt bytes are an allocation
// of their own (the image's payload descriptors address it).
struct StreamLayout {
    uint64_t o_hap, o_poff, o_rlen, o_res, o_tb, o_ab, o_code, o_sp, o_ln, o_sr, o_hoff, o_hlen, total;
};
static StreamLayout stream_layout(uint64_t n_h, uint64_t n_tx, uint64_t n_tk, bool fasta)
{
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    StreamLayout L;
    L.o_hap = carve((n_h + 1) * 8); L.o_poff = carve(n_tx * 8); L.o_rlen = carve(n_tx * 4); L.o_res = carve(n_tx * 4);
    L.o_tb = carve((n_tx + 2) * 8); L.o_ab = carve((n_tx + 2) * 8); L.o_code = carve(n_tk + 64); L.o_sp = carve((n_tk + 16) * 4); L.o_ln = carve((n_tk + 16) * 4);
    L.o_sr = carve((n_tk + 16) * 4); L.o_hoff = carve(fasta ? n_tx * 8 : 0); L.o_hlen = carve(fasta ? n_tx * 4 : 0);
    L.total = off;
    return L;
}
// the (dst offset, source, bytes) of every array of a host stream in that layout
struct StreamPiece { uint64_t off; const void* src; uint64_t bytes; const char* what; };
static uint32_t stream_pieces(const v2p_txstream* s, const StreamLayout& L, bool fasta, StreamPiece* out)
{
    const uint64_t n_tx = s->n_tx, n_tk = s->n_tasks, n_h = s->n_haps;
    uint32_t n = 0;
    auto add = [&](uint64_t off, const void* src, uint64_t bytes, const char* what) { if (bytes) out[n++] = StreamPiece{off, src, bytes, what}; };
    add(L.o_hap, s->hap_tx_begin, (n_h + 1) * 8, "H2D(hap_tx_begin)"); add(L.o_poff, s->tx_proteome_off, n_tx * 8, "H2D(tx_proteome_off)");
    add(L.o_rlen, s->tx_ref_len, n_tx * 4, "H2D(tx_ref_len)"); add(L.o_res, s->tx_res_len, n_tx * 4, "H2D(tx_res_len)");
    add(L.o_tb, s->tx_task_begin, s->tx_task_begin ? (n_tx + 1) * 8 : 0, "H2D(tx_task_begin)"); add(L.o_ab, s->tx_alt_begin, s->tx_alt_begin ? (n_tx + 1) * 8 : 0, "H2D(tx_alt_begin)");
    add(L.o_code, s->code, n_tk, "H2D(code)"); add(L.o_sp, s->start_pos, n_tk * 4, "H2D(start_pos)"); add(L.o_ln, s->length, n_tk * 4, "H2D(length)");
    add(L.o_sr, s->start_pos_res, n_tk * 4, "H2D(start_pos_res)");
    if (fasta) { add(L.o_hoff, s->tx_header_off, n_tx * 8, "H2D(tx_header_off)"); add(L.o_hlen, s->tx_header_len, n_tx * 4, "H2D(tx_header_len)"); }
    return n;
}
static void stream_view(const v2p_txstream* s, const StreamLayout& L, bool fasta, uint8_t* d, const uint8_t* d_alt, DevStreamView& v)
{
    v = DevStreamView();
    v.n_haps = s->n_haps; v.n_tx = s->n_tx; v.n_tasks = s->n_tasks; v.n_alt = s->n_alt;
    v.hap_tx_begin = reinterpret_cast<const uint64_t*>(d + L.o_hap); v.tx_proteome_off = reinterpret_cast<const uint64_t*>(d + L.o_poff);
    v.tx_ref_len = reinterpret_cast<const uint32_t*>(d + L.o_rlen); v.tx_res_len = reinterpret_cast<const uint32_t*>(d + L.o_res);
    v.tx_task_begin = reinterpret_cast<const uint64_t*>(d + L.o_tb); v.tx_alt_begin = reinterpret_cast<const uint64_t*>(d + L.o_ab);
    v.code = d + L.o_code; v.start_pos = reinterpret_cast<const uint32_t*>(d + L.o_sp); v.length = reinterpret_cast<const uint32_t*>(d + L.o_ln);
    v.start_pos_res = reinterpret_cast<const uint32_t*>(d + L.o_sr); v.alt = d_alt;
    v.tx_header_off = fasta ? reinterpret_cast<const uint64_t*>(d + L.o_hoff) : nullptr;
    v.tx_header_len = fasta ? reinterpret_cast<const uint32_t*>(d + L.o_hlen) : nullptr;
    v.fasta = fasta;
}

// The stream's arrays into one device allocation (`buf`, carved) + its alt bytes (`altbuf`), on `stream`; v receives the device pointers.
static int upload_stream(v2p_ctx* c, const v2p_txstream* s, bool fasta, DevBuf& buf, DevBuf& altbuf, DevStreamView& v, hipStream_t stream, bool with_stats = true)
{
    const StreamLayout L = stream_layout(s->n_haps, s->n_tx, s->n_tasks, fasta);
    HIP_TRY(c, buf.ensure(L.total), "hipMalloc(stream)");
    uint8_t* const d = buf.ptr();
    HIP_TRY(c, altbuf.ensure(s->n_alt), "hipMalloc(alt)");
    StreamPiece pc[12];
    const uint32_t np = stream_pieces(s, L, fasta, pc);
    // (pageable memory as it comes: the runtime pins the caller's pages piece by piece and the copy runs at the link's rate -- 57 GB/s on
    // C3 whole's 9.6 GB.  A ring of pinned slots filled by a thread team was measured: 45 GB/s on a first upload, 28 on the next ones.)
    for (uint32_t k = 0; k < np; ++k) HIP_TRY(c, hipMemcpyAsync(d + pc[k].off, pc[k].src, pc[k].bytes, hipMemcpyHostToDevice, stream), pc[k].what);
    if (s->n_alt) HIP_TRY(c, hipMemcpyAsync(altbuf.ptr(), s->alt, s->n_alt, hipMemcpyHostToDevice, stream), "H2D(alt)");
    stream_view(s, L, fasta, d, altbuf.ptr(), v);
    if (with_stats) stream_item_stats(s, v);            // (walks the Task arrays through the tables: only behind their check)
    return V2P_OK;
}

static void rows_args_of(const DevStreamView& v, v2p_ctx* c, uint32_t K, uint64_t n_tiles, RowsArgs& a)
{
    a = RowsArgs{};
    a.n_tx = v.n_tx; a.n_tasks = v.n_tasks; a.n_alt = v.n_alt; a.n_haps = v.n_haps;
    a.hap_tx_begin = v.hap_tx_begin; a.tx_proteome_off = v.tx_proteome_off; a.tx_ref_len = v.tx_ref_len; a.tx_res_len = v.tx_res_len;
    a.tx_task_begin = v.tx_task_begin; a.tx_alt_begin = v.tx_alt_begin;
    a.code = v.code; a.start_pos = v.start_pos; a.length = v.length; a.start_pos_res = v.start_pos_res; a.alt = v.alt;
    a.tx_header_off = v.tx_header_off; a.tx_header_len = v.tx_header_len;
    a.proteome_len = c->proteome_len; a.headers_len = c->headers_len; a.K = K; a.n_tiles = n_tiles;
}

// Called by v2p_batch_build_on_device / v2p_batch_build_from_stream (kernel 6: wave image, 7: dense) with the stream's tables already
// checked and c->mu held.  own_stream_copy: the stream sits in b->d_build (uploaded for this build) and is released with it.
static int build_rows_image(v2p_batch* b, const DevStreamView& v, int mode, float* build_ms, bool own_stream_copy)
{
    v2p_ctx* c = b->ctx;
    const bool fasta = v.fasta;
    const uint64_t n_tx = v.n_tx, n_h = v.n_haps;
    const uint32_t K = rows_pick_k(v, mode);
    const uint64_t n_tiles = n_tx ? (n_tx + K - 1) / K : 1;
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    int rc = init_status(c, b->d_status);
    if (rc) return rc;
    struct Cleanup {
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        DevBuf tiles, scratch, cover, pad;
        v2p_batch* b;
        bool ok = false, own;
        Cleanup(v2p_batch* b_, bool own_) : b(b_), own(own_) {}
        ~Cleanup() {
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            tiles.release(); scratch.release(); cover.release(); pad.release();
            if (own) b->d_build.release();
            if (!ok) { b->img.hap_out_begin.assign(1, 0); b->n_desc = b->n_chunks = b->n_payload = b->out_bytes = b->n_haps = 0; b->payload_dev = nullptr; }
        }
    } guard(b, own_stream_copy);
    for (hipEvent_t& e : guard.ev) HIP_TRY(c, hipEventCreate(&e), "hipEventCreate");
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    const uint64_t o_tbytes = carve(n_tiles * 8), o_tbase = carve((n_tiles + 1) * 8), o_tcount = carve((n_tiles + 1) * 4), o_tdbase = carve((n_tiles + 2) * 8),
                   o_totals = carve(64), o_scan = carve((rows_scan_scratch_entries(n_tiles) + scan_tiles_for(n_tiles + 1)) * 8);
    HIP_TRY(c, guard.tiles.ensure_exact(off), "hipMalloc(tile tables)");
    uint8_t* const d = guard.tiles.ptr();
    RowsArgs a;
    rows_args_of(v, c, K, n_tiles, a);
    a.tile_bytes = reinterpret_cast<uint64_t*>(d + o_tbytes); a.tile_res_base = reinterpret_cast<uint64_t*>(d + o_tbase);
    a.tile_count = reinterpret_cast<uint32_t*>(d + o_tcount); a.tile_desc_base = reinterpret_cast<uint64_t*>(d + o_tdbase);
    a.totals = reinterpret_cast<uint64_t*>(d + o_totals);
    a.status = reinterpret_cast<unsigned long long*>(b->d_status.ptr());
    uint64_t* const scan_scratch = reinterpret_cast<uint64_t*>(d + o_scan);
    // 1. res_counter per tile (haplotype_instruction.rs:90,132 as a scan); the host needs the arena's size for the row map
    HIP_TRY(c, hipEventRecord(guard.ev[0], c->stream), "hipEventRecord");
    HIP_TRY(c, launch_rows_tile_bytes(a, scan_scratch, c->stream), "launch(tile bytes)");
    HIP_TRY(c, hipEventRecord(guard.ev[1], c->stream), "hipEventRecord");
    uint64_t out_bytes = 0;
    HIP_TRY(c, hipMemcpyAsync(&out_bytes, d + o_tbase + n_tiles * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(out_bytes)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    const uint64_t n_rows = (out_bytes + ROW_BYTES - 1) / ROW_BYTES, n_segs = (n_rows + ROWS_SEG - 1) / ROWS_SEG;
    if (n_rows > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 rows in one batch");
    a.out_bytes = out_bytes; a.n_rows = n_rows; a.n_segs = n_segs;
    const uint32_t chunk_pad = rows_chunk_pad_for(out_bytes, v.n_tasks + (fasta ? 2 * n_tx : 0), mode);
    a.chunk_pad = chunk_pad;
    // scratch of the parse and the count pass of the cutter: row map, chunks per segment + their scan
    const uint64_t c_cover = 0, c_segc = up8((n_rows + 1) * 8), c_segb = c_segc + up8((n_segs + 1) * 4), c_tiles = c_segb + up8((n_segs + 2) * 8),
                   c_cpad = c_tiles + up8(scan_tiles_for(n_segs + 1) * 8), c_end = c_cpad + up8(n_segs * chunk_pad * sizeof(Chunk));
    HIP_TRY(c, guard.cover.ensure_exact(c_end), "hipMalloc(row map)");
    a.cover = reinterpret_cast<uint64_t*>(guard.cover.ptr() + c_cover);
    a.seg_count = reinterpret_cast<uint32_t*>(guard.cover.ptr() + c_segc);
    a.seg_base = reinterpret_cast<const uint64_t*>(guard.cover.ptr() + c_segb);
    a.chunks_pad = reinterpret_cast<Chunk*>(guard.cover.ptr() + c_cpad);
    HIP_TRY(c, b->d_hap.ensure((n_h + 1) * 8), "hipMalloc(hap_begin)");
    a.hap_out_begin = reinterpret_cast<uint64_t*>(b->d_hap.ptr());
    HIP_TRY(c, hipMemsetAsync(d + o_totals, 0, 64, c->stream), "hipMemset(totals)");
    // 2. the parse.  One pass: every tile's descriptors into its slots of a padded array, the tiles' counts scanned, a copy kernel
    // compacts.  A tile that does not fit its slots (a transcript with hundreds of tasks ...): the two-pass form -- count, scan, write.
    bool two_pass = n_tiles >= (1ull << 25);
    uint64_t n_desc = 0;
    float ms_parse = 0.f;
    for (int attempt = 0; ; ++attempt) {
        HIP_TRY(c, hipEventRecord(guard.ev[2], c->stream), "hipEventRecord");
        if (!two_pass) {
            // (2 KiB per tile -- C3 whole: 3.2 GB next to a 36 GB arena; when the device cannot spare it the two-pass form, which
            // needs no padded array, builds the same image)
            const hipError_t pe = guard.pad.ensure_exact(n_tiles * ROWS_PAD_SLOTS * 8);
            if (pe == hipErrorOutOfMemory) { (void)hipGetLastError(); two_pass = true; }
            else if (pe != hipSuccess) return c->hip_fail(pe, "hipMalloc(padded descriptors)");
            else a.desc_pad = reinterpret_cast<uint64_t*>(guard.pad.ptr());
        }
        HIP_TRY(c, launch_rows_parse(a, mode, fasta, two_pass ? 1 : 0, c->stream), "launch(parse)");
        HIP_TRY(c, launch_scan_u32(a.tile_count, n_tiles, a.tile_desc_base, scan_scratch + rows_scan_scratch_entries(n_tiles), c->stream), "launch(scan)");
        HIP_TRY(c, hipEventRecord(guard.ev[3], c->stream), "hipEventRecord");
        HIP_TRY(c, hipMemcpyAsync(&n_desc, d + o_tdbase + n_tiles * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(n_desc)");
        unsigned long long st = STATUS_CLEAN;
        HIP_TRY(c, hipMemcpyAsync(&st, b->d_status.ptr(), 8, hipMemcpyDeviceToHost, c->stream), "D2H(status)");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, guard.ev[2], guard.ev[3]);
        ms_parse += ms;
        const uint32_t reason = st == STATUS_CLEAN ? 0u : uint32_t(st & 0xFFu);
        if (reason == STATUS_ROWS_STAGE && !two_pass && attempt == 0) {
            two_pass = true;
            guard.pad.release();
            rc = init_status(c, b->d_status);
            if (rc) return rc;
            continue;
        }
        if (reason == STATUS_ROWS_SPAN) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return c->fail(V2P_ERR_UNSUPPORTED, "64 consecutive transcripts with more than 2 GiB of result", int64_t(st >> 8)); }
        rc = collect_status(c, b->d_status);              // what the reference would panic on
        if (rc) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return rc; }
        break;
    }
    HIP_TRY(c, b->d_desc.ensure((n_desc ? n_desc : 1) * 8), "hipMalloc(desc)");
    a.desc = reinterpret_cast<uint64_t*>(b->d_desc.ptr()); a.desc_cap = n_desc;
    HIP_TRY(c, hipEventRecord(guard.ev[4], c->stream), "hipEventRecord");
    if (two_pass) HIP_TRY(c, launch_rows_parse(a, mode, fasta, 2, c->stream), "launch(parse: write)");
    else HIP_TRY(c, launch_rows_compact(a, c->stream), "launch(compact)");
    HIP_TRY(c, launch_rows_hap_begin(a, c->stream), "launch(hap_begin)");
    HIP_TRY(c, launch_rows_cut(a, mode, 2, c->stream), "launch(cut)");
    HIP_TRY(c, launch_scan_u32(a.seg_count, n_segs, const_cast<uint64_t*>(a.seg_base), reinterpret_cast<uint64_t*>(guard.cover.ptr() + c_tiles), c->stream), "launch(scan)");
    HIP_TRY(c, hipEventRecord(guard.ev[5], c->stream), "hipEventRecord");
    uint64_t totals[4] = {0, 0, 0, 0}, n_chunks = 0;
    HIP_TRY(c, hipMemcpyAsync(totals, d + o_totals, 32, hipMemcpyDeviceToHost, c->stream), "D2H(totals)");
    HIP_TRY(c, hipMemcpyAsync(&n_chunks, guard.cover.ptr() + c_segb + n_segs * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(n_chunks)");
    {
        unsigned long long st = STATUS_CLEAN;
        HIP_TRY(c, hipMemcpyAsync(&st, b->d_status.ptr(), 8, hipMemcpyDeviceToHost, c->stream), "D2H(status)");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        if (st != STATUS_CLEAN && uint32_t(st & 0xFFu) == STATUS_ROWS_TOO_MANY) {
            (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream);
            return c->fail(V2P_ERR_UNSUPPORTED, mode == ROWS_DENSE ? "a 1 KiB row of the result holds more than 1024 descriptors"
                                                                   : "a 1 KiB row of the result holds more than 64 descriptors: not a wave image (kernel 7 builds a dense one)", int64_t(st >> 8));
        }
        rc = collect_status(c, b->d_status);
        if (rc) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return rc; }
    }
    guard.pad.release();
    const uint64_t last_dst = totals[2];
    if (n_chunks > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch");
    // 3. the chunk table: emit, keys, XCD / window order inside blocks of the arena (as the host packer's)
    const uint64_t cap = n_chunks ? n_chunks : 1;
    const uint64_t n_blocks_cap = order_blocks_thread_blocks(cap, XCD_ORDER_MAX_BLOCKS);
    const uint64_t n_sub_cap = uint64_t(XCD_SUB) * n_blocks_cap;
    const uint64_t s_tmp = 0, s_bucket = s_tmp + up8(cap * 16), s_hist = s_bucket + up8(cap),
                   s_sub = s_hist + up8((n_blocks_cap + 1) * 8 * 4), s_tmp2 = s_sub + up8(cap), s_bucket2 = s_tmp2 + up8(cap * 16),
                   s_subhist = s_bucket2 + up8(cap), s_substart = s_subhist + up8(n_sub_cap * 4), s_subtiles = s_substart + up8((n_sub_cap + 1) * 8),
                   s_tot = s_subtiles + up8(scan_tiles_for(n_sub_cap) * 8), s_end = s_tot + up8(uint64_t(XCD_ORDER_MAX_BLOCKS) * 8 * 4);
    DevBuf& scratch = guard.scratch;
    HIP_TRY(c, scratch.ensure_exact(s_end), "hipMalloc(build scratch)");
    HIP_TRY(c, b->d_chunks.ensure(cap * sizeof(Chunk)), "hipMalloc(chunks)");
    HIP_TRY(c, b->d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_digest.ensure((n_h ? n_h : 1) * 8), "hipMalloc(digest)");
    a.chunks_tmp = reinterpret_cast<Chunk*>(scratch.ptr() + s_tmp); a.chunk_cap = cap;
    a.bucket = scratch.ptr() + s_bucket;
    a.sub = scratch.ptr() + s_sub;
    HIP_TRY(c, hipEventRecord(guard.ev[6], c->stream), "hipEventRecord");
    if (totals[3]) HIP_TRY(c, launch_rows_cut(a, mode, 1, c->stream), "launch(cut: emit)");      // (a segment with more chunks than the padded table's slots)
    else HIP_TRY(c, launch_rows_chunk_compact(a, c->stream), "launch(chunk table)");
    const bool reorder = !(c->flags & V2P_FLAG_RESULT_ORDER) && n_chunks >= 16 && c->proteome_len != 0 && n_desc != 0;
    if (reorder) {
        a.hap_major = xcd_order_window_major(last_dst, n_desc) ? 0u : 1u;
        HIP_TRY(c, launch_rows_keys(a, n_chunks, n_desc, c->stream), "launch(keys)");
        const uint32_t nb = xcd_order_blocks(last_dst, c->proteome_len, n_chunks, XCD_ORDER_MAX_BLOCKS, n_desc);
        HIP_TRY(c, launch_order_blocks(a.chunks_tmp, a.bucket, a.sub, n_chunks, nb, reinterpret_cast<uint32_t*>(scratch.ptr() + s_subhist),
                                       reinterpret_cast<uint64_t*>(scratch.ptr() + s_substart), reinterpret_cast<uint64_t*>(scratch.ptr() + s_subtiles),
                                       reinterpret_cast<Chunk*>(scratch.ptr() + s_tmp2), scratch.ptr() + s_bucket2, reinterpret_cast<uint32_t*>(scratch.ptr() + s_hist),
                                       reinterpret_cast<uint32_t*>(scratch.ptr() + s_tot), reinterpret_cast<Chunk*>(b->d_chunks.ptr()), c->stream), "launch(order)");
    } else if (n_chunks) HIP_TRY(c, hipMemcpyAsync(b->d_chunks.ptr(), a.chunks_tmp, n_chunks * sizeof(Chunk), hipMemcpyDeviceToDevice, c->stream), "D2D(chunks)");
    HIP_TRY(c, hipEventRecord(guard.ev[7], c->stream), "hipEventRecord");
    b->img.hap_out_begin.assign(n_h + 1, 0);
    HIP_TRY(c, hipMemcpyAsync(b->img.hap_out_begin.data(), b->d_hap.ptr(), (n_h + 1) * 8, hipMemcpyDeviceToHost, c->stream), "D2H(hap_begin)");
    rc = collect_status(c, b->d_status);
    if (rc) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return rc; }
    float ms0 = 0.f, ms1 = 0.f, ms2 = 0.f;
    (void)hipEventElapsedTime(&ms0, guard.ev[0], guard.ev[1]);
    (void)hipEventElapsedTime(&ms1, guard.ev[4], guard.ev[5]);
    (void)hipEventElapsedTime(&ms2, guard.ev[6], guard.ev[7]);
    if (build_ms) *build_ms = ms0 + ms_parse + ms1 + ms2;
    guard.ok = true;
    b->n_desc = n_desc; b->n_chunks = n_chunks; b->n_payload = v.n_alt; b->payload_dev = v.alt; b->out_bytes = out_bytes; b->n_haps = n_h;
    b->launch_hint = (mode == ROWS_DENSE ? 2 : 4) | 8 | 16 | 32 | (1 << 6) | (1 << 8);
    b->uses_proteome = true;
    b->finalized = true;
    b->n_slices = 0;
    return V2P_OK;
}

// The kernels index device memory through EVERY entry of the stream's offset tables (16-byte slab loads of a transcript's tasks, the
// unaligned 8-byte load of an immediate payload, tx_res_base[hap_tx_begin[h]]): a table that is not ascending from 0 or leaves
// its array is refused here, with the offending index, before anything is uploaded.  Optionally: every transcript's arena length
// (FASTA, the grid builders), the arena offset of every haplotype (res_counter of haplotype_instruction.rs:90,132 on the host).
struct CheckErr { std::string msg; int64_t index = -1; int operator()(int code, const std::string& m, int64_t i = -1) { msg = m; index = i; return code; } };
// (reads the context's header table and nothing else of it: callable without c->mu -- the streamed pipeline's submitters check their slices while the
// runner holds the context through a one call; the error goes to `fail`, the caller hands it to c->fail under the lock)
static int check_stream_nolock(const v2p_ctx* c, const v2p_txstream* s, bool* fasta_out, std::vector<uint32_t>* arena_len, std::vector<uint64_t>* hap_out_begin, CheckErr& fail)
{
    if (!s->hap_tx_begin || (s->n_tx && (!s->tx_proteome_off || !s->tx_ref_len || !s->tx_res_len || !s->tx_task_begin || !s->tx_alt_begin)) ||
        (s->n_tasks && (!s->code || !s->start_pos || !s->length || !s->start_pos_res)) || (s->n_alt && !s->alt))
        return fail(V2P_ERR_INVALID_ARG, "null argument");
    if (s->hap_tx_begin[s->n_haps] != s->n_tx || (s->n_tx && (s->tx_task_begin[s->n_tx] != s->n_tasks || s->tx_alt_begin[s->n_tx] != s->n_alt)))
        return fail(V2P_ERR_INVALID_ARG, "stream offsets do not add up");
    if (s->hap_tx_begin[0] != 0) return fail(V2P_ERR_INVALID_ARG, "hap_tx_begin does not start at 0", 0);
    for (uint64_t h = 0; h < s->n_haps; ++h)
        if (s->hap_tx_begin[h + 1] < s->hap_tx_begin[h] || s->hap_tx_begin[h + 1] > s->n_tx)
            return fail(V2P_ERR_INVALID_ARG, "hap_tx_begin is not ascending inside [0, n_tx] at haplotype " + std::to_string(h), int64_t(h));
    if (s->n_tx && (s->tx_task_begin[0] != 0 || s->tx_alt_begin[0] != 0)) return fail(V2P_ERR_INVALID_ARG, "tx_task_begin / tx_alt_begin do not start at 0", 0);
    for (uint64_t t = 0; t < s->n_tx; ++t) {
        if (s->tx_task_begin[t + 1] < s->tx_task_begin[t] || s->tx_task_begin[t + 1] > s->n_tasks)
            return fail(V2P_ERR_INVALID_ARG, "tx_task_begin is not ascending inside [0, n_tasks] at transcript " + std::to_string(t), int64_t(t));
        if (s->tx_alt_begin[t + 1] < s->tx_alt_begin[t] || s->tx_alt_begin[t + 1] > s->n_alt)
            return fail(V2P_ERR_INVALID_ARG, "tx_alt_begin is not ascending inside [0, n_alt] at transcript " + std::to_string(t), int64_t(t));
        if (s->tx_proteome_off[t] + s->tx_ref_len[t] < s->tx_proteome_off[t])
            return fail(V2P_ERR_INVALID_ARG, "tx_proteome_off + tx_ref_len wraps at transcript " + std::to_string(t), int64_t(t));
    }
    // FASTA emit: every record header inside the resident header table and ending in a line feed (the record's own line feed is read
    // from there); a transcript's arena length is then header + residues + line feed
    const bool fasta = s->tx_header_off && s->tx_header_len;
    if ((s->tx_header_off == nullptr) != (s->tx_header_len == nullptr)) return fail(V2P_ERR_INVALID_ARG, "tx_header_off and tx_header_len come together");
    if (fasta) {
        if (arena_len) arena_len->resize(s->n_tx);
        for (uint64_t t = 0; t < s->n_tx; ++t) {
            const uint64_t ho = s->tx_header_off[t], hl = s->tx_header_len[t];
            if (hl && (ho + hl > c->headers_len || ho + hl < ho)) return fail(V2P_ERR_SRC_OOB, "record header outside the resident header table at transcript " + std::to_string(t), int64_t(t));
            if (hl && c->headers_host[ho + hl - 1] != '\n') return fail(V2P_ERR_INVALID_ARG, "a record header must end in a line feed (transcript " + std::to_string(t) + ")", int64_t(t));
            const uint64_t al = uint64_t(s->tx_res_len[t]) + (hl ? hl + 1u : 0u);
            if (al > 0xFFFFFFFFull) return fail(V2P_ERR_UNSUPPORTED, "a record of more than 4 GiB", int64_t(t));
            if (arena_len) (*arena_len)[t] = uint32_t(al);
        }
    }
    if (hap_out_begin) {
        hap_out_begin->assign(s->n_haps + 1, 0);
        uint64_t at = 0;
        for (uint64_t h = 0; h < s->n_haps; ++h) {
            (*hap_out_begin)[h] = at;
            for (uint64_t t = s->hap_tx_begin[h]; t < s->hap_tx_begin[h + 1]; ++t) {
                const uint32_t hl = fasta ? s->tx_header_len[t] : 0u;
                at += uint64_t(s->tx_res_len[t]) + (hl ? hl + 1u : 0u);
            }
        }
        (*hap_out_begin)[s->n_haps] = at;
    }
    *fasta_out = fasta;
    return V2P_OK;
}

static int check_stream(v2p_ctx* c, const v2p_txstream* s, bool* fasta_out, std::vector<uint32_t>* arena_len, std::vector<uint64_t>* hap_out_begin)
{
    CheckErr err;
    const int rc = check_stream_nolock(c, s, fasta_out, arena_len, hap_out_begin, err);
    return rc == V2P_OK ? V2P_OK : c->fail(rc, err.msg, err.index);
}

// ---- image build on the device ---------------------------------------------------------------------------------------------
int v2p_batch_build_on_device(v2p_batch* b, const v2p_txstream* s, uint32_t window_bytes, int kernel, float* build_ms)
{
    if (!b || !s) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if (b->hap_open || b->img.n_haplotypes()) return c->fail(V2P_ERR_STATE, "the batch already holds host-built haplotypes");
#ifndef V2P_BENCH_VARIANTS
    // the product builds ROWS images (round 4: one pass, whole descriptors, chunks cut on 1 KiB rows afterwards; no window).  The grid builders
    // of rounds 2-3 (kernel 1 .. 5) and PATCH images (8) were never picked by a routing rule: they live in libv2p_bench.so (csrc/bench/v2p_bench.h)
    if (kernel != 6 && kernel != 7) return c->fail(V2P_ERR_INVALID_ARG, "v2p_batch_build_on_device builds rows images: kernel 6 (wave) or 7 (dense)");
    (void)window_bytes;
    const bool rows = true;
#else
    const bool split = kernel == 5;                       // wave windows that may split once (65 .. 127 descriptors -> two chunks)
    if (split) kernel = 4;
    const bool rows = kernel == 6 || kernel == 7 || kernel == 8;   // ROWS images (round 4): one pass, whole descriptors, chunks cut on 1 KiB rows afterwards; no window -- 8: a PATCH image (round 5)
    if (!rows && (window_bytes == 0 || window_bytes % (kernel == 4 ? 1024u : 4096u) || window_bytes > CHUNK_BYTES - 4080u))
        return c->fail(V2P_ERR_INVALID_ARG, "window_bytes must be a multiple of 4096 (wave images: of 1024), at most 61440");
    if (kernel == 1 && window_bytes > CHUNK_BYTES_LONG) return c->fail(V2P_ERR_INVALID_ARG, "the long-run kernel takes windows of at most 32768 bytes");
    if (kernel == 4 && window_bytes > CHUNK_BYTES_WAVE) return c->fail(V2P_ERR_INVALID_ARG, "a wave image takes windows of at most 10240 bytes (one chunk = ten 1 KiB rows of one wave)");
    if (split && window_bytes < 2048u) return c->fail(V2P_ERR_INVALID_ARG, "a window that may split holds at least two 1 KiB rows");
    // (a grid chunk starts on a multiple of 4096: no 16-byte phase, so 12288 bytes fill the kernel's LDS image exactly)
    if (kernel == 3 && window_bytes > 12288u) return c->fail(V2P_ERR_INVALID_ARG, "a dense image takes windows of 4096, 8192 or 12288 bytes (one chunk = one 12 KiB LDS image)");
#endif
    bool fasta = false;
    std::vector<uint32_t> arena_len;
    {
        const int crc = check_stream(c, s, &fasta, rows ? nullptr : &arena_len, nullptr);
        if (crc) return crc;
    }
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (rows) {
        DevStreamView v;
        const int urc = upload_stream(c, s, fasta, b->d_build, b->d_payload, v, c->stream);
        if (urc) { b->d_build.release(); return urc; }
#ifdef V2P_BENCH_VARIANTS
        if (kernel == 8) return build_patch_image(b, v, build_ms, true, ~0ull);
#endif
        return build_rows_image(b, v, kernel == 7 ? ROWS_DENSE : ROWS_WAVE, build_ms, true);
    }
#ifndef V2P_BENCH_VARIANTS
    return c->fail(V2P_ERR_INVALID_ARG, "kernel");
#else
    const uint64_t n_tx = s->n_tx, n_tk = s->n_tasks, n_h = s->n_haps;
    const uint64_t n_tiles = (n_tx + 1023) / 1024 + 2;
    // one device allocation, carved: stream arrays, then scratch
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    const uint64_t o_hap = carve((n_h + 1) * 8), o_poff = carve(n_tx * 8), o_rlen = carve(n_tx * 4), o_res = carve(n_tx * 4),
                   o_tb = carve((n_tx + 1) * 8), o_ab = carve((n_tx + 1) * 8), o_code = carve(n_tk), o_sp = carve(n_tk * 4), o_ln = carve(n_tk * 4),
                   o_sr = carve(n_tk * 4), o_base = carve((n_tx + 1) * 8), o_cnt = carve(n_tx * 4), o_dbase = carve((n_tx + 1) * 8),
                   o_tiles = carve(n_tiles * 8), o_meta = carve(32),
                   o_hoff = carve(fasta ? n_tx * 8 : 0), o_hlen = carve(fasta ? n_tx * 4 : 0), o_alen = carve(fasta ? n_tx * 4 : 0);
    HIP_TRY(c, b->d_build.ensure(off), "hipMalloc(build)");
    uint8_t* const d = b->d_build.ptr();
    HIP_TRY(c, b->d_payload.ensure(s->n_alt), "hipMalloc(alt)");
#define UP(dst_off, src, bytes, what) do { if (bytes) HIP_TRY(c, hipMemcpyAsync(d + (dst_off), (src), (bytes), hipMemcpyHostToDevice, c->stream), what); } while (0)
    UP(o_hap, s->hap_tx_begin, (n_h + 1) * 8, "H2D(hap_tx_begin)"); UP(o_poff, s->tx_proteome_off, n_tx * 8, "H2D(tx_proteome_off)");
    UP(o_rlen, s->tx_ref_len, n_tx * 4, "H2D(tx_ref_len)"); UP(o_res, s->tx_res_len, n_tx * 4, "H2D(tx_res_len)");
    UP(o_tb, s->tx_task_begin, (n_tx + 1) * 8, "H2D(tx_task_begin)"); UP(o_ab, s->tx_alt_begin, (n_tx + 1) * 8, "H2D(tx_alt_begin)");
    UP(o_code, s->code, n_tk, "H2D(code)"); UP(o_sp, s->start_pos, n_tk * 4, "H2D(start_pos)"); UP(o_ln, s->length, n_tk * 4, "H2D(length)");
    UP(o_sr, s->start_pos_res, n_tk * 4, "H2D(start_pos_res)");
    if (fasta) { UP(o_hoff, s->tx_header_off, n_tx * 8, "H2D(tx_header_off)"); UP(o_hlen, s->tx_header_len, n_tx * 4, "H2D(tx_header_len)"); UP(o_alen, arena_len.data(), n_tx * 4, "H2D(arena_len)"); }
#undef UP
    if (s->n_alt) HIP_TRY(c, hipMemcpyAsync(b->d_payload.ptr(), s->alt, s->n_alt, hipMemcpyHostToDevice, c->stream), "H2D(alt)");
    HIP_TRY(c, hipMemsetAsync(d + o_meta, 0, 32, c->stream), "hipMemset(meta)");
    int rc = init_status(c, b->d_status);
    if (rc) return rc;
    // two brackets: the counting kernels, then -- after the host has read the two totals and allocated the image -- the emitting ones
    // (every return below leaves through these: the events, the scratch tables and the uploaded stream are released, and a batch
    // that failed holds nothing -- the caller may retry it with a smaller window)
    struct Cleanup {
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        DevBuf scratch;
        v2p_batch* b;
        bool ok = false;
        explicit Cleanup(v2p_batch* b_) : b(b_) {}
        ~Cleanup() {
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            scratch.release();
            b->d_build.release();                         // nothing reads the stream copy after the emit pass
            if (!ok) { b->img.hap_out_begin.assign(1, 0); b->n_desc = b->n_chunks = b->n_payload = b->out_bytes = b->n_haps = 0; }
        }
    } guard(b);
    for (hipEvent_t& e : guard.ev) HIP_TRY(c, hipEventCreate(&e), "hipEventCreate");
    hipEvent_t &e0 = guard.ev[0], &e1 = guard.ev[1], &e2 = guard.ev[2], &e3 = guard.ev[3];
    BuildArgs a{};
    a.n_haps = n_h; a.n_tx = n_tx; a.n_tasks = s->n_tasks;
    a.hap_tx_begin = reinterpret_cast<const uint64_t*>(d + o_hap); a.tx_proteome_off = reinterpret_cast<const uint64_t*>(d + o_poff);
    a.tx_ref_len = reinterpret_cast<const uint32_t*>(d + o_rlen); a.tx_res_len = reinterpret_cast<const uint32_t*>(d + o_res);
    a.tx_task_begin = reinterpret_cast<const uint64_t*>(d + o_tb); a.tx_alt_begin = reinterpret_cast<const uint64_t*>(d + o_ab);
    a.code = d + o_code; a.start_pos = reinterpret_cast<const uint32_t*>(d + o_sp); a.length = reinterpret_cast<const uint32_t*>(d + o_ln);
    a.start_pos_res = reinterpret_cast<const uint32_t*>(d + o_sr); a.alt = b->d_payload.ptr();
    a.tx_header_off = fasta ? reinterpret_cast<const uint64_t*>(d + o_hoff) : nullptr;
    a.tx_header_len = fasta ? reinterpret_cast<const uint32_t*>(d + o_hlen) : nullptr;
    a.proteome_len = c->proteome_len; a.window = window_bytes; a.long_run = kernel == 1 || kernel == 4; a.dense = kernel == 3; a.wave = kernel == 4; a.split = split;
    a.tx_res_base = reinterpret_cast<const uint64_t*>(d + o_base); a.tx_desc_count = reinterpret_cast<uint32_t*>(d + o_cnt);
    a.desc_base = reinterpret_cast<const uint64_t*>(d + o_dbase); a.meta = reinterpret_cast<uint32_t*>(d + o_meta);
    a.status = reinterpret_cast<unsigned long long*>(b->d_status.ptr());
    // step 5: res_counter as a prefix scan; then count, scan, emit
    HIP_TRY(c, hipEventRecord(e0, c->stream), "hipEventRecord");
    HIP_TRY(c, launch_scan_u32(fasta ? reinterpret_cast<const uint32_t*>(d + o_alen) : a.tx_res_len, n_tx, reinterpret_cast<uint64_t*>(d + o_base), reinterpret_cast<uint64_t*>(d + o_tiles), c->stream), "launch(scan)");
    HIP_TRY(c, launch_build(a, 0, 0, 0, 0, c->stream), "launch(count)");
    HIP_TRY(c, launch_scan_u32(a.tx_desc_count, n_tx, reinterpret_cast<uint64_t*>(d + o_dbase), reinterpret_cast<uint64_t*>(d + o_tiles), c->stream), "launch(scan)");
    HIP_TRY(c, hipEventRecord(e1, c->stream), "hipEventRecord");
    uint64_t totals[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&totals[0], d + o_base + n_tx * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(out_bytes)");
    HIP_TRY(c, hipMemcpyAsync(&totals[1], d + o_dbase + n_tx * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(n_desc)");
    rc = collect_status(c, b->d_status);                  // what the reference would panic on surfaces here, before anything is emitted
    if (rc) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return rc; }
    const uint64_t out_bytes = totals[0], n_desc = totals[1];
    const uint64_t n_windows = (out_bytes + window_bytes - 1) / window_bytes;
    if (n_windows > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch");
    DevBuf& scratch = guard.scratch;                      // chunk_first, chunks in result order, slices, per-block histograms
    const uint64_t cap = split ? 2 * n_windows : n_windows;   // chunk slots: a window that splits adds one behind the n_windows first ones
    if (cap > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch");
    const uint64_t n_blocks_cap = order_blocks_thread_blocks(cap, XCD_ORDER_MAX_BLOCKS);   // thread blocks of the two sorts, all blocks of the table together
    const uint64_t n_sub_cap = uint64_t(XCD_SUB) * n_blocks_cap;      // counters of the window sort
    const uint64_t s_first = 0, s_tmp = up8(n_windows * 8), s_bucket = s_tmp + up8(cap * 16), s_hist = s_bucket + up8(cap),
                   s_sub = s_hist + up8((n_blocks_cap + 1) * 8 * 4), s_tmp2 = s_sub + up8(cap), s_bucket2 = s_tmp2 + up8(cap * 16),
                   s_subhist = s_bucket2 + up8(cap), s_substart = s_subhist + up8(n_sub_cap * 4), s_subtiles = s_substart + up8((n_sub_cap + 1) * 8),
                   s_tot = s_subtiles + up8(scan_tiles_for(n_sub_cap) * 8), s_end = s_tot + up8(uint64_t(XCD_ORDER_MAX_BLOCKS) * 8 * 4);
    HIP_TRY(c, scratch.ensure(s_end), "hipMalloc(build scratch)");
    HIP_TRY(c, b->d_desc.ensure(n_desc * 8), "hipMalloc(desc)");
    HIP_TRY(c, b->d_chunks.ensure(cap * sizeof(Chunk)), "hipMalloc(chunks)");
    HIP_TRY(c, b->d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_hap.ensure((n_h + 1) * 8), "hipMalloc(hap_begin)");
    HIP_TRY(c, b->d_digest.ensure((n_h ? n_h : 1) * 8), "hipMalloc(digest)");
    a.desc = reinterpret_cast<uint64_t*>(b->d_desc.ptr());
    a.chunk_first = reinterpret_cast<uint64_t*>(scratch.ptr() + s_first);
    a.chunks_tmp = reinterpret_cast<Chunk*>(scratch.ptr() + s_tmp);
    a.bucket = scratch.ptr() + s_bucket;
    a.sub = scratch.ptr() + s_sub;
    a.hap_out_begin = reinterpret_cast<uint64_t*>(b->d_hap.ptr());
    HIP_TRY(c, hipEventRecord(e2, c->stream), "hipEventRecord");
    HIP_TRY(c, launch_build(a, n_windows, n_desc, out_bytes, 1, c->stream), "launch(emit)");
    uint64_t n_chunks = n_windows;
    if (split) {                                          // how many windows split: the sorts below run over every chunk
        uint32_t extra = 0;
        HIP_TRY(c, hipMemcpyAsync(&extra, d + o_meta + 16, 4, hipMemcpyDeviceToHost, c->stream), "D2H(split windows)");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        n_chunks = n_windows + extra;
    }
    const bool reorder = !(c->flags & V2P_FLAG_RESULT_ORDER) && n_chunks >= 16 && c->proteome_len != 0 && n_desc != 0;
    if (reorder) {
        // the XCD / window order inside blocks of the arena (sir_pack.hpp: order_chunks_for_xcds): the first n_windows entries are in
        // arena order and are dealt inside their blocks, all blocks in one launch per sort step (launch_order_blocks); the second
        // chunks of split windows (behind them, in no particular order) are one more range
        Chunk* by_window = reinterpret_cast<Chunk*>(scratch.ptr() + s_tmp2);
        const uint32_t nb = xcd_order_blocks(n_windows ? (n_windows - 1) * uint64_t(window_bytes) : 0, c->proteome_len, n_windows, XCD_ORDER_MAX_BLOCKS, n_desc);   // (span = the last window's offset: what the host rule sees)
        HIP_TRY(c, launch_order_blocks(a.chunks_tmp, a.bucket, a.sub, n_windows, nb, reinterpret_cast<uint32_t*>(scratch.ptr() + s_subhist),
                                       reinterpret_cast<uint64_t*>(scratch.ptr() + s_substart), reinterpret_cast<uint64_t*>(scratch.ptr() + s_subtiles),
                                       by_window, scratch.ptr() + s_bucket2, reinterpret_cast<uint32_t*>(scratch.ptr() + s_hist),
                                       reinterpret_cast<uint32_t*>(scratch.ptr() + s_tot), reinterpret_cast<Chunk*>(b->d_chunks.ptr()), c->stream), "launch(order)");
        if (n_chunks > n_windows) {                           // the second chunks of split windows: one more range, dealt by itself
            const uint64_t k0 = n_windows, nt = n_chunks - n_windows;
            if (nt < 16) HIP_TRY(c, hipMemcpyAsync(b->d_chunks.ptr() + k0 * sizeof(Chunk), a.chunks_tmp + k0, nt * sizeof(Chunk), hipMemcpyDeviceToDevice, c->stream), "D2D(chunks)");
            else {
                HIP_TRY(c, launch_sub_order(a.chunks_tmp + k0, a.bucket + k0, a.sub + k0, nt, reinterpret_cast<uint32_t*>(scratch.ptr() + s_subhist),
                                            reinterpret_cast<uint64_t*>(scratch.ptr() + s_substart), reinterpret_cast<uint64_t*>(scratch.ptr() + s_subtiles),
                                            by_window, scratch.ptr() + s_bucket2, c->stream), "launch(window order)");
                HIP_TRY(c, launch_xcd_order(by_window, scratch.ptr() + s_bucket2, nt, reinterpret_cast<uint32_t*>(scratch.ptr() + s_hist),
                                            reinterpret_cast<Chunk*>(b->d_chunks.ptr()) + k0, c->stream), "launch(xcd order)");
            }
        }
    }
    else if (n_chunks) HIP_TRY(c, hipMemcpyAsync(b->d_chunks.ptr(), a.chunks_tmp, n_chunks * sizeof(Chunk), hipMemcpyDeviceToDevice, c->stream), "D2D(chunks)");
    HIP_TRY(c, hipEventRecord(e3, c->stream), "hipEventRecord");
    uint32_t meta[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(meta, d + o_meta, 16, hipMemcpyDeviceToHost, c->stream), "D2H(meta)");
    b->img.hap_out_begin.assign(n_h + 1, 0);
    HIP_TRY(c, hipMemcpyAsync(b->img.hap_out_begin.data(), b->d_hap.ptr(), (n_h + 1) * 8, hipMemcpyDeviceToHost, c->stream), "D2H(hap_begin)");
    rc = collect_status(c, b->d_status);
    float ms = 0.f, ms2 = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventElapsedTime(&ms2, e2, e3);
    ms += ms2;
    if (rc) { (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream); return rc; }
    guard.ok = true;
    if (build_ms) *build_ms = ms;
    b->n_desc = n_desc; b->n_chunks = n_chunks; b->n_payload = s->n_alt; b->payload_dev = b->d_payload.ptr(); b->out_bytes = out_bytes; b->n_haps = n_h;
    b->n_slices = 0;
    const int tpt = meta[3] <= 256u ? 1 : (meta[3] <= 512u ? 2 : 4);
    b->launch_hint = ((meta[0] & 2u) ? 2 : 0) | ((meta[0] & 4u) ? 4 : 0) | ((meta[0] & 1u) ? 0 : 16) | (meta[2] ? 0 : 32) | ((meta[1] ? 2 : 1) << 6) | (tpt << 8);
    b->uses_proteome = true;
    b->finalized = true;
    return V2P_OK;
#endif   // V2P_BENCH_VARIANTS: the grid builders
}

#ifdef V2P_BENCH_VARIANTS
// ---- PATCH images (patch_image.h; kernel 8): deep Task vectors as segments + patches, ONE build kernel, one workgroup per 8 KiB window ----
// V2P_ERR_UNSUPPORTED: the format declines the stream (a window with more segments / patches than its slots, sources beyond 16 GB) --
// the batch is left empty and the caller builds a dense rows image (kernel 7) instead.
static int build_patch_image(v2p_batch* b, const DevStreamView& v, float* build_ms, bool own_stream_copy, uint64_t known_out_bytes)
{
    v2p_ctx* c = b->ctx;
    const uint64_t n_tx = v.n_tx, n_h = v.n_haps;
    struct Cleanup {
        hipEvent_t ev[2] = {nullptr, nullptr};
        DevBuf scratch;
        v2p_batch* b;
        bool ok = false, own;
        Cleanup(v2p_batch* b_, bool own_) : b(b_), own(own_) {}
        ~Cleanup() {
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            scratch.release();
            if (own) b->d_build.release();
            if (!ok) { b->img.hap_out_begin.assign(1, 0); b->n_desc = b->n_chunks = b->n_payload = b->out_bytes = b->n_haps = 0; b->payload_dev = nullptr; b->is_patch = false; }
        }
    } guard(b, own_stream_copy);
    if (c->proteome_len + c->headers_len > PATCH_SRC_MAX || v.n_alt > PATCH_SRC_MAX) return c->fail(V2P_ERR_UNSUPPORTED, "a patch image addresses 16 GB of reference / alt bytes");
    int rc = init_status(c, b->d_status);
    if (rc) return rc;
    for (hipEvent_t& e : guard.ev) HIP_TRY(c, hipEventCreate(&e), "hipEventCreate");
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    // positions first: the arena's size decides everything else
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    const uint64_t o_alen = carve((n_tx + 1) * 4), o_rbase = carve((n_tx + 2) * 8), o_scan = carve(scan_tiles_for(n_tx + 1) * 8), o_totals = carve(64);
    // (the chunk-level scratch follows once out_bytes is known; sized here for the largest arena the stream's result lengths allow)
    PatchBuildArgs a{};
    a.n_tx = n_tx; a.n_tasks = v.n_tasks; a.n_haps = n_h;
    a.tx_proteome_off = v.tx_proteome_off; a.tx_ref_len = v.tx_ref_len; a.tx_res_len = v.tx_res_len; a.tx_task_begin = v.tx_task_begin; a.tx_alt_begin = v.tx_alt_begin;
    a.code = v.code; a.start_pos = v.start_pos; a.length = v.length; a.start_pos_res = v.start_pos_res; a.alt = v.alt;
    a.tx_header_off = v.tx_header_off; a.tx_header_len = v.tx_header_len; a.hap_tx_begin = v.hap_tx_begin;
    a.proteome_len = c->proteome_len; a.alt_len = v.n_alt;
    a.status = reinterpret_cast<unsigned long long*>(b->d_status.ptr());
    DevBuf& pos = b->d_tiles;                            // (recycled by a batch that is rebuilt)
    HIP_TRY(c, pos.ensure_exact(off), "hipMalloc(positions)");
    uint8_t* const d = pos.ptr();
    a.tx_arena_len = reinterpret_cast<uint32_t*>(d + o_alen); a.tx_res_base = reinterpret_cast<uint64_t*>(d + o_rbase);
    a.totals = reinterpret_cast<uint64_t*>(d + o_totals);
    HIP_TRY(c, hipEventRecord(guard.ev[0], c->stream), "hipEventRecord");
    HIP_TRY(c, hipMemsetAsync(d + o_totals, 0, 64, c->stream), "hipMemset(totals)");
    HIP_TRY(c, launch_patch_positions(a, reinterpret_cast<uint64_t*>(d + o_scan), c->stream), "launch(positions)");
    uint64_t out_bytes = known_out_bytes;
    if (known_out_bytes == ~0ull) {
        HIP_TRY(c, hipMemcpyAsync(&out_bytes, d + o_rbase + n_tx * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(out_bytes)");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    const uint64_t n_chunks = (out_bytes + PATCH_G - 1) / PATCH_G;
    if (n_chunks > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch");
    a.out_bytes = out_bytes; a.n_chunks = n_chunks;
    const uint64_t cap = n_chunks ? n_chunks : 1;
    const uint64_t n_blocks_cap = order_blocks_thread_blocks(cap, XCD_ORDER_MAX_BLOCKS);
    const uint64_t n_sub_cap = uint64_t(XCD_SUB) * n_blocks_cap;
    const uint64_t s_tmp = 0, s_bucket = s_tmp + up8(cap * 16), s_hist = s_bucket + up8(cap),
                   s_sub = s_hist + up8((n_blocks_cap + 1) * 8 * 4), s_tmp2 = s_sub + up8(cap), s_bucket2 = s_tmp2 + up8(cap * 16),
                   s_subhist = s_bucket2 + up8(cap), s_substart = s_subhist + up8(n_sub_cap * 4), s_subtiles = s_substart + up8((n_sub_cap + 1) * 8),
                   s_tot = s_subtiles + up8(scan_tiles_for(n_sub_cap) * 8), s_ctx = s_tot + up8(uint64_t(XCD_ORDER_MAX_BLOCKS) * 8 * 4), s_end = s_ctx + up8((cap + 2) * 8);
    DevBuf& scratch = b->d_order;
    HIP_TRY(c, scratch.ensure_exact(s_end), "hipMalloc(build scratch)");
    a.chunk_tx = reinterpret_cast<uint64_t*>(scratch.ptr() + s_ctx);
    HIP_TRY(c, b->d_desc.ensure_exact(cap * PATCH_SEG_CAP * 8), "hipMalloc(segments)");
    HIP_TRY(c, b->d_patch.ensure_exact(cap * PATCH_PATCH_CAP * 4), "hipMalloc(patches)");
    HIP_TRY(c, b->d_chunks.ensure(cap * sizeof(Chunk)), "hipMalloc(chunks)");
    HIP_TRY(c, b->d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_hap.ensure((n_h + 1) * 8), "hipMalloc(hap_begin)");
    HIP_TRY(c, b->d_digest.ensure((n_h ? n_h : 1) * 8), "hipMalloc(digest)");
    uint8_t* const sc = scratch.ptr();
    a.seg = reinterpret_cast<uint64_t*>(b->d_desc.ptr()); a.patch = reinterpret_cast<uint32_t*>(b->d_patch.ptr());
    a.chunks = reinterpret_cast<Chunk*>(sc + s_tmp); a.bucket = sc + s_bucket; a.sub = sc + s_sub;
    a.hap_out_begin = reinterpret_cast<uint64_t*>(b->d_hap.ptr());
    HIP_TRY(c, launch_patch_hap_begin(a, c->stream), "launch(hap_begin)");
    HIP_TRY(c, launch_patch_build(a, c->stream), "launch(patch build)");
    const bool reorder = !(c->flags & V2P_FLAG_RESULT_ORDER) && n_chunks >= 16 && c->proteome_len != 0;
    if (reorder) {
        // (the XCD / window order of the chunk table, as for every other image; a declined or failed build is noticed below, before anything executes)
        const uint32_t nb = xcd_order_blocks(out_bytes, c->proteome_len, n_chunks, XCD_ORDER_MAX_BLOCKS, n_chunks * 400u);
        HIP_TRY(c, launch_order_blocks(a.chunks, a.bucket, a.sub, n_chunks, nb, reinterpret_cast<uint32_t*>(sc + s_subhist),
                                       reinterpret_cast<uint64_t*>(sc + s_substart), reinterpret_cast<uint64_t*>(sc + s_subtiles),
                                       reinterpret_cast<Chunk*>(sc + s_tmp2), sc + s_bucket2, reinterpret_cast<uint32_t*>(sc + s_hist),
                                       reinterpret_cast<uint32_t*>(sc + s_tot), reinterpret_cast<Chunk*>(b->d_chunks.ptr()), c->stream), "launch(order)");
    } else if (n_chunks) HIP_TRY(c, hipMemcpyAsync(b->d_chunks.ptr(), a.chunks, n_chunks * sizeof(Chunk), hipMemcpyDeviceToDevice, c->stream), "D2D(chunks)");
    HIP_TRY(c, hipEventRecord(guard.ev[1], c->stream), "hipEventRecord");
    uint64_t totals[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(totals, d + o_totals, 16, hipMemcpyDeviceToHost, c->stream), "D2H(totals)");
    b->img.hap_out_begin.assign(n_h + 1, 0);
    HIP_TRY(c, hipMemcpyAsync(b->img.hap_out_begin.data(), b->d_hap.ptr(), (n_h + 1) * 8, hipMemcpyDeviceToHost, c->stream), "D2H(hap_begin)");
    unsigned long long st = STATUS_CLEAN;
    HIP_TRY(c, hipMemcpyAsync(&st, b->d_status.ptr(), 8, hipMemcpyDeviceToHost, c->stream), "D2H(status)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st != STATUS_CLEAN) {
        (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, c->stream);
        if (uint32_t(st & 0xFFu) == STATUS_PATCH_DECLINED) return c->fail(V2P_ERR_UNSUPPORTED, "a 8 KiB window of the result holds more segments or patches than a patch image's chunk (kernel 7 builds a dense image)", int64_t(st >> 8));
        const int code = reason_to_err(uint32_t(st & 0xFFu));          // what the reference would panic on
        return c->fail(code, std::string("device: ") + err_name(code) + " at descriptor " + std::to_string(st >> 8), int64_t(st >> 8));
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, guard.ev[0], guard.ev[1]);
    if (build_ms) *build_ms = ms;
    guard.ok = true;
    b->n_desc = totals[0]; b->patch_segs = totals[0]; b->patch_patches = totals[1];
    b->n_chunks = n_chunks; b->n_payload = v.n_alt; b->payload_dev = v.alt; b->out_bytes = out_bytes; b->n_haps = n_h;
    b->launch_hint = 0; b->is_patch = true;
    b->uses_proteome = true;
    b->finalized = true;
    b->n_slices = 0;
    return V2P_OK;
}
#endif   // V2P_BENCH_VARIANTS: patch images

// ---- a transcript stream RESIDENT on the device; Task vectors -> result bytes in ONE call -----------------------------------
// v2p_batch_build_on_device takes a host stream, uploads it, builds, and the caller executes afterwards: two calls, the stream's H2D
// inside the first, the build and the execute strictly one after the other.  What a cohort pays ONCE is exactly that sequence
// (haplotype_instruction.rs:75-137 -> gir.rs:197-241), so here it is one call on a stream that is already in HBM:
//   v2p_stream_upload             tables checked on the host (check_stream), arrays to the device once, arena offsets of the haplotypes
//   v2p_batch_build_from_stream   the one-piece builder (build_rows_image) on it: no H2D
//   v2p_batch_build_and_execute   the image built SLICE BY SLICE on a second HIP stream while the slice before it is being stitched on the
//                                 context's stream: the parse is bound by instruction issue, the stitch by its stores
struct v2p_stream {
    v2p_ctx* ctx = nullptr;
    DevBuf buf, alt;
    DevStreamView v;
    uint64_t out_bytes = 0;
    std::vector<uint64_t> hap_out_begin;       // res_counter (haplotype_instruction.rs:90,132) at every haplotype's first transcript, from the host tables
    // res_counter per TILE of the rows builder (K transcripts: rows_pick_k for the image kind the routing rule picks) and per haplotype, on
    // the device: tables of the stream, made once behind its upload -- v2p_batch_build_and_execute then starts with the parse
    DevBuf tiles;
    uint32_t tile_K = 0;
    uint64_t n_tiles = 0;
    const uint64_t* tile_res_base = nullptr;   // [n_tiles + 1]
    std::vector<uint64_t> h_tile_res_base;     // ... and on the host (a call that builds in slices cuts at these offsets without asking the device)
    const uint64_t* d_hap_out_begin = nullptr; // [n_haps + 1]
    std::vector<v2p_batch*> batches;           // the batches whose images read this stream's alt bytes (under ctx->mu)
};

// every stream a context launches on: its own (or the caller's, v2p_set_stream) and the one call's build / compaction streams -- NOT the
// device: other contexts of the process keep running (the finding of round 4 for v2p_init, of round 5 for v2p_stream_destroy)
static void sync_ctx_streams(v2p_ctx* c)
{
    (void)hipStreamSynchronize(c->stream);
    if (c->stream != c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->build_stream) (void)hipStreamSynchronize(c->build_stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->exec_aux) (void)hipStreamSynchronize(c->exec_aux);
}
static void stream_detach(v2p_batch* b)
{
    if (v2p_stream* st = b->from_stream) {
        for (size_t i = 0; i < st->batches.size(); ++i) if (st->batches[i] == b) { st->batches[i] = st->batches.back(); st->batches.pop_back(); break; }
        b->from_stream = nullptr;
    }
}
static void stream_attach(v2p_batch* b, const v2p_stream* st)
{
    stream_detach(b);
    v2p_stream* s = const_cast<v2p_stream*>(st);
    s->batches.push_back(b);
    b->from_stream = s; b->orphaned = false;
}

static int rows_mode_for(const v2p_stream* st, int kernel);
static int build_tiles(v2p_batch* b, const v2p_stream* st, bool execute, bool* fallback, float* build_ms, uint32_t n_slices = 1);

int v2p_stream_upload(v2p_ctx* c, const v2p_txstream* s, v2p_stream** out)
{
    if (!c || !s || !out) return V2P_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(c->mu);
    v2p_stream* st = new (std::nothrow) v2p_stream();
    if (!st) return c->fail(V2P_ERR_HIP, "out of host memory");
    st->ctx = c;
    bool fasta = false;
    // The tables are checked BESIDE the upload (a thread of its own: C3 whole's 20 M transcripts take it 95 ms, the copy 170): what the
    // kernels index device memory through is refused before anything is built from it, and a stream that fails has been copied for
    // nothing.  Null arrays are caught first -- the copy never reads through one -- and the Task arrays are only sampled behind the check.
    int rc = V2P_OK;
    const bool nulls = !s->hap_tx_begin || (s->n_tx && (!s->tx_proteome_off || !s->tx_ref_len || !s->tx_res_len || !s->tx_task_begin || !s->tx_alt_begin)) ||
                       (s->n_tasks && (!s->code || !s->start_pos || !s->length || !s->start_pos_res)) || (s->n_alt && !s->alt) ||
                       ((s->tx_header_off == nullptr) != (s->tx_header_len == nullptr));
    if (nulls) rc = check_stream(c, s, &fasta, nullptr, &st->hap_out_begin);          // (it says which)
    if (rc == V2P_OK && nulls) rc = c->fail(V2P_ERR_INVALID_ARG, "null argument");
    if (rc == V2P_OK && hipSetDevice(c->device) != hipSuccess) rc = c->fail(V2P_ERR_HIP, "hipSetDevice");
    if (rc == V2P_OK) {
        CheckErr cerr;
        int crc = V2P_OK;
        auto check = [&] {
            try { crc = check_stream_nolock(c, s, &fasta, nullptr, &st->hap_out_begin, cerr); }
            catch (...) { crc = cerr(V2P_ERR_HIP, "out of host memory"); }
        };
        std::thread checker;
        try { checker = std::thread(check); } catch (...) { check(); }                  // (no thread to be had: checked first, as before)
        if (checker.joinable() || crc == V2P_OK) rc = upload_stream(c, s, s->tx_header_off && s->tx_header_len, st->buf, st->alt, st->v, c->stream, false);
        if (checker.joinable()) checker.join();
        if (crc != V2P_OK) rc = c->fail(crc, cerr.msg, cerr.index);                    // (the check's verdict first: it is what the caller can act on)
        else if (rc == V2P_OK) stream_item_stats(s, st->v);
    }
    if (rc == V2P_OK) {
        // the tile tables of the image kind a call with kernel = 0 builds (a call that asks for the other kind makes its own)
        st->out_bytes = st->hap_out_begin.back();
        const uint32_t K = rows_pick_k(st->v, rows_mode_for(st, 0));
        const uint64_t n_tiles = st->v.n_tx ? (st->v.n_tx + K - 1) / K : 1;
        auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
        const uint64_t o_tbytes = 0, o_tbase = up8(n_tiles * 8), o_hap = o_tbase + up8((n_tiles + 1) * 8), o_scan = o_hap + up8((st->v.n_haps + 1) * 8),
                       o_end = o_scan + up8(rows_scan_scratch_entries(n_tiles) * 8);
        if (st->tiles.ensure_exact(o_end) != hipSuccess) { (void)hipGetLastError(); }    // (no room: the calls make their own tables)
        else {
            RowsArgs a;
            rows_args_of(st->v, c, K, n_tiles, a);
            uint8_t* const d = st->tiles.ptr();
            a.tile_bytes = reinterpret_cast<uint64_t*>(d + o_tbytes); a.tile_res_base = reinterpret_cast<uint64_t*>(d + o_tbase);
            a.hap_out_begin = reinterpret_cast<uint64_t*>(d + o_hap);
            a.status = nullptr;                                    // (a tile of more than 2 GiB is found again, and reported, by the parse)
            hipError_t e = launch_rows_tile_bytes(a, reinterpret_cast<uint64_t*>(d + o_scan), c->stream);
            if (e == hipSuccess) e = launch_rows_hap_begin(a, c->stream);
            if (e != hipSuccess) rc = c->hip_fail(e, "launch(tile tables)");
            else {
                st->tile_K = K; st->n_tiles = n_tiles; st->tile_res_base = a.tile_res_base; st->d_hap_out_begin = a.hap_out_begin;
                st->h_tile_res_base.assign(n_tiles + 1, 0);
                if (hipMemcpyAsync(st->h_tile_res_base.data(), a.tile_res_base, (n_tiles + 1) * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); st->h_tile_res_base.clear(); }
            }
        }
    }
    if (rc == V2P_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = c->fail(V2P_ERR_HIP, "hipStreamSynchronize");
    if (rc != V2P_OK) { st->buf.release(); st->alt.release(); st->tiles.release(); delete st; return rc; }
    *out = st;
    return V2P_OK;
}

void v2p_stream_destroy(v2p_stream* st)
{
    if (!st) return;
    std::lock_guard<std::mutex> lk(st->ctx->mu);
    (void)hipSetDevice(st->ctx->device);
    // batches built from it may still execute on the context's streams (their payload descriptors read its alt bytes): those streams are
    // waited for, and the batches are orphaned -- v2p_batch_execute on one is V2P_ERR_STATE from here on, never a read of freed memory
    sync_ctx_streams(st->ctx);
    for (v2p_batch* b : st->batches) { b->from_stream = nullptr; b->orphaned = true; b->payload_dev = nullptr; }
    st->batches.clear();
    st->buf.release(); st->alt.release(); st->tiles.release();
    delete st;
}

int v2p_stream_counts(const v2p_stream* st, uint64_t* n_haps, uint64_t* n_tx, uint64_t* n_tasks, uint64_t* out_bytes)
{
    if (!st) return V2P_ERR_INVALID_ARG;
    if (n_haps) *n_haps = st->v.n_haps;
    if (n_tx) *n_tx = st->v.n_tx;
    if (n_tasks) *n_tasks = st->v.n_tasks;
    if (out_bytes) *out_bytes = st->out_bytes;
    return V2P_OK;
}

// a TILE image takes sources below 2 GiB (a piece's 31-bit source field) and transcripts a tile can hold
static bool tiles_possible(const v2p_stream* st)
{
    const v2p_ctx* c = st->ctx;
    uint32_t slots;
    return c->proteome_len + c->headers_len + 64u <= PIECE_SRC_MAX && st->v.n_alt + 64u <= PIECE_SRC_MAX && st->v.n_tx != 0 && rows_pick_k_tiles(st->v, &slots) != 0u;
}
// kernel 0: the routing rule (sir_pack.hpp: WAVE_BYTES_PER_TASK result bytes per Task and more -> a wave image; below -- deep Task
// vectors -- a TILE image where the form takes the stream, else a dense rows image; libv2p_bench.so's variant 28 (A/B): never a tile image)
static int rows_mode_for(const v2p_stream* st, int kernel)
{
    if (kernel == 6) return ROWS_WAVE;
    if (kernel == 7) return ROWS_DENSE;
    if (kernel == 9) return ROWS_TILES;
    const double bpt = double(st->out_bytes) / double(st->v.n_tasks ? st->v.n_tasks : 1);
    if (bpt >= double(WAVE_BYTES_PER_TASK)) return ROWS_WAVE;
    return ctx_variant(st->ctx) != 28u && tiles_possible(st) ? ROWS_TILES : ROWS_DENSE;
}

// (c->mu held.  wait: the context's stream is waited for -- the streamed pipeline's runner recycles a slot's batch whose last
// kernels and copies are known to be done, and must not wait for the slice another slot has on the GPU)
static int batch_reset_locked(v2p_batch* b, bool wait)
{
    v2p_ctx* c = b->ctx;
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (wait) HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    b->img = ImageBuilder();
    b->finalized = false; b->uses_proteome = false; b->hap_open = false;
    b->n_desc = b->n_chunks = b->n_payload = b->out_bytes = b->n_haps = 0;
    b->payload_dev = nullptr; b->n_slices = 0; b->launch_hint = 0; b->is_patch = false; b->patch_segs = b->patch_patches = 0;
    b->is_tiles = false; b->tile_slots = 0; b->tiles_n = 0; b->tiles_count = nullptr; b->tiles_res_base = nullptr;
    b->pad_image = false; b->desc_slots = 0; b->pad_tdbase = nullptr; b->pieces_state = 0; b->executed = false;
    b->os_kernel = 0; b->os_build_ms = 0.f; b->os_wall_ms = 0.0; b->os_ahead = false;      // (v2p_batch_oneshot_info: no call to report on)
    stream_detach(b); b->orphaned = false;
    if (b->desc_swapped) { std::swap(b->d_desc, b->d_pad); b->desc_swapped = false; }     // (the large allocation is the padded array's again)
    return V2P_OK;
}

int v2p_batch_reset(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    return batch_reset_locked(b, true);
}

int v2p_batch_build_from_stream(v2p_batch* b, const v2p_stream* st, int kernel, float* build_ms)
{
    if (!b || !st) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    if (st->ctx != c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if (b->hap_open || b->img.n_haplotypes()) return c->fail(V2P_ERR_STATE, "the batch already holds host-built haplotypes");
    if (kernel != 0 && kernel != 6 && kernel != 7 && kernel != 8 && kernel != 9) return c->fail(V2P_ERR_INVALID_ARG, "a resident stream builds rows images (kernel 6, 7), a patch image (8), a tile image (9) or what the routing rule picks (0)");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    b->os_kernel = 0;                                   // (not a one call: v2p_batch_oneshot_info has nothing to report)
#ifdef V2P_BENCH_VARIANTS
    if (kernel == 8) { const int prc = build_patch_image(b, st->v, build_ms, false, st->out_bytes); if (prc == V2P_OK) stream_attach(b, st); return prc; }
#else
    if (kernel == 8) return c->fail(V2P_ERR_INVALID_ARG, "patch images (kernel 8) live in libv2p_bench.so");
#endif
    int mode = rows_mode_for(st, kernel);
    if (mode == ROWS_TILES) {                           // deep Task vectors: a tile image (dense_pieces.h), not executed here
        bool fb = false;
        const int trc = build_tiles(b, st, false, &fb, build_ms);
        if (trc != V2P_OK) return trc;
        if (!fb) { stream_attach(b, st); return V2P_OK; }
        if (kernel == 9) return c->fail(V2P_ERR_UNSUPPORTED, "the stream does not fit a tile image: kernel 0 or 7 builds a dense rows image");
        mode = ROWS_DENSE;
    }
    int rc = build_rows_image(b, st->v, mode, build_ms, false);
    if (rc == V2P_ERR_UNSUPPORTED && kernel == 0 && mode == ROWS_WAVE) rc = build_rows_image(b, st->v, ROWS_DENSE, build_ms, false);     // (a row with more than 64 descriptors)
    if (rc == V2P_OK) stream_attach(b, st);
    return rc;
}

#ifdef V2P_BENCH_VARIANTS
static hipError_t patch_execute(v2p_batch* b, hipStream_t stream)
{
    v2p_ctx* c = b->ctx;
    PatchExecArgs a{reinterpret_cast<const uint64_t*>(b->d_desc.ptr()), reinterpret_cast<const uint32_t*>(b->d_patch.ptr()), reinterpret_cast<const Chunk*>(b->d_chunks.ptr()),
                    uint32_t(b->n_chunks), c->proteome.ptr(), c->proteome_len + c->headers_len, b->payload_dev, b->n_payload,
                    b->d_out.ptr(), b->out_bytes, reinterpret_cast<unsigned long long*>(b->d_status.ptr())};
    return launch_stitch_patch(a, stream, !(c->flags & V2P_FLAG_TEMPORAL));
}
#endif

// v2p_set_launch_opts: variant 16 / 17 / 18 = how a context's phased wave images are launched (A/B switches: tools/phase_ab.py):
// ONE launch for all phases / the read-ahead as kernels of its own / no read-ahead
static uint32_t touch_of(uint32_t variant) { return variant == 16u ? 4u : (variant == 17u ? 2u : (variant == 18u ? 1u : 0u)); }

// libv2p_bench.so's variant 19: the phases of a context's wave images in halves on two launch streams (launch_stitch: dual)
static void dual_of(v2p_ctx* c, StitchArgs& a)
{
    if (ctx_variant(c) != 19u) return;
    if (!c->exec_aux && hipStreamCreateWithFlags(&c->exec_aux, hipStreamNonBlocking) != hipSuccess) { c->exec_aux = nullptr; return; }
    for (hipEvent_t& e : c->ev_exec) if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return; }
    a.aux_stream = c->exec_aux; a.ev_fork = c->ev_exec[0]; a.ev_join = c->ev_exec[1]; a.opt_dual = 1u;
}

static hipError_t ensure_event(hipEvent_t& e) { return e ? hipSuccess : hipEventCreate(&e); }

// Staging buffers for an image launch_stitch() will run in the form that stages (a pure wave rows image in phases): sized by the
// launcher's own rule.  libv2p_bench.so's variant 23 (A/B): no staging -- the stitch waves read descriptors where the image has them;
// 25: dense rows images are staged as well.
static hipError_t attach_stage(v2p_batch* b, StitchArgs& a, int hint)
{
    // A padded image is staged; so is a dense one that is rich and reads a small reference (the north star's cohort): at the 28 MB phases
    // its lines need to be found in the caches, staging buys it nothing (7.38 against 7.42 ms), but staged descriptors allow 44 MB
    // phases -- a third fewer launches, tails and read-ahead hand-overs: C3 whole 7.42 -> 7.23 ms per execute; read in place the image
    // falls off a cliff there (7.3 ... 7.8 ms at 40-48 MB by run).  C4 whole (56 MB of reference) and C2 (thin: half-filled staging rows)
    // gain nothing from either and are read in place (tools/stage_probe.py, profiles/r05_staged_steady_state.json).
    const uint64_t idesc = a.img_desc ? a.img_desc : a.n_desc, ibytes = a.img_bytes ? a.img_bytes : a.out_len;
    // (... and large: what staging buys is FEWER phases; an image of three phases gains nothing from being one of two and pays the copy --
    // the routing sweep's 1.5 GB images ran 1-3 % slower staged, profiles/r05_routing_sweep.json)
    const bool small_rich = image_is_rich(idesc, ibytes) && a.src0_len <= PHASE_STAGED_SMALL_REF && 8u * idesc + 16u * uint64_t(a.n_chunks) >= 8u * PHASE_BYTES_RICH;
    if (ctx_variant(b->ctx) == 23u || !(b->pad_image || ctx_variant(b->ctx) == 25u || small_rich)) return hipSuccess;
    const uint32_t rows = stitch_stage_chunks(a, hint);
    if (rows == 0) return hipSuccess;
    const hipError_t e = b->d_stage.ensure(uint64_t(2) * rows * 64u * 8u);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }       // (no room: launched without)
    a.stage = reinterpret_cast<uint64_t*>(b->d_stage.ptr()); a.stage_chunks = rows;
    return hipSuccess;
}

// One slice's share of launch_stitch's routing (phases, store policy) follows the slice, not the table it is a range of
static hipError_t stitch_range(v2p_batch* b, uint64_t desc_bound, uint64_t chunk0, uint64_t n_chunks, uint64_t img_desc, uint64_t img_bytes, hipStream_t stream)
{
    v2p_ctx* c = b->ctx;
    if (n_chunks == 0) return hipSuccess;
    StitchArgs a{reinterpret_cast<const uint64_t*>(b->d_desc.ptr()), desc_bound, reinterpret_cast<const Chunk*>(b->d_chunks.ptr()) + chunk0,
                 uint32_t(n_chunks), c->proteome.ptr(), c->proteome_len + c->headers_len, b->payload_dev, b->n_payload,
                 b->d_out.ptr(), b->out_bytes, reinterpret_cast<unsigned long long*>(b->d_status.ptr())};
    a.opt_phase_bytes = c->launch_opts.phase_bytes; a.opt_phase_min_chunks = c->launch_opts.phase_min_chunks; a.opt_store_sc1 = c->launch_opts.store_sc1;
    a.opt_touch = touch_of(ctx_variant(c));
    dual_of(c, a);
    a.img_desc = img_desc; a.img_bytes = img_bytes;
    const int hint = int(!(c->flags & V2P_FLAG_TEMPORAL)) | b->launch_hint;
    (void)attach_stage(b, a, hint);
    return launch_stitch(a, stream, hint, 0);
}

// The sliced builder.  Global tables first (arena bytes per tile, their scan, the haplotypes' offsets: 0.2 ms), then per slice of tiles
// [T_j, T_j+1) on the BUILD stream: parse into the slice's padded array, scan of its tiles' counts (continuing the slices' running
// total), compaction into the one descriptor array, the cutter over the 640-row segments the slice completes, scan of their chunk
// counts -- one host round trip (counts, status) -- chunk table, keys, XCD / window order inside the slice; then the slice's chunks
// are stitched on the context's stream while the build stream is already parsing the next slice.  Descriptors, chunk records and
// haplotype offsets are the one-piece builder's (the cutter is deterministic per segment); only the blocks inside which the
// chunk table is dealt to the XCDs are the slice's own.  *fallback: a stream this path does not take (a tile that overflows its 256
// slots, a row with more descriptors than the kernel's chunk, more than 2^25 tiles): nothing is lost, the caller builds in one piece.
static int build_and_execute_rows(v2p_batch* b, const v2p_stream* st, int mode, uint32_t n_slices, bool* fallback)
{
    v2p_ctx* c = b->ctx;
    const DevStreamView& v = st->v;
    *fallback = false;
    const uint64_t n_tx = v.n_tx, n_h = v.n_haps;
    const uint32_t K = rows_pick_k(v, mode);
    const uint64_t n_tiles = n_tx ? (n_tx + K - 1) / K : 1;
    if (n_tiles >= (1ull << 25)) { *fallback = true; return V2P_OK; }
    const uint64_t out_bytes = st->out_bytes;
    const uint64_t n_rows = (out_bytes + ROW_BYTES - 1) / ROW_BYTES, n_segs = (n_rows + ROWS_SEG - 1) / ROWS_SEG;
    if (n_rows > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 rows in one batch");
    // (n_slices = 0: ONE slice.  Slices were built to overlap the build of slice j + 1 with the stitch of slice j; measured on C3 whole and
    // C2 -- profiles/r05_oneshot_slices.json -- every added slice costs: a slice's build takes three times as long next to a running
    // stitch (cold stream reads between its stores, the effect of DESIGN.md section 3) and the stitch slows down as well.)
    // (a padded image -- the rule is below, where its buffers are -- is built in slices of its own kind: see `ahead`)
    const uint32_t bvar = ctx_variant(c);
    const bool rich_stream = out_bytes <= uint64_t(PAD_BYTES_PER_TASK_MAX) * v.n_tasks;
    const bool pad = mode == ROWS_WAVE && n_tiles < (1ull << 24) && bvar != 22u && (rich_stream || bvar == 24u);
    uint32_t S = n_slices ? n_slices : (pad && bvar == 27u ? 3u : 1u);
    if (S > V2P_MAX_SLICES) S = V2P_MAX_SLICES;
    while (S > 1 && n_tiles / S < 64) --S;
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    // (the call's scratch and its image: when the device has no room for the one-pass form -- its padded descriptor array is up to 2 KiB per
    // tile -- nothing has been launched yet and the call builds in one piece instead, whose builder has a two-pass form without it)
#define ensure_exact(n) ensure_os(b->grow, (n))
#define OS_ALLOC(expr, what) do { hipError_t e__ = (expr); if (e__ == hipErrorOutOfMemory) { (void)hipGetLastError(); b->d_tiles.release(); b->d_cover.release(); b->d_pad.release(); b->d_order.release(); *fallback = true; return V2P_OK; } \
                                  if (e__ != hipSuccess) return c->hip_fail(e__, what); } while (0)
    // ---- memory: everything before the first kernel (a batch that is rebuilt recycles all of it) ----
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    const uint64_t o_tbytes = carve(n_tiles * 8), o_tbase = carve((n_tiles + 1) * 8), o_tcount = carve((n_tiles + 1) * 4), o_tdbase = carve((n_tiles + 2) * 8),
                   o_totals = carve(64), o_scan = carve((rows_scan_scratch_entries(n_tiles) + scan_tiles_for(n_tiles + 1)) * 8);
    OS_ALLOC(bvar == 29u ? hipErrorOutOfMemory : b->d_tiles.ensure_exact(off), "hipMalloc(tile tables)");      // (variant 29, tests: as if the device had no room)
    uint8_t* const d = b->d_tiles.ptr();
    const uint32_t chunk_pad = rows_chunk_pad_for(out_bytes, v.n_tasks + (v.fasta ? 2 * n_tx : 0), mode);
    const uint64_t c_cover = 0, c_segc = up8((n_rows + 1) * 8), c_segb = c_segc + up8((n_segs + 1) * 4), c_tiles = c_segb + up8((n_segs + 2) * 8),
                   c_cpad = c_tiles + up8(scan_tiles_for(n_segs + 1) * 8), c_end = c_cpad + up8(n_segs * chunk_pad * sizeof(Chunk));
    OS_ALLOC(b->d_cover.ensure_exact(c_end), "hipMalloc(row map)");
    uint64_t T[V2P_MAX_SLICES + 1], max_tiles = 0;
    for (uint32_t j = 0; j <= S; ++j) T[j] = n_tiles * j / S;
    for (uint32_t j = 0; j < S; ++j) if (T[j + 1] - T[j] > max_tiles) max_tiles = T[j + 1] - T[j];
    // A RICH wave image (many descriptors per result byte: C3, C4) stays PADDED (sir_pack.hpp): the parse writes every tile's descriptors to
    // its slots of d_desc and there they stay -- no compaction pass (0.74 ms and 3.6 GB of traffic of the north star's cohort's build).
    // Read in place by the stitch waves the padded array costs every execute 6 % (a chunk's descriptors in two places, lines 55 % used:
    // profiles/r05_padded_image.txt); so the launcher STAGES them (stitch_kernels.h): the read-ahead of a phase, which reads the
    // array as a stream anyway, copies the phase's descriptors into a buffer in launch order, 64 slots per chunk, and that is what
    // stitchw_kernel reads -- steady state as the dense image's, build 0.6 ms shorter.  A thin image (C2: 28 descriptors per chunk,
    // 64 MB phases) would half-fill its staging rows and double what a phase keeps in the caches: it keeps the compaction, and so do
    // dense images (chunks of up to 1 024 descriptors).  The rule looks at the stream (result bytes per Task); variants 22 / 24 force
    // the compacted / the padded form (A/B).
    if (!pad) OS_ALLOC(b->d_pad.ensure_exact(max_tiles * ROWS_PAD_SLOTS * 8), "hipMalloc(padded descriptors)");
    const uint64_t desc_cap = n_tiles * ROWS_PAD_SLOTS;              // (one-pass tiles hold at most their 256 slots)
    OS_ALLOC(b->d_desc.ensure_exact(desc_cap * 8), "hipMalloc(desc)");
    const uint64_t cap = n_segs ? n_segs * chunk_pad : 1;
    const uint64_t n_blocks_cap = order_blocks_thread_blocks(cap, XCD_ORDER_MAX_BLOCKS);
    const uint64_t n_sub_cap = uint64_t(XCD_SUB) * n_blocks_cap;
    const uint64_t s_tmp = 0, s_bucket = s_tmp + up8(cap * 16), s_hist = s_bucket + up8(cap),
                   s_sub = s_hist + up8((n_blocks_cap + 1) * 8 * 4), s_tmp2 = s_sub + up8(cap), s_bucket2 = s_tmp2 + up8(cap * 16),
                   s_subhist = s_bucket2 + up8(cap), s_substart = s_subhist + up8(n_sub_cap * 4), s_subtiles = s_substart + up8((n_sub_cap + 1) * 8),
                   s_tot = s_subtiles + up8(scan_tiles_for(n_sub_cap) * 8), s_end = s_tot + up8(uint64_t(XCD_ORDER_MAX_BLOCKS) * 8 * 4);
    OS_ALLOC(b->d_order.ensure_exact(s_end), "hipMalloc(build scratch)");
    OS_ALLOC(b->d_chunks.ensure_exact(cap * sizeof(Chunk)), "hipMalloc(chunks)");
    HIP_TRY(c, b->d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_hap.ensure((n_h + 1) * 8), "hipMalloc(hap_begin)");
    HIP_TRY(c, b->d_digest.ensure((n_h ? n_h : 1) * 8), "hipMalloc(digest)");
    if (!c->build_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking), "hipStreamCreate(build)");
    if (!c->aux_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking), "hipStreamCreate(aux)");
    for (uint32_t k = 0; k < 2 + 2 * S; ++k) HIP_TRY(c, ensure_event(b->ev_os[k]), "hipEventCreate");
    for (hipEvent_t& e : b->ev_aux) HIP_TRY(c, ensure_event(e), "hipEventCreate");
    for (uint32_t k = 0; k < S; ++k) HIP_TRY(c, ensure_event(b->ev_par[k]), "hipEventCreate");
    HIP_TRY(c, b->h_sum.ensure(64), "hipHostMalloc(summary)");
    hipStream_t A = c->stream, B = c->build_stream, X = c->aux_stream;
    int rc = init_status(c, b->d_status);             // (on A)
    if (rc) return rc;
    uint8_t* const sc = b->d_order.ptr();
    RowsArgs a;
    rows_args_of(v, c, K, n_tiles, a);
    a.tile_bytes = reinterpret_cast<uint64_t*>(d + o_tbytes); a.tile_res_base = reinterpret_cast<uint64_t*>(d + o_tbase);
    // (A/B switches of the builder, v2p_set_launch_opts: variant 20 = tile = workgroup index in the parse, 21 = tile tables made inside the
    // call: round 5's first form; 22 / 24 = the compacted / the padded form of a wave image whatever the rule says)
    a.xcd_tiles = bvar == 20u ? 0u : 1u;
    const bool cached = bvar != 21u && st->tile_K == K && st->n_tiles == n_tiles && st->tile_res_base != nullptr && (S == 1 || st->h_tile_res_base.size() == n_tiles + 1);
    if (cached) a.tile_res_base = const_cast<uint64_t*>(st->tile_res_base);      // (read only from here on)
    a.tile_count = reinterpret_cast<uint32_t*>(d + o_tcount); a.tile_desc_base = reinterpret_cast<uint64_t*>(d + o_tdbase);
    a.totals = reinterpret_cast<uint64_t*>(d + o_totals);
    a.status = reinterpret_cast<unsigned long long*>(b->d_status.ptr());
    uint64_t* const scan_scratch = reinterpret_cast<uint64_t*>(d + o_scan);
    a.out_bytes = out_bytes; a.n_rows = n_rows; a.n_segs = n_segs;
    a.cover = reinterpret_cast<uint64_t*>(b->d_cover.ptr() + c_cover);
    a.seg_count = reinterpret_cast<uint32_t*>(b->d_cover.ptr() + c_segc);
    a.seg_base = reinterpret_cast<const uint64_t*>(b->d_cover.ptr() + c_segb);
    a.chunks_pad = reinterpret_cast<Chunk*>(b->d_cover.ptr() + c_cpad); a.chunk_pad = chunk_pad;
    a.hap_out_begin = reinterpret_cast<uint64_t*>(b->d_hap.ptr());
    a.desc_pad = reinterpret_cast<uint64_t*>(b->d_pad.ptr());
    a.desc = reinterpret_cast<uint64_t*>(b->d_desc.ptr()); a.desc_cap = desc_cap;
    a.pad_chunks = pad ? 1u : 0u;
    Chunk* const chunks_tmp = reinterpret_cast<Chunk*>(sc + s_tmp);
    uint8_t* const bucket = sc + s_bucket; uint8_t* const sub = sc + s_sub;
    // the batch describes the image from here on (a failure below resets it)
    b->n_payload = v.n_alt; b->payload_dev = v.alt; b->out_bytes = out_bytes; b->n_haps = n_h;
    b->launch_hint = (mode == ROWS_DENSE ? 2 : 4) | 8 | 16 | 32 | (1 << 6) | (1 << 8);
    b->pad_image = pad; b->desc_slots = pad ? desc_cap : 0; b->pad_n_tiles = n_tiles; b->pad_K = K; b->pad_tdbase = a.tile_desc_base;
    auto fail_reset = [&](int code) {
        (void)hipStreamSynchronize(A); (void)hipStreamSynchronize(B); (void)hipStreamSynchronize(X);
        (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, A);
        b->pad_image = false; b->desc_slots = 0; b->pad_tdbase = nullptr;
        b->img.hap_out_begin.assign(1, 0); b->n_desc = b->n_chunks = b->n_payload = b->out_bytes = b->n_haps = 0; b->payload_dev = nullptr; b->n_slices = 0;
        return code;
    };
#define OS_TRY(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail_reset(c->hip_fail(e__, what)); } while (0)
    // ---- global tables ----
    OS_TRY(hipEventRecord(b->ev_os[0], A), "hipEventRecord");
    OS_TRY(hipStreamWaitEvent(B, b->ev_os[0], 0), "hipStreamWaitEvent");
    OS_TRY(hipMemsetAsync(d + o_totals, 0, 64, B), "hipMemset(totals)");
    uint64_t R[V2P_MAX_SLICES + 1];                                 // arena offset of every slice's first tile
    if (cached) {
        // res_counter per tile and per haplotype came with the stream (v2p_stream_upload): no kernel, no host round trip before the parse
        OS_TRY(hipMemcpyAsync(b->d_hap.ptr(), st->d_hap_out_begin, (n_h + 1) * 8, hipMemcpyDeviceToDevice, B), "D2D(hap_begin)");
        R[0] = 0; R[S] = out_bytes;
        for (uint32_t j = 1; j < S; ++j) R[j] = st->h_tile_res_base[T[j]];
    } else {
        OS_TRY(launch_rows_tile_bytes(a, scan_scratch, B), "launch(tile bytes)");
        OS_TRY(launch_rows_hap_begin(a, B), "launch(hap_begin)");
        for (uint32_t j = 0; j <= S; ++j) OS_TRY(hipMemcpyAsync(&R[j], d + o_tbase + T[j] * 8, 8, hipMemcpyDeviceToHost, B), "D2H(slice offsets)");
        OS_TRY(hipStreamSynchronize(B), "hipStreamSynchronize");
        if (R[S] != out_bytes) return fail_reset(c->fail(V2P_ERR_STATE, "the resident stream's tables changed since its upload"));
    }
    uint64_t SG[V2P_MAX_SLICES + 1];                                // segments [SG_j, SG_j+1) are complete -- their rows' cover entries and the entry of the
    SG[0] = 0;                                                      // row behind them written -- once slice j is parsed
    for (uint32_t j = 1; j < S; ++j) { SG[j] = R[j] ? (R[j] - 1) / (uint64_t(ROWS_SEG) * ROW_BYTES) : 0; if (SG[j] < SG[j - 1]) SG[j] = SG[j - 1]; }
    SG[S] = n_segs;
    for (uint32_t j = 1; j < S; ++j) if (SG[j] > n_segs) SG[j] = n_segs;
    const bool reorder = !(c->flags & V2P_FLAG_RESULT_ORDER) && c->proteome_len != 0;
    uint64_t desc0 = 0, chunk0 = 0;
    b->os_build_ms = 0.f;
    // AHEAD (a padded image in more than one slice: n_slices, or three with A/B variant 27): every slice's parse -- the build's one heavy
    // kernel -- is launched at once on a stream of its own, the count scans chained on the device, and what follows a parse (cutter,
    // chunk table, the host's look at the counts, keys, XCD order: 0.75 ms of serial walks and small sorts) runs for slice j on the
    // build stream WHILE slice j + 1 is parsed.  The descriptors need no second pass (padded), so nothing of a slice waits for the
    // slices behind it; the stitch starts when the last slice's chunk table stands.  MEASURED AND NOT THE DEFAULT: C3 whole's build
    // 3.96 -> 4.34 ms, C4 whole's 2.75 -> 3.07 with three slices -- the small kernels take wave slots and issue cycles from the parse,
    // which is balanced on all of its units, and give back less than they take.  Overlap does not pay on this chip, again (section 4).
    const bool ahead = pad && S > 1;
    b->os_ahead = ahead;
    if (ahead) {
        OS_TRY(hipStreamWaitEvent(X, b->ev_os[0], 0), "hipStreamWaitEvent");
        OS_TRY(hipEventRecord(b->ev_os[2], X), "hipEventRecord");
        for (uint32_t j = 0; j < S; ++j) {
            a.tile0 = T[j]; a.tile1 = T[j + 1];
            const uint64_t nt = T[j + 1] - T[j];
            a.desc_pad = a.desc + T[j] * ROWS_PAD_SLOTS;
            if (nt) OS_TRY(launch_rows_parse(a, mode, v.fasta, 0, X), "launch(parse)");
            if (j == 0) OS_TRY(launch_scan_u32_from(a.tile_count, nt, a.tile_desc_base, scan_scratch + rows_scan_scratch_entries(n_tiles), 0, X), "launch(scan)");
            else OS_TRY(launch_scan_u32_chained(a.tile_count + T[j], nt, a.tile_desc_base + T[j], scan_scratch + rows_scan_scratch_entries(n_tiles), X), "launch(scan)");
            OS_TRY(hipEventRecord(b->ev_par[j], X), "hipEventRecord");
        }
    }
    for (uint32_t j = 0; j < S; ++j) {
        a.tile0 = T[j]; a.tile1 = T[j + 1]; a.seg0 = SG[j]; a.seg1 = SG[j + 1];
        const uint64_t nt = T[j + 1] - T[j], ns = SG[j + 1] - SG[j];
        if (ahead) OS_TRY(hipStreamWaitEvent(B, b->ev_par[j], 0), "hipStreamWaitEvent");
        if (!ahead || j > 0) OS_TRY(hipEventRecord(b->ev_os[2 + 2 * j], B), "hipEventRecord");     // (ahead: slice 0's interval starts with its parse, on X)
        if (nt && !ahead) {
            if (pad) a.desc_pad = a.desc + T[j] * ROWS_PAD_SLOTS;               // (the slice's tiles' own slots of the one array)
            OS_TRY(launch_rows_parse(a, mode, v.fasta, 0, B), "launch(parse)");
            OS_TRY(launch_scan_u32_from(a.tile_count + T[j], nt, a.tile_desc_base + T[j], scan_scratch + rows_scan_scratch_entries(n_tiles), desc0, B), "launch(scan)");
            if (!pad) {
                // the compaction (a copy at the memory's rate) on a stream of its own, next to the cutter (one wave per 640 rows: latency), the
                // scan of its counts and the host's round trip for them: the cutter reads the row map alone, only the keys need the descriptors
                OS_TRY(hipEventRecord(b->ev_aux[0], B), "hipEventRecord");
                OS_TRY(hipStreamWaitEvent(X, b->ev_aux[0], 0), "hipStreamWaitEvent");
                OS_TRY(launch_rows_compact(a, X), "launch(compact)");
                OS_TRY(hipEventRecord(b->ev_aux[1], X), "hipEventRecord");
            }
        }
        if (ns) OS_TRY(launch_rows_cut(a, mode, 2, B), "launch(cut)");
        OS_TRY(launch_scan_u32_from(a.seg_count + SG[j], ns, const_cast<uint64_t*>(a.seg_base) + SG[j], reinterpret_cast<uint64_t*>(b->d_cover.ptr() + c_tiles), chunk0, B), "launch(scan)");
        // (the chunk table in arena order before the host's one look at the counts: its grid is the segments', and for a padded image it
        // is also where a chunk that does not fit the slot-addressed form is found -- totals[3])
        a.chunks_tmp = chunks_tmp; a.chunk_cap = cap;
        a.bucket = pad && reorder ? bucket : nullptr; a.sub = pad && reorder ? sub : nullptr;      // (a padded image: the chunks' keys in the same pass)
        if (ns) OS_TRY(launch_rows_chunk_compact(a, B), "launch(chunk table)");
        // (the host's one look per slice: the GPU idles while it waits, so what it reads is gathered on the device and comes back as ONE
        // 64-byte copy into pinned memory -- four pageable copies took 90 us of C3 whole's call, this takes 35)
        OS_TRY(launch_rows_summary(nt ? reinterpret_cast<const uint64_t*>(d + o_tdbase) + T[j + 1] : nullptr, reinterpret_cast<const uint64_t*>(b->d_cover.ptr() + c_segb) + SG[j + 1],
                                   reinterpret_cast<const unsigned long long*>(b->d_status.ptr()), a.totals, B), "launch(summary)");
        uint64_t* const hs = reinterpret_cast<uint64_t*>(b->h_sum.p);
        OS_TRY(hipMemcpyAsync(hs, d + o_totals, 64, hipMemcpyDeviceToHost, B), "D2H(totals)");
        OS_TRY(hipStreamSynchronize(B), "hipStreamSynchronize");
        const uint64_t totals[4] = {hs[0], hs[1], hs[2], hs[3]};
        const uint64_t desc_end = nt ? hs[4] : desc0, chunk_end = hs[5];
        const unsigned long long stw = hs[6];
        if (stw != STATUS_CLEAN || totals[3] != 0) {
            const uint32_t reason = stw == STATUS_CLEAN ? 0u : uint32_t(stw & 0xFFu);
            if (reason == 0u || reason == STATUS_ROWS_STAGE || reason == STATUS_ROWS_TOO_MANY) { *fallback = true; return fail_reset(V2P_OK); }
            if (reason == STATUS_ROWS_SPAN) return fail_reset(c->fail(V2P_ERR_UNSUPPORTED, "64 consecutive transcripts with more than 2 GiB of result", int64_t(stw >> 8)));
            const int code = reason_to_err(reason);                  // what the reference would panic on (update_task, Task::execute)
            return fail_reset(c->fail(code, std::string("device: ") + err_name(code) + " at descriptor " + std::to_string(stw >> 8), int64_t(stw >> 8)));
        }
        const uint64_t nd = desc_end - desc0, nc = chunk_end - chunk0;
        if (chunk_end > 0xFFFFFFFFull) return fail_reset(c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch"));
        // (the keys read the chunks' first descriptors.  Reading them from the padded array instead, so that the keys and the sorts run
        // beside the compaction as well, was measured: nothing -- the copy runs at the memory's rate and what runs beside it waits for it)
        if (nt && !pad) OS_TRY(hipStreamWaitEvent(B, b->ev_aux[1], 0), "hipStreamWaitEvent");      // (the descriptors are in place)
        Chunk* const out_chunks = reinterpret_cast<Chunk*>(b->d_chunks.ptr()) + chunk0;
        if (reorder && nc >= 16 && desc_end != 0) {
            RowsArgs ak = a;
            ak.chunks_tmp = chunks_tmp + chunk0; ak.bucket = bucket + chunk0; ak.sub = sub + chunk0;
            ak.hap_major = xcd_order_window_major(ns * uint64_t(ROWS_SEG) * ROW_BYTES, nd) ? 0u : 1u;      // (a padded image is rich: window-major, keys made in the table pass)
            if (!pad) OS_TRY(launch_rows_keys(ak, nc, desc_end, B), "launch(keys)");
            const uint32_t nb = xcd_order_blocks(ns * uint64_t(ROWS_SEG) * ROW_BYTES, c->proteome_len, nc, XCD_ORDER_MAX_BLOCKS, nd);
            OS_TRY(launch_order_blocks(ak.chunks_tmp, ak.bucket, ak.sub, nc, nb, reinterpret_cast<uint32_t*>(sc + s_subhist),
                                       reinterpret_cast<uint64_t*>(sc + s_substart), reinterpret_cast<uint64_t*>(sc + s_subtiles),
                                       reinterpret_cast<Chunk*>(sc + s_tmp2), sc + s_bucket2, reinterpret_cast<uint32_t*>(sc + s_hist),
                                       reinterpret_cast<uint32_t*>(sc + s_tot), out_chunks, B), "launch(order)");
        } else if (nc) OS_TRY(hipMemcpyAsync(out_chunks, chunks_tmp + chunk0, nc * sizeof(Chunk), hipMemcpyDeviceToDevice, B), "D2D(chunks)");
        OS_TRY(hipEventRecord(b->ev_os[3 + 2 * j], B), "hipEventRecord");
        b->slice_chunk0[j] = chunk0; b->slice_desc[j] = nd; b->slice_bytes[j] = ns * uint64_t(ROWS_SEG) * ROW_BYTES;
        if (!ahead) {
            // ---- the slice is built: stitch it on the context's stream while the build stream goes on ----
            OS_TRY(hipStreamWaitEvent(A, b->ev_os[3 + 2 * j], 0), "hipStreamWaitEvent");
            OS_TRY(stitch_range(b, pad ? desc_cap : desc_end, chunk0, nc, nd, b->slice_bytes[j], A), "launch(stitch)");
        }
        desc0 = desc_end; chunk0 = chunk_end;
    }
    if (ahead) {
        // ---- every slice is built: the stitch, slice after slice (each table is dealt to the XCDs inside its own slice) ----
        OS_TRY(hipStreamWaitEvent(A, b->ev_os[3 + 2 * (S - 1)], 0), "hipStreamWaitEvent");
        b->slice_chunk0[S] = chunk0;
        for (uint32_t j = 0; j < S; ++j)
            OS_TRY(stitch_range(b, desc_cap, b->slice_chunk0[j], b->slice_chunk0[j + 1] - b->slice_chunk0[j], b->slice_desc[j], b->slice_bytes[j], A), "launch(stitch)");
    }
#undef OS_TRY
#undef OS_ALLOC
#undef ensure_exact
    b->slice_chunk0[S] = chunk0;
    HIP_TRY(c, hipEventRecord(b->ev_os[1], A), "hipEventRecord");
    b->n_desc = desc0; b->n_chunks = chunk0; b->n_slices = S;
    b->img.hap_out_begin = st->hap_out_begin;
    b->uses_proteome = true;
    b->finalized = true;
    return V2P_OK;
}

static hipError_t tiles_execute(v2p_batch* b, hipStream_t stream)
{
    v2p_ctx* c = b->ctx;
    TileExecArgs a{reinterpret_cast<const uint64_t*>(b->d_desc.ptr()), b->tile_slots, b->tiles_count, b->tiles_res_base, b->tiles_n,
                   c->proteome.ptr(), b->payload_dev, b->d_out.ptr(), b->out_bytes, reinterpret_cast<const unsigned long long*>(b->d_status.ptr())};
    return launch_stitch_tiles(a, stream, !(c->flags & V2P_FLAG_TEMPORAL));
}

// A TILE image (dense_pieces.h), built -- and, in the one call, executed -- on the context's stream: [res_counter per tile and per haplotype
// unless the resident stream carries them] the parse (pieces into the tiles' slots), the scan of the tiles' counts (their total: the
// batch's counts), ONE look of the host at the status word, the executor.  *fallback: a stream the form does not take after all (a tile
// whose result or whose pieces do not fit: the sample that sized the tiles missed it; no room for the piece slots) -- nothing is
// lost, the caller builds a dense rows image.  c->mu held.
static int build_tiles(v2p_batch* b, const v2p_stream* st, bool execute, bool* fallback, float* build_ms, uint32_t n_slices)
{
    v2p_ctx* c = b->ctx;
    const DevStreamView& v = st->v;
    *fallback = false;
    uint32_t slots = 0;
    const uint32_t K = tiles_possible(st) ? rows_pick_k_tiles(v, &slots) : 0u;
    if (K == 0u) { *fallback = true; return V2P_OK; }
    const uint64_t n_tx = v.n_tx, n_h = v.n_haps, out_bytes = st->out_bytes;
    const uint64_t n_tiles = (n_tx + K - 1) / K;
    if (n_tiles >= (1ull << 31)) { *fallback = true; return V2P_OK; }
    auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) { const uint64_t o = off; off += up8(bytes); return o; };
    const uint64_t o_tbytes = carve(n_tiles * 8), o_tbase = carve((n_tiles + 1) * 8), o_tcount = carve((n_tiles + 1) * 4), o_tdbase = carve((n_tiles + 2) * 8),
                   o_totals = carve(64), o_scan = carve((rows_scan_scratch_entries(n_tiles) + scan_tiles_for(n_tiles + 1)) * 8);
    auto oom = [&](hipError_t e) { if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return true; } return false; };
    {
        hipError_t e = b->d_tiles.ensure_os(b->grow, off);
        if (e == hipSuccess) e = b->d_desc.ensure_os(b->grow, n_tiles * slots * 8);
        if (oom(e)) { *fallback = true; return V2P_OK; }
        HIP_TRY(c, e, "hipMalloc(tile image)");
    }
    HIP_TRY(c, b->d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_hap.ensure((n_h + 1) * 8), "hipMalloc(hap_begin)");
    HIP_TRY(c, b->d_digest.ensure((n_h ? n_h : 1) * 8), "hipMalloc(digest)");
    for (uint32_t k = 0; k < 4; ++k) HIP_TRY(c, ensure_event(b->ev_os[k]), "hipEventCreate");
    HIP_TRY(c, b->h_sum.ensure(64), "hipHostMalloc(summary)");
    hipStream_t A = c->stream;
    int rc = init_status(c, b->d_status);
    if (rc) return rc;
    uint8_t* const d = b->d_tiles.ptr();
    RowsArgs a;
    rows_args_of(v, c, K, n_tiles, a);
    a.tile_slots = slots; a.tile_span_max = TILE_SPAN_MAX;
    a.tile_bytes = reinterpret_cast<uint64_t*>(d + o_tbytes); a.tile_res_base = reinterpret_cast<uint64_t*>(d + o_tbase);
    a.tile_count = reinterpret_cast<uint32_t*>(d + o_tcount); a.tile_desc_base = reinterpret_cast<uint64_t*>(d + o_tdbase);
    a.totals = reinterpret_cast<uint64_t*>(d + o_totals);
    a.status = reinterpret_cast<unsigned long long*>(b->d_status.ptr());
    a.desc_pad = reinterpret_cast<uint64_t*>(b->d_desc.ptr());
    a.hap_out_begin = reinterpret_cast<uint64_t*>(b->d_hap.ptr());
    a.out_bytes = out_bytes;
    uint64_t* const scan_scratch = reinterpret_cast<uint64_t*>(d + o_scan);
    const bool cached = ctx_variant(c) != 21u && st->tile_K == K && st->n_tiles == n_tiles && st->tile_res_base != nullptr && st->d_hap_out_begin != nullptr;
    HIP_TRY(c, hipEventRecord(b->ev_os[0], A), "hipEventRecord");
    if (cached) {
        a.tile_res_base = const_cast<uint64_t*>(st->tile_res_base);
        HIP_TRY(c, hipMemcpyAsync(b->d_hap.ptr(), st->d_hap_out_begin, (n_h + 1) * 8, hipMemcpyDeviceToDevice, A), "D2D(hap_begin)");
    } else {
        HIP_TRY(c, launch_rows_tile_bytes(a, scan_scratch, A), "launch(tile bytes)");
        HIP_TRY(c, launch_rows_hap_begin(a, A), "launch(hap_begin)");
    }
    HIP_TRY(c, hipEventRecord(b->ev_os[2], A), "hipEventRecord");
    bool executed_in_slices = false;
#ifdef V2P_BENCH_VARIANTS
    // (A/B, libv2p_bench.so: the tiles in S slices -- a tile is the executor's work item as it leaves the parse, so slice j is executed on a
    // second stream while slice j + 1 is parsed; the executor reads the status word itself and the host's look comes behind everything.
    // Measured (profiles/r06_tile_slices.txt): C5 whole 12.1 -> 11.7 / 11.4 / 11.2 ms with 4 / 8 / 16 slices, C5 fifth 2.58 -> 2.51 with 4,
    // 2.74 with 16; the same with the executor's stream at the highest priority.  The parse fills every wave slot; not the product's form.)
    if (execute && n_slices > 1 && n_tiles >= n_slices) {
        const uint32_t S = n_slices < V2P_MAX_SLICES ? n_slices : V2P_MAX_SLICES;
        if (!c->build_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking), "hipStreamCreate(build)");
        for (uint32_t k = 0; k < S; ++k) HIP_TRY(c, ensure_event(b->ev_par[k]), "hipEventCreate");
        HIP_TRY(c, ensure_event(b->ev_aux[0]), "hipEventCreate");
        hipStream_t B = c->build_stream;
        uint64_t* const pad0 = a.desc_pad;
        for (uint32_t j = 0; j < S; ++j) {
            RowsArgs aj = a;
            aj.tile0 = n_tiles * j / S; aj.tile1 = n_tiles * (j + 1) / S;
            aj.desc_pad = pad0 + aj.tile0 * slots;
            HIP_TRY(c, launch_rows_parse(aj, ROWS_TILES, v.fasta, 0, A), "launch(parse: pieces)");
            HIP_TRY(c, hipEventRecord(b->ev_par[j], A), "hipEventRecord");
            HIP_TRY(c, hipStreamWaitEvent(B, b->ev_par[j], 0), "hipStreamWaitEvent");
            TileExecArgs x{reinterpret_cast<const uint64_t*>(b->d_desc.ptr()), slots, a.tile_count, a.tile_res_base, aj.tile1 - aj.tile0,
                           c->proteome.ptr(), v.alt, b->d_out.ptr(), out_bytes, reinterpret_cast<const unsigned long long*>(b->d_status.ptr()), aj.tile0};
            HIP_TRY(c, launch_stitch_tiles(x, B, !(c->flags & V2P_FLAG_TEMPORAL)), "launch(stitch: tile image)");
        }
        HIP_TRY(c, hipEventRecord(b->ev_aux[0], B), "hipEventRecord");
        HIP_TRY(c, hipStreamWaitEvent(A, b->ev_aux[0], 0), "hipStreamWaitEvent");
        executed_in_slices = true;
    } else
#else
    (void)n_slices;
#endif
    HIP_TRY(c, launch_rows_parse(a, ROWS_TILES, v.fasta, 0, A), "launch(parse: pieces)");
    HIP_TRY(c, launch_scan_u32(a.tile_count, n_tiles, a.tile_desc_base, scan_scratch + rows_scan_scratch_entries(n_tiles), A), "launch(scan)");
    HIP_TRY(c, launch_rows_summary(a.tile_desc_base + n_tiles, a.tile_res_base + n_tiles, a.status, a.totals, A), "launch(summary)");
    uint64_t* const hs = reinterpret_cast<uint64_t*>(b->h_sum.p);
    HIP_TRY(c, hipMemcpyAsync(hs, d + o_totals, 64, hipMemcpyDeviceToHost, A), "D2H(totals)");
    HIP_TRY(c, hipEventRecord(b->ev_os[3], A), "hipEventRecord");
    HIP_TRY(c, hipStreamSynchronize(A), "hipStreamSynchronize");
    const unsigned long long stw = hs[6];
    if (stw != STATUS_CLEAN) {
        (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, 8, A);
        const uint32_t reason = uint32_t(stw & 0xFFu);
        if (reason == STATUS_ROWS_STAGE || reason == STATUS_ROWS_TOO_MANY) { *fallback = true; return V2P_OK; }
        if (reason == STATUS_ROWS_SPAN) return c->fail(V2P_ERR_UNSUPPORTED, "64 consecutive transcripts with more than 2 GiB of result", int64_t(stw >> 8));
        const int code = reason_to_err(reason);                  // what the reference would panic on (update_task, Task::execute)
        return c->fail(code, std::string("device: ") + err_name(code) + " at descriptor " + std::to_string(stw >> 8), int64_t(stw >> 8));
    }
    if (hs[5] != out_bytes) return c->fail(V2P_ERR_STATE, "the resident stream's tables changed since its upload");
    b->is_tiles = true; b->tile_slots = slots; b->tiles_n = n_tiles; b->tiles_count = a.tile_count; b->tiles_res_base = a.tile_res_base;
    b->n_desc = hs[4]; b->n_chunks = n_tiles; b->n_payload = v.n_alt; b->payload_dev = v.alt; b->out_bytes = out_bytes; b->n_haps = n_h;
    b->pieces_src0_len = c->proteome_len + c->headers_len; b->pieces_src1_len = v.n_alt;
    b->launch_hint = 0; b->pad_image = false; b->desc_slots = 0;
    b->img.hap_out_begin = st->hap_out_begin;
    b->uses_proteome = true; b->finalized = true;
    b->n_slices = 1; b->slice_chunk0[0] = 0; b->slice_chunk0[1] = n_tiles; b->os_ahead = false;
    if (execute && !executed_in_slices) HIP_TRY(c, tiles_execute(b, A), "launch(stitch: tile image)");
    HIP_TRY(c, hipEventRecord(b->ev_os[1], A), "hipEventRecord");
    if (build_ms) { *build_ms = 0.f; (void)hipEventElapsedTime(build_ms, b->ev_os[0], b->ev_os[3]); }
    return V2P_OK;
}

// (c->mu held by the caller: v2p_batch_build_and_execute, and the streamed pipeline's runner)
static int build_and_execute_locked(v2p_batch* b, const v2p_stream* st, int kernel, uint32_t n_slices)
{
    v2p_ctx* c = b->ctx;
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized (v2p_batch_reset recycles it)");
    if (b->hap_open || b->img.n_haplotypes()) return c->fail(V2P_ERR_STATE, "the batch already holds host-built haplotypes");
    if (kernel != 0 && kernel != 6 && kernel != 7 && kernel != 8 && kernel != 9) return c->fail(V2P_ERR_INVALID_ARG, "a resident stream builds rows images (kernel 6, 7), a patch image (8), a tile image (9) or what the routing rule picks (0)");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    const auto t0 = std::chrono::steady_clock::now();
#ifndef V2P_BENCH_VARIANTS
    if (kernel == 8) return c->fail(V2P_ERR_INVALID_ARG, "patch images (kernel 8) live in libv2p_bench.so");
    if (n_slices > 1) return c->fail(V2P_ERR_INVALID_ARG, "n_slices: 0 or 1 (building in slices was measured slower on every cohort and lives in libv2p_bench.so)");
#else
    if (kernel == 8) {
        // a PATCH image: one build kernel behind the positions' scan, then one stitch kernel, on the context's stream
        for (uint32_t k = 0; k < 2; ++k) HIP_TRY(c, ensure_event(b->ev_os[k]), "hipEventCreate");
        HIP_TRY(c, hipEventRecord(b->ev_os[0], c->stream), "hipEventRecord");
        float ms = 0.f;
        const int prc = build_patch_image(b, st->v, &ms, false, st->out_bytes);
        if (prc != V2P_OK) return prc;
        const hipError_t pe = patch_execute(b, c->stream);
        if (pe != hipSuccess) return c->hip_fail(pe, "launch(stitch: patch image)");
        HIP_TRY(c, hipEventRecord(b->ev_os[1], c->stream), "hipEventRecord");
        b->os_build_ms = ms; b->os_kernel = 8; b->n_slices = 0;
        b->os_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stream_attach(b, st);
        return V2P_OK;
    }
#endif
    int mode = rows_mode_for(st, kernel);
    bool fallback = false;
    if (mode == ROWS_TILES) {
        // deep Task vectors: pieces straight from the parse, the tiles executed as they stand (dense_pieces.h)
        const int trc = build_tiles(b, st, true, &fallback, nullptr, n_slices);
        if (trc != V2P_OK) return trc;
        if (!fallback) {
            stream_attach(b, st);
            b->executed = true; b->os_kernel = 9; b->os_build_ms = 0.f;
            b->os_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return V2P_OK;
        }
        if (kernel == 9) return c->fail(V2P_ERR_UNSUPPORTED, "the stream does not fit a tile image (a tile of transcripts with more than 16 368 result bytes or 2 048 pieces, sources beyond 2 GiB): kernel 0 or 7 builds a dense rows image");
        mode = ROWS_DENSE; fallback = false;
    }
    int rc = build_and_execute_rows(b, st, mode, n_slices, &fallback);
    if (rc == V2P_OK && fallback) {
        // streams the sliced builder does not take: the one-piece builder (its two-pass form, or a dense image), then one execute
        for (uint32_t k = 0; k < 2; ++k) HIP_TRY(c, ensure_event(b->ev_os[k]), "hipEventCreate");
        HIP_TRY(c, hipEventRecord(b->ev_os[0], c->stream), "hipEventRecord");
        float ms = 0.f;
        rc = build_rows_image(b, st->v, mode, &ms, false);
        if (rc == V2P_ERR_UNSUPPORTED && kernel == 0 && mode == ROWS_WAVE) { mode = ROWS_DENSE; rc = build_rows_image(b, st->v, mode, &ms, false); }
        if (rc == V2P_OK) {
            b->os_build_ms = ms;
            const hipError_t e = stitch_range(b, b->n_desc, 0, b->n_chunks, 0, 0, c->stream);
            if (e != hipSuccess) return c->hip_fail(e, "launch(stitch)");
            HIP_TRY(c, hipEventRecord(b->ev_os[1], c->stream), "hipEventRecord");
        }
    }
    if (rc == V2P_OK) {
        stream_attach(b, st);
        b->executed = true;
        b->os_kernel = mode == ROWS_DENSE ? 7 : 6;
        b->os_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

int v2p_batch_build_and_execute(v2p_batch* b, const v2p_stream* st, int kernel, uint32_t n_slices)
{
    if (!b || !st) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    if (st->ctx != c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return build_and_execute_locked(b, st, kernel, n_slices);
}

int v2p_batch_oneshot_info(v2p_batch* b, v2p_oneshot_info* info)
{
    if (!b || !info) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized || !b->ev_os[0] || !b->ev_os[1] || b->os_kernel == 0) return c->fail(V2P_ERR_STATE, "no v2p_batch_build_and_execute on this batch");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    HIP_TRY(c, hipEventSynchronize(b->ev_os[1]), "hipEventSynchronize");
    memset(info, 0, sizeof *info);
    info->kernel = b->os_kernel; info->n_slices = b->n_slices; info->call_wall_ms = b->os_wall_ms;
    HIP_TRY(c, hipEventElapsedTime(&info->total_ms, b->ev_os[0], b->ev_os[1]), "hipEventElapsedTime");
    float sum = 0.f;
    for (uint32_t j = 0; j < b->n_slices && j < V2P_MAX_SLICES; ++j) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, b->ev_os[2 + 2 * j], b->ev_os[3 + 2 * j]), "hipEventElapsedTime");
        info->slice_build_ms[j] = ms; sum += ms;
    }
    info->build_ms = b->n_slices ? sum : b->os_build_ms;
    if (b->n_slices && b->os_ahead) HIP_TRY(c, hipEventElapsedTime(&info->build_ms, b->ev_os[2], b->ev_os[3 + 2 * (b->n_slices - 1)]), "hipEventElapsedTime");   // (the slices overlap: first parse .. last chunk table)
    if (b->n_slices) HIP_TRY(c, hipEventElapsedTime(&info->tables_ms, b->ev_os[0], b->ev_os[2]), "hipEventElapsedTime");
    return V2P_OK;
}

#ifdef V2P_BENCH_VARIANTS
int v2p_batch_download_patch_image(v2p_batch* b, uint64_t* seg, uint32_t* patch, v2p_chunk* chunks, uint64_t* n_segments, uint64_t* n_patches)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized || !b->is_patch) return c->fail(V2P_ERR_STATE, "not a finalized patch image");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (n_segments) *n_segments = b->patch_segs;
    if (n_patches) *n_patches = b->patch_patches;
    if (seg && b->n_chunks) HIP_TRY(c, hipMemcpyAsync(seg, b->d_desc.ptr(), b->n_chunks * PATCH_SEG_CAP * 8, hipMemcpyDeviceToHost, c->stream), "D2H(segments)");
    if (patch && b->n_chunks) HIP_TRY(c, hipMemcpyAsync(patch, b->d_patch.ptr(), b->n_chunks * PATCH_PATCH_CAP * 4, hipMemcpyDeviceToHost, c->stream), "D2H(patches)");
    if (chunks && b->n_chunks) HIP_TRY(c, hipMemcpyAsync(chunks, b->d_chunks.ptr(), b->n_chunks * sizeof(Chunk), hipMemcpyDeviceToHost, c->stream), "D2H(chunks)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return V2P_OK;
}
#endif

// A padded image becomes the dense one (on the context's stream, for good): the compaction the one call skipped, into the batch's spare
// descriptor buffer, and the chunk records translated in place.  Called by whoever needs the dense form -- a download, and the first
// RE-execute of the batch: read in place the padded array costs a steady-state execute 6 %, staged 1-2 % (the read-ahead also writes),
// so a batch that is executed again pays the 0.74 ms once and runs the dense image from then on; a cohort that is executed once -- the
// one call -- never pays it.
static int densify(v2p_batch* b)
{
    v2p_ctx* c = b->ctx;
    if (!b->pad_image) return V2P_OK;
    {   // (no room for the dense copy: the image stays padded -- it executes in that form too, its descriptors staged)
        const hipError_t me = b->d_pad.ensure((b->n_desc ? b->n_desc : 1) * 8);
        if (me == hipErrorOutOfMemory) { (void)hipGetLastError(); return V2P_OK; }
        HIP_TRY(c, me, "hipMalloc(dense descriptors)");
    }
    if (b->n_desc) {
        RowsArgs a{};
        a.n_tiles = b->pad_n_tiles; a.K = b->pad_K; a.tile0 = 0; a.tile1 = b->pad_n_tiles;
        a.tile_desc_base = const_cast<uint64_t*>(b->pad_tdbase);
        a.desc_pad = reinterpret_cast<uint64_t*>(b->d_desc.ptr()); a.desc = reinterpret_cast<uint64_t*>(b->d_pad.ptr()); a.desc_cap = b->n_desc;
        HIP_TRY(c, launch_rows_compact(a, c->stream), "launch(compact)");
    }
    if (b->n_chunks) HIP_TRY(c, launch_rows_chunks_dense(reinterpret_cast<const Chunk*>(b->d_chunks.ptr()), b->n_chunks, b->pad_tdbase, reinterpret_cast<Chunk*>(b->d_chunks.ptr()), c->stream), "launch(chunk records)");
    std::swap(b->d_desc, b->d_pad);
    b->desc_swapped = !b->desc_swapped;
    b->pad_image = false; b->desc_slots = 0;
    return V2P_OK;
}

int v2p_batch_download_image(v2p_batch* b, uint64_t* desc, v2p_chunk* chunks, uint64_t* hap_out_begin)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    if (b->is_patch && desc) return c->fail(V2P_ERR_STATE, "a patch image has segments and patches, not descriptors: v2p_batch_download_patch_image");
    if (b->is_tiles && (desc || chunks)) return c->fail(V2P_ERR_STATE, "a tile image has pieces in its tiles' slots, not descriptors and chunks (v2p_batch_build_from_stream with kernel 7 builds the dense rows image of the same stream)");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (b->pad_image) { const int rc = densify(b); if (rc) return rc; }       // (a padded image leaves the device in the dense form)
    if (b->pad_image) return c->fail(V2P_ERR_HIP, "no device memory for the dense copy of a padded image");
    if (desc && b->n_desc) HIP_TRY(c, hipMemcpyAsync(desc, b->d_desc.ptr(), b->n_desc * 8, hipMemcpyDeviceToHost, c->stream), "D2H(desc)");
    if (chunks && b->n_chunks) HIP_TRY(c, hipMemcpyAsync(chunks, b->d_chunks.ptr(), b->n_chunks * sizeof(Chunk), hipMemcpyDeviceToHost, c->stream), "D2H(chunks)");
    if (hap_out_begin) HIP_TRY(c, hipMemcpyAsync(hap_out_begin, b->d_hap.ptr(), (b->n_haps + 1) * 8, hipMemcpyDeviceToHost, c->stream), "D2H(hap_begin)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return V2P_OK;
}

int v2p_batch_set_packed(v2p_batch* b,
                         const uint64_t* desc, uint64_t n_desc,
                         const v2p_chunk* chunks, uint64_t n_chunks,
                         const uint8_t* payload, uint64_t n_payload,
                         const uint64_t* hap_out_begin, uint64_t n_haps)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if ((n_desc && !desc) || (n_chunks && !chunks) || (n_payload && !payload) || !hap_out_begin)
        return c->fail(V2P_ERR_INVALID_ARG, "null argument");
    {   // a caller-packed image is checked before it is adopted: the kernels trust these tables
        const int rc = check_packed(c, n_desc, reinterpret_cast<const Chunk*>(chunks), n_chunks, hap_out_begin, n_haps);
        if (rc) return rc;
    }
    b->img = ImageBuilder();
    b->img.desc.assign(desc, desc + n_desc);
    b->img.chunks.resize(n_chunks);
    if (n_chunks) memcpy(b->img.chunks.data(), chunks, n_chunks * sizeof(Chunk));
    b->img.payload.assign(payload, payload + n_payload);
    b->img.hap_out_begin.assign(hap_out_begin, hap_out_begin + n_haps + 1);
    b->uses_proteome = true;
    return V2P_OK;
}

int v2p_batch_finalize(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (b->finalized) return c->fail(V2P_ERR_STATE, "batch already finalized");
    if (b->hap_open) return c->fail(V2P_ERR_STATE, "a haplotype is still open");
    b->img.finish();
    if (b->uses_proteome && !(c->flags & V2P_FLAG_RESULT_ORDER))
        order_chunks_for_xcds(b->img.chunks.data(), b->img.chunks.size(), b->img.desc.data(), b->img.desc.size(), c->proteome_len);
    if (b->img.chunks.size() > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one batch");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    b->n_desc = b->img.desc.size(); b->n_chunks = b->img.chunks.size(); b->n_payload = b->img.payload.size();
    b->launch_hint = stitch_launch_bits(b->img.chunks.data(), b->img.chunks.size());
    b->out_bytes = b->img.out_size(); b->n_haps = b->img.n_haplotypes();
    HIP_TRY(c, b->d_desc.ensure(b->n_desc * 8), "hipMalloc(desc)");
    HIP_TRY(c, b->d_chunks.ensure(b->n_chunks * sizeof(Chunk)), "hipMalloc(chunks)");
    HIP_TRY(c, b->d_payload.ensure(b->n_payload), "hipMalloc(payload)");
    HIP_TRY(c, b->d_out.ensure((b->out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, b->d_hap.ensure((b->n_haps + 1) * 8), "hipMalloc(hap_begin)");
    HIP_TRY(c, b->d_digest.ensure((b->n_haps ? b->n_haps : 1) * 8), "hipMalloc(digest)");
    if (b->n_desc) HIP_TRY(c, hipMemcpyAsync(b->d_desc.ptr(), b->img.desc.data(), b->n_desc * 8, hipMemcpyHostToDevice, c->stream), "H2D(desc)");
    if (b->n_chunks) HIP_TRY(c, hipMemcpyAsync(b->d_chunks.ptr(), b->img.chunks.data(), b->n_chunks * sizeof(Chunk), hipMemcpyHostToDevice, c->stream), "H2D(chunks)");
    if (b->n_payload) HIP_TRY(c, hipMemcpyAsync(b->d_payload.ptr(), b->img.payload.data(), b->n_payload, hipMemcpyHostToDevice, c->stream), "H2D(payload)");
    HIP_TRY(c, hipMemcpyAsync(b->d_hap.ptr(), b->img.hap_out_begin.data(), (b->n_haps + 1) * 8, hipMemcpyHostToDevice, c->stream), "H2D(hap_begin)");
    b->payload_dev = b->d_payload.ptr();
    int rc = init_status(c, b->d_status);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    // the host copy of the image is no longer needed
    std::vector<uint64_t>().swap(b->img.desc);
    std::vector<Chunk>().swap(b->img.chunks);
    std::vector<uint8_t>().swap(b->img.payload);
    b->finalized = true;
    return V2P_OK;
}

// A dense rows image (every chunk a dense chunk on a 1 KiB row: launch_hint bits 1, 3, 4, 5 and not 2) that is executed AGAIN is
// re-written as pieces first (dense_pieces.h), once; an image the form does not take (sources beyond 2 GiB, a descriptor the dense
// kernel would refuse and report) stays on the dense kernel.  libv2p_bench.so's variant 28 (A/B): never.
static bool pieces_eligible(const v2p_batch* b)
{
    const int h = b->launch_hint;
    return b->finalized && !b->is_patch && !b->is_tiles && (h & 2) && (h & 8) && !(h & 4) && (h & 16) && (h & 32) && b->n_chunks != 0 && ctx_variant(b->ctx) != 28u &&
           ctx_variant(b->ctx) != 3u && ctx_variant(b->ctx) != 8u;
}
static int to_pieces(v2p_batch* b)
{
    v2p_ctx* c = b->ctx;
    if (b->pieces_state != 0) return V2P_OK;
    b->pieces_state = -1;
    DevBuf cnt, basebuf, scratch, st;
    struct Rel { DevBuf& a; DevBuf& b; DevBuf& c; DevBuf& d; ~Rel() { a.release(); b.release(); c.release(); d.release(); } } rel{cnt, basebuf, scratch, st};
    const uint64_t nc = b->n_chunks;
    // (no room for the conversion's scratch: not converted -- the dense kernel goes on executing the image)
    if (cnt.ensure_exact((nc + 1) * 4) != hipSuccess || basebuf.ensure_exact((nc + 2) * 8) != hipSuccess ||
        scratch.ensure_exact(scan_tiles_for(nc + 1) * 8 + 64) != hipSuccess || st.ensure_exact(8) != hipSuccess) { (void)hipGetLastError(); return V2P_OK; }
    HIP_TRY(c, hipMemsetAsync(st.ptr(), 0xFF, 8, c->stream), "hipMemset(status)");
    PieceBuildArgs a{};
    a.desc = reinterpret_cast<const uint64_t*>(b->d_desc.ptr()); a.n_desc = b->n_desc;
    a.chunks = reinterpret_cast<const Chunk*>(b->d_chunks.ptr()); a.n_chunks = uint32_t(nc);
    a.src0_len = c->proteome_len + c->headers_len; a.src1_len = b->n_payload; a.out_len = b->out_bytes;
    a.count = reinterpret_cast<uint32_t*>(cnt.ptr()); a.base = reinterpret_cast<const uint64_t*>(basebuf.ptr());
    a.status = reinterpret_cast<unsigned long long*>(st.ptr());
    HIP_TRY(c, launch_pieces_build(a, 0, c->stream), "launch(pieces: count)");
    HIP_TRY(c, launch_scan_u32(a.count, nc, const_cast<uint64_t*>(a.base), reinterpret_cast<uint64_t*>(scratch.ptr()), c->stream), "launch(scan)");
    uint64_t total = 0;
    unsigned long long stw = ~0ull;
    HIP_TRY(c, hipMemcpyAsync(&total, basebuf.ptr() + nc * 8, 8, hipMemcpyDeviceToHost, c->stream), "D2H(pieces)");
    HIP_TRY(c, hipMemcpyAsync(&stw, st.ptr(), 8, hipMemcpyDeviceToHost, c->stream), "D2H(status)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (stw != ~0ull || total == 0 || total >= (1ull << 40)) return V2P_OK;            // (not converted: the dense kernel executes the image, and reports what it refuses)
    if (b->d_pieces.ensure(total * 8) != hipSuccess || b->d_chunks2.ensure(nc * sizeof(Chunk)) != hipSuccess) { (void)hipGetLastError(); return V2P_OK; }
    b->n_pieces = total; b->pieces_src0_len = a.src0_len; b->pieces_src1_len = a.src1_len;
    a.pieces = reinterpret_cast<uint64_t*>(b->d_pieces.ptr()); a.chunks2 = reinterpret_cast<Chunk*>(b->d_chunks2.ptr());
    HIP_TRY(c, launch_pieces_build(a, 1, c->stream), "launch(pieces: write)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");                 // (the scratch above is released on return)
    b->pieces_state = 1;
    return V2P_OK;
}

int v2p_batch_execute(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    if (b->orphaned) return c->fail(V2P_ERR_STATE, "the v2p_stream this batch was built from has been destroyed: its image cannot be executed again (the arena stays readable)");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
#ifdef V2P_BENCH_VARIANTS
    if (b->is_patch) {
        const hipError_t pe = patch_execute(b, c->stream);
        if (pe != hipSuccess) return c->hip_fail(pe, "launch(stitch: patch image)");
        return V2P_OK;
    }
#endif
    if (b->is_tiles) {
        // (the executor checks no source bound: the parse did, against the reference of that moment)
        if (c->proteome_len + c->headers_len < b->pieces_src0_len) return c->fail(V2P_ERR_SRC_OOB, "the resident reference is shorter than the one this tile image was built against: rebuild the batch");
        HIP_TRY(c, tiles_execute(b, c->stream), "launch(stitch: tile image)");
        b->executed = true;
        return V2P_OK;
    }
    // (variants 23 / 26, A/B: a padded image stays padded -- read in place / staged)
    if (b->pad_image && ctx_variant(c) != 23u && ctx_variant(c) != 26u) { const int rc = densify(b); if (rc) return rc; }
    if (pieces_eligible(b) && b->executed) {
        if (b->pieces_state == 0) { const int rc = to_pieces(b); if (rc) return rc; }
        // (the piece kernel checks no source bound: to_pieces did, against the reference of that moment -- behind a v2p_upload_reference
        // of a SHORTER reference the image goes back to the dense kernel, which checks every execute and reports V2P_ERR_SRC_OOB)
        if (b->pieces_state == 1 && (c->proteome_len + c->headers_len < b->pieces_src0_len || b->n_payload < b->pieces_src1_len)) b->pieces_state = -1;
        if (b->pieces_state == 1) {
            PieceExecArgs pa{reinterpret_cast<const uint64_t*>(b->d_pieces.ptr()), reinterpret_cast<const Chunk*>(b->d_chunks2.ptr()), uint32_t(b->n_chunks),
                             c->proteome.ptr(), b->payload_dev, b->d_out.ptr(), b->out_bytes};
            HIP_TRY(c, launch_stitch_pieces(pa, c->stream, !(c->flags & V2P_FLAG_TEMPORAL)), "launch(stitch: pieces)");
            return V2P_OK;
        }
    }
    StitchArgs a{reinterpret_cast<const uint64_t*>(b->d_desc.ptr()), b->pad_image ? b->desc_slots : b->n_desc, reinterpret_cast<const Chunk*>(b->d_chunks.ptr()),
                 uint32_t(b->n_chunks), c->proteome.ptr(), c->proteome_len + c->headers_len, b->payload_dev, b->n_payload,
                 b->d_out.ptr(), b->out_bytes, reinterpret_cast<unsigned long long*>(b->d_status.ptr())};
    a.opt_phase_bytes = c->launch_opts.phase_bytes; a.opt_phase_min_chunks = c->launch_opts.phase_min_chunks; a.opt_store_sc1 = c->launch_opts.store_sc1;
    a.opt_touch = touch_of(ctx_variant(c));
    a.img_desc = b->pad_image ? b->n_desc : 0;          // (the routing looks at the descriptors the image holds, not at the array's slots)
    dual_of(c, a);
    const int hint = int(!(c->flags & V2P_FLAG_TEMPORAL)) | b->launch_hint;
    (void)attach_stage(b, a, hint);
    HIP_TRY(c, launch_stitch(a, c->stream, hint, 0), "launch(stitch)");
    b->executed = true;
    return V2P_OK;
}

int v2p_batch_sync(v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = collect_status(c, b->d_status);
    if (rc) (void)hipMemsetAsync(b->d_status.ptr(), 0xFF, sizeof(unsigned long long), c->stream);
    return rc;
}

int v2p_batch_counts(const v2p_batch* b, uint64_t* n_haps, uint64_t* n_desc, uint64_t* n_chunks,
                     uint64_t* out_bytes, uint64_t* payload_bytes)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    const bool f = b->finalized;
    if (n_haps) *n_haps = f ? b->n_haps : b->img.n_haplotypes();
    if (n_desc) *n_desc = f ? b->n_desc : b->img.desc.size();
    if (n_chunks) *n_chunks = f ? b->n_chunks : b->img.chunks.size();
    if (out_bytes) *out_bytes = f ? b->out_bytes : b->img.out_size();
    if (payload_bytes) *payload_bytes = f ? b->n_payload : b->img.payload.size();
    return V2P_OK;
}

int v2p_batch_image_form(const v2p_batch* b)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    return (b->pad_image ? 1 : 0) | (b->pieces_state == 1 ? 2 : 0) | (b->d_stage.ptr() ? 4 : 0) | (b->is_tiles ? 8 : 0);
}

int v2p_batch_hap_range(const v2p_batch* b, uint64_t h, uint64_t* begin, uint64_t* len)
{
    if (!b || !begin || !len) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    const std::vector<uint64_t>& hb = b->img.hap_out_begin;
    if (h + 1 >= hb.size()) return V2P_ERR_INVALID_ARG;
    *begin = hb[h]; *len = hb[h + 1] - hb[h];
    return V2P_OK;
}

int v2p_batch_download(v2p_batch* b, uint64_t begin, uint64_t len, uint8_t* out)
{
    if (!b || (len && !out)) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    if (begin + len > b->out_bytes) return c->fail(V2P_ERR_INVALID_ARG, "range outside the arena");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (len) HIP_TRY(c, hipMemcpyAsync(out, b->d_out.ptr() + begin, len, hipMemcpyDeviceToHost, c->stream), "D2H(out)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return V2P_OK;
}

int v2p_batch_digests(v2p_batch* b, uint64_t* digests, uint64_t n_haps)
{
    if (!b || (n_haps && !digests)) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    if (n_haps != b->n_haps) return c->fail(V2P_ERR_INVALID_ARG, "n_haps mismatch");
    if (n_haps == 0) return V2P_OK;
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    HIP_TRY(c, hipMemsetAsync(b->d_digest.ptr(), 0, n_haps * 8, c->stream), "hipMemset(digest)");
    DigestArgs a{b->d_out.ptr(), reinterpret_cast<const uint64_t*>(b->d_hap.ptr()), n_haps, reinterpret_cast<uint64_t*>(b->d_digest.ptr())};
    HIP_TRY(c, launch_digest(a, b->out_bytes, c->stream), "launch(digest)");
    HIP_TRY(c, hipMemcpyAsync(digests, b->d_digest.ptr(), n_haps * 8, hipMemcpyDeviceToHost, c->stream), "D2H(digest)");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    return V2P_OK;
}

void* v2p_batch_device_out(v2p_batch* b) { return (b && b->finalized) ? b->d_out.ptr() : nullptr; }

int v2p_batch_scribble(v2p_batch* b, int byte)
{
    if (!b) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!b->finalized) return c->fail(V2P_ERR_STATE, "batch not finalized");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (b->out_bytes) HIP_TRY(c, hipMemsetAsync(b->d_out.ptr(), byte & 0xFF, b->out_bytes, c->stream), "hipMemset(arena)");
    return V2P_OK;
}

// ---- streamed pipeline: H2D / kernel / D2H of successive images overlap ---------------
// Two kinds of submission share the slots.
//   v2p_pipeline_submit          a host-PACKED image (round 2): staged, uploaded, stitched, downloaded on the slot's stream.
//   v2p_pipeline_submit_stream   a slice of the TRANSCRIPT STREAM -- Task vectors exactly as step 4b returns them (round 6): the caller's
//                                thread checks the slice's tables and copies its arrays into the slot's pinned staging (a small team of
//                                copy threads), the H2D runs on the slot's stream, the pipeline's RUNNER thread takes the slices in
//                                submission order through the one call (image build + execute on the context's streams -- its looks at
//                                the counts block the runner, nobody else), and the arena travels into the slot's pinned result buffer on
//                                the slot's second stream.  What the reference does per sample (get_g_rep(..).execute(engine), then the
//                                bytes to the host: personalized_genome.rs:61-69, parts/exec.rs:23-42) with nothing packed on the host.

enum : int { SLOT_FREE = 0, SLOT_QUEUED = 1, SLOT_LAUNCHED = 2, SLOT_FAILED = 3 };

struct PipeSlot {
    hipStream_t stream = nullptr;
    hipStream_t d2h = nullptr;    // stream submissions: the result's way home (the slot's `stream` carries the slice's H2D)
    hipEvent_t done = nullptr, ev_h2d = nullptr, ev_exec = nullptr;
    DevBuf d_desc, d_chunks, d_payload, d_out, d_status;
    PinnedBuf h_in, h_out;
    uint64_t out_bytes = 0;
    unsigned long long status = STATUS_CLEAN;
    bool busy = false;        // holds a result the caller has not released
    bool in_flight = false;   // work was enqueued on `stream` and `done` has not been waited for
    // ---- stream submissions ----
    bool is_stream = false;
    int state = SLOT_FREE;                 // (under v2p_pipeline::pmu)
    v2p_stream* rs = nullptr;              // the slice on the device: buffers kept from submission to submission
    v2p_batch* batch = nullptr;            // ... and its image / arena
    StreamLayout lay{};
    v2p_txstream shape{};                  // the slice's counts (pointers cleared) for stream_view
    DevStreamView stats;                   // the slice's Task shapes, sampled by the submitter (stream_item_stats)
    bool fasta = false;
    int kernel = 0;
    unsigned sflags = 0;
    uint64_t n_haps = 0;
    std::vector<uint64_t> hap_out_begin;   // offsets of the slice's haplotypes inside the result
    int rc = V2P_OK;
    std::string err;
    int64_t err_index = -1;
    uint64_t o_status = 0, o_digests = 0;  // where the status word and the digests sit in h_out
    double t_stage_ms = 0, t_queue_ms = 0, t_gpu_ms = 0;
};

struct v2p_pipeline {
    v2p_ctx* ctx = nullptr;
    std::vector<PipeSlot> slots;
    uint32_t next = 0;
    // stream submissions
    std::mutex pmu;
    std::condition_variable cv;
    std::vector<uint32_t> jobs;            // slots queued for the runner, in submission order
    bool stop = false;
    std::thread runner;
    uint32_t copy_threads = 8;
};

static void pipeline_runner(v2p_pipeline* p);

int v2p_pipeline_create(v2p_ctx* c, uint32_t n_slots, v2p_pipeline** out)
{
    if (!c || !out || n_slots == 0 || n_slots > 16) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    v2p_pipeline* p = new (std::nothrow) v2p_pipeline();
    if (!p) return c->fail(V2P_ERR_HIP, "out of host memory");
    p->ctx = c;
    p->slots.resize(n_slots);
    for (PipeSlot& s : p->slots) {
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e != hipSuccess) { delete p; return c->hip_fail(e, "pipeline stream/event"); }
    }
    *out = p;
    return V2P_OK;
}

void v2p_pipeline_destroy(v2p_pipeline* p)
{
    if (!p) return;
    if (p->runner.joinable()) {
        { std::lock_guard<std::mutex> lk(p->pmu); p->stop = true; }
        p->cv.notify_all();
        p->runner.join();
    }
    (void)hipSetDevice(p->ctx->device);
    for (PipeSlot& s : p->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.d2h) (void)hipStreamSynchronize(s.d2h);
        if (s.batch) v2p_batch_destroy(s.batch);
        if (s.rs) v2p_stream_destroy(s.rs);
        s.d_desc.release(); s.d_chunks.release(); s.d_payload.release(); s.d_out.release(); s.d_status.release();
        s.h_in.release(); s.h_out.release();
        for (hipEvent_t e : {s.done, s.ev_h2d, s.ev_exec}) if (e) (void)hipEventDestroy(e);
        if (s.d2h) (void)hipStreamDestroy(s.d2h);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete p;
}

int v2p_pipeline_submit(v2p_pipeline* p,
                        const uint64_t* desc, uint64_t n_desc,
                        const v2p_chunk* chunks, uint64_t n_chunks,
                        const uint8_t* payload, uint64_t n_payload,
                        uint64_t out_bytes, uint32_t* ticket)
{
    if (!p || !ticket || (n_desc && !desc) || (n_chunks && !chunks) || (n_payload && !payload)) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (n_chunks > 0xFFFFFFFFull) return c->fail(V2P_ERR_UNSUPPORTED, "more than 2^32 chunks in one image");
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    uint32_t t;
    {
        std::lock_guard<std::mutex> pl(p->pmu);
        t = p->next;
        if (p->slots[t].busy) return c->fail(V2P_ERR_STATE, "pipeline slot still holds an unreleased result");
    }
    PipeSlot& s = p->slots[t];
    {
        const int rc = check_packed(c, n_desc, reinterpret_cast<const Chunk*>(chunks), n_chunks, nullptr, 0, out_bytes);
        if (rc) return rc;
    }
    // the slot's staging and device buffers are about to be rewritten (and possibly reallocated): whatever was
    // enqueued on it before -- a submit that failed half way, a result released without a wait -- must be done
    if (s.in_flight) { HIP_TRY(c, hipStreamSynchronize(s.stream), "hipStreamSynchronize(slot)"); if (s.d2h) HIP_TRY(c, hipStreamSynchronize(s.d2h), "hipStreamSynchronize(slot)"); s.in_flight = false; }
    const size_t b_desc = size_t(n_desc) * 8, b_chunks = size_t(n_chunks) * sizeof(Chunk);
    const size_t o_chunks = (b_desc + 15) & ~size_t(15), o_payload = (o_chunks + b_chunks + 15) & ~size_t(15);
    HIP_TRY(c, s.h_in.ensure(o_payload + n_payload), "hipHostMalloc(in)");
    HIP_TRY(c, s.h_out.ensure(out_bytes + 16), "hipHostMalloc(out)");            // result + the 8-byte status word behind it, 8-aligned
    HIP_TRY(c, s.d_desc.ensure(b_desc), "hipMalloc(desc)");
    HIP_TRY(c, s.d_chunks.ensure(b_chunks), "hipMalloc(chunks)");
    HIP_TRY(c, s.d_payload.ensure(n_payload), "hipMalloc(payload)");
    HIP_TRY(c, s.d_out.ensure((out_bytes + 15) & ~15ull), "hipMalloc(out)");
    HIP_TRY(c, s.d_status.ensure(sizeof(unsigned long long)), "hipMalloc(status)");
    // stage the image in pinned memory so the three copies are truly asynchronous
    if (b_desc) memcpy(s.h_in.p, desc, b_desc);
    if (b_chunks) {
        memcpy(s.h_in.p + o_chunks, chunks, b_chunks);
        if (!(c->flags & V2P_FLAG_RESULT_ORDER))
            order_chunks_for_xcds(reinterpret_cast<Chunk*>(s.h_in.p + o_chunks), n_chunks, desc, n_desc, c->proteome_len);
    }
    if (n_payload) memcpy(s.h_in.p + o_payload, payload, n_payload);
    s.in_flight = true;       // from here on the stream may hold work that reads h_in / writes h_out
    s.is_stream = false;
    if (b_desc) HIP_TRY(c, hipMemcpyAsync(s.d_desc.ptr(), s.h_in.p, b_desc, hipMemcpyHostToDevice, s.stream), "H2D(desc)");
    if (b_chunks) HIP_TRY(c, hipMemcpyAsync(s.d_chunks.ptr(), s.h_in.p + o_chunks, b_chunks, hipMemcpyHostToDevice, s.stream), "H2D(chunks)");
    if (n_payload) HIP_TRY(c, hipMemcpyAsync(s.d_payload.ptr(), s.h_in.p + o_payload, n_payload, hipMemcpyHostToDevice, s.stream), "H2D(payload)");
    HIP_TRY(c, hipMemsetAsync(s.d_status.ptr(), 0xFF, sizeof(unsigned long long), s.stream), "hipMemset(status)");
    StitchArgs a{reinterpret_cast<const uint64_t*>(s.d_desc.ptr()), n_desc, reinterpret_cast<const Chunk*>(s.d_chunks.ptr()),
                 uint32_t(n_chunks), c->proteome.ptr(), c->proteome_len + c->headers_len, s.d_payload.ptr(), n_payload,
                 s.d_out.ptr(), out_bytes, reinterpret_cast<unsigned long long*>(s.d_status.ptr())};
    HIP_TRY(c, launch_stitch(a, s.stream, int(!(c->flags & V2P_FLAG_TEMPORAL)) | stitch_launch_bits(reinterpret_cast<const Chunk*>(chunks), n_chunks), 0), "launch(stitch)");
    if (out_bytes) HIP_TRY(c, hipMemcpyAsync(s.h_out.p, s.d_out.ptr(), out_bytes, hipMemcpyDeviceToHost, s.stream), "D2H(out)");
    HIP_TRY(c, hipMemcpyAsync(s.h_out.p + ((out_bytes + 7) & ~7ull), s.d_status.ptr(), sizeof(unsigned long long), hipMemcpyDeviceToHost, s.stream), "D2H(status)");
    HIP_TRY(c, hipEventRecord(s.done, s.stream), "hipEventRecord");
    s.out_bytes = out_bytes; s.o_status = (out_bytes + 7) & ~7ull;
    {
        std::lock_guard<std::mutex> pl(p->pmu);
        s.busy = true; s.state = SLOT_LAUNCHED; s.rc = V2P_OK;
        p->next = (t + 1) % uint32_t(p->slots.size());
    }
    *ticket = t;
    return V2P_OK;
}

// pageable -> pinned, a team of threads over 8 MiB pieces (one thread's memcpy moves 6-10 GB/s; the link takes 50)
static void team_copy(uint8_t* dst_base, const StreamPiece* pc, uint32_t np, uint32_t n_threads, const std::function<void()>& leader_first)
{
    struct Job { uint8_t* d; const uint8_t* s; uint64_t n; };
    std::vector<Job> jobs;
    constexpr uint64_t PIECE = 8ull << 20;
    uint64_t total = 0;
    for (uint32_t k = 0; k < np; ++k) {
        total += pc[k].bytes;
        for (uint64_t o = 0; o < pc[k].bytes; o += PIECE)
            jobs.push_back(Job{dst_base + pc[k].off + o, static_cast<const uint8_t*>(pc[k].src) + o, pc[k].bytes - o < PIECE ? pc[k].bytes - o : PIECE});
    }
    uint32_t T = n_threads ? n_threads : 1;
    if (T > jobs.size()) T = uint32_t(jobs.size());
    if (total < (32ull << 20) || T <= 1) { leader_first(); for (const Job& j : jobs) memcpy(j.d, j.s, j.n); return; }
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t i = next++; i < jobs.size(); i = next++) memcpy(jobs[i].d, jobs[i].s, jobs[i].n); };
    std::vector<std::thread> team;
    for (uint32_t t = 1; t < T; ++t) { try { team.emplace_back(work); } catch (...) { break; } }      // (no thread to be had: the others take its share)
    leader_first();                                     // (the calling thread: the table checks, then its share of the copy)
    work();
    for (std::thread& th : team) th.join();
}

int v2p_pipeline_reserve(v2p_pipeline* p, uint64_t stream_bytes, uint64_t out_bytes, uint32_t copy_threads)
{
    if (!p) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipSetDevice(c->device), "hipSetDevice");
    if (copy_threads) p->copy_threads = copy_threads > 64 ? 64 : copy_threads;
    for (PipeSlot& s : p->slots) {
        if (s.busy || s.in_flight) return c->fail(V2P_ERR_STATE, "v2p_pipeline_reserve: a slot is in use");
        if (stream_bytes) HIP_TRY(c, s.h_in.ensure(stream_bytes), "hipHostMalloc(in)");
        if (out_bytes) HIP_TRY(c, s.h_out.ensure(out_bytes + 64), "hipHostMalloc(out)");
    }
    return V2P_OK;
}

int v2p_pipeline_submit_stream(v2p_pipeline* p, const v2p_txstream* slice, int kernel, unsigned flags, uint32_t* ticket)
{
    if (!p || !slice || !ticket) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    if (kernel != 0 && kernel != 6 && kernel != 7 && kernel != 9) { std::lock_guard<std::mutex> lk(c->mu); return c->fail(V2P_ERR_INVALID_ARG, "a stream slice builds rows images (kernel 6, 7), a tile image (9) or what the routing rule picks (0)"); }
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t t;
    {
        std::lock_guard<std::mutex> pl(p->pmu);
        // the first free slot from `next` on (one submitter that releases in order sees them round-robin; workers that release as they
        // finish take whichever is free); every slot in use: V2P_BUSY, nothing staged -- wait for a ticket, release it, submit again
        const uint32_t ns = uint32_t(p->slots.size());
        uint32_t k = 0;
        while (k < ns && p->slots[(p->next + k) % ns].busy) ++k;
        if (k == ns) return V2P_BUSY;
        t = (p->next + k) % ns;
        p->slots[t].busy = true;                       // claimed: concurrent submitters take the slots behind it
        p->slots[t].state = SLOT_FREE;
        p->next = (t + 1) % ns;
    }
    PipeSlot& s = p->slots[t];
    auto unclaim = [&](int rc) { std::lock_guard<std::mutex> pl(p->pmu); s.busy = false; return rc; };
    auto hip_unclaim = [&](hipError_t e, const char* what) { std::lock_guard<std::mutex> lk(c->mu); return unclaim(c->hip_fail(e, what)); };
    (void)hipSetDevice(c->device);
    if (!s.d2h || !s.ev_h2d || !s.ev_exec || !s.rs || !s.batch) {               // (a slot's first stream submission)
        hipError_t e = hipSuccess;
        if (!s.d2h) e = hipStreamCreateWithFlags(&s.d2h, hipStreamNonBlocking);
        if (e == hipSuccess && !s.ev_h2d) e = hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming);
        if (e == hipSuccess && !s.ev_exec) e = hipEventCreateWithFlags(&s.ev_exec, hipEventDisableTiming);
        if (e != hipSuccess) return hip_unclaim(e, "pipeline stream/event");
        if (!s.rs) { s.rs = new (std::nothrow) v2p_stream(); if (s.rs) s.rs->ctx = c; }
        if (!s.batch) { s.batch = new (std::nothrow) v2p_batch(); if (s.batch) { s.batch->ctx = c; s.batch->grow = true; } }
        if (!s.rs || !s.batch) { std::lock_guard<std::mutex> lk(c->mu); return unclaim(c->fail(V2P_ERR_HIP, "out of host memory")); }
    }
    // a result released without a wait, a submission that failed half way: the slot's streams must be idle before its buffers are rewritten
    if (s.in_flight) { (void)hipStreamSynchronize(s.stream); (void)hipStreamSynchronize(s.d2h); s.in_flight = false; }
    if (!slice->hap_tx_begin) { std::lock_guard<std::mutex> lk(c->mu); return unclaim(c->fail(V2P_ERR_INVALID_ARG, "null argument")); }
    const bool fasta_shape = slice->tx_header_off && slice->tx_header_len;
    const StreamLayout L = stream_layout(slice->n_haps, slice->n_tx, slice->n_tasks, fasta_shape);
    const uint64_t o_alt = (L.total + 255) & ~255ull;
    {
        hipError_t e = s.h_in.ensure(o_alt + slice->n_alt + 64);
        if (e == hipSuccess) e = s.rs->buf.ensure(L.total);
        if (e == hipSuccess) e = s.rs->alt.ensure(slice->n_alt);
        if (e != hipSuccess) return hip_unclaim(e, "pipeline buffers");
    }
    // ---- the caller's thread and its copy team.  The team copies the slice's arrays into pinned staging, in the device layout, WHILE this
    // thread checks the slice's tables -- what the kernels index device memory through is refused before anything is uploaded (as
    // v2p_stream_upload; without the context's lock: the runner holds it through a whole one call) -- and samples its Task shapes; a slice
    // that fails the check has been copied for nothing (null arrays are caught first: the team never reads through one).
    bool fasta = false;
    std::vector<uint64_t> hob;
    CheckErr cerr;
    int crc = V2P_OK;
    if ((slice->n_tx && (!slice->tx_proteome_off || !slice->tx_ref_len || !slice->tx_res_len || !slice->tx_task_begin || !slice->tx_alt_begin)) ||
        (slice->n_tasks && (!slice->code || !slice->start_pos || !slice->length || !slice->start_pos_res)) || (slice->n_alt && !slice->alt) ||
        ((slice->tx_header_off == nullptr) != (slice->tx_header_len == nullptr)))
        crc = check_stream_nolock(c, slice, &fasta, nullptr, &hob, cerr);                 // (it says which)
    if (crc == V2P_OK) {
        StreamPiece pc[13];
        uint32_t np = stream_pieces(slice, L, fasta_shape, pc);
        if (slice->n_alt) pc[np++] = StreamPiece{o_alt, slice->alt, slice->n_alt, "H2D(alt)"};
        team_copy(s.h_in.p, pc, np, p->copy_threads, [&] {
            crc = check_stream_nolock(c, slice, &fasta, nullptr, &hob, cerr);
            if (crc == V2P_OK) stream_item_stats(slice, s.stats);
        });
    }
    if (crc != V2P_OK) { std::lock_guard<std::mutex> lk(c->mu); return unclaim(c->fail(crc == V2P_OK ? V2P_ERR_INVALID_ARG : crc, cerr.msg, cerr.index)); }
    const uint64_t out_bytes = hob.back();
    const uint64_t o_status = (out_bytes + 63) & ~63ull, o_dig = o_status + 64;
    {
        const hipError_t e = s.h_out.ensure(o_dig + ((flags & V2P_SUBMIT_DIGESTS) ? slice->n_haps * 8 : 0) + 64);
        if (e != hipSuccess) return hip_unclaim(e, "pipeline buffers");
    }
    s.lay = L; s.fasta = fasta; s.kernel = kernel; s.sflags = flags; s.n_haps = slice->n_haps;
    s.shape = *slice;
    s.shape.hap_tx_begin = nullptr; s.shape.tx_proteome_off = nullptr; s.shape.tx_ref_len = nullptr; s.shape.tx_res_len = nullptr; s.shape.tx_task_begin = nullptr;
    s.shape.tx_alt_begin = nullptr; s.shape.code = nullptr; s.shape.start_pos = nullptr; s.shape.length = nullptr; s.shape.start_pos_res = nullptr; s.shape.alt = nullptr;
    s.shape.tx_header_off = nullptr; s.shape.tx_header_len = nullptr;
    s.hap_out_begin.swap(hob);
    s.out_bytes = out_bytes; s.o_status = o_status; s.o_digests = o_dig;
    s.rc = V2P_OK; s.err.clear(); s.err_index = -1; s.is_stream = true;
    s.t_stage_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // ---- H2D on the slot's stream: two copies from pinned memory, truly asynchronous ----
    s.in_flight = true;
    hipError_t e = hipSuccess;
    if (L.total) e = hipMemcpyAsync(s.rs->buf.ptr(), s.h_in.p, L.total, hipMemcpyHostToDevice, s.stream);
    if (e == hipSuccess && slice->n_alt) e = hipMemcpyAsync(s.rs->alt.ptr(), s.h_in.p + o_alt, slice->n_alt, hipMemcpyHostToDevice, s.stream);
    if (e == hipSuccess) e = hipEventRecord(s.ev_h2d, s.stream);
    if (e != hipSuccess) return hip_unclaim(e, "H2D(stream slice)");
    {
        std::unique_lock<std::mutex> pl(p->pmu);
        if (!p->runner.joinable()) {
            try { p->runner = std::thread(pipeline_runner, p); }
            catch (...) { pl.unlock(); std::lock_guard<std::mutex> lk(c->mu); return unclaim(c->fail(V2P_ERR_HIP, "no thread for the pipeline's runner")); }
        }
        s.state = SLOT_QUEUED;
        p->jobs.push_back(t);
    }
    p->cv.notify_all();
    *ticket = t;
    return V2P_OK;
}

// The runner: slice after slice, in submission order, through the one call.  Holds the context while it enqueues (and while the one
// call looks at its counts); submitters stage and upload meanwhile, waiters sleep on the slots' events.
static void pipeline_runner(v2p_pipeline* p)
{
    v2p_ctx* c = p->ctx;
    for (;;) {
        uint32_t t;
        {
            std::unique_lock<std::mutex> pl(p->pmu);
            p->cv.wait(pl, [&] { return p->stop || !p->jobs.empty(); });
            if (p->jobs.empty()) return;               // (stop: what is queued is still run -- its waiters are owed an answer)
            t = p->jobs.front();
            p->jobs.erase(p->jobs.begin());
        }
        PipeSlot& s = p->slots[t];
        const auto t0 = std::chrono::steady_clock::now();
        int rc = V2P_OK;
        {
            std::lock_guard<std::mutex> lk(c->mu);
            auto hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == V2P_OK) rc = c->hip_fail(e, what); return e == hipSuccess; };
            v2p_stream* st = s.rs;
            v2p_batch* b = s.batch;
            hip(hipSetDevice(c->device), "hipSetDevice");
            if (rc == V2P_OK) rc = batch_reset_locked(b, false);
            if (rc == V2P_OK) hip(hipStreamWaitEvent(c->stream, s.ev_h2d, 0), "hipStreamWaitEvent");
            if (rc == V2P_OK) {
                // the slice as a resident stream (v2p_stream_upload's second half): the view, res_counter per tile and per haplotype
                stream_view(&s.shape, s.lay, s.fasta, st->buf.ptr(), st->alt.ptr(), st->v);
                copy_stats(s.stats, st->v);
                st->hap_out_begin = s.hap_out_begin;
                st->out_bytes = s.out_bytes;
                st->tile_K = 0; st->n_tiles = 0; st->tile_res_base = nullptr; st->d_hap_out_begin = nullptr; st->h_tile_res_base.clear();
                const uint32_t K = rows_pick_k(st->v, rows_mode_for(st, s.kernel));
                const uint64_t n_tiles = st->v.n_tx ? (st->v.n_tx + K - 1) / K : 1;
                auto up8 = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
                const uint64_t o_tbytes = 0, o_tbase = up8(n_tiles * 8), o_hap = o_tbase + up8((n_tiles + 1) * 8), o_scan = o_hap + up8((st->v.n_haps + 1) * 8),
                               o_end = o_scan + up8(rows_scan_scratch_entries(n_tiles) * 8);
                if (st->tiles.ensure(o_end) != hipSuccess) (void)hipGetLastError();            // (no room: the call makes its own tables)
                else {
                    RowsArgs a;
                    rows_args_of(st->v, c, K, n_tiles, a);
                    uint8_t* const d = st->tiles.ptr();
                    a.tile_bytes = reinterpret_cast<uint64_t*>(d + o_tbytes); a.tile_res_base = reinterpret_cast<uint64_t*>(d + o_tbase);
                    a.hap_out_begin = reinterpret_cast<uint64_t*>(d + o_hap);
                    a.status = nullptr;
                    if (hip(launch_rows_tile_bytes(a, reinterpret_cast<uint64_t*>(d + o_scan), c->stream), "launch(tile tables)") &&
                        hip(launch_rows_hap_begin(a, c->stream), "launch(tile tables)")) {
                        st->tile_K = K; st->n_tiles = n_tiles; st->tile_res_base = a.tile_res_base; st->d_hap_out_begin = a.hap_out_begin;
                    }
                }
            }
            if (rc == V2P_OK) rc = build_and_execute_locked(b, st, s.kernel, 0);
            if (rc == V2P_OK) {
                if ((s.sflags & V2P_SUBMIT_DIGESTS) && b->n_haps) {
                    hip(hipMemsetAsync(b->d_digest.ptr(), 0, b->n_haps * 8, c->stream), "hipMemset(digest)");
                    DigestArgs da{b->d_out.ptr(), reinterpret_cast<const uint64_t*>(b->d_hap.ptr()), b->n_haps, reinterpret_cast<uint64_t*>(b->d_digest.ptr())};
                    hip(launch_digest(da, b->out_bytes, c->stream), "launch(digest)");
                }
                hip(hipEventRecord(s.ev_exec, c->stream), "hipEventRecord");
                hip(hipStreamWaitEvent(s.d2h, s.ev_exec, 0), "hipStreamWaitEvent");
                // the way home: arena, status word, digests -- into pinned memory, on a stream of the slot's own (the next slice's H2D and its
                // one call run beside it)
                if (rc == V2P_OK && b->out_bytes) hip(hipMemcpyAsync(s.h_out.p, b->d_out.ptr(), b->out_bytes, hipMemcpyDeviceToHost, s.d2h), "D2H(out)");
                if (rc == V2P_OK) hip(hipMemcpyAsync(s.h_out.p + s.o_status, b->d_status.ptr(), 8, hipMemcpyDeviceToHost, s.d2h), "D2H(status)");
                if (rc == V2P_OK && (s.sflags & V2P_SUBMIT_DIGESTS) && b->n_haps) hip(hipMemcpyAsync(s.h_out.p + s.o_digests, b->d_digest.ptr(), b->n_haps * 8, hipMemcpyDeviceToHost, s.d2h), "D2H(digests)");
                hip(hipEventRecord(s.done, s.d2h), "hipEventRecord");
                if (rc == V2P_OK && b->out_bytes != s.out_bytes) rc = c->fail(V2P_ERR_STATE, "a slice's arena is not the sum of its transcripts' result sizes");
            }
            if (rc != V2P_OK) { s.err = c->err; s.err_index = c->err_index; }
        }
        {
            std::lock_guard<std::mutex> pl(p->pmu);
            s.rc = rc;
            s.t_gpu_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            s.state = rc == V2P_OK ? SLOT_LAUNCHED : SLOT_FAILED;
        }
        p->cv.notify_all();
    }
}

int v2p_pipeline_wait(v2p_pipeline* p, uint32_t ticket, const uint8_t** result, uint64_t* n)
{
    if (!p || ticket >= p->slots.size() || !result || !n) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    PipeSlot& s = p->slots[ticket];
    {
        std::unique_lock<std::mutex> pl(p->pmu);
        if (!s.busy) { pl.unlock(); std::lock_guard<std::mutex> lk(c->mu); return c->fail(V2P_ERR_STATE, "nothing submitted on this ticket"); }
        p->cv.wait(pl, [&] { return s.state == SLOT_LAUNCHED || s.state == SLOT_FAILED; });       // (a stream slice: the runner has taken it through the one call)
        if (s.state == SLOT_FAILED) {
            const int rc = s.rc;
            pl.unlock();
            std::lock_guard<std::mutex> lk(c->mu);
            return c->fail(rc, s.err, s.err_index);
        }
    }
    // (the context is NOT held while the copies finish: submitters and the runner go on)
    (void)hipSetDevice(c->device);
    const hipError_t e = hipEventSynchronize(s.done);
    if (e != hipSuccess) { std::lock_guard<std::mutex> lk(c->mu); return c->hip_fail(e, "hipEventSynchronize"); }
    s.in_flight = false;
    unsigned long long st;
    memcpy(&st, s.h_out.p + s.o_status, sizeof st);
    *result = s.h_out.p;
    *n = s.out_bytes;
    if (st != STATUS_CLEAN) {
        const int code = reason_to_err(uint32_t(st & 0xFFu));
        std::lock_guard<std::mutex> lk(c->mu);
        return c->fail(code, std::string("device: ") + err_name(code) + " at descriptor " + std::to_string(st >> 8), int64_t(st >> 8));
    }
    return V2P_OK;
}

int v2p_pipeline_result_info(v2p_pipeline* p, uint32_t ticket, const uint64_t** hap_out_begin, uint64_t* n_haps, const uint64_t** digests, v2p_slice_times* times)
{
    if (!p || ticket >= p->slots.size()) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    PipeSlot& s = p->slots[ticket];
    std::lock_guard<std::mutex> pl(p->pmu);
    if (!s.busy || !s.is_stream || s.state != SLOT_LAUNCHED || s.in_flight) { std::lock_guard<std::mutex> lk(c->mu); return c->fail(V2P_ERR_STATE, "v2p_pipeline_result_info: a stream slice that has been waited for"); }
    if (hap_out_begin) *hap_out_begin = s.hap_out_begin.data();
    if (n_haps) *n_haps = s.n_haps;
    if (digests) *digests = (s.sflags & V2P_SUBMIT_DIGESTS) ? reinterpret_cast<const uint64_t*>(s.h_out.p + s.o_digests) : nullptr;
    if (times) { times->stage_ms = s.t_stage_ms; times->runner_ms = s.t_gpu_ms; }
    return V2P_OK;
}

int v2p_pipeline_release(v2p_pipeline* p, uint32_t ticket)
{
    if (!p || ticket >= p->slots.size()) return V2P_ERR_INVALID_ARG;
    v2p_ctx* c = p->ctx;
    PipeSlot& s = p->slots[ticket];
    {
        // released without a wait: a queued slice is taken through the one call first, then the copies into / out of the pinned buffers must end
        std::unique_lock<std::mutex> pl(p->pmu);
        if (s.busy && s.is_stream) p->cv.wait(pl, [&] { return s.state == SLOT_LAUNCHED || s.state == SLOT_FAILED; });
    }
    if (s.in_flight) {
        (void)hipSetDevice(c->device);
        hipError_t e = hipSuccess;
        if (s.state == SLOT_LAUNCHED) e = hipEventSynchronize(s.done);
        else { e = hipStreamSynchronize(s.stream); if (e == hipSuccess && s.d2h) e = hipStreamSynchronize(s.d2h); }
        if (e != hipSuccess) { std::lock_guard<std::mutex> lk(c->mu); return c->hip_fail(e, "hipEventSynchronize"); }
        s.in_flight = false;
    }
    std::lock_guard<std::mutex> pl(p->pmu);
    s.busy = false; s.state = SLOT_FREE;
    return V2P_OK;
}

// ---- raw launchers -------------------------------------------------------------

int v2p_stitch_launch_opts(void* hip_stream,
                           const uint64_t* d_desc, uint64_t n_desc, const v2p_chunk* d_chunks, uint32_t n_chunks,
                           const uint8_t* d_src0, uint64_t src0_len,
                           const uint8_t* d_src1, uint64_t src1_len,
                           uint8_t* d_out, uint64_t out_len,
                           uint64_t* d_status, const v2p_launch_opts* opts)
{
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) || !d_status || !opts) return V2P_ERR_INVALID_ARG;
    if (opts->routing & ~0xFFFu) return V2P_ERR_INVALID_ARG;      // (v2p_stitch_launch_bits() fills bits 1 .. 11)
    StitchArgs a{d_desc, n_desc, reinterpret_cast<const Chunk*>(d_chunks), n_chunks, d_src0, src0_len, d_src1, src1_len,
                 d_out, out_len, reinterpret_cast<unsigned long long*>(d_status)};
    a.opt_phase_bytes = opts->phase_bytes; a.opt_phase_min_chunks = opts->phase_min_chunks; a.opt_store_sc1 = opts->store_sc1;
#ifdef V2P_BENCH_VARIANTS
    // (the development library: `reserved` = 3 / 8 route every per-block chunk to the per-block / the dense kernel -- routing-only A/B switches)
    const int flags = int(opts->nontemporal ? 1u : 0u) | int(opts->routing & 0xFFEu) | int((opts->reserved == 3u || opts->reserved == 8u ? opts->reserved : 0u) << 12);
    if (opts->reserved != 0u && opts->reserved != 3u && opts->reserved != 8u) return V2P_ERR_INVALID_ARG;
#else
    const int flags = int(opts->nontemporal ? 1u : 0u) | int(opts->routing & 0xFFEu);
    if (opts->reserved != 0u) return V2P_ERR_INVALID_ARG;
#endif
    return launch_stitch(a, reinterpret_cast<hipStream_t>(hip_stream), flags, opts->max_blocks) == hipSuccess ? V2P_OK : V2P_ERR_HIP;
}

int v2p_set_launch_opts(v2p_ctx* c, const v2p_launch_opts* opts)
{
    if (!c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (opts && opts->reserved != 0u) return c->fail(V2P_ERR_INVALID_ARG, "v2p_launch_opts.reserved must be 0");
    if (opts) c->launch_opts = *opts; else c->launch_opts = v2p_launch_opts{1u, 0u, 0ull, 0u, -1, 0u, 0u};
    return V2P_OK;
}

#ifdef V2P_BENCH_VARIANTS
// libv2p_bench.so only (csrc/bench/v2p_bench.h): the A/B switches 16 .. 29 of the builders and launchers
int v2p_bench_set_variant(v2p_ctx* c, uint32_t variant)
{
    if (!c) return V2P_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->variant = variant;
    return V2P_OK;
}
#endif

#ifdef V2P_BENCH_VARIANTS
// libv2p_bench.so only (csrc/bench/v2p_bench.h): the launcher with the packed flag word -- kernel variants, timing-only ablations,
// idle LDS, waves per workgroup -- and the A/B switches read from the environment at every call
int v2p_stitch_launch(void* hip_stream,
                      const uint64_t* d_desc, uint64_t n_desc, const v2p_chunk* d_chunks, uint32_t n_chunks,
                      const uint8_t* d_src0, uint64_t src0_len,
                      const uint8_t* d_src1, uint64_t src1_len,
                      uint8_t* d_out, uint64_t out_len,
                      uint64_t* d_status, int nontemporal, uint32_t max_blocks)
{
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) || !d_status) return V2P_ERR_INVALID_ARG;
    StitchArgs a{d_desc, n_desc, reinterpret_cast<const Chunk*>(d_chunks), n_chunks, d_src0, src0_len, d_src1, src1_len,
                 d_out, out_len, reinterpret_cast<unsigned long long*>(d_status)};
    if (const char* e = getenv("V2P_PHASE_BYTES")) { const uint64_t v = strtoull(e, nullptr, 10); a.opt_phase_bytes = v ? v : ~0ull; }   // 0: one phase, no touch
    if (const char* e = getenv("V2P_WAVE_SC1")) a.opt_store_sc1 = atoi(e) != 0;
    if (const char* e = getenv("V2P_PHASE_MIN_CHUNKS")) a.opt_phase_min_chunks = uint32_t(strtoul(e, nullptr, 10));
    a.opt_touch = (getenv("V2P_PHASE_NO_TOUCH") ? 1u : 0u) | (getenv("V2P_PHASE_OWN_TOUCH") ? 2u : 0u) | (getenv("V2P_PHASE_ONE_LAUNCH") ? 4u : 0u);
    return launch_stitch(a, reinterpret_cast<hipStream_t>(hip_stream), nontemporal, max_blocks) == hipSuccess ? V2P_OK : V2P_ERR_HIP;
}
#endif

int v2p_digest_launch(void* hip_stream, const uint8_t* d_out, const uint64_t* d_hap_begin, uint64_t n_haps,
                      uint64_t out_bytes, uint64_t* d_digests)
{
    DigestArgs a{d_out, d_hap_begin, n_haps, d_digests};
    return launch_digest(a, out_bytes, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? V2P_OK : V2P_ERR_HIP;
}

int v2p_stitch_launch_bits(const v2p_chunk* chunks, uint64_t n_chunks)
{
    if (n_chunks && !chunks) return V2P_ERR_INVALID_ARG;
    return stitch_launch_bits(reinterpret_cast<const Chunk*>(chunks), n_chunks);
}

int v2p_routing_rules(uint64_t n_desc, uint64_t n_chunks, uint64_t result_bytes, uint64_t proteome_len, int wave_image, v2p_routing* out)
{
    if (!out) return V2P_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    const bool rich = image_is_rich(n_desc, result_bytes);
    out->wave_bytes_per_task = WAVE_BYTES_PER_TASK;
    out->rich = rich ? 1u : 0u;
    out->phased = (wave_image && n_chunks >= PHASE_MIN_CHUNKS) ? 1u : 0u;       // (launch_stitch: dense and per-block images are one launch)
    out->store_sc1 = rich ? 0u : 1u;
    out->phase_bytes = rich ? PHASE_BYTES_RICH : PHASE_BYTES_DEFAULT;
    out->order_blocks = (n_chunks < 16 || proteome_len == 0) ? 1u : xcd_order_blocks(result_bytes, proteome_len, n_chunks, XCD_ORDER_MAX_BLOCKS, n_desc);
    out->order_windows = xcd_order_window_major(result_bytes, n_desc) ? 1u : 0u;
    return V2P_OK;
}

int v2p_order_chunks_for_xcds(v2p_chunk* chunks, uint64_t n_chunks, const uint64_t* desc, uint64_t n_desc, uint64_t proteome_len)
{
    if ((n_chunks && !chunks) || (n_desc && !desc)) return V2P_ERR_INVALID_ARG;
#ifdef V2P_BENCH_VARIANTS
    const char* e = getenv("V2P_ORDER_WINDOWS");                   // (experiments: 0 = haplotype-major inside a slice)
    const char* eb = getenv("V2P_ORDER_MAX_BLOCKS");               // (experiments: 1 = one order for the whole table)
    order_chunks_for_xcds(reinterpret_cast<Chunk*>(chunks), n_chunks, desc, n_desc, proteome_len, 8, !(e && e[0] == '0'),
                          eb ? uint32_t(strtoul(eb, nullptr, 10)) : XCD_ORDER_MAX_BLOCKS);
#else
    order_chunks_for_xcds(reinterpret_cast<Chunk*>(chunks), n_chunks, desc, n_desc, proteome_len);
#endif
    return V2P_OK;
}

}  // extern "C"
