// ppgg_gpu.hpp -- C++ mirror of the reference's step-6 host interface, gpu arm implemented.
//
// The reference host is Rust (crate `ppgg`); this image has no Rust toolchain, so the host
// side above the C ABI is written in C++ with the reference's names, argument meaning and
// error behaviour (paths under /root/reference/src/data_structures/InternalRep):
//   Engine, Engine::from_str      engines.rs:15-29
//   Task                          task.rs:2-18
//   GIR, GIR::execute(engine)     gir.rs:15-46, 197-241   (Engine::GPU arm = v2p_execute_gir)
//   execute(probands, engine)     parts/exec.rs:23-42     (thread pool over probands, one
//                                 engine context per worker, as Rayon workers would hold)
// A reference panic!() is a C++ exception here (`Panic`); the Rust shim in INTEGRATION.md
// turns the same status codes back into panic!().
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/vcf2prot_hip.h"

namespace ppgg {

struct Panic : std::runtime_error {
    int code; int64_t index;
    Panic(int c, const std::string& m, int64_t i = -1) : std::runtime_error(m), code(c), index(i) {}
};

enum class Engine { ST = V2P_ENGINE_ST, MT = V2P_ENGINE_MT, GPU = V2P_ENGINE_GPU };

inline Engine engine_from_str(const std::string& name)          // engines.rs:17-29
{
    int e = 0;
    if (v2p_engine_from_str(name.c_str(), &e) != V2P_OK) throw std::invalid_argument(name + " is not a supported engine");
    return static_cast<Engine>(e);
}

struct Task {                                                    // task.rs:2-9
    uint8_t exe_code; uint64_t start_pos, length, start_pos_res;
    Task(uint8_t c, uint64_t s, uint64_t l, uint64_t r) : exe_code(c), start_pos(s), length(l), start_pos_res(r) {}
};

// One engine context per worker thread (the C ABI's threading contract).
class GpuContext {
public:
    explicit GpuContext(int device = 0, bool debug_gpu = std::getenv("DEBUG_GPU") != nullptr)   // README.md:156-157
    {
        if (v2p_init(device, debug_gpu ? V2P_FLAG_DEBUG_GPU : 0u, &ctx_) != V2P_OK)
            throw Panic(V2P_ERR_HIP, v2p_last_error(nullptr));
    }
    ~GpuContext() { v2p_destroy(ctx_); }
    GpuContext(const GpuContext&) = delete;
    GpuContext& operator=(const GpuContext&) = delete;
    v2p_ctx* raw() const { return ctx_; }
private:
    v2p_ctx* ctx_ = nullptr;
};

using Annotation = std::map<std::string, std::pair<uint64_t, uint64_t>>;

class GIR {                                                      // gir.rs:15-23
public:
    GIR(std::vector<Task> g_rep, Annotation annotation, std::u32string alt_stream, std::u32string ref_stream,
        std::u32string res_array)
        : g_rep_(std::move(g_rep)), annotation_(std::move(annotation)), alt_(std::move(alt_stream)),
          ref_(std::move(ref_stream)), res_(std::move(res_array)) {}

    const std::vector<Task>& get_tasks() const { return g_rep_; }
    const Annotation& get_annotation() const { return annotation_; }
    uint64_t get_results_max() const                              // gir.rs:156-167
    {
        uint64_t m = 0;
        for (const auto& kv : annotation_) m = kv.second.second > m ? kv.second.second : m;
        return m;
    }

    // gir.rs:197-241: consumes the representation, returns (res_array, annotation).
    // Only the GPU arm lives here; ST/MT are the reference's own CPU code.
    std::pair<std::u32string, Annotation> execute(Engine engine, GpuContext& ctx) &&
    {
        if (engine != Engine::GPU) throw std::logic_error("the st/mt engines are the reference's CPU code");
        const size_t n = g_rep_.size();
        std::vector<uint8_t> code(n);
        std::vector<uint64_t> sp(n), ln(n), sr(n);
        for (size_t i = 0; i < n; ++i) {                          // gir.rs:283-299
            code[i] = g_rep_[i].exe_code; sp[i] = g_rep_[i].start_pos; ln[i] = g_rep_[i].length; sr[i] = g_rep_[i].start_pos_res;
        }
        const int rc = v2p_execute_gir(ctx.raw(), code.data(), sp.data(), ln.data(), sr.data(), n,
                                       reinterpret_cast<const uint32_t*>(ref_.data()), ref_.size(),
                                       reinterpret_cast<const uint32_t*>(alt_.data()), alt_.size(),
                                       reinterpret_cast<uint32_t*>(&res_[0]), res_.size());
        if (rc != V2P_OK) throw Panic(rc, v2p_last_error(ctx.raw()), v2p_last_error_index(ctx.raw()));
        return {std::move(res_), std::move(annotation_)};
    }

    // The same arm when the workers share one context (the reference's own shape: every Rayon worker calls GIR::execute,
    // parts/exec.rs:36-39): v2p_execute_gir_shared coalesces the concurrent calls; exec codes go as the marshaller has them (u64).
    std::pair<std::u32string, Annotation> execute_shared(Engine engine, GpuContext& ctx) &&
    {
        if (engine != Engine::GPU) throw std::logic_error("the st/mt engines are the reference's CPU code");
        const size_t n = g_rep_.size();
        std::vector<uint64_t> code(n), sp(n), ln(n), sr(n);
        for (size_t i = 0; i < n; ++i) {                          // gir.rs:283-299
            code[i] = g_rep_[i].exe_code; sp[i] = g_rep_[i].start_pos; ln[i] = g_rep_[i].length; sr[i] = g_rep_[i].start_pos_res;
        }
        int64_t row = -1;
        const int rc = v2p_execute_gir_shared(ctx.raw(), code.data(), sp.data(), ln.data(), sr.data(), n,
                                              reinterpret_cast<const uint32_t*>(ref_.data()), ref_.size(),
                                              reinterpret_cast<const uint32_t*>(alt_.data()), alt_.size(),
                                              reinterpret_cast<uint32_t*>(&res_[0]), res_.size(), &row);
        if (rc != V2P_OK) throw Panic(rc, v2p_last_error(ctx.raw()), row);
        return {std::move(res_), std::move(annotation_)};
    }

    // ... and in two halves (v2p_gir_submit / v2p_gir_collect): the worker that calls GIR::execute for haplotype 1 and then for
    // haplotype 2 (personalized_genome.rs:64-65) submits the first, packs and submits the second while the first batch is on the
    // GPU, then collects both.  The GIR owns the arrays the engine reads until collect() has returned.
    // returns false when every batch of the queue is in flight (V2P_BUSY): collect an earlier GIR, then submit again
    bool submit(Engine engine, GpuContext& ctx)
    {
        if (engine != Engine::GPU) throw std::logic_error("the st/mt engines are the reference's CPU code");
        const size_t n = g_rep_.size();
        soa_.resize(4 * n);
        uint64_t* code = soa_.data(), *sp = code + n, *ln = sp + n, *sr = ln + n;
        for (size_t i = 0; i < n; ++i) { code[i] = g_rep_[i].exe_code; sp[i] = g_rep_[i].start_pos; ln[i] = g_rep_[i].length; sr[i] = g_rep_[i].start_pos_res; }
        const int rc = v2p_gir_submit(ctx.raw(), code, sp, ln, sr, n, reinterpret_cast<const uint32_t*>(ref_.data()), ref_.size(),
                                      reinterpret_cast<const uint32_t*>(alt_.data()), alt_.size(), reinterpret_cast<uint32_t*>(&res_[0]), res_.size(), &ticket_);
        if (rc == V2P_BUSY) return false;
        if (rc != V2P_OK) throw Panic(rc, v2p_last_error(ctx.raw()), v2p_last_error_index(ctx.raw()));
        return true;
    }
    std::pair<std::u32string, Annotation> collect(GpuContext& ctx) &&
    {
        int64_t row = -1;
        const int rc = v2p_gir_collect(ctx.raw(), ticket_, &row);
        ticket_ = nullptr;
        if (rc != V2P_OK) throw Panic(rc, v2p_last_error(ctx.raw()), row);
        return {std::move(res_), std::move(annotation_)};
    }

private:
    std::vector<Task> g_rep_;
    Annotation annotation_;
    std::u32string alt_, ref_, res_;
    std::vector<uint64_t> soa_;                                   // the marshalled Task vector of a submitted GIR
    v2p_gir_ticket* ticket_ = nullptr;
};

// parts/exec.rs:23-42 for Engine::GPU: jobs (haplotype GIRs) are pulled by a pool of workers,
// each worker owning one engine context, the way Rayon workers would each hold one.
template <class MakeGir, class Consume>
void execute(uint64_t n_jobs, int n_threads, int device, MakeGir make_gir, Consume consume)
{
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> pool;
    std::vector<std::string> errors(size_t(n_threads > 0 ? n_threads : 1));
    for (int t = 0; t < (n_threads > 0 ? n_threads : 1); ++t)
        pool.emplace_back([&, t] {
            try {
                GpuContext ctx(device);
                for (uint64_t j = next++; j < n_jobs; j = next++) {
                    auto out = make_gir(j).execute(Engine::GPU, ctx);
                    consume(j, std::move(out.first), std::move(out.second));
                }
            } catch (const std::exception& e) { errors[size_t(t)] = e.what(); }
        });
    for (auto& th : pool) th.join();
    for (const auto& e : errors) if (!e.empty()) throw Panic(V2P_ERR_HIP, e);
}

// ---- N devices in ONE process -------------------------------------------------------------------------------------------------
// The reference's host is one process with a Rayon pool (parts/exec.rs:34-40); its natural shape on a node of GPUs is one engine
// context per device inside that process -- no launcher, no collective: haplotypes are independent units
// (haplotype_instruction.rs:94-133), the proteome is replicated, every device builds and executes a contiguous range of
// haplotypes and the cohort-wide offsets are a prefix sum the host already knows.

// contiguous haplotype ranges of equal result bytes -- the prefix sum cut at its k / world quantiles (the rule of
// vcf2prot_amd/shard.py::shard_by_bytes, in exact integer arithmetic: haplotype h goes to the left of cut k while the middle of
// its bytes lies before total * k / world)
inline std::vector<std::pair<uint64_t, uint64_t>> shard_by_bytes(const std::vector<uint64_t>& result_bytes, int world)
{
    if (world < 1) throw std::invalid_argument("world < 1");
    unsigned __int128 total = 0;
    for (uint64_t b : result_bytes) total += b;
    std::vector<uint64_t> cuts{0};
    unsigned __int128 acc = 0;
    uint64_t h = 0;
    const uint64_t n = result_bytes.size();
    for (int k = 1; k < world; ++k) {
        while (h < n && (2 * acc + result_bytes[h]) * unsigned(world) < 2 * total * unsigned(k)) { acc += result_bytes[h]; ++h; }
        cuts.push_back(h);
    }
    cuts.push_back(n);
    std::vector<std::pair<uint64_t, uint64_t>> out;
    for (int r = 0; r < world; ++r) out.emplace_back(cuts[size_t(r)], cuts[size_t(r) + 1]);
    return out;
}

struct DeviceShard {
    int rank = 0, device = 0;
    uint64_t h0 = 0, h1 = 0;                   // haplotypes [h0, h1) of the cohort
    uint64_t byte_offset = 0, bytes = 0;       // the shard's arena inside the cohort-wide result (prefix sum over the shards before it)
    double seconds = 0;                        // v2p_batch_build_and_execute + sync on this device (the stream already resident)
    float oneshot_ms = 0;                      // ... by HIP events
};

// One worker thread per entry of `devices` (a device may appear more than once: its shards then share it), each with its own
// context: proteome upload, make_stream(h0, h1) -> the range's transcript stream (a shared_ptr that keeps its arrays alive),
// v2p_stream_upload, ONE v2p_batch_build_and_execute, sync, then consume(shard, ctx, batch) on the worker thread (digests,
// download, file writes).  Throws Panic with the first shard's error.
template <class MakeStream, class Consume>
std::vector<DeviceShard> execute_sharded(const std::vector<uint64_t>& hap_bytes, const std::vector<int>& devices,
                                         const uint8_t* proteome, uint64_t proteome_len, MakeStream make_stream, Consume consume)
{
    const int world = int(devices.size());
    const auto ranges = shard_by_bytes(hap_bytes, world);
    std::vector<DeviceShard> shards;
    shards.resize(size_t(world));
    uint64_t off = 0;
    for (int r = 0; r < world; ++r) {
        DeviceShard& s = shards[size_t(r)];
        s.rank = r; s.device = devices[size_t(r)]; s.h0 = ranges[size_t(r)].first; s.h1 = ranges[size_t(r)].second; s.byte_offset = off;
        for (uint64_t h = s.h0; h < s.h1; ++h) s.bytes += hap_bytes[h];
        off += s.bytes;
    }
    std::vector<std::string> errors;
    errors.resize(size_t(world));
    std::vector<int> codes;
    codes.resize(size_t(world), 0);
    std::vector<std::thread> pool;
    for (int r = 0; r < world; ++r)
        pool.emplace_back([&, r] {
            DeviceShard& s = shards[size_t(r)];
            try {
                GpuContext ctx(s.device);
                auto fail = [&](int rc) { throw Panic(rc, v2p_last_error(ctx.raw()), v2p_last_error_index(ctx.raw())); };
                int rc = v2p_upload_proteome(ctx.raw(), proteome, proteome_len);
                if (rc != V2P_OK) fail(rc);
                std::shared_ptr<const v2p_txstream> host = make_stream(s.h0, s.h1);
                v2p_stream* rs = nullptr;
                if ((rc = v2p_stream_upload(ctx.raw(), host.get(), &rs)) != V2P_OK) fail(rc);
                host.reset();
                v2p_batch* b = nullptr;
                if ((rc = v2p_batch_create(ctx.raw(), &b)) != V2P_OK) { v2p_stream_destroy(rs); fail(rc); }
                const auto t0 = std::chrono::steady_clock::now();
                rc = v2p_batch_build_and_execute(b, rs, 0, 0);
                if (rc == V2P_OK) rc = v2p_batch_sync(b);
                s.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rc != V2P_OK) { const std::string m = v2p_last_error(ctx.raw()); const int64_t i = v2p_last_error_index(ctx.raw()); v2p_batch_destroy(b); v2p_stream_destroy(rs); throw Panic(rc, m, i); }
                v2p_oneshot_info info;
                if (v2p_batch_oneshot_info(b, &info) == V2P_OK) s.oneshot_ms = info.total_ms;
                uint64_t out_bytes = 0;
                v2p_batch_counts(b, nullptr, nullptr, nullptr, &out_bytes, nullptr);
                if (out_bytes != s.bytes) { v2p_batch_destroy(b); v2p_stream_destroy(rs); throw Panic(V2P_ERR_STATE, "a shard's arena is not the sum of its haplotypes' result sizes"); }
                consume(s, ctx.raw(), b);
                v2p_batch_destroy(b);
                v2p_stream_destroy(rs);
            } catch (const Panic& p) { errors[size_t(r)] = p.what(); codes[size_t(r)] = p.code; }
            catch (const std::exception& e) { errors[size_t(r)] = e.what(); codes[size_t(r)] = V2P_ERR_HIP; }
        });
    for (auto& th : pool) th.join();
    for (int r = 0; r < world; ++r) if (!errors[size_t(r)].empty()) throw Panic(codes[size_t(r)], "shard " + std::to_string(r) + ": " + errors[size_t(r)]);
    return shards;
}

// ---- ... and with the results returning to the host: one STREAM PIPELINE per device --------------------------------------------------
// What parts/exec.rs:23-42 hands its caller is host memory (Vec<PersonalizedGenome>; the writer is next, personalized_genome.rs:90-113).
// Each device's worker cuts its shard into slices of about `slice_bytes` of result and feeds them through v2p_pipeline_submit_stream:
// the slice's Task vectors are checked and staged on this worker's thread, uploaded, built + executed by the one call, and the arena comes
// back into pinned host memory while the next slices are on their way -- eight devices are eight PCIe links.  consume(shard, h0, h1,
// bytes, n, hap_out_begin, digests) runs on the worker thread with the slice's result (valid until it returns), slices in order.
inline std::vector<std::pair<uint64_t, uint64_t>> cut_by_bytes(const std::vector<uint64_t>& hap_bytes, uint64_t h0, uint64_t h1, uint64_t budget)
{
    std::vector<std::pair<uint64_t, uint64_t>> cuts;          // (vcf2prot_amd/driver.py::cut_by_bytes)
    uint64_t begin = h0, acc = 0;
    for (uint64_t h = h0; h < h1; ++h) {
        if (h > begin && acc + hap_bytes[h] > budget) { cuts.emplace_back(begin, h); begin = h; acc = 0; }
        acc += hap_bytes[h];
    }
    if (h1 > begin) cuts.emplace_back(begin, h1);
    return cuts;
}

template <class MakeStream, class ConsumeSlice>
std::vector<DeviceShard> execute_streamed(const std::vector<uint64_t>& hap_bytes, const std::vector<int>& devices,
                                          const uint8_t* proteome, uint64_t proteome_len, const uint8_t* record_headers, uint64_t n_headers,
                                          MakeStream make_stream, ConsumeSlice consume, uint64_t slice_bytes = 1152ull << 20, uint32_t slots = 4,
                                          uint32_t copy_threads = 8)
{
    const int world = int(devices.size());
    const auto ranges = shard_by_bytes(hap_bytes, world);
    std::vector<DeviceShard> shards;
    shards.resize(size_t(world));
    uint64_t off = 0;
    for (int r = 0; r < world; ++r) {
        DeviceShard& s = shards[size_t(r)];
        s.rank = r; s.device = devices[size_t(r)]; s.h0 = ranges[size_t(r)].first; s.h1 = ranges[size_t(r)].second; s.byte_offset = off;
        for (uint64_t h = s.h0; h < s.h1; ++h) s.bytes += hap_bytes[h];
        off += s.bytes;
    }
    std::vector<std::string> errors;
    errors.resize(size_t(world));
    std::vector<int> codes;
    codes.resize(size_t(world), 0);
    std::vector<std::thread> pool;
    for (int r = 0; r < world; ++r)
        pool.emplace_back([&, r] {
            DeviceShard& s = shards[size_t(r)];
            v2p_pipeline* pipe = nullptr;
            try {
                GpuContext ctx(s.device);
                auto ck = [&](int rc) { if (rc != V2P_OK) throw Panic(rc, v2p_last_error(ctx.raw()), v2p_last_error_index(ctx.raw())); };
                ck(v2p_upload_reference(ctx.raw(), proteome, proteome_len, record_headers, n_headers));
                ck(v2p_pipeline_create(ctx.raw(), slots, &pipe));
                ck(v2p_pipeline_reserve(pipe, 0, 0, copy_threads));
                const auto slices = cut_by_bytes(hap_bytes, s.h0, s.h1, slice_bytes);
                struct InFlight { uint32_t ticket; uint64_t h0, h1; };
                std::vector<InFlight> inflight;
                const auto t0 = std::chrono::steady_clock::now();
                auto finish = [&](const InFlight& f) {
                    const uint8_t* bytes = nullptr; uint64_t n = 0, nh = 0;
                    const uint64_t* hob = nullptr; const uint64_t* dig = nullptr;
                    ck(v2p_pipeline_wait(pipe, f.ticket, &bytes, &n));
                    ck(v2p_pipeline_result_info(pipe, f.ticket, &hob, &nh, &dig, nullptr));
                    if (nh != f.h1 - f.h0) throw Panic(V2P_ERR_STATE, "a slice came back with another number of haplotypes");
                    consume(s, f.h0, f.h1, bytes, n, hob, dig);
                    ck(v2p_pipeline_release(pipe, f.ticket));
                };
                for (const auto& sl : slices) {
                    if (inflight.size() == slots) { finish(inflight.front()); inflight.erase(inflight.begin()); }
                    std::shared_ptr<const v2p_txstream> host = make_stream(sl.first, sl.second);
                    uint32_t t = 0;
                    ck(v2p_pipeline_submit_stream(pipe, host.get(), 0, V2P_SUBMIT_DIGESTS, &t));       // (staged on return: the host copy goes)
                    inflight.push_back(InFlight{t, sl.first, sl.second});
                }
                for (const InFlight& f : inflight) finish(f);
                s.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                v2p_pipeline_destroy(pipe);
                pipe = nullptr;
            } catch (const Panic& p) { errors[size_t(r)] = p.what(); codes[size_t(r)] = p.code; if (pipe) v2p_pipeline_destroy(pipe); }
            catch (const std::exception& e) { errors[size_t(r)] = e.what(); codes[size_t(r)] = V2P_ERR_HIP; if (pipe) v2p_pipeline_destroy(pipe); }
        });
    for (auto& th : pool) th.join();
    for (int r = 0; r < world; ++r) if (!errors[size_t(r)].empty()) throw Panic(codes[size_t(r)], "shard " + std::to_string(r) + ": " + errors[size_t(r)]);
    return shards;
}

}  // namespace ppgg
