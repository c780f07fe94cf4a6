// v2p_harness.cpp -- C++ host harness that plays the role of the reference's exec::execute
// (parts/exec.rs:23-42) on top of the C ABI: it builds haplotype GIRs exactly as steps 4-5
// would hand them over (synthetic cohorts, include/v2p_cohort.h), runs every one through
// GIR::execute(Engine::GPU) from a pool of worker threads, and reports wall-clock throughput
// of this GIR-faithful mode (tapes cross PCIe as Rust chars, 4 bytes per residue, both ways).
//
//   v2p_harness kat                          reference known-answer tests through the mirror
//   v2p_harness run <preset> <haps> <threads>   e.g. run C2 64 8
//   v2p_harness vcf <in.vcf> <reference.fasta> <outdir> [--no-test] [-a] [-c] [--slice-kb K]   VCF -> one FASTA(.gz) per proband, no Rust anywhere;
//                                            steps 4-5 produce slices of probands that stream through v2p_pipeline_submit_stream while the next are made
//   v2p_harness sharded <preset> <samples> --devices N [--oversubscribe] [--threads T] [--streamed [--slice-mb M]]
//                                            the cohort over N devices in THIS process (ppgg::execute_sharded): N contexts, N worker
//                                            threads, ranges of equal result bytes, one v2p_batch_build_and_execute each;
//                                            --streamed (ppgg::execute_streamed): one v2p_pipeline per device, the shard's Task vectors in
//                                            slices of M MiB of result through v2p_pipeline_submit_stream, the results IN HOST MEMORY --
//                                            the digests printed are then digests of the host bytes
//   v2p_harness shard <world> <bytes...>     the ranges ppgg::shard_by_bytes cuts (no GPU)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "../../../include/v2p_cohort.h"
#include "../../../include/v2p_step4a.h"
#include "ppgg_gpu.hpp"

using namespace ppgg;

static std::u32string u32(const char* s)
{
    std::u32string o;
    for (; *s; ++s) o.push_back(char32_t(static_cast<unsigned char>(*s)));
    return o;
}

static std::string narrow(const std::u32string& s)
{
    std::string o;
    for (char32_t c : s) o.push_back(char(c));
    return o;
}

static int kat()
{
    GpuContext ctx(0);
    int fails = 0;
    {   // task.rs:118-144 (descending result offsets: ordered path, untouched cells keep 'x')
        GIR g({Task(0, 1, 1, 8), Task(0, 4, 1, 4), Task(0, 6, 2, 6)}, {}, u32("HGFEFCBA"), u32("ABCFEFGH"), u32("xxxxxxxxxx"));
        auto r = std::move(g).execute(engine_from_str("gpu"), ctx);
        if (narrow(r.first) != "xxxxExGHBx") { std::printf("FAIL task.rs test_execute: %s\n", narrow(r.first).c_str()); ++fails; }
    }
    {   // gir.rs:172-196
        Annotation a; a["Seq_1"] = {0, 5};
        GIR g({Task(0, 0, 4, 0), Task(1, 0, 1, 4)}, a, u32("G"), u32("TEST"), u32("....."));
        auto r = std::move(g).execute(Engine::GPU, ctx);
        if (narrow(r.first) != "TESTG" || r.second.at("Seq_1").second != 5) { std::printf("FAIL gir.rs doc example\n"); ++fails; }
    }
    {   // out-of-bounds source: the reference panics (task.rs:43); nothing may be written
        GIR g({Task(0, 3, 5, 0)}, {}, u32("XY"), u32("ABCDE"), u32("........"));
        try { (void)std::move(g).execute(Engine::GPU, ctx); std::printf("FAIL: out-of-bounds task did not raise\n"); ++fails; }
        catch (const Panic& p) { if (p.code != V2P_ERR_SRC_OOB || p.index != 0) { std::printf("FAIL: wrong panic %d\n", p.code); ++fails; } }
    }
    try { (void)engine_from_str("cuda"); std::printf("FAIL: engine_from_str accepted 'cuda'\n"); ++fails; }
    catch (const std::invalid_argument&) {}
    std::printf(fails ? "kat: %d failure(s)\n" : "kat: ok\n", fails);
    return fails ? 1 : 0;
}

static uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static int run(const char* preset, uint64_t n_haps, int threads, bool shared, bool async)
{
    v2p_cohort_params p;
    if (v2p_cohort_preset(preset, &p)) { std::fprintf(stderr, "unknown preset %s\n", preset); return 2; }
    v2p_cohort* c = nullptr;
    if (v2p_cohort_create(&p, &c)) return 2;
    if (n_haps > v2p_cohort_n_haplotypes(c)) n_haps = v2p_cohort_n_haplotypes(c);
    std::vector<uint64_t> digest(n_haps, 0);
    std::atomic<uint64_t> aa{0};
    // per-thread generator buffers: make_gir runs on the worker that executes the job
    auto make_gir = [&](uint64_t h) {
        thread_local v2p_hapbuf* hb = v2p_hapbuf_create();
        v2p_hap_view v;
        v2p_cohort_generate(c, h, hb, &v);
        std::vector<Task> tasks;
        tasks.reserve(v.n_tasks);
        for (uint64_t i = 0; i < v.n_tasks; ++i) tasks.emplace_back(v.code[i], v.start_pos[i], v.length[i], v.start_pos_res[i]);
        std::u32string ref(v.n_ref, U'\0'), alt(v.n_alt, U'\0');
        v2p_cohort_ref_tape_u32(c, &v, reinterpret_cast<uint32_t*>(&ref[0]));          // what step 5 builds
        for (uint64_t i = 0; i < v.n_alt; ++i) alt[i] = v.alt[i];
        Annotation ann;
        for (uint64_t t = 0; t < v.n_tx; ++t) {
            char name[32];
            std::snprintf(name, sizeof name, "ENST%011u", v.tx_id[t]);
            ann[name] = {v.tx_res_begin[t], v.tx_res_end[t]};
        }
        aa += v.n_res;
        return GIR(std::move(tasks), std::move(ann), std::move(alt), std::move(ref), std::u32string(v.n_res, U'.'));   // haplotype_instruction.rs:78
    };
    auto consume = [&](uint64_t h, std::u32string res, Annotation) {
        uint64_t s = 0;
        for (size_t i = 0; i < res.size(); ++i) s += ((uint64_t(res[i] & 0xFFu) + 1ull) << (8 * (i & 7))) * mix64(i >> 3);   // vcf2prot_hip.h: v2p_batch_digests
        digest[h] = s;
    };
    // the GIRs are built first (steps 4-5 are not the engine's), then every GIR::execute(Engine::GPU) is timed: wall clock of the
    // worker pool from its first call to its last return -- the quantity the CPU baseline (oracle, same Task boundary) reports
    std::vector<std::unique_ptr<GIR>> girs(n_haps);
    {
        std::atomic<uint64_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back([&] { for (uint64_t h = next++; h < n_haps; h = next++) girs[h].reset(new GIR(make_gir(h))); });
        for (auto& th : pool) th.join();
    }
    const uint64_t aa_timed = aa.load();                          // residues of the prebuilt GIRs (warm-up calls below are not counted)
    std::vector<std::u32string> results(n_haps);
    std::vector<std::string> errors{size_t(threads), std::string()};
    std::atomic<uint64_t> next{0};
    std::atomic<int> ready{0};
    std::chrono::steady_clock::time_point t0;
    {
        // a worker = one engine context for its whole life, as a Rayon worker would hold it; its first call (stream, pinned and
        // device buffers) is a warm-up outside the clock
        // (--shared: ONE context for all workers, the reference's own shape -- v2p_execute_gir_shared coalesces their calls)
        std::unique_ptr<GpuContext> one(shared ? new GpuContext(0) : nullptr);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                try {
                    if (shared) {
                        { GIR warm = make_gir(uint64_t(t) % n_haps); (void)std::move(warm).execute_shared(Engine::GPU, *one); }
                        if (++ready == threads) t0 = std::chrono::steady_clock::now();
                        while (ready.load() < threads) std::this_thread::yield();
                        if (async) {
                            // two haplotypes per worker in flight, as a sample's two are (personalized_genome.rs:64-65): the second is
                            // marshalled and staged while the first one's batch is on the GPU
                            uint64_t prev = ~0ull;
                            for (uint64_t h = next++; h < n_haps; h = next++) {
                                if (!girs[h]->submit(Engine::GPU, *one)) {            // every batch in flight: take the first haplotype's result, then go on
                                    if (prev != ~0ull) { results[prev] = std::move(*girs[prev]).collect(*one).first; prev = ~0ull; }
                                    while (!girs[h]->submit(Engine::GPU, *one)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (a batch is on the GPU: nothing to gain from asking faster)
                                }
                                if (prev != ~0ull) results[prev] = std::move(*girs[prev]).collect(*one).first;
                                prev = h;
                            }
                            if (prev != ~0ull) results[prev] = std::move(*girs[prev]).collect(*one).first;
                            return;
                        }
                        for (uint64_t h = next++; h < n_haps; h = next++) {
                            auto out = std::move(*girs[h]).execute_shared(Engine::GPU, *one);
                            results[h] = std::move(out.first);
                        }
                        return;
                    }
                    GpuContext ctx(0);
                    { GIR warm = make_gir(uint64_t(t) % n_haps); (void)std::move(warm).execute(Engine::GPU, ctx); }
                    if (++ready == threads) t0 = std::chrono::steady_clock::now();
                    while (ready.load() < threads) std::this_thread::yield();
                    for (uint64_t h = next++; h < n_haps; h = next++) {
                        auto out = std::move(*girs[h]).execute(Engine::GPU, ctx);
                        results[h] = std::move(out.first);
                    }
                } catch (const std::exception& e) { errors[size_t(t)] = e.what(); ready = threads; }
            });
        for (auto& th : pool) th.join();
    }
    for (const auto& e : errors) if (!e.empty()) { std::fprintf(stderr, "engine error: %s\n", e.c_str()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint64_t h = 0; h < n_haps; ++h) consume(h, std::move(results[h]), Annotation());
    std::printf("{\"mode\": \"gir-faithful: GIR::execute(Engine::GPU) on prebuilt GIRs (Rust chars in and out), %d worker threads, %s\", \"preset\": \"%s\", "
                "\"haplotypes\": %llu, \"aa\": %llu, \"seconds\": %.6f, \"aa_per_s\": %.4e, \"digests\": [",
                threads, shared ? (async ? "ONE shared ctx, calls coalesced, two GIRs per worker in flight (v2p_gir_submit / v2p_gir_collect)" : "ONE shared ctx, calls coalesced (v2p_execute_gir_shared)") : "one ctx each", preset,
                (unsigned long long)n_haps, (unsigned long long)aa_timed, secs, double(aa_timed) / secs);
    for (uint64_t h = 0; h < n_haps; ++h) std::printf("%s%llu", h ? ", " : "", (unsigned long long)digest[h]);
    std::printf("]}\n");
    v2p_cohort_destroy(c);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// vcf <in.vcf> <reference.fasta> <outdir> [--no-test]: the whole of `vcf2prot -f .. -r .. -o .. -g gpu` (main.rs:10-61)
// above the C ABI: record index, GPU bitmask decode, grouping (v2p_frontend.h), steps 4a / 4b (v2p_step4a.h, v2p_step4b.h),
// step 5 in the image builder, step 6 + FASTA emit on the GPU, one <proband>.fasta per proband (personalized_genome.rs:72-117).
static std::string slurp(const char* path)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("could not read ") + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::string s(size_t(n < 0 ? 0 : n), '\0');
    const size_t got = s.empty() ? 0 : std::fread(&s[0], 1, s.size(), f);
    std::fclose(f);
    if (got != s.size()) throw std::runtime_error(std::string("short read on ") + path);
    return s;
}

// readers.rs:37-76
static std::map<std::string, std::string> read_fasta(const std::string& text)
{
    std::map<std::string, std::string> rec;
    std::string header, seq;
    bool started = false;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(pos, e - pos);
        pos = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (header.empty() && !started) header = line.substr(1);
            else { rec[header] = seq; header = line.substr(1); seq.clear(); }
            started = true;
        } else {
            seq += line;
        }
    }
    rec[header] = seq;
    return rec;
}

static int vcf_mode(const char* vcf_path, const char* fasta_path, const char* outdir, bool no_test, bool write_all, bool compressed, bool host_build, uint64_t slice_bytes)
{
    using clk = std::chrono::steady_clock;
    auto since = [](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    const auto t_start = clk::now();
    auto t0 = clk::now();
    double t_read, t_index, t_decode, t_group, t_build, t_exec, t_write;
    const std::string vcf = slurp(vcf_path);
    const auto ref = read_fasta(slurp(fasta_path));
    const uint8_t* text = reinterpret_cast<const uint8_t*>(vcf.data());
    t_read = since(t0); t0 = clk::now();
    v2p_vcf_index* idx = nullptr;
    if (v2p_vcf_index_build(text, vcf.size(), &idx) != 0) {
        std::fprintf(stderr, "reading the file failed: %s\n", v2p_vcf_index_error(idx));
        return 101;
    }
    const uint64_t S = v2p_vcf_index_n_samples(idx), R = v2p_vcf_index_n_records(idx);
    t_index = since(t0); t0 = clk::now();
    GpuContext ctx;
    v2p_decode* dec = nullptr;
    if (v2p_decode_run(ctx.raw(), text, vcf.size(), v2p_vcf_index_row_begin(idx), v2p_vcf_index_row_end(idx), R, S,
                       v2p_vcf_index_csq_begin(idx), v2p_vcf_index_csq_supported(idx), &dec) != V2P_OK) {
        std::fprintf(stderr, "panicked: %s\n", v2p_last_error(ctx.raw()));
        return 101;
    }
    std::vector<uint64_t> hap_begin(2 * S + 1);
    if (v2p_decode_counts(dec, hap_begin.data()) != V2P_OK) { std::fprintf(stderr, "panicked: decode counts unavailable\n"); return 101; }
    std::vector<uint32_t> ids(hap_begin.back() + 1);
    if (v2p_decode_download(dec, ids.data()) != V2P_OK) { std::fprintf(stderr, "panicked: %s\n", v2p_last_error(ctx.raw())); return 101; }
    float kms[4] = {0, 0, 0, 0};
    v2p_decode_timing(dec, &kms[0], &kms[1], &kms[2], &kms[3]);
    v2p_decode_destroy(dec);
    t_decode = since(t0); t0 = clk::now();
    v2p_groups* g = nullptr;
    if (v2p_groups_build(idx, text, hap_begin.data(), ids.data(), 2 * S, 0, &g) != 0) {
        std::fprintf(stderr, "panicked: %s\n", v2p_groups_error(g));
        return 101;
    }
    t_group = since(t0); t0 = clk::now();
    // resident reference: the transcripts the file touches + their two record headers
    const uint64_t n_tx = v2p_groups_n_transcripts(g);
    std::vector<std::string> names(n_tx);
    std::vector<int64_t> tx_off(n_tx, -1);
    std::vector<uint64_t> tx_len(n_tx, 0), hdr_off(2 * n_tx, 0);
    std::string proteome, headers = "\n";
    struct RefTx { uint64_t off, len, hdr[2]; int64_t rank; };
    std::map<std::string, RefTx> all;                              // -a: every transcript of the reference, sorted like the groups
    for (uint64_t r = 0; r < n_tx; ++r) {
        uint64_t b, n;
        v2p_groups_transcript(g, r, &b, &n);
        names[r] = vcf.substr(b, n);
    }
    auto place = [&](const std::string& name, const std::string& seq, int64_t rank) {
        RefTx t{proteome.size(), seq.size(), {0, 0}, rank};
        proteome += seq;
        for (int h = 0; h < 2; ++h) { t.hdr[h] = headers.size(); headers += ">" + name + "_" + char('1' + h) + "\n"; }
        all.emplace(name, t);
        return t;
    };
    for (uint64_t r = 0; r < n_tx; ++r) {
        auto it = ref.find(names[r]);
        if (it == ref.end()) continue;
        const RefTx t = place(names[r], it->second, int64_t(r));
        tx_off[r] = int64_t(t.off); tx_len[r] = t.len; hdr_off[2 * r] = t.hdr[0]; hdr_off[2 * r + 1] = t.hdr[1];
    }
    if (write_all)
        for (const auto& kv : ref)
            if (!all.count(kv.first)) place(kv.first, kv.second, -1);
    auto chk = [&](int rc) { if (rc != V2P_OK) { std::fprintf(stderr, "panicked: %s\n", v2p_last_error(ctx.raw())); std::exit(101); } };
    chk(v2p_upload_reference(ctx.raw(), reinterpret_cast<const uint8_t*>(proteome.data()), proteome.size(),
                             reinterpret_cast<const uint8_t*>(headers.data()), headers.size()));
    v2p_batch* b = nullptr;
    chk(v2p_batch_create(ctx.raw(), &b));
    // Step 5, the image packing and the record text are built ON the device from the per-transcript GIRs collected here as they come out
    // of step 4b -- in SLICES of whole probands (about slice_bytes of FASTA text each) that go through v2p_pipeline_submit_stream as soon
    // as they are complete: while the host runs steps 4a / 4b for the next probands, the slice before is uploaded, built, executed and
    // its text comes back into pinned memory, from where the probands' files are written (parts/exec.rs:23-42 + personalized_genome.rs:
    // 72-117 as a pipeline).  --host-build keeps the host builder (v2p_batch_add_transcript), one image, one download: same bytes.
    struct TxStreamHost {
        std::vector<uint64_t> hap_tx_begin{0}, off, task_begin{0}, alt_begin{0}, hdr_off;
        std::vector<uint32_t> ref_len, res_len, hdr_len, sp, ln, sr;
        std::vector<uint8_t> code, alt;
        uint64_t result_bytes = 0;
        void add(const uint8_t* c, const uint64_t* p, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t o, uint64_t rl,
                 const uint8_t* a, uint64_t na, uint64_t res, uint64_t ho, uint32_t hl) {
            for (uint64_t i = 0; i < n; ++i) { code.push_back(c[i]); sp.push_back(uint32_t(p[i])); ln.push_back(uint32_t(l[i])); sr.push_back(uint32_t(r[i])); }
            alt.insert(alt.end(), a, a + na);
            off.push_back(o); ref_len.push_back(uint32_t(rl)); res_len.push_back(uint32_t(res)); hdr_off.push_back(ho); hdr_len.push_back(hl);
            task_begin.push_back(code.size()); alt_begin.push_back(alt.size());
            result_bytes += res + (hl ? hl + 1u : 0u);
        }
        // (the builder's slab loads read a few entries past a transcript's last task / alt byte: 64 entries of slack behind the arrays)
        v2p_txstream view() {
            if (!padded) { for (int k = 0; k < 64; ++k) { code.push_back(0); sp.push_back(0); ln.push_back(0); sr.push_back(0); alt.push_back(0); } padded = true; }
            v2p_txstream st{};
            st.n_haps = hap_tx_begin.size() - 1; st.n_tx = off.size(); st.n_tasks = code.size() - 64; st.n_alt = alt.size() - 64;
            st.hap_tx_begin = hap_tx_begin.data(); st.tx_proteome_off = off.data(); st.tx_ref_len = ref_len.data(); st.tx_res_len = res_len.data();
            st.tx_task_begin = task_begin.data(); st.tx_alt_begin = alt_begin.data();
            st.code = code.data(); st.start_pos = sp.data(); st.length = ln.data(); st.start_pos_res = sr.data(); st.alt = alt.data();
            st.tx_header_off = hdr_off.data(); st.tx_header_len = hdr_len.data();
            return st;
        }
        bool padded = false;
    };
    std::unique_ptr<TxStreamHost> txs_p(new TxStreamHost());
    // ---- the writer: the probands [s0, s1) of a slice, haplotype h of proband s at bytes [hob[2 (s - s0) + h], hob[.. + 1]) ----
    uint64_t written = 0;
    double t_write_acc = 0;
    auto write_probands = [&](uint64_t s0, uint64_t s1, const uint8_t* bytes, const uint64_t* hob) -> bool {
        const auto tw = clk::now();
        for (uint64_t s = s0; s < s1; ++s) {
            uint64_t nb, nl;
            v2p_vcf_index_sample(idx, s, &nb, &nl);
            const std::string path = std::string(outdir) + "/" + vcf.substr(nb, nl) + (compressed ? ".fasta.gz" : ".fasta");   // personalized_genome.rs:76-80
            std::ofstream f;
            gzFile gz = nullptr;
            if (compressed) gz = gzopen(path.c_str(), "wb9");                        // GzEncoder, Compression::best() (:90)
            else f.open(path, std::ios::binary);
            if (compressed ? gz == nullptr : !f) { std::fprintf(stderr, "Could not create %s\n", path.c_str()); return false; }
            for (int h = 0; h < 2; ++h) {
                const uint64_t k = 2 * (s - s0) + uint64_t(h), begin = hob[k], len = hob[k + 1] - hob[k];
                bool ok = true;
                if (compressed) {
                    for (uint64_t o = 0; o < len && ok; o += 1u << 30) {
                        const unsigned part = unsigned(std::min<uint64_t>(len - o, 1u << 30));
                        ok = gzwrite(gz, bytes + begin + o, part) == int(part);
                    }
                } else {
                    f.write(reinterpret_cast<const char*>(bytes + begin), std::streamsize(len));
                    ok = bool(f);
                }
                if (!ok) { std::fprintf(stderr, "Could not write %s\n", path.c_str()); if (gz) gzclose(gz); return false; }   // (a full disk is an error, not a short file)
                written += len;
            }
            if (gz && gzclose(gz) != Z_OK) { std::fprintf(stderr, "Could not write %s\n", path.c_str()); return false; }
            if (!compressed) { f.close(); if (!f) { std::fprintf(stderr, "Could not write %s\n", path.c_str()); return false; } }
        }
        t_write_acc += since(tw);
        return true;
    };
    // ---- a slice the rows builders refuse (a 1 KiB row with more than 1 024 descriptors): the HOST builder takes any stream (step 5 on the
    // host: v2p_batch_begin_haplotype / _add_transcript / _end_haplotype + v2p_batch_finalize), on a batch of its own; its arena is downloaded whole ----
    auto blocking_slice = [&](TxStreamHost& tx, std::vector<uint8_t>& bytes, std::vector<uint64_t>& hob) {
        v2p_txstream st = tx.view();
        v2p_batch* fb = nullptr;
        chk(v2p_batch_create(ctx.raw(), &fb));
        std::vector<uint64_t> sp64, ln64, sr64;
        for (uint64_t h = 0; h < st.n_haps; ++h) {
            chk(v2p_batch_begin_haplotype(fb));
            for (uint64_t t = st.hap_tx_begin[h]; t < st.hap_tx_begin[h + 1]; ++t) {
                const uint64_t k0 = st.tx_task_begin[t], k1 = st.tx_task_begin[t + 1], n = k1 - k0;
                sp64.assign(st.start_pos + k0, st.start_pos + k1); ln64.assign(st.length + k0, st.length + k1); sr64.assign(st.start_pos_res + k0, st.start_pos_res + k1);
                chk(v2p_batch_add_transcript(fb, st.code + k0, sp64.data(), ln64.data(), sr64.data(), n, st.tx_proteome_off[t], st.tx_ref_len[t],
                                             st.alt + st.tx_alt_begin[t], st.tx_alt_begin[t + 1] - st.tx_alt_begin[t], st.tx_res_len[t],
                                             st.tx_header_off[t], st.tx_header_len[t]));
            }
            chk(v2p_batch_end_haplotype(fb));
        }
        chk(v2p_batch_finalize(fb));
        chk(v2p_batch_execute(fb));
        chk(v2p_batch_sync(fb));
        hob.assign(st.n_haps + 1, 0);
        for (uint64_t h = 0; h < st.n_haps; ++h) { uint64_t begin, len; chk(v2p_batch_hap_range(fb, h, &begin, &len)); hob[h] = begin; hob[h + 1] = begin + len; }
        bytes.resize(hob.back());
        if (!bytes.empty()) chk(v2p_batch_download(fb, 0, bytes.size(), bytes.data()));
        v2p_batch_destroy(fb);
    };
    // ---- the pipeline ----
    constexpr uint32_t SLOTS = 3;
    v2p_pipeline* pipe = nullptr;
    if (!host_build) chk(v2p_pipeline_create(ctx.raw(), SLOTS, &pipe));
    // (every way out of this function -- a transcript the reference would panic on, a full disk -- ends the pipeline's runner before the
    // context it works on goes)
    struct PipeGuard { v2p_pipeline*& p; ~PipeGuard() { if (p) { v2p_pipeline_destroy(p); p = nullptr; } } } pipe_guard{pipe};
    struct Job { uint32_t ticket; uint64_t s0, s1; std::unique_ptr<TxStreamHost> tx; };
    std::vector<Job> inflight;
    uint64_t n_slices = 0, n_fallback = 0;
    auto finish = [&](Job& j) -> bool {
        const uint8_t* bytes = nullptr; uint64_t n = 0, nh = 0;
        const uint64_t* hob = nullptr;
        const int rc = v2p_pipeline_wait(pipe, j.ticket, &bytes, &n);
        if (rc == V2P_ERR_UNSUPPORTED) {
            chk(v2p_pipeline_release(pipe, j.ticket));
            std::vector<uint8_t> fb_bytes; std::vector<uint64_t> fb_hob;
            blocking_slice(*j.tx, fb_bytes, fb_hob);
            ++n_fallback;
            return write_probands(j.s0, j.s1, fb_bytes.data(), fb_hob.data());
        }
        chk(rc);
        chk(v2p_pipeline_result_info(pipe, j.ticket, &hob, &nh, nullptr, nullptr));
        if (nh != 2 * (j.s1 - j.s0)) { std::fprintf(stderr, "panicked: a slice came back with another number of haplotypes\n"); std::exit(101); }
        const bool ok = write_probands(j.s0, j.s1, bytes, hob);
        chk(v2p_pipeline_release(pipe, j.ticket));
        return ok;
    };
    uint64_t slice_s0 = 0;
    auto flush = [&](uint64_t s1) -> bool {                      // probands [slice_s0, s1) are complete: off they go
        if (s1 == slice_s0) return true;
        if (inflight.size() == SLOTS) { if (!finish(inflight.front())) return false; inflight.erase(inflight.begin()); }
        v2p_txstream st = txs_p->view();
        uint32_t t = 0;
        chk(v2p_pipeline_submit_stream(pipe, &st, 0, 0, &t));
        inflight.push_back(Job{t, slice_s0, s1, std::move(txs_p)});
        txs_p.reset(new TxStreamHost());
        slice_s0 = s1; ++n_slices;
        return true;
    };
    TxStreamHost* txs = nullptr;
    auto add_transcript = [&](const uint8_t* c, const uint64_t* p, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t o, uint64_t rl,
                              const uint8_t* a, uint64_t na, uint64_t res, uint64_t ho, uint32_t hl) {
        if (host_build) chk(v2p_batch_add_transcript(b, c, p, l, r, n, o, rl, a, na, res, ho, hl));
        else txs->add(c, p, l, r, n, o, rl, a, na, res, ho, hl);
    };
    const uint64_t* hgb = v2p_groups_hap_group_begin(g);
    const uint32_t* gtx = v2p_groups_group_transcript(g);
    const uint64_t* gmb = v2p_groups_group_member_begin(g);
    const uint32_t* mid = v2p_groups_member_ids(g);
    const uint32_t flags = no_test ? 0u : (V2P_4A_INSPECT_INS_GEN | V2P_4A_PANIC_INSPECT_ERR);     // cli.rs:275-368
    std::vector<v2p_mutation_view> views;
    std::vector<v2p_instruction> ins;
    std::vector<uint8_t> code, alt;
    std::vector<uint64_t> sp, ln, sr;
    const uint8_t ref_code = 0;
    const uint64_t zero = 0;
    auto reference_copy = [&](const RefTx& t, uint64_t hap, size_t name_len) {        // personalized_genome.rs:176-183: not altered -> as in the reference
        add_transcript(&ref_code, &zero, &t.len, &zero, 1, t.off, t.len, nullptr, 0, t.len, t.hdr[hap & 1], uint32_t(name_len + 4));
    };
    for (uint64_t hap = 0; hap < 2 * S; ++hap) {
        txs = txs_p.get();
        if (host_build) chk(v2p_batch_begin_haplotype(b));
        // the haplotype's groups, and with -a the rest of the reference around them, in sorted transcript order
        std::vector<std::pair<const std::pair<const std::string, RefTx>*, int64_t>> todo;     // (reference entry, group index or -1)
        if (write_all) {
            std::map<std::string, int64_t> mine;
            for (uint64_t k = hgb[hap]; k < hgb[hap + 1]; ++k) mine[names[gtx[k]]] = int64_t(k);
            for (const auto& kv : all) { auto it = mine.find(kv.first); todo.emplace_back(&kv, it == mine.end() ? -1 : it->second); }
        } else {
            for (uint64_t k = hgb[hap]; k < hgb[hap + 1]; ++k) { auto it = all.find(names[gtx[k]]); if (it != all.end()) todo.emplace_back(&*it, int64_t(k)); }
        }
        for (const auto& td : todo) {
            if (td.second < 0) { reference_copy(td.first->second, hap, td.first->first.size()); continue; }
            const uint64_t k = uint64_t(td.second);
            const uint32_t r = gtx[k];
            if (tx_off[r] < 0) continue;                                       // transcript_instructions.rs:37-41
            const uint64_t n = gmb[k + 1] - gmb[k];
            views.resize(n); ins.resize(n + 1);
            for (uint64_t i = 0; i < n; ++i) v2p_groups_mutation_view(g, mid[gmb[k] + i], &views[i]);
            uint64_t n_ins = 0;
            const int rc4a = v2p_transcript_instructions(views.data(), n, flags, ins.data(), ins.size(), &n_ins);
            if (rc4a == V2P_4A_SKIP) { if (write_all) reference_copy(td.first->second, hap, names[r].size()); continue; }
            if (rc4a != V2P_4A_OK) { std::fprintf(stderr, "panicked: instruction generation for transcript %s\n", names[r].c_str()); return 101; }
            uint64_t payload = 8;
            for (uint64_t i = 0; i < n_ins; ++i) payload += 2 * ins[i].data_len;
            const uint64_t cap = 3 * n_ins + 4;
            code.resize(cap); sp.resize(cap); ln.resize(cap); sr.resize(cap); alt.resize(payload);
            uint64_t n_tasks = 0, n_alt = 0, res_len = 0;
            const int rc4b = v2p_transcript_g_rep(ins.data(), n_ins, tx_len[r], code.data(), sp.data(), ln.data(), sr.data(), cap, &n_tasks,
                                                  alt.data(), payload, &n_alt, &res_len);
            if (rc4b == V2P_4B_MUST_BE_LAST) { if (write_all) reference_copy(td.first->second, hap, names[r].size()); continue; }   // haplotype_instruction.rs:100-104
            if (rc4b != V2P_4B_OK) { std::fprintf(stderr, "panicked: task generation for transcript %s (%d)\n", names[r].c_str(), rc4b); return 101; }
            if (!no_test && v2p_inspect_transcript_tasks(ln.data(), sr.data(), n_tasks, res_len, nullptr) != V2P_4B_INSPECT_OK) {   // INSPECT_TXP
                std::fprintf(stderr, "panicked: size mismatched / non-contiguous tasks in transcript %s\n", names[r].c_str());
                return 101;
            }
            add_transcript(code.data(), sp.data(), ln.data(), sr.data(), n_tasks, uint64_t(tx_off[r]), tx_len[r],
                           alt.data(), n_alt, res_len, hdr_off[2 * r + (hap & 1)], uint32_t(names[r].size() + 4));
        }
        if (host_build) chk(v2p_batch_end_haplotype(b));
        else {
            txs->hap_tx_begin.push_back(txs->off.size());
            if ((hap & 1) && (txs->result_bytes >= slice_bytes || hap + 1 == 2 * S) && !flush(hap / 2 + 1)) return 101;   // a proband is complete
        }
    }
    t_build = since(t0) - t_write_acc; t0 = clk::now();
    if (host_build) {
        chk(v2p_batch_finalize(b));
        chk(v2p_batch_execute(b));
        chk(v2p_batch_sync(b));
        t_exec = since(t0); t0 = clk::now();
        std::vector<uint64_t> hob(2 * S + 1, 0);
        for (uint64_t h = 0; h < 2 * S; ++h) { uint64_t begin, len; chk(v2p_batch_hap_range(b, h, &begin, &len)); hob[h] = begin; hob[h + 1] = begin + len; }
        std::vector<uint8_t> bytes(hob.back());
        if (!bytes.empty()) chk(v2p_batch_download(b, 0, bytes.size(), bytes.data()));
        if (!write_probands(0, S, bytes.data(), hob.data())) return 101;
    } else {
        const double w0 = t_write_acc;
        for (Job& j : inflight) if (!finish(j)) return 101;
        inflight.clear();
        v2p_pipeline_destroy(pipe);
        pipe = nullptr;
        t_exec = since(t0) - (t_write_acc - w0); t0 = clk::now();
    }
    t_write = host_build ? since(t0) : t_write_acc;
    std::printf("vcf: %llu records, %llu probands, %llu bytes of FASTA written to %s\n", (unsigned long long)R, (unsigned long long)S,
                (unsigned long long)written, outdir);
    std::printf("{\"records\": %llu, \"probands\": %llu, \"fasta_bytes\": %llu, \"slices\": %llu, \"slices_through_the_host_builder\": %llu, \"seconds\": {\"read_files\": %.4f, \"index\": %.4f, "
                "\"decode_incl_h2d\": %.4f, \"grouping\": %.4f, \"steps_4a_4b_5\": %.4f, \"h2d_step6_sync\": %.4f, \"d2h_write\": %.4f, \"total\": %.4f}, "
                "\"decode_kernels_ms\": {\"parse\": %.3f, \"count\": %.3f, \"scan\": %.3f, \"emit\": %.3f}}\n",
                (unsigned long long)R, (unsigned long long)S, (unsigned long long)written, (unsigned long long)n_slices, (unsigned long long)n_fallback,
                t_read, t_index, t_decode, t_group, t_build, t_exec, t_write,
                since(t_start), kms[0], kms[1], kms[2], kms[3]);
    v2p_batch_destroy(b);
    v2p_groups_destroy(g);
    v2p_vcf_index_destroy(idx);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// sharded: parts/exec.rs:34-40 over the devices of one node, in one process (ppgg::execute_sharded)
// the checker's digest (include/vcf2prot_hip.h: v2p_batch_digests' definition) of bytes in host memory
static uint64_t host_digest(const uint8_t* p, uint64_t n)
{
    auto mix = [](uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
    uint64_t s = 0;
    const uint64_t nw = n >> 3;
    for (uint64_t k = 0; k < nw; ++k) { uint64_t w; std::memcpy(&w, p + 8 * k, 8); s += (w + 0x0101010101010101ull) * mix(k); }
    uint64_t t = 0;
    for (uint64_t i = 8 * nw; i < n; ++i) t += (uint64_t(p[i]) + 1ull) << (8 * (i & 7));
    if (n & 7) s += t * mix(nw);
    return s;
}

static int sharded(const char* preset, uint32_t samples, int n_devices, bool oversubscribe, int threads, bool streamed, uint64_t slice_mb)
{
    v2p_cohort_params p;
    if (v2p_cohort_preset(preset, &p)) { std::fprintf(stderr, "unknown preset %s\n", preset); return 2; }
    if (samples) p.n_samples = samples;
    v2p_cohort* c = nullptr;
    if (v2p_cohort_create(&p, &c)) return 2;
    const uint64_t n_haps = v2p_cohort_n_haplotypes(c);
    const int have = v2p_device_count();
    if (have <= 0) { std::fprintf(stderr, "no HIP device: the gpu engine has no CPU fallback\n"); return 1; }
    if (n_devices > have && !oversubscribe) { std::fprintf(stderr, "%d devices asked for, %d present (--oversubscribe shares them)\n", n_devices, have); return 2; }
    std::vector<int> devices;
    for (int d = 0; d < n_devices; ++d) devices.push_back(d % have);
    std::vector<uint64_t> sizes(n_haps);
    if (v2p_cohort_result_sizes(c, 0, n_haps, threads, sizes.data())) return 2;
    std::vector<uint64_t> digest(n_haps, 0);
    const int per = threads / n_devices > 0 ? threads / n_devices : 1;
    auto make_stream = [&](uint64_t h0, uint64_t h1) -> std::shared_ptr<const v2p_txstream> {
        struct Owned { v2p_txstream_buf buf; v2p_txstream view; };
        auto o = std::shared_ptr<Owned>(new Owned(), [](Owned* q) { v2p_txstream_free(&q->buf); delete q; });
        if (v2p_cohort_txstream(c, h0, h1, per, &o->buf)) throw std::runtime_error("v2p_cohort_txstream failed");
        const v2p_txstream_buf& b = o->buf;
        o->view = v2p_txstream{b.n_haps, b.n_tx, b.n_tasks, b.n_alt, b.hap_tx_begin, b.tx_proteome_off, b.tx_ref_len, b.tx_res_len, b.tx_task_begin, b.tx_alt_begin,
                               b.code, b.start_pos, b.length, b.start_pos_res, b.alt, b.tx_header_off, b.tx_header_len};
        return std::shared_ptr<const v2p_txstream>(o, &o->view);
    };
    auto consume = [&](const DeviceShard& s, v2p_ctx* ctx, v2p_batch* b) {
        // the shard's haplotypes sit in its arena at the offsets the cohort-wide prefix sum gives them, minus the shard's own
        uint64_t at = 0;
        for (uint64_t h = s.h0; h < s.h1; ++h) {
            uint64_t begin = 0, len = 0;
            if (v2p_batch_hap_range(b, h - s.h0, &begin, &len) != V2P_OK || begin != at || len != sizes[h]) throw Panic(V2P_ERR_STATE, "haplotype range of a shard disagrees with the cohort's result sizes");
            at += len;
        }
        if (s.h1 > s.h0 && v2p_batch_digests(b, digest.data() + s.h0, s.h1 - s.h0) != V2P_OK) throw Panic(V2P_ERR_HIP, v2p_last_error(ctx));
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<DeviceShard> shards;
    std::atomic<uint64_t> host_bytes{0}, n_slices{0};
    auto consume_slice = [&](const DeviceShard&, uint64_t h0, uint64_t h1, const uint8_t* bytes, uint64_t n, const uint64_t* hob, const uint64_t* dig) {
        // what crossed the link: every haplotype digested HERE, on the host, and compared with the device's digest of its arena
        if (hob[h1 - h0] != n) throw Panic(V2P_ERR_STATE, "a slice's offsets do not end at its size");
        for (uint64_t h = h0; h < h1; ++h) {
            if (hob[h - h0 + 1] - hob[h - h0] != sizes[h]) throw Panic(V2P_ERR_STATE, "haplotype range of a slice disagrees with the cohort's result sizes");
            digest[h] = host_digest(bytes + hob[h - h0], sizes[h]);
            if (dig && dig[h - h0] != digest[h]) throw Panic(V2P_ERR_STATE, "the host bytes of a haplotype digest differently from its arena on the device");
        }
        host_bytes += n; ++n_slices;
    };
    try {
        if (streamed) shards = execute_streamed(sizes, devices, v2p_cohort_proteome(c), v2p_cohort_proteome_len(c), nullptr, 0, make_stream, consume_slice,
                                                slice_mb << 20, 4, uint32_t(per > 16 ? 16 : per));
        else shards = execute_sharded(sizes, devices, v2p_cohort_proteome(c), v2p_cohort_proteome_len(c), make_stream, consume);
    }
    catch (const Panic& e) { std::fprintf(stderr, "panicked: %s\n", e.what()); return 101; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t total = 0;
    for (uint64_t b : sizes) total += b;
    std::printf("{\"mode\": \"sharded: %d device context(s) in one process, ranges of equal result bytes, %s\", \"preset\": \"%s\", "
                "\"samples\": %u, \"haplotypes\": %llu, \"result_bytes\": %llu, \"devices_present\": %d, \"wall_seconds_incl_generation_and_upload\": %.6f, "
                "\"streamed\": %s, \"slices\": %llu, \"host_bytes\": %llu, \"shards\": [",
                n_devices, streamed ? "one v2p_pipeline per device: Task-vector slices in, host bytes out (v2p_pipeline_submit_stream)" : "one v2p_batch_build_and_execute each",
                preset, p.n_samples, (unsigned long long)n_haps, (unsigned long long)total, have, secs, streamed ? "true" : "false",
                (unsigned long long)n_slices.load(), (unsigned long long)host_bytes.load());
    for (size_t r = 0; r < shards.size(); ++r) {
        const DeviceShard& s = shards[r];
        std::printf("%s{\"rank\": %d, \"device\": %d, \"h0\": %llu, \"h1\": %llu, \"byte_offset\": %llu, \"bytes\": %llu, \"seconds\": %.6f, \"oneshot_ms\": %.4f}", r ? ", " : "",
                    s.rank, s.device, (unsigned long long)s.h0, (unsigned long long)s.h1, (unsigned long long)s.byte_offset, (unsigned long long)s.bytes, s.seconds, double(s.oneshot_ms));
    }
    std::printf("], \"digests\": [");
    for (uint64_t h = 0; h < n_haps; ++h) std::printf("%s%llu", h ? ", " : "", (unsigned long long)digest[h]);
    std::printf("]}\n");
    v2p_cohort_destroy(c);
    return 0;
}

static uint64_t vcf_slice_kb = 0;       // vcf --slice-kb K: slices of K KiB of FASTA text (tests: several slices out of a small file)

int main(int argc, char** argv)
{
    if (argc >= 5 && !std::strcmp(argv[1], "vcf")) {
        bool no_test = false, write_all = false, compressed = false, host_build = false;
        uint64_t slice_mb = 256;
        for (int i = 5; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--slice-kb") && i + 1 < argc) { slice_mb = 0; vcf_slice_kb = std::strtoull(argv[++i], nullptr, 10); continue; }
            no_test |= !std::strcmp(argv[i], "--no-test");
            host_build |= !std::strcmp(argv[i], "--host-build");
            write_all |= !std::strcmp(argv[i], "--write-all") || !std::strcmp(argv[i], "-a");
            compressed |= !std::strcmp(argv[i], "--write-compressed") || !std::strcmp(argv[i], "-c");
        }
        try { return vcf_mode(argv[2], argv[3], argv[4], no_test, write_all, compressed, host_build, slice_mb ? slice_mb << 20 : (vcf_slice_kb ? vcf_slice_kb << 10 : 1)); }
        catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 101; }
    }
    if (argc >= 3 && !std::strcmp(argv[1], "shard")) {              // the cut rule alone (no GPU): one "begin end" line per rank
        std::vector<uint64_t> sizes;
        for (int i = 3; i < argc; ++i) sizes.push_back(std::strtoull(argv[i], nullptr, 10));
        for (const auto& r : shard_by_bytes(sizes, std::atoi(argv[2]))) std::printf("%llu %llu\n", (unsigned long long)r.first, (unsigned long long)r.second);
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "sharded")) {
        int n_devices = 1, threads = int(std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8);
        bool over = false, streamed = false;
        uint64_t slice_mb = 1152;
        for (int i = 4; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) n_devices = std::atoi(argv[++i]);
            else if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = std::atoi(argv[++i]);
            else if (!std::strcmp(argv[i], "--slice-mb") && i + 1 < argc) slice_mb = std::strtoull(argv[++i], nullptr, 10);
            else if (!std::strcmp(argv[i], "--oversubscribe")) over = true;
            else if (!std::strcmp(argv[i], "--streamed")) streamed = true;
        }
        if (slice_mb < 1) slice_mb = 1;
        if (n_devices < 1 || n_devices > 64) { std::fprintf(stderr, "--devices 1 .. 64\n"); return 2; }
        if (threads > 64) threads = 64;
        return sharded(argv[2], uint32_t(std::strtoul(argv[3], nullptr, 10)), n_devices, over, threads, streamed, slice_mb);
    }
    if (argc >= 2 && !std::strcmp(argv[1], "kat")) return kat();
    if (argc >= 5 && !std::strcmp(argv[1], "run"))
        return run(argv[2], std::strtoull(argv[3], nullptr, 10), std::atoi(argv[4]), argc >= 6 && (!std::strcmp(argv[5], "--shared") || !std::strcmp(argv[5], "--async")),
                   argc >= 6 && !std::strcmp(argv[5], "--async"));
    std::fprintf(stderr, "usage: v2p_harness kat | run <preset> <haplotypes> <threads> [--shared | --async] | sharded <preset> <samples> --devices N [--oversubscribe] [--streamed [--slice-mb M]] | shard <world> <bytes...>\n");
    return 2;
}
