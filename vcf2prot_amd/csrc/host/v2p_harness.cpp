// v2p_harness.cpp -- C++ host harness that plays the role of the reference's exec::execute
// (parts/exec.rs:23-42) on top of the C ABI: it builds haplotype GIRs exactly as steps 4-5
// would hand them over (synthetic cohorts, include/v2p_cohort.h), runs every one through
// GIR::execute(Engine::GPU) from a pool of worker threads, and reports wall-clock throughput
// of this GIR-faithful mode (tapes cross PCIe as Rust chars, 4 bytes per residue, both ways).
//
//   v2p_harness kat                          reference known-answer tests through the mirror
//   v2p_harness run <preset> <haps> <threads>   e.g. run C2 64 8
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../../include/v2p_cohort.h"
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

static int run(const char* preset, uint64_t n_haps, int threads)
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
        for (size_t i = 0; i < res.size(); ++i) s += (uint64_t(res[i] & 0xFFu) + 1ull) * mix64(i);
        digest[h] = s;
    };
    const auto t0 = std::chrono::steady_clock::now();
    try { execute(n_haps, threads, 0, make_gir, consume); }
    catch (const std::exception& e) { std::fprintf(stderr, "engine error: %s\n", e.what()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"mode\": \"gir-faithful (u32 tapes over PCIe, %d worker threads, incl. host GIR build)\", \"preset\": \"%s\", "
                "\"haplotypes\": %llu, \"aa\": %llu, \"seconds\": %.6f, \"aa_per_s\": %.4e, \"digests\": [",
                threads, preset, (unsigned long long)n_haps, (unsigned long long)aa.load(), secs, double(aa.load()) / secs);
    for (uint64_t h = 0; h < n_haps; ++h) std::printf("%s%llu", h ? ", " : "", (unsigned long long)digest[h]);
    std::printf("]}\n");
    v2p_cohort_destroy(c);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && !std::strcmp(argv[1], "kat")) return kat();
    if (argc >= 5 && !std::strcmp(argv[1], "run")) return run(argv[2], std::strtoull(argv[3], nullptr, 10), std::atoi(argv[4]));
    std::fprintf(stderr, "usage: v2p_harness kat | run <preset> <haplotypes> <threads>\n");
    return 2;
}
