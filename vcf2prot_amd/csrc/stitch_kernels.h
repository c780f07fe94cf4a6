// stitch_kernels.h -- argument blocks and host launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sir_pack.hpp"

namespace v2p {

// device status word: min over offending rows of (row << 8 | reason); ~0ull = clean
enum : uint32_t {
    STATUS_BAD_CODE = 1,        // exe_code not in {0,1}
    STATUS_RES_OOB = 2,         // result range outside the result tape / arena
    STATUS_SRC_OOB = 3,         // source range outside its tape
    STATUS_NOT_CONTIGUOUS = 4   // gir.rs:208-226 predicate
};
constexpr unsigned long long STATUS_CLEAN = ~0ull;
constexpr int STITCH_TASKS_PER_LANE = 4;               // default descriptors per lane (any chunk <= 1024 tasks is legal)
constexpr uint32_t DOTS_BYTES = 64u * 1024u + 128u;   // per-device buffer of '.' that fill descriptors gather from

struct StitchArgs {
    const uint64_t* desc;      // packed descriptors (sir_pack.hpp)
    uint64_t        n_desc;    // chunks pointing outside desc[0, n_desc) are refused on the device
    const Chunk*    chunks;
    uint32_t        n_chunks;
    const uint8_t*  src0;      // space 0: resident proteome / the GIR's ref tape; 32 readable bytes before and after
    uint64_t        src0_len;
    const uint8_t*  src1;      // space 1: payload arena / the GIR's alt tape;    32 readable bytes before and after
    uint64_t        src1_len;
    uint8_t*        out;       // result arena, 16-byte aligned
    uint64_t        out_len;
    unsigned long long* status;
    const uint8_t*  dots;      // set by launch_stitch(): DOTS_BYTES of '.'
    const Chunk*    next_chunks = nullptr;   // set by launch_stitch(): the chunk records of the NEXT phase, whose image the trailing
    uint32_t        n_next = 0;              // workgroups of a wave launch read ahead (stitch_wave.hip); 0: none
    uint32_t        store_sc1 = 0;           // set by launch_stitch(): the wave kernel's row stores "sc1 nt" instead of "nt" (thin descriptor streams)
    // launch options (include/vcf2prot_hip.h: v2p_launch_opts; 0 / -1 = the library's choice)
    uint64_t        opt_phase_bytes = 0;     // image bytes per phase of a wave / long-run image (~0ull: one launch, no read-ahead)
    uint32_t        opt_phase_min_chunks = 0;// images with fewer chunks are launched at once
    int32_t         opt_store_sc1 = -1;      // wave images: force (1) / forbid (0) "sc1 nt" row stores
    uint32_t        opt_touch = 0;           // bench builds: 1 = no read-ahead, 2 = the read-ahead as kernels of its own, 4 = one launch for all phases
    uint32_t        rows = 0;                // set by launch_stitch(): a rows image (sir_pack.hpp: every chunk carries CHUNK_CLIP) -- stitchw_kernel's ROWS instance
    uint64_t        img_desc = 0, img_bytes = 0; // the chunk table is a RANGE of a larger image (v2p_batch_build_and_execute: one slice): descriptors and result
                                             // bytes of the range, for the routing (0: n_desc, out_len)
    // two launch streams (set by the batch's execute): a pure wave image's phases are cut in halves that alternate between `stream` and
    // `aux_stream`, staggered by half a piece -- while one stream's kernel drains and the next one's waves ramp up, the other stream's
    // kernel is in mid-flight and takes the freed wave slots (launch_stitch: "dual"); nullptr: one stream
    hipStream_t     aux_stream = nullptr;
    hipEvent_t      ev_fork = nullptr, ev_join = nullptr;
    uint32_t        opt_dual = 0;            // 1: use them
    // STAGED descriptors (round 5; pure wave rows images in the ride form): two buffers of stage_chunks x 64 descriptor slots; the read-ahead
    // of a phase copies its chunks' descriptors into one of them, row = chunk index inside the phase, and stitchw_kernel reads them there
    uint64_t*       stage = nullptr;         // [2 * stage_chunks * 64] (nullptr: descriptors are read where the image has them)
    uint32_t        stage_chunks = 0;        // rows per buffer: at least stitch_stage_chunks() of the same arguments
    const uint64_t* stage_cur = nullptr;     // set by launch_stitch(): this phase's buffer / the next phase's
    uint64_t*       stage_next = nullptr;
    uint32_t        phase_chunks = 0;        // set by launch_stitch(): != 0 -- ONE wave launch for all phases of that many chunks, the read-ahead
                                             // workgroups of phase g + 1 placed in the grid before the stitch workgroups of phase g
};

// rows per staging buffer launch_stitch() would use for these arguments (0: this image is not launched in the form that stages)
uint32_t stitch_stage_chunks(const StitchArgs& args, int nontemporal);

struct OrderedArgs {
    const uint8_t*  code;
    const uint64_t* start_pos;
    const uint64_t* length;
    const uint64_t* start_pos_res;
    uint64_t        n_tasks;
    const uint8_t*  ref;
    const uint8_t*  alt;
    uint8_t*        res;
    uint64_t        esize;     // bytes per tape element
};

struct ValidateArgs {
    const uint8_t*  code;
    const uint64_t* start_pos;
    const uint64_t* length;
    const uint64_t* start_pos_res;
    uint64_t        n_tasks;
    uint64_t        n_ref, n_alt, n_res;
    unsigned long long* status;
};

struct DigestArgs {
    const uint8_t*  out;
    const uint64_t* hap_begin;   // [n_haps + 1]
    uint64_t        n_haps;
    uint64_t*       digest;      // [n_haps], zeroed by the caller
};

hipError_t launch_stitch(const StitchArgs& a, hipStream_t stream, int nontemporal, uint32_t max_blocks);
hipError_t launch_stitch_wave(const StitchArgs& a, hipStream_t stream, bool nt, int waves_per_group);   // stitch_wave.hip; `a.dots` set by the caller
hipError_t launch_ordered(const OrderedArgs& a, hipStream_t stream);
hipError_t launch_validate(const ValidateArgs& a, hipStream_t stream);
hipError_t launch_digest(const DigestArgs& a, uint64_t out_bytes, hipStream_t stream);
// the translation units' code objects loaded now instead of at their first launch (v2p_init)
hipError_t preload_stitch_kernels(hipStream_t stream);
hipError_t preload_stitch_wave(hipStream_t stream);
hipError_t preload_build_kernels(hipStream_t stream);
hipError_t preload_build_rows(hipStream_t stream);
}  // namespace v2p

namespace v2p {
// descriptors per lane the stitch kernel needs for chunks of at most `max_chunk_tasks` descriptors
inline int tasks_per_lane_for(uint32_t max_chunk_tasks) { return max_chunk_tasks <= 256u ? 1 : (max_chunk_tasks <= 512u ? 2 : 4); }
}
