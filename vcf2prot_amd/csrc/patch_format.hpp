// patch_format.hpp -- the PATCH image format (patch_image.h: what it is for; patch_image.hip: the kernels): constants and word layouts shared by
// the device code and the host restatement (patch_image_host.hpp).  No HIP here.
#pragma once
#include <stdint.h>
#include "sir_pack.hpp"

namespace v2p {

constexpr uint32_t PATCH_ROWS = 8;                         // 1 KiB rows of a chunk: the workgroup's LDS image
constexpr uint32_t PATCH_G = PATCH_ROWS * ROW_BYTES;       // 8 192 bytes of arena per chunk
constexpr uint32_t PATCH_SEG_CAP = 1024;                   // segment slots per chunk
constexpr uint32_t PATCH_PATCH_CAP = 1024;                 // patch slots per chunk
constexpr uint32_t PATCH_SRC_BITS = 34;                    // a segment's source offset: 16 GB of proteome (+ record headers) / alt bytes
constexpr uint64_t PATCH_SRC_MAX = (1ull << PATCH_SRC_BITS) - 1;
constexpr uint32_t PATCH_IMM_MAX = 4;                      // literal bytes a segment word carries (space 3): the low 32 bits of its source field
constexpr uint64_t CHUNK_PATCH = CHUNK_DENSE | CHUNK_WAVE; // chunk header flags of a patch image (no other image sets both)
constexpr uint32_t STATUS_PATCH_DECLINED = 9;              // not an error of the stream: the format does not take it (slots, source range)

V2P_HOST_DEVICE inline uint64_t patch_seg(uint64_t src, uint32_t start, uint32_t len, unsigned space)
{
    return (src & PATCH_SRC_MAX) | (uint64_t(start & 0x3FFFu) << 34) | (uint64_t(len & 0x3FFFu) << 48) | (uint64_t(space & 3u) << 62);
}
V2P_HOST_DEVICE inline uint64_t patch_seg_src(uint64_t w) { return w & PATCH_SRC_MAX; }
V2P_HOST_DEVICE inline uint32_t patch_seg_start(uint64_t w) { return uint32_t(w >> 34) & 0x3FFFu; }
V2P_HOST_DEVICE inline uint32_t patch_seg_len(uint64_t w) { return uint32_t(w >> 48) & 0x3FFFu; }
V2P_HOST_DEVICE inline unsigned patch_seg_space(uint64_t w) { return unsigned(w >> 62); }
V2P_HOST_DEVICE inline uint32_t patch_word(uint32_t pos, uint32_t byte) { return (pos & 0x3FFFu) | ((byte & 0xFFu) << 16); }
// chunk record: task_begin = first segment slot (42 bits) | patches << 42; dst_n = arena offset | segments << 48 | CHUNK_PATCH
V2P_HOST_DEVICE inline uint32_t patch_chunk_patches(uint64_t task_begin) { return uint32_t(task_begin >> TB_IDX_BITS) & 0xFFFu; }

}  // namespace v2p
