// frontend_common.hpp -- constants shared by the host front-end pieces.
#pragma once
#include <string_view>

namespace v2p_frontend {

// Constants.rs:3-8 (SUP_TYPE), same order; MutationType::from_str accepts exactly these spellings (mutation_ds.rs:19-46)
inline constexpr std::string_view SUP_TYPE[22] = {
    "missense", "*missense", "frameshift", "*frameshift", "inframe_insertion", "*inframe_insertion", "inframe_deletion",
    "*inframe_deletion", "stop_gained", "stop_lost", "*missense&inframe_altering", "*frameshift&stop_retained",
    "*stop_gained&inframe_altering", "frameshift&stop_retained", "inframe_deletion&stop_retained",
    "inframe_insertion&stop_retained", "stop_gained&inframe_altering", "start_lost", "*stop_gained", "stop_lost&frameshift",
    "missense&inframe_altering", "start_lost&splice_region"};

inline int sup_type_index(std::string_view t)
{
    for (int i = 0; i < 22; ++i)
        if (SUP_TYPE[i] == t) return i;
    return -1;
}

}  // namespace v2p_frontend
