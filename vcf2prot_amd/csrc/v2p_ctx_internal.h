// v2p_ctx_internal.h -- what other translation units of libvcf2prot_hip.so may do with a v2p_ctx (defined in v2p_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

struct v2p_ctx;

namespace v2p {
hipStream_t ctx_stream(v2p_ctx* c);
int ctx_device(v2p_ctx* c);
int ctx_fail(v2p_ctx* c, int code, const std::string& msg, int64_t index);
void ctx_lock(v2p_ctx* c);
void ctx_unlock(v2p_ctx* c);
}  // namespace v2p
