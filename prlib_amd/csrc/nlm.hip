// nlm.hip — NL-means entry points (implementation lands in a later commit of this round).
#include "prl_internal.h"

using namespace prl_hip;

extern "C" {

int prl_hip_nlm_planes_device(int, int, float, const uint8_t*, size_t, size_t, int, int, uint8_t*, size_t,
                              size_t, void*)
{
    set_error_detail("prl_hip_nlm_planes_device: not implemented yet");
    return PRL_ERR_HIP;
}

int prl_hip_denoise_batch_device(int, int, float, const uint8_t*, size_t, size_t, int, int, uint8_t*,
                                 size_t, size_t, void*)
{
    set_error_detail("prl_hip_denoise_batch_device: not implemented yet");
    return PRL_ERR_HIP;
}

int prl_hip_denoise_host(int, float, const uint8_t*, size_t, int, int, uint8_t*, size_t)
{
    set_error_detail("prl_hip_denoise_host: not implemented yet");
    return PRL_ERR_HIP;
}
}
