// binarize_fused.hip — placeholder until the fused sliding-window kernel lands (next commit).
#include "prl_internal.h"

namespace prl_hip {

size_t fused_small_bytes(int) { return 0; }
bool fused_supports(const ThrParams&) { return false; }
int fused_run(const ThrParams&, const PageSet&, int, const PageSetOut&, void*, PageGlobals*, hipStream_t)
{
    set_error_detail("fused path not built");
    return PRL_ERR_HIP;
}

}  // namespace prl_hip
