// GroupNorm apply -> proj_in -> attn1 -> attn2 -> attn_temporal -> ff of a Transformer3DModel as one launch: instance <2, 1, 1> of
// tattn_kernel.h in its own translation unit.
#include "tattn_kernel.h"

int uav_tattn_run_block_pi(const void* tattn_args, dim3 grid, hipStream_t stream) {
    const TattnArgs& a = *(const TattnArgs*)tattn_args;
    static UavDynLds lds4;
    if (int rc = uav_set_dyn_lds(lds4, (const void*)tattn_sublayer_kernel<2, 1, 1>, TSMEM)) return rc;
    hipLaunchKernelGGL((tattn_sublayer_kernel<2, 1, 1>), grid, dim3(256), TSMEM, stream, a);
    return uav_launch_status();
}
