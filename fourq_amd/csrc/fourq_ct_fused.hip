// Third translation unit of libfourq_amd.so: the fused variable-base kernels with CONSTANT-TIME table selection
// (FOURQ_CT_SELECT / fourq_ctx_set_ct_select; kernels.hip.h, curve.hip.h).  Same build flavour as fourq_amd.hip
// (FQ_CHAIN=0); the lane's 8-entry table is loaded into registers once and every ladder step scans all of it.
#ifndef FQ_CHAIN
#define FQ_CHAIN 0
#endif
#include "kernels.hip.h"

namespace fq {

int ct_launch_fused(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (algo == ENDO) {
        if (dh) hipLaunchKernelGGL((ladder_kernel<ENDO, FUSED, true, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((ladder_kernel<ENDO, FUSED, false, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    } else {
        if (dh) hipLaunchKernelGGL((ladder_kernel<WINDOWED, FUSED, true, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((ladder_kernel<WINDOWED, FUSED, false, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    }
    return (int)hipGetLastError();
}

namespace {
template <int ALGO, bool DH, int LPE> int launch_pair_ct(bool fixed, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (fixed) hipLaunchKernelGGL((pair_kernel<ALGO, DH, true, true, LPE>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((pair_kernel<ALGO, DH, true, false, LPE>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return (int)hipGetLastError();
}
template <int LPE> int launch_pair_ct_lpe(int algo, bool dh, bool fixed, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (algo == ENDO) return dh ? launch_pair_ct<ENDO, true, LPE>(fixed, grid, stream, a) : launch_pair_ct<ENDO, false, LPE>(fixed, grid, stream, a);
    return dh ? launch_pair_ct<WINDOWED, true, LPE>(fixed, grid, stream, a) : launch_pair_ct<WINDOWED, false, LPE>(fixed, grid, stream, a);
}
}  // namespace
int ct_launch_pair(int algo, bool dh, bool fixed, bool quad, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    return quad ? launch_pair_ct_lpe<4>(algo, dh, fixed, grid, stream, a) : launch_pair_ct_lpe<2>(algo, dh, fixed, grid, stream, a);
}
int ct_launch_pair_mixed(bool quad, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (quad) hipLaunchKernelGGL((pair_kernel<ENDO, false, true, false, 4, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((pair_kernel<ENDO, false, true, false, 2, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return (int)hipGetLastError();
}
int ct_launch_comb_quad(bool quad, unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, u32 n) {
    if (quad) hipLaunchKernelGGL((comb_quad_kernel<true, 4>), dim3(grid), dim3(BLOCK), 0, stream, scalars, comb_limbs, out, status, n);
    else hipLaunchKernelGGL((comb_quad_kernel<true, 2>), dim3(grid), dim3(BLOCK), 0, stream, scalars, comb_limbs, out, status, n);
    return (int)hipGetLastError();
}
int ct_launch_mixed_queue(unsigned grid, hipStream_t stream, const LadderArgs& a, const u32* var_list, const u32* fix_list, const u32* counts, u32* queue_head) {
    hipLaunchKernelGGL(mixed_queue_kernel<true>, dim3(grid), dim3(BLOCK), 0, stream, a, var_list, fix_list, counts, queue_head);
    return (int)hipGetLastError();
}

}  // namespace fq
