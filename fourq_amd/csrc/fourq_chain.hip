// Second translation unit of libfourq_amd.so: the kernels that profit from chained carries (FQ_CHAIN=1, see
// kernels.hip.h): fixed-base ladders (table in LDS), the two-kernel route for large variable-base batches
// (prep_kernel + ladder_kernel<PREBUILT>) and the fixed-base comb.  Only launchers are exported to the other
// translation unit; the C ABI lives in fourq_amd.hip.
#ifndef FQ_CHAIN
#define FQ_CHAIN 1
#endif
#include "kernels.hip.h"

namespace fq {

namespace {
template <int ALGO, int SRC, bool DH> int launch(unsigned grid, hipStream_t stream, const LadderArgs& a) {
    hipLaunchKernelGGL((ladder_kernel<ALGO, SRC, DH>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return (int)hipGetLastError();
}
template <int ALGO, int SRC> int launch_dh(bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    return dh ? launch<ALGO, SRC, true>(grid, stream, a) : launch<ALGO, SRC, false>(grid, stream, a);
}
}  // namespace

int chain_launch_ladder(int algo, int src, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (src == LDS) return algo == ENDO ? launch_dh<ENDO, LDS>(dh, grid, stream, a) : launch_dh<WINDOWED, LDS>(dh, grid, stream, a);
    return algo == ENDO ? launch_dh<ENDO, PREBUILT>(dh, grid, stream, a) : launch_dh<WINDOWED, PREBUILT>(dh, grid, stream, a);
}
int chain_launch_prep(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (algo == ENDO) {
        if (dh) hipLaunchKernelGGL((prep_kernel<ENDO, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((prep_kernel<ENDO, false>), dim3(grid), dim3(BLOCK), 0, stream, a);
    } else {
        if (dh) hipLaunchKernelGGL((prep_kernel<WINDOWED, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((prep_kernel<WINDOWED, false>), dim3(grid), dim3(BLOCK), 0, stream, a);
    }
    return (int)hipGetLastError();
}
int chain_launch_comb(unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, u32 n) {
    hipLaunchKernelGGL(comb_kernel, dim3(grid), dim3(BLOCK), 0, stream, scalars, comb_limbs, out, status, n);
    return (int)hipGetLastError();
}

}  // namespace fq
