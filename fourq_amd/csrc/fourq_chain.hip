// Second translation unit of libfourq_amd.so: the kernels that profit from chained carries (FQ_CHAIN=1, see
// kernels.hip.h): fixed-base ladders (table in LDS), the two-kernel route for large variable-base batches
// (prep_kernel + ladder_kernel<PREBUILT>), the fixed-base comb and the batched normalisation.  Only launchers are exported to the other
// translation unit; the C ABI lives in fourq_amd.hip.
#ifndef FQ_CHAIN
#define FQ_CHAIN 1
#endif
#include "kernels.hip.h"

namespace fq {

namespace {
template <int ALGO, int SRC, bool DH, bool DEFER> int launch(unsigned grid, hipStream_t stream, const LadderArgs& a) {
    hipLaunchKernelGGL((ladder_kernel<ALGO, SRC, DH, DEFER>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return (int)hipGetLastError();
}
template <int ALGO> int launch_algo(int src, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (src == LDS) {
        if (!dh) return launch<ALGO, LDS, false, false>(grid, stream, a);
        return a.proj ? launch<ALGO, LDS, true, true>(grid, stream, a) : launch<ALGO, LDS, true, false>(grid, stream, a);
    }
    if (!dh) return launch<ALGO, PREBUILT, false, false>(grid, stream, a);
    if (!a.proj) return (int)hipErrorInvalidValue;           // the PREBUILT DH ladder always defers normalisation
    return launch<ALGO, PREBUILT, true, true>(grid, stream, a);
}
}  // namespace

int chain_launch_ladder(int algo, int src, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    return algo == ENDO ? launch_algo<ENDO>(src, dh, grid, stream, a) : launch_algo<WINDOWED>(src, dh, grid, stream, a);
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
// The comb's table needs more dynamic LDS than the 64 KB a kernel gets by default: raised once per device, at context creation
// (a launch then only enqueues, so the _dev entry points stay capturable into a graph).
int chain_setup_device() {
    hipError_t e = hipFuncSetAttribute((const void*)comb_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMB_FAST_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)comb_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMB_FAST_LDS_BYTES);
    return (int)e;
}
// One block per CU holds the whole 144 KB table; its width follows the batch so that a small batch still reaches every CU.
int chain_launch_comb(unsigned cus, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, uint4* proj, u32 proj_stride, u32 n) {
    unsigned width = 64;
    while (width < (unsigned)COMB_BLOCK_MAX && (size_t)width * cus < n) width *= 2;
    const unsigned blocks = (n + width - 1) / width, grid = blocks < cus ? blocks : cus;
    if (proj) hipLaunchKernelGGL(comb_kernel<true>, dim3(grid), dim3(width), COMB_FAST_LDS_BYTES, stream, scalars, comb_limbs, out, status, proj, proj_stride, n);
    else hipLaunchKernelGGL(comb_kernel<false>, dim3(grid), dim3(width), COMB_FAST_LDS_BYTES, stream, scalars, comb_limbs, out, status, proj, proj_stride, n);
    return (int)hipGetLastError();
}
int chain_launch_normalize(int k, hipStream_t stream, const uint4* proj, u32 proj_stride, u64* out, uint8_t* status, u32 n) {
    const unsigned grid = ((n + k - 1) / k + BLOCK - 1) / BLOCK;
    if (k == 8) hipLaunchKernelGGL(normalize_kernel<8>, dim3(grid), dim3(BLOCK), 0, stream, proj, proj_stride, out, status, n);
    else if (k == 4) hipLaunchKernelGGL(normalize_kernel<4>, dim3(grid), dim3(BLOCK), 0, stream, proj, proj_stride, out, status, n);
    else if (k == 2) hipLaunchKernelGGL(normalize_kernel<2>, dim3(grid), dim3(BLOCK), 0, stream, proj, proj_stride, out, status, n);
    else hipLaunchKernelGGL(normalize_kernel<1>, dim3(grid), dim3(BLOCK), 0, stream, proj, proj_stride, out, status, n);
    return (int)hipGetLastError();
}

}  // namespace fq
