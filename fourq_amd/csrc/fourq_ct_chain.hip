// Fourth translation unit of libfourq_amd.so: the fixed-base (LDS) ladders and the comb with CONSTANT-TIME table
// selection (FOURQ_CT_SELECT / fourq_ctx_set_ct_select).  Same build flavour as fourq_chain.hip (FQ_CHAIN=1): every
// entry of the shared table is read by every lane at every step (same addresses across the wave: LDS broadcasts).
#ifndef FQ_CHAIN
#define FQ_CHAIN 1
#endif
#include "kernels.hip.h"

namespace fq {

namespace {
template <int ALGO> int launch_lds(bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    if (!dh) hipLaunchKernelGGL((ladder_kernel<ALGO, LDS, false, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else if (a.proj) hipLaunchKernelGGL((ladder_kernel<ALGO, LDS, true, true, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((ladder_kernel<ALGO, LDS, true, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return (int)hipGetLastError();
}
}  // namespace

int ct_launch_lds(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a) {
    return algo == ENDO ? launch_lds<ENDO>(dh, grid, stream, a) : launch_lds<WINDOWED>(dh, grid, stream, a);
}
int ct_launch_comb(unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, uint4* proj, u32 proj_stride, u32 n) {
    constexpr size_t lds_bytes = (size_t)CombScan::POINTS * COMB_LDS_U32 * sizeof(u32);       // 11.25 KB: four blocks per CU
    if (proj) hipLaunchKernelGGL((comb_kernel<true, true>), dim3(grid), dim3(BLOCK), lds_bytes, stream, scalars, comb_limbs, out, status, proj, proj_stride, n);
    else hipLaunchKernelGGL((comb_kernel<false, true>), dim3(grid), dim3(BLOCK), lds_bytes, stream, scalars, comb_limbs, out, status, proj, proj_stride, n);
    return (int)hipGetLastError();
}

int ct_launch_split_counts(hipStream_t stream, u32* counts, u32 lanes, u32 limit) {
    hipLaunchKernelGGL(split_counts_kernel<0>, dim3(1), dim3(64), 0, stream, counts, lanes, limit);
    return (int)hipGetLastError();
}
// the overflow ids' tables (whole entries, working limbs: what ScanMem reads), then fixed-base elements + overflow ids in one launch
int ct_launch_mixed_tail(unsigned prep_grid, unsigned tail_grid, hipStream_t stream, const LadderArgs& a, const u32* fix_list, const u32* var_list, u32* counts,
                         u32* over_scratch, u32 lanes, u32 limit) {
    (void)lanes; (void)limit;
    LadderArgs ap = a;
    ap.index = var_list; ap.base = 0; ap.base_dev = counts + 4; ap.n_dev = counts + 5; ap.scratch = over_scratch;
    hipLaunchKernelGGL((prep_kernel<ENDO, false, LimbSlots>), dim3(prep_grid), dim3(BLOCK), 0, stream, ap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mixed_ct_tail_kernel<0>, dim3(tail_grid), dim3(BLOCK), 0, stream, a, fix_list, var_list, counts, over_scratch);
    return (int)hipGetLastError();
}

}  // namespace fq
