// libfourq_amd.so -- first translation unit: the fused variable-base kernels, the small service kernels
// (tables, partition, wire format, primitives) and the C ABI declared in include/fourq_amd.h.  The
// fixed-base, two-kernel-route and comb kernels live in fourq_chain.hip; shared device code in kernels.hip.h.
//
// Kernel design (gfx950): one wavefront lane owns one (scalar, point) pair for the whole scalar
// multiplication.  Lanes never communicate; a workgroup is 256 lanes; the grid is sized to the
// number of resident lanes and strides over the batch.  (The exception: variable-base batches of at most
// half a generation, and remainders past whole generations, run TWO lanes per element -- real parts in even
// lanes, imaginary parts in odd lanes, halves exchanged by DPP -- at 0.66 of the latency: pair.hip.h.)
//
//   variable base : the lane builds its own 8-entry R2 table (table_endo / table_windowed): N, D of every entry into
//                   a 768-byte slot of HBM scratch (96 bytes per entry = two memory sectors), E, F into its rows of
//                   LDS; each ladder step then gathers the coordinates of the entry its digit selects (wavefront-
//                   level gather, one entry per lane) a whole doubling ahead of their use.  Large batches build the
//                   tables in a kernel of their own into packed 128-byte entries and ladder at four waves per SIMD.
//   fixed base    : the 8-entry table is staged once per workgroup into LDS (padded to dodge bank
//                   conflicts) and gathered from there.
//   selection     : as the reference's selectpt (curve4q.py:193-206): the sign of a digit is applied by masked selects
//                   (N/D exchanged by one v_bitop3_b32 per limb, F negated by a two-op conditional negation) and never
//                   becomes an address; the table index is a per-lane address, as in the reference (curve4q.py:232,
//                   :440).  With FOURQ_CT_SELECT / fourq_ctx_set_ct_select every step reads the whole table instead
//                   and selects the entry by masks too (fourq_ct_*.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#ifndef FQ_CHAIN
#define FQ_CHAIN 0
#endif
#include "kernels.hip.h"
#include "pipeline_plan.h"

using namespace fq;

namespace {

// packed table (8 x 16 words) -> working limbs (8 x 48 u32)
__global__ void table_unpack_kernel(const u64* packed, u32* limbs, u32* slots) {
    int k = threadIdx.x;
    if (k < 8) {
        const R2 t = load_r2_packed(packed + 16 * k);
        store_r2_limbs(limbs + k * R2_LIMBS, t);
        store_r2<PrebuiltSlots>(slots + k * PrebuiltSlots::ENTRY, t);     // what a mixed batch's per-lane pointer reads
    }
}
// working limbs of one lane's scratch slot -> packed table
__global__ void table_build_kernel(int algo, const u64* p_r1, u32* scratch, u64* packed) {
    if (threadIdx.x != 0) return;
    R1 P = load_r1(p_r1);
    if (algo == ENDO) build_table_endo(P, scratch); else build_table_windowed(P, scratch);
    for (int k = 0; k < 8; k++) store_r2_packed(packed + 16 * k, load_r2_limbs(scratch + k * R2_LIMBS));
}
// Mixed batches: compacts the ids of the variable-base elements of one round (flags[i] != 0) into var_list (any
// order) and records, per element of the round, which scratch slot holds its table: slot_of[i] = rank in var_list,
// or ~0 for a fixed-base element (shared table).  Sixteen flags per lane, a block scan in LDS and one atomic per
// 4 096 elements; one atomic per element (or per wave) serialises on the counter.
constexpr int PART_PER_LANE = 16;
// fix_list (optional; constant-time mode runs the two kinds as two launches): the ids of the fixed-base elements,
// counted in counter[1].
__global__ __launch_bounds__(BLOCK) void partition_kernel(const uint8_t* flags, u32 n, u32 first_id, u32* var_list, u32* slot_of, u32* counter,
                                                          u32* fix_list) {
    __shared__ u32 scan[BLOCK], base, base_fix;
    const u32 t = threadIdx.x;
    const u32 first = (blockIdx.x * BLOCK + t) * PART_PER_LANE;
    const u32 valid = first < n ? (n - first < (u32)PART_PER_LANE ? n - first : (u32)PART_PER_LANE) : 0u;
    uint8_t f[PART_PER_LANE];
    if (valid == PART_PER_LANE && (reinterpret_cast<uintptr_t>(flags) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(flags + first);
        const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int k = 0; k < PART_PER_LANE; k++) f[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    } else {
#pragma unroll
        for (int k = 0; k < PART_PER_LANE; k++) f[k] = (u32)k < valid ? flags[first + k] : (uint8_t)0;
    }
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < PART_PER_LANE; k++) mine += ((u32)k < valid && f[k] != 0) ? 1u : 0u;
    scan[t] = mine;
    __syncthreads();
    for (u32 off = 1; off < BLOCK; off <<= 1) {           // inclusive scan
        const u32 below = t >= off ? scan[t - off] : 0u;
        __syncthreads();
        scan[t] += below;
        __syncthreads();
    }
    if (t == BLOCK - 1) {
        base = atomicAdd(counter, scan[t]);
        if (fix_list) {                                   // fixed-base elements of this block = its valid elements - variable ones
            const u32 block_first = blockIdx.x * BLOCK * PART_PER_LANE;
            const u32 block_valid = block_first < n ? (n - block_first < (u32)(BLOCK * PART_PER_LANE) ? n - block_first : (u32)(BLOCK * PART_PER_LANE)) : 0u;
            base_fix = atomicAdd(counter + 1, block_valid - scan[t]);
        }
    }
    __syncthreads();
    u32 rank = base + scan[t] - mine;
    // fixed-base elements before this lane inside the block: elements before it minus variable ones before it
    u32 rank_fix = fix_list ? base_fix + (t * PART_PER_LANE - (scan[t] - mine)) : 0u;
#pragma unroll
    for (int k = 0; k < PART_PER_LANE; k++) {
        if ((u32)k >= valid) break;
        if (f[k] != 0) { var_list[rank] = first_id + first + k; slot_of[first + k] = rank++; }
        else {
            slot_of[first + k] = ~0u;
            if (fix_list) fix_list[rank_fix++] = first_id + first + k;
        }
    }
}

// One lane per point of the table object (both shapes, recode.hip.h): [2^(e j) (1 + u0 2^d + u1 2^2d + ...)] B by the ordinary
// variable-base MUL_endo, normalised to affine and stored as (x+y, y-x, 2d x y), 12 packed words.
__global__ __launch_bounds__(BLOCK) void comb_table_kernel(const u64* p_r1, u32* scratch, u64* comb) {
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (u32)COMB_POINTS) return;
    u64 m[4];
    if (t < (u32)CombFast::POINTS) comb_point_scalar<CombFast>(t, m); else comb_point_scalar<CombScan>(t - CombFast::POINTS, m);
    R1 P = load_r1(p_r1);
    u32* slot = scratch + (size_t)t * SLOT_U32;
    build_table_endo(P, slot);
    u64 v[4];
    decompose(m, v);
    R1 Q = ladder_endo(recode(v), slot, R2_LIMBS);
    Fe2<1> x, y;
    r1_to_affine(Q, x, y);
    u64* dst = comb + 12 * (size_t)t;
    store_fe2(dst, fe2_add(x, y));
    store_fe2(dst + 4, fe2_sub(y, x));
    store_fe2(dst + 8, fe2_mul(fe2_mul(x, y), fe2_two_d()));
}
__global__ void comb_unpack_kernel(const u64* packed, u32* limbs) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= COMB_POINTS) return;
    for (int c = 0; c < 3; c++) store_fe2_limbs(limbs + k * COMB_ENTRY_U32 + c * COORD_U32, load_fe2(packed + 12 * k + 4 * c));
}
// ---- point compression ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void encode_kernel(const u64* affine, u64* out, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 w[4];
    point_encode(load_fe2(affine + 8 * (size_t)i), load_fe2(affine + 8 * (size_t)i + 4), w);
    uint4* dst = reinterpret_cast<uint4*>(out + 4 * (size_t)i);
    dst[0] = make_uint4((u32)w[0], (u32)(w[0] >> 32), (u32)w[1], (u32)(w[1] >> 32));
    dst[1] = make_uint4((u32)w[2], (u32)(w[2] >> 32), (u32)w[3], (u32)(w[3] >> 32));
}
__global__ __launch_bounds__(BLOCK) void decode_kernel(const u64* in, u64* affine, uint8_t* status, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 w[4];
    load_scalar(in + 4 * (size_t)i, w);
    Fe2<1> x, y;
    int st = point_decode(w, x, y);
    u64 o[8];
    store_fe2_words(o, x); store_fe2_words(o + 4, y);
    uint4* dst = reinterpret_cast<uint4*>(affine + 8 * (size_t)i);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u64 lo = st ? 0 : o[2 * k], hi = st ? 0 : o[2 * k + 1];
        dst[k] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    }
    status[i] = (uint8_t)st;
}

// Protocol step, last stage: encode the shared point and merge the per-stage verdicts into one status byte
//   0 ok | FOURQ_DH_* of the DH stage | 16 + FOURQ_DECODE_* of the decode stage (which wins); out32 is zero unless 0.
__global__ __launch_bounds__(BLOCK) void encode_status_kernel(const u64* affine, const uint8_t* st_decode, const uint8_t* st_dh,
                                                              u64* out, uint8_t* status, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t sd = st_decode[i], sh = st_dh ? st_dh[i] : 0;        // st_dh == NULL: a stage that cannot fail (MUL_*)
    const uint8_t st = sd ? (uint8_t)(16 + sd) : sh;
    u64 w[4];
    point_encode(load_fe2(affine + 8 * (size_t)i), load_fe2(affine + 8 * (size_t)i + 4), w);
    if (st) w[0] = w[1] = w[2] = w[3] = 0;
    uint4* dst = reinterpret_cast<uint4*>(out + 4 * (size_t)i);
    dst[0] = make_uint4((u32)w[0], (u32)(w[0] >> 32), (u32)w[1], (u32)(w[1] >> 32));
    dst[1] = make_uint4((u32)w[2], (u32)(w[2] >> 32), (u32)w[3], (u32)(w[3] >> 32));
    status[i] = st;
}
// one affine point replicated n times (the public base of a batch of exchanges); the point travels as a kernel argument,
// so the call neither copies from pageable host memory nor keeps a host buffer alive behind the caller's back
struct AffineArg { u64 w[FOURQ_AFFINE_WORDS]; };
__global__ __launch_bounds__(BLOCK) void broadcast_point_kernel(AffineArg point, u64* out, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint4* dst = reinterpret_cast<uint4*>(out + 8 * (size_t)i);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = make_uint4((u32)point.w[2 * k], (u32)(point.w[2 * k] >> 32), (u32)point.w[2 * k + 1], (u32)(point.w[2 * k + 1] >> 32));
}
// MUL_* with affine I/O ("affine-only I/O variant"): AffineToR1 in front (curve4q.py:100-101), R1toAffine behind
// (curve4q.py:103-106, one GFp2.inv per element: ~2 % of a MUL_endo's multiply-adds).  The R1 rows in between stay on the device.
__global__ __launch_bounds__(BLOCK) void lift_affine_kernel(const u64* affine, u64* r1, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(affine + 8 * (size_t)i);
    uint4* dst = reinterpret_cast<uint4*>(r1 + 20 * (size_t)i);
    const uint4 x0 = src[0], x1 = src[1], y0 = src[2], y1 = src[3];
    dst[0] = x0; dst[1] = x1; dst[2] = y0; dst[3] = y1;                                  // X, Y
    dst[4] = make_uint4(1, 0, 0, 0); dst[5] = make_uint4(0, 0, 0, 0);                   // Z = 1
    dst[6] = x0; dst[7] = x1; dst[8] = y0; dst[9] = y1;                                  // Ta = X, Tb = Y
}
// R1toAffine (curve4q.py:103-106) behind a MUL_*, K elements per lane with ONE GFp.inv (Montgomery's trick inside a lane, as normalize_kernel
// does for DH batches; fields.py:66-106, :193-199): lane t owns elements t, t + T, ..., t + (K-1) T with T = ceil(n / K).  Z != 0 for every
// point of the curve (the addition law is complete) -- but MUL_* accepts any pair of field elements (curve4q.py never checks), and the
// all-zero point a failed decode is lifted to DOES come out with Z = 0: an element whose norm is zero contributes a 1 to the lane's product,
// so it cannot touch its neighbours, and gets (0, 0) itself -- what conj(0) * 0^(p-2) gives the reference (fields.py:193-199).  A lane's
// slots past the end of the batch redo its first element and store nothing.  K = 1 for chunks within two generations (one wave per SIMD: the chain's latency is the time either
// way), K = 4 beyond (throughput: 0.29 -> 0.12 ms per 2^20 elements).  ENC: the result leaves as a 32-byte encoding + status byte
// (curve4q.py:41-47; 16 + decode status where the input did not decode) instead of 64 bytes of affine words.
template <int K, bool ENC>
__global__ __launch_bounds__(BLOCK) void lower_kernel(const u64* r1, u32 stride, const uint8_t* st_decode, u64* out, uint8_t* status, u32 n) {   // stride: 20 (R1 rows) or 12 ((X, Y, Z) rows)
    const u32 T = (n + K - 1) / K;
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    Fe2<1> z[K];
    Fe<1> nz[K], pre[K];
#pragma clang loop unroll(full)
    for (int j = 0; j < K; j++) {
        const u32 id = t + (u32)j * T, at = id < n ? id : t;
        z[j] = load_fe2(r1 + stride * (size_t)at + 8);
    }
    Fe<1> one;
    one.l[0] = 1; one.l[1] = one.l[2] = one.l[3] = one.l[4] = 0;
#pragma clang loop unroll(full)
    for (int j = 0; j < K; j++) {
        const Fe<1> norm = fe_carry(fe_add(fe_sqr(z[j].re), fe_sqr(z[j].im)));
        nz[j] = (K > 1) ? fe_select(fe_is_zero(norm) ? 0u : ~0u, norm, one) : norm;
        if (j == 0) pre[0] = nz[0]; else pre[j] = fe_mul(pre[j - 1], nz[j]);
    }
    Fe<1> inv = fe_inv(pre[K - 1]);
#pragma clang loop unroll(full)
    for (int j = K - 1; j >= 0; j--) {
        const u32 id = t + (u32)j * T, at = id < n ? id : t;
        const Fe2<1> X = load_fe2(r1 + stride * (size_t)at), Y = load_fe2(r1 + stride * (size_t)at + 4);
        Fe<1> ninv = inv;                                              // 1 / |Z_j|^2
        if (j > 0) { ninv = fe_mul(inv, pre[j - 1]); inv = fe_mul(inv, nz[j]); }
        Fe2<1> zi;
        zi.re = fe_mul(ninv, z[j].re);                                 // conj(Z) / |Z|^2     fields.py:193-199
        zi.im = fe_mul(ninv, fe_neg(z[j].im));
        const Fe2<1> ax = fe2_mul(X, zi), ay = fe2_mul(Y, zi);
        if (id >= n) continue;
        if constexpr (ENC) {
            const uint8_t sd = st_decode[id];
            const uint8_t st = sd ? (uint8_t)(16 + sd) : (uint8_t)0;
            u64 w[4];
            point_encode(ax, ay, w);
            if (st) w[0] = w[1] = w[2] = w[3] = 0;
            uint4* dst = reinterpret_cast<uint4*>(out + 4 * (size_t)id);
            dst[0] = make_uint4((u32)w[0], (u32)(w[0] >> 32), (u32)w[1], (u32)(w[1] >> 32));
            dst[1] = make_uint4((u32)w[2], (u32)(w[2] >> 32), (u32)w[3], (u32)(w[3] >> 32));
            status[id] = st;
        } else {
            u64 o[8];
            store_fe2(o, ax); store_fe2(o + 4, ay);
            uint4* dst = reinterpret_cast<uint4*>(out + 8 * (size_t)id);
#pragma clang loop unroll(full)
            for (int k = 0; k < 4; k++) dst[k] = make_uint4((u32)o[2 * k], (u32)(o[2 * k] >> 32), (u32)o[2 * k + 1], (u32)(o[2 * k + 1] >> 32));
        }
    }
}
// The 32-byte I/O flavour of MUL_* as THREE kernels per chunk instead of five (round 5): decode + AffineToR1 in one (the decoded point
// never exists as an affine row), R1toAffine + encode + status in one.  Same values as decode_kernel -> lift_affine_kernel and
// lower_r1_kernel -> encode_status_kernel: a point that does not decode is lifted as all-zero coordinates with Z = 1, its product is
// computed like any other and its output zeroed by the status (lower_kernel<K, true>).
__global__ __launch_bounds__(BLOCK) void decode_lift_kernel(const u64* in, u64* r1, uint8_t* status, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    u64 w[4];
    load_scalar(in + 4 * (size_t)i, w);
    Fe2<1> x, y;
    const int st = point_decode(w, x, y);
    u64 o[8];
    store_fe2_words(o, x); store_fe2_words(o + 4, y);
    uint4 q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 lo = st ? 0 : o[2 * k], hi = st ? 0 : o[2 * k + 1];
        q[k] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    }
    uint4* dst = reinterpret_cast<uint4*>(r1 + 20 * (size_t)i);
    dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3];                          // X, Y
    dst[4] = make_uint4(1, 0, 0, 0); dst[5] = make_uint4(0, 0, 0, 0);                   // Z = 1
    dst[6] = q[0]; dst[7] = q[1]; dst[8] = q[2]; dst[9] = q[3];                          // Ta = X, Tb = Y
    status[i] = (uint8_t)st;
}
// status of a two-stage exchange: the first failure of either half (the second half already zeroed its output)
// Diagnostic (fourq_diag_clock): one wave per block stamps the shader-cycle counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime), sleeps until the latter has advanced by `ticks`, and stamps again: shader cycles per 10 ns = the clock the chip holds
// while whatever ELSE is running runs.  No product kernel carries a stamp.  Every wave leaves: the wait is on a free-running counter and
// bounded by a spin limit besides.
__global__ __launch_bounds__(64) void clock_probe_kernel(u64* stamps, u64 ticks) {
    const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u64 r1 = r0;
    for (u32 spin = 0; spin < (1u << 24) && r1 - r0 < ticks; spin++) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const u64 c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// The bracket form (fourq_diag_clock_begin / _stop / _end): NO probe stays resident beside the work -- a sleeping wave still holds eight
// registers of its SIMD, and a kernel that fills the register file (the LDS-table ladders: 2 x 256 VGPRs per SIMD) then loses a whole
// workgroup slot on that CU: config 3 ran 30 % slower under the round-6 first attempt (profiles/r06_clock_probe.txt).  Instead two launches
// of this kernel ON THE CONTEXT'S STREAM, one before and one after the bracketed work, each wave recording its CU's cycle counter and the
// global 100 MHz counter.  s_memtime is a PER-CU counter (tools/microbench/stamp_coherence.hip: 256 CUs, each coherent within 113 cycles,
// offsets between CUs up to 4 x 10^7), so the host pairs the two launches' stamps CU by CU (key = XCC_ID, and SE / SH / CU of HW_ID).
__global__ __launch_bounds__(64) void clock_stamp_kernel(u64* stamps) {
    u64 t, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
    const u32 xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;        // HW_REG_XCC_ID[3:0]
    const u32 hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) & 0xFF00;      // HW_REG_HW_ID: cu 11:8, sh 12, se 15:13
    if (threadIdx.x == 0) { u64* o = stamps + 3 * (size_t)blockIdx.x; o[0] = (u64)xcc << 16 | hw; o[1] = t; o[2] = r; }
}

__global__ __launch_bounds__(BLOCK) void merge_status_kernel(const uint8_t* first, uint8_t* status, u32 n) {
    u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n && first[i]) status[i] = first[i];
}

// ---- primitives ----------------------------------------------------------------------------------
FQ_DEV Fe<1> ld_fe(const u64* w) { return fe_unpack(w[0], w[1]); }
template <int B> FQ_DEV void st_fe(u64* w, const Fe<B>& a) { fe_canon(a, w[0], w[1]); }
template <int BX, int BY, int BZ> FQ_DEV void st_proj(u64* w, const Proj<BX, BY, BZ>& p) {
    store_fe2(w, p.X); store_fe2(w + 4, p.Y); store_fe2(w + 8, p.Z);
}
FQ_DEV R2s ld_r2s(const u64* w) { return as_signed(load_r2_packed(w)); }

__global__ __launch_bounds__(64) void prim_kernel(int op, const u64* in, u64* out, u32 n, u32 iw, u32 ow) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64* x = in + (size_t)i * iw;
    u64* y = out + (size_t)i * ow;
    switch (op) {
    case FOURQ_FP_ADD: st_fe(y, fe_add(ld_fe(x), ld_fe(x + 2))); break;
    case FOURQ_FP_SUB: st_fe(y, fe_sub(ld_fe(x), ld_fe(x + 2))); break;
    case FOURQ_FP_MUL: st_fe(y, fe_mul(ld_fe(x), ld_fe(x + 2))); break;
    case FOURQ_FP_SQR: st_fe(y, fe_sqr(ld_fe(x))); break;
    case FOURQ_FP_NEG: st_fe(y, fe_neg(ld_fe(x))); break;
    case FOURQ_FP_INV: st_fe(y, fe_inv(fe_carry(ld_fe(x)))); break;
    case FOURQ_FP_INVSQRT: st_fe(y, fe_invsqrt(fe_carry(ld_fe(x)))); break;
    case FOURQ_FP_SELECT:
    case FOURQ_FP2_SELECT: {
        // (mask * c) mod 2^128 = -c mod 2^128 for the reference's mask = 2^512 - 1; bit operations on the raw words,
        // no reduction (the reference's select does none either)
        const u64 m_lo = 0 - x[0], m_hi = ~x[1] + (x[0] == 0 ? 1 : 0);
        const int parts = op == FOURQ_FP_SELECT ? 1 : 2;
        const u64* a = x + 2;
        const u64* b = x + 2 + 2 * parts;
        for (int k = 0; k < parts; k++) {
            y[2 * k] = b[2 * k] ^ (m_lo & (a[2 * k] ^ b[2 * k]));
            y[2 * k + 1] = b[2 * k + 1] ^ (m_hi & (a[2 * k + 1] ^ b[2 * k + 1]));
        }
        break;
    }
    case FOURQ_FP2_ADD: store_fe2(y, fe2_add(load_fe2(x), load_fe2(x + 4))); break;
    case FOURQ_FP2_SUB: store_fe2(y, fe2_sub(load_fe2(x), load_fe2(x + 4))); break;
    case FOURQ_FP2_MUL: store_fe2(y, fe2_mul(load_fe2(x), load_fe2(x + 4))); break;
    case FOURQ_FP2_SQR: store_fe2(y, fe2_sqr(load_fe2(x))); break;
    case FOURQ_FP2_NEG: store_fe2(y, fe2_neg(load_fe2(x))); break;
    case FOURQ_FP2_CONJ: store_fe2(y, fe2_conj(load_fe2(x))); break;
    case FOURQ_FP2_INV: store_fe2(y, fe2_inv(load_fe2(x))); break;
    case FOURQ_PT_DBL: store_r1(y, dbl(load_r1(x))); break;
    case FOURQ_PT_ADD: store_r1(y, add(load_r1(x), ld_r2s(x + 20))); break;
    case FOURQ_PT_ADD_CORE: {
        R2 p = load_r2_packed(x);
        R3 p3; p3.N = widen<2>(p.N); p3.D = widen<3>(p.D); p3.E = p.E; p3.F = p.F;
        store_r1(y, add_core(p3, ld_r2s(x + 16)));
        break;
    }
    case FOURQ_PT_R1TOR2: store_r2_packed(y, r1_to_r2(load_r1(x))); break;
    case FOURQ_PT_R1TOR3: {
        R3 r = r1_to_r3(load_r1(x));
        store_fe2(y, r.N); store_fe2(y + 4, r.D); store_fe2(y + 8, r.E); store_fe2(y + 12, r.F);
        break;
    }
    case FOURQ_PT_R2TOR4: st_proj(y, r2_to_r4(ld_r2s(x))); break;
    case FOURQ_PT_TAU: st_proj(y, tau(load_fe2(x), load_fe2(x + 4), load_fe2(x + 8))); break;
    case FOURQ_PT_TAU_DUAL:
        store_r1(y, tau_dual(widen<2>(load_fe2(x)), widen<2>(load_fe2(x + 4)), widen<2>(load_fe2(x + 8))));
        break;
    case FOURQ_PT_UPSILON:
    case FOURQ_PT_CHI: {
        Proj<1, 2, 1> p; p.X = load_fe2(x); p.Y = widen<2>(load_fe2(x + 4)); p.Z = load_fe2(x + 8);
        if (op == FOURQ_PT_UPSILON) st_proj(y, upsilon(p)); else st_proj(y, chi(p));
        break;
    }
    case FOURQ_PT_PHI: store_r1(y, phi(load_r1(x))); break;
    case FOURQ_PT_PSI: store_r1(y, psi(load_r1(x))); break;
    case FOURQ_PT_ON_CURVE: y[0] = point_on_curve(load_fe2(x), load_fe2(x + 4)) ? 1 : 0; break;
    case FOURQ_PT_COFACTOR392: store_r1(y, clear_cofactor_392(load_fe2(x), load_fe2(x + 4))); break;
    case FOURQ_PT_R1TOAFFINE: {
        Fe2<1> ax, ay;
        r1_to_affine(load_r1(x), ax, ay);
        store_fe2(y, ax); store_fe2(y + 4, ay);
        break;
    }
    case FOURQ_SC_DECOMPOSE: {
        u64 m[4] = { x[0], x[1], x[2], x[3] }, v[4];
        decompose(m, v);
        y[0] = v[0]; y[1] = v[1]; y[2] = v[2]; y[3] = v[3];
        break;
    }
    case FOURQ_SC_RECODE: {
        u64 v[4] = { x[0], x[1], x[2], x[3] };
        EndoDigits e = recode(v);
        y[0] = e.sign; y[1] = e.d[0]; y[2] = e.d[1]; y[3] = e.d[2]; y[4] = e.top;
        break;
    }
    case FOURQ_SC_WINDOWED: {
        u64 m[4] = { x[0], x[1], x[2], x[3] };
        WinScalar w = win_reduce(m);
        uint8_t* b = reinterpret_cast<uint8_t*>(y);
        for (int k = 0; k < 62; k++) b[k] = (uint8_t)win_code_from_window(win_window(w, k));
        b[62] = (uint8_t)win_top_code(w);
        b[63] = 0;
        break;
    }
    default: break;
    }
}

struct PrimShape { int op; size_t in_words, out_words; };
const PrimShape PRIMS[] = {
    { FOURQ_FP_ADD, 4, 2 }, { FOURQ_FP_SUB, 4, 2 }, { FOURQ_FP_MUL, 4, 2 }, { FOURQ_FP_SQR, 4, 2 }, { FOURQ_FP_NEG, 4, 2 }, { FOURQ_FP_INV, 4, 2 }, { FOURQ_FP_INVSQRT, 4, 2 },
    { FOURQ_FP_SELECT, 6, 2 }, { FOURQ_FP2_SELECT, 10, 4 },
    { FOURQ_FP2_ADD, 8, 4 }, { FOURQ_FP2_SUB, 8, 4 }, { FOURQ_FP2_MUL, 8, 4 }, { FOURQ_FP2_SQR, 8, 4 }, { FOURQ_FP2_NEG, 8, 4 },
    { FOURQ_FP2_CONJ, 8, 4 }, { FOURQ_FP2_INV, 8, 4 },
    { FOURQ_PT_DBL, 20, 20 }, { FOURQ_PT_ADD, 36, 20 }, { FOURQ_PT_ADD_CORE, 32, 20 }, { FOURQ_PT_R1TOR2, 20, 16 },
    { FOURQ_PT_R1TOR3, 20, 16 }, { FOURQ_PT_R2TOR4, 16, 12 }, { FOURQ_PT_TAU, 12, 12 }, { FOURQ_PT_TAU_DUAL, 12, 20 },
    { FOURQ_PT_UPSILON, 12, 12 }, { FOURQ_PT_CHI, 12, 12 }, { FOURQ_PT_PHI, 20, 20 }, { FOURQ_PT_PSI, 20, 20 },
    { FOURQ_PT_ON_CURVE, 8, 1 }, { FOURQ_PT_COFACTOR392, 8, 20 }, { FOURQ_PT_R1TOAFFINE, 20, 8 },
    { FOURQ_SC_DECOMPOSE, 4, 4 }, { FOURQ_SC_RECODE, 4, 5 }, { FOURQ_SC_WINDOWED, 4, 8 },
};
const PrimShape* find_prim(int op) {
    for (const PrimShape& p : PRIMS) if (p.op == op) return &p;
    return nullptr;
}

}  // namespace

// ====================================================================================== C ABI
constexpr int PIPE_SLOTS_MAX = 6;
struct fourq_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int cus = 0;
    size_t lanes = 0;              // resident lanes of the fused variable-base kernels (1 wave per SIMD)
    size_t lanes_w4 = 0;           // resident lanes of the 128-VGPR kernels (4 waves per SIMD)
    size_t split_min = 0;          // variable-base batches of at least this many elements take the prep + ladder route
    size_t split_chunk = 0;        // elements per prep + ladder round (<= lanes_w4)
    bool split_all = false;        // FOURQ_SPLIT_ALL=1: route plain MUL_endo through prep + ladder from split_min on (tests)
    size_t split_endo_min = 0;     // plain MUL_endo batches of at least this many elements take prep + ladder (0: never)
    u32* scratch = nullptr;        // per-lane / per-element table slots: the largest of the three users (fourq_ctx_create)
    u32* table_limbs = nullptr;    // 8 x 48: the staged fixed-base table as working limbs
    u32* table_slots = nullptr;    // the same in the PrebuiltSlots layout
    u64* table_packed = nullptr;   // 128 words
    u32* comb_limbs = nullptr;     // FOURQ_COMB_POINTS x 36 working limbs of the staged comb table
    u64* comb_packed = nullptr;    // FOURQ_COMB_POINTS x 12 words
    uint64_t table_shadow[FOURQ_TABLE_WORDS];   // host copies of what table_limbs / comb_limbs currently hold
    uint64_t comb_shadow[FOURQ_COMB_WORDS];
    bool table_staged = false, comb_staged = false;
    bool comb_known = false;           // comb_shadow holds a caller's table (it survives a change of stream; the device copy is staged again from it)
    hipEvent_t shadow_read = nullptr;   // recorded behind every upload from a shadow: a shadow is rewritten only after its last upload has read it
    u32* part_counter = nullptr;   // mixed batches: number of variable-base elements of the current round (device side)
    u32* part_list = nullptr;      // their ids, split_chunk entries
    u32* part_slot = nullptr;      // per element of the round: scratch slot of its table, ~0 = shared table
    u32* part_fix = nullptr;       // constant-time mode: ids of the round's fixed-base elements
    bool ct = false;               // constant-time table selection (FOURQ_CT_SELECT / fourq_ctx_set_ct_select)
    int mixed_queue = -1;          // mixed batches through the persistent work-queue kernel: 1 always, 0 never, -1 = where it measured faster
    size_t quad_max = 0;           // ... and of at most this many, four lanes per element (a quarter generation fills the chip)
    size_t pair_max = 0;           // variable-base batches (and tails past whole generations) of at most this many elements run two lanes per element
    uint4* proj = nullptr;         // deferred normalisation of DH batches: PROJ_PLANES planes of proj_capacity uint4, grown on demand
    size_t proj_capacity = 0;
    int norm_k = -1;               // FOURQ_NORM_K: 0 = always invert per element, 2/4/8 = always batch; -1 = by batch size
    void* stage = nullptr;         // staging for the small host-pointer calls (tables, primitives)
    size_t stage_bytes = 0;
    char* work = nullptr;          // intermediates of the protocol-level calls (decoded points, first-half results)
    size_t work_bytes = 0;
    // host-pointer batches: pipe_slots device slots (and pinned bounce slots for pageable callers) cycled through
    // H2D copy -> kernels -> D2H copy on three streams
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t in_done[PIPE_SLOTS_MAX] = {}, kernels_done[PIPE_SLOTS_MAX] = {}, out_done[PIPE_SLOTS_MAX] = {};
    std::vector<hipEvent_t> ticks;     // timing events around the copies, four per chunk of the call: recorded only under host_timing
    int pipe_slots = 0;            // FOURQ_PIPE_SLOTS (test hook): slots in flight, 2..PIPE_SLOTS_MAX; 0 = 4 when the GPU hands slots on, 3 when the host does
    int pipe_gens = 0;             // FOURQ_PIPE_GENS (test hook): > 0 = that many kernel generations per inner chunk instead of the planned sizes (pipeline_plan.h)
    bool pipe_host_poll = false;   // FOURQ_PIPE_HOST_POLL=1 (test hook): the host's wait for a slot spins on hipEventQuery instead of hipEventSynchronize
    bool pipe_host_wait = false;   // FOURQ_PIPE_HOST_WAIT=1 (test hook): the host waits for a slot's last use before refilling it, as rounds 2-4 did
    bool host_timing = false;      // fourq_ctx_set_host_timing: time the chunk copies with HIP events (h2d_ms / d2h_ms of fourq_host_stats)
    // Measured planner inputs (round 6): every multi-chunk host-array call times ONE middle chunk (six events, `probe_ticks`) and leaves the
    // kernel time per element of its route -- per selection mode -- and the link's rate each way here; the next call of that route is planned
    // with them.  0 = not measured yet: the first call plans with the KT_* guesses (x KT_CT_GUESS in constant-time mode) and 48 GB/s.
    double plan_kt[32][2] = {};
    double plan_link_in = 0, plan_link_out = 0;     // bytes per nanosecond
    hipEvent_t probe_ticks[6] = {};
    bool fused_io = true;          // FOURQ_FUSED_IO=0 (test hook): the affine / encoded flavours of MUL_* always through lift + full R1 rows, as round 5
    int plan_measure = 1;          // FOURQ_PIPE_MEASURE=0 (test hook): plan with the compiled-in guesses only, as round 5 did
    std::vector<float> chunk_stamps;   // under host_timing: six stamps per chunk of the last call, ms since its first event (fourq_ctx_host_chunk_stamps)
    char* pipe_dev = nullptr;      size_t pipe_dev_bytes = 0;
    char* pipe_pin = nullptr;      size_t pipe_pin_bytes = 0;
    u64* diag_stamps = nullptr;    // fourq_diag_clock: 2 x 16 stamps, allocated at its first call
    hipEvent_t diag_mark = nullptr; // fourq_diag_clock: the end of the stream's backlog at the time of the call
    u64* diag_bracket = nullptr;   // fourq_diag_clock_begin / _stop: 2 x DIAG_STAMP_BLOCKS x {CU key, memtime, memrealtime}
    bool diag_open = false;        // the first stamp launch of a bracket has been enqueued
    bool diag_stopped = false;     // ... and the second
    char* zero_copy = nullptr;     // 64 KiB of pinned host memory the kernels of a TINY host call read and write directly (no copy engine)
    fourq_host_stats host_stats = {};
    bool host_bounce = true;       // FOURQ_HOST_BOUNCE=0: hand pageable arrays to hipMemcpyAsync directly (measurement knob)
    bool host_zero_copy = true;    // FOURQ_HOST_ZERO_COPY=0: tiny host calls through hipMemcpyAsync like the others (measurement knob)
    char err[256] = { 0 };
    mutable std::recursive_mutex mu;   // CtxGuard: calls on one context take turns
};

namespace {

int fail(fourq_ctx* c, hipError_t e, const char* what) {
    if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? FOURQ_ERR_NOMEM : FOURQ_ERR_HIP;
}
#define HIP_TRY(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), e_, #call); } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// What every entry point holds for its duration: the context's lock, then the context's device.  The reference's functions are
// pure (curve4q.py), so a drop-in caller may well call them from several threads at once -- fourq_amd.curve4q does so on ONE
// process-wide context -- while a context is one stream, one set of scratch slots and one set of pipeline buffers.  Calls on the
// same context therefore take turns (host-array calls for their whole duration, _dev calls for their enqueue); contexts are
// independent of each other.  Recursive, because host-array entry points are built from the _dev ones.
struct CtxGuard {
    std::unique_lock<std::recursive_mutex> lock;
    DeviceGuard dev;
    explicit CtxGuard(const fourq_ctx* c) : lock(c->mu), dev(c->device) {}
};

int ensure_stage(fourq_ctx* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return FOURQ_OK;
    if (c->stage) { HIP_TRY(c, hipFree(c->stage)); c->stage = nullptr; c->stage_bytes = 0; }
    size_t want = bytes + bytes / 4;
    HIP_TRY(c, hipMalloc(&c->stage, want));
    c->stage_bytes = want;
    return FOURQ_OK;
}

// device arrays are accessed as 16-byte vectors: every array pointer of the _dev API must be 16-byte aligned
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define HIPRC_TRY(c, expr) do { hipError_t e_ = (hipError_t)(expr); if (e_ != hipSuccess) return fail((c), e_, #expr); } while (0)

template <int ALGO, int SRC, bool DH> int launch_ladder(fourq_ctx* c, LadderArgs a) {
    if (a.n == 0) return FOURQ_OK;
    size_t blocks_needed = ((size_t)a.n + BLOCK - 1) / BLOCK;
    size_t blocks_max = (SRC == FUSED ? c->lanes : c->lanes_w4) / BLOCK;
    unsigned grid = (unsigned)(blocks_needed < blocks_max ? blocks_needed : blocks_max);
    a.scratch = c->scratch;
    a.table = c->table_limbs;
    a.table_slots = c->table_slots;
    if (c->ct) {                            // constant-time selection: fourq_ct_fused.hip / fourq_ct_chain.hip
        if (SRC == PREBUILT) return FOURQ_ERR_INVALID;    // the two-kernel route is not taken in this mode
        HIPRC_TRY(c, SRC == FUSED ? ct_launch_fused(ALGO, DH, grid, c->stream, a) : ct_launch_lds(ALGO, DH, grid, c->stream, a));
    } else if (SRC == FUSED) {              // this translation unit's code object (FQ_CHAIN=0)
        hipLaunchKernelGGL((ladder_kernel<ALGO, FUSED, DH>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
        HIP_TRY(c, hipGetLastError());
    } else {                                // fourq_chain.hip's (FQ_CHAIN=1)
        HIPRC_TRY(c, chain_launch_ladder(ALGO, SRC, DH, grid, c->stream, a));
    }
    return FOURQ_OK;
}
// Which variable-base batches take the two-kernel route (prep_kernel + ladder_kernel<PREBUILT>, up to 4 waves per SIMD).
// Measured on MI355X (profiles/r02_split_route.txt, profiles/r03_cliff.txt): MUL_windowed gains 8 % and DH_* 2-3 % at 2^20, and
// both 3-6 % from the first element past one resident generation of the fused kernels.  Plain MUL_endo, whose 64-step ladder
// hardly amortises the second launch, lost 7 % on this route in round 1 (192-byte entries: a round's tables, 486 MB, fell out of
// the Infinity Cache), gained 1-3 % once the entries were packed, and loses 3-4 % again at 2^20 since the fused kernel keeps E, F
// in LDS and runs on signed limbs (5.21 ms against 5.43): it stays fused (FOURQ_SPLIT_ENDO_MIN = smallest batch that would take
// the route, 0 = never).
bool takes_split_route(const fourq_ctx* c, int algo, bool dh, size_t n) {
    if (c->ct) return false;        // constant-time mode keeps the lane's table in registers: fused kernels only
    if (algo == WINDOWED || dh || c->split_all) return n >= c->split_min;
    return c->split_endo_min && n >= c->split_endo_min;
}
// The variable-base kernels with two lanes per element (pair.hip.h): 0.66 of the one-lane kernels' latency for at most half a
// generation of elements.  A batch that small runs on them alone; a batch of q generations + r elements, 0 < r <= pair_max, on the
// fused route runs q fused generations and then the r elements two lanes each, so the element past a generation costs half a
// generation instead of a whole one.  Both selection modes.
// At most quad_max (a QUARTER generation) elements: FOUR lanes per element -- the element's two pairs take every other product of a
// formula level in the ladder steps -- 0.73 of the two-lane latency again (profiles/r03_quadlane.txt).
// FIXED: the staged fixed-base table (c->table_limbs) instead of a table built per element.
template <int ALGO, bool DH, bool FIXED = false> int launch_pair(fourq_ctx* c, LadderArgs a) {
    if (a.n == 0) return FOURQ_OK;
    const bool quad = a.n <= c->quad_max;   // at most a quarter generation: four lanes per element, the ladder steps' products shared between two pairs
    const size_t per_block = BLOCK / (quad ? 4 : 2);
    const size_t blocks = ((size_t)a.n + per_block - 1) / per_block;
    const unsigned grid = (unsigned)(blocks < (size_t)c->cus ? blocks : (size_t)c->cus);
    a.table = c->table_limbs;
    if (c->ct) {                            // fourq_ct_fused.hip: the same kernels with the lane's table scanned at every step
        HIPRC_TRY(c, ct_launch_pair(ALGO, DH, FIXED, quad, grid, c->stream, a));
        return FOURQ_OK;
    }
    if (quad) hipLaunchKernelGGL((pair_kernel<ALGO, DH, false, FIXED, 4>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL((pair_kernel<ALGO, DH, false, FIXED>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
// A mixed batch of at most half a generation: the same kernels with the table chosen per element (kernels.hip.h, pair_kernel<..., MIXED>)
int launch_pair_mixed(fourq_ctx* c, LadderArgs a) {
    const bool quad = a.n <= c->quad_max;
    const size_t per_block = BLOCK / (quad ? 4 : 2);
    const size_t blocks = ((size_t)a.n + per_block - 1) / per_block;
    const unsigned grid = (unsigned)(blocks < (size_t)c->cus ? blocks : (size_t)c->cus);
    a.table = c->table_limbs;
    if (c->ct) {
        HIPRC_TRY(c, ct_launch_pair_mixed(quad, grid, c->stream, a));
        return FOURQ_OK;
    }
    if (quad) hipLaunchKernelGGL((pair_kernel<ENDO, false, false, false, 4, true>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL((pair_kernel<ENDO, false, false, false, 2, true>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
// fixed-base batches of at most half a generation: the same two-lanes-per-element kernels on the caller's table (the LDS ladders
// of fourq_chain.hip take over above that: one lane per element, up to four waves per SIMD)
bool fixed_takes_pair(const fourq_ctx* c, size_t n) { return c->pair_max && n <= c->pair_max; }
// The route of a variable-base batch.  PAIR_TAIL: whole fused generations, then the remainder two lanes per element (also a batch
// that is nothing but such a remainder).  Below two generations it beats the two-kernel route for every entry point (65 792
// elements: DH_endo 0.69 against 0.74 ms, MUL_windowed 1.10 against 1.19); from two generations on the two-kernel route's four
// wave slots per SIMD absorb a remainder by themselves.
enum Route { ROUTE_FUSED, ROUTE_PAIR_TAIL, ROUTE_SPLIT };
Route variable_route(const fourq_ctx* c, int algo, bool dh, size_t n, bool indexed) {
    const size_t tail = n % c->lanes;
    const bool pair_tail = !indexed && c->pair_max && tail != 0 && tail <= c->pair_max;
    if (pair_tail && n < 2 * c->lanes) return ROUTE_PAIR_TAIL;
    if (takes_split_route(c, algo, dh, n)) return ROUTE_SPLIT;
    return pair_tail ? ROUTE_PAIR_TAIL : ROUTE_FUSED;
}
template <int ALGO, bool DH> int launch_variable(fourq_ctx* c, LadderArgs a) {
    const Route route = variable_route(c, ALGO, DH, a.n, a.index != nullptr);
    if (route == ROUTE_PAIR_TAIL) {
        const u32 tail = (u32)(a.n % c->lanes);
        LadderArgs whole = a, rest = a;
        whole.n = a.n - tail;
        rest.base = a.base + whole.n; rest.n = tail;
        int rc = whole.n ? launch_ladder<ALGO, FUSED, DH>(c, whole) : FOURQ_OK;
        return rc ? rc : launch_pair<ALGO, DH>(c, rest);
    }
    if (route == ROUTE_FUSED) return launch_ladder<ALGO, FUSED, DH>(c, a);
    const u32 total = a.n;
    for (u32 off = 0; off < total; off += (u32)c->split_chunk) {
        LadderArgs part = a;
        part.base = a.base + off;
        part.n = total - off < (u32)c->split_chunk ? total - off : (u32)c->split_chunk;
        part.scratch = c->scratch;
        HIPRC_TRY(c, chain_launch_prep(ALGO, DH, (part.n + BLOCK - 1) / BLOCK, c->stream, part));
        int rc = launch_ladder<ALGO, PREBUILT, DH>(c, part);
        if (rc) return rc;
    }
    return FOURQ_OK;
}

// Mixed batches: the persistent work-queue kernel where it measured faster than compaction + prep + pointer-selected ladder
// (profiles/r03_mixed_queue.txt).  A round that fits one generation of resident lanes is one launch instead of three (2^16
// elements: 0.39 ms against 0.42, constant-time mode 0.48 against 0.81).  Past that the queue hands a wave a third item as soon
// as the kinds do not split evenly -- BASELINE config 5's 65 550 variable-base elements of 2^17: 0.88 ms against 0.65 -- where the
// two-kernel route's ladder still has three free wave slots per SIMD.  FOURQ_MIXED_QUEUE=0|1 forces either.
// Mixed rounds through the persistent work-queue kernel (BASELINE config 5's mechanism): every round in the default mode -- since its items
// run over the two lists laid end to end it beats compaction + prep + pointer-selected ladder at every size (config 5: 0.609 against 0.630 ms,
// 2^20 elements 4.40 against 4.53, profiles/r04_mixed_queue.txt) -- and rounds of at most one generation in constant-time mode, where the
// fused kernel + mixed_ct_tail_kernel pair is as fast for larger ones (0.800 both).
bool mixed_queue_default(const fourq_ctx* c, size_t round) { return !c->ct || round <= c->lanes; }

// DH outputs are affine: from two resident generations of lanes upwards each lane meets several elements, and
// the inversions of K of them are merged into one (normalize_kernel).  Returns K (0: invert per element).
int normalize_group(const fourq_ctx* c, size_t n) {
    if (c->norm_k >= 0) return n >= (size_t)c->norm_k ? c->norm_k : 0;
    for (int k = 8; k >= 2; k >>= 1) if (n >= (size_t)k * c->lanes) return k;
    return 0;
}
int ensure_proj(fourq_ctx* c, size_t n) {
    if (n <= c->proj_capacity) return FOURQ_OK;
    if (c->proj) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(c->proj)); c->proj = nullptr; c->proj_capacity = 0; }
    size_t want = n + n / 4;
    HIP_TRY(c, hipMalloc(&c->proj, want * PROJ_PLANES * sizeof(uint4)));
    c->proj_capacity = want;
    return FOURQ_OK;
}

// The working-limb copy of the caller's fixed-base table stays staged between calls: a call with the same 1 KiB
// (compared on the host) skips the copy and the unpack launch.
// The uploads read the context's shadow copies asynchronously.  HIP stages a copy from pageable memory before hipMemcpyAsync
// returns, but the API does not promise it: the event makes "the previous upload has read its shadow" explicit before a shadow
// is overwritten (only when the caller's table changes, i.e. never in a steady-state or captured sequence).
int shadow_free(fourq_ctx* c) {
    HIP_TRY(c, hipEventSynchronize(c->shadow_read));
    return FOURQ_OK;
}
// A table must be staged OUTSIDE a stream capture: the upload would be captured reading the context's mutable shadow at
// replay time, and shadow_read would become a captured event that no later hipEventSynchronize accepts.  A call that finds
// its table staged captures fine (it enqueues kernels only); one that would have to stage is refused with a message.
int staging_allowed(fourq_ctx* c) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (c->stream && hipStreamIsCapturing(c->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        snprintf(c->err, sizeof c->err, "a table would have to be staged while the stream is being captured: stage it before "
                 "the capture (one call with the same table, or fourq_comb_stage), then capture");
        return FOURQ_ERR_HIP;                                 // what the runtime itself would answer (hipErrorStreamCaptureUnsupported), with the reason
    }
    return FOURQ_OK;
}
int stage_table(fourq_ctx* c, const uint64_t* table_host) {
    if (c->table_staged && memcmp(c->table_shadow, table_host, sizeof c->table_shadow) == 0) return FOURQ_OK;
    if (int rc = staging_allowed(c)) return rc;
    c->table_staged = false;
    if (int rc = shadow_free(c)) return rc;
    memcpy(c->table_shadow, table_host, sizeof c->table_shadow);
    HIP_TRY(c, hipMemcpyAsync(c->table_packed, c->table_shadow, FOURQ_TABLE_WORDS * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->shadow_read, c->stream));
    hipLaunchKernelGGL(table_unpack_kernel, dim3(1), dim3(64), 0, c->stream, c->table_packed, c->table_limbs, c->table_slots);
    HIP_TRY(c, hipGetLastError());
    c->table_staged = true;
    return FOURQ_OK;
}

int mul_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
            const u32* index, size_t n, u32 io = 0) {
    if (!c || !scalars || !out || (!points && !table) || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(out) || !aligned16(points)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    LadderArgs a = {};
    a.scalars = scalars; a.points = points; a.out = out; a.index = index; a.n = (u32)n;
    if (io) {                                       // only the fused one-lane kernels read LadderArgs::io: the caller has checked the route (fused_io)
        if (!points || index || variable_route(c, algo, false, n, false) != ROUTE_FUSED) return FOURQ_ERR_INVALID;
        a.io = io;
    }
    if (points) return algo == ENDO ? launch_variable<ENDO, false>(c, a) : launch_variable<WINDOWED, false>(c, a);
    int rc = stage_table(c, table);
    if (rc) return rc;
    if (!index && fixed_takes_pair(c, n)) return algo == ENDO ? launch_pair<ENDO, false, true>(c, a) : launch_pair<WINDOWED, false, true>(c, a);
    return algo == ENDO ? launch_ladder<ENDO, LDS, false>(c, a) : launch_ladder<WINDOWED, LDS, false>(c, a);
}

int dh_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
           uint8_t* status, size_t n) {
    if (!c || !scalars || !points || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(points) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    LadderArgs a = {};
    a.scalars = scalars; a.points = points; a.out = out; a.status = status; a.n = (u32)n;
    if (table && fixed_takes_pair(c, n)) {         // small fixed-base DH batches: two lanes per element, inversion in the kernel
        int rc = stage_table(c, table);
        if (rc) return rc;
        return algo == ENDO ? launch_pair<ENDO, true, true>(c, a) : launch_pair<WINDOWED, true, true>(c, a);
    }
    int group = normalize_group(c, n);
    if (!table) {                                  // the prep + ladder route always defers; the fused kernels from two generations on (K = 2, 4, 8), except
        const Route route = variable_route(c, algo, true, n, false);     // with a two-lane tail, whose kernels invert in place
        group = route == ROUTE_SPLIT ? (group ? group : 1) : (route == ROUTE_FUSED && FQ_FUSED_DEFER ? group : 0);
    }
    int rc = group ? ensure_proj(c, n) : FOURQ_OK;
    if (rc) return rc;
    a.proj = group ? c->proj : nullptr;
    a.proj_stride = (u32)c->proj_capacity;
    if (!table) {
        rc = algo == ENDO ? launch_variable<ENDO, true>(c, a) : launch_variable<WINDOWED, true>(c, a);
    } else {
        rc = stage_table(c, table);
        if (rc) return rc;
        rc = algo == ENDO ? launch_ladder<ENDO, LDS, true>(c, a) : launch_ladder<WINDOWED, LDS, true>(c, a);
    }
    if (rc || !group) return rc;
    HIPRC_TRY(c, chain_launch_normalize(group, c->stream, c->proj, (u32)c->proj_capacity, out, status, (u32)n));
    return FOURQ_OK;
}

// ---- host-pointer batches ---------------------------------------------------------------------------------
// The caller's arrays are cut into chunks of whole kernel generations; chunk k uses slot k mod pipe_slots and goes
// H2D copy (stream copy_in) -> kernels (the context's stream) -> D2H copy (stream copy_out), the three stages of
// consecutive chunks overlapping.  Arrays in pinned host memory (fourq_host_alloc, hipHostMalloc, torch pin_memory)
// are copied by DMA straight from / to the caller's buffer; pageable arrays go through pinned bounce slots filled
// and drained by a few host threads (a single memcpy stream would be slower than the link).
//
// Round 5.  What a call costs beyond its kernels is (a) the first chunk's copy in and the last chunk's copy out, which
// nothing can overlap, and (b) whatever keeps the kernel stream waiting between chunks.  So: the FIRST and the LAST chunk are one generation
// (the smallest unit that fills the chip), the chunks in between `pipe_gens` generations (fewer stream hops per element); a slot is handed
// on ON THE GPU -- the copy-in stream waits for the event behind the slot's previous copy-out -- so the host enqueues every chunk without
// ever blocking and the queues never run dry because the host was asleep in hipEventSynchronize; and the six timing events per chunk
// are recorded only when the caller asked for copy durations (fourq_ctx_set_host_timing).  Pageable callers keep the host-side hand-over:
// their bounce slots are filled and drained by the host anyway.
constexpr int PIPE_MAX_ARRAYS = 4;
struct PipeArray {
    const char* src;     // input array (host) or NULL
    char* dst;           // output array (host) or NULL
    size_t stride;       // bytes per element
};
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool is_pinned(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}
void host_copy(char* dst, const char* src, size_t bytes) {
    constexpr size_t SLICE = 2u << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t threads = bytes / SLICE;
    if (threads > 6) threads = 6;
    if (hw && threads > hw) threads = hw;
    if (threads <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> pool;
    const size_t per = (bytes / threads + 63) & ~(size_t)63;
    size_t done_to = per;                      // [0, per) is this thread's slice; helpers take the rest
    try {
        pool.reserve(threads - 1);
        for (size_t t = 1; t < threads; t++) {
            const size_t lo = t * per, hi = (t + 1 == threads) ? bytes : (t + 1) * per;
            pool.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
            done_to = hi;
        }
    } catch (...) {                            // no more threads to be had: nothing may cross the C ABI; copy the remainder here
        done_to = pool.empty() ? per : done_to;
    }
    memcpy(dst, src, per);
    for (auto& th : pool) th.join();
    if (done_to < bytes) memcpy(dst + done_to, src + done_to, bytes - done_to);
}
int grow(fourq_ctx* c, char** buf, size_t* have, size_t want, bool pinned) {
    if (want <= *have) return FOURQ_OK;
    if (*buf) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, pinned ? hipHostFree(*buf) : hipFree(*buf));
        *buf = nullptr; *have = 0;
    }
    void* p = nullptr;
    HIP_TRY(c, pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want));
    *buf = (char*)p; *have = want;
    return FOURQ_OK;
}
int ensure_work(fourq_ctx* c, size_t bytes) { return grow(c, &c->work, &c->work_bytes, bytes + bytes / 4, false); }
// intermediates of the protocol-level calls for n elements (decoded keys / first-half results, their status bytes)
size_t dh_bytes_work_bytes(size_t n) { return 2 * n * 64 + 2 * align256(n); }
size_t exchange_work_bytes(size_t n) { return 2 * n * 64 + align256(n); }
size_t mul_affine_work_bytes(size_t n) { return 2 * n * 160 + n * 64 + align256(n); }      // R1 in, R1 out, decoded points, decode status

int ensure_ticks(fourq_ctx* c, size_t count) {
    while (c->ticks.size() < count) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreate(&e));
        try { c->ticks.push_back(e); } catch (...) { (void)hipEventDestroy(e); return FOURQ_ERR_NOMEM; }
    }
    return FOURQ_OK;
}

using ChunkLaunch = std::function<int(char* const* in_dev, char* const* out_dev, size_t m)>;

// Which call this is, for the measured planner inputs, and the guess of its kernel time per element for the context's first call of it
enum PipeRouteId { PR_MUL_VAR = 0, PR_MUL_FIX = 2, PR_DH_VAR = 4, PR_DH_FIX = 6, PR_MIXED = 8, PR_COMB = 9, PR_ENCODE = 10, PR_DECODE = 11, PR_DHB_VAR = 12,
                   PR_DHB_FIX = 14, PR_AFF = 16, PR_BYTES = 18, PR_EXCH = 20, PR_EXCH_COMB = 22, PR_COUNT = 23 };       // + algo (0 / 1) where two follow each other
struct PipeRoute { int id; double kt_guess; };
constexpr double KT_CT_GUESS = 1.3;                 // constant-time selection: x 1.07 - 1.5 by route until the context has measured it
using PipeReserve = std::function<int(size_t big)>; // sizes the context's intermediates for the largest chunk BEFORE the first chunk is enqueued
int run_pipeline_inner(fourq_ctx* c, const PipeArray* in, int n_in, const PipeArray* out, int n_out, size_t n, size_t chunk, PipeRoute route, const ChunkLaunch& launch, size_t chunk_bounce, const PipeReserve& reserve);
// `chunk`: elements of one kernel generation of the route the call takes; `ns_per_elem`: that route's kernel time per element (the
// KT_* constants below: what sizes the chunks is the ratio of kernel time to copy time, pipeline_plan.h).  `chunk_bounce` (0 = `chunk`):
// the uniform chunk of a call from PAGEABLE arrays -- its pace is the host's bounce copies (and, where the call allocates its result, the
// page faults of 160 fresh bytes per element), not the link: such calls keep the chunk shape they were measured with in rounds 2-4
// (cfg3's call, result array reused: 10.3-10.4 ms at 2^20 then and now; profiles/r02_host_api.txt, profiles/r05_pipeline.txt).
int run_pipeline(fourq_ctx* c, const PipeArray* in, int n_in, const PipeArray* out, int n_out, size_t n, size_t chunk, PipeRoute route, const ChunkLaunch& launch,
                 size_t chunk_bounce = 0, const PipeReserve& reserve = nullptr) {
    const int rc = run_pipeline_inner(c, in, n_in, out, n_out, n, chunk, route, launch, chunk_bounce, reserve);
    if (rc != FOURQ_OK) {                      // a chunk failed half way: nothing of this call may still be in flight when the caller
        (void)hipStreamSynchronize(c->copy_in);    // gets its buffers (and the context its slots) back
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->copy_out);
    }
    return rc;
}
int run_pipeline_inner(fourq_ctx* c, const PipeArray* in, int n_in, const PipeArray* out, int n_out, size_t n, size_t chunk, PipeRoute route, const ChunkLaunch& launch, size_t chunk_bounce, const PipeReserve& reserve) {
    if (n_in > PIPE_MAX_ARRAYS || n_out > PIPE_MAX_ARRAYS || chunk == 0 || route.id < 0 || route.id >= 32) return FOURQ_ERR_INVALID;
    bool is_pin_in[PIPE_MAX_ARRAYS], is_pin_out[PIPE_MAX_ARRAYS], any_pageable = false;
    for (int i = 0; i < n_in; i++) { is_pin_in[i] = is_pinned(in[i].src); any_pageable |= !is_pin_in[i]; }
    for (int i = 0; i < n_out; i++) { is_pin_out[i] = is_pinned(out[i].dst); any_pageable |= !is_pin_out[i]; }
    const bool uniform = any_pageable && c->host_bounce && chunk_bounce > chunk && n > chunk_bounce;
    if (uniform) chunk = chunk_bounce;
    if (chunk > n) chunk = n;
    size_t off_in[PIPE_MAX_ARRAYS], off_out[PIPE_MAX_ARRAYS], slot = 0;
    bool pin_in[PIPE_MAX_ARRAYS], pin_out[PIPE_MAX_ARRAYS], bounce = false;
    const size_t chunks = (n + chunk - 1) / chunk;
    // "direct" = hipMemcpyAsync on the caller's array.  Pinned arrays always; pageable ones when the batch is a single
    // chunk (nothing to overlap with, and the runtime's own staging of a pageable copy beats a bounce pass: 0.92 vs
    // 1.29 ms at 2^16 variable-base elements), through the bounce slots otherwise (8.5 vs 13.9 ms at 2^20).
    const bool direct_pageable = !c->host_bounce || chunks == 1;
    fourq_host_stats st = {};
    st.chunks = (uint32_t)chunks;
    st.pinned_in = st.pinned_out = 1;
    for (int i = 0; i < n_in; i++) {
        const bool pinned = is_pin_in[i];
        off_in[i] = slot; slot += align256(chunk * in[i].stride); pin_in[i] = pinned || direct_pageable; bounce |= !pin_in[i];
        st.pinned_in &= pinned ? 1 : 0;
    }
    for (int i = 0; i < n_out; i++) {
        const bool pinned = is_pin_out[i];
        off_out[i] = slot; slot += align256(chunk * out[i].stride); pin_out[i] = pinned || direct_pageable; bounce |= !pin_out[i];
        st.pinned_out &= pinned ? 1 : 0;
    }
    // A single chunk (down to the reference-shaped call, a batch of one): copies and kernels in order on the context's own stream.
    // Nothing can overlap, so the three-stream choreography below would only add its cross-stream event hops (50 us) to the call.
    // The four timing events around the copies are recorded only under fourq_ctx_set_host_timing; otherwise copy durations read 0.
    // Tiny calls (the batch of one above all): the arrays fit 64 KiB of pinned host memory that the kernels read and write in
    // place -- two CPU memcpys of a few hundred bytes instead of three trips through the copy engine (profiles/r03_single_call.txt).
    constexpr size_t ZERO_COPY_BYTES = 64u << 10;
    if (chunks == 1 && slot <= ZERO_COPY_BYTES && c->host_zero_copy) {
        if (!c->zero_copy) {
            void* zp = nullptr;
            HIP_TRY(c, hipHostMalloc(&zp, ZERO_COPY_BYTES, hipHostMallocDefault));
            c->zero_copy = (char*)zp;
        }
        char *din[PIPE_MAX_ARRAYS], *dout[PIPE_MAX_ARRAYS];
        for (int i = 0; i < n_in; i++) {
            din[i] = c->zero_copy + off_in[i];
            memcpy(din[i], in[i].src, n * in[i].stride);              // a CPU copy: no h2d / d2h bytes or times are reported for it
        }
        for (int i = 0; i < n_out; i++) dout[i] = c->zero_copy + off_out[i];
        if (int rc = launch(din, dout, n)) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < n_out; i++) {
            memcpy(out[i].dst, dout[i], n * out[i].stride);
        }
        c->host_stats = st;
        return FOURQ_OK;
    }
    if (chunks == 1) {
        int rc = grow(c, &c->pipe_dev, &c->pipe_dev_bytes, slot, false);
        if (rc) return rc;
        const bool timed = c->host_timing;
        if (timed && (rc = ensure_ticks(c, 4))) return rc;
        char *din[PIPE_MAX_ARRAYS], *dout[PIPE_MAX_ARRAYS];
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[0], c->stream));
        for (int i = 0; i < n_in; i++) {
            din[i] = c->pipe_dev + off_in[i];
            HIP_TRY(c, hipMemcpyAsync(din[i], in[i].src, n * in[i].stride, hipMemcpyHostToDevice, c->stream));
            st.h2d_bytes += n * in[i].stride;
        }
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[1], c->stream));
        for (int i = 0; i < n_out; i++) dout[i] = c->pipe_dev + off_out[i];
        if ((rc = launch(din, dout, n))) return rc;
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[2], c->stream));
        for (int i = 0; i < n_out; i++) {
            HIP_TRY(c, hipMemcpyAsync(out[i].dst, dout[i], n * out[i].stride, hipMemcpyDeviceToHost, c->stream));
            st.d2h_bytes += n * out[i].stride;
        }
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[3], c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (timed) {
            float ms = 0;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[0], c->ticks[1]));
            st.h2d_ms = ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[2], c->ticks[3]));
            st.d2h_ms = ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[1], c->ticks[2]));
            st.kernels_ms = st.kernels_span_ms = ms;
        }
        c->host_stats = st;
        return FOURQ_OK;
    }
    // The chunks of this call (pipeline_plan.h): one generation first and last, growing in between as far as the copies' lead allows
    size_t bytes_in = 0, bytes_out = 0;
    for (int i = 0; i < n_in; i++) bytes_in += in[i].stride;
    for (int i = 0; i < n_out; i++) bytes_out += out[i].stride;
    using fq_plan::Piece;
    std::vector<Piece> plan;
    // what the plan is priced with: this context's own measurements of this route in this selection mode, or the first call's guesses
    const int mode = c->ct ? 1 : 0;
    fq_plan::Rates rates;
    const bool measured = c->plan_measure && c->plan_kt[route.id][mode] > 0;
    rates.ns_per_elem = measured ? c->plan_kt[route.id][mode] : route.kt_guess * (c->ct ? KT_CT_GUESS : 1.0);
    rates.link_in = c->plan_measure && c->plan_link_in > 0 ? c->plan_link_in : fq_plan::LINK_BYTES_PER_NS;
    rates.link_out = c->plan_measure && c->plan_link_out > 0 ? c->plan_link_out : fq_plan::LINK_BYTES_PER_NS;
    st.planned_kernel_ns_per_elem = rates.ns_per_elem; st.planned_link_in_gbs = rates.link_in; st.planned_link_out_gbs = rates.link_out;
    st.planned_from_measurement = measured ? 1 : 0;
    try { plan = fq_plan::plan_pieces(n, chunk, bytes_in, bytes_out, rates, (bounce && !c->pipe_gens) ? 1 : c->pipe_gens); } catch (...) { return FOURQ_ERR_NOMEM; }
    const size_t pieces = plan.size();
    size_t big = 0;
    for (const Piece& pc : plan) if (pc.m > big) big = pc.m;
    // planned chunks grow within a call; the context's intermediates (work, proj) are sized ONCE for the largest, so that no
    // launch() inside the loop reaches grow() -- which would drain the kernel stream and block the host in mid-pipeline
    if (reserve) { if (int rrc = reserve(big)) return rrc; }
    if (big != chunk) {                                                 // the slot layout above was made for `chunk` elements: redo it for `big`
        slot = 0;
        for (int i = 0; i < n_in; i++) { off_in[i] = slot; slot += align256(big * in[i].stride); }
        for (int i = 0; i < n_out; i++) { off_out[i] = slot; slot += align256(big * out[i].stride); }
    }
    st.chunks = (uint32_t)pieces;
    // who hands a slot on: the GPU when every array is copied directly, the host when it has bounce slots to fill and drain
    const bool host_wait = bounce || c->pipe_host_wait;
    // Slots in flight: four -- with the GPU's hand-over, and with the host's when it fills and drains bounce slots anyway (pageable callers:
    // raw R1 7.28 against 7.78 ms at 2^20 with three, the other formats unchanged, profiles/r06_pipeline.txt).  The host's hand-over on PINNED arrays
    // (FOURQ_PIPE_HOST_WAIT, a test hook) keeps three: with four or more the copy out of chunk slots + 2 -- the one the host is waiting for
    // with nothing left to enqueue -- makes almost no progress while the three chunks queued behind it run their kernels and completes only
    // once they are done (1.66 ms instead of 0.42; per-chunk stamps in the same file; a spinning hipEventQuery instead of the blocking wait
    // changes nothing), so the slot it frees comes 0.6 ms late.  The device copy out is a blit kernel on this runtime; handing slots on
    // ON THE GPU never leaves that queue without a successor, and does not show it.
    const int want_slots = c->pipe_slots ? c->pipe_slots : ((host_wait && !bounce) ? 3 : 4);
    const int slots = pieces < (size_t)want_slots ? (int)pieces : want_slots;
    int rc = grow(c, &c->pipe_dev, &c->pipe_dev_bytes, slot * slots, false);
    if (rc) return rc;
    if (bounce && (rc = grow(c, &c->pipe_pin, &c->pipe_pin_bytes, slot * slots, true))) return rc;
    // timing (on request): six events per chunk -- around its copy in, its kernels, its copy out -- for the first TIMED_MAX chunks; a call of
    // more chunks reports their copy durations scaled to the call's bytes (a 2^28-element call has 4 096 chunks: it does not get 24 576 events)
    constexpr size_t TIMED_MAX = 256, TK = 6;
    const size_t timed_pieces = c->host_timing ? (pieces < TIMED_MAX ? pieces : TIMED_MAX) : 0;
    if (timed_pieces && (rc = ensure_ticks(c, TK * timed_pieces))) return rc;
    uint64_t timed_h2d = 0, timed_d2h = 0;
    // the measured planner inputs: ONE chunk of whole generations from the middle of the call, where both directions of the link are busy
    const size_t whole_pieces = pieces - ((n % chunk) ? 1 : 0);
    const size_t probe = (c->plan_measure && whole_pieces >= 1) ? (whole_pieces >= 3 ? whole_pieces / 2 : 0) : (size_t)-1;
    if (probe != (size_t)-1)
        for (hipEvent_t& e : c->probe_ticks) if (!e) HIP_TRY(c, hipEventCreate(&e));
    auto drain = [&](size_t k) -> int {          // chunk k has left the device: hand a pageable caller its bytes
        const int b = (int)(k % slots);
        if (c->pipe_host_poll) {                 // FOURQ_PIPE_HOST_POLL=1 (experiment, profiles/r06_pipeline.txt): spin on hipEventQuery instead of blocking
            hipError_t q;
            while ((q = hipEventQuery(c->out_done[b])) == hipErrorNotReady) {}
            HIP_TRY(c, q);
        } else {
            HIP_TRY(c, hipEventSynchronize(c->out_done[b]));
        }
        for (int i = 0; i < n_out; i++)
            if (!pin_out[i]) host_copy(out[i].dst + plan[k].off * out[i].stride, c->pipe_pin + (size_t)b * slot + off_out[i], plan[k].m * out[i].stride);
        return FOURQ_OK;
    };
    for (size_t k = 0; k < pieces; k++) {
        const int b = (int)(k % slots);
        const size_t off = plan[k].off, m = plan[k].m;
        if (k >= (size_t)slots) {                                       // slot b is free again once chunk k - slots has been copied out of it
            if (host_wait) { if ((rc = drain(k - slots))) return rc; }
            else HIP_TRY(c, hipStreamWaitEvent(c->copy_in, c->out_done[b], 0));    // ... which the kernels of chunk k inherit through in_done[b]
        }
        char* dev = c->pipe_dev + (size_t)b * slot;
        char* pin = bounce ? c->pipe_pin + (size_t)b * slot : nullptr;
        char *din[PIPE_MAX_ARRAYS], *dout[PIPE_MAX_ARRAYS];
        for (int i = 0; i < n_in; i++) {
            din[i] = dev + off_in[i];
            if (!pin_in[i]) host_copy(pin + off_in[i], in[i].src + off * in[i].stride, m * in[i].stride);
        }
        const bool timed = k < timed_pieces, probed = k == probe;
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[0], c->copy_in));
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k], c->copy_in));
        for (int i = 0; i < n_in; i++) {
            const char* src = pin_in[i] ? in[i].src + off * in[i].stride : pin + off_in[i];
            HIP_TRY(c, hipMemcpyAsync(din[i], src, m * in[i].stride, hipMemcpyHostToDevice, c->copy_in));
            st.h2d_bytes += m * in[i].stride;
            if (timed) timed_h2d += m * in[i].stride;
        }
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k + 1], c->copy_in));
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[1], c->copy_in));
        HIP_TRY(c, hipEventRecord(c->in_done[b], c->copy_in));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->in_done[b], 0));
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k + 4], c->stream));      // the kernel stream is past its wait: the chunk's bytes are there
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[2], c->stream));
        for (int i = 0; i < n_out; i++) dout[i] = dev + off_out[i];
        if ((rc = launch(din, dout, m))) return rc;
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[3], c->stream));
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k + 5], c->stream));
        HIP_TRY(c, hipEventRecord(c->kernels_done[b], c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->copy_out, c->kernels_done[b], 0));
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[4], c->copy_out));
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k + 2], c->copy_out));
        for (int i = 0; i < n_out; i++) {
            char* dst = pin_out[i] ? out[i].dst + off * out[i].stride : pin + off_out[i];
            HIP_TRY(c, hipMemcpyAsync(dst, dout[i], m * out[i].stride, hipMemcpyDeviceToHost, c->copy_out));
            st.d2h_bytes += m * out[i].stride;
            if (timed) timed_d2h += m * out[i].stride;
        }
        if (timed) HIP_TRY(c, hipEventRecord(c->ticks[TK * k + 3], c->copy_out));
        if (probed) HIP_TRY(c, hipEventRecord(c->probe_ticks[5], c->copy_out));
        HIP_TRY(c, hipEventRecord(c->out_done[b], c->copy_out));
    }
    if (host_wait) {
        for (size_t k = pieces > (size_t)slots ? pieces - slots : 0; k < pieces; k++)
            if ((rc = drain(k))) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->copy_out));      // in order behind every chunk's copy out, which is behind its kernels and its copy in
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (probe != (size_t)-1) {                          // what the next call of this route will be planned with
        float t_in = 0, t_k = 0, t_out = 0;
        HIP_TRY(c, hipEventElapsedTime(&t_in, c->probe_ticks[0], c->probe_ticks[1]));
        HIP_TRY(c, hipEventElapsedTime(&t_k, c->probe_ticks[2], c->probe_ticks[3]));
        HIP_TRY(c, hipEventElapsedTime(&t_out, c->probe_ticks[4], c->probe_ticks[5]));
        const double m = (double)plan[probe].m;
        if (t_k > 0) c->plan_kt[route.id][mode] = (double)t_k * 1e6 / m;
        // the link's rates only from a chunk that had neighbours on both sides (a call of one or two chunks copies with the link to itself),
        // and only from a chunk copied directly (a bounce slot's copy is paced by the host threads that fill it)
        if (whole_pieces >= 3 && !bounce) {
            if (t_in > 0 && bytes_in) c->plan_link_in = m * (double)bytes_in / ((double)t_in * 1e6);
            if (t_out > 0 && bytes_out) c->plan_link_out = m * (double)bytes_out / ((double)t_out * 1e6);
        }
        st.measured_kernel_ns_per_elem = c->plan_kt[route.id][mode];
    }
    try { c->chunk_stamps.clear(); } catch (...) {}
    if (timed_pieces) {
        try {
            c->chunk_stamps.resize(TK * timed_pieces);
            for (size_t i = 0; i < TK * timed_pieces; i++) HIP_TRY(c, hipEventElapsedTime(&c->chunk_stamps[i], c->ticks[0], c->ticks[i]));
        } catch (...) { c->chunk_stamps.clear(); }
        for (size_t k = 0; k < timed_pieces; k++) {
            float ms = 0;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[TK * k], c->ticks[TK * k + 1]));
            st.h2d_ms += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[TK * k + 2], c->ticks[TK * k + 3]));
            st.d2h_ms += ms;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ticks[TK * k + 4], c->ticks[TK * k + 5]));
            st.kernels_ms += ms;
        }
        float span = 0;
        HIP_TRY(c, hipEventElapsedTime(&span, c->ticks[4], c->ticks[TK * (timed_pieces - 1) + 5]));
        st.kernels_span_ms = span;
        if (timed_h2d && timed_h2d < st.h2d_bytes) st.h2d_ms *= (double)st.h2d_bytes / (double)timed_h2d;
        if (timed_d2h && timed_d2h < st.d2h_bytes) st.d2h_ms *= (double)st.d2h_bytes / (double)timed_d2h;
    }
    c->host_stats = st;
    return FOURQ_OK;
}

// chunk of a host-pointer batch: whole generations of the kernels that will run it
size_t pipe_chunk(const fourq_ctx* c, bool fused_route) { return fused_route ? c->lanes : c->lanes_w4; }
// Kernel time per element of each route in nanoseconds, device-resident at 2^20 elements (profiles/r04_perf_probe.txt, tools/perf_probe.py).
// They only size the chunks of the host-array calls (pipeline_plan.h): a kernel slower than its figure (constant-time mode, a slower box)
// gets chunks smaller than it could have had, one faster by more than the plan's 15 % margin a short stall on the first chunks.
constexpr double KT_ENDO_VAR = 4.63, KT_WIN_VAR = 8.18, KT_DH_VAR = 4.90, KT_ENDO_FIXED = 3.58, KT_WIN_FIXED = 7.65, KT_DH_FIXED = 3.65, KT_COMB = 1.09,
                 KT_LIFT_LOWER = 0.10, KT_CODEC = 0.30, KT_DECODE = 0.25, KT_ENCODE = 0.05;
int mul_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out, size_t n) {
    if (!c || !scalars || !out || (!points && !table) || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[2] = { { (const char*)scalars, nullptr, 32 }, { (const char*)points, nullptr, 160 } };
    PipeArray o[1] = { { nullptr, (char*)out, 160 } };
    // variable-base MUL_endo: chunks of one fused generation overlap the copies better than rounds of the two-kernel route
    // would (16 chunks instead of 4 at 2^20 elements: 6.7 ms against ~7.8)
    const bool fused = points && (algo == ENDO || !takes_split_route(c, algo, false, n));
    // fixed base: the LDS ladders hold two waves per SIMD, so HALF of lanes_w4 is one generation of theirs -- the unit that sizes the first
    // chunk's copy in and the last chunk's copy out (cfg3's call: 160 B out per element, 21 MB instead of 42 behind the last kernel)
    const size_t unit = points ? pipe_chunk(c, fused) : c->lanes_w4 / 2;
    const double kt = points ? (algo == ENDO ? KT_ENDO_VAR : KT_WIN_VAR) : (algo == ENDO ? KT_ENDO_FIXED : KT_WIN_FIXED);
    return run_pipeline(c, in, points ? 2 : 1, o, 1, n, unit, PipeRoute{ (points ? PR_MUL_VAR : PR_MUL_FIX) + (algo == ENDO ? 0 : 1), kt }, [&](char* const* di, char* const* dout, size_t m) {
        return mul_dev(c, algo, (const uint64_t*)di[0], points ? (const uint64_t*)di[1] : nullptr, table, (uint64_t*)dout[0], nullptr, m);
    }, pipe_chunk(c, fused));
}
int dh_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
            uint8_t* status, size_t n) {
    if (!c || !scalars || !points || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[2] = { { (const char*)scalars, nullptr, 32 }, { (const char*)points, nullptr, 64 } };
    PipeArray o[2] = { { nullptr, (char*)out, 64 }, { nullptr, (char*)status, 1 } };
    const bool fused = !table && !takes_split_route(c, algo, true, n);
    const size_t unit = table ? c->lanes_w4 / 2 : pipe_chunk(c, fused);
    const double kt = table ? KT_DH_FIXED : (algo == ENDO ? KT_DH_VAR : KT_WIN_VAR + 0.3);
    return run_pipeline(c, in, 2, o, 2, n, unit, PipeRoute{ (table ? PR_DH_FIX : PR_DH_VAR) + (algo == ENDO ? 0 : 1), kt }, [&](char* const* di, char* const* dout, size_t m) {
        return dh_dev(c, algo, (const uint64_t*)di[0], (const uint64_t*)di[1], table, (uint64_t*)dout[0], (uint8_t*)dout[1], m);
    }, pipe_chunk(c, fused), [&](size_t big) { return ensure_proj(c, big); });
}

int table_host(fourq_ctx* c, int algo, const uint64_t* p_r1, uint64_t* table) {
    if (!c || !p_r1 || !table) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    int rc = ensure_stage(c, 160);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, p_r1, 160, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(table_build_kernel, dim3(1), dim3(64), 0, c->stream, algo, (const u64*)c->stage, c->scratch, c->table_packed);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(table, c->table_packed, FOURQ_TABLE_WORDS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}

}  // namespace

extern "C" {

#define FQ_API __attribute__((visibility("default")))

FQ_API int fourq_version(void) { return 600; }    // 0.6.0; fourq_amd/_lib.py checks it at load time
#ifndef FQ_BUILD_ID
#define FQ_BUILD_ID "unknown"
#endif
FQ_API const char* fourq_build_id(void) { return FQ_BUILD_ID; }   // fourq_amd/build.py: hash of the sources and flags of this build

FQ_API const char* fourq_strerror(int code) {
    switch (code) {
    case FOURQ_OK: return "ok";
    case FOURQ_ERR_INVALID: return "invalid argument";
    case FOURQ_ERR_NODEVICE: return "no usable gfx950 HIP device";
    case FOURQ_ERR_NOMEM: return "out of memory";
    case FOURQ_ERR_HIP: return "HIP runtime error";
    default: return "unknown error";
    }
}
FQ_API const char* fourq_last_error(const fourq_ctx* ctx) { return ctx ? ctx->err : "no context"; }

FQ_API int fourq_device_count(int* count) {
    if (!count) return FOURQ_ERR_INVALID;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return FOURQ_OK; }
    for (int d = 0; d < n; d++) {                   // devices are numbered as HIP numbers them; a non-gfx950 device ends the usable prefix
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;
        *count = d + 1;
    }
    return FOURQ_OK;
}

// Environment.  ONE product option is read from the environment: FOURQ_CT_SELECT (include/fourq_amd.h).  Everything else that
// changes which kernels a call takes is a TEST HOOK -- it exists so that the tests, tools/fuzz.py and the A/B probes can reach every
// route at small sizes -- and is read only when FOURQ_DEBUG_ROUTES=1 is set as well (tools/README.md lists the hooks); a production
// process cannot be re-routed by a stray variable.
static const char* route_env(const char* name) {
    const char* gate = getenv("FOURQ_DEBUG_ROUTES");
    return (gate && atoi(gate) != 0) ? getenv(name) : nullptr;
}
FQ_API int fourq_ctx_create(int device, fourq_ctx** out) {
    if (!out) return FOURQ_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return FOURQ_ERR_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FOURQ_ERR_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FOURQ_ERR_NODEVICE;   // the code object is gfx950 only
    fourq_ctx* c = new (std::nothrow) fourq_ctx();
    if (!c) return FOURQ_ERR_NOMEM;
    c->device = device;
    c->cus = prop.multiProcessorCount;
    DeviceGuard g(device);
    int rc = FOURQ_OK;
    do {
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = FOURQ_ERR_HIP; break; }
        c->stream = c->own_stream;
        if (hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking) != hipSuccess) { rc = FOURQ_ERR_HIP; break; }
        if (hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking) != hipSuccess) { rc = FOURQ_ERR_HIP; break; }
        for (int i = 0; i < PIPE_SLOTS_MAX && rc == FOURQ_OK; i++) {
            if (hipEventCreateWithFlags(&c->in_done[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->kernels_done[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->out_done[i], hipEventDisableTiming) != hipSuccess) rc = FOURQ_ERR_HIP;
        }
        if (rc) break;
        if (hipEventCreateWithFlags(&c->shadow_read, hipEventDisableTiming) != hipSuccess) { rc = FOURQ_ERR_HIP; break; }
        if (chain_setup_device() != 0) { rc = FOURQ_ERR_HIP; break; }
        // resident blocks per CU of the fused variable-base kernels (they own the per-lane scratch slots)
        int occ = 8, o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ladder_kernel<ENDO, FUSED, false>, BLOCK, 0) == hipSuccess && o > 0 && o < occ) occ = o;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ladder_kernel<WINDOWED, FUSED, true>, BLOCK, 0) == hipSuccess && o > 0 && o < occ) occ = o;
        if (const char* env = route_env("FOURQ_BLOCKS_PER_CU")) { int v = atoi(env); if (v > 0 && v <= 8) occ = v; }
        c->lanes = (size_t)c->cus * occ * BLOCK;
        c->lanes_w4 = (size_t)c->cus * 4 * BLOCK;
        // The two-kernel route (prep_kernel + ladder_kernel<PREBUILT>) for MUL_windowed / DH_* past one fused generation was worth
        // 3-8 % through round 3; since the fused kernels' ladders run on the hand-scheduled bodies (ladder_asm.hip.h) the fused route is
        // 3-6 % ahead at every size and config 4's step 4.7 % faster on it (profiles/r04_routes.txt).  The route stays reachable for
        // experiments (FOURQ_SPLIT_MIN = smallest batch that takes it); mixed batches keep their own use of both kernels.
        c->split_min = ~(size_t)0;
        if (const char* env = route_env("FOURQ_SPLIT_MIN")) { long v = atol(env); if (v > 0) c->split_min = (size_t)v; }
        if (const char* env = route_env("FOURQ_SPLIT_ALL")) c->split_all = atoi(env) != 0;
        c->split_endo_min = 0;
        if (const char* env = route_env("FOURQ_SPLIT_ENDO_MIN")) { long v = atol(env); if (v >= 0) c->split_endo_min = (size_t)v; }
        if (const char* env = route_env("FOURQ_HOST_BOUNCE")) c->host_bounce = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_HOST_ZERO_COPY")) c->host_zero_copy = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_PIPE_SLOTS")) { int v = atoi(env); if (v >= 2 && v <= PIPE_SLOTS_MAX) c->pipe_slots = v; }
        if (const char* env = route_env("FOURQ_PIPE_GENS")) { int v = atoi(env); if (v >= 0 && v <= 64) c->pipe_gens = v; }
        if (const char* env = route_env("FOURQ_PIPE_HOST_WAIT")) c->pipe_host_wait = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_PIPE_HOST_POLL")) c->pipe_host_poll = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_PIPE_MEASURE")) c->plan_measure = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_FUSED_IO")) c->fused_io = atoi(env) != 0;
        if (const char* env = getenv("FOURQ_CT_SELECT")) c->ct = atoi(env) != 0;
        if (const char* env = route_env("FOURQ_MIXED_QUEUE")) { int v = atoi(env); if (v == 0 || v == 1) c->mixed_queue = v; }
        c->pair_max = c->lanes / 2;                        // two lanes per element: half a generation fills the chip
        if (const char* env = route_env("FOURQ_PAIR_MAX")) { long v = atol(env); if (v >= 0 && (size_t)v <= c->lanes / 2) c->pair_max = (size_t)v; }
        c->quad_max = c->pair_max < c->lanes / 4 ? c->pair_max : c->lanes / 4;
        if (const char* env = route_env("FOURQ_QUAD_MAX")) { long v = atol(env); if (v >= 0 && (size_t)v <= c->quad_max) c->quad_max = (size_t)v; }
        if (const char* env = route_env("FOURQ_NORM_K")) { int v = atoi(env); if (v == 0 || v == 2 || v == 4 || v == 8) c->norm_k = v; }
        c->split_chunk = c->lanes_w4;
        if (const char* env = route_env("FOURQ_SPLIT_CHUNK")) { long v = atol(env); if (v >= BLOCK && (size_t)v <= c->lanes_w4) c->split_chunk = (size_t)v; }
        size_t scratch_u32 = c->lanes * NDSlots::SLOT;                                   // fused kernels: N, D per resident lane
        if (c->lanes_w4 * PrebuiltSlots::SLOT > scratch_u32) scratch_u32 = c->lanes_w4 * PrebuiltSlots::SLOT;   // two-kernel route: per element of a round
        if ((size_t)COMB_POINTS * SLOT_U32 > scratch_u32) scratch_u32 = (size_t)COMB_POINTS * SLOT_U32;         // comb_table_kernel: whole entries
        const size_t ct_tail_u32 = c->lanes * NDSlots::SLOT + (c->lanes / 8) * LimbSlots::SLOT;                 // constant-time mixed rounds: fused slots + the overflow ids' whole entries
        if (ct_tail_u32 > scratch_u32) scratch_u32 = ct_tail_u32;
        if (hipMalloc(&c->scratch, scratch_u32 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->table_limbs, 8 * R2_LIMBS * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->table_slots, 8 * R2_LIMBS * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->table_packed, FOURQ_TABLE_WORDS * 8) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->part_counter, 8 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }   // n_var, n_fix, queue head, -, fused, overflow
        if (hipMalloc(&c->part_fix, c->lanes_w4 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->part_list, c->lanes_w4 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->part_slot, c->lanes_w4 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->comb_limbs, COMB_POINTS * COMB_ENTRY_U32 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->comb_packed, FOURQ_COMB_WORDS * 8) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
    } while (0);
    if (rc) { fourq_ctx_destroy(c); return rc; }
    *out = c;
    return FOURQ_OK;
}

FQ_API int fourq_ctx_destroy(fourq_ctx* c) {
    if (!c) return FOURQ_ERR_INVALID;
    { CtxGuard last(c); }                      // a call still running on another thread finishes first; the caller must not start new ones
    DeviceGuard g(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->proj) (void)hipFree(c->proj);
    if (c->table_limbs) (void)hipFree(c->table_limbs);
    if (c->table_slots) (void)hipFree(c->table_slots);
    if (c->table_packed) (void)hipFree(c->table_packed);
    if (c->part_counter) (void)hipFree(c->part_counter);
    if (c->part_list) (void)hipFree(c->part_list);
    if (c->part_slot) (void)hipFree(c->part_slot);
    if (c->part_fix) (void)hipFree(c->part_fix);
    if (c->comb_limbs) (void)hipFree(c->comb_limbs);
    if (c->comb_packed) (void)hipFree(c->comb_packed);
    if (c->stage) (void)hipFree(c->stage);
    if (c->work) (void)hipFree(c->work);
    if (c->pipe_dev) (void)hipFree(c->pipe_dev);
    if (c->pipe_pin) (void)hipHostFree(c->pipe_pin);
    if (c->zero_copy) (void)hipHostFree(c->zero_copy);
    if (c->diag_stamps) (void)hipFree(c->diag_stamps);
    if (c->diag_bracket) (void)hipFree(c->diag_bracket);
    if (c->diag_mark) (void)hipEventDestroy(c->diag_mark);
    if (c->shadow_read) (void)hipEventDestroy(c->shadow_read);
    for (hipEvent_t e : c->ticks) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->probe_ticks) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < PIPE_SLOTS_MAX; i++) {
        if (c->in_done[i]) (void)hipEventDestroy(c->in_done[i]);
        if (c->kernels_done[i]) (void)hipEventDestroy(c->kernels_done[i]);
        if (c->out_done[i]) (void)hipEventDestroy(c->out_done[i]);
    }
    if (c->copy_in) (void)hipStreamDestroy(c->copy_in);
    if (c->copy_out) (void)hipStreamDestroy(c->copy_out);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return FOURQ_OK;
}

FQ_API int fourq_ctx_set_stream(fourq_ctx* c, void* hip_stream) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (next != c->stream) {
        // the staged fixed-base / comb tables were unpacked by launches on the old stream: nothing orders the new
        // stream behind them, so they are staged again (1 KiB / 7.5 KiB) by the next call that needs them
        c->table_staged = false;
        c->comb_staged = false;
    }
    c->stream = next;
    return FOURQ_OK;
}
FQ_API int fourq_ctx_set_ct_select(fourq_ctx* c, int on) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    c->ct = on != 0;
    return FOURQ_OK;
}
FQ_API int fourq_ctx_get_ct_select(const fourq_ctx* c, int* on) {
    if (!c || !on) return FOURQ_ERR_INVALID;
    *on = c->ct ? 1 : 0;
    return FOURQ_OK;
}
FQ_API int fourq_ctx_sync(fourq_ctx* c) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
// Pre-sizes every buffer a _dev call for up to n elements can need (the deferred-normalisation planes and the intermediates
// of the protocol-level calls), so that those calls only enqueue: without it the first batch larger than any seen before
// synchronises the stream and reallocates inside the call (not capturable into a graph).
FQ_API int fourq_ctx_reserve(fourq_ctx* c, size_t n) {
    if (!c || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    int rc = ensure_proj(c, n);
    if (rc) return rc;
    size_t need = dh_bytes_work_bytes(n);
    if (exchange_work_bytes(n) > need) need = exchange_work_bytes(n);
    if (mul_affine_work_bytes(n) > need) need = mul_affine_work_bytes(n);
    return ensure_work(c, need);
}
FQ_API int fourq_ctx_lanes(const fourq_ctx* c, size_t* lanes) {
    if (!c || !lanes) return FOURQ_ERR_INVALID;
    *lanes = c->lanes;
    return FOURQ_OK;
}

FQ_API int fourq_dev_alloc(fourq_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipMalloc(out, bytes ? bytes : 16));
    return FOURQ_OK;
}
FQ_API int fourq_dev_free(fourq_ctx* c, void* ptr) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipFree(ptr));
    return FOURQ_OK;
}
FQ_API int fourq_dev_upload(fourq_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
FQ_API int fourq_dev_download(fourq_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}

FQ_API int fourq_table_windowed(fourq_ctx* c, const uint64_t* p, uint64_t* t) { return table_host(c, WINDOWED, p, t); }
FQ_API int fourq_table_endo(fourq_ctx* c, const uint64_t* p, uint64_t* t) { return table_host(c, ENDO, p, t); }

FQ_API int fourq_mul_endo_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) {
    return p ? mul_host(c, ENDO, s, p, nullptr, o, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_windowed_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) {
    return p ? mul_host(c, WINDOWED, s, p, nullptr, o, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_endo_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) {
    return p ? mul_dev(c, ENDO, s, p, nullptr, o, nullptr, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_windowed_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) {
    return p ? mul_dev(c, WINDOWED, s, p, nullptr, o, nullptr, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_endo_fixed_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* t, uint64_t* o, size_t n) {
    return t ? mul_host(c, ENDO, s, nullptr, t, o, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_windowed_fixed_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* t, uint64_t* o, size_t n) {
    return t ? mul_host(c, WINDOWED, s, nullptr, t, o, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_endo_fixed_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* t, uint64_t* o, size_t n) {
    return t ? mul_dev(c, ENDO, s, nullptr, t, o, nullptr, n) : FOURQ_ERR_INVALID;
}
FQ_API int fourq_mul_windowed_fixed_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* t, uint64_t* o, size_t n) {
    return t ? mul_dev(c, WINDOWED, s, nullptr, t, o, nullptr, n) : FOURQ_ERR_INVALID;
}

FQ_API int fourq_mul_endo_mixed_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint8_t* flags,
                                          const uint64_t* table, uint64_t* o, size_t n) {
    if (!c || !s || !p || !flags || !table || !o || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(s) || !aligned16(p) || !aligned16(o)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    int rc = stage_table(c, table);
    if (rc) return rc;
    // At most half a generation (and no route forced): one launch of the two- / four-lane kernels, each element on its own kind of
    // table -- 0.15-0.23 ms where the work-queue kernel takes a one-lane ladder's 0.35 (profiles/r03_quadlane.txt).
    if (c->mixed_queue < 0 && c->pair_max && n <= c->pair_max) {
        LadderArgs a = {};
        a.scalars = s; a.points = p; a.out = o; a.n = (u32)n; a.flags = flags;
        return launch_pair_mixed(c, a);
    }
    // Rounds of up to split_chunk elements.  Per round: compact the ids of both kinds (counts stay on the device), then -- the default --
    // ONE persistent kernel whose waves pull 64-element work items off a device-side queue (mixed_queue_kernel), or, behind the test hook
    // FOURQ_MIXED_QUEUE=0, round 2's route: the variable-base ids' tables built into scratch slots (prep_kernel over the compacted list)
    // and ONE ladder launch over all elements of the round in their natural order, each lane reading its table through a pointer -- its
    // own slot or the shared fixed-base table -- so that fixed and variable elements share wavefronts without divergence.
    const size_t per_block = (size_t)BLOCK * PART_PER_LANE;
    for (size_t off = 0; off < n; off += c->split_chunk) {
        const u32 m = (u32)(n - off < c->split_chunk ? n - off : c->split_chunk);
        const bool queue = c->mixed_queue >= 0 ? c->mixed_queue != 0 : mixed_queue_default(c, m);
        HIP_TRY(c, hipMemsetAsync(c->part_counter, 0, 8 * sizeof(u32), c->stream));
        hipLaunchKernelGGL(partition_kernel, dim3((unsigned)((m + per_block - 1) / per_block)), dim3(BLOCK), 0, c->stream,
                           flags + off, m, (u32)off, c->part_list, c->part_slot, c->part_counter, (c->ct || queue) ? c->part_fix : nullptr);
        HIP_TRY(c, hipGetLastError());
        LadderArgs a = {};
        a.scalars = s; a.points = p; a.out = o; a.n = m;
        a.scratch = c->scratch; a.table = c->table_limbs; a.table_slots = c->table_slots;
        if (queue) {
            // BASELINE config 5's mechanism: one persistent kernel, one block per CU, every wave pulling 64-element work items
            // (variable-base first) from a device-side queue until it is empty (kernels.hip.h, mixed_queue_kernel)
            const size_t items = ((size_t)m + 63) / 64, blocks = (items + 3) / 4;
            const unsigned grid = (unsigned)(blocks < (size_t)c->cus ? blocks : (size_t)c->cus);
            if (c->ct) {
                HIPRC_TRY(c, ct_launch_mixed_queue(grid, c->stream, a, c->part_list, c->part_fix, c->part_counter, c->part_counter + 2));
            } else {
                hipLaunchKernelGGL(mixed_queue_kernel<false>, dim3(grid), dim3(BLOCK), 0, c->stream, a, c->part_list, c->part_fix, c->part_counter, c->part_counter + 2);
                HIP_TRY(c, hipGetLastError());
            }
            continue;
        }
        if (c->ct) {
            // constant-time selection: which elements are fixed-base is public, the digits are not.  The variable-base
            // ids go through the fused kernel (table in registers), the fixed-base ids through the LDS kernel; both
            // read their element counts on the device.
            // A remainder of at most lanes / 8 ids past whole generations of the fused kernel is cut off on the device and
            // runs with the fixed-base elements (kernels.hip.h, mixed_ct_tail_kernel): config 5's 65 550 variable-base ids
            // are one fused generation + 14 riders, not two generations.
            const u32 limit = (u32)(c->lanes / 8);
            u32* over_scratch = c->scratch + c->lanes * NDSlots::SLOT;           // behind the fused kernels' per-lane slots
            HIPRC_TRY(c, ct_launch_split_counts(c->stream, c->part_counter, (u32)c->lanes, limit));
            LadderArgs av = a;
            av.index = c->part_list; av.n_dev = c->part_counter + 4;
            if ((rc = launch_ladder<ENDO, FUSED, false>(c, av))) return rc;
            const size_t tail_blocks = ((size_t)m + BLOCK - 1) / BLOCK, tail_max = c->lanes_w4 / BLOCK;
            HIPRC_TRY(c, ct_launch_mixed_tail((limit + BLOCK - 1) / BLOCK, (unsigned)(tail_blocks < tail_max ? tail_blocks : tail_max), c->stream, a,
                                              c->part_fix, c->part_list, c->part_counter, over_scratch, (u32)c->lanes, limit));
            continue;
        }
        LadderArgs ap = a;
        ap.index = c->part_list; ap.n_dev = c->part_counter;
        HIPRC_TRY(c, chain_launch_prep(ENDO, false, (m + BLOCK - 1) / BLOCK, c->stream, ap));
        a.base = (u32)off; a.slot_of = c->part_slot;
        rc = launch_ladder<ENDO, PREBUILT, false>(c, a);
        if (rc) return rc;
    }
    return FOURQ_OK;
}
FQ_API int fourq_mul_endo_mixed_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint8_t* flags,
                                      const uint64_t* table, uint64_t* o, size_t n) {
    if (!c || !s || !p || !flags || !table || !o || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[3] = { { (const char*)s, nullptr, 32 }, { (const char*)p, nullptr, 160 }, { (const char*)flags, nullptr, 1 } };
    PipeArray out[1] = { { nullptr, (char*)o, 160 } };
    // one pass of the work-queue kernel's 4 x CUs waves over 64-element items is `lanes` elements: the unit that lets the copies of a
    // config-5-sized call (2 x lanes) overlap its kernels; raw R1 in and out keeps the chunks at that size (pipeline_plan.h)
    const size_t unit = c->split_chunk < c->lanes ? c->split_chunk : c->lanes;
    return run_pipeline(c, in, 3, out, 1, n, unit, PipeRoute{ PR_MIXED, KT_ENDO_VAR }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_mul_endo_mixed_batch_dev(c, (const uint64_t*)di[0], (const uint64_t*)di[1], (const uint8_t*)di[2], table, (uint64_t*)dout[0], m);
    }, c->split_chunk);
}

FQ_API int fourq_dh_endo_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint64_t* t, uint64_t* o, uint8_t* st, size_t n) {
    return dh_host(c, ENDO, s, p, t, o, st, n);
}
FQ_API int fourq_dh_windowed_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint64_t* t, uint64_t* o, uint8_t* st, size_t n) {
    return dh_host(c, WINDOWED, s, p, t, o, st, n);
}
FQ_API int fourq_dh_endo_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint64_t* t, uint64_t* o, uint8_t* st, size_t n) {
    return dh_dev(c, ENDO, s, p, t, o, st, n);
}
FQ_API int fourq_dh_windowed_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, const uint64_t* t, uint64_t* o, uint8_t* st, size_t n) {
    return dh_dev(c, WINDOWED, s, p, t, o, st, n);
}

FQ_API int fourq_comb_table(fourq_ctx* c, const uint64_t* p_r1, uint64_t* comb) {
    if (!c || !p_r1 || !comb) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    int rc = ensure_stage(c, 160);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, p_r1, 160, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(comb_table_kernel, dim3((COMB_POINTS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, (const u64*)c->stage, c->scratch, c->comb_packed);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(comb, c->comb_packed, FOURQ_COMB_WORDS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
// The staged comb stays on the device between calls.  fourq_comb_stage does the 106 KB compare-and-upload once; batch calls
// that pass comb == NULL use what is staged and do no host-side work on the table at all (a non-NULL comb is compared with the
// staged copy on every call, which costs microseconds of host time per call on the key-generation path).
// comb == NULL: "the table fourq_comb_stage was given" -- staged again from the shadow if the stream has changed since
static int stage_comb(fourq_ctx* c, const uint64_t* comb) {
    if (!comb) {
        if (!c->comb_known) return FOURQ_ERR_INVALID;          // NULL = "the staged table": there must be one
        if (c->comb_staged) return FOURQ_OK;
        if (int rc = staging_allowed(c)) return rc;
    } else {
        if (c->comb_staged && memcmp(c->comb_shadow, comb, sizeof c->comb_shadow) == 0) return FOURQ_OK;      // as stage_table
        if (int rc = staging_allowed(c)) return rc;
        c->comb_staged = false;
        if (int rc = shadow_free(c)) return rc;
        memcpy(c->comb_shadow, comb, sizeof c->comb_shadow);
        c->comb_known = true;
    }
    HIP_TRY(c, hipMemcpyAsync(c->comb_packed, c->comb_shadow, FOURQ_COMB_WORDS * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->shadow_read, c->stream));
    hipLaunchKernelGGL(comb_unpack_kernel, dim3((COMB_POINTS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, c->stream, c->comb_packed, c->comb_limbs);
    HIP_TRY(c, hipGetLastError());
    c->comb_staged = true;
    return FOURQ_OK;
}
FQ_API int fourq_comb_stage(fourq_ctx* c, const uint64_t* comb) {
    if (!c || !comb) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    return stage_comb(c, comb);
}
FQ_API int fourq_comb_mul_batch_dev(fourq_ctx* c, const uint64_t* scalars, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !scalars || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    if (int rc = stage_comb(c, comb)) return rc;
    const int group = normalize_group(c, n);
    // At most a quarter generation, selection by address: four lanes per element, entries gathered where the table lies (kernels.hip.h,
    // comb_quad_kernel): a batch of one 0.107 -> 0.05x ms (profiles/r03_quadlane.txt)
    if (!group && n <= c->pair_max) {         // two lanes per element between a quarter and half a generation: 0.100 -> 0.06x ms
        const bool quad = n <= c->quad_max;
        const size_t per_block = BLOCK / (quad ? 4 : 2);
        const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
        if (c->ct) {
            HIPRC_TRY(c, ct_launch_comb_quad(quad, grid, c->stream, scalars, c->comb_limbs, out, status, (u32)n));
            return FOURQ_OK;
        }
        if (quad) hipLaunchKernelGGL((comb_quad_kernel<false, 4>), dim3(grid), dim3(BLOCK), 0, c->stream, scalars, c->comb_limbs, out, status, (u32)n);
        else hipLaunchKernelGGL((comb_quad_kernel<false, 2>), dim3(grid), dim3(BLOCK), 0, c->stream, scalars, c->comb_limbs, out, status, (u32)n);
        HIP_TRY(c, hipGetLastError());
        return FOURQ_OK;
    }
    int rc = group ? ensure_proj(c, n) : FOURQ_OK;
    if (rc) return rc;
    size_t blocks = (n + BLOCK - 1) / BLOCK, blocks_max = c->lanes_w4 / BLOCK;
    // constant-time mode: 256-lane blocks, four per CU, over the small shape; otherwise one block per CU with the fast shape's table
    const unsigned grid_or_cus = c->ct ? (unsigned)(blocks < blocks_max ? blocks : blocks_max) : (unsigned)c->cus;
    HIPRC_TRY(c, (c->ct ? ct_launch_comb : chain_launch_comb)(grid_or_cus, c->stream, scalars, c->comb_limbs, out, status,
                                                             group ? c->proj : nullptr, (u32)c->proj_capacity, (u32)n));
    if (group) HIPRC_TRY(c, chain_launch_normalize(group, c->stream, c->proj, (u32)c->proj_capacity, out, status, (u32)n));
    return FOURQ_OK;
}
FQ_API int fourq_comb_mul_batch(fourq_ctx* c, const uint64_t* scalars, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !scalars || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    if (int rc = stage_comb(c, comb)) return rc;                                        // compared once, not once per chunk
    comb = nullptr;
    PipeArray in[1] = { { (const char*)scalars, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)out, 64 }, { nullptr, (char*)status, 1 } };
    return run_pipeline(c, in, 1, o, 2, n, c->lanes_w4, PipeRoute{ PR_COMB, KT_COMB }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_comb_mul_batch_dev(c, (const uint64_t*)di[0], comb, (uint64_t*)dout[0], (uint8_t*)dout[1], m);
    });
}

FQ_API int fourq_encode_batch_dev(fourq_ctx* c, const uint64_t* affine, uint8_t* out32, size_t n) {
    if (!c || !affine || !out32 || n > FOURQ_MAX_BATCH || !aligned16(affine) || !aligned16(out32)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, affine, (u64*)out32, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_decode_batch_dev(fourq_ctx* c, const uint8_t* in32, uint64_t* affine, uint8_t* status, size_t n) {
    if (!c || !in32 || !affine || !status || n > FOURQ_MAX_BATCH || !aligned16(in32) || !aligned16(affine)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)in32, affine, status, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_encode_batch(fourq_ctx* c, const uint64_t* affine, uint8_t* out32, size_t n) {
    if (!c || !affine || !out32 || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[1] = { { (const char*)affine, nullptr, 64 } };
    PipeArray o[1] = { { nullptr, (char*)out32, 32 } };
    return run_pipeline(c, in, 1, o, 1, n, 4 * c->lanes_w4, PipeRoute{ PR_ENCODE, KT_ENCODE }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_encode_batch_dev(c, (const uint64_t*)di[0], (uint8_t*)dout[0], m);
    });
}
FQ_API int fourq_decode_batch(fourq_ctx* c, const uint8_t* in32, uint64_t* affine, uint8_t* status, size_t n) {
    if (!c || !in32 || !affine || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[1] = { { (const char*)in32, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)affine, 64 }, { nullptr, (char*)status, 1 } };
    return run_pipeline(c, in, 1, o, 2, n, c->lanes_w4, PipeRoute{ PR_DECODE, KT_DECODE }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_decode_batch_dev(c, (const uint8_t*)di[0], (uint64_t*)dout[0], (uint8_t*)dout[1], m);
    });
}

// ---- protocol-level calls: every intermediate stays on the device -------------------------------------------
// decode -> DH_<algo> -> encode (draft-ladd-cfrg-4q.md:707-723; curve4q.py:49-96, :446-462, :41-46)
static int dh_bytes_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint8_t* keys32, const uint64_t* table, uint8_t* out32,
                        uint8_t* status, size_t n) {
    if (!c || !scalars || !keys32 || !out32 || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(keys32) || !aligned16(out32)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    const size_t nb = align256(n);
    int rc = ensure_work(c, dh_bytes_work_bytes(n));
    if (rc) return rc;
    uint64_t* pts = (uint64_t*)c->work;                       // decoded public keys
    uint64_t* shared = (uint64_t*)(c->work + n * 64);         // affine shared points
    uint8_t* st_decode = (uint8_t*)(c->work + 2 * n * 64);
    uint8_t* st_dh = st_decode + nb;
    if ((rc = fourq_decode_batch_dev(c, keys32, pts, st_decode, n))) return rc;
    if ((rc = dh_dev(c, algo, scalars, pts, table, shared, st_dh, n))) return rc;    // an undecodable key is (0, 0): rejected again here
    hipLaunchKernelGGL(encode_status_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, shared, st_decode, st_dh,
                       (u64*)out32, status, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
static int dh_bytes_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint8_t* keys32, const uint64_t* table, uint8_t* out32,
                         uint8_t* status, size_t n) {
    if (!c || !scalars || !keys32 || !out32 || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[2] = { { (const char*)scalars, nullptr, 32 }, { (const char*)keys32, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)out32, 32 }, { nullptr, (char*)status, 1 } };
    const bool fused = !table && !takes_split_route(c, algo, true, n);
    const size_t unit = table ? c->lanes_w4 / 2 : pipe_chunk(c, fused);
    const double kt = (table ? KT_DH_FIXED : (algo == ENDO ? KT_DH_VAR : KT_WIN_VAR + 0.3)) + KT_CODEC;
    return run_pipeline(c, in, 2, o, 2, n, unit, PipeRoute{ (table ? PR_DHB_FIX : PR_DHB_VAR) + (algo == ENDO ? 0 : 1), kt }, [&](char* const* di, char* const* dout, size_t m) {
        return dh_bytes_dev(c, algo, (const uint64_t*)di[0], (const uint8_t*)di[1], table, (uint8_t*)dout[0], (uint8_t*)dout[1], m);
    }, pipe_chunk(c, fused), [&](size_t big) { if (int r = ensure_proj(c, big)) return r; return ensure_work(c, dh_bytes_work_bytes(big)); });
}
// ---- MUL_* with affine / encoded I/O: R1toAffine(MUL_<algo>(m, AffineToR1(P))) and encode(.) of it --------------------------
// 160 (96) bytes per operation across the ABI instead of the raw-R1 form's 352: the host-array calls are bound by the link, not by the
// kernels ("PCIe-inclusive").  Parity level L1 (canonical affine); the raw-R1 entry points are untouched.
// R1toAffine (+ encode) behind a MUL_*: one inversion per element while the chunk is at most two generations, one per four beyond
static int launch_lower(fourq_ctx* c, bool enc, const uint64_t* r1, u32 stride, const uint8_t* st_decode, uint64_t* out, uint8_t* status, size_t n) {
    const bool batched = n > 2 * c->lanes;
    const size_t lanes = batched ? (n + 3) / 4 : n;
    const unsigned grid = (unsigned)((lanes + BLOCK - 1) / BLOCK);
    if (enc) {
        if (batched) hipLaunchKernelGGL((lower_kernel<4, true>), dim3(grid), dim3(BLOCK), 0, c->stream, r1, stride, st_decode, out, status, (u32)n);
        else hipLaunchKernelGGL((lower_kernel<1, true>), dim3(grid), dim3(BLOCK), 0, c->stream, r1, stride, st_decode, out, status, (u32)n);
    } else {
        if (batched) hipLaunchKernelGGL((lower_kernel<4, false>), dim3(grid), dim3(BLOCK), 0, c->stream, r1, stride, st_decode, out, status, (u32)n);
        else hipLaunchKernelGGL((lower_kernel<1, false>), dim3(grid), dim3(BLOCK), 0, c->stream, r1, stride, st_decode, out, status, (u32)n);
    }
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
// Does the whole batch run on the fused one-lane-per-element kernels?  Only those take the affine-in / (X, Y, Z)-out flags (LadderArgs::io):
// then the lane lifts its own point and leaves the three coordinates R1toAffine reads -- no lift kernel, no R1 rows.  Batches with a
// two-lane tail or on the two-kernel route keep the separate lift and full R1 rows (FOURQ_FUSED_IO=0, a test hook, forces that everywhere).
static bool fused_io(const fourq_ctx* c, int algo, size_t n) { return c->fused_io && variable_route(c, algo, false, n, false) == ROUTE_FUSED; }
static int mul_affine_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points_affine, uint64_t* out_affine, size_t n) {
    if (!c || !scalars || !points_affine || !out_affine || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(points_affine) || !aligned16(out_affine)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    int rc = ensure_work(c, mul_affine_work_bytes(n));
    if (rc) return rc;
    uint64_t* r1_in = (uint64_t*)c->work;
    uint64_t* r1_out = (uint64_t*)(c->work + n * 160);
    if (fused_io(c, algo, n)) {
        if ((rc = mul_dev(c, algo, scalars, points_affine, nullptr, r1_out, nullptr, n, LADDER_IO_AFFINE_IN | LADDER_IO_XYZ_OUT))) return rc;
        return launch_lower(c, false, r1_out, 12, nullptr, out_affine, nullptr, n);
    }
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(lift_affine_kernel, dim3(grid), dim3(BLOCK), 0, c->stream, points_affine, r1_in, (u32)n);
    HIP_TRY(c, hipGetLastError());
    if ((rc = mul_dev(c, algo, scalars, r1_in, nullptr, r1_out, nullptr, n))) return rc;
    return launch_lower(c, false, r1_out, 20, nullptr, out_affine, nullptr, n);
}
static int mul_affine_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points_affine, uint64_t* out_affine, size_t n) {
    if (!c || !scalars || !points_affine || !out_affine || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[2] = { { (const char*)scalars, nullptr, 32 }, { (const char*)points_affine, nullptr, 64 } };
    PipeArray o[1] = { { nullptr, (char*)out_affine, 64 } };
    return run_pipeline(c, in, 2, o, 1, n, pipe_chunk(c, true), PipeRoute{ PR_AFF + (algo == ENDO ? 0 : 1), (algo == ENDO ? KT_ENDO_VAR : KT_WIN_VAR) + KT_LIFT_LOWER }, [&](char* const* di, char* const* dout, size_t m) {
        return mul_affine_dev(c, algo, (const uint64_t*)di[0], (const uint64_t*)di[1], (uint64_t*)dout[0], m);
    }, 0, [&](size_t big) { return ensure_work(c, mul_affine_work_bytes(big)); });
}
// decode -> MUL_<algo> -> encode.  status: 0 ok | 16 + FOURQ_DECODE_* (out32 is zero then); MUL_* itself cannot fail.
static int mul_bytes_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint8_t* points32, uint8_t* out32, uint8_t* status, size_t n) {
    if (!c || !scalars || !points32 || !out32 || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(points32) || !aligned16(out32)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    int rc = ensure_work(c, mul_affine_work_bytes(n));
    if (rc) return rc;
    uint64_t* r1_in = (uint64_t*)c->work;                       // decoded (and lifted) points
    uint64_t* r1_out = (uint64_t*)(c->work + n * 160);
    uint8_t* st_decode = (uint8_t*)(c->work + 2 * n * 160 + n * 64);
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    if (fused_io(c, algo, n)) {                                 // decode to affine rows; the ladder's lanes lift them and leave (X, Y, Z)
        hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(BLOCK), 0, c->stream, (const u64*)points32, r1_in, st_decode, (u32)n);
        HIP_TRY(c, hipGetLastError());
        if ((rc = mul_dev(c, algo, scalars, r1_in, nullptr, r1_out, nullptr, n, LADDER_IO_AFFINE_IN | LADDER_IO_XYZ_OUT))) return rc;
        return launch_lower(c, true, r1_out, 12, st_decode, (uint64_t*)out32, status, n);
    }
    hipLaunchKernelGGL(decode_lift_kernel, dim3(grid), dim3(BLOCK), 0, c->stream, (const u64*)points32, r1_in, st_decode, (u32)n);
    HIP_TRY(c, hipGetLastError());
    if ((rc = mul_dev(c, algo, scalars, r1_in, nullptr, r1_out, nullptr, n))) return rc;
    return launch_lower(c, true, r1_out, 20, st_decode, (uint64_t*)out32, status, n);
}
static int mul_bytes_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint8_t* points32, uint8_t* out32, uint8_t* status, size_t n) {
    if (!c || !scalars || !points32 || !out32 || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    PipeArray in[2] = { { (const char*)scalars, nullptr, 32 }, { (const char*)points32, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)out32, 32 }, { nullptr, (char*)status, 1 } };
    return run_pipeline(c, in, 2, o, 2, n, pipe_chunk(c, true), PipeRoute{ PR_BYTES + (algo == ENDO ? 0 : 1), (algo == ENDO ? KT_ENDO_VAR : KT_WIN_VAR) + KT_LIFT_LOWER + KT_CODEC }, [&](char* const* di, char* const* dout, size_t m) {
        return mul_bytes_dev(c, algo, (const uint64_t*)di[0], (const uint8_t*)di[1], (uint8_t*)dout[0], (uint8_t*)dout[1], m);
    }, 0, [&](size_t big) { return ensure_work(c, mul_affine_work_bytes(big)); });
}
FQ_API int fourq_mul_endo_affine_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) { return mul_affine_dev(c, ENDO, s, p, o, n); }
FQ_API int fourq_mul_windowed_affine_batch_dev(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) { return mul_affine_dev(c, WINDOWED, s, p, o, n); }
FQ_API int fourq_mul_endo_affine_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) { return mul_affine_host(c, ENDO, s, p, o, n); }
FQ_API int fourq_mul_windowed_affine_batch(fourq_ctx* c, const uint64_t* s, const uint64_t* p, uint64_t* o, size_t n) { return mul_affine_host(c, WINDOWED, s, p, o, n); }
FQ_API int fourq_mul_endo_bytes_batch_dev(fourq_ctx* c, const uint64_t* s, const uint8_t* p, uint8_t* o, uint8_t* st, size_t n) { return mul_bytes_dev(c, ENDO, s, p, o, st, n); }
FQ_API int fourq_mul_windowed_bytes_batch_dev(fourq_ctx* c, const uint64_t* s, const uint8_t* p, uint8_t* o, uint8_t* st, size_t n) { return mul_bytes_dev(c, WINDOWED, s, p, o, st, n); }
FQ_API int fourq_mul_endo_bytes_batch(fourq_ctx* c, const uint64_t* s, const uint8_t* p, uint8_t* o, uint8_t* st, size_t n) { return mul_bytes_host(c, ENDO, s, p, o, st, n); }
FQ_API int fourq_mul_windowed_bytes_batch(fourq_ctx* c, const uint64_t* s, const uint8_t* p, uint8_t* o, uint8_t* st, size_t n) { return mul_bytes_host(c, WINDOWED, s, p, o, st, n); }

FQ_API int fourq_dh_endo_bytes_batch_dev(fourq_ctx* c, const uint64_t* s, const uint8_t* k, const uint64_t* t, uint8_t* o, uint8_t* st, size_t n) {
    return dh_bytes_dev(c, ENDO, s, k, t, o, st, n);
}
FQ_API int fourq_dh_windowed_bytes_batch_dev(fourq_ctx* c, const uint64_t* s, const uint8_t* k, const uint64_t* t, uint8_t* o, uint8_t* st, size_t n) {
    return dh_bytes_dev(c, WINDOWED, s, k, t, o, st, n);
}
FQ_API int fourq_dh_endo_bytes_batch(fourq_ctx* c, const uint64_t* s, const uint8_t* k, const uint64_t* t, uint8_t* o, uint8_t* st, size_t n) {
    return dh_bytes_host(c, ENDO, s, k, t, o, st, n);
}
FQ_API int fourq_dh_windowed_bytes_batch(fourq_ctx* c, const uint64_t* s, const uint8_t* k, const uint64_t* t, uint8_t* o, uint8_t* st, size_t n) {
    return dh_bytes_host(c, WINDOWED, s, k, t, o, st, n);
}

// dh_exchange: DH_endo(a_i, DH_endo(b_i, base [, table392])) with the first half's public keys kept on the device
FQ_API int fourq_dh_exchange_batch_dev(fourq_ctx* c, const uint64_t* a, const uint64_t* b, const uint64_t* base_affine, const uint64_t* table392,
                                       uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !a || !b || !base_affine || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    int rc = ensure_work(c, exchange_work_bytes(n));
    if (rc) return rc;
    uint64_t* base = (uint64_t*)c->work;                      // the base point, once per exchange
    uint64_t* mid = (uint64_t*)(c->work + n * 64);            // DH(b_i, base): the public keys
    uint8_t* st_first = (uint8_t*)(c->work + 2 * n * 64);
    AffineArg point;
    memcpy(point.w, base_affine, sizeof point.w);             // read once, here: the launch carries it
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(broadcast_point_kernel, dim3(grid), dim3(BLOCK), 0, c->stream, point, base, (u32)n);
    HIP_TRY(c, hipGetLastError());
    if ((rc = dh_dev(c, ENDO, b, base, table392, mid, st_first, n))) return rc;
    if ((rc = dh_dev(c, ENDO, a, mid, nullptr, out, status, n))) return rc;
    hipLaunchKernelGGL(merge_status_kernel, dim3(grid), dim3(BLOCK), 0, c->stream, st_first, status, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_dh_exchange_batch(fourq_ctx* c, const uint64_t* a, const uint64_t* b, const uint64_t* base_affine, const uint64_t* table392,
                                   uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !a || !b || !base_affine || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    uint64_t base_copy[8];
    memcpy(base_copy, base_affine, sizeof base_copy);         // the caller's buffer is read once, here
    PipeArray in[2] = { { (const char*)a, nullptr, 32 }, { (const char*)b, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)out, 64 }, { nullptr, (char*)status, 1 } };
    // one generation of the fixed-base half (two waves per SIMD) = two of the variable-base half (one wave per SIMD)
    return run_pipeline(c, in, 2, o, 2, n, c->lanes_w4 / 2, PipeRoute{ PR_EXCH + (table392 ? 1 : 0), (table392 ? KT_DH_FIXED : KT_DH_VAR) + KT_DH_VAR }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_dh_exchange_batch_dev(c, (const uint64_t*)di[0], (const uint64_t*)di[1], base_copy, table392, (uint64_t*)dout[0], (uint8_t*)dout[1], m);
    }, c->lanes_w4, [&](size_t big) { if (int r = ensure_proj(c, big)) return r; return ensure_work(c, exchange_work_bytes(big)); });
}

// dh_exchange with the key-generation half through the comb (bench.py's cfg4 step as one call)
FQ_API int fourq_dh_exchange_comb_batch_dev(fourq_ctx* c, const uint64_t* a, const uint64_t* b, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !a || !b || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    int rc = ensure_work(c, exchange_work_bytes(n));
    if (rc) return rc;
    uint64_t* mid = (uint64_t*)(c->work + n * 64);            // [b_i]B: the public keys (same place as in fourq_dh_exchange_batch_dev)
    uint8_t* st_first = (uint8_t*)(c->work + 2 * n * 64);
    if ((rc = fourq_comb_mul_batch_dev(c, b, comb, mid, st_first, n))) return rc;
    if ((rc = dh_dev(c, ENDO, a, mid, nullptr, out, status, n))) return rc;
    hipLaunchKernelGGL(merge_status_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, st_first, status, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_dh_exchange_comb_batch(fourq_ctx* c, const uint64_t* a, const uint64_t* b, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !a || !b || !out || !status || n > FOURQ_MAX_BATCH) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    if (int rc = stage_comb(c, comb)) return rc;                            // compared once, not once per chunk
    PipeArray in[2] = { { (const char*)a, nullptr, 32 }, { (const char*)b, nullptr, 32 } };
    PipeArray o[2] = { { nullptr, (char*)out, 64 }, { nullptr, (char*)status, 1 } };
    return run_pipeline(c, in, 2, o, 2, n, c->lanes_w4 / 2, PipeRoute{ PR_EXCH_COMB, KT_COMB + KT_DH_VAR }, [&](char* const* di, char* const* dout, size_t m) {
        return fourq_dh_exchange_comb_batch_dev(c, (const uint64_t*)di[0], (const uint64_t*)di[1], nullptr, (uint64_t*)dout[0], (uint8_t*)dout[1], m);
    }, c->lanes_w4, [&](size_t big) { if (int r = ensure_proj(c, big)) return r; return ensure_work(c, exchange_work_bytes(big)); });
}

// ---- pinned host memory and transfer statistics of the host-pointer calls ---------------------------------------
FQ_API int fourq_host_alloc(fourq_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return FOURQ_OK;
}
FQ_API int fourq_host_free(fourq_ctx* c, void* ptr) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    HIP_TRY(c, hipHostFree(ptr));
    return FOURQ_OK;
}
FQ_API int fourq_ctx_set_host_timing(fourq_ctx* c, int on) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    c->host_timing = on != 0;
    return FOURQ_OK;
}
// The shader clock the chip holds right now, from inside a kernel (MI355X_MICROARCH "DVFS give-back" item 6): a probe of FOURQ_DIAG_BLOCKS
// single-wave blocks on a stream of the context's own (not the one its kernels are enqueued on), each timing `window_us` of the 100 MHz
// counter in shader cycles.  Call it while the context's stream has work queued for longer than the window (the _dev calls only enqueue)
// and the answer is the clock under THAT load: what turns a time measured on one box into cycles comparable with another's.
namespace {
constexpr int DIAG_BLOCKS = 16;                     // consecutive workgroups go to consecutive XCDs: two probes on each of the eight
int diag_alloc(fourq_ctx* c) {
    if (!c->diag_stamps) HIP_TRY(c, hipMalloc(&c->diag_stamps, 2 * DIAG_BLOCKS * sizeof(u64)));      // kept: hipFree would wait for the whole device
    return FOURQ_OK;
}
int diag_reduce(fourq_ctx* c, double* mhz_median, double* mhz_min, double* mhz_max, double* window_us) {
    u64 host[2 * DIAG_BLOCKS];
    HIP_TRY(c, hipMemcpyAsync(host, c->diag_stamps, sizeof host, hipMemcpyDeviceToHost, c->copy_out));
    HIP_TRY(c, hipStreamSynchronize(c->copy_out));
    double mhz[DIAG_BLOCKS], win = 0.0;
    for (int i = 0; i < DIAG_BLOCKS; i++) {
        mhz[i] = host[2 * i + 1] ? (double)host[2 * i] / ((double)host[2 * i + 1] / 100.0) : 0.0;
        win += (double)host[2 * i + 1] / 100.0 / DIAG_BLOCKS;
    }
    for (int i = 1; i < DIAG_BLOCKS; i++) for (int j = i; j > 0 && mhz[j] < mhz[j - 1]; j--) { const double t = mhz[j]; mhz[j] = mhz[j - 1]; mhz[j - 1] = t; }
    *mhz_median = 0.5 * (mhz[DIAG_BLOCKS / 2 - 1] + mhz[DIAG_BLOCKS / 2]);
    if (mhz_min) *mhz_min = mhz[0];
    if (mhz_max) *mhz_max = mhz[DIAG_BLOCKS - 1];
    if (window_us) *window_us = win;
    return FOURQ_OK;
}
}  // namespace
FQ_API int fourq_diag_clock(fourq_ctx* c, uint32_t window_us, double* mhz_median, double* mhz_min, double* mhz_max, int* under_load) {
    if (!c || !mhz_median || window_us == 0 || window_us > 1000000) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    if (int rc = diag_alloc(c)) return rc;
    // the end of the stream's backlog as it stands now: still pending when the window has closed = the window lay inside the load
    if (!c->diag_mark) HIP_TRY(c, hipEventCreateWithFlags(&c->diag_mark, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->diag_mark, c->stream));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(DIAG_BLOCKS), dim3(64), 0, c->copy_out, c->diag_stamps, (u64)window_us * 100u);
    HIP_TRY(c, hipGetLastError());
    int rc = diag_reduce(c, mhz_median, mhz_min, mhz_max, nullptr);
    // Was the context's stream still busy when the window closed?  If the backlog was shorter than the window -- or the probe shared a hardware
    // queue with the kernels and waited behind them -- it timed an idle chip: the caller must be able to tell.  (An event, not
    // hipStreamQuery: that one still answered hipErrorNotReady 19 ms after a 1 ms backlog had drained.)
    if (under_load) {
        const hipError_t q = hipEventQuery(c->diag_mark);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(c, q, "hipEventQuery(diag_mark)");
        *under_load = q == hipErrorNotReady ? 1 : 0;
    }
    return rc;
}
// The bracket form: begin and stop each enqueue one launch of clock_stamp_kernel on the context's stream -- before and behind the work the
// caller enqueues in between, so the host never waits; end waits for the stream, pairs the stamps per CU and reduces.
namespace {
constexpr int DIAG_STAMP_BLOCKS = 1024;             // single-wave workgroups: about four per CU, so that (nearly) every CU is in both launches
int diag_stamp(fourq_ctx* c, int which) {
    if (!c->diag_bracket) HIP_TRY(c, hipMalloc(&c->diag_bracket, 2 * 3 * DIAG_STAMP_BLOCKS * sizeof(u64)));
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(DIAG_STAMP_BLOCKS), dim3(64), 0, c->stream, c->diag_bracket + (size_t)which * 3 * DIAG_STAMP_BLOCKS);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
}  // namespace
FQ_API int fourq_diag_clock_begin(fourq_ctx* c) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    if (c->diag_open) return FOURQ_ERR_INVALID;
    if (int rc = diag_stamp(c, 0)) return rc;
    c->diag_open = true;
    c->diag_stopped = false;
    return FOURQ_OK;
}
FQ_API int fourq_diag_clock_stop(fourq_ctx* c) {
    if (!c) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    if (!c->diag_open || c->diag_stopped) return FOURQ_ERR_INVALID;
    if (int rc = diag_stamp(c, 1)) return rc;
    c->diag_stopped = true;
    return FOURQ_OK;
}
FQ_API int fourq_diag_clock_end(fourq_ctx* c, double* mhz_median, double* mhz_min, double* mhz_max, double* window_us) {
    if (!c || !mhz_median) return FOURQ_ERR_INVALID;
    CtxGuard g(c);
    if (!c->diag_open) return FOURQ_ERR_INVALID;
    if (!c->diag_stopped) { if (int rc = diag_stamp(c, 1)) return rc; }
    c->diag_open = c->diag_stopped = false;
    std::vector<u64> h;
    try { h.resize(2 * 3 * DIAG_STAMP_BLOCKS); } catch (...) { return FOURQ_ERR_NOMEM; }
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->diag_bracket, h.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // one (memtime, memrealtime) pair per CU and launch -- the first block that landed on it; then per CU seen in both: cycles per 10 ns
    const u64 *a = h.data(), *b = h.data() + 3 * DIAG_STAMP_BLOCKS;
    std::vector<double> mhz;
    double win = 0.0;
    try {
        std::vector<std::pair<u64, int>> first;          // (CU key, block of the first launch)
        for (int i = 0; i < DIAG_STAMP_BLOCKS; i++) first.emplace_back(a[3 * i], i);
        std::sort(first.begin(), first.end());
        std::vector<bool> used(DIAG_STAMP_BLOCKS, false);
        for (int j = 0; j < DIAG_STAMP_BLOCKS; j++) {
            auto it = std::lower_bound(first.begin(), first.end(), std::make_pair(b[3 * j], 0));
            if (it == first.end() || it->first != b[3 * j]) continue;
            const int slot = (int)(it - first.begin());
            if (used[slot]) continue;                    // this CU is paired already
            used[slot] = true;
            const int i = it->second;
            const double ticks = (double)(b[3 * j + 2] - a[3 * i + 2]);
            if (ticks <= 0) continue;
            mhz.push_back((double)(b[3 * j + 1] - a[3 * i + 1]) / (ticks / 100.0));
            win += ticks / 100.0;
        }
    } catch (...) { return FOURQ_ERR_NOMEM; }
    if (mhz.size() < 8) { snprintf(c->err, sizeof c->err, "fourq_diag_clock_end: only %zu CUs were stamped by both launches", mhz.size()); return FOURQ_ERR_HIP; }
    std::sort(mhz.begin(), mhz.end());
    const size_t m = mhz.size();
    *mhz_median = 0.5 * (mhz[(m - 1) / 2] + mhz[m / 2]);
    if (mhz_min) *mhz_min = mhz[m / 20];                 // 5th and 95th percentile over the CUs: the spread of the chip's clock domains, not of stragglers
    if (mhz_max) *mhz_max = mhz[m - 1 - m / 20];
    if (window_us) *window_us = win / (double)m;
    return FOURQ_OK;
}
FQ_API int fourq_ctx_host_stats(const fourq_ctx* c, fourq_host_stats* out) {
    if (!c || !out) return FOURQ_ERR_INVALID;
    *out = c->host_stats;
    return FOURQ_OK;
}

FQ_API int fourq_ctx_host_chunk_stamps(const fourq_ctx* c, uint32_t chunk, double out_ms[6]) {
    if (!c || !out_ms) return FOURQ_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock(c->mu);
    if ((size_t)chunk * 6 + 6 > c->chunk_stamps.size()) return FOURQ_ERR_INVALID;
    // event order in the pipeline: copy-in start, copy-in end, copy-out start, copy-out end, kernels start, kernels end
    for (int i = 0; i < 6; i++) out_ms[i] = c->chunk_stamps[(size_t)chunk * 6 + i];
    return FOURQ_OK;
}
FQ_API int fourq_ctx_host_stats_sized(const fourq_ctx* c, void* out, size_t size) {
    if (!c || !out) return FOURQ_ERR_INVALID;
    memcpy(out, &c->host_stats, size < sizeof c->host_stats ? size : sizeof c->host_stats);
    return FOURQ_OK;
}

FQ_API int fourq_prim_words(int op, size_t* in_words, size_t* out_words) {
    const PrimShape* p = find_prim(op);
    if (!p || !in_words || !out_words) return FOURQ_ERR_INVALID;
    *in_words = p->in_words; *out_words = p->out_words;
    return FOURQ_OK;
}
FQ_API int fourq_prim_batch(fourq_ctx* c, int op, const uint64_t* in, uint64_t* out, size_t n) {
    const PrimShape* p = find_prim(op);
    if (!c || !p || !in || !out || n > 0x7fffffffu) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    CtxGuard g(c);
    size_t ib = n * p->in_words * 8, ob = n * p->out_words * 8;
    int rc = ensure_stage(c, ib + ob);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, in, ib, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(prim_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, op, (const u64*)base, (u64*)(base + ib),
                       (u32)n, (u32)p->in_words, (u32)p->out_words);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, base + ib, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}

}  // extern "C"
