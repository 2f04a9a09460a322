// libfourq_amd.so -- first translation unit: the fused variable-base kernels, the small service kernels
// (tables, partition, wire format, primitives) and the C ABI declared in include/fourq_amd.h.  The
// fixed-base, two-kernel-route and comb kernels live in fourq_chain.hip; shared device code in kernels.hip.h.
//
// Kernel design (gfx950): one wavefront lane owns one (scalar, point) pair for the whole scalar
// multiplication.  Lanes never communicate; a workgroup is 256 lanes; the grid is sized to the
// number of resident lanes and strides over the batch.
//
//   variable base : the lane builds its own 8-entry R2 table (table_endo / table_windowed) into a 1 856-byte
//                   slot of HBM scratch (192 bytes per entry), then each ladder step gathers the coordinates
//                   of the entry its digit selects (wavefront-level gather, one entry per lane) a whole
//                   doubling ahead of their use.
//   fixed base    : the 8-entry table is staged once per workgroup into LDS (padded to dodge bank
//                   conflicts) and gathered from there.
//   selection     : the sign of a digit is applied branch-free (address choice for the N/D swap, a two-op
//                   conditional negation for F); the table index is a per-lane address, as in the reference
//                   (curve4q.py:232, :440).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#ifndef FQ_CHAIN
#define FQ_CHAIN 0
#endif
#include "kernels.hip.h"

using namespace fq;

namespace {

// packed table (8 x 16 words) -> working limbs (8 x 48 u32)
__global__ void table_unpack_kernel(const u64* packed, u32* limbs) {
    int k = threadIdx.x;
    if (k < 8) store_r2_limbs(limbs + k * R2_LIMBS, load_r2_packed(packed + 16 * k));
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
__global__ __launch_bounds__(BLOCK) void partition_kernel(const uint8_t* flags, u32 n, u32 first_id, u32* var_list, u32* slot_of, u32* counter) {
    __shared__ u32 scan[BLOCK], base;
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
    if (t == BLOCK - 1) base = atomicAdd(counter, scan[t]);
    __syncthreads();
    u32 rank = base + scan[t] - mine;
#pragma unroll
    for (int k = 0; k < PART_PER_LANE; k++) {
        if ((u32)k >= valid) break;
        if (f[k] != 0) { var_list[rank] = first_id + first + k; slot_of[first + k] = rank++; }
        else slot_of[first + k] = ~0u;
    }
}

// One lane per table entry: P[j][u] = [2^(e j) (1 + u0 2^d + u1 2^2d + u2 2^3d + u3 2^4d)] B by the ordinary
// variable-base MUL_endo, normalised to affine and stored as (x+y, y-x, 2d x y), 12 packed words.
__global__ __launch_bounds__(128) void comb_table_kernel(const u64* p_r1, u32* scratch, u64* comb) {
    const u32 t = threadIdx.x;
    if (t >= COMB_POINTS) return;
    const u32 j = t >> (COMB_W - 1), u = t & ((1u << (COMB_W - 1)) - 1);
    u64 m[4] = { 0, 0, 0, 0 };
    auto set_bit = [&](int bit) { m[bit >> 6] |= 1ull << (bit & 63); };
    set_bit(COMB_E * j);
    for (int r = 0; r < COMB_W - 1; r++) if ((u >> r) & 1) set_bit(COMB_E * j + (r + 1) * COMB_D);
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
    int k = threadIdx.x;
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
struct fourq_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int cus = 0;
    size_t lanes = 0;              // resident lanes of the fused variable-base kernels (1 wave per SIMD)
    size_t lanes_w4 = 0;           // resident lanes of the 128-VGPR kernels (4 waves per SIMD)
    size_t split_min = 0;          // variable-base batches of at least this many elements take the prep + ladder route
    size_t split_chunk = 0;        // elements per prep + ladder round (<= lanes_w4)
    bool split_all = false;        // FOURQ_SPLIT_ALL=1: also route plain MUL_endo through prep + ladder (tests)
    u32* scratch = nullptr;        // max(lanes, lanes_w4) x SLOT_U32
    u32* table_limbs = nullptr;    // 8 x 40
    u64* table_packed = nullptr;   // 128 words
    u32* comb_limbs = nullptr;     // 80 x 36 working limbs of the staged comb table
    u64* comb_packed = nullptr;    // 80 x 12 words
    uint64_t table_shadow[FOURQ_TABLE_WORDS];   // host copies of what table_limbs / comb_limbs currently hold
    uint64_t comb_shadow[FOURQ_COMB_WORDS];
    bool table_staged = false, comb_staged = false;
    u32* part_counter = nullptr;   // mixed batches: number of variable-base elements of the current round (device side)
    u32* part_list = nullptr;      // their ids, split_chunk entries
    u32* part_slot = nullptr;      // per element of the round: scratch slot of its table, ~0 = shared table
    uint4* proj = nullptr;         // deferred normalisation of DH batches: PROJ_PLANES planes of proj_capacity uint4, grown on demand
    size_t proj_capacity = 0;
    int norm_k = -1;               // FOURQ_NORM_K: 0 = always invert per element, 2/4/8 = always batch; -1 = by batch size
    void* stage = nullptr;         // staging for the host-pointer API
    size_t stage_bytes = 0;
    char err[256] = { 0 };
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
    if (SRC == FUSED) {                     // this translation unit's code object (FQ_CHAIN=0)
        hipLaunchKernelGGL((ladder_kernel<ALGO, FUSED, DH>), dim3(grid), dim3(BLOCK), 0, c->stream, a);
        HIP_TRY(c, hipGetLastError());
    } else {                                // fourq_chain.hip's (FQ_CHAIN=1)
        HIPRC_TRY(c, chain_launch_ladder(ALGO, SRC, DH, grid, c->stream, a));
    }
    return FOURQ_OK;
}
// Measured on MI355X at 2^20 elements: the two-kernel route gains for MUL_windowed and DH_* and is on par for
// plain MUL_endo, whose 64-step ladder hardly amortises the second launch and the colder table gathers (4 waves
// per SIMD put 486 MB of tables in flight, past the Infinity Cache); plain MUL_endo therefore stays fused.
bool takes_split_route(const fourq_ctx* c, int algo, bool dh, size_t n) {
    return (algo == WINDOWED || dh || c->split_all) && n >= c->split_min;
}
template <int ALGO, bool DH> int launch_variable(fourq_ctx* c, LadderArgs a) {
    if (!takes_split_route(c, ALGO, DH, a.n)) return launch_ladder<ALGO, FUSED, DH>(c, a);
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
int stage_table(fourq_ctx* c, const uint64_t* table_host) {
    if (c->table_staged && memcmp(c->table_shadow, table_host, sizeof c->table_shadow) == 0) return FOURQ_OK;
    c->table_staged = false;
    memcpy(c->table_shadow, table_host, sizeof c->table_shadow);
    HIP_TRY(c, hipMemcpyAsync(c->table_packed, c->table_shadow, FOURQ_TABLE_WORDS * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(table_unpack_kernel, dim3(1), dim3(64), 0, c->stream, c->table_packed, c->table_limbs);
    HIP_TRY(c, hipGetLastError());
    c->table_staged = true;
    return FOURQ_OK;
}

int mul_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
            const u32* index, size_t n) {
    if (!c || !scalars || !out || (!points && !table) || n > 0xffffffffu) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(out) || !aligned16(points)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    LadderArgs a = {};
    a.scalars = scalars; a.points = points; a.out = out; a.index = index; a.n = (u32)n;
    if (points) return algo == ENDO ? launch_variable<ENDO, false>(c, a) : launch_variable<WINDOWED, false>(c, a);
    int rc = stage_table(c, table);
    if (rc) return rc;
    return algo == ENDO ? launch_ladder<ENDO, LDS, false>(c, a) : launch_ladder<WINDOWED, LDS, false>(c, a);
}

int dh_dev(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
           uint8_t* status, size_t n) {
    if (!c || !scalars || !points || !out || !status || n > 0xffffffffu) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(points) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    LadderArgs a = {};
    a.scalars = scalars; a.points = points; a.out = out; a.status = status; a.n = (u32)n;
    int group = normalize_group(c, n);
    if (!table) {                                  // fused kernels invert in place; the prep + ladder route always defers
        const bool split = takes_split_route(c, algo, true, n);
        group = split ? (group ? group : 1) : 0;
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

// host-pointer wrappers: one staging buffer carved into [scalars | points | out | status]
int mul_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out, size_t n) {
    if (!c || !scalars || !out || (!points && !table)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    size_t sb = n * 32, pb = points ? n * 160 : 0, ob = n * 160;
    int rc = ensure_stage(c, sb + pb + ob);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, scalars, sb, hipMemcpyHostToDevice, c->stream));
    if (points) HIP_TRY(c, hipMemcpyAsync(base + sb, points, pb, hipMemcpyHostToDevice, c->stream));
    rc = mul_dev(c, algo, (const uint64_t*)base, points ? (const uint64_t*)(base + sb) : nullptr, table, (uint64_t*)(base + sb + pb), nullptr, n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, base + sb + pb, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
int dh_host(fourq_ctx* c, int algo, const uint64_t* scalars, const uint64_t* points, const uint64_t* table, uint64_t* out,
            uint8_t* status, size_t n) {
    if (!c || !scalars || !points || !out || !status) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    size_t sb = n * 32, pb = n * 64, ob = n * 64, tb = (n + 15) / 16 * 16;
    int rc = ensure_stage(c, sb + pb + ob + tb);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, scalars, sb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + sb, points, pb, hipMemcpyHostToDevice, c->stream));
    rc = dh_dev(c, algo, (const uint64_t*)base, (const uint64_t*)(base + sb), table, (uint64_t*)(base + sb + pb),
                (uint8_t*)(base + sb + pb + ob), n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, base + sb + pb, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, base + sb + pb + ob, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}

int table_host(fourq_ctx* c, int algo, const uint64_t* p_r1, uint64_t* table) {
    if (!c || !p_r1 || !table) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
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

FQ_API int fourq_version(void) { return 100; }

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
        // resident blocks per CU of the fused variable-base kernels (they own the per-lane scratch slots)
        int occ = 8, o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ladder_kernel<ENDO, FUSED, false>, BLOCK, 0) == hipSuccess && o > 0 && o < occ) occ = o;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, ladder_kernel<WINDOWED, FUSED, true>, BLOCK, 0) == hipSuccess && o > 0 && o < occ) occ = o;
        if (const char* env = getenv("FOURQ_BLOCKS_PER_CU")) { int v = atoi(env); if (v > 0 && v <= 8) occ = v; }
        c->lanes = (size_t)c->cus * occ * BLOCK;
        c->lanes_w4 = (size_t)c->cus * 4 * BLOCK;
        c->split_min = 2 * c->lanes;                       // below two full waves of fused work the second launch does not pay
        if (const char* env = getenv("FOURQ_SPLIT_MIN")) { long v = atol(env); if (v > 0) c->split_min = (size_t)v; }
        if (const char* env = getenv("FOURQ_SPLIT_ALL")) c->split_all = atoi(env) != 0;
        if (const char* env = getenv("FOURQ_NORM_K")) { int v = atoi(env); if (v == 0 || v == 2 || v == 4 || v == 8) c->norm_k = v; }
        c->split_chunk = c->lanes_w4;
        if (const char* env = getenv("FOURQ_SPLIT_CHUNK")) { long v = atol(env); if (v >= BLOCK && (size_t)v <= c->lanes_w4) c->split_chunk = (size_t)v; }
        size_t slots = c->lanes > c->lanes_w4 ? c->lanes : c->lanes_w4;
        if (hipMalloc(&c->scratch, slots * SLOT_U32 * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->table_limbs, 8 * R2_LIMBS * sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->table_packed, FOURQ_TABLE_WORDS * 8) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
        if (hipMalloc(&c->part_counter, sizeof(u32)) != hipSuccess) { rc = FOURQ_ERR_NOMEM; break; }
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
    DeviceGuard g(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->proj) (void)hipFree(c->proj);
    if (c->table_limbs) (void)hipFree(c->table_limbs);
    if (c->table_packed) (void)hipFree(c->table_packed);
    if (c->part_counter) (void)hipFree(c->part_counter);
    if (c->part_list) (void)hipFree(c->part_list);
    if (c->part_slot) (void)hipFree(c->part_slot);
    if (c->comb_limbs) (void)hipFree(c->comb_limbs);
    if (c->comb_packed) (void)hipFree(c->comb_packed);
    if (c->stage) (void)hipFree(c->stage);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return FOURQ_OK;
}

FQ_API int fourq_ctx_set_stream(fourq_ctx* c, void* hip_stream) {
    if (!c) return FOURQ_ERR_INVALID;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return FOURQ_OK;
}
FQ_API int fourq_ctx_sync(fourq_ctx* c) {
    if (!c) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
FQ_API int fourq_ctx_lanes(const fourq_ctx* c, size_t* lanes) {
    if (!c || !lanes) return FOURQ_ERR_INVALID;
    *lanes = c->lanes;
    return FOURQ_OK;
}

FQ_API int fourq_dev_alloc(fourq_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMalloc(out, bytes ? bytes : 16));
    return FOURQ_OK;
}
FQ_API int fourq_dev_free(fourq_ctx* c, void* ptr) {
    if (!c) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipFree(ptr));
    return FOURQ_OK;
}
FQ_API int fourq_dev_upload(fourq_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
FQ_API int fourq_dev_download(fourq_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return FOURQ_ERR_INVALID;
    DeviceGuard g(c->device);
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
    if (!c || !s || !p || !flags || !table || !o || n > 0xffffffffu) return FOURQ_ERR_INVALID;
    if (!aligned16(s) || !aligned16(p) || !aligned16(o)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    int rc = stage_table(c, table);
    if (rc) return rc;
    // Rounds of up to split_chunk elements.  Per round: compact the variable-base ids (count stays on the device),
    // build their tables into scratch slots (prep_kernel over the compacted list), then ONE ladder launch over all
    // elements of the round in their natural order: each lane reads its table through a pointer -- its own slot or
    // the shared fixed-base table -- so fixed and variable elements share wavefronts without divergence.
    const size_t per_block = (size_t)BLOCK * PART_PER_LANE;
    for (size_t off = 0; off < n; off += c->split_chunk) {
        const u32 m = (u32)(n - off < c->split_chunk ? n - off : c->split_chunk);
        HIP_TRY(c, hipMemsetAsync(c->part_counter, 0, sizeof(u32), c->stream));
        hipLaunchKernelGGL(partition_kernel, dim3((unsigned)((m + per_block - 1) / per_block)), dim3(BLOCK), 0, c->stream,
                           flags + off, m, (u32)off, c->part_list, c->part_slot, c->part_counter);
        HIP_TRY(c, hipGetLastError());
        LadderArgs a = {};
        a.scalars = s; a.points = p; a.out = o; a.n = m;
        a.scratch = c->scratch; a.table = c->table_limbs;
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
    if (!c || !s || !p || !flags || !table || !o) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    size_t sb = n * 32, pb = n * 160, ob = n * 160, fb = (n + 15) / 16 * 16;
    int rc = ensure_stage(c, sb + pb + ob + fb);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, s, sb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + sb, p, pb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + sb + pb + ob, flags, n, hipMemcpyHostToDevice, c->stream));
    rc = fourq_mul_endo_mixed_batch_dev(c, (const uint64_t*)base, (const uint64_t*)(base + sb), (const uint8_t*)(base + sb + pb + ob),
                                        table, (uint64_t*)(base + sb + pb), n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(o, base + sb + pb, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
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
    DeviceGuard g(c->device);
    int rc = ensure_stage(c, 160);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->stage, p_r1, 160, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(comb_table_kernel, dim3(1), dim3(128), 0, c->stream, (const u64*)c->stage, c->scratch, c->comb_packed);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(comb, c->comb_packed, FOURQ_COMB_WORDS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
FQ_API int fourq_comb_mul_batch_dev(fourq_ctx* c, const uint64_t* scalars, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !scalars || !comb || !out || !status || n > 0xffffffffu) return FOURQ_ERR_INVALID;
    if (!aligned16(scalars) || !aligned16(out)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    if (!c->comb_staged || memcmp(c->comb_shadow, comb, sizeof c->comb_shadow) != 0) {      // as stage_table
        c->comb_staged = false;
        memcpy(c->comb_shadow, comb, sizeof c->comb_shadow);
        HIP_TRY(c, hipMemcpyAsync(c->comb_packed, c->comb_shadow, FOURQ_COMB_WORDS * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(comb_unpack_kernel, dim3(1), dim3(128), 0, c->stream, c->comb_packed, c->comb_limbs);
        HIP_TRY(c, hipGetLastError());
        c->comb_staged = true;
    }
    const int group = normalize_group(c, n);
    int rc = group ? ensure_proj(c, n) : FOURQ_OK;
    if (rc) return rc;
    size_t blocks = (n + BLOCK - 1) / BLOCK, blocks_max = c->lanes_w4 / BLOCK;
    HIPRC_TRY(c, chain_launch_comb((unsigned)(blocks < blocks_max ? blocks : blocks_max), c->stream, scalars, c->comb_limbs, out, status,
                                   group ? c->proj : nullptr, (u32)c->proj_capacity, (u32)n));
    if (group) HIPRC_TRY(c, chain_launch_normalize(group, c->stream, c->proj, (u32)c->proj_capacity, out, status, (u32)n));
    return FOURQ_OK;
}
FQ_API int fourq_comb_mul_batch(fourq_ctx* c, const uint64_t* scalars, const uint64_t* comb, uint64_t* out, uint8_t* status, size_t n) {
    if (!c || !scalars || !comb || !out || !status) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    size_t sb = n * 32, ob = n * 64;
    int rc = ensure_stage(c, sb + ob + n + 16);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, scalars, sb, hipMemcpyHostToDevice, c->stream));
    rc = fourq_comb_mul_batch_dev(c, (const uint64_t*)base, comb, (uint64_t*)(base + sb), (uint8_t*)(base + sb + ob), n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, base + sb, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, base + sb + ob, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}

FQ_API int fourq_encode_batch_dev(fourq_ctx* c, const uint64_t* affine, uint8_t* out32, size_t n) {
    if (!c || !affine || !out32 || n > 0xffffffffu || !aligned16(affine) || !aligned16(out32)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, affine, (u64*)out32, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_decode_batch_dev(fourq_ctx* c, const uint8_t* in32, uint64_t* affine, uint8_t* status, size_t n) {
    if (!c || !in32 || !affine || !status || n > 0xffffffffu || !aligned16(in32) || !aligned16(affine)) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)in32, affine, status, (u32)n);
    HIP_TRY(c, hipGetLastError());
    return FOURQ_OK;
}
FQ_API int fourq_encode_batch(fourq_ctx* c, const uint64_t* affine, uint8_t* out32, size_t n) {
    if (!c || !affine || !out32) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    int rc = ensure_stage(c, n * 96);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, affine, n * 64, hipMemcpyHostToDevice, c->stream));
    rc = fourq_encode_batch_dev(c, (const uint64_t*)base, (uint8_t*)(base + n * 64), n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out32, base + n * 64, n * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FOURQ_OK;
}
FQ_API int fourq_decode_batch(fourq_ctx* c, const uint8_t* in32, uint64_t* affine, uint8_t* status, size_t n) {
    if (!c || !in32 || !affine || !status) return FOURQ_ERR_INVALID;
    if (n == 0) return FOURQ_OK;
    DeviceGuard g(c->device);
    int rc = ensure_stage(c, n * 97 + 16);
    if (rc) return rc;
    char* base = (char*)c->stage;
    HIP_TRY(c, hipMemcpyAsync(base, in32, n * 32, hipMemcpyHostToDevice, c->stream));
    rc = fourq_decode_batch_dev(c, (const uint8_t*)base, (uint64_t*)(base + n * 32), (uint8_t*)(base + n * 96), n);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(affine, base + n * 32, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, base + n * 96, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    DeviceGuard g(c->device);
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
