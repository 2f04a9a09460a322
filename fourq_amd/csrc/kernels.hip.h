// Shared device code of libfourq_amd.so: ladder arguments, table construction, the ladders and the kernels
// built from them.  Included by four translation units; the first two differ only in FQ_CHAIN (fp127.hip.h), the last
// two are their builds with constant-time table selection (ladder_kernel<..., CT = true>):
//   fourq_amd.hip       FQ_CHAIN=0  fused variable-base kernels (table + ladder in one launch), primitives, C ABI
//   fourq_chain.hip     FQ_CHAIN=1  fixed-base (LDS), two-kernel route (prep + PREBUILT ladder), comb
//   fourq_ct_fused.hip  FQ_CHAIN=0  fused kernels, the lane's table scanned in registers
//   fourq_ct_chain.hip  FQ_CHAIN=1  fixed-base (LDS) ladders and comb, the whole table read at every step
// Measured on MI355X (2^20 elements): chaining each column's carry into the next column's first multiply-add
// gains 4-10 % for every kernel of the second group and costs the fused kernels 12-15 % when applied wholesale;
// the fused kernels therefore use it only in their ladders, with preloaded table entries (see "the ladders").
// Designs that were measured and rejected are kept as patches under tools/experiments/, not as switches in this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/fourq_amd.h"
#include "curve.hip.h"
#include "ladder_asm.hip.h"
#include "recode.hip.h"
#include "pair.hip.h"

namespace fq {

constexpr int BLOCK = 256;
constexpr int SLOT_U32 = LimbSlots::SLOT;   // a whole-entry scratch slot (table_build_kernel, comb_table_kernel): 8 x 48 dwords + two parked points
// Layout of the slots that prep_kernel fills and ladder_kernel<PREBUILT> gathers from: 2^18 elements in flight make
// 486 MB of 192-byte entries, past the 256 MiB Infinity Cache, and the DH ladder of BASELINE config 4 then pulled
// 4 TB/s from HBM; packed 128-byte entries cut that by a third and let a round's tables stay in the cache.
// (The fused kernels keep ready-to-use limbs: N, D in dense NDSlots -- 2^16 slots = 42 MB, inside the Infinity Cache --
// and E, F in LDS, below.)
typedef PackedSlots PrebuiltSlots;
constexpr int PROJ_PLANES = 8;            // deferred normalisation: the 30 working limbs of (X, Y, Z) in eight uint4 planes
constexpr int LDS_ENTRY_U32 = 52;      // 48 + 4 pad: entry k starts at bank 52k mod 64 -> eight entries never share a b128 bank group

// A second copy of the E and F coordinates of the lane's eight entries in LDS (fused kernels, one wave per SIMD: 8 entries x
// 2 coordinates x 40 bytes x 256 lanes = exactly the CU's 160 KiB).  A ladder step then gathers only N and D from the lane's
// HBM slot -- bytes 0..95 of a 192-byte entry: two 64-byte sectors instead of three -- and reads E and F from LDS, with no
// extra arithmetic.  Layout [entry][coordinate][limb pair][lane] x 8 bytes: a wave's 64 lanes hit 64 distinct bank pairs
// whatever their digits are.  The constant-time fused kernels use the same layout (ScanSplit, curve.hip.h).
constexpr int EF_LDS_U32 = 8 * 2 * 5 * 2 * 256;               // dwords: 163 840 bytes
struct LdsEF {
    static constexpr bool ON = true;
    uint2* lane;                                               // the block's array + threadIdx.x
    FQ_DEV void put1(int k, int c, const Fe2<1>& v) const {
        uint2* q = lane + (size_t)((k * 2 + c) * 5) * 256;
        q[0] = make_uint2(v.re.l[0], v.re.l[1]); q[256] = make_uint2(v.re.l[2], v.re.l[3]); q[512] = make_uint2(v.re.l[4], v.im.l[0]);
        q[768] = make_uint2(v.im.l[1], v.im.l[2]); q[1024] = make_uint2(v.im.l[3], v.im.l[4]);
    }
    FQ_DEV void put(int k, const R2& t) const { put1(k, 0, t.E); put1(k, 1, t.F); }
    // The ten rows of an entry whose E, F are not there yet are free storage: table_endo parks its working values in them
    // (build_table_endo_lds) instead of making round trips through the HBM slot.
    FQ_DEV void park_nd(int k, const Fe2<1>& n, const Fe2<1>& d) const { put1(k, 0, n); put1(k, 1, d); }
    FQ_DEV void unpark_nd(int k, Fe2<1>& n, Fe2<1>& d) const { n = get((u32)k, 0); d = get((u32)k, 1); }
    FQ_DEV void park_xyz(int k, const Fe2<1>& x, const Fe2<1>& y, const Fe2<1>& z) const { put1(k, 0, x); put1(k, 1, y); put1(k + 1, 0, z); }
    FQ_DEV void unpark_xyz(int k, Fe2<1>& x, Fe2<1>& y, Fe2<1>& z) const { x = get((u32)k, 0); y = get((u32)k, 1); z = get((u32)k + 1, 0); }
    FQ_DEV Fe2<1> get(u32 k, int c) const {
        const uint2* q = lane + (size_t)((k * 2 + c) * 5) * 256;
        const uint2 a = q[0], b = q[256], d = q[512], e = q[768], f = q[1024];
        Fe2<1> r;
        r.re.l[0] = a.x; r.re.l[1] = a.y; r.re.l[2] = b.x; r.re.l[3] = b.y; r.re.l[4] = d.x;
        r.im.l[0] = d.y; r.im.l[1] = e.x; r.im.l[2] = e.y; r.im.l[3] = f.x; r.im.l[4] = f.y;
        return r;
    }
};
struct NoEF {
    static constexpr bool ON = false;
    FQ_DEV void put(int, const R2&) const {}
    FQ_DEV Fe2<1> get(u32, int) const { return Fe2<1>{}; }
    FQ_DEV void park_nd(int, const Fe2<1>&, const Fe2<1>&) const {}
    FQ_DEV void unpark_nd(int, Fe2<1>&, Fe2<1>&) const {}
    FQ_DEV void park_xyz(int, const Fe2<1>&, const Fe2<1>&, const Fe2<1>&) const {}
    FQ_DEV void unpark_xyz(int, Fe2<1>&, Fe2<1>&, Fe2<1>&) const {}
};

// entry k of a lane's table: with an LdsEF only N and D go to (come from) the HBM slot, E and F live in LDS alone
template <typename L, typename EF> FQ_DEV void store_entry(u32* slot, int k, const R2& t, const EF& ef) {
    if constexpr (EF::ON) {
        store_nd<L>(slot + k * L::ENTRY, t.N, t.D);
        ef.put(k, t);
    } else {
        store_r2<L>(slot + k * L::ENTRY, t);
    }
}
template <typename L, typename EF> FQ_DEV R2 load_entry_r2(const u32* slot, int k, const EF& ef) {
    if constexpr (EF::ON) {
        R2 t;
        load_nd<L>(slot + k * L::ENTRY, t.N, t.D);
        t.E = ef.get((u32)k, 0); t.F = ef.get((u32)k, 1);
        return t;
    } else {
        return load_r2<L>(slot + k * L::ENTRY);
    }
}

enum Algo { ENDO = 0, WINDOWED = 1 };
// where the ladder finds its table:
//   FUSED     built by the same lane into its scratch slot just before the ladder (small batches: one launch)
//   LDS       one shared table staged into LDS (fixed base)
//   PREBUILT  built per element by prep_kernel into scratch slot `pos` (large batches: the ladder kernel then
//             fits 128 VGPRs and runs 4 waves per SIMD instead of 1)
enum Src { FUSED = 0, LDS = 1, PREBUILT = 2 };

struct LadderArgs {
    const u64* scalars;    // n x 4
    const u64* points;     // variable base: n x 20 (R1) ; DH: n x 8 (affine) ; else unused
    u64* out;              // n x 20 (R1) or n x 8 (affine, DH)
    uint8_t* status;       // DH only
    const uint8_t* flags;  // mixed batches on the two- and four-lane kernels: flags[i] != 0 = variable base (points[i]), 0 = the fixed-base table
    const u32* index;      // optional: element ids to process (prep_kernel over the variable-base ids of a mixed batch); NULL = identity
    u32 base;              // first position of this launch (chunked large batches)
    const u32* base_dev;   // optional: added to `base`, read on the device (the overflow part of a list whose split is decided on the device)
    const u32* n_dev;      // optional: element count read on the device (mixed batches: no host round trip)
    const u32* slot_of;    // PREBUILT, optional (mixed batches): per position, the scratch slot of its table or ~0 = `table`
    const u32* table;      // fixed base: 8 x 48 working limbs (global), staged to LDS
    const u32* table_slots;// the same table in the PrebuiltSlots layout (mixed batches: read through a per-lane pointer)
    u32* scratch;          // variable base: NDSlots::SLOT dwords per resident lane (FUSED) or PrebuiltSlots::SLOT per position of the chunk (PREBUILT)
    uint4* proj;           // DH, optional: PROJ_PLANES x proj_stride; non-NULL selects the kernels that leave (X, Y, Z)
                           // there for normalize_kernel (always the case on the PREBUILT route)
    u32 proj_stride;       // elements per plane
    u32 n;
    u32 io;                // fused MUL_* kernels (one lane per element) only -- LADDER_IO_AFFINE_IN: `points` is n x 8 affine words and the lane lifts
                           // it itself (curve4q.py:100-101); LADDER_IO_XYZ_OUT: `out` is n x 12 words, (X, Y, Z) of the result only (what
                           // R1toAffine reads, curve4q.py:103-106).  Wave-uniform branches in front of and behind the ladder: the affine / encoded
                           // I/O flavours of MUL_* need no lift kernel and no 160-byte R1 rows on either side (round 6)
};
constexpr u32 LADDER_IO_AFFINE_IN = 1, LADDER_IO_XYZ_OUT = 2;

// ---- small helpers -----------------------------------------------------------------------------
FQ_DEV void store_xyz(u32* dst, const Fe2<1>& X, const Fe2<1>& Y, const Fe2<1>& Z) {
    const Fe2<1>* f[3] = { &X, &Y, &Z };
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int i = 0; i < 5; i++) { dst[10 * k + i] = f[k]->re.l[i]; dst[10 * k + 5 + i] = f[k]->im.l[i]; }
    }
}
FQ_DEV void load_xyz(const u32* src, Fe2<1>& X, Fe2<1>& Y, Fe2<1>& Z) {
    Fe2<1>* f[3] = { &X, &Y, &Z };
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int i = 0; i < 5; i++) { f[k]->re.l[i] = src[10 * k + i]; f[k]->im.l[i] = src[10 * k + 5 + i]; }
    }
}

// (X, Y, Z) of element `id` for normalize_kernel: plane p holds limbs 4p .. 4p+3 of the 30-limb record
// X.re X.im Y.re Y.im Z.re Z.im, so that consecutive elements are consecutive uint4s in every plane (coalesced
// for the ladder's stores and for the normaliser's loads alike); Z occupies planes 5..7.
FQ_DEV void store_proj(uint4* proj, u32 stride, u32 id, const Fe2<1>& X, const Fe2<1>& Y, const Fe2<1>& Z) {
    u32 w[32];
    store_xyz(w, X, Y, Z);
    w[30] = w[31] = 0;
#pragma unroll
    for (int p = 0; p < PROJ_PLANES; p++) proj[(size_t)p * stride + id] = make_uint4(w[4 * p], w[4 * p + 1], w[4 * p + 2], w[4 * p + 3]);
}
FQ_DEV Fe2<1> load_proj_z(const uint4* proj, u32 stride, u32 id) {
    uint4 a = proj[(size_t)5 * stride + id], b = proj[(size_t)6 * stride + id], c = proj[(size_t)7 * stride + id];
    Fe2<1> z;
    z.re.l[0] = a.x; z.re.l[1] = a.y; z.re.l[2] = a.z; z.re.l[3] = a.w; z.re.l[4] = b.x;
    z.im.l[0] = b.y; z.im.l[1] = b.z; z.im.l[2] = b.w; z.im.l[3] = c.x; z.im.l[4] = c.y;
    return z;
}
FQ_DEV void load_proj_xy(const uint4* proj, u32 stride, u32 id, Fe2<1>& X, Fe2<1>& Y) {
    u32 w[20];
#pragma unroll
    for (int p = 0; p < 5; p++) {
        uint4 v = proj[(size_t)p * stride + id];
        w[4 * p] = v.x; w[4 * p + 1] = v.y; w[4 * p + 2] = v.z; w[4 * p + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) { X.re.l[i] = w[i]; X.im.l[i] = w[5 + i]; Y.re.l[i] = w[10 + i]; Y.im.l[i] = w[15 + i]; }
}

// T[0] = R1toR2(P); T[i] = R1toR2(ADD(DBL(P), T[i-1]))                       curve4q.py:179-185
template <typename L = LimbSlots, typename EF = NoEF> FQ_DEV void build_table_windowed(const R1& P, u32* tbl, const EF& ef = EF()) {
    R3 twoP = r1_to_r3(dbl(P));
    R2 t = r1_to_r2(P);
    store_entry<L>(tbl, 0, t, ef);
#pragma unroll 1
    for (int i = 1; i < 8; i++) {
        t = r1_to_r2(add_core(twoP, as_signed(t)));
        store_entry<L>(tbl, i, t, ef);
    }
}

// T[k] = P + k0*phi(P) + k1*psi(P) + k2*psi(phi(P)), built in the reference's order      curve4q.py:385-403
//   step 0: Q = phi(P) -> T[1] = Q + T[0]
//   step 1: R = psi(P) -> T[2] = R + T[0], T[3] = R + T[1]
//   step 2: S = psi(Q) -> T[4..7] = S + T[0..3]
// tau and tau_dual are shared by the three steps (one code instance each): the working points P and
// Q are parked in the lane's scratch slot so that they do not pin 60 VGPRs across the endomorphisms.
template <typename L = LimbSlots> FQ_DEV void build_table_endo(const R1& P, u32* slot) {
    constexpr int SLOT_P = L::PARK_P, SLOT_Q = L::PARK_Q;
    static_assert(SLOT_Q > SLOT_P && SLOT_Q + 30 <= L::SLOT, "this builder parks two points in the slot");
    store_r2<L>(slot, r1_to_r2(P));
    store_xyz(slot + SLOT_P, P.X, P.Y, P.Z);
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        Fe2<1> X, Y, Z;
        load_xyz(slot + (step == 2 ? SLOT_Q : SLOT_P), X, Y, Z);
        Proj<1, 2, 1> t;
        if (step == 1) {                 // tau(P), parked by step 0 in P's place (phi and psi share it, curve4q.py:318-322)
            t.X = X; t.Y = widen<2>(Y); t.Z = Z;
        } else {
            t = tau(X, Y, Z);
            if (step == 0) store_xyz(slot + SLOT_P, t.X, fe2_carry(t.Y), t.Z);
        }
        Proj<2, 2, 2> u;
        if (step == 0) {
            u = upsilon(t);
        } else {
            Proj<1, 1, 1> c = chi(t);
            u.X = widen<2>(c.X); u.Y = widen<2>(c.Y); u.Z = widen<2>(c.Z);
        }
        R1 V = tau_dual(u.X, u.Y, u.Z);
        if (step == 0) store_xyz(slot + SLOT_Q, V.X, V.Y, V.Z);
        R3 V3 = r1_to_r3(V);
        int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 base = load_r2<L>(slot + m * L::ENTRY);
            store_r2<L>(slot + (half + m) * L::ENTRY, r1_to_r2(add_core(V3, as_signed(base))));
        }
    }
}

// The same table with its slot traffic reordered (prep_kernel): on gfx950 loads and stores retire through one in-order counter,
// so a load issued after a store waits for that store to drain to L2 as well (microseconds when 64 lanes scatter 16 bytes each to
// their own lines).  build_table_endo above reads back ten times right behind its own stores.  Here every read-back is issued
// BEFORE the stores of the stretch it belongs to: the base T[0] and the next step's working point ahead of this step's parking /
// result stores, the base T[m+1] ahead of the store of T[half+m]; step 0 takes P from registers.  Same values, same table.
template <typename L = LimbSlots> FQ_DEV void build_table_endo_pipelined(const R1& P, u32* slot) {
    constexpr int SLOT_P = L::PARK_P, SLOT_Q = L::PARK_Q;
    static_assert(SLOT_Q > SLOT_P && SLOT_Q + 30 <= L::SLOT, "this builder parks two points in the slot");
    store_r2<L>(slot, r1_to_r2(P));
    Fe2<1> X = P.X, Y = P.Y, Z = P.Z;
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        Proj<1, 2, 1> t;
        if (step == 1) {                 // tau(P), parked by step 0 (phi and psi share it, curve4q.py:318-322)
            t.X = X; t.Y = widen<2>(Y); t.Z = Z;
        } else {
            t = tau(X, Y, Z);
            if (step == 0) store_xyz(slot + SLOT_P, t.X, fe2_carry(t.Y), t.Z);
        }
        Proj<2, 2, 2> u;
        if (step == 0) {
            u = upsilon(t);
        } else {
            Proj<1, 1, 1> c = chi(t);
            u.X = widen<2>(c.X); u.Y = widen<2>(c.Y); u.Z = widen<2>(c.Z);
        }
        R1 V = tau_dual(u.X, u.Y, u.Z);
        R2 base = load_r2<L>(slot);                                              // T[0]: behind stores that drained an endomorphism ago
        if (step < 2) load_xyz(slot + (step == 0 ? SLOT_P : SLOT_Q), X, Y, Z);   // next step's input; step 1 reads the Q parked by step 0
        if (step == 0) store_xyz(slot + SLOT_Q, V.X, V.Y, V.Z);
        R3 V3 = r1_to_r3(V);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 next = base;
            if (m + 1 < half) next = load_r2<L>(slot + (m + 1) * L::ENTRY);      // ahead of this iteration's store
            store_r2<L>(slot + (half + m) * L::ENTRY, r1_to_r2(add_core(V3, as_signed(base))));
            base = next;
        }
    }
}

// ---- the ladders -------------------------------------------------------------------------------
// Every ladder runs the products with chained carries on SIGNED limbs (CH = 2, fp127.hip.h "signed flavour": 5 % fewer
// instructions per step; per-kernel A/Bs in DESIGN.md section 9 -- round 1 adopted it for the LDS and mixed-batch ladders,
// round 2's layouts made it pay in the fused and the split ladders too).
// The fused kernels run one wave per SIMD: their ladder preloads the entry of a step into registers a whole doubling ahead
// (PRELOAD; measured 0.416 ms per 2^16 batch against 0.423 with plain products and 0.463 chained without the preload: hipcc
// cannot hoist loads across the opaque partial sums).
constexpr int LADDER_CH = 2;
// CH = 3: the same signed DAG as hand-scheduled asm bodies (ladder_asm.hip.h, tools/asmgen/gen_ladder_step.py): the fused
// kernels' ladders, whose lone wave per SIMD pays for every issue slot (profiles/r04_ladder_step.txt).
#ifndef FQ_LADDER_ASM
#define FQ_LADDER_ASM 1
#endif
// the fixed-base ladders (table in LDS) on the same bodies: 256 VGPRs, two waves per SIMD instead of four
#ifndef FQ_LDS_ASM
#define FQ_LDS_ASM 1
#endif
// the two-kernel route's MUL ladder likewise (the packed entry requested a doubling ahead, unpacked behind it): mixed batches
// +3.7 %; its DH flavour, whose gathers run at the memory system's pace, loses 2.5-4 % with two wave slots and keeps four
#ifndef FQ_PREBUILT_ASM
#define FQ_PREBUILT_ASM 1
#endif
// the constant-time ladders on the asm bodies: the whole-table scan (select trees, compiler code) produces the entry, the bodies consume it
#ifndef FQ_CT_LDS_ASM
#define FQ_CT_LDS_ASM 1
#endif
#ifndef FQ_CT_FUSED_ASM
#define FQ_CT_FUSED_ASM 1
#endif
// TOUCH: the two-kernel route's ladder reads a lane's table entry (one 128-byte line of its scratch slot) inside the
// addition, with no registers to spare for issuing the eight loads a doubling ahead (128-VGPR budget).  A one-dword load of the
// line at the top of the step, result unused, starts the HBM / Infinity-Cache fetch early: the real loads then hit L2.
// Same-box A/B (profiles/r02_split_route.txt): cfg5 +2.2 % (two waves per SIMD), cfg4 +0..1 % (four waves hide the latency themselves).
// The touch is an ordinary load that hipcc sees: it allocates the landing register and places the s_waitcnt that covers it
// itself (round 2 issued the load from inline asm, where a spill of the untracked register would have corrupted a live limb).
// touch_done() after the addition is its only consumer; `after` is a limb of the sum's X, which depends on all four coordinates
// of the entry, so the consumer -- and with it the wait -- cannot move up in front of the addition's own loads, and the
// scheduling barrier behind the load keeps the load itself at the top of the step.
template <typename TP> FQ_DEV u32 touch_line(const TP* p) {
    const u32 landing = *reinterpret_cast<const u32*>(p);
    __builtin_amdgcn_sched_barrier(0);
    return landing;
}
FQ_DEV void touch_done(u32 landing, u32& after) { asm volatile("" : "+v"(after) : "v"(landing)); }
// a ladder on signed limbs (CH == 2) hands its result back with non-negative limbs
template <int CH> FQ_DEV R1 ladder_result(const R1& Q) {
    if constexpr (CH >= 2) {
        R1 r;
        r.X = fe2_unsign(Q.X); r.Y = fe2_unsign(Q.Y); r.Z = fe2_unsign(Q.Z);
        r.Ta = widen<4>(fe2_unsign(Q.Ta)); r.Tb = widen<2>(fe2_unsign(Q.Tb));
        return r;
    } else {
        return Q;
    }
}
template <int CH = (FQ_CHAIN != 0) ? 1 : 0, bool PRELOAD = false, typename L = LimbSlots, typename EF = NoEF, bool TOUCH = false, typename TP> FQ_DEV R1 ladder_endo(const EndoDigits& e, const TP* tbl, int stride, const EF& ef = EF()) {   // curve4q.py:436-442
    Proj<1, 1, 1> q4 = start_table<L>(tbl + (e.top & 7) * stride, 0u);        // s[64] = 1: the entry itself
    if constexpr (EF::ON) q4.Z = ef.get(e.top & 7, 0);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        const u32 digit = endo_digit(e, i);
        const TP* entry = tbl + digit * stride;
        if constexpr (CH == 3) {                  // the hand-scheduled bodies (ladder_asm.hip.h); the entry is requested ahead of the doubling
            static_assert(PRELOAD, "the asm bodies take the entry in registers");
            const u32 neg = endo_neg_mask(e, i);
            EntryRegs t = load_entry<L>(entry, neg, digit, ef);
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, t, neg);
        } else if (PRELOAD) {
            const u32 neg = endo_neg_mask(e, i);
            EntryRegs t = load_entry<L>(entry, neg, digit, ef);
            Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_entry<CH>(Q, t, neg);
        } else {
            u32 landing = 0;
            if (TOUCH) landing = touch_line(entry);
            Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_table<CH, L>(Q, entry, endo_neg_mask(e, i));
            if (TOUCH) touch_done(landing, Q.X.re.l[0]);
        }
    }
    return ladder_result<CH>(Q);
}
// The fused kernels' ladder on the nibble stream (recode_nibbles) and with the lane's entries addressed as a 32-bit dword offset on the
// kernel's scratch pointer (`base` is wave-uniform: the loads take it as their scalar base).  Same steps, same bodies, same R1 tuple as
// ladder_endo<3, true, L, EF>; what changes is the glue between the bodies: per step one bit-field extract, one arithmetic shift (the
// negation mask), one shift of the stream and two multiply-adds for the two addresses, instead of four 64-bit shifts, five bit
// operations, two multiplications and a 64-bit address addition.
#ifndef FQ_NIBBLE_LADDER
#define FQ_NIBBLE_LADDER 1
#endif
// DH batches of at least two generations on the fused kernels leave (X, Y, Z) for normalize_kernel<K> as the other routes do (round 6), instead
// of inverting per element at one wave per SIMD
#ifndef FQ_FUSED_DEFER
#define FQ_FUSED_DEFER 1
#endif
// SIGNED_OUT: hand the ladder's point back on signed limbs (bound 1 each) -- for store_r1_signed, which folds the bias into the canonical
// reduction of the final store
template <typename L, typename EF, bool SIGNED_OUT = false> FQ_DEV R1 ladder_endo_nibbles(const EndoNibbles& e, const u32* base, u32 lane_off, const EF& ef) {   // curve4q.py:436-442
    static_assert(EF::ON, "the fused kernels' ladder: E, F in the lane's LDS rows");
    Proj<1, 1, 1> q4 = start_table<L>(base + (size_t)(lane_off + (e.top & 7) * (u32)L::ENTRY), 0u);     // s[64] = 1: the entry itself
    q4.Z = ef.get(e.top & 7, 0);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
    u32 w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) w[t] = e.w[t];
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        u32 cur = w[7];
#pragma unroll
        for (int t = 7; t > 0; t--) w[t] = w[t - 1];
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const u32 digit = (cur >> 28) & 7;                                    // top nibble, bits 0..2
            const u32 neg = (u32)((int32_t)cur >> 31);                            // its bit 3, spread: ~0 when the step subtracts
            cur <<= 4;
            EntryRegs t = load_entry<L>(base + (size_t)(lane_off + digit * (u32)L::ENTRY), neg, digit, ef);
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, t, neg);
        }
    }
    if constexpr (SIGNED_OUT) return Q;
    else return ladder_result<3>(Q);
}
// canonical words of a point the asm bodies left on SIGNED limbs (every limb of every coordinate within bound 1): bias, then fe_canon's
// own carry -- what ladder_result<3> + store_r1 compute, with one carry chain per field element less
template <int B> FQ_DEV void store_fe2_signed(u64* w, const Fe2<B>& a) {
    fe_canon(fe_unsign_wide(a.re), w[0], w[1]); fe_canon(fe_unsign_wide(a.im), w[2], w[3]);
}
FQ_DEV void store_r1_signed(u64* w, const R1& q) {
    store_fe2_signed(w, q.X); store_fe2_signed(w + 4, q.Y); store_fe2_signed(w + 8, q.Z); store_fe2_signed(w + 12, q.Ta); store_fe2_signed(w + 16, q.Tb);
}
// The same ladder over any table a pointer and an entry stride describe -- the shared fixed-base table in LDS (ladder_kernel<ENDO, LDS>) or in
// global memory (the fixed-base items of mixed_queue_kernel): the nibble stream's glue with ladder_endo's addressing.
template <typename L, typename EF, typename TP> FQ_DEV R1 ladder_endo_nibbles_at(const EndoNibbles& e, const TP* tbl, int stride, const EF& ef = EF()) {   // curve4q.py:436-442
    Proj<1, 1, 1> q4 = start_table<L>(tbl + (e.top & 7) * stride, 0u);
    if constexpr (EF::ON) q4.Z = ef.get(e.top & 7, 0);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
    u32 w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) w[t] = e.w[t];
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        u32 cur = w[7];
#pragma unroll
        for (int t = 7; t > 0; t--) w[t] = w[t - 1];
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const u32 digit = (cur >> 28) & 7;
            const u32 neg = (u32)((int32_t)cur >> 31);
            cur <<= 4;
            EntryRegs t = load_entry<L>(tbl + digit * stride, neg, digit, ef);
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, t, neg);
        }
    }
    return ladder_result<3>(Q);
}
template <int CH = (FQ_CHAIN != 0) ? 1 : 0, bool PRELOAD = false, typename L = LimbSlots, typename EF = NoEF, bool TOUCH = false, typename TP> FQ_DEV R1 ladder_windowed(const WinScalar& w, const TP* tbl, int stride, const EF& ef = EF()) {   // curve4q.py:228-235
    u32 code = win_top_code(w);
    Proj<1, 1, 1> q4 = start_table<L>(tbl + (code & 7) * stride, (code >> 3) - 1u);
    if constexpr (EF::ON) q4.Z = ef.get(code & 7, 0);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
#pragma unroll 1
    for (int i = 61; i >= 0; i--) {
        code = win_code_from_window(win_window(w, i));
        const TP* entry = tbl + (code & 7) * stride;
        const u32 neg = (code >> 3) - 1u;
        if constexpr (CH == 3) {
            static_assert(PRELOAD, "the asm bodies take the entry in registers");
            EntryRegs t = load_entry<L>(entry, neg, code & 7, ef);
#pragma unroll 1
            for (int k = 0; k < 3; k++) dbl_asm(Q.X, Q.Y, Q.Z);
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, t, neg);
        } else if (PRELOAD) {
            EntryRegs t = load_entry<L>(entry, neg, code & 7, ef);
#pragma unroll 1
            for (int k = 0; k < 4; k++) Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_entry<CH>(Q, t, neg);
        } else {
            u32 landing = 0;
            if (TOUCH) landing = touch_line(entry);
#pragma unroll 1
            for (int k = 0; k < 4; k++) Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_table<CH, L>(Q, entry, neg);
            if (TOUCH) touch_done(landing, Q.X.re.l[0]);
        }
    }
    return ladder_result<CH>(Q);
}

// The ladders with constant-time selection (curve.hip.h, "constant-time selection"): `src` reads the whole table at
// every step.  Same digits, same DAG, same R1 tuple as the ladders above.
// the four coordinates of entry `digit` out of a scanned table, as they are stored (add_asm applies the sign: masked selects, no address)
template <typename SRC> FQ_DEV EntryRegs scan_entry(const SRC& src, u32 digit_value) {
    const typename SRC::Bits digit(digit_value);
    EntryRegs t;
    t.N = src.coord(digit, 0); t.D = src.coord(digit, 1); t.E = src.coord(digit, 2); t.F = src.coord(digit, 3);
    return t;
}
template <int CH, typename SRC> FQ_DEV R1 ladder_endo_scan(const EndoDigits& e, const SRC& src) {
    Proj<1, 1, 1> q4 = start_scan(src, e.top & 7, 0u);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        if constexpr (CH == 3) {                  // the asm bodies: the scan's select trees hand them the entry as operands
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, scan_entry(src, endo_digit(e, i)), endo_neg_mask(e, i));
        } else {
            Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_scan<CH>(Q, src, endo_digit(e, i), endo_neg_mask(e, i));
        }
    }
    return ladder_result<CH>(Q);
}
template <int CH, typename SRC> FQ_DEV R1 ladder_windowed_scan(const WinScalar& w, const SRC& src) {
    u32 code = win_top_code(w);
    Proj<1, 1, 1> q4 = start_scan(src, code & 7, (code >> 3) - 1u);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
#pragma unroll 1
    for (int i = 61; i >= 0; i--) {
        code = win_code_from_window(win_window(w, i));
        if constexpr (CH == 3) {
#pragma unroll 1
            for (int k = 0; k < 3; k++) dbl_asm(Q.X, Q.Y, Q.Z);
            Fe2<1> T;
            dblt_asm(Q.X, Q.Y, Q.Z, T);
            add_asm(Q, T, scan_entry(src, code & 7), (code >> 3) - 1u);
        } else {
#pragma unroll 1
            for (int k = 0; k < 4; k++) Q = dbl<CH>(Q.X, Q.Y, Q.Z);
            Q = add_scan<CH>(Q, src, code & 7, (code >> 3) - 1u);
        }
    }
    return ladder_result<CH>(Q);
}

// table_endo for the fused kernels that keep E, F in LDS (LdsEF): the same DAG and the same order as above, but nothing is
// ever read back from the HBM slot.  The working points and the N, D of the entries that later additions need are parked in
// the LDS rows of entries that do not exist yet (each entry owns 80 bytes per lane):
//     tau(P)      entries 6, 7   step 0 -> read at the end of step 0
//     Q = phi(P)  entries 4, 5   step 0 -> read at the end of step 1
//     N, D of T[k], k = 1, 2, 3: entry 8 - k, written when T[k] is stored, read as the base of T[2 + k] / T[4 + k] before
//                 that entry's own E, F arrive (an iteration loads its next base BEFORE it stores the previous result)
// T[0]'s N, D stay in registers.  The HBM slot receives N, D of every entry, write-only; loads and stores share vmcnt on
// gfx950, so a table phase without loads has nothing to wait for until the ladder's first gather.
template <typename L, typename EF> FQ_DEV void build_table_endo_lds(const R1& P, u32* slot, const EF& ef) {
    static_assert(EF::ON, "needs the LDS copy of E and F");
    R2 result = r1_to_r2(P);                   // T[0]; `result` holds the one entry not stored yet
    int result_at = 0;
    const Fe2<1> n0 = result.N, d0 = result.D;
    Fe2<1> X = P.X, Y = P.Y, Z = P.Z;          // step 0: P; step 1: tau(P); step 2: phi(P)
    auto store_result = [&]() {
        store_entry<L>(slot, result_at, result, ef);
        if (result_at >= 1 && result_at <= 3) ef.park_nd(8 - result_at, result.N, result.D);
    };
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        store_result();
        Proj<1, 2, 1> t;
        if (step == 1) {                       // phi and psi share tau(P), curve4q.py:318-322
            t.X = X; t.Y = widen<2>(Y); t.Z = Z;
        } else {
            t = tau(X, Y, Z);
            if (step == 0) ef.park_xyz(6, t.X, fe2_carry(t.Y), t.Z);
        }
        Proj<2, 2, 2> u;
        if (step == 0) {
            u = upsilon(t);
        } else {
            Proj<1, 1, 1> c = chi(t);
            u.X = widen<2>(c.X); u.Y = widen<2>(c.Y); u.Z = widen<2>(c.Z);
        }
        R1 V = tau_dual(u.X, u.Y, u.Z);
        R3 V3 = r1_to_r3(V);
        R2 base;
        base.N = n0; base.D = d0; base.E = ef.get(0, 0); base.F = ef.get(0, 1);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 next = base;
            if (m + 1 < half) { ef.unpark_nd(8 - (m + 1), next.N, next.D); next.E = ef.get((u32)(m + 1), 0); next.F = ef.get((u32)(m + 1), 1); }
            if (m > 0) store_result();
            if (m == 0 && step == 0) ef.park_xyz(4, V.X, V.Y, V.Z);
            if (m == half - 1 && step < 2) ef.unpark_xyz(step == 0 ? 6 : 4, X, Y, Z);   // for the next step
            result = r1_to_r2(add_core(V3, as_signed(base)));
            result_at = half + m;
            base = next;
        }
    }
    store_result();                            // T[7]
}

// The same construction with every formula a generated body (ladder_asm.hip.h: tau / upsilon / chi / tau_dual + R1toR3 / R1toR2 /
// table addition, 22 500 instructions per table against hipcc's 24 400, no register shuffles between products): same parking
// schedule in LDS, same stores, same tight non-negative entries.  FQ_TABLE_ASM selects it for the fused kernels.
#ifndef FQ_TABLE_ASM
#define FQ_TABLE_ASM 1
#endif
template <typename L, typename EF> FQ_DEV void build_table_endo_lds_asm(const R1& P, u32* slot, const EF& ef) {
    static_assert(EF::ON, "needs the LDS copy of E and F");
    R2 result = r1_to_r2_asm(P);               // T[0]
    int result_at = 0;
    const Fe2<1> n0 = result.N, d0 = result.D;
    Fe2<1> X = P.X, Y = P.Y, Z = P.Z;          // step 0: P; step 1: tau(P); step 2: phi(P)
    auto store_result = [&]() {
        store_entry<L>(slot, result_at, result, ef);
        if (result_at >= 1 && result_at <= 3) ef.park_nd(8 - result_at, result.N, result.D);
    };
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        store_result();
        if (step != 1) {                       // phi and psi share tau(P), curve4q.py:318-322
            tau_asm(X, Y, Z);
            if (step == 0) ef.park_xyz(6, X, Y, Z);
        }
        if (step == 0) upsilon_asm(X, Y, Z); else chi_asm(X, Y, Z);
        Fe2<2> N3, D3;
        Fe2<1> F3;
        taudual_asm(X, Y, Z, N3, D3, F3);      // (X, Y, Z) = phi(P) / psi(P) / psi(phi(P)); (N3, D3, Z, F3) its R3 form
        const Fe2<1> E3 = Z;
        R2 base;
        base.N = n0; base.D = d0; base.E = ef.get(0, 0); base.F = ef.get(0, 1);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 next = base;
            if (m + 1 < half) { ef.unpark_nd(8 - (m + 1), next.N, next.D); next.E = ef.get((u32)(m + 1), 0); next.F = ef.get((u32)(m + 1), 1); }
            if (m > 0) store_result();
            if (m == 0 && step == 0) ef.park_xyz(4, X, Y, Z);
            if (m == half - 1 && step < 2) ef.unpark_xyz(step == 0 ? 6 : 4, X, Y, Z);   // for the next step
            result = base;
            table_add_asm(result, N3, D3, E3, F3);
            result_at = half + m;
            base = next;
        }
    }
    store_result();                            // T[7]
}

FQ_DEV void load_scalar(const u64* p, u64 m[4]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    m[0] = (u64)a.x | ((u64)a.y << 32); m[1] = (u64)a.z | ((u64)a.w << 32);
    m[2] = (u64)b.x | ((u64)b.y << 32); m[3] = (u64)b.z | ((u64)b.w << 32);
}

namespace {   // kernels: one private copy per translation unit (their code objects differ by FQ_CHAIN)

// Large variable-base batches, first half: per element, (DH: membership test, cofactor clearing,) table
// construction into scratch slot `pos`.  Kept apart from the ladder so that the endomorphisms' register
// appetite (256 VGPRs) does not set the ladder's occupancy.
template <int ALGO, bool DH, typename SL = PrebuiltSlots>
__global__ __launch_bounds__(BLOCK) void prep_kernel(LadderArgs a) {
    const u32 pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= (a.n_dev ? *a.n_dev : a.n)) return;
    const u32 first = a.base + (a.base_dev ? *a.base_dev : 0u);
    const u32 id = a.index ? a.index[first + pos] : first + pos;
    R1 P;
    if (DH) {
        Fe2<1> x = load_fe2(a.points + 8 * (size_t)id), y = load_fe2(a.points + 8 * (size_t)id + 4);
        a.status[id] = point_on_curve(x, y) ? FOURQ_DH_OK : FOURQ_DH_NOT_ON_CURVE;
        P = clear_cofactor_392(x, y);
    } else {
        P = load_r1(a.points + 20 * (size_t)id);
    }
    u32* slot = a.scratch + (size_t)pos * SL::SLOT;
    if (ALGO == ENDO) build_table_endo_pipelined<SL>(P, slot);
    else build_table_windowed<SL>(P, slot);
}

// ALGO: ENDO / WINDOWED.  SRC: where the table is.  DH: affine in, cofactor clearing, affine out + status.
// DEFER (DH only): leave (X, Y, Z) in a.proj for normalize_kernel instead of inverting Z here.
// CT: constant-time table selection (every entry read at every step); FUSED keeps N, D of the lane's table in registers and
// scans E, F in the lane's LDS rows, LDS scans the shared table where it lies (PREBUILT is not taken in this mode).
constexpr int ladder_waves(int src, bool dh, bool ct) {       // wave slots per SIMD a ladder kernel is built for
    if (src == FUSED) return 1;                                // the endomorphisms' 256 VGPRs and the CU's whole LDS
    if (!ct && FQ_LADDER_ASM && ((src == LDS && FQ_LDS_ASM) || (src == PREBUILT && FQ_PREBUILT_ASM && !dh))) return 2;   // the asm bodies: 256 VGPRs
    if (ct && src == LDS && FQ_LADDER_ASM && FQ_CT_LDS_ASM) return 2;
    if (ct && src == LDS && dh) return 3;                      // the scan's select tree + DH's epilogue spill at 128 VGPRs (52-80 bytes per lane)
    return 4;
}
template <int ALGO, int SRC, bool DH, bool DEFER = false, bool CT = false>
__global__ __launch_bounds__(BLOCK, ladder_waves(SRC, DH, CT)) void ladder_kernel(LadderArgs a) {
    static_assert(!DEFER || DH, "only DH outputs are normalised");
    static_assert(!(CT && SRC == PREBUILT), "the constant-time mode does not take the two-kernel route");
    constexpr bool USE_EF = SRC == FUSED;                      // LdsEF: the CU's whole LDS for one block of a fused kernel
    __shared__ __attribute__((aligned(16))) u32 lds_table[SRC == LDS ? 8 * LDS_ENTRY_U32 : (USE_EF ? EF_LDS_U32 : 4)];
    using EF = typename std::conditional<USE_EF, LdsEF, NoEF>::type;
    using L = typename std::conditional<SRC == PREBUILT, PrebuiltSlots, NDSlots>::type;      // slot layout (unused with SRC == LDS)
    EF ef;
    if constexpr (USE_EF) ef.lane = reinterpret_cast<uint2*>(lds_table) + threadIdx.x;
    if (SRC == LDS) {
        for (int i = threadIdx.x; i < 8 * R2_LIMBS; i += BLOCK)
            lds_table[(i / R2_LIMBS) * LDS_ENTRY_U32 + (i % R2_LIMBS)] = a.table[i];
        __syncthreads();
    }
    const u32 n = a.n_dev ? *a.n_dev : a.n;
    const u32 lane_slot = blockIdx.x * BLOCK + threadIdx.x;
    const u32 lanes = gridDim.x * BLOCK;
    const u32 n_round = (n + BLOCK - 1) / BLOCK * BLOCK;
#pragma unroll 1
    for (u32 it = lane_slot; it < n_round; it += lanes) {
        const bool live = it < n;
        const u32 pos = live ? it : n - 1;                // idle tail lanes redo the last element, store nothing
        const u32 id = a.index ? a.index[a.base + pos] : a.base + pos;
        u64 m[4];
        load_scalar(a.scalars + 4 * (size_t)id, m);
        u32* slot = SRC == LDS ? nullptr : a.scratch + (size_t)(SRC == FUSED ? lane_slot : pos) * L::SLOT;
        const u32* tbl = slot;
        if (SRC == PREBUILT && a.slot_of) {                              // mixed batch: own table or the shared one
            const u32 own = a.slot_of[pos];
            tbl = own == ~0u ? a.table_slots : a.scratch + (size_t)own * L::SLOT;
        }

        uint8_t st = FOURQ_DH_OK;
        if (SRC == PREBUILT) {
            if (DH) st = a.status[id];                                  // membership verdict of prep_kernel
        } else {
            R1 P;
            if (DH) {
                Fe2<1> x = load_fe2(a.points + 8 * (size_t)id), y = load_fe2(a.points + 8 * (size_t)id + 4);
                if (!point_on_curve(x, y)) st = FOURQ_DH_NOT_ON_CURVE;  // keep going branch-free; masked at the end
                if (SRC == FUSED) P = clear_cofactor_392(x, y);         // with a table the reference discards [392]P (curve4q.py:209)
            } else if (SRC == FUSED) {
                if (a.io & LADDER_IO_AFFINE_IN) P = affine_to_r1(load_fe2(a.points + 8 * (size_t)id), load_fe2(a.points + 8 * (size_t)id + 4));
                else P = load_r1(a.points + 20 * (size_t)id);
            }
            if constexpr (SRC == FUSED) {
                if constexpr (ALGO == ENDO && FQ_TABLE_ASM && FQ_LADDER_ASM) build_table_endo_lds_asm<L>(P, slot, ef);
                else if (ALGO == ENDO) build_table_endo_lds<L>(P, slot, ef);
                else build_table_windowed<L>(P, slot, ef);
            }
        }
        R1 Q;
        constexpr bool CT_ASM = CT && FQ_LADDER_ASM && ((SRC == LDS && FQ_CT_LDS_ASM) || (SRC == FUSED && FQ_CT_FUSED_ASM));
        constexpr int CH = (((SRC == FUSED || (SRC == LDS && FQ_LDS_ASM) || (SRC == PREBUILT && FQ_PREBUILT_ASM && !DH)) && !CT && FQ_LADDER_ASM) || CT_ASM) ? 3 : LADDER_CH;
        if constexpr (ALGO == ENDO && SRC == FUSED && !CT && CH == 3 && FQ_NIBBLE_LADDER) {
            u64 v[4];
            decompose(m, v);
            Q = ladder_endo_nibbles<L, EF, !DH>(recode_nibbles(v), a.scratch, lane_slot * (u32)L::SLOT, ef);      // MUL_*: signed limbs, see the store
        } else if constexpr (ALGO == ENDO && SRC == LDS && !CT && CH == 3 && FQ_NIBBLE_LADDER) {
            u64 v[4];
            decompose(m, v);
            Q = ladder_endo_nibbles_at<LimbSlots, NoEF>(recode_nibbles(v), lds_table, LDS_ENTRY_U32);
        } else if (ALGO == ENDO) {
            u64 v[4];
            decompose(m, v);
            const EndoDigits e = recode(v);
            if constexpr (CT && SRC == FUSED) {
                ScanSplit<EF> regs;
                regs.ef = ef;
                regs.template load<L>(tbl);
                Q = ladder_endo_scan<CH>(e, regs);
            } else if constexpr (CT) {
                Q = ladder_endo_scan<CH>(e, ScanMem<8, u32>{ lds_table, LDS_ENTRY_U32 });
            } else if constexpr (SRC == LDS) {
                Q = ladder_endo<CH, CH == 3>(e, lds_table, LDS_ENTRY_U32);
            } else {
                Q = ladder_endo<CH, SRC == FUSED || CH == 3, L, EF, SRC == PREBUILT && CH != 3>(e, tbl, L::ENTRY, ef);
            }
        } else {
            const WinScalar w = win_reduce(m);
            if constexpr (CT && SRC == FUSED) {
                ScanSplit<EF> regs;
                regs.ef = ef;
                regs.template load<L>(tbl);
                Q = ladder_windowed_scan<CH>(w, regs);
            } else if constexpr (CT) {
                Q = ladder_windowed_scan<CH>(w, ScanMem<8, u32>{ lds_table, LDS_ENTRY_U32 });
            } else if constexpr (SRC == LDS) {
                Q = ladder_windowed<CH, CH == 3>(w, lds_table, LDS_ENTRY_U32);
            } else {
                Q = ladder_windowed<CH, SRC == FUSED || CH == 3, L, EF, SRC == PREBUILT && CH != 3>(w, tbl, L::ENTRY, ef);
            }
        }
        if (DH && (DEFER || (FQ_FUSED_DEFER && SRC == FUSED && a.proj != nullptr))) {     // one inversion per K elements, later (FUSED: by a wave-uniform flag, no second instance)
            if (live) {
                store_proj(a.proj, a.proj_stride, id, Q.X, Q.Y, Q.Z);
                a.status[id] = st;
            }
        } else if (DH) {
            Fe2<1> ax, ay;
            r1_to_affine(Q, ax, ay);
            u64 o[8];
            store_fe2(o, ax); store_fe2(o + 4, ay);
            bool neutral = (o[0] | o[1] | o[2] | o[3] | o[5] | o[6] | o[7]) == 0 && o[4] == 1;   // (Ox, Oy) = ((0,0),(1,0))
            if (st == FOURQ_DH_OK && neutral) st = FOURQ_DH_NEUTRAL;
            if (live) {
                uint4* dst = reinterpret_cast<uint4*>(a.out + 8 * (size_t)id);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u64 lo = st ? 0 : o[2 * k], hi = st ? 0 : o[2 * k + 1];
                    dst[k] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
                }
                a.status[id] = st;
            }
        } else if (live) {
            u64 o[20];
            if constexpr (ALGO == ENDO && SRC == FUSED && !CT && CH == 3 && FQ_NIBBLE_LADDER) store_r1_signed(o, Q);
            else store_r1(o, Q);
            const bool xyz = SRC == FUSED && (a.io & LADDER_IO_XYZ_OUT);
            uint4* dst = reinterpret_cast<uint4*>(a.out + (xyz ? 12 : 20) * (size_t)id);
#pragma unroll
            for (int k = 0; k < 10; k++)
                if (k < 6 || !xyz) dst[k] = make_uint4((u32)o[2 * k], (u32)(o[2 * k] >> 32), (u32)o[2 * k + 1], (u32)(o[2 * k + 1] >> 32));
        }
    }
}

// ---- two lanes per element (pair.hip.h): the variable-base kernels for small batches and for the tail past whole generations --
// 128 elements per 256-lane block (the pair's table fills the CU's LDS), one block per CU, grid-stride over the batch.  Each lane
// loads, computes and stores ITS half (real or imaginary parts) of every coordinate; the scalar is recoded in both lanes.
// CT: constant-time table selection -- the lane scans its eight entries in LDS at every step (PairTable::scan_entry).
// FIXED: the caller's table (a.table, working limbs) instead of one built from a point: every lane copies its halves of the
// eight entries into its LDS rows once, then the same ladders run (with a table the reference ignores the point: curve4q.py:209,
// :426; DH still tests it, curve4q.py:447-448).
// LPE = 4: four lanes per element (pair.hip.h, "four lanes per element"): 64 elements per block, both pairs of an element keep the
// whole table and every value; the ladder steps share their products between the pairs; the first pair stores.
// MIXED (BASELINE config 5's batch shape, small): element i takes its table from its own point (flags[i] != 0) or from the caller's
// fixed-base table -- every lane builds a table (from zeros where the element is fixed-base: no divergence in the long part), then the
// fixed-base elements overwrite their LDS rows with the caller's entries.  Which kind an element is, is public.
template <int ALGO, bool DH, bool CT = false, bool FIXED = false, int LPE = 2, bool MIXED = false>
__global__ __launch_bounds__(BLOCK, 1) void pair_kernel(LadderArgs a) {
    static_assert(LPE == 2 || LPE == 4, "two or four lanes per element");
    static_assert(!MIXED || (ALGO == ENDO && !DH && !FIXED), "mixed batches are MUL_endo batches");
    constexpr bool QUAD = LPE == 4;
    const QuadLane ql{ QUAD && (threadIdx.x & 2) != 0 };
    const bool writer = !ql.second;
    __shared__ __attribute__((aligned(16))) u32 lds_pair[PAIR_LDS_U32];
    PairTable tbl;
    tbl.two = reinterpret_cast<uint2*>(lds_pair) + threadIdx.x;
    tbl.one = lds_pair + PAIR_VALUES * 2 * 256 * 2 + threadIdx.x;             // behind the region of limb pairs
    const u32 odd = threadIdx.x & 1;
    const PairLane pl{ odd - 1u, 0u - odd };
    constexpr u32 PER_BLOCK = BLOCK / LPE;
    constexpr int IN_WORDS = DH ? 8 : 20, OUT_WORDS = DH ? 8 : 20;
    const u32 n = a.n, n_round = (n + PER_BLOCK - 1) / PER_BLOCK * PER_BLOCK;
    if constexpr (FIXED) {                       // entry k, coordinate c: 12 dwords = re limbs 0..4 | im limbs 0..4 | 2 pad (load_fe2_limbs)
#pragma unroll 1
        for (int v = 0; v < PAIR_VALUES; v++) {
            const u32* src = a.table + (v >> 2) * R2_LIMBS + (v & 3) * COORD_U32 + 5 * odd;
            PF<1> h;
#pragma unroll
            for (int i = 0; i < 5; i++) { h.l[i] = src[i]; FQ_SIGN_UNKNOWN(h.l[i]); }
            tbl.put(v, h);
        }
    }
#pragma unroll 1
    for (u32 it = blockIdx.x * PER_BLOCK + threadIdx.x / LPE; it < n_round; it += gridDim.x * PER_BLOCK) {
        const bool live = it < n && writer;
        const u32 id = a.base + (it < n ? it : n - 1);           // idle tail pairs redo the last element, store nothing
        u64 m[4];
        load_scalar(a.scalars + 4 * (size_t)id, m);
        bool variable = true;
        if constexpr (MIXED) variable = a.flags[id] != 0;
        auto half = [&](int c) {                                  // this lane's half of coordinate c of the input point: words 4c + 2 odd, + 1
            const uint4 w = *reinterpret_cast<const uint4*>(a.points + IN_WORDS * (size_t)id + 4 * c + 2 * odd);
            const Fe<1> f = fe_unpack((u64)w.x | ((u64)w.y << 32), (u64)w.z | ((u64)w.w << 32));
            PF<1> r;
#pragma unroll
            for (int i = 0; i < 5; i++) { r.l[i] = (!MIXED || variable) ? f.l[i] : 0u; FQ_SIGN_UNKNOWN(r.l[i]); }
            return r;
        };
        auto store_half = [&](int c, u64 lo, u64 hi) {
            *reinterpret_cast<uint4*>(a.out + OUT_WORDS * (size_t)id + 4 * c + 2 * odd) = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
        };
        uint8_t st = FOURQ_DH_OK;
        PR1 P;
        if constexpr (DH) {
            const PF<1> x = half(0), y = half(1);
            if (!pair_point_on_curve(x, y, pl)) st = FOURQ_DH_NOT_ON_CURVE;      // keep going branch-free; masked at the end
            if constexpr (!FIXED) P = pair_clear_cofactor_392<QUAD>(x, y, pl, ql);
        } else if constexpr (!FIXED) {
            P.X = half(0); P.Y = half(1); P.Z = half(2); P.Ta = pwiden<3>(half(3)); P.Tb = pwiden<2>(half(4));
        }
        PR1 Q;
        if constexpr (ALGO == ENDO) {
            if constexpr (!FIXED) pair_build_table_endo<QUAD>(P, tbl, pl, ql);
            if constexpr (MIXED) {
                if (!variable) {
#pragma unroll 1
                    for (int v = 0; v < PAIR_VALUES; v++) {
                        const u32* src = a.table + (v >> 2) * R2_LIMBS + (v & 3) * COORD_U32 + 5 * odd;
                        PF<1> h;
#pragma unroll
                        for (int i = 0; i < 5; i++) { h.l[i] = src[i]; FQ_SIGN_UNKNOWN(h.l[i]); }
                        tbl.put(v, h);
                    }
                }
            }
            u64 v[4];
            decompose(m, v);
            Q = pair_ladder_endo<CT, QUAD>(recode(v), tbl, pl, ql);
        } else {
            if constexpr (!FIXED) pair_build_table_windowed<QUAD>(P, tbl, pl, ql);
            Q = pair_ladder_windowed<CT, QUAD>(win_reduce(m), tbl, pl, ql);
        }
        if constexpr (DH) {
            PF<1> ax, ay;
            pair_to_affine(Q, pl, ax, ay);
            u64 x0, x1, y0, y1;
            pair_canon(ax, x0, x1); pair_canon(ay, y0, y1);
            // (Ox, Oy) = ((0, 0), (1, 0)): this lane's half of it is (0, 1) on even lanes and (0, 0) on odd ones
            const u32 neutral = pair_both(((x0 | x1 | y1) == 0 && y0 == (u64)(pl.even & 1u)) ? 1u : 0u);
            if (st == FOURQ_DH_OK && neutral) st = FOURQ_DH_NEUTRAL;
            if (live) {
                store_half(0, st ? 0 : x0, st ? 0 : x1);
                store_half(1, st ? 0 : y0, st ? 0 : y1);
                if (!odd) a.status[id] = st;
            }
        } else if (live) {
            u64 lo, hi;
            pair_canon(Q.X, lo, hi); store_half(0, lo, hi);
            pair_canon(Q.Y, lo, hi); store_half(1, lo, hi);
            pair_canon(Q.Z, lo, hi); store_half(2, lo, hi);
            pair_canon(Q.Ta, lo, hi); store_half(3, lo, hi);
            pair_canon(Q.Tb, lo, hi); store_half(4, lo, hi);
        }
    }
}

// ---- mixed batches: ONE persistent kernel pulling work from a device-side queue (BASELINE config 5) ----------------
// partition_kernel has compacted the ids of the round's variable-base and fixed-base elements (counts[0], counts[1]).  A work
// item is 64 consecutive positions of the two lists laid end to end, variable-base first: items wholly inside the first list are
// variable-base (fused table_endo + ladder, as ladder_kernel<ENDO, FUSED>), items wholly inside the second fixed-base (the shared
// table, gathered through L1 a doubling ahead), and the ONE item that straddles the boundary runs as a variable-base item whose
// fixed-base lanes copy the shared table into their own slot and LDS rows instead of building one -- so 65 550 + 65 522 elements
// are 2 048 items, two per resident wave, not 1 025 + 1 024 with a third item (a whole ladder) for one wave (round 3's version:
// profiles/r03_mixed_queue.txt).  One block per CU owns the CU's LDS (E, F of the lanes' tables) and a scratch slot per lane; each
// of its four waves loops on its own: take the next item with one atomic, run it, until the queue is empty.  Except in the
// boundary item no wave diverges, and the long items are handed out first.  CT: constant-time table selection for both kinds
// (the kind of an element is public, its digits are not).  Every wave leaves the loop: the queue head only grows and `total` is
// fixed before the launch.
template <bool CT>
__global__ __launch_bounds__(BLOCK, 1) void mixed_queue_kernel(LadderArgs a, const u32* var_list, const u32* fix_list, const u32* counts, u32* queue_head) {
    __shared__ __attribute__((aligned(16))) u32 lds_ef[EF_LDS_U32];
    LdsEF ef;
    ef.lane = reinterpret_cast<uint2*>(lds_ef) + threadIdx.x;
    const u32 n_var = counts[0], n_all = n_var + counts[1];
    const u32 total = (n_all + 63) / 64;
    u32* slot = a.scratch + (size_t)(blockIdx.x * BLOCK + threadIdx.x) * NDSlots::SLOT;
    const u32 lane = threadIdx.x & 63;
    constexpr int CH = (FQ_LADDER_ASM && (!CT || FQ_CT_FUSED_ASM)) ? 3 : LADDER_CH;
#pragma unroll 1
    for (;;) {
        u32 item = 0;
        if (lane == 0) item = atomicAdd(queue_head, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= total) break;
        const u32 first = item * 64;
        const bool all_fixed = first >= n_var;                      // wave-uniform
        const bool live = first + lane < n_all;
        const u32 pos = live ? first + lane : n_all - 1;            // idle tail lanes redo the last element, store nothing
        const bool variable = pos < n_var;                          // per lane: differs inside the wave in the boundary item only
        const u32 id = variable ? var_list[pos] : fix_list[pos - n_var];
        u64 m[4];
        load_scalar(a.scalars + 4 * (size_t)id, m);
        R1 Q;
        if (!all_fixed) {
            if (variable) {
                const R1 P = load_r1(a.points + 20 * (size_t)id);
                if constexpr (FQ_TABLE_ASM && FQ_LADDER_ASM) build_table_endo_lds_asm<NDSlots>(P, slot, ef);
                else build_table_endo_lds<NDSlots>(P, slot, ef);
            } else {                                                // a fixed-base lane of the boundary item: the shared table as its own
#pragma unroll 1
                for (int k = 0; k < 8; k++) store_entry<NDSlots>(slot, k, load_r2_limbs(a.table + k * R2_LIMBS), ef);
            }
            u64 v[4];
            decompose(m, v);
            if constexpr (CT) {
                const EndoDigits e = recode(v);
                ScanSplit<LdsEF> regs;
                regs.ef = ef;
                regs.template load<NDSlots>(slot);
                Q = ladder_endo_scan<CH>(e, regs);
            } else if constexpr (CH == 3 && FQ_NIBBLE_LADDER) {          // the headline kernel's ladder (nibble stream, 32-bit entry offsets)
                Q = ladder_endo_nibbles<NDSlots, LdsEF>(recode_nibbles(v), a.scratch, (blockIdx.x * BLOCK + threadIdx.x) * (u32)NDSlots::SLOT, ef);
            } else {
                Q = ladder_endo<CH, true, NDSlots, LdsEF>(recode(v), (const u32*)slot, NDSlots::ENTRY, ef);
            }
        } else {
            u64 v[4];
            decompose(m, v);
            if constexpr (CT) Q = ladder_endo_scan<CH>(recode(v), ScanMem<8, u32>{ a.table, R2_LIMBS });      // wave-uniform addresses
            else if constexpr (CH == 3 && FQ_NIBBLE_LADDER) Q = ladder_endo_nibbles_at<LimbSlots, NoEF>(recode_nibbles(v), a.table, R2_LIMBS);
            else Q = ladder_endo<CH, true, LimbSlots, NoEF>(recode(v), a.table, R2_LIMBS);
        }
        if (live) {
            u64 o[20];
            store_r1(o, Q);
            uint4* dst = reinterpret_cast<uint4*>(a.out + 20 * (size_t)id);
#pragma unroll
            for (int k = 0; k < 10; k++) dst[k] = make_uint4((u32)o[2 * k], (u32)(o[2 * k] >> 32), (u32)o[2 * k + 1], (u32)(o[2 * k + 1] >> 32));
        }
    }
}

// Mixed batches in CONSTANT-TIME mode, rounds larger than one generation.  The variable-base ids run through the fused constant-
// time kernel (table in registers + LDS, one wave per SIMD), whose generations hold exactly `lanes` elements: config 5's 65 550
// ids would cost a second generation of the whole chip for 14 elements.  split_counts_kernel therefore cuts a small remainder
// (at most `limit` ids) off the list on the device; prep_kernel<ENDO, false, LimbSlots> builds those ids' tables into scratch, and
// this kernel -- 128 registers, 1.6 KB of LDS, up to four waves per SIMD -- runs them TOGETHER with the round's fixed-base
// elements: a wave whose 64 elements are all fixed-base scans the shared table in LDS (as ladder_kernel<ENDO, LDS, CT>), any other
// wave scans each lane's own table through a per-lane pointer in global memory (every entry read at every step: no address depends
// on a digit).  Which kind an element is, is public.
template <int UNIT_ONLY = 0>        // a template so that only the translation unit that launches it carries a copy
__global__ void split_counts_kernel(u32* counts, u32 lanes, u32 limit) {         // counts: [n_var, n_fix, -, -, fused, over]
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const u32 n_var = counts[0], rem = n_var % lanes, whole = n_var - rem;
    const bool cut = whole > 0 && rem > 0 && rem <= limit;
    counts[4] = cut ? whole : n_var;
    counts[5] = cut ? rem : 0u;
}
template <int UNIT_ONLY = 0>
__global__ __launch_bounds__(BLOCK, (FQ_LADDER_ASM && FQ_CT_LDS_ASM) ? 2 : 4) void mixed_ct_tail_kernel(LadderArgs a, const u32* fix_list, const u32* var_list, const u32* counts, const u32* over_scratch) {
    __shared__ __attribute__((aligned(16))) u32 lds_table[8 * LDS_ENTRY_U32];
    for (int i = threadIdx.x; i < 8 * R2_LIMBS; i += BLOCK)
        lds_table[(i / R2_LIMBS) * LDS_ENTRY_U32 + (i % R2_LIMBS)] = a.table[i];
    __syncthreads();
    const u32 n_fix = counts[1], fused = counts[4], total = n_fix + counts[5];
    const u32 lanes = gridDim.x * BLOCK, n_round = (total + BLOCK - 1) / BLOCK * BLOCK;
    constexpr int CH = (FQ_LADDER_ASM && FQ_CT_LDS_ASM) ? 3 : LADDER_CH;
#pragma unroll 1
    for (u32 it = blockIdx.x * BLOCK + threadIdx.x; it < n_round; it += lanes) {
        const bool live = it < total;
        const u32 pos = live ? it : total - 1;                // idle tail lanes redo the last element, store nothing
        const bool fixed = pos < n_fix;
        const u32 id = fixed ? fix_list[pos] : var_list[fused + (pos - n_fix)];
        u64 m[4], v[4];
        load_scalar(a.scalars + 4 * (size_t)id, m);
        decompose(m, v);
        const EndoDigits e = recode(v);
        R1 Q;
        if (__all(fixed)) {                                   // wave-uniform, and public
            Q = ladder_endo_scan<CH>(e, ScanMem<8, u32>{ lds_table, LDS_ENTRY_U32 });
        } else {
            const u32* tbl = fixed ? a.table : over_scratch + (size_t)(pos - n_fix) * LimbSlots::SLOT;
            Q = ladder_endo_scan<CH>(e, ScanMem<8, u32>{ tbl, LimbSlots::ENTRY });
        }
        if (live) {
            u64 o[20];
            store_r1(o, Q);
            uint4* dst = reinterpret_cast<uint4*>(a.out + 20 * (size_t)id);
#pragma unroll
            for (int k = 0; k < 10; k++) dst[k] = make_uint4((u32)o[2 * k], (u32)(o[2 * k] >> 32), (u32)o[2 * k + 1], (u32)(o[2 * k + 1] >> 32));
        }
    }
}

// ---- fixed-base comb -------------------------------------------------------------
constexpr int COMB_POINTS = COMB_POINTS_ALL;                  // 1 024 (by address) + 80 (constant-time mode)
static_assert(COMB_POINTS == FOURQ_COMB_POINTS, "include/fourq_amd.h and recode.hip.h disagree on the comb's shape");
constexpr int COMB_ENTRY_U32 = 3 * COORD_U32;                 // (x+y, y-x, 2dxy)
constexpr int COMB_LDS_U32 = COMB_ENTRY_U32;                  // stride in LDS (pads of 0, 4, 8, 12 dwords measured alike)
constexpr int COMB_BLOCK_MAX = 1024;                          // the fast shape's table is 144 KB of LDS: ONE block per CU, up to 16 waves wide
constexpr size_t COMB_FAST_LDS_BYTES = (size_t)CombFast::POINTS * COMB_LDS_U32 * sizeof(u32);
static_assert(COMB_FAST_LDS_BYTES <= 160 * 1024, "the fast comb table must fit the CU's LDS");

// ---- key generation for small batches: the comb (recode.hip.h, CombFast: 6 doublings + 27 mixed additions) four lanes per element ----
// comb_kernel stages the 144 KB table into LDS once per block -- nothing when a block works through thousands of elements, a third of
// the 0.11 ms of a batch of one.  Here every lane gathers ITS HALF of an entry (15 dwords) from the table where it lies (L2), one
// column ahead of its use, and the element's two pairs share the products of the doubling and of the mixed addition (pair.hip.h):
// 64 elements per block, no LDS in the default mode.
// CT: the constant-time shape (80 points, 16-entry blocks; recode.hip.h, CombScan), its 11.5 KB staged into LDS; an addition reads its WHOLE
// block -- every lane the same addresses but for its half -- and keeps one entry by the select tree (curve.hip.h): no address depends on the scalar.
template <bool CT, int P, typename TREE, typename BITS> FQ_DEV void comb_scan_pairs(const u32* blk, u32 odd, const BITS& bits, TREE& tree, u32 out[15]) {
    u32 v0[15], v1[15];
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            v0[5 * cidx + i] = blk[(2 * P) * COMB_ENTRY_U32 + cidx * COORD_U32 + 5 * odd + i];
            v1[5 * cidx + i] = blk[(2 * P + 1) * COMB_ENTRY_U32 + cidx * COORD_U32 + 5 * odd + i];
        }
    }
    tree.template feed<P>(bits, v0, v1, out);
    __builtin_amdgcn_sched_barrier(0);                         // two entries per round, as ScanMem
    if constexpr (2 * P + 2 < CombScan::BLOCK_POINTS) comb_scan_pairs<CT, P + 1>(blk, odd, bits, tree, out);
}
// LPE = 2: two lanes per element (batches between a quarter and half a generation): the same walk on the pair code, T = Ta*Tb inside the addition.
template <bool CT = false, int LPE = 4>
__global__ __launch_bounds__(BLOCK) void comb_quad_kernel(const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, u32 n) {
    static_assert(LPE == 2 || LPE == 4, "two or four lanes per element");
    using S = typename std::conditional<CT, CombScan, CombFast>::type;
    __shared__ __attribute__((aligned(16))) u32 lds_scan[CT ? CombScan::POINTS * COMB_ENTRY_U32 : 4];
    if constexpr (CT) {
        const u32* sub = comb_limbs + CombFast::POINTS * COMB_ENTRY_U32;
        for (u32 i = threadIdx.x; i < (u32)(CombScan::POINTS * COMB_ENTRY_U32 / 4); i += BLOCK)
            reinterpret_cast<uint4*>(lds_scan)[i] = reinterpret_cast<const uint4*>(sub)[i];
        __syncthreads();
    }
    const u32 odd = threadIdx.x & 1;
    const PairLane pl{ odd - 1u, 0u - odd };
    const QuadLane ql{ LPE == 4 && (threadIdx.x & 2) != 0 };
    const u32 it = blockIdx.x * (BLOCK / LPE) + threadIdx.x / LPE;
    const bool live = it < n && !ql.second;
    const u32 id = it < n ? it : n - 1;                       // idle tail quads redo the last element, store nothing
    u64 m[4];
    load_scalar(scalars + 4 * (size_t)id, m);
    const CombDigits<S> c = comb_recode<S>(m);
    auto column = [&](int t) { return S::E * (t % S::V) + (S::E - 1 - t / S::V); };          // the t-th column in processing order
    auto entry = [&](int t, PF<1>& hN, PF<1>& hD, PF<1>& hF) {  // this lane's halves of the three coordinates of the entry of the t-th column
        const int col = column(t);
        u32 w[15];
        if constexpr (CT) {
            const DigitBits<S::W - 1> bits(comb_index(c, col));
            SelectTree<15, S::W - 1> tree;
            comb_scan_pairs<CT, 0>(lds_scan + (((u32)(t % S::V)) << (S::W - 1)) * COMB_ENTRY_U32, odd, bits, tree, w);
        } else {
            const u32* e = comb_limbs + (size_t)((((u32)(t % S::V)) << (S::W - 1)) + comb_index(c, col)) * COMB_ENTRY_U32 + 5 * odd;
#pragma unroll
            for (int cidx = 0; cidx < 3; cidx++) {
#pragma unroll
                for (int i = 0; i < 5; i++) w[5 * cidx + i] = e[cidx * COORD_U32 + i];
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            hN.l[i] = w[i]; hD.l[i] = w[5 + i]; hF.l[i] = w[10 + i];
            FQ_SIGN_UNKNOWN(hN.l[i]); FQ_SIGN_UNKNOWN(hD.l[i]); FQ_SIGN_UNKNOWN(hF.l[i]);
        }
    };
    PF<1> aN, aD, aF;
    entry(0, aN, aD, aF);
    PR1 Q;
    PF<1> T;
    {                                                           // +-A as a point with Z = 2: (N - D, N + D, 2), T = X*Y/Z
        const u32 neg = comb_neg_mask(c, column(0));
        PF<1> N, D;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            N.l[k] = __builtin_amdgcn_bitop3_b32(neg, aD.l[k], aN.l[k], 0xCA);
            D.l[k] = __builtin_amdgcn_bitop3_b32(neg, aN.l[k], aD.l[k], 0xCA);
        }
        Q.X = ptighten(psub(N, D)); Q.Y = ptighten(padd(N, D));
        Q.Z.l[0] = pl.even & 2u; Q.Z.l[1] = Q.Z.l[2] = Q.Z.l[3] = Q.Z.l[4] = 0;
        const PF<1> y_half = pmul_const(Q.Y, fe2_half_const(), pl);
        Q.Ta = pwiden<3>(Q.X); Q.Tb = pwiden<2>(y_half);        // Ta*Tb = X*Y/Z
        T = pmul(Q.X, y_half, pl);
    }
#pragma unroll 1
    for (int t = 1; t < S::V * S::E; t++) {
        entry(t, aN, aD, aF);
        if constexpr (LPE == 4) {
            if (t % S::V == 0) Q = qdbl_point<true>(Q.X, Q.Y, Q.Z, pl, ql, T);
            Q = qadd_affine_entry(Q, T, aN, aD, aF, comb_neg_mask(c, column(t)), pl, ql);
        } else {
            if (t % S::V == 0) Q = pdbl_point(Q.X, Q.Y, Q.Z, pl);
            Q = padd_affine_entry(Q, aN, aD, aF, comb_neg_mask(c, column(t)), pl);
        }
    }
    PF<1> ax, ay;
    pair_to_affine(Q, pl, ax, ay);
    ax = pcneg(ax, c.negate);                                   // even scalar: [k]B = -[N - k]B, -(x, y) = (-x, y)
    u64 x0, x1, y0, y1;
    pair_canon(ax, x0, x1); pair_canon(ay, y0, y1);
    const u32 neutral = pair_both(((x0 | x1 | y1) == 0 && y0 == (u64)(pl.even & 1u)) ? 1u : 0u);
    if (live) {
        auto store_half = [&](int k, u64 lo, u64 hi) {
            *reinterpret_cast<uint4*>(out + 8 * (size_t)id + 4 * k + 2 * odd) = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
        };
        store_half(0, neutral ? 0 : x0, neutral ? 0 : x1);
        store_half(1, neutral ? 0 : y0, neutral ? 0 : y1);
        if (!odd) status[id] = neutral ? FOURQ_DH_NEUTRAL : FOURQ_DH_OK;
    }
}

#if FQ_CHAIN   // only fourq_chain.hip launches it
// [m]B, affine, from the comb: 6 doublings + 27 mixed additions per element (constant-time mode: 9 + 49 on the small shape).
// The block width is the launcher's choice (blockDim.x, a multiple of 64): the table is staged once per block.
constexpr int COMB_MODE = LADDER_CH;                          // the comb's additions run on signed limbs like every ladder
template <bool DEFER, bool CT = false>
__global__ __launch_bounds__(CT ? BLOCK : COMB_BLOCK_MAX, CT ? 4 : 1) void comb_kernel(const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, uint4* proj, u32 proj_stride, u32 n) {
    using S = typename std::conditional<CT, CombScan, CombFast>::type;       // the constant-time mode scans the small sub-table
    extern __shared__ __attribute__((aligned(16))) u32 lds[];                // S::POINTS * COMB_LDS_U32 dwords, sized by the launcher
    const u32* sub = comb_limbs + (CT ? CombFast::POINTS * COMB_ENTRY_U32 : 0);
    const u32 width = blockDim.x;
    if (COMB_LDS_U32 == COMB_ENTRY_U32) {
        for (u32 i = threadIdx.x; i < (u32)(S::POINTS * COMB_ENTRY_U32 / 4); i += width)
            reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(sub)[i];
    } else {
        for (u32 i = threadIdx.x; i < (u32)(S::POINTS * COMB_ENTRY_U32); i += width)
            lds[(i / COMB_ENTRY_U32) * COMB_LDS_U32 + (i % COMB_ENTRY_U32)] = sub[i];
    }
    __syncthreads();
    const u32 lanes = gridDim.x * width;
    const u32 n_round = (n + width - 1) / width * width;
#pragma unroll 1
    for (u32 it = blockIdx.x * width + threadIdx.x; it < n_round; it += lanes) {
        const bool live = it < n;
        const u32 id = live ? it : n - 1;
        u64 m[4];
        load_scalar(scalars + 4 * (size_t)id, m);
        CombDigits<S> c = comb_recode<S>(m);
        R1 Q;
#pragma unroll 1
        for (int i = S::E - 1; i >= 0; i--) {
#pragma unroll 1
            for (int j = 0; j < S::V; j++) {
                const int col = S::E * j + i;
                const u32 neg = comb_neg_mask(c, col);
                if constexpr (CT) {                      // all entries of block j are read; the index only forms masks
                    const ScanMem<S::BLOCK_POINTS, u32> block{ lds + (j << (S::W - 1)) * COMB_LDS_U32, COMB_LDS_U32 };
                    if (i == S::E - 1 && j == 0) { Q = affine_scan_start(block, comb_index(c, col), neg); continue; }
                    if (j == 0) Q = dbl<COMB_MODE>(Q.X, Q.Y, Q.Z);
                    Q = add_affine_scan<COMB_MODE>(Q, block, comb_index(c, col), neg);
                } else {
                    const u32* entry = lds + ((j << (S::W - 1)) + comb_index(c, col)) * COMB_LDS_U32;
                    if (i == S::E - 1 && j == 0) { Q = affine_table_start(entry, neg); continue; }
                    if (j == 0) Q = dbl<COMB_MODE>(Q.X, Q.Y, Q.Z);
                    Q = add_affine_table<COMB_MODE>(Q, entry, neg);
                }
            }
        }
        Q = ladder_result<COMB_MODE>(Q);
        if (DEFER) {                                          // deferred normalisation, see normalize_kernel
            if (live) {
                store_proj(proj, proj_stride, id, fe2_carry(fe2_cneg(Q.X, c.negate)), Q.Y, Q.Z);   // even scalar: -(x, y) = (-x, y)
                status[id] = FOURQ_DH_OK;
            }
            continue;
        }
        Fe2<1> ax, ay;
        r1_to_affine(Q, ax, ay);
        ax = fe2_carry(fe2_cneg(ax, c.negate));               // even scalar: [k]B = -[N - k]B, -(x, y) = (-x, y)
        u64 o[8];
        store_fe2(o, ax); store_fe2(o + 4, ay);
        const bool neutral = (o[0] | o[1] | o[2] | o[3] | o[5] | o[6] | o[7]) == 0 && o[4] == 1;
        if (live) {
            uint4* dst = reinterpret_cast<uint4*>(out + 8 * (size_t)id);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u64 lo = neutral ? 0 : o[2 * k], hi = neutral ? 0 : o[2 * k + 1];
                dst[k] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            }
            status[id] = neutral ? FOURQ_DH_NEUTRAL : FOURQ_DH_OK;
        }
    }
}

// R1toAffine for a whole batch with one GF(p) inversion per K elements (Montgomery's trick; SURVEY 8f row 4,
// curve4q.py:103-106, fields.py:66-106, :193-199).  Lane t owns elements t, t + T, ..., t + (K-1)T with
// T = ceil(n / K): it multiplies up the norms |Z|^2, inverts the product once and peels the individual inverses
// off on the way back.  Elements already rejected (status != 0) and the ragged tail contribute a 1.  The affine
// result is canonical, hence identical to the per-element inversion.  Z != 0 for every point of the curve (the
// addition law is complete), so a product is zero only for rejected elements, which are masked.
template <int K>
__global__ __launch_bounds__(BLOCK) void normalize_kernel(const uint4* proj, u32 stride, u64* out, uint8_t* status, u32 n) {
    const u32 T = (n + K - 1) / K;
    const u32 t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    // every loop over the K elements is unrolled in full: left to its size heuristics the compiler keeps z[], nz[] and pre[] in
    // scratch memory and walks them by a run-time index (656 bytes per lane at K = 8), a memory round trip per use in a kernel
    // that is nothing but one dependent chain per lane
    Fe<1> one;
    one.l[0] = 1; one.l[1] = one.l[2] = one.l[3] = one.l[4] = 0;
    Fe2<1> z[K];
    uint8_t st[K];
#pragma clang loop unroll(full)
    for (int j = 0; j < K; j++) {                     // all loads first: the lane's K elements are K separate cache lines
        const u32 id = t + (u32)j * T, at = id < n ? id : t;
        z[j] = load_proj_z(proj, stride, at);
        st[j] = id < n ? status[at] : (uint8_t)FOURQ_DH_NOT_ON_CURVE;
    }
    Fe<1> nz[K], pre[K];
#pragma clang loop unroll(full)
    for (int j = 0; j < K; j++) {
        Fe<1> norm = fe_carry(fe_add(fe_sqr(z[j].re), fe_sqr(z[j].im)));
        nz[j] = fe_select(st[j] == FOURQ_DH_OK ? ~0u : 0u, norm, one);
        if (j == 0) pre[0] = nz[0]; else pre[j] = fe_mul(pre[j - 1], nz[j]);
    }
    Fe<1> inv = fe_inv_inline(pre[K - 1]);
#pragma clang loop unroll(full)
    for (int j = K - 1; j >= 0; j--) {
        const u32 id = t + (u32)j * T, at = id < n ? id : t;
        Fe2<1> X, Y, zi;
        load_proj_xy(proj, stride, at, X, Y);
        Fe<1> ninv = inv;                                              // 1 / |Z_j|^2
        if (j > 0) { ninv = fe_mul(inv, pre[j - 1]); inv = fe_mul(inv, nz[j]); }
        zi.re = fe_mul(ninv, z[j].re);                                 // conj(Z) / |Z|^2     fields.py:193-199
        zi.im = fe_mul(ninv, fe_neg(z[j].im));
        u64 o[8];
        store_fe2(o, fe2_mul(X, zi));
        store_fe2(o + 4, fe2_mul(Y, zi));
        uint8_t s = st[j];
        if (s == FOURQ_DH_OK && (o[0] | o[1] | o[2] | o[3] | o[5] | o[6] | o[7]) == 0 && o[4] == 1) s = FOURQ_DH_NEUTRAL;   // curve4q.py:459-460
        if (id < n) {
            uint4* dst = reinterpret_cast<uint4*>(out + 8 * (size_t)id);
#pragma clang loop unroll(full)
            for (int k = 0; k < 4; k++) {
                u64 lo = s ? 0 : o[2 * k], hi = s ? 0 : o[2 * k + 1];
                dst[k] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            }
            status[id] = s;
        }
    }
}

#endif

}  // namespace

// launchers implemented in fourq_chain.hip (FQ_CHAIN=1 code objects)
int chain_launch_ladder(int algo, int src, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a);
int chain_launch_prep(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a);
int chain_setup_device();       // per-device function attributes (the comb's dynamic LDS); called by fourq_ctx_create
int chain_launch_comb(unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, uint4* proj, u32 proj_stride, u32 n);
int chain_launch_normalize(int k, hipStream_t stream, const uint4* proj, u32 proj_stride, u64* out, uint8_t* status, u32 n);   // k in {1, 2, 4, 8}
// constant-time selection builds of the same kernels: fourq_ct_fused.hip (FQ_CHAIN=0) and fourq_ct_chain.hip (FQ_CHAIN=1)
int ct_launch_fused(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a);
int ct_launch_pair(int algo, bool dh, bool fixed, bool quad, unsigned grid, hipStream_t stream, const LadderArgs& a);
int ct_launch_pair_mixed(bool quad, unsigned grid, hipStream_t stream, const LadderArgs& a);
int ct_launch_comb_quad(bool quad, unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, u32 n);
int ct_launch_mixed_tail(unsigned prep_grid, unsigned tail_grid, hipStream_t stream, const LadderArgs& a, const u32* fix_list, const u32* var_list, u32* counts,
                         u32* over_scratch, u32 lanes, u32 limit);     // split_counts_kernel must already have run: see fourq_ct_chain.hip
int ct_launch_split_counts(hipStream_t stream, u32* counts, u32 lanes, u32 limit);
int ct_launch_mixed_queue(unsigned grid, hipStream_t stream, const LadderArgs& a, const u32* var_list, const u32* fix_list, const u32* counts, u32* queue_head);
int ct_launch_lds(int algo, bool dh, unsigned grid, hipStream_t stream, const LadderArgs& a);        // defers when a.proj != NULL
int ct_launch_comb(unsigned grid, hipStream_t stream, const u64* scalars, const u32* comb_limbs, u64* out, uint8_t* status, uint4* proj, u32 proj_stride, u32 n);

}  // namespace fq
