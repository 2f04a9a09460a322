// Where the headline kernel (ladder_kernel<ENDO, FUSED>) spends its cycles: the same device functions, same build flavour
// (FQ_CHAIN=0, chained ladder with preloaded entries, NDSlots + LdsEF), one wave per SIMD on every CU, with s_memtime
// stamps between the phases.  Variants isolate the memory side of the ladder:
//   0  the kernel as shipped (N, D gathered from the lane's HBM slot, E, F from LDS)
//   1  the ladder gathers ALWAYS entry 0 (same instruction stream, all of a lane's gathers hit one address)
//   2  no table construction (slots pre-filled by variant 0's previous launch): ladder only
// Round 4: build it through the placement pass too (python: fourq_amd.build.compile_unit(path, obj, flags, place=True), then hipcc obj -o exe).
// Diagnostic build, never part of the product:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -o phases phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define FQ_CHAIN 0
#include "../../fourq_amd/csrc/kernels.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

FQ_DEV uint64_t stamp() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

// the product's MUL_endo ladder (ladder_endo<3, true, NDSlots, LdsEF> of kernels.hip.h) with the entry index of the HBM-side gather (N, D)
// and of the LDS read (E, F) masked by run-time values: 7 = by digit as shipped, 0 = always entry 0 (same instructions, one address)
FQ_DEV R1 ladder_probe(const EndoDigits& e, const u32* tbl, const LdsEF& ef, u32 gmask, u32 lmask) {
    Proj<1, 1, 1> q4 = start_table<NDSlots>(tbl + (e.top & 7) * NDSlots::ENTRY, 0u);
    q4.Z = ef.get(e.top & 7, 0);
    R1 Q; Q.X = q4.X; Q.Y = q4.Y; Q.Z = q4.Z; Q.Ta = widen<4>(q4.X); Q.Tb = widen<2>(q4.Y);
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        const u32 digit = endo_digit(e, i);
        const u32 neg = endo_neg_mask(e, i);
        EntryRegs t = load_entry<NDSlots>(tbl + (digit & gmask) * NDSlots::ENTRY, neg, digit & lmask, ef);
        Fe2<1> T;
        dblt_asm(Q.X, Q.Y, Q.Z, T);
        add_asm(Q, T, t, neg);
    }
    return ladder_result<3>(Q);
}
template <int VARIANT> __global__ __launch_bounds__(256, 1) void k(const u64* scalars, const u64* points, u64* out, u32* scratch, uint64_t* stamps, u32 gmask, u32 lmask) {
    __shared__ __attribute__((aligned(16))) u32 lds_mem[EF_LDS_U32];
    LdsEF ef; ef.lane = reinterpret_cast<uint2*>(lds_mem) + threadIdx.x;
    const u32 id = blockIdx.x * 256 + threadIdx.x;
    if (id >= (1u << 16)) return;                  // "dead" blocks of an over-sized grid (variant 3 is launched with 512 blocks)
    uint64_t t0 = stamp();
    u64 m[4];
    load_scalar(scalars + 4 * (size_t)id, m);
    R1 P = load_r1(points + 20 * (size_t)id);
    u32* slot = scratch + (size_t)id * NDSlots::SLOT;
    uint64_t t1 = stamp();
#if FQ_TABLE_ASM
    if (VARIANT != 2) build_table_endo_lds_asm<NDSlots>(P, slot, ef);         // -DFQ_TABLE_ASM=1: the generated table bodies
#else
    if (VARIANT != 2) build_table_endo_lds<NDSlots>(P, slot, ef);
#endif
    else { for (int kk = 0; kk < 8; kk++) { R2 t = r1_to_r2(P); ef.put(kk, t); } }      // LDS filled, HBM slot left from the previous launch
    uint64_t t2 = stamp();
    u64 v[4];
    decompose(m, v);
    EndoDigits e = recode(v);
    if (VARIANT == 1) { e.d[0] = e.d[1] = e.d[2] = 0; e.top = 0; }
    uint64_t t3 = stamp();
    R1 Q = (VARIANT >= 3) ? ladder_probe(e, (const u32*)slot, ef, gmask, lmask)
                          : ladder_endo<FQ_LADDER_ASM ? 3 : LADDER_CH, true, NDSlots>(e, (const u32*)slot, NDSlots::ENTRY, ef);      // the product kernel's ladder: the asm bodies since round 4
    uint64_t t4 = stamp();
    u64 o[20];
    store_r1(o, Q);
    uint4* dst = reinterpret_cast<uint4*>(out + 20 * (size_t)id);
#pragma unroll
    for (int q = 0; q < 10; q++) dst[q] = make_uint4((u32)o[2 * q], (u32)(o[2 * q] >> 32), (u32)o[2 * q + 1], (u32)(o[2 * q + 1] >> 32));
    uint64_t t5 = stamp();
    if ((threadIdx.x & 63) == 0) {
        uint64_t* w = stamps + 8 * (size_t)(id >> 6);
        w[0] = t1 - t0; w[1] = t2 - t1; w[2] = t3 - t2; w[3] = t4 - t3; w[4] = t5 - t4; w[5] = t5 - t0;
    }
}
int main() {
    const int n = 1 << 16;
    u64 *s, *p, *o; u32* scr; uint64_t* st;
    CHECK(hipMalloc(&s, n * 32)); CHECK(hipMalloc(&p, n * 160)); CHECK(hipMalloc(&o, n * 160));
    CHECK(hipMalloc(&scr, (size_t)n * PackedSlots::SLOT * 4)); CHECK(hipMalloc(&st, 1024 * 64));
    std::vector<u64> hs(n * 4), hp(n * 20);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto& v : hs) v = rnd();
    for (size_t i = 0; i < hp.size(); i++) hp[i] = (i & 1) ? (rnd() >> 1) : rnd();          // any residues: timing does not care whether they are points
    CHECK(hipMemcpy(s, hs.data(), n * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(p, hp.data(), n * 160, hipMemcpyHostToDevice));
    const char* names[6] = { "as shipped", "ladder gathers always entry 0", "no table construction", "probe ladder, by digit (= shipped)",
                             "probe: N, D (HBM side) always entry 0", "probe: E, F (LDS) always entry 0" };
    for (int variant = 0; variant < 6; variant++) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
        for (int rep = 0; rep < 60; rep++) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, s, p, o, scr, st, 7u, 7u);
            else if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, s, p, o, scr, st, 7u, 7u);
            else if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, s, p, o, scr, st, 7u, 7u);
            else hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, s, p, o, scr, st, variant == 4 ? 0u : 7u, variant == 5 ? 0u : 7u);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 40 && ms < best) best = ms;
        }
        std::vector<uint64_t> h(1024 * 8);
        CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        printf("%-32s kernel %.4f ms; median cycles per wave:", names[variant], best);
        const char* ph[6] = { "load", "table_endo", "decompose+recode", "ladder", "store", "total" };
        for (int q = 0; q < 6; q++) {
            std::vector<uint64_t> c; for (int w = 0; w < 1024; w++) c.push_back(h[8 * w + q]);
            std::sort(c.begin(), c.end());
            printf("  %s %llu", ph[q], (unsigned long long)c[512]);
        }
        printf("\n");
    }
    return 0;
}
