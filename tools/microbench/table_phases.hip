// s_memtime stamps INSIDE table_endo of the fused kernels (a copy of build_table_endo_lds with stamps; diagnostic only):
// cycles per sub-formula for a lone wave, against their instruction counts.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o table_phases table_phases.hip
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
constexpr int NS = 24;
template <typename L, typename EF> FQ_DEV void build_stamped(const R1& P, u32* slot, const EF& ef, uint64_t* ts) {
    int q = 0;
    ts[q++] = stamp();
    R2 result = r1_to_r2(P);
    int result_at = 0;
    const Fe2<1> n0 = result.N, d0 = result.D;
    Fe2<1> X = P.X, Y = P.Y, Z = P.Z;
    auto store_result = [&]() {
        store_entry<L>(slot, result_at, result, ef);
        if (result_at >= 1 && result_at <= 3) ef.park_nd(8 - result_at, result.N, result.D);
    };
    ts[q++] = stamp();                                   // r1_to_r2(P)
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        store_result();
        Proj<1, 2, 1> t;
        if (step == 1) { t.X = X; t.Y = widen<2>(Y); t.Z = Z; }
        else { t = tau(X, Y, Z); if (step == 0) ef.park_xyz(6, t.X, fe2_carry(t.Y), t.Z); }
        ts[q++] = stamp();                               // tau
        Proj<2, 2, 2> u;
        if (step == 0) u = upsilon(t);
        else { Proj<1, 1, 1> c = chi(t); u.X = widen<2>(c.X); u.Y = widen<2>(c.Y); u.Z = widen<2>(c.Z); }
        ts[q++] = stamp();                               // upsilon / chi
        R1 V = tau_dual(u.X, u.Y, u.Z);
        R3 V3 = r1_to_r3(V);
        ts[q++] = stamp();                               // tau_dual + r1_to_r3
        R2 base;
        base.N = n0; base.D = d0; base.E = ef.get(0, 0); base.F = ef.get(0, 1);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 next = base;
            if (m + 1 < half) { ef.unpark_nd(8 - (m + 1), next.N, next.D); next.E = ef.get((u32)(m + 1), 0); next.F = ef.get((u32)(m + 1), 1); }
            if (m > 0) store_result();
            if (m == 0 && step == 0) ef.park_xyz(4, V.X, V.Y, V.Z);
            if (m == half - 1 && step < 2) ef.unpark_xyz(step == 0 ? 6 : 4, X, Y, Z);
            result = r1_to_r2(add_core(V3, as_signed(base)));
            result_at = half + m;
            base = next;
        }
        ts[q++] = stamp();                               // the step's additions
    }
    store_result();
    ts[q++] = stamp();
}
// the same stamps around the GENERATED table bodies (build_table_endo_lds_asm of kernels.hip.h; -DFQ_TABLE_ASM=1): a body's instruction count is known
// exactly (tools/asmgen/gen_ladder_step.py --stats), so cycles - 4 x instructions is what the memory side and the glue cost
template <typename L, typename EF> FQ_DEV void build_stamped_asm(const R1& P, u32* slot, const EF& ef, uint64_t* ts) {
    int q = 0;
    ts[q++] = stamp();
    R2 result = r1_to_r2_asm(P);
    int result_at = 0;
    const Fe2<1> n0 = result.N, d0 = result.D;
    Fe2<1> X = P.X, Y = P.Y, Z = P.Z;
    auto store_result = [&]() {
        store_entry<L>(slot, result_at, result, ef);
        if (result_at >= 1 && result_at <= 3) ef.park_nd(8 - result_at, result.N, result.D);
    };
    ts[q++] = stamp();
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        store_result();
        if (step != 1) { tau_asm(X, Y, Z); if (step == 0) ef.park_xyz(6, X, Y, Z); }
        ts[q++] = stamp();
        if (step == 0) upsilon_asm(X, Y, Z); else chi_asm(X, Y, Z);
        ts[q++] = stamp();
        Fe2<2> N3, D3; Fe2<1> F3;
        taudual_asm(X, Y, Z, N3, D3, F3);
        const Fe2<1> E3 = Z;
        ts[q++] = stamp();
        R2 base;
        base.N = n0; base.D = d0; base.E = ef.get(0, 0); base.F = ef.get(0, 1);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            R2 next = base;
            if (m + 1 < half) { ef.unpark_nd(8 - (m + 1), next.N, next.D); next.E = ef.get((u32)(m + 1), 0); next.F = ef.get((u32)(m + 1), 1); }
            if (m > 0) store_result();
            if (m == 0 && step == 0) ef.park_xyz(4, X, Y, Z);
            if (m == half - 1 && step < 2) ef.unpark_xyz(step == 0 ? 6 : 4, X, Y, Z);
            result = base;
            table_add_asm(result, N3, D3, E3, F3);
            result_at = half + m;
            base = next;
        }
        ts[q++] = stamp();
    }
    store_result();
    ts[q++] = stamp();
}
__global__ __launch_bounds__(256, 1) void k(const u64* points, u32* scratch, uint64_t* stamps, u64* sink) {
    __shared__ __attribute__((aligned(16))) u32 lds_mem[EF_LDS_U32];
    LdsEF ef; ef.lane = reinterpret_cast<uint2*>(lds_mem) + threadIdx.x;
    const u32 id = blockIdx.x * 256 + threadIdx.x;
    R1 P = load_r1(points + 20 * (size_t)id);
    uint64_t ts[NS];
    for (int i = 0; i < NS; i++) ts[i] = 0;
#if FQ_TABLE_ASM
    build_stamped_asm<NDSlots>(P, scratch + (size_t)id * NDSlots::SLOT, ef, ts);
#else
    build_stamped<NDSlots>(P, scratch + (size_t)id * NDSlots::SLOT, ef, ts);
#endif
    Fe2<1> e = ef.get(7, 0);
    if (e.re.l[0] == 0x7fffffffu) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < NS; i++) stamps[(size_t)(id >> 6) * NS + i] = ts[i];
}
int main() {
    const int n = 1 << 16;
    u64 *p, *sink; u32* scr; uint64_t* st;
    CHECK(hipMalloc(&p, n * 160)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&scr, (size_t)n * NDSlots::SLOT * 4)); CHECK(hipMalloc(&st, 1024 * NS * 8));
    std::vector<u64> hp(n * 20);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < hp.size(); i++) hp[i] = (i & 1) ? (rnd() >> 1) : rnd();
    CHECK(hipMemcpy(p, hp.data(), n * 160, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 50; rep++) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, p, scr, st, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h(1024 * NS);
    CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[] = { "r1_to_r2(P)", "s0 tau", "s0 upsilon", "s0 tau_dual", "s0 1 add", "s1 (tau shared)", "s1 chi", "s1 tau_dual", "s1 2 adds", "s2 tau", "s2 chi", "s2 tau_dual", "s2 4 adds", "last store" };
    // multiply-adds of each part (M = 100, S = 50): r1_to_r2 2M; tau 5M+3S; upsilon 20M+5S; tau_dual+r1_to_r3 6M+3S; add_core+r1_to_r2 9M; chi 9M+3S
    const int mads[] = { 200, 650, 2250, 750, 900, 0, 1050, 750, 1800, 650, 1050, 750, 3600, 0 };
    // instructions of the generated bodies in each part (gen_ladder_step.py --stats): R1TOR2 472; TAU 1 021; UPSILON 3 199; TAUDUAL 1 196; TABLEADD 1 473; CHI 1 445
    const int asm_instr[] = { 472, 1021, 3199, 1196, 1473, 0, 1445, 1196, 2 * 1473, 1021, 1445, 1196, 4 * 1473, 0 };
    uint64_t total = 0;
    for (int i = 0; i < 14; i++) {
        std::vector<uint64_t> c; for (int w = 0; w < 1024; w++) c.push_back(h[(size_t)w * NS + i + 1] - h[(size_t)w * NS + i]);
        std::sort(c.begin(), c.end());
        total += c[512];
        printf("%-18s %8llu cycles  %5d multiply-adds  %s\n", names[i], (unsigned long long)c[512], mads[i], mads[i] ? "" : "");
        if (mads[i]) printf("                   -> %.1f cycles per multiply-add\n", (double)c[512] / mads[i]);
#if FQ_TABLE_ASM
        if (asm_instr[i]) printf("                   -> %d body instructions x 4 = %d cycles; the rest (glue, LDS / HBM side, stamp): %lld\n", asm_instr[i], 4 * asm_instr[i], (long long)c[512] - 4 * asm_instr[i]);
#endif
    }
    printf("sum %llu cycles\n", (unsigned long long)total);
    return 0;
}
