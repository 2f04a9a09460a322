// Do VGPR bank conflicts (vgpr number mod 4) change v_mad_u64_u32 throughput on gfx950?  Kernel wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
// five accumulators v[10:11] v[12:13] v[14:15] v[16:17] v[18:19]; sources chosen per MODE
#define BODY(S0, S1, S2, S3, S4, T0, T1, T2, T3, T4) \
    "v_mad_u64_u32 v[10:11], vcc, " S0 ", " T0 ", v[10:11]\n\t" \
    "v_mad_u64_u32 v[12:13], vcc, " S1 ", " T1 ", v[12:13]\n\t" \
    "v_mad_u64_u32 v[14:15], vcc, " S2 ", " T2 ", v[14:15]\n\t" \
    "v_mad_u64_u32 v[16:17], vcc, " S3 ", " T3 ", v[16:17]\n\t" \
    "v_mad_u64_u32 v[18:19], vcc, " S4 ", " T4 ", v[18:19]\n\t"
template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint32_t x = seed + threadIdx.x;
    asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\t"
                 "v_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\t"
                 "v_mov_b32 v10, %0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, 0"
                 :: "v"(x) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0)      // conflict-free: acc banks (2,3),(0,1),(2,3),(0,1),(2,3); sources in the two other banks
            asm volatile(BODY("v20","v22","v24","v26","v28", "v21","v23","v25","v27","v29") BODY("v20","v22","v24","v26","v28", "v21","v23","v25","v27","v29")
                         ::: "vcc","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        else if (MODE == 1) // both sources in the same bank as each other (bank 0), acc in (2,3)/(0,1)
            asm volatile(BODY("v20","v24","v28","v32","v20", "v24","v28","v32","v20","v24") BODY("v20","v24","v28","v32","v20", "v24","v28","v32","v20","v24")
                         ::: "vcc","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        else                // sources collide with the accumulator's banks: acc v[10:11] (banks 2,3) with v22 (2), v23 (3) etc.
            asm volatile(BODY("v22","v20","v22","v20","v22", "v23","v21","v23","v21","v23") BODY("v22","v20","v22","v20","v22", "v23","v21","v23","v21","v23")
                         ::: "vcc","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
    }
    uint32_t s;
    asm volatile("v_add_u32 %0, v10, v12\n\tv_add_u32 %0, %0, v14\n\tv_add_u32 %0, %0, v16\n\tv_add_u32 %0, %0, v18" : "=v"(s));
    if (s == 0x12345678u) out[0] = s;
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    const char* names[3] = {"conflict-free banks", "src0/src1 same bank", "sources on accumulator banks"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        printf("%-30s:", names[mode]);
        for (int w : {1, 2, 4}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<2>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  W=%d %.2f ns/mad/SIMD", w, ms * 1e6 / ((double)ITERS * 10 * w));
        }
        printf("\n");
    }
    return 0;
}
