// Feasibility probe for a GATED host-array pipeline ("what remains"): ONE persistent launch covers all generations of a
// call; before generation g a wave spins on a device-side counter that the copy-in stream bumps behind H2D copy g, and after its stores it
// bumps done[g], on which a one-wave kernel on the copy-out stream spins before D2H copy g is allowed to start.  Questions:
//   1. does a running kernel see bytes that a DMA copy wrote AFTER the kernel started (stale L2 lines of an earlier call are the hazard)?
//      -- with plain hipMalloc memory, and with hipDeviceMallocUncached memory;
//   2. what does a call cost against the chunked pipeline of the library (one launch per generation, events between the streams)?
// The "work" of a generation is a dependent multiply-add loop of a chosen length per lane.  Every spin is bounded by the 100 MHz counter
// (0.3 s) and an abort flag that makes every wave leave.
//   hipcc -O3 --offload-arch=gfx950 -o gated gated.hip && ./gated [iters_per_gen]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int LANES = 65536, BLOCK = 256, IN_U4 = 12, OUT_U4 = 10;      // 192 B in, 160 B out per element, as MUL_endo
constexpr uint64_t TIMEOUT_TICKS = 30000000ull;                          // 0.3 s of s_memrealtime

struct Gate { uint32_t ready, abort_, pad[14]; uint32_t done[64]; };

__device__ __forceinline__ uint4 work(const uint4* in, int iters) {
    uint64_t a0 = in[0].x | ((uint64_t)in[0].y << 32), a1 = in[1].z, a2 = in[5].w, a3 = in[11].x;
    uint32_t b = in[3].x | 1, c = in[7].y | 3;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %5, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %4, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
    }
    uint32_t mix = 0;
    for (int k = 0; k < IN_U4; k++) mix ^= in[k].x * 3u + in[k].y * 5u + in[k].z * 7u + in[k].w * 11u;      // every input word counts
    return make_uint4((uint32_t)a0 ^ mix, (uint32_t)(a0 >> 32) + (uint32_t)a1, (uint32_t)a2 ^ (uint32_t)(a2 >> 32), (uint32_t)a3 + mix);
}

// generation g of the batch: elements [g * LANES, (g + 1) * LANES)
__global__ __launch_bounds__(BLOCK, 1) void gen_kernel(const uint4* in, uint4* out, int first_gen, int gens, int iters, Gate* gate) {
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    for (int g = first_gen; g < first_gen + gens; g++) {
        if (gate) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(&gate->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (uint32_t)(g + 1)) {
                if (__hip_atomic_load(&gate->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
                if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) { __hip_atomic_store(&gate->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        const size_t e = (size_t)g * LANES + lane;
        uint4 v[IN_U4];
        for (int k = 0; k < IN_U4; k++) v[k] = in[e * IN_U4 + k];
        const uint4 r = work(v, iters);
        for (int k = 0; k < OUT_U4; k++) out[e * OUT_U4 + k] = make_uint4(r.x + k, r.y, r.z, r.w ^ (uint32_t)e);
        if (gate) {
            __threadfence_system();
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&gate->done[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void publish_kernel(Gate* gate, uint32_t value) { __hip_atomic_store(&gate->ready, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void wait_done_kernel(Gate* gate, int g, uint32_t expected) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(&gate->done[g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < expected) {
        if (__hip_atomic_load(&gate->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT_TICKS) { __hip_atomic_store(&gate->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ void touch_kernel(const uint4* p, size_t n, uint32_t* sink) {        // pulls the buffer's current contents into the caches
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}

static void fill(uint32_t* p, size_t words, uint32_t seed) { uint32_t x = seed * 2654435761u + 1; for (size_t i = 0; i < words; i++) { x = x * 1664525u + 1013904223u; p[i] = x; } }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    const int GENS = 16;
    const size_t n = (size_t)GENS * LANES, in_bytes = n * IN_U4 * 16, out_bytes = n * OUT_U4 * 16;
    hipStream_t kern, cin, cout;
    CHECK(hipStreamCreateWithFlags(&kern, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&cin, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&cout, hipStreamNonBlocking));
    uint4 *h_in, *h_out, *h_ref;
    CHECK(hipHostMalloc((void**)&h_in, in_bytes)); CHECK(hipHostMalloc((void**)&h_out, out_bytes)); CHECK(hipHostMalloc((void**)&h_ref, out_bytes));
    uint32_t* sink; CHECK(hipMalloc((void**)&sink, 4));
    for (int mode = 0; mode < 2; mode++) {
        uint4 *d_in = nullptr, *d_out = nullptr; Gate* gate = nullptr;
        hipError_t e1, e2, e3;
        if (mode == 0) { e1 = hipMalloc((void**)&d_in, in_bytes); e2 = hipMalloc((void**)&d_out, out_bytes); }
        else { e1 = hipExtMallocWithFlags((void**)&d_in, in_bytes, hipDeviceMallocUncached); e2 = hipExtMallocWithFlags((void**)&d_out, out_bytes, hipDeviceMallocUncached); }
        e3 = hipExtMallocWithFlags((void**)&gate, sizeof(Gate), hipDeviceMallocUncached);
        if (e3 != hipSuccess) { (void)hipGetLastError(); e3 = hipExtMallocWithFlags((void**)&gate, sizeof(Gate), hipDeviceMallocFinegrained); printf("gate: uncached allocation refused, fine-grained: %s\n", hipGetErrorString(e3)); }
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { printf("mode %d: allocation failed (%s / %s / %s)\n", mode, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3)); (void)hipGetLastError(); continue; }
        printf("== data buffers: %s\n", mode == 0 ? "hipMalloc" : "hipExtMallocWithFlags(hipDeviceMallocUncached)");
        for (int variant = 0; variant < 3; variant++) {          // 0: one launch, data resident (floor); 1: chunked, one launch per generation; 2: gated
            double best = 1e9; size_t bad_total = 0; int aborted = 0;
            for (int call = 0; call < 5; call++) {
                fill((uint32_t*)h_in, in_bytes / 4, 1000u * mode + 10u * variant + call);
                memset(h_out, 0, out_bytes);
                // reference outputs of THIS call's inputs: the plain kernel on resident data (and it leaves this call's ... no: a scratch copy)
                uint4 *r_in, *r_out; CHECK(hipMalloc((void**)&r_in, in_bytes)); CHECK(hipMalloc((void**)&r_out, out_bytes));
                CHECK(hipMemcpy(r_in, h_in, in_bytes, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(gen_kernel, dim3(LANES / BLOCK), dim3(BLOCK), 0, kern, r_in, r_out, 0, GENS, iters, (Gate*)nullptr);
                CHECK(hipStreamSynchronize(kern));
                CHECK(hipMemcpy(h_ref, r_out, out_bytes, hipMemcpyDeviceToHost));
                CHECK(hipFree(r_in)); CHECK(hipFree(r_out));
                // the device buffers still hold the PREVIOUS call's bytes: pull them into the caches, as an earlier kernel would have left them
                hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, kern, d_in, in_bytes / 16, sink);
                hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, kern, d_out, out_bytes / 16, sink);
                CHECK(hipMemset(gate, 0, sizeof(Gate)));
                CHECK(hipDeviceSynchronize());
                const size_t gi = (size_t)LANES * IN_U4, go = (size_t)LANES * OUT_U4;
                auto t0 = std::chrono::steady_clock::now();
                if (variant == 0) {
                    CHECK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, kern));
                    auto t1 = std::chrono::steady_clock::now();
                    CHECK(hipStreamSynchronize(kern));
                    t0 = std::chrono::steady_clock::now(); (void)t1;
                    hipLaunchKernelGGL(gen_kernel, dim3(LANES / BLOCK), dim3(BLOCK), 0, kern, d_in, d_out, 0, GENS, iters, (Gate*)nullptr);
                    CHECK(hipStreamSynchronize(kern));
                    auto t2 = std::chrono::steady_clock::now();
                    best = std::min(best, std::chrono::duration<double, std::milli>(t2 - t0).count());
                    CHECK(hipMemcpy(h_out, d_out, out_bytes, hipMemcpyDeviceToHost));
                } else if (variant == 1) {
                    std::vector<hipEvent_t> in_done(GENS), k_done(GENS);
                    for (int g = 0; g < GENS; g++) { CHECK(hipEventCreateWithFlags(&in_done[g], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&k_done[g], hipEventDisableTiming)); }
                    t0 = std::chrono::steady_clock::now();
                    for (int g = 0; g < GENS; g++) {
                        CHECK(hipMemcpyAsync(d_in + g * gi, h_in + g * gi, gi * 16, hipMemcpyHostToDevice, cin));
                        CHECK(hipEventRecord(in_done[g], cin));
                        CHECK(hipStreamWaitEvent(kern, in_done[g], 0));
                        hipLaunchKernelGGL(gen_kernel, dim3(LANES / BLOCK), dim3(BLOCK), 0, kern, d_in, d_out, g, 1, iters, (Gate*)nullptr);
                        CHECK(hipEventRecord(k_done[g], kern));
                        CHECK(hipStreamWaitEvent(cout, k_done[g], 0));
                        CHECK(hipMemcpyAsync(h_out + g * go, d_out + g * go, go * 16, hipMemcpyDeviceToHost, cout));
                    }
                    CHECK(hipStreamSynchronize(cout));
                    auto t2 = std::chrono::steady_clock::now();
                    best = std::min(best, std::chrono::duration<double, std::milli>(t2 - t0).count());
                    for (int g = 0; g < GENS; g++) { CHECK(hipEventDestroy(in_done[g])); CHECK(hipEventDestroy(k_done[g])); }
                } else {
                    t0 = std::chrono::steady_clock::now();
                    hipLaunchKernelGGL(gen_kernel, dim3(LANES / BLOCK), dim3(BLOCK), 0, kern, d_in, d_out, 0, GENS, iters, gate);
                    for (int g = 0; g < GENS; g++) {
                        CHECK(hipMemcpyAsync(d_in + g * gi, h_in + g * gi, gi * 16, hipMemcpyHostToDevice, cin));
                        hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, cin, gate, (uint32_t)(g + 1));
                        hipLaunchKernelGGL(wait_done_kernel, dim3(1), dim3(1), 0, cout, gate, g, (uint32_t)(LANES / 64));
                        CHECK(hipMemcpyAsync(h_out + g * go, d_out + g * go, go * 16, hipMemcpyDeviceToHost, cout));
                    }
                    CHECK(hipStreamSynchronize(cout)); CHECK(hipStreamSynchronize(kern)); CHECK(hipStreamSynchronize(cin));
                    auto t2 = std::chrono::steady_clock::now();
                    best = std::min(best, std::chrono::duration<double, std::milli>(t2 - t0).count());
                    Gate hg; CHECK(hipMemcpy(&hg, gate, sizeof hg, hipMemcpyDeviceToHost));
                    aborted += hg.abort_ != 0;
                }
                size_t bad = 0;
                for (size_t i = 0; i < out_bytes / 16; i++) bad += memcmp(&h_out[i], &h_ref[i], 16) != 0;
                bad_total += bad;
            }
            printf("  %-44s best %.3f ms   wrong output words: %zu of %zu x 5 calls%s\n",
                   variant == 0 ? "one launch, inputs resident (kernels only)" : variant == 1 ? "chunked: a launch per generation, events" : "GATED: one persistent launch, flags",
                   best, bad_total, out_bytes / 16, aborted ? "   (ABORTED by a spin timeout)" : "");
        }
        CHECK(hipFree(d_in)); CHECK(hipFree(d_out)); CHECK(hipFree(gate));
    }
    printf("done\n");
    return 0;
}
