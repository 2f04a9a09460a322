// Is the host link full duplex for this runtime?  H2D alone, D2H alone, both at once on two streams (pinned memory, 64 MiB each),
// with a few extra streams created AND USED first (argv[1]) to see whether the answer depends on how streams map onto hardware queues.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int extra = argc > 1 ? atoi(argv[1]) : 0;
    const size_t B = 64u << 20;
    std::vector<hipStream_t> pad(extra);
    void* scratch; CHECK(hipMalloc(&scratch, 4096));
    for (auto& s : pad) { CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHECK(hipMemsetAsync(scratch, 0, 4096, s)); }   // USED: a stream takes its hardware queue at first use
    CHECK(hipDeviceSynchronize());
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *ha, *hb, *da, *db;
    CHECK(hipHostMalloc(&ha, B, hipHostMallocDefault)); CHECK(hipHostMalloc(&hb, B, hipHostMallocDefault));
    CHECK(hipMalloc(&da, B)); CHECK(hipMalloc(&db, B));
    auto run = [&](bool up, bool down) {
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (up) CHECK(hipMemcpyAsync(da, ha, B, hipMemcpyHostToDevice, s1));
            if (down) CHECK(hipMemcpyAsync(hb, db, B, hipMemcpyDeviceToHost, s2));
            CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        return best;
    };
    const double u = run(true, false), d = run(false, true), b = run(true, true);
    printf("extra streams %d: h2d %.3f ms (%.1f GB/s)  d2h %.3f ms (%.1f GB/s)  both %.3f ms (%.1f GB/s summed)  -> %s\n", extra, u, B / u / 1e6, d, B / d / 1e6, b, 2 * B / b / 1e6,
           b < 0.75 * (u + d) ? "overlap" : "serial");
    return 0;
}
