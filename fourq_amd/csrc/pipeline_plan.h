// Chunk plan of a host-array call (run_pipeline in fourq_amd.hip).  Plain C++, no HIP: tests/test_pipeline_plan.py compiles it with g++.
//
// A call of W whole kernel generations (+ a tail of less than one) is cut into chunks that go copy-in -> kernels -> copy-out on three
// streams.  What the call costs beyond its kernels is
//   (a) the first chunk's copy-in and the last chunk's copy-out, which nothing overlaps        -> first and last chunk: ONE generation;
//   (b) a fixed price per chunk boundary on the kernel stream (an event record, a cross-stream wait, the launch gaps of the chunk's
//       kernels: ~20 us measured, profiles/r05_pipeline.txt)                                   -> as few chunks as possible;
//   (c) stalls: the kernels of a chunk cannot start before its LAST input byte has arrived     -> a chunk may only be as large as the
//       copy engine's lead allows.
// With h = copy-in time, k = kernel time, d = copy-out time of one generation: the copy-in stream runs ahead of the kernels at h per
// generation, so a chunk that covers generations [a, b) does not stall if  h b <= h + k a,  i.e.  b <= 1 + (k / h) a  -- the boundaries may
// grow geometrically with ratio k / h from the front.  From the back the mirror image holds for the copies out: chunk j's copy-out (d s_j)
// must be done when chunk j+1's kernels (k s_{j+1}) are, or the last copy-out starts late: s_j <= (k / d) s_{j+1}.  Formats that move few
// bytes per operation (affine, encoded points) reach large chunks within two or three steps; raw R1 in and out (k / h ~ 1.15) stays at one
// generation per chunk for a long while -- which is exactly what the sweep measured (R1: 1 generation per chunk best, affine 2, bytes 4).
#pragma once
#include <cstddef>
#include <vector>

namespace fq_plan {

struct Piece { size_t off, m; };

// Link rate both directions busy (tools/microbench/link_duplex.hip: 96.5 GB/s summed, 48 each) in bytes per nanosecond, and the share of
// the theoretical lead a plan may use (copies of other chunks share the link, rates vary by box).
constexpr double LINK_BYTES_PER_NS = 48.0;
constexpr double SAFETY = 0.85;
constexpr size_t SLOT_BYTES_MAX = 64u << 20;        // a slot holds the largest chunk's inputs and outputs: at most this much
constexpr size_t GENS_MAX = 8;

inline size_t clamp_gens(double v, size_t lo, size_t hi) {
    if (!(v >= (double)lo)) return lo;
    if (v >= (double)hi) return hi;
    return (size_t)v;
}

// Sizes, in generations, of the chunks that cover `whole` generations.  `ns_per_elem`: kernel time per element of the route; `in_bytes`,
// `out_bytes`: bytes per element copied in / out; `unit`: elements per generation; `gens_fixed` > 0 forces the round 5 uniform shape
// (first and last chunk one generation, `gens_fixed` in between: the FOURQ_PIPE_GENS test hook).
inline std::vector<size_t> plan_generations(size_t whole, size_t unit, size_t in_bytes, size_t out_bytes, double ns_per_elem, int gens_fixed) {
    std::vector<size_t> sizes;
    if (whole == 0) return sizes;
    size_t cap = SLOT_BYTES_MAX / (unit * (in_bytes + out_bytes) + 1);
    if (cap < 1) cap = 1;
    if (cap > GENS_MAX) cap = GENS_MAX;
    if (gens_fixed > 0) {
        const size_t g = (size_t)gens_fixed;
        sizes.push_back(1);
        size_t inner = whole >= 2 ? whole - 2 : 0;
        while (inner) { const size_t s = inner < g ? inner : g; sizes.push_back(s); inner -= s; }
        if (whole >= 2) sizes.push_back(1);
        return sizes;
    }
    const double rin = in_bytes ? SAFETY * ns_per_elem * LINK_BYTES_PER_NS / (double)in_bytes : 1e9;
    const double rout = out_bytes ? SAFETY * ns_per_elem * LINK_BYTES_PER_NS / (double)out_bytes : 1e9;
    if (rin < 1.0 || rout < 1.0) {          // a copy direction is (nearly) as slow as the kernels: the link is the pace, and the smallest chunks
        sizes.assign(whole, 1);             // keep it busy from the first generation to the last
        return sizes;
    }
    // Chunks are laid from both ends towards the middle, alternately.  A front chunk at generations [a, a + f) obeys the copy-in rule
    // (a + f <= 1 + rin a); a back chunk obeys the copy-out rule against the chunk behind it (b <= rout * next) AND, its position being
    // known, the copy-in rule; the chunk that closes the gap has a known neighbour on both sides and obeys both.
    std::vector<size_t> front, back;
    size_t used = 0, fb = 0;
    while (used < whole) {
        size_t f = front.empty() ? 1 : clamp_gens(1.0 + rin * (double)fb - (double)fb, 1, cap);
        if (f >= whole - used) {                                    // closes the gap: the chunk behind it is back.back()
            f = whole - used;
            if (!back.empty()) { const size_t lim = clamp_gens(rout * (double)back.back(), 1, cap); if (f > lim) f = lim; }
        }
        front.push_back(f); fb += f; used += f;
        if (used == whole) break;
        size_t b = back.empty() ? 1 : clamp_gens(rout * (double)back.back(), 1, cap);
        if (b > whole - used) b = whole - used;
        const size_t end = whole - (used - fb);                     // generations in front of the chunks already laid at the back
        while (b > 1 && (double)end > 1.0 + rin * (double)(end - b)) b--;
        back.push_back(b); used += b;
    }
    sizes = front;
    for (size_t i = back.size(); i-- > 0;) sizes.push_back(back[i]);
    return sizes;
}

// The pieces (offset, length in elements) of a call of n > unit elements: the generations above, then the tail of less than one.
inline std::vector<Piece> plan_pieces(size_t n, size_t unit, size_t in_bytes, size_t out_bytes, double ns_per_elem, int gens_fixed) {
    std::vector<Piece> plan;
    const size_t whole = n / unit, tail = n - whole * unit;
    size_t off = 0;
    for (size_t g : plan_generations(whole, unit, in_bytes, out_bytes, ns_per_elem, gens_fixed)) { plan.push_back({ off, g * unit }); off += g * unit; }
    if (tail) plan.push_back({ off, tail });
    return plan;
}

}  // namespace fq_plan
