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
//
// Round 6.  (1) The rates are INPUTS now, not constants of one box and one mode: the context measures the kernel time per element of each
// route and the link's rate in each direction on one middle chunk of every multi-chunk call and plans the next call of that route with
// them (fourq_amd.hip, "measured planner inputs"); the constants below are only the first call's guess.  (2) The plan is the best of
// three under the model played through (play()): one generation per chunk, the round-5 greedy ramps, and a dynamic program over the
// chunk boundaries that prices a stall at its duration and a chunk boundary at GAP_NS -- it is allowed to stall a little where that is
// cheaper than another boundary, which is where raw R1 I/O gains (k / h ~ 1.1: [1 x 6, 2 x 3, 1 x 4] instead of 16 x 1).
#pragma once
#include <cstddef>
#include <vector>

namespace fq_plan {

struct Piece { size_t off, m; };

// Link rate both directions busy (tools/microbench/link_duplex.hip: 96.5 GB/s summed, 48 each) in bytes per nanosecond: the guess for a
// context's first call; and the share of the theoretical lead the greedy ramps may use (copies of other chunks share the link).
constexpr double LINK_BYTES_PER_NS = 48.0;
constexpr double SAFETY = 0.85;
constexpr double MODEL_MARGIN = 0.95;               // the dynamic program trusts measured rates to this much (copy times / 0.95)
constexpr double GAP_NS = 20e3;                     // a chunk boundary on the kernel stream: event record, cross-stream wait, launch gaps (profiles/r05_pipeline.txt)
constexpr size_t SLOT_BYTES_MAX = 64u << 20;        // a slot holds the largest chunk's inputs and outputs: at most this much
constexpr size_t GENS_MAX = 8;

// What a plan is priced with.  ns_per_elem: kernel time per element of the route; link_in / link_out: bytes per nanosecond each way.
struct Rates { double ns_per_elem, link_in, link_out; };

inline size_t clamp_gens(double v, size_t lo, size_t hi) {
    if (!(v >= (double)lo)) return lo;
    if (v >= (double)hi) return hi;
    return (size_t)v;
}
inline size_t gens_cap(size_t unit, size_t in_bytes, size_t out_bytes) {
    size_t cap = SLOT_BYTES_MAX / (unit * (in_bytes + out_bytes) + 1);
    if (cap < 1) cap = 1;
    if (cap > GENS_MAX) cap = GENS_MAX;
    return cap;
}

// The model, played through: copies in back to back at h per generation; a chunk's kernels start when its last byte is there and the chunk
// before it is done, and cost k per generation plus `gap` per chunk; copies out behind them at d per generation.  Returns the time of the
// last output byte (nanoseconds).
inline double play(const std::vector<size_t>& sizes, double h, double k, double d, double gap) {
    double t_in = 0, t_k = 0, t_out = 0;
    for (size_t g : sizes) {
        t_in += h * (double)g;
        t_k = (t_k > t_in ? t_k : t_in) + k * (double)g + gap;
        t_out = (t_out > t_k ? t_out : t_k) + d * (double)g;
    }
    return t_out;
}

// round 5's plan: geometric ramps from both ends that never stall under the stated rates (kept: a candidate of plan_generations)
inline std::vector<size_t> plan_greedy(size_t whole, size_t cap, double rin, double rout) {
    std::vector<size_t> sizes;
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

// Dynamic program over the chunk boundaries: f[a] = the earliest time the kernels can be done with generations [0, a), over all ways to
// cut them into chunks of at most `cap`.  The tail (whose copies out nothing overlaps) is settled by trying every size of the last two
// chunks against the full play-through.
inline std::vector<size_t> plan_dp(size_t whole, size_t cap, double h, double k, double d, double gap) {
    std::vector<double> f(whole + 1, 0.0);
    std::vector<size_t> cut(whole + 1, 0);
    for (size_t a = 1; a <= whole; a++) {
        double best = 1e300; size_t arg = 1;
        for (size_t g = 1; g <= cap && g <= a; g++) {
            const double in_done = h * (double)a, prev = f[a - g];
            const double t = (prev > in_done ? prev : in_done) + k * (double)g + gap;
            if (t < best - 1e-9) { best = t; arg = g; }             // ties: the smaller chunk (its copy out starts earlier)
        }
        f[a] = best; cut[a] = arg;
    }
    auto prefix = [&](size_t a) { std::vector<size_t> r; while (a) { r.push_back(cut[a]); a -= cut[a]; } return std::vector<size_t>(r.rbegin(), r.rend()); };
    std::vector<size_t> best_plan = prefix(whole);
    double best_t = play(best_plan, h, k, d, gap);
    for (size_t l1 = 1; l1 <= cap && l1 <= whole; l1++)
        for (size_t l2 = 0; l2 <= cap && l1 + l2 <= whole; l2++) {
            std::vector<size_t> cand = prefix(whole - l1 - l2);
            if (l2) cand.push_back(l2);
            cand.push_back(l1);
            const double t = play(cand, h, k, d, gap);
            if (t < best_t - 1e-9 || (t < best_t + 1e-9 && cand.size() < best_plan.size())) { best_t = t; best_plan = cand; }
        }
    return best_plan;
}

// Sizes, in generations, of the chunks that cover `whole` generations.  `in_bytes`, `out_bytes`: bytes per element copied in / out; `unit`:
// elements per generation; `gens_fixed` > 0 forces the round 5 uniform shape (first and last chunk one generation, `gens_fixed` in between:
// the FOURQ_PIPE_GENS test hook).
inline std::vector<size_t> plan_generations(size_t whole, size_t unit, size_t in_bytes, size_t out_bytes, const Rates& r, int gens_fixed) {
    std::vector<size_t> sizes;
    if (whole == 0) return sizes;
    const size_t cap = gens_cap(unit, in_bytes, out_bytes);
    if (gens_fixed > 0) {
        const size_t g = (size_t)gens_fixed;
        sizes.push_back(1);
        size_t inner = whole >= 2 ? whole - 2 : 0;
        while (inner) { const size_t s = inner < g ? inner : g; sizes.push_back(s); inner -= s; }
        if (whole >= 2) sizes.push_back(1);
        return sizes;
    }
    const double li = r.link_in > 0 ? r.link_in : LINK_BYTES_PER_NS, lo = r.link_out > 0 ? r.link_out : LINK_BYTES_PER_NS;
    const double rin = in_bytes ? SAFETY * r.ns_per_elem * li / (double)in_bytes : 1e9;
    const double rout = out_bytes ? SAFETY * r.ns_per_elem * lo / (double)out_bytes : 1e9;
    // nanoseconds per generation; the copies priced with a margin (rates are measured, but the link is shared with the other direction)
    const double h = (double)unit * (double)in_bytes / li / MODEL_MARGIN, d = (double)unit * (double)out_bytes / lo / MODEL_MARGIN, k = (double)unit * r.ns_per_elem;
    std::vector<size_t> ones(whole, 1), greedy = plan_greedy(whole, cap, rin, rout), dp = plan_dp(whole, cap, h, k, d, GAP_NS);
    const double t_ones = play(ones, h, k, d, GAP_NS), t_greedy = play(greedy, h, k, d, GAP_NS), t_dp = play(dp, h, k, d, GAP_NS);
    if (t_dp <= t_greedy && t_dp <= t_ones) return dp;
    return t_greedy <= t_ones ? greedy : ones;
}

// The pieces (offset, length in elements) of a call of n > unit elements: the generations above, then the tail of less than one.
inline std::vector<Piece> plan_pieces(size_t n, size_t unit, size_t in_bytes, size_t out_bytes, const Rates& r, int gens_fixed) {
    std::vector<Piece> plan;
    const size_t whole = n / unit, tail = n - whole * unit;
    size_t off = 0;
    for (size_t g : plan_generations(whole, unit, in_bytes, out_bytes, r, gens_fixed)) { plan.push_back({ off, g * unit }); off += g * unit; }
    if (tail) plan.push_back({ off, tail });
    return plan;
}
inline std::vector<Piece> plan_pieces(size_t n, size_t unit, size_t in_bytes, size_t out_bytes, double ns_per_elem, int gens_fixed) {
    return plan_pieces(n, unit, in_bytes, out_bytes, Rates{ ns_per_elem, LINK_BYTES_PER_NS, LINK_BYTES_PER_NS }, gens_fixed);
}

}  // namespace fq_plan
