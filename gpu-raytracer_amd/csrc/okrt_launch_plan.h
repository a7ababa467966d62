// okrt_launch_plan.h — how many frames the next launch set of a progressive accumulation traces, and whether it rides the
// pipeline slots (okrt_context.hip: renderBatched / the deferred-render reservation). Plain arithmetic, no HIP: the same
// functions answer okrt_debug_launch_plan, which the CPU tests call (tests/test_launch_plan.py).
#pragma once
#include <algorithm>
#include <cstdint>

namespace okrt {

struct LaunchPlanParams {
    uint32_t tileRows, width;       // the pixels this context owns
    uint32_t batchLimit;            // frames per launch set the allocator has granted so far (<= kMaxFramesPerLaunch)
    uint32_t pipelineMaxFrames;     // OKRT_OPT_PIPELINE: sets up to this many frames take turns on the queue slots as they are
    uint32_t pipelineDepth;         // slots in use
    uint32_t splitParts;            // sub-sets a small-tile set is issued as
    uint64_t splitRays;             // ... when it traces at most this many camera rays
    bool pipelineOn, batchable;     // pipelining possible at all (library-owned stream + framebuffers, fast traversal) / frames may share launches
};

// A launch set of n frames is issued in one piece (1) or as that many equal sub-sets that ride the pipeline slots. Sub-sets
// pay for themselves where a set's fixed cost — the last dependent chains of each kernel on an emptied GPU, ~0.35 ms on the
// Sponza-class scene — is a large share of its time: one rank's stripes of an 8-way split, 20 frames: 1.67 -> 1.58 ms;
// on a full 1080p frame the two persistent grids cost each other more than the overlap returns (+4.5 %).
inline uint32_t launchSubSets(const LaunchPlanParams& p, uint32_t n) {
    if (!p.pipelineOn || !p.batchable || p.splitParts < 2) return 1;
    if (n <= p.pipelineMaxFrames || n < 2 * p.splitParts) return 1;  // (small sets take turns on the slots as they are)
    if ((uint64_t)p.tileRows * p.width * n > p.splitRays) return 1;
    return std::min(p.splitParts, p.pipelineDepth);
}

struct LaunchCursor {
    uint32_t setLeft = 0, subSize = 0;  // a set being issued in sub-sets: frames of it still to go, frames per sub-set
};
struct LaunchStep {
    uint32_t frames;
    bool subSet;  // part of a larger set: rides the slots whatever its size
};

inline LaunchStep nextLaunch(const LaunchPlanParams& p, LaunchCursor& cur, uint32_t framesLeft) {
    if (cur.setLeft) return {std::min(cur.setLeft, cur.subSize), true};
    uint32_t n = p.batchable ? std::min(framesLeft, p.batchLimit) : 1u;
    const uint32_t parts = launchSubSets(p, n);
    if (parts > 1) {
        cur.subSize = (n + parts - 1) / parts;
        cur.setLeft = n;
        n = cur.subSize;
    }
    return {n, cur.setLeft != 0};
}
inline void launched(LaunchCursor& cur, uint32_t n) {
    if (cur.setLeft) cur.setLeft -= std::min(n, cur.setLeft);
}
// does a step ride the pipeline slots (its own stream, frame slots, ordered accumulation)?
inline bool ridesSlots(const LaunchPlanParams& p, const LaunchStep& s) { return p.pipelineOn && p.batchable && (s.subSet || s.frames <= p.pipelineMaxFrames); }

}  // namespace okrt
