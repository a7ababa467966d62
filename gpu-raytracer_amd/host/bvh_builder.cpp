// bvh_builder.cpp — see bvh_builder.h. Follows reference source/Graphics/BvhBuilder.cpp:12-223.
#include "bvh_builder.h"

#include <stack>

namespace okh {

namespace {
constexpr uint32_t NUM_TESTS = 100;  // BvhBuilder.cpp:74

inline float sahCost(uint32_t leftCount, const AABB& leftBox, uint32_t rightCount, const AABB& rightBox) {
    // BvhBuilder.cpp:68-69. An empty side has area +inf and count 0 -> NaN -> FLT_MAX.
    float cost = leftCount * leftBox.getArea() + rightCount * rightBox.getArea();
    return cost > 0.f ? cost : FLT_MAX;
}
inline void merge(AABB& a, const AABB& b) {
    a.min = vmin(b.min, a.min);
    a.max = vmax(b.max, a.max);
}
}  // namespace

void BvhBuilder::buildTree(const Triangle* tris, uint32_t numTris, const AABB& rootBox) {
    m_nodes.clear();
    m_tris = tris;

    m_nodes.emplace_back();
    BvhNode& root = m_nodes.back();
    root.boundingBox = rootBox;  // BvhBuilder.cpp:23
    root.triIndicies.resize(numTris);
    for (uint32_t i = 0; i < numTris; i++) root.triIndicies[i] = i;

    // BvhBuilder.cpp:113-119 (OkayMath::getMiddle, SMath.h:9-12)
    m_triMiddles.resize(numTris);
    m_triBoxes.resize(numTris);
    for (uint32_t i = 0; i < numTris; i++) {
        const Triangle& t = tris[i];
        m_triMiddles[i] = (t.position[0] + t.position[1] + t.position[2]) * (1.f / 3.f);
        AABB b;
        b.growTo(t.position[0]);
        b.growTo(t.position[1]);
        b.growTo(t.position[2]);
        m_triBoxes[i] = b;
    }

    struct NodeStack {
        uint32_t nodeIndex, depth;
    };
    std::stack<NodeStack> stack;
    stack.push({0u, 0u});
    std::vector<uint32_t> leftIndicies, rightIndicies;

    while (!stack.empty()) {  // BvhBuilder.cpp:136-204
        NodeStack nodeData = stack.top();
        stack.pop();
        uint32_t nodeNumTris = (uint32_t)m_nodes[nodeData.nodeIndex].triIndicies.size();
        if (nodeData.depth >= m_maxDepth - 1u || nodeNumTris <= m_maxLeafTriangles) continue;  // :145

        uint32_t axis = 0;
        float splitPos = 0.f;
        float splitCost = findBestSplitPlane(m_nodes[nodeData.nodeIndex], axis, splitPos);
        float parentCost = nodeNumTris * m_nodes[nodeData.nodeIndex].boundingBox.getArea();
        if (splitCost >= parentCost) continue;  // :150

        leftIndicies.clear();
        rightIndicies.clear();
        {
            const std::vector<uint32_t>& idx = m_nodes[nodeData.nodeIndex].triIndicies;
            for (uint32_t i = 0; i < nodeNumTris; i++) {
                if (m_triMiddles[idx[i]][axis] < splitPos)
                    leftIndicies.push_back(idx[i]);
                else
                    rightIndicies.push_back(idx[i]);
            }
        }
        if (leftIndicies.empty() || rightIndicies.empty()) continue;  // :175

        m_nodes.emplace_back();
        m_nodes.emplace_back();
        uint32_t first = (uint32_t)m_nodes.size() - 2u;
        BvhNode& cur = m_nodes[nodeData.nodeIndex];
        cur.firstChildIdx = first;
        std::vector<uint32_t>().swap(cur.triIndicies);  // interior nodes need no list (the reference keeps it)
        m_nodes[first].triIndicies = leftIndicies;
        m_nodes[first + 1].triIndicies = rightIndicies;
        findAABB(m_nodes[first]);
        findAABB(m_nodes[first + 1]);
        stack.push({first, nodeData.depth + 1u});       // :202
        stack.push({first + 1u, nodeData.depth + 1u});  // :203  -> right child is processed first
    }
    m_triMiddles.clear();
    m_triBoxes.clear();
}

float BvhBuilder::findBestSplitPlane(const BvhNode& node, uint32_t& outAxis, float& outSplitPos) {  // :72-97
    const std::vector<uint32_t>& idx = node.triIndicies;
    const uint32_t triCount = (uint32_t)idx.size();
    float bestCost = FLT_MAX;

    float pos[NUM_TESTS];  // pos[1..99]
    uint32_t binCount[NUM_TESTS + 1];
    AABB binBox[NUM_TESTS + 1];
    AABB suffixBox[NUM_TESTS + 2];
    uint32_t suffixCount[NUM_TESTS + 2];

    for (uint32_t axis = 0; axis < 3; axis++) {
        float boundsMin = node.boundingBox.min[(int)axis];
        float boundsMax = node.boundingBox.max[(int)axis];
        if (boundsMin == boundsMax) continue;

        bool monotone = true;
        for (uint32_t i = 1; i < NUM_TESTS; i++) {
            float a = i / (float)NUM_TESTS;
            pos[i] = boundsMin * (1.f - a) + boundsMax * a;  // glm::mix(x, y, a) = x*(1-a) + y*a
            if (i > 1 && !(pos[i - 1] <= pos[i])) monotone = false;
        }

        if (!monotone) {
            // Not sweepable: evaluate every plane against every triangle like the reference (:38-70).
            for (uint32_t i = 1; i < NUM_TESTS; i++) {
                AABB leftBox, rightBox;
                uint32_t leftCount = 0, rightCount = 0;
                for (uint32_t t = 0; t < triCount; t++) {
                    if (m_triMiddles[idx[t]][(int)axis] < pos[i]) {
                        leftCount++;
                        merge(leftBox, m_triBoxes[idx[t]]);
                    } else {
                        rightCount++;
                        merge(rightBox, m_triBoxes[idx[t]]);
                    }
                }
                float cost = sahCost(leftCount, leftBox, rightCount, rightBox);
                if (cost < bestCost) {
                    outSplitPos = pos[i];
                    outAxis = axis;
                    bestCost = cost;
                }
            }
            continue;
        }

        // bin k (1..100): triangles whose centroid is left of planes k..99 and right of 1..k-1.
        for (uint32_t k = 0; k <= NUM_TESTS; k++) {
            binCount[k] = 0;
            binBox[k] = AABB();
        }
        for (uint32_t t = 0; t < triCount; t++) {
            float c = m_triMiddles[idx[t]][(int)axis];
            uint32_t lo = 1, hi = NUM_TESTS;  // smallest i in [1,99] with c < pos[i], else 100
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (c < pos[mid])
                    hi = mid;
                else
                    lo = mid + 1;
            }
            binCount[lo]++;
            merge(binBox[lo], m_triBoxes[idx[t]]);
        }
        suffixBox[NUM_TESTS + 1] = AABB();
        suffixCount[NUM_TESTS + 1] = 0;
        for (uint32_t k = NUM_TESTS; k >= 1; k--) {
            suffixBox[k] = suffixBox[k + 1];
            merge(suffixBox[k], binBox[k]);
            suffixCount[k] = suffixCount[k + 1] + binCount[k];
        }
        AABB leftBox;
        uint32_t leftCount = 0;
        for (uint32_t i = 1; i < NUM_TESTS; i++) {
            merge(leftBox, binBox[i]);
            leftCount += binCount[i];
            float cost = sahCost(leftCount, leftBox, suffixCount[i + 1], suffixBox[i + 1]);
            if (cost < bestCost) {
                outSplitPos = pos[i];
                outAxis = axis;
                bestCost = cost;
            }
        }
    }
    return bestCost;
}

void BvhBuilder::findAABB(BvhNode& node) {  // :209-223
    for (uint32_t t : node.triIndicies) merge(node.boundingBox, m_triBoxes[t]);
}

}  // namespace okh
