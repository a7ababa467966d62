#include "okrt_fast.h"
namespace okrt {
struct FastScene {};
bool fastBuild(FastScene**, const FastInputs&, hipStream_t, std::string& err) { err = "not built yet"; return false; }
void fastDestroy(FastScene*) {}
void launchFast(const FastScene&, const SceneView&, const okrt_render_data&, const TileView&, float4*, uint32_t*, unsigned long long*, hipStream_t) {}
}
