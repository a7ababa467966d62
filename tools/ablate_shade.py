"""Where the shading kernels' time goes (developer tool): Sponza-class scene with and without texture taps / lights."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
W, H = 1920, 1080
INV = 0xFFFFFFFF
for label in ("full", "no textures", "no normal maps", "albedo only"):
    desc = scenes.sponza_proxy()
    for e in desc.entities:
        m = e.material
        if label == "no textures":
            for k in ("albedoTextureId", "roughnessTextureId", "metallicTextureId", "specularTextureId", "normalMapIdx"): m[k] = INV
        elif label == "no normal maps":
            m["normalMapIdx"] = INV
        elif label == "albedo only":
            for k in ("roughnessTextureId", "metallicTextureId", "specularTextureId", "normalMapIdx"): m[k] = INV
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.render(); c = rt.ctx.counters(); rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    rt.renderFrames(32)
    rt.ctx.set_option(abi.OPT_TIMING, 8)
    rt.renderFrames(128)
    tr, sh, ot, n = rt.ctx.stage_ms()
    print(f"{label:16s} tex taps/ray {c['tex_taps']/c['segments']:.2f}  trace {tr/n:.4f}  shade {sh/n:.4f}  other {ot/n:.4f} ms/frame ({n} frames)", flush=True)
    rt.close()
