"""bench.py's bookkeeping that decides whether a committed PMC measurement may be attached to a bench line
(roofline.traffic), and the roofline object's invariants — on CPU, with synthetic inputs."""
import json
import os
import sys
import types

import pytest

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_kernel_source_fingerprint_is_stable_and_sensitive(tmp_path, monkeypatch):
    a = bench.kernel_source_sha()
    assert a == bench.kernel_source_sha() and len(a) == 16
    # a different tree (an extra device source) gives a different fingerprint
    fake = tmp_path / "gpu-raytracer_amd" / "csrc"
    fake.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (fake / "k.hip").write_text("__global__ void k() {}\n")
    (tmp_path / "include" / "okrt.h").write_text("/* */\n")
    (tmp_path / "gpu-raytracer_amd" / "build.py").write_text("\n")
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    b = bench.kernel_source_sha()
    (fake / "k.hip").write_text("__global__ void k() { }\n")
    assert b != a and bench.kernel_source_sha() != b


def test_stale_or_mismatched_pmc_files_are_not_attached(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "feedfacefeedface")
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    got, why = bench.load_pmc("sponza", 20)
    assert got is None and "no profiles" in why
    good = {"workload": "sponza", "frames_per_launch": 20, "source_sha": "feedfacefeedface", "kernels": {"traceKernel": {"hbm_bytes_per_launch": 1.0}}}
    json.dump(dict(good, source_sha="0000000000000000"), open(tmp_path / "profiles" / "r01_pmc_sponza_f20.json", "w"))
    got, why = bench.load_pmc("sponza", 20)
    assert got is None and "other kernel sources" in why            # measured before the kernels changed
    json.dump(dict(good, frames_per_launch=64), open(tmp_path / "profiles" / "r02_pmc_sponza_f64.json", "w"))
    got, why = bench.load_pmc("sponza", 20)
    assert got is None and "launch sets of 64" in why                # right kernels, other launch-set size
    json.dump(good, open(tmp_path / "profiles" / "r03_pmc_sponza_f20.json", "w"))
    got, why = bench.load_pmc("sponza", 20)
    assert why is None and got[0] == "r03_pmc_sponza_f20.json" and got[1]["kernels"]["traceKernel"]["hbm_bytes_per_launch"] == 1.0
    assert bench.load_pmc("dragon", 20)[0] is None                   # another workload's file is never used


def test_roofline_report_is_a_fraction_and_names_its_basis(monkeypatch):
    monkeypatch.setattr(bench, "load_pmc", lambda w, f: (None, "test"))
    args = types.SimpleNamespace(traversal="fast", max_bounces=1, workload="sponza")
    px, rays = 2_073_600, 4_099_000
    # one Sponza-class 1080p frame: 32 400 camera tiles x (18.9 node steps + 7.7 triangle tests) x 64 lanes, 2.03 M bounce
    # rays x (15.85 + 4.45) lane-level steps
    fast = dict(pixels=px, segments=rays, fast_nodes=39_100_000 + 32_200_000, fast_tris=16_000_000 + 9_000_000, mesh_hits=4_000_000, tex_taps=4_100_000, env_taps=90_000)
    cam = dict(fast, segments=px, fast_nodes=39_100_000, fast_tris=16_000_000, mesh_hits=2_030_000, tex_taps=2_080_000, env_taps=43_000)
    ref = dict(segments=rays)
    build = dict(num_nodes=65_500, num_triangles=262_267)
    rep = bench.roofline_report(args, 1, False, [12.7, 0.01, 17.5, 6.1], 1, 64, build, ref, fast, cam, 117.9e9, 1e9, 0.47, 0.095, 0.0)
    assert rep["kernel"].startswith("traceKernel") and rep["frames_per_launch"] == 64 and rep["traffic"] is None and rep["traffic_stale"]
    assert 0 < rep["frac"] <= 1 and "algorithmic" in rep["frac_basis"]
    for k in rep["kernels"]:
        assert 0 < k["algorithmic_frac"] <= 1 and k["l2_frac"] is None and k["l1_gather_bytes_per_launch"] > 0, k  # L2 figures are measured or absent
    assert rep["reference_work_ratio"] > 1  # the reference's traversal order would touch far more bytes than HBM could deliver
    # with a matching measurement the fraction is the measured one
    pmc = {"kernels": {"traceKernel": {"hbm_bytes_per_launch": 17.1e9, "l2_hit_pct": 84.5, "valu_lane_util": 0.565, "l2_read_requests_per_launch": 893e6}}}
    monkeypatch.setattr(bench, "load_pmc", lambda w, f: (("r99_pmc_sponza_f64.json", pmc), None))
    rep = bench.roofline_report(args, 1, False, [12.7, 0.01, 17.5, 6.1], 1, 64, build, ref, fast, cam, 117.9e9, 1e9, 0.47, 0.095, 0.0)
    assert rep["traffic"] == int(17.1e9) and not rep["traffic_stale"] and "measured" in rep["frac_basis"]
    assert abs(rep["frac"] - 17.1e9 / 17.5e-3 / 8e12) < 1e-3 and rep["valu_lane_util"] == 0.565
    # measured L1 -> L2 read requests (128-byte lines): as a share of the L2 peak and of the rate a pure gather sustains
    assert abs(rep["l2_frac"] - 893e6 * 128 / 17.5e-3 / 34.5e12) < 1e-3 and abs(rep["gather_ceiling_frac"] - 893e6 / 17.5e-3 / bench.gather_ceiling(0.845)) < 1e-3
    assert bench.gather_ceiling(1.0) == bench.GATHER_L2_HIT_REQUESTS_PER_S and abs(bench.gather_ceiling(0.0) - bench.GATHER_L2_MISS_REQUESTS_PER_S) < 1
    # round 5: the last-segment shortcut changes the kernels' names and algorithmic bytes (hit-or-miss record: 32 B per ray + 4 per
    # miss), the PMC file's issue fractions travel into the line and name what binds each kernel
    pmc = {"kernels": {"traceKernel": {"hbm_bytes_per_launch": 8.0e9, "l2_hit_pct": 88.9, "valu_lane_util": 0.568, "l2_read_requests_per_launch": 890e6,
                                       "valu_issue_frac": 1.19, "salu_issue_frac": 0.52, "wait_any_share": 0.46, "wait_inst_share": 0.25},
                       "missKernel": {"hbm_bytes_per_launch": 6.5e9, "l2_hit_pct": 19.4, "valu_lane_util": 0.33, "valu_issue_frac": 0.36}}}
    monkeypatch.setattr(bench, "load_pmc", lambda w, f: (("r99_pmc_sponza_f64.json", pmc), None))
    rep = bench.roofline_report(args, 1, False, [12.7, 0.01, 17.5, 1.6], 1, 64, dict(build, flags=4), ref, fast, cam, 117.9e9, 1e9, 0.47, 0.095, 0.0, True)
    names = [k["kernel"].split(" ")[0] for k in rep["kernels"]]
    assert names == ["tracePacketKernel<FUSED>", "traceKernel<MODE", "missKernel"], names
    tk = rep["kernels"][1]
    bounce = rays - px
    assert tk["algorithmic_bytes_per_launch"] == int((32 * bounce + 4 * (90_000 - 43_000)) * 64 + 64 * 65_500 + 48 * 262_267)
    assert tk["bound"] == "valu-issue" and tk["valu_issue_frac"] == 1.19 and tk["wait_mem_share"] == 0.46 and rep["binding_resource"] == "valu-issue"
    assert rep["kernels"][2]["bound"].startswith("fabric") and rep["bound"] == "hbm" and rep["camera_segment"].startswith("packets of 64 rays")
    rep = bench.roofline_report(args, 1, False, [12.7, 0.01, 17.5, 1.6], 1, 64, dict(build, flags=4 | 8), ref, fast, cam, 117.9e9, 1e9, 0.47, 0.095, 0.0, True)
    assert rep["kernels"][0]["kernel"].startswith("traceKernelCamera") and rep["camera_segment"] == "per ray (probe)"


def test_pmc_tool_sorts_kernel_names_into_families():
    """tools/pmc_traffic.py attributes rocprofv3 rows to kernel families by name: the counting variants are left out, the per-ray camera
    kernel (traceKernel MODE 0 beside a MODE 1 launch) is kept apart from the bounce kernel, missKernel is a family of its own."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(REPO, "tools", "pmc_traffic.py"))
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    ns = "void okrt::fastdetail::(anonymous namespace)::"
    assert pmc.family(ns + "traceKernel<false, 1, >(okrt::fastdetail::FastView, ...)")[0] == "traceKernel"
    assert pmc.family(ns + "traceKernel<false, 1>(okrt::fastdetail::FastView, ...)") == ("traceKernel", False)
    assert pmc.family(ns + "traceKernel<false, 0>(okrt::fastdetail::FastView, ...)") == ("traceKernel0", False)
    assert pmc.family(ns + "traceKernel<true, 0>(okrt::fastdetail::FastView, ...)") == ("traceKernel0", True)
    assert pmc.family(ns + "tracePacketKernel<false, true>(HIP_vector_type<unsigned int, 4u> const*, ...)") == ("tracePacketKernel", False)
    assert pmc.family(ns + "tracePacketKernel<true, false>(HIP_vector_type<unsigned int, 4u> const*, ...)") == ("tracePacketKernel", True)
    assert pmc.family(ns + "shadeKernel<false, true>(okrt::SceneView, ...)") == ("shadeKernel", True)
    assert pmc.family("okrt::fastdetail::(anonymous namespace)::missKernel(okrt::SceneView, ...)") == ("missKernel", False)
    assert pmc.family("okrt::fastdetail::(anonymous namespace)::orderTilesKernel(unsigned int const*, ...)") == (None, False)
    assert pmc.family("__amd_rocclr_fillBufferAligned") == (None, False)
    assert [n for n, _ in pmc.PASSES] == ["stats", "fetch", "write", "sq", "l1", "sq2", "sq3"]
    assert all(len([c for c in flags if c.startswith("SQ_")]) <= 8 for _, flags in pmc.PASSES)  # 8 SQ slots per pass (MI355X_MICROARCH.md)
