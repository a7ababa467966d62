"""Shared fixtures. `-m "not gpu"` runs on a machine without a GPU (oracle, host logic, ABI
surface); `-m gpu` tests are the parity tests proper and call the HIP path through the C ABI."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

abi = importlib.import_module("gpu-raytracer_amd._abi")
host = importlib.import_module("gpu-raytracer_amd.host")
scenes = importlib.import_module("gpu-raytracer_amd.scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def host_threads():
    """Threads worth starting for the CPU checker: CPU count cut down to the affinity mask and the cgroup quota (a
    one-GPU box shows 256 CPUs and grants its job a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    return n


class Oracle:
    """The CPU oracle (oracle/liboracle.so): test infrastructure only."""

    def __init__(self):
        path = os.environ.get("OKRT_ORACLE_LIBRARY") or os.path.join(REPO, "oracle", "liboracle.so")  # (override: `make asan`)
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True)
        self.lib = C.CDLL(path)
        self.fns = abi.bind(self.lib, "okrt_oracle_")
        self.lib.okrt_oracle_get_api.restype = C.c_void_p
        self.api = self.lib.okrt_oracle_get_api()
        self.lib.okrt_oracle_create.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.okrt_oracle_render_rows.argtypes = [C.c_void_p, C.POINTER(abi.RenderData), C.c_uint32, C.c_uint32]
        self.lib.okrt_oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
        self.lib.okrt_oracle_read_counter_images.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]

    def context(self):
        h = C.c_void_p()
        assert self.lib.okrt_oracle_create(C.byref(h)) == 0
        ctx = host.Context(fns=self.fns, handle=h)
        ctx._own = True
        return ctx

    def instantiate(self, desc, w, h):
        return scenes.instantiate(desc, w, h, api=self.api, fns=self.fns)

    def counter_images(self, ctx):
        n = ctx.tile_rows() * ctx.W
        bb = np.zeros(n, np.uint32)
        tri = np.zeros(n, np.uint32)
        self.lib.okrt_oracle_read_counter_images(ctx.h, bb.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p))
        return bb.reshape(-1, ctx.W), tri.reshape(-1, ctx.W)


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def device():
    """The HIP product; fails (not skips) if libokrt.so is missing."""
    return abi.load_device_library()


def compare_accum(gpu, cpu, frames=1):
    """Per-pixel RGB L2 of acc/N between two float accumulations; returns (frac_within_1e3, rmse, nan_mismatch)."""
    g = gpu[..., :3].astype(np.float64) / frames
    c = cpu[..., :3].astype(np.float64) / frames
    fin = np.isfinite(g).all(-1) & np.isfinite(c).all(-1)
    nan_mismatch = float((np.isfinite(g).all(-1) != np.isfinite(c).all(-1)).mean())
    d = np.where(fin[..., None], g - c, 0.0)
    l2 = np.sqrt((d * d).sum(-1))
    frac = float((l2[fin] <= 1e-3).mean()) if fin.any() else 1.0
    rmse = float(np.sqrt((d * d).mean()))
    return frac, rmse, nan_mismatch, l2
