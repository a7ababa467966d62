"""gpu-raytracer_amd — MI355X-native path-tracing core behind the GPU-Raytracer upload interface.

The package directory name contains a hyphen (it mirrors the reference's repository name), so import
it with `importlib.import_module("gpu-raytracer_amd")`; `tests/conftest.py`, `bench.py` and
`__graft_entry__.py` do exactly that.

  csrc/   HIP kernels + the C ABI of include/okrt.h            -> libokrt.so        (the product)
  host/   C++ mirror of the reference's RayTracer host driver  -> libokrt_host.so
  _abi.py / host.py   ctypes views of the two libraries (test + bench harness)
  scenes.py           seeded procedural stand-ins for the benchmark scenes
  build.py            the build recipe (__graft_entry__.build() runs it)
"""
from . import _abi  # noqa: F401
from ._abi import OkrtError, RenderData, load_device_library  # noqa: F401
