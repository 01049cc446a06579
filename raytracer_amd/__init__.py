"""raytracer_amd: MI355X (gfx950) implementation of the LewisBray/RayTracer per-pixel render loop.

The product is raytracer_amd/lib/librt_amd.so (HIP kernels + C ABI, include/rt_amd.h);
`raytracer_amd.api` is its ctypes binding for the tests and the benchmark.
"""
from . import abi  # noqa: F401
