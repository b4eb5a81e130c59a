import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    return importlib.import_module("raytracer-go_amd")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("raytracer-go_amd.scenes")


@pytest.fixture(scope="session")
def bridge():
    from oracle import bridge as b
    b.build()
    return b


@pytest.fixture(scope="session")
def backend():
    return importlib.import_module("raytracer-go_amd.backend")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def c1(rt):
    """ExampleRegression(640, 380) — the scene of the reference's only test (raytracer_test.go:88-102)."""
    return rt.ExampleRegression(640, 380, GOLD)


@pytest.fixture(scope="session")
def golden_rgba(rt):
    with open(os.path.join(GOLD, "samples_images", "code_example.png"), "rb") as f:
        return rt.LoadRGBAImage(f).Pix  # the reference loads its golden through LoadRGBAImage too (raytracer_test.go:97)


def compare_images(img, exp):
    """compareImages, raytracer_test.go:32-80: a pixel differs if any of R,G,B is off by more than
    20*257 in 16-bit space (= 20/255) or alpha differs; returns (different, allowed)."""
    import numpy as np
    assert img.shape == exp.shape, "Generated image does not match expected image size"
    d = np.abs(img.astype(np.int64) * 257 - exp.astype(np.int64) * 257)
    diff = (d[..., :3] > 20 * 257).any(-1) | (img[..., 3] != exp[..., 3])
    return int(diff.sum()), int(img.shape[0] * img.shape[1] * 0.01)


class HipBuffers:
    """Device buffers for tests that hand DEVICE pointers to the C-ABI, through the HIP runtime the library itself
    is linked against (found among the process's mapped libraries once librtgo_hip.so is loaded).  torch is kept
    out of it: its wheel bundles a HIP runtime of its own, and two runtimes in one process do not both see the GPU
    when the library's comes first."""

    def __init__(self):
        import ctypes as C
        import importlib

        importlib.import_module("raytracer-go_amd.backend").lib()
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line and "/torch/" not in line:
                    path = line.split()[-1]
                    break
        assert path, "librtgo_hip.so did not bring a HIP runtime with it"
        self.C, self.hip = C, C.CDLL(path)
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._live = []

    def zeros(self, nbytes):
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), nbytes) == 0
        assert self.hip.hipMemset(p, 0, nbytes) == 0
        self._live.append(p)
        return p.value

    def to_host(self, ptr, shape, dtype):
        import numpy as np

        out = np.empty(shape, dtype=dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def close(self):
        for p in self._live:
            self.hip.hipFree(p)
        self._live = []


@pytest.fixture()
def hip_buffers():
    h = HipBuffers()
    yield h
    h.close()
