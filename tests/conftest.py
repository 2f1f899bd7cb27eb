import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The shared library is a build artefact (git-ignored). Build it once if this checkout does not have it yet and a
    # hipcc is around (it cross-compiles gfx950 without a GPU); the tests themselves never fall back to anything else.
    lib = os.path.join(ROOT, "flatquant_amd", "lib", "libfqhip.so")
    if not os.path.exists(lib):
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
            subprocess.run(["make", "-C", os.path.join(ROOT, "flatquant_amd", "csrc"), "-j8"], check=True, env=env)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` with no -m."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def hadk_matrix(K):
    """Sign matrix of the K x K Hadamard factor shipped in flatquant_amd/data/hadk.npz."""
    z = np.load(os.path.join(ROOT, "flatquant_amd", "data", "hadk.npz"))
    bits = np.unpackbits(z[f"had{K}"])[: K * K].reshape(K, K)
    return (bits.astype(np.float32) * 2 - 1).astype(np.float16)


def same_bits(a, b):
    """fp16 arrays equal BIT FOR BIT (np.array_equal would let -0.0 pass for +0.0): fake-quant outputs are compared this
    way — the reference's round_ste never returns -0.0 and neither may the kernels (rule 10)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype == np.float16 and np.array_equal(a.view(np.uint16), b.view(np.uint16))
