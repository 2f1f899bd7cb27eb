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


_FLIPS = []


def flip_ok(qa, qb, site, bound=1e-3, max_step=1):
    """INT4 digits (or fake-quant values) of two routes that share the arithmetic up to rounding noise: the fraction that differs is
    RECORDED (profiles/r06_flip_rates.txt is the session's list, written to gpurun_out/flip_rates.txt on the GPU box) and bounded by
    SURVEY 7's 1e-3 — `bound` is larger only where the site says why (a population of a few rows, where one row with a one-ulp scale
    difference is already 1e-3 of the digits)."""
    qa, qb = np.asarray(qa), np.asarray(qb)
    rate = float(np.mean(qa != qb))
    _FLIPS.append((site, int(qa.size), rate, bound))
    ok = rate <= bound
    if ok and max_step is not None and qa.dtype.kind in "iu":
        ok = int(np.max(np.abs(qa.astype(np.int32) - qb.astype(np.int32)))) <= max_step
    return ok


def pytest_sessionfinish(session, exitstatus):
    if not _FLIPS:
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "flip_rates.txt"), "w") as fh:
            fh.write("# site | digits compared | fraction that differs | bound asserted\n")
            for site, n, rate, bound in _FLIPS:
                fh.write(f"{site:70s} {n:10d} {rate:.3e} {bound:.1e}\n")
    except OSError:
        pass


# The small-sample sites (14 - 300 rows) that asserted 2e-3 through round 5. Measured in round 6 (profiles/r06_flip_rates.txt): the largest is
# 7.9e-4 (hadamard_quant n = 6144, 37 rows), every Kronecker / RMSNorm site is below 3.1e-4 — so they all assert SURVEY 7's 1e-3 now; the
# >= 1e6-digit populations of tests/test_gpu_flip_rates.py hold the same routes to it on populations where one row cannot move the figure.
BOUND37 = 1e-3
