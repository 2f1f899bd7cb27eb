"""A kernel whose time is bimodal is not done (two tables of round 4 held 4x outliers, 144x192 at 792 us where
three other runs had 204-207). Every Kronecker kernel family runs its BASELINE / benchmark-list shape 200 times through the C ABI with
pre-allocated outputs, one HIP-event pair per launch; the run fails when the slowest launch exceeds twice the median. One retry
budget: a host hiccup between two launches (the event pair brackets the gap in front of a launch as well) does not repeat three
times in a row, a slow mode of the kernel (a meeting that spins, a starved claim) does."""
import os
import statistics
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

CASES = [  # (M, N, tokens, kernel family)
    (64, 64, 16384, "fq_kron64_kernel (C2 headline)"),
    (64, 128, 16384, "fq_kron_wave_kernel (Llama-2-70B hidden)"),
    (112, 128, 16384, "fq_kron_trio_kernel (Llama-3-8B ffn)"),
    (128, 224, 8192, "fq_kron_duo_kernel (Llama-2-70B ffn)"),
    (172, 64, 16384, "fq_kron_tall_kernel (Hadamard 11008 as a Kronecker pair)"),
    (80, 112, 16384, "fq_kron_tiles_kernel 4 groups x 4 waves"),
    (128, 144, 8192, "fq_kron_tiles_kernel 2 x 5"),
    (144, 192, 8192, "fq_kron_tiles_kernel 2 x 6, R streamed (792 us outlier of r04_final_shapes_table.txt)"),
    (168, 176, 8192, "fq_kron_tiles_kernel 1 x 6 (1117 us outlier of the mid-round table)"),
]


@pytest.mark.parametrize("M,N,rows,family", CASES, ids=[f"{m}x{n}" for m, n, _, _ in CASES])
def test_no_launch_beyond_twice_the_median(M, N, rows, family):
    import time_dist
    launch = time_dist.prepare(M, N, rows)
    worst = []
    for attempt in range(3):
        us = time_dist.distribution(launch, 200, warm=30)
        med = statistics.median(us)
        worst.append((max(us), med))
        if max(us) <= 2.0 * med:
            return
    pytest.fail(f"{family}: slowest / median launch of three runs of 200: " + ", ".join(f"{a:.1f} / {b:.1f} us" for a, b in worst))
