"""bench.py's roofline object (host logic, no GPU): the dominant kernel is credited the job's algorithmic bytes split over its
launches; for whole-buffer Huffman the one-lane heap kernel is reported as latency-bound and the longest
streaming kernel carries the roofline with its own bytes."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_dominant_kernel_gets_the_jobs_bytes_per_launch():
    b = _bench()
    n, c, steps, dt = 10**9, 516_875_560, 10, 0.458
    kt = [{"name": "k_lz2_find", "ms": 6.6, "launches": 40}, {"name": "k_lz_parse_emit", "ms": 5.5, "launches": 40}]
    r = b.roofline_of("deflate-h", n, c, kt, steps, dt)
    assert r["kernel"] == "k_lz2_find" and r["launches_per_step"] == 4.0
    assert r["algorithmic_bytes_per_launch"] == (n + c) // 4
    assert abs(r["achieved"] - (n + c) / 4 / 6.6e-3 / 1e9) < 0.01
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-5
    assert abs(r["whole_step_frac"] - (n + c) / (dt / steps) / 1e9 / 8000.0) < 1e-5
    assert "latency_bound" not in r


def test_huffman_heap_kernel_is_not_credited_the_jobs_bytes():
    b = _bench()
    n, c = 10**8, 56_000_000
    kt = [{"name": "k_huff_hist", "ms": 0.08, "launches": 10}, {"name": "k_huff_build", "ms": 0.18, "launches": 10},
          {"name": "k_huff_encode", "ms": 0.15, "launches": 10}]
    r = b.roofline_of("huffman", n, c, kt, 10, 0.0046)
    assert r["kernel"] == "k_huff_encode"
    assert r["algorithmic_bytes_per_launch"] == n + c
    assert abs(r["achieved"] - (n + c) / 0.15e-3 / 1e9) < 0.01
    assert r["latency_bound"]["kernel"] == "k_huff_build" and abs(r["latency_bound"]["ms_per_step"] - 0.18) < 1e-9
    # the whole step still counts both passes over the input
    assert abs(r["whole_step_frac"] - (2 * n + c) / 0.00046 / 1e9 / 8000.0) < 1e-4


def test_no_kernel_times_no_roofline():
    assert _bench().roofline_of("deflate", 1, 1, [], 1, 1.0) is None
