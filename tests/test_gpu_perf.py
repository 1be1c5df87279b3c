"""Wall-clock bounds, kept OUT of the parity gate: collected only by `pytest -m perf` on an MI355X
(tests/conftest.py deselects them under every other selection).  The numbers of record are the driver's
BENCH_r*.json and the rocprofv3 summaries under profiles/; these bounds are tripwires an order of magnitude out of
the way of clock-speed differences between boxes (C2: 6.9 us on one box, 8.6 on the driver's)."""
import pytest

from test_gpu_bench import run_bench

pytestmark = pytest.mark.perf


def test_c2_pass_takes_microseconds_not_tens_of_them():
    """25 us per pass of two launches in round 2; 6.2-8.6 us measured since round 3 (one launch)."""
    out, _ = run_bench("--workload", "C2", "--steps", "400", "--warmup", "20", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic")
    assert out["ms_per_step"] <= 0.020, out["ms_per_step"]
    assert out["ms_per_step"] <= out["roofline"]["kernel_ms_between_events"] + 0.005


def test_graded_planes_cost_less_than_two_sweeps_and_exact_unweighted_half_of_the_weighted_walk():
    out, _ = run_bench("--steps", "3", "--warmup", "1", "--secondary-steps", "2", "--no-cpu-baseline", "--no-live-traffic")
    sec = out["secondary"]
    assert sec[1]["ms_per_step"] < sec[5]["ms_per_step"] < 2.0 * sec[1]["ms_per_step"]   # (two sweeps cost 1.85 x)
    assert sec[6]["ms_per_step"] < 0.7 * sec[0]["ms_per_step"]       # pair_exact_unw_kernel 10 ms, pair_exact64_skip_kernel 21


def test_round_4_figures_have_not_gone_backwards():
    """Tripwires a fifth out of the way of the figures of record (profiles/r04_bench_default.json): the headline's
    fraction of the vector-ALU roofline (0.87), weighted EXACT64 at C3 (20.8-21.5 ms; 30.1 before
    pair_exact64_skip_kernel), the exact unweighted kernel (9.9-10.0 ms; 30 before it existed), C4 and C5 on one GPU."""
    out, _ = run_bench("--steps", "20", "--warmup", "3", "--secondary-steps", "3", "--no-cpu-baseline", "--no-live-traffic")
    sec = out["secondary"]
    assert out["roofline"]["frac"] >= 0.80, out["roofline"]["frac"]
    assert out["reference_width"]["ms_per_step"] <= 25.0, out["reference_width"]["ms_per_step"]
    assert sec[6]["ms_per_step"] <= 12.0, sec[6]["ms_per_step"]
    assert sec[3]["roofline"]["frac"] >= 0.80 and sec[4]["roofline"]["frac"] >= 0.80
