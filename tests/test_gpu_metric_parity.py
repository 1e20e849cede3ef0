"""North_star's "+-0.001 on HR@50 / NDCG@50" as far as it can be checked without the Netflix files: one initialisation and
one batch stream trained by the fp64 restatement of the reference, the HIP float32 path and the HIP bf16 path, then ranked on
a held-out set (tests/metric_parity.py)."""
import pytest

pytestmark = pytest.mark.gpu


def test_trained_ranking_metrics_match_the_reference_arithmetic():
    from tests.metric_parity import run
    res = run()
    ref = res["ref"]
    assert ref["loss_last"] < ref["loss_first"] - 0.05, "the proxy task did not train"
    assert 0.05 < ref["H50"] < 0.98, ref               # metrics away from the trivial ends
    for k in ("H50", "N50", "H10", "N10", "H100", "N100"):
        assert abs(res["f32"]["delta_vs_ref"][k]) <= 1e-3, (k, res["f32"], ref)     # the stated +-0.001
    # bf16 activations: reported (profiles/r02_metric_parity.json holds a run); bounded here at 0.002 absolute (measured: <= 0.0005)
    for k in ("H50", "N50"):
        assert abs(res["bf16"]["delta_vs_ref"][k]) <= 2e-3, (k, res["bf16"], ref)
    print({m: {k: round(v, 5) for k, v in res[m].items() if k != "delta_vs_ref"} for m in ("ref", "f32", "bf16")})
