"""CPU-side checks: the C-ABI library builds/loads and exports every declared symbol, the host logic
(masker, shard bounds, model construction) is right, and the product path refuses to run without a GPU."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from easydgl_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "easydgl_hip.h")).read()
    declared = set(re.findall(r"\b(edgl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.edgl_version() >= 100
    assert _lib.lib.edgl_score_chunks(10240, 20001) >= 1
    assert _lib.lib.edgl_bimau_pack_bytes(128, 8, 16, _lib.BF16) > 0


def test_abi_argument_validation_without_gpu():
    from easydgl_amd import _lib
    # null pointers / bad shapes are rejected before any launch
    rc = _lib.lib.edgl_gemm(None, None, None, 4, 4, 4, 4, 4, 4, 1, 1, None, None, 0, 1, None, 0, None)
    assert rc == -4 and b"null" in _lib.lib.edgl_last_error()
    rc = _lib.lib.edgl_bimau_pack(None, None, None, None, 128, 8, 16, None, 0, None)
    assert rc == -4
    assert _lib.lib.edgl_bimau_bwd_workspace(4, 11, 32, 3, 4, 0) == -1   # C % H != 0
    # the attention entry points of the TGAT / TiSASRec path: nulls, head dims, interval table limits
    L, one = _lib.lib, 1   # `one`: any non-null address — the checks below fail before a pointer is dereferenced
    rc = L.edgl_tattn_fwd(None, 48, None, 48, None, 16, None, 16, None, 2, 8, 1, 48, 16, 0.25, 0.0, None, 0, None, 16, None, 1, 0, None)
    assert rc == -4 and b"null" in L.edgl_last_error()
    rc = L.edgl_tattn_fwd(one, 40, one, 40, one, 16, one, 16, one, 2, 8, 1, 40, 16, 0.25, 0.0, None, 0, one, 16, None, 1, 0, None)
    assert rc == -1 and b"multiples of 16" in L.edgl_last_error()                      # Dq = 40
    rc = L.edgl_tattn_fwd(one, 48, one, 48, one, 16, one, 16, one, 2, 8, 1, 48, 16, 0.25, 0.5, None, 0, one, 16, None, 1, 0, None)
    assert rc == -4 and b"rng_state" in L.edgl_last_error()                            # dropout without a generator state
    rc = L.edgl_tattn_fwd(one, 48, one, 48, one, 16, one, 16, one, 2, 8, 1, 48, 16, 0.25, 0.0, None, 0, one, 16, None, 1, 7, None)
    assert rc == -2                                                                    # dtype code 7
    rc = L.edgl_tiattn_fwd(one, 32, one, 64, one, 64, one, 32, one, one, one, one, 300, 2, 8, 2, 16, 0.25, 1.0, 300, 0.0, None, 0,
                           one, 32, None, None, 1, 0, None)
    assert rc == -1 and b"timelen" in L.edgl_last_error()                              # timelen > 256
    assert L.edgl_tattn_saved_bytes(2, 8, 1, 16) > 0 and L.edgl_tattn_saved_bytes(0, 8, 1, 16) == -1
    assert L.edgl_tiattn_bucket_elems(2, 8, 2, 256) == 2 * 8 * 2 * 272


def test_ops_refuse_cpu_tensors():
    from easydgl_amd import _lib, ops
    with pytest.raises(_lib.EdglError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4), 4, 4, 4, 4, 4, True, True, torch.float32)


def test_model_constructs_with_reference_flags_and_names():
    import easydgl_amd
    F = SimpleNamespace(model="EasyDGL", num_items=50, num_units=32, num_heads=2, num_blocks=2, seqslen=10, masklen=3,
                        time_scale=86400.0, learning_rate=1e-3, l2_reg=1e-4, ct_reg=1e-3, hidden_dropout_rate=0.1,
                        attention_probs_dropout_rate=0.1, mark_table=O.synthetic_mark_table(50, 4), compute_dtype="f32",
                        num_train_steps=None, num_warmup_steps=None)
    m = easydgl_amd.ranking(F)
    assert m.mask == 50 and m.num_items == 51 and m.seqslen == 11          # EasyDGL.py:39-41
    cfg = O.Config(num_items=50, seqslen=10, num_units=32, num_heads=2, num_blocks=2, num_events=4)
    want = O.init_params(cfg, np.random.default_rng(0))
    got = m.tf_variable_map()
    assert set(got) == set(want)
    for k in want:
        assert tuple(got[k].shape) == want[k].shape, k
    with pytest.raises(NotImplementedError):
        easydgl_amd.ranking(SimpleNamespace(model="GRU4REC"))
    # initialisers (temporal.py:393: N(0, 0.02); biases zero; LayerNorm gamma one)
    assert abs(float(got["layer_0/attention/self/TMAU/dense/kernel"].std()) - 0.02) < 0.002
    assert float(got["CSTMA/output_bias"].abs().max()) == 0.0
    assert float(got["layer_0/output/LayerNorm/gamma"].min()) == 1.0


def test_masker_semantics():
    from easydgl_amd import data as D
    g = torch.Generator().manual_seed(0)
    mp = D.draw_masked_positions(64, 21, 5, generator=g)
    assert mp.shape == (64, 5) and int(mp.min()) >= 1 and int(mp.max()) <= 20      # dataloader.py:34-36
    assert all(len(set(r.tolist())) == 5 for r in mp)
    assert len({tuple(sorted(r.tolist())) for r in mp}) > 32
    ids, ts = D.synthetic_batch(100, 20, 64, seed=1)
    cfg = O.Config(num_items=100, seqslen=20, num_units=8)
    f, lab = D.mask_random(torch.tensor(ids), torch.tensor(ts), 100, mp)
    fo, labo = O.mask_random(cfg, ids, ts, mp.numpy())
    np.testing.assert_array_equal(f["seqs_i"].numpy(), fo["seqs_i"])
    np.testing.assert_array_equal(lab.numpy(), labo)
    f, lab = D.mask_last(torch.tensor(ids), torch.tensor(ts), 100)
    fo, labo = O.mask_last(cfg, ids, ts)
    np.testing.assert_array_equal(f["seqs_i"].numpy(), fo["seqs_i"])
    np.testing.assert_array_equal(lab.numpy(), labo)
    # the synthetic generators agree with the oracle's (same seed, same stream)
    np.testing.assert_array_equal(D.synthetic_mark_table(30, 4, True), O.synthetic_mark_table(30, 4, True))
    assert ids.min() == 0 and ids.max() < 100 and (np.diff(ts, axis=1)[ids[:, 1:] > 0] >= 0).all()


def test_shard_bounds_cover_the_table():
    from easydgl_amd.parallel import shard_bounds
    for n, w in [(20001, 8), (1001, 3), (7, 8)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(lo % 8 == 0 for lo, hi in b if hi > lo)      # empty shards may start at num_rows
        if n > 8 * w:
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) < 16


def test_graft_entry_build():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(os.path.join(ROOT, "easydgl_amd", "libeasydgl_hip.so"))


def test_bench_flop_model_matches_the_survey_numbers():
    """179.9 MFLOP per sequence forward at the headline configuration (M = 20), 108.2 at M = 6; x3 for fwd+bwd."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f20 = bench.flops_per_seq(bench.HEADLINE)
    f6 = bench.flops_per_seq(dict(bench.HEADLINE, masklen=6))
    assert abs(f20 / 1e6 - 179.9) < 0.2 and abs(f6 / 1e6 - 108.2) < 0.2
    assert abs(3 * f20 * 512 / 1e9 - 276.3) < 0.5


def test_key_mask_forms_accepted_by_the_attention_units():
    """BiMAU / MAU keep the reference's positional signature, whose `masks` is the [h*B, T, T] float key mask
    (EasyDGL.py:94-95); the kernels take a [B, T] int64 vector.  The module reduces the reference forms and rejects anything else."""
    import pytest
    import torch
    from easydgl_amd.module.temporal import key_ids_from_masks
    B, T, h = 3, 5, 2
    ids = torch.tensor([[0, 0, 4, 9, 2], [0, 7, 7, 1, 3], [5, 6, 1, 2, 8]])
    want = (ids != 0).to(torch.int64)
    assert key_ids_from_masks(ids, B, T, h) is not None and torch.equal(key_ids_from_masks(ids, B, T, h), ids)
    ref_mask = (ids != 0).float().unsqueeze(1).repeat(h, T, 1)            # tf.tile(expand_dims(.., 1), [h, T, 1])
    assert ref_mask.shape == (h * B, T, T)
    for m in (ref_mask, ref_mask[:B], (ids != 0).float().unsqueeze(1), (ids != 0).float(), (ids != 0)):
        got = key_ids_from_masks(m, B, T, h)
        assert got.dtype == torch.int64 and got.shape == (B, T) and torch.equal(got, want)
    for bad in (torch.zeros(B, T + 1), torch.zeros(h * B, T), torch.zeros(B, 2, T), torch.zeros(4, 4, 4, 4)):
        with pytest.raises(ValueError):
            key_ids_from_masks(bad, B, T, h)
    with pytest.raises(TypeError):
        key_ids_from_masks([[1, 2]], B, T, h)
