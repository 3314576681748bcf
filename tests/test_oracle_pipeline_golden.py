"""CPU: the oracle's a4 / a7 / K7 restatements against tests/golden/pipeline.npz -- outputs of the reference's OWN
DeepfakeDetectionPipeline.process_audio_batch / .retrieve_similar_vectors (pipeline.py:392-414, :449-532) and
VectorDatabase._maybe_normalize / .add_vectors / .search_batch (vector_database.py:100-188), run by tests/golden/make_golden.py."""
import os
import sys

import numpy as np
import pytest

from oracle import radad_oracle as O
from oracle import synth


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "pipeline.npz"))


@pytest.fixture(scope="module")
def mk(golden_dir):
    """the generator module (stdlib + numpy at import time): its stand-in encoder a4_extract is shared with the fixture"""
    if golden_dir not in sys.path:
        sys.path.insert(0, golden_dir)
    import make_golden
    return make_golden


def a4_inputs(g, mk):
    w = synth.rows(0, mk.A4_HOP, mk.A4_F, int(g["a4_w_seed"])) * np.float32(0.05)
    waves = [synth.audio(i, 1, int(n), int(g["a4_audio_seed"]))[0] for i, n in enumerate(g["a4_lengths"])]
    return w, waves


@pytest.mark.parametrize("mode", ["max", "avg"])
def test_process_audio_batch_composition_matches_reference(g, mk, mode):
    w, waves = a4_inputs(g, mk)
    seg, hop = O.segment_lengths()
    out = O.process_audio_batch(waves, seg, hop, lambda s: mk.a4_extract(s, w), tuple(g["a4_levels"]), mode)
    ref = g[f"a4_{mode}_out"]
    assert out.shape == ref.shape == (len(waves), 7 * mk.A4_F)
    # max pooling is exact; the mean over <= 3 float32 segment vectors is a float32 sum in the reference, float64 here
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-7)
    assert bool(g["a4_none_raises_runtime_error"])
    with pytest.raises(RuntimeError):
        O.process_audio_batch([None], seg, hop, lambda s: s)


def a7_inputs(g):
    K, D, N = int(g["a7_K"]), int(g["a7_D"]), int(g["a7_N"])
    stored = synth.rows(0, N, D, int(g["a7_stored_seed"]))
    return K, D, N, stored, [str(p) for p in g["a7_paths"]], [float(x) for x in g["a7_labels"]]


A7_CASES = {
    # name: (lists handed back by the search, keyword arguments, k the reference asked the search for)
    "self_paths": ("15", dict(exclude_self=True, use_paths=True), 15),
    "self_train_ids": ("15", dict(exclude_self=True, use_paths=False), 15),
    "no_exclusion": ("5", dict(exclude_self=False, use_paths=True), 5),
}


@pytest.mark.parametrize("name", sorted(A7_CASES))
def test_retrieve_postprocess_matches_reference(g, name):
    K, D, N, stored, paths, labels = a7_inputs(g)
    which, kw, k_search = A7_CASES[name]
    dists, idxs = g["a7_d" + which], g["a7_i" + which]
    np.testing.assert_array_equal(g[f"a7_{name}_search_k"], [k_search])          # K + (10 if exclude_self else 0), pipeline.py:478
    assert str(g[f"a7_{name}_query_dtype"][0]) == "<f4"                            # queries reach search_batch as float32 (:456)
    qp = [str(p) for p in g["a7_query_paths"]] if kw["use_paths"] else None
    v, l, p, d = O.retrieve_postprocess(dists, idxs, stored, paths, labels, K, D, query_paths=qp, exclude_self=kw["exclude_self"],
                                        training_file_ids=set(str(x) for x in g["a7_training_file_ids"]))
    np.testing.assert_array_equal(v, g[f"a7_{name}_vec"])
    np.testing.assert_array_equal(l, g[f"a7_{name}_lbl"])
    np.testing.assert_array_equal(d, g[f"a7_{name}_dist"])                         # NaN padding compares equal here
    assert [list(r) for r in p] == [[str(x) for x in r] for r in g[f"a7_{name}_paths"]]
    np.testing.assert_array_equal(g[f"a7_{name}_arities"], [2, 3, 3, 1])           # the four return arities carry the same tensors


def test_unfilled_slots_wrap_in_the_reference(g):
    """pipeline.py:495: an id of -1 (faiss's unfilled slot) indexes vector_paths[-1] and reconstruct(-1) -- the fixture shows the
    reference returning the LAST stored row for such slots.  The oracle reproduces that to the letter (skip_unfilled=False); this
    build skips such slots instead (skip_unfilled=True; HotPathPipeline.retrieve_similar_vectors, tests/test_gpu_reference_fixtures.py)."""
    K, D, N, stored, paths, labels = a7_inputs(g)
    row = 3                                                                        # every id of row 3 is -1
    assert (g["a7_i15"][row] == -1).all()
    ref_paths = [str(x) for x in g["a7_self_paths_paths"][row]]
    assert ref_paths == [paths[-1]] * K                                            # wrapped, K times
    np.testing.assert_array_equal(g["a7_self_paths_vec"][row], np.stack([stored[-1]] * K))
    v, l, p, d = O.retrieve_postprocess(g["a7_d15"], g["a7_i15"], stored, paths, labels, K, D,
                                        query_paths=[str(x) for x in g["a7_query_paths"]], skip_unfilled=True)
    assert p[row] == [""] * K and np.isnan(d[row]).all() and not v[row].any()


def test_swallowed_search_failure_and_empty_index(g):
    K, D, N, stored, paths, labels = a7_inputs(g)
    B = int(g["a7_B"])
    # search_batch raising is swallowed (pipeline.py:479-483): every row is padding
    v, l, p, d = O.retrieve_postprocess(np.zeros((B, 0), np.float32), np.zeros((B, 0), np.int64), stored, paths, labels, K, D,
                                        query_paths=[str(x) for x in g["a7_query_paths"]])
    np.testing.assert_array_equal(v, g["a7_search_raises_vec"])
    np.testing.assert_array_equal(l, g["a7_search_raises_lbl"])
    np.testing.assert_array_equal(d, g["a7_search_raises_dist"])
    assert [list(r) for r in p] == [[str(x) for x in r] for r in g["a7_search_raises_paths"]]
    # an index that holds nothing (pipeline.py:465-476): the search is not called at all
    assert g["a7_empty_index_search_k"].size == 0
    v, l, p, d = O.retrieve_postprocess(np.zeros((B, 0), np.float32), np.zeros((B, 0), np.int64), stored, paths, labels, K, D,
                                        index_ntotal=0)
    np.testing.assert_array_equal(v, g["a7_empty_index_vec"])
    np.testing.assert_array_equal(l, g["a7_empty_index_lbl"])
    np.testing.assert_array_equal(d, g["a7_empty_index_dist"])


@pytest.mark.parametrize("cosine", [0, 1])
def test_maybe_normalize_and_index_shell_match_reference(g, cosine):
    rows = g["k7_rows"]
    ref = g[f"k7_norm_{cosine}"]
    assert str(g[f"k7_norm_{cosine}_dtype"]) == "float32"
    out = O.maybe_normalize(rows, bool(cosine))
    np.testing.assert_allclose(out, ref, rtol=3e-7, atol=1e-30)                    # float32 division in the reference, float64 here
    if not cosine:
        np.testing.assert_array_equal(ref, rows)
    else:
        assert not ref[5].any()                                                    # the zero row stays zero: 0 / (0 + 1e-12)
    # what reached index.add: the normalised rows, in batches of vector_add_batch_size (vector_database.py:118-119, 132-138)
    np.testing.assert_array_equal(g[f"shell_{cosine}_added"], ref)
    np.testing.assert_array_equal(g[f"shell_{cosine}_batch_sizes"], [10, 10, 3])
    assert bool(g[f"shell_{cosine}_paths_ok"])
    # what reached index.search (vector_database.py:163-169)
    q = g["k7_query"]
    for (k_req, k_passed, n_rows, n_cols), qq in zip(g[f"shell_{cosine}_k"], (q, q, q, q, q[0])):
        qo, ko = O.search_batch_shell(23, qq, None if k_req < 0 else int(k_req), top_k=5, cosine=bool(cosine))
        assert (ko, qo.shape[0]) == (k_passed, n_rows) and n_cols == k_passed
    np.testing.assert_allclose(O.search_batch_shell(23, q, 15, cosine=bool(cosine))[0], g[f"shell_{cosine}_q_passed"], rtol=3e-7)
    np.testing.assert_allclose(O.search_batch_shell(23, q[0], 3, cosine=bool(cosine))[0], g[f"shell_{cosine}_q1d_passed"], rtol=3e-7)
    # nothing stored: ([B, 0] float32, [B, 0] int64) and index.search is never called (:169-172); no index at all: ValueError (:160-161)
    np.testing.assert_array_equal(g[f"shell_{cosine}_empty"], [4, 0, 4, 0, 1, 1, 0])
    assert O.search_batch_shell(0, q, 5)[1] == 0
    assert bool(g[f"shell_{cosine}_none_raises"])
