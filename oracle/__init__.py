"""CPU oracle for the RADAD segment -> embed -> retrieve hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`radad_retrievalaugmenteddeepfakeaudiodetection_amd/`) imports this package; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the checker /
the reported baseline, never as the thing shipped.

Parity status (see DESIGN.md "Oracle"):
  * segmenter, temporal pyramid pooling, segment mean, projection layer, RADAD model shell: PINNED by
    golden vectors produced by importing the reference's own modules (tests/golden/make_golden.py).
  * zero-mean/unit-variance and log-mel front-ends: PINNED by golden vectors from the HuggingFace
    feature extractors the reference calls (default constructors, transformers 5.15.0).
  * brute-force kNN: PARITY UNPINNED by the reference -- its arithmetic lives in faiss, which is neither in
    /root/reference nor installed.  The oracle restates faiss IndexFlat's published semantics
    (squared L2 ascending / inner product descending, int64 ids in insertion order, -1 fill) in
    float64 with (distance, index) lexicographic order.
"""
