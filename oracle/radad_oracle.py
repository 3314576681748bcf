"""CPU restatement of the RADAD hot path (segment -> embed -> retrieve).  TEST INFRASTRUCTURE ONLY.

Every function cites the reference lines it follows (paths relative to the RADAD repository, or
`transformers/...` for the HuggingFace front-ends the reference calls).  Arithmetic is float64 unless
the reference's own dtype behaviour is the thing being restated.
"""
import os
import numpy as np

# ------------------------------------------------------------------------------------------------
# a1  segmenter.py:8-39
# ------------------------------------------------------------------------------------------------

def segment_lengths(sample_rate=16000, segment_length_s=2.0, overlap=0.5):
    """segmenter.py:11-13 -- (segment_length, hop_length) in samples."""
    seg = int(segment_length_s * sample_rate)
    hop = int(seg * (1 - overlap))
    return seg, hop


def segment_count(n_samples, seg, hop):
    """segmenter.py:25."""
    return max(1, (n_samples - seg) // hop + 1)


def segment_audio(audio, seg, hop):
    """segmenter.py:15-39.  Returns the list of segments exactly as the reference does: views of the
    input when no padding is needed, float64 arrays (np.zeros default dtype) when the clip is shorter
    than one segment; samples past the last full window are dropped."""
    audio = np.asarray(audio)
    if audio.ndim > 1:
        raise ValueError("Expected 1D audio array")
    total = len(audio)
    out = []
    for i in range(segment_count(total, seg, hop)):
        start = i * hop
        end = min(start + seg, total)
        s = audio[start:end]
        if len(s) < seg:
            s = np.concatenate([s, np.zeros(seg - len(s))])
        out.append(s)
    return out


def segment_plan(clip_lengths, seg, hop):
    """CSR form of the same rule for a batch: (seg_clip, seg_start_in_clip, seg_valid, clip_seg_offsets)."""
    seg_clip, seg_start, seg_valid, offs = [], [], [], [0]
    for b, n in enumerate(clip_lengths):
        ns = segment_count(int(n), seg, hop)
        for i in range(ns):
            st = i * hop
            seg_clip.append(b)
            seg_start.append(st)
            seg_valid.append(max(0, min(seg, int(n) - st)))
        offs.append(offs[-1] + ns)
    return (np.asarray(seg_clip, np.int64), np.asarray(seg_start, np.int64),
            np.asarray(seg_valid, np.int32), np.asarray(offs, np.int64))


# ------------------------------------------------------------------------------------------------
# a2  front-ends the reference's extractors call
# ------------------------------------------------------------------------------------------------

def zero_mean_unit_var(x):
    """feature_extractor.py:25-30 -> transformers/models/wav2vec2/feature_extraction_wav2vec2.py:95
    (no attention mask: the segments are already equal length): (x - mean) / sqrt(var + 1e-7)."""
    x = np.asarray(x, np.float64)
    return (x - x.mean(axis=-1, keepdims=True)) / np.sqrt(x.var(axis=-1, keepdims=True) + 1e-7)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)


def mel_filter_bank(n_bins=201, n_mels=80, fmin=0.0, fmax=8000.0, sr=16000):
    """transformers/audio_utils.py:638-729 with norm='slaney', mel_scale='slaney' (the arguments
    transformers/models/whisper/feature_extraction_whisper.py:94-103 passes).  float64 [n_bins, n_mels]."""
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    f_pts = _mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    diff = np.diff(f_pts)
    slopes = f_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb


def hann_periodic(n):
    """torch.hann_window(n) (periodic=True), used at feature_extraction_whisper.py:141."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def log_mel(x, padded_samples=0, n_fft=400, hop=160, n_mels=80, mel_filters=None):
    """transformers/models/whisper/feature_extraction_whisper.py:135-168 for ONE waveform.
    x: [L] samples.  padded_samples: 0 -> spectrogram of x itself; else x is zero-padded to that
    length first (HF pads to 480000, feature_extraction_whisper.py:200,303).
    torch.stft(center=True, pad_mode='reflect'): reflect-pad n_fft//2 both sides, 1+P//hop frames,
    the last one dropped (:151).  Returns float64 [frames, n_mels] (HF returns its transpose)."""
    x = np.asarray(x, np.float64)
    if padded_samples and padded_samples > len(x):
        x = np.concatenate([x, np.zeros(padded_samples - len(x))])
    if mel_filters is None:
        mel_filters = mel_filter_bank(n_fft // 2 + 1, n_mels)
    half = n_fft // 2
    xp = np.pad(x, (half, half), mode="reflect")
    n_frames = 1 + len(x) // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = xp[idx] * hann_periodic(n_fft)[None, :]
    spec = np.fft.rfft(frames, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:-1]          # drop last frame (:151)
    mel = power @ np.asarray(mel_filters, np.float64)        # (:154) mel_filters.T @ magnitudes, transposed
    logs = np.log10(np.maximum(mel, 1e-10))                  # (:156)
    logs = np.maximum(logs, logs.max() - 8.0)                # (:161)
    return (logs + 4.0) / 4.0                                # (:162)


def frame_projection(feats, w, b):
    """The stage that stands where the reference runs its pretrained encoder (feature_extractor.py:33,
    :110, :167): a dense projection of each log-mel frame, [T, M] @ [M, F] + b."""
    return np.asarray(feats, np.float64) @ np.asarray(w, np.float64) + np.asarray(b, np.float64)[None, :]


# ------------------------------------------------------------------------------------------------
# a3  pooling.py:66-103, 119-122
# ------------------------------------------------------------------------------------------------

def adaptive_bins(T, level):
    """torch adaptive_{max,avg}_pool1d bin rule used by pooling.py:76,79: bin i = [floor(i*T/l), ceil((i+1)*T/l))."""
    return [((i * T) // level, -((-(i + 1) * T) // level)) for i in range(level)]


def tpp(features, levels=(1, 2, 4), mode="max"):
    """pooling.py:88-103: per level adaptive pooling over time, flattened bin-major / feature-minor
    (:84), levels concatenated in order (:103).  features [T, F] -> [sum(levels)*F]."""
    f = np.asarray(features)
    T = f.shape[0]
    outs = []
    for l in levels:
        for lo, hi in adaptive_bins(T, l):
            if mode == "max":
                outs.append(f[lo:hi].max(axis=0))
            elif mode == "avg":
                outs.append(f[lo:hi].astype(np.float64).mean(axis=0))
            else:
                raise ValueError(f"Unsupported pooling type: {mode}")
    return np.concatenate(outs)


def tpp_output_dim(levels, feature_dim):
    """pooling.py:119-122."""
    return sum(levels) * feature_dim


def segment_mean(seg_vectors):
    """pipeline.py:411: mean over the clip's segment vectors."""
    return np.mean(np.stack([np.asarray(v, np.float64) for v in seg_vectors]), axis=0)


def process_audio_batch(waves, seg, hop, extract_fn, levels=(1, 2, 4), mode="max"):
    """pipeline.py:392-414 with the extractor as a parameter (the reference's is a pretrained transformer): per clip
    segment_audio (:396-400) -> extract_fn(segment) -> [T, F] per segment (:402-406) -> tpp per segment (:410) -> mean over the
    clip's segments (:411) -> stacked [B, D] (:414).  A clip that loads as None raises RuntimeError (:398-399).
    Pinned by tests/golden/pipeline.npz (the reference's own method run with its AudioSegmenter and TemporalPyramidPooling)."""
    out = []
    for wav in waves:
        if wav is None:
            raise RuntimeError("Failed to load clip")
        vecs = [tpp(extract_fn(s), levels, mode) for s in segment_audio(np.asarray(wav), seg, hop)]
        out.append(segment_mean(vecs))
    return np.stack(out)


def embed_clips(waves, seg, hop, w, b, levels=(1,), mode="max", normalize=True, padded_samples=0,
                mel_filters=None):
    """process_audio_batch with this build's extractor (normalise -> log-mel -> frame projection): list of 1-D clips -> float64 [B, D]."""
    def extract(s):
        s = zero_mean_unit_var(s) if normalize else np.asarray(s, np.float64)
        return frame_projection(log_mel(s, padded_samples=padded_samples, mel_filters=mel_filters), w, b)
    return process_audio_batch(waves, seg, hop, extract, levels, mode)


# ------------------------------------------------------------------------------------------------
# a5/a6  vector_database.py:56-105, 159-188  (faiss IndexFlat semantics; see oracle/__init__.py)
# ------------------------------------------------------------------------------------------------

def maybe_normalize(arr, cosine):
    """vector_database.py:100-105."""
    arr = np.asarray(arr, np.float64)
    if cosine:
        arr = arr / (np.linalg.norm(arr, axis=1, keepdims=True) + 1e-12)
    return arr


def knn(db, q, k, metric="L2", chunk=8192):
    """Brute-force top-k in float64 with (distance, index) lexicographic order.
    metric 'L2': squared Euclidean ascending; 'IP': inner product descending; 'COSINE': both sides
    normalised first (vector_database.py:97,118,166), then IP.
    k is clamped to ntotal (vector_database.py:169); k<=0 -> empty (:170-172).
    Returns (dist float64 [nq,k], idx int64 [nq,k])."""
    metric = metric.upper()
    db = np.asarray(db, np.float64)
    q = np.asarray(q, np.float64)
    if q.ndim == 1:
        q = q.reshape(1, -1)                                     # vector_database.py:164-165
    if metric == "COSINE":
        db, q = maybe_normalize(db, True), maybe_normalize(q, True)
    n = db.shape[0]
    k = min(int(k), n)
    if k <= 0:
        return np.zeros((len(q), 0)), np.zeros((len(q), 0), np.int64)
    best_d = np.full((len(q), 0), 0.0)
    best_i = np.zeros((len(q), 0), np.int64)
    for s in range(0, n, chunk):
        blk = db[s:s + chunk]
        if metric == "L2":
            d = ((q[:, None, :] - blk[None, :, :]) ** 2).sum(-1) if blk.shape[0] * len(q) * db.shape[1] < 2e7 \
                else (q * q).sum(1)[:, None] + (blk * blk).sum(1)[None, :] - 2.0 * (q @ blk.T)
            key = d
        else:
            d = q @ blk.T
            key = -d
        ids = np.arange(s, s + blk.shape[0], dtype=np.int64)[None, :].repeat(len(q), 0)
        cat_key = np.concatenate([(-best_d if metric != "L2" else best_d), key], axis=1)
        cat_d = np.concatenate([best_d, d], axis=1)
        cat_i = np.concatenate([best_i, ids], axis=1)
        order = np.lexsort((cat_i, cat_key), axis=1)[:, :k]     # primary key: distance, then index
        best_d = np.take_along_axis(cat_d, order, 1)
        best_i = np.take_along_axis(cat_i, order, 1)
    return best_d, best_i


def knn_exact_l2_chunked(db, q, k, chunk=4096):
    """Same as knn(..., 'L2') but always with the direct sum((q-y)^2) form (no cancellation)."""
    db = np.asarray(db, np.float64)
    q = np.asarray(q, np.float64)
    k = min(int(k), db.shape[0])
    bd = np.zeros((len(q), 0))
    bi = np.zeros((len(q), 0), np.int64)
    for s in range(0, db.shape[0], chunk):
        blk = db[s:s + chunk]
        d = np.stack([((blk - qq[None, :]) ** 2).sum(1) for qq in q])
        ids = np.arange(s, s + blk.shape[0], dtype=np.int64)[None, :].repeat(len(q), 0)
        cd, ci = np.concatenate([bd, d], 1), np.concatenate([bi, ids], 1)
        order = np.lexsort((ci, cd), axis=1)[:, :k]
        bd, bi = np.take_along_axis(cd, order, 1), np.take_along_axis(ci, order, 1)
    return bd, bi


def ivf_search(db, assign, centroids, q, k, nprobe):
    """What faiss.IndexIVFFlat(IndexFlatL2, METRIC_L2).search returns for given centroids and list assignments
    (vector_database.py:65-70,174-181): the exact top-k by squared L2 among the rows whose list is one of the `nprobe`
    centroids nearest to the query; fewer candidates than k -> id -1 / +inf.  float64, (distance, id) order.
    faiss is not in the repository (requirements.txt:7,10): restated from its published behaviour, parity unpinned."""
    db = np.asarray(db, np.float64)
    q = np.asarray(q, np.float64)
    _, probes = knn(centroids, q, min(nprobe, len(centroids)), "L2")
    D = np.full((len(q), k), np.inf)
    I = np.full((len(q), k), -1, np.int64)
    assign = np.asarray(assign)
    for i in range(len(q)):
        cand = np.flatnonzero(np.isin(assign, probes[i]))
        if len(cand) == 0:
            continue
        d = ((db[cand] - q[i][None, :]) ** 2).sum(1)
        order = np.lexsort((cand, d))[:k]
        D[i, :len(order)] = d[order]
        I[i, :len(order)] = cand[order]
    return D, I


def rank_gaps(dist_sorted):
    """Minimum gap between consecutive ranks per query (used to prove 'bit-exact indices' is testable)."""
    d = np.asarray(dist_sorted, np.float64)
    return np.abs(np.diff(d, axis=1)).min(axis=1) if d.shape[1] > 1 else np.full(len(d), np.inf)


def merge_topk(dists, idxs, k, metric="L2"):
    """Merge P partial lists [P, nq, k'] -> [nq, k] by (distance, index); idx -1 sorts last."""
    d = np.concatenate(list(dists), axis=1).astype(np.float64)
    i = np.concatenate(list(idxs), axis=1).astype(np.int64)
    key = d.copy() if metric.upper() == "L2" else -d
    key = np.where(i < 0, np.inf, key)
    order = np.lexsort((i, key), axis=1)[:, :k]
    return np.take_along_axis(d, order, 1), np.take_along_axis(i, order, 1)


# ------------------------------------------------------------------------------------------------
# a7  pipeline.py:449-532 (retrieve_similar_vectors post-processing)
# ------------------------------------------------------------------------------------------------

def retrieve_postprocess(dists, idxs, stored_vectors, vector_paths, vector_labels, K, D, query_paths=None,
                         exclude_self=True, training_file_ids=(), index_ntotal=None, skip_unfilled=False):
    """pipeline.py:449-532 after the search: empty store -> zeros / 0 / '' / NaN (:465-476); per row keep the first K hits whose
    basename is not excluded (:491-509), fetch the STORED vector (:503), pad to K with zeros / 0.0 / '' / NaN (:511-515).
    dists / idxs: what search_batch returned for k_search = K + (10 if exclude_self else 0) (:478); None or a malformed result is
    the swallowed failure (:481-487).  Returns (vec [B,K,D] f32, lbl [B,K] f32, paths, dist [B,K] f32).
    skip_unfilled=False follows the reference to the letter: an unfilled slot (id -1) indexes vector_paths[-1] (:495, Python
    negative indexing) and reconstructs row -1; True is this build's documented deviation (such slots are skipped).
    Pinned by tests/golden/pipeline.npz (the reference's own method, run on seeded search results)."""
    B = len(idxs) if idxs is not None else 0
    if index_ntotal is not None and index_ntotal == 0:
        B = len(dists) if idxs is None else B
        return (np.zeros((B, K, D), np.float32), np.zeros((B, K), np.float32), [[""] * K for _ in range(B)],
                np.full((B, K), np.nan, np.float32))
    exclude_ids = set()
    if exclude_self and query_paths is not None:
        exclude_ids = {os.path.basename(p) for p in query_paths}         # :463
    all_v, all_l, all_p, all_d = [], [], [], []
    for row_inds, row_d in zip(idxs, dists):
        cv, cl, cp, cd = [], [], [], []
        for ii, dd in zip(row_inds, row_d):
            ii = int(ii)
            if ii < 0 and skip_unfilled:
                continue
            fname = os.path.basename(vector_paths[ii])                   # :495 (negative ii wraps, as in the reference)
            if exclude_self:
                if query_paths is not None:
                    if fname in exclude_ids:
                        continue
                elif fname in training_file_ids:
                    continue
            cv.append(np.asarray(stored_vectors[ii], np.float32))
            cl.append(vector_labels[ii])
            cp.append(vector_paths[ii])
            cd.append(float(dd))
            if len(cv) == K:
                break
        while len(cv) < K:
            cv.append(np.zeros(D, np.float32)); cl.append(0.0); cp.append(""); cd.append(float("nan"))
        all_v.append(cv); all_l.append(cl); all_p.append(cp); all_d.append(cd)
    return (np.stack([np.stack(v) for v in all_v]).astype(np.float32), np.asarray(all_l, np.float32),
            all_p, np.asarray(all_d, np.float32))


def search_batch_shell(ntotal, query, k=None, top_k=5, cosine=False):
    """vector_database.py:159-182 around index.search: k defaults to config.top_k (:163), a 1-D query becomes [1, D] (:164-165),
    queries are normalised when the index is cosine (:166), k is clamped to ntotal (:169), k <= 0 returns ([B, 0] f32, [B, 0] i64)
    without calling the index (:170-172).  Returns (q as handed to index.search, k handed over or 0)."""
    k = int(k if k is not None else top_k)
    q = np.asarray(query, np.float32)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    q = maybe_normalize(q, cosine)
    return q, max(0, min(k, int(ntotal)))


# ------------------------------------------------------------------------------------------------
# a8  projection.py:68-106 (+ radad_model.py:38-39)
# ------------------------------------------------------------------------------------------------

def projection_forward(x, p):
    """ProjectionLayer eval forward.  x [B,K,D]; p: dict of torch-layout weights
    (attention_score.weight [H,D] ...).  float64."""
    f = lambda a: np.asarray(a, np.float64)
    x = f(x)
    h = np.tanh(x @ f(p["attention_score.weight"]).T + f(p["attention_score.bias"]))        # :69-70
    s = h @ f(p["attention_final.weight"]).T + f(p["attention_final.bias"])                 # :71  [B,K,1]
    c = np.maximum(x @ f(p["cst_hidden.weight"]).T + f(p["cst_hidden.bias"]), 0.0)           # :74-75
    c = c @ f(p["cst_output.weight"]).T + f(p["cst_output.bias"])                            # :76  [B,K,D]
    s = s - s.max(axis=1, keepdims=True)
    a = np.exp(s) / np.exp(s).sum(axis=1, keepdims=True)                                     # :87
    u = (a * c).sum(axis=1)                                                                  # :88-89
    y = u @ f(p["weight_sum.weight"]).T + f(p["weight_sum.bias"])                            # :94
    mu = y.mean(axis=1, keepdims=True)
    var = ((y - mu) ** 2).mean(axis=1, keepdims=True)
    y = (y - mu) / np.sqrt(var + 1e-6) * f(p["normalization.weight"]) + f(p["normalization.bias"])  # :99, eps :51
    return y @ f(p["unified_embedding.weight"]).T + f(p["unified_embedding.bias"])           # :101


def radad_model_forward(neighbor_vecs, tpp_vecs, p, bn_eps=1e-5):
    """RADADModel eval forward (radad_model.py:32-41) with the reference's state_dict names; float64.
    Returns (proj [B,P], fused [B,P], logits [B])."""
    f = lambda a: np.asarray(a, np.float64)
    proj = projection_forward(neighbor_vecs, {k[len("projection_layer."):]: v for k, v in p.items()
                                              if k.startswith("projection_layer.")})               # :38
    fused = np.concatenate([f(tpp_vecs), proj], axis=1) @ f(p["fuse.weight"]).T + f(p["fuse.bias"])  # :39
    # detection_model.py:45-72: nn.Sequential indices in order; Linear, then (BatchNorm1d eval, ReLU) except last
    idx = sorted({int(k.split(".")[2]) for k in p if k.startswith("detection_model.model.")})
    lin = [i for i in idx if np.ndim(p[f"detection_model.model.{i}.weight"]) == 2]
    h = fused
    for n, i in enumerate(lin):
        h = h @ f(p[f"detection_model.model.{i}.weight"]).T + f(p[f"detection_model.model.{i}.bias"])
        if n + 1 < len(lin):
            b = i + 1
            if f"detection_model.model.{b}.running_mean" in p:
                h = (h - f(p[f"detection_model.model.{b}.running_mean"])) / np.sqrt(
                    f(p[f"detection_model.model.{b}.running_var"]) + bn_eps)
                h = h * f(p[f"detection_model.model.{b}.weight"]) + f(p[f"detection_model.model.{b}.bias"])
            h = np.maximum(h, 0.0)
    return proj, fused, (h[:, 0] if h.shape[1] == 1 else h)                                         # squeeze(-1) :125
