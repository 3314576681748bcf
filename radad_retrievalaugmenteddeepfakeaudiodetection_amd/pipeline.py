"""Hot-path shell: the two pipeline methods whose loops this build collapses, with the reference's signatures
(pipeline.py:392-414 process_audio_batch, :449-532 retrieve_similar_vectors).  Training, metrics, plots, wandb
and the CLI stay in the reference; this class only wires segment -> embed -> retrieve.
"""
import os
from typing import List, Optional, Sequence

import numpy as np

from .feature_extractor import build_feature_extractor
from .pooling import TemporalPyramidPooling
from .segmenter import AudioSegmenter
from .vector_database import VectorDatabase


class HotPathPipeline:
    """The slice of DeepfakeDetectionPipeline.__init__ (pipeline.py:70-93) the hot path needs."""

    def __init__(self, config, feature_extractor=None):
        import torch
        self.config = config
        self.device = torch.device(config.device)
        self.audio_segmenter = AudioSegmenter(config)                                   # pipeline.py:78
        self.feature_extractor = feature_extractor or build_feature_extractor(config)   # pipeline.py:81
        config.feature_dim = self.feature_extractor.feature_dim                         # pipeline.py:85
        self.tpp = TemporalPyramidPooling(config)                                       # pipeline.py:89
        self.vector_db = VectorDatabase(config)                                         # pipeline.py:90
        self.training_file_ids = set()

    # ---- segment + embed --------------------------------------------------------------------------------
    def embed_waves(self, wave, clip_offsets: Sequence[int]):
        """Device-resident form: `wave` holds the clips back to back on the GPU -> [B, D]."""
        fe = self.feature_extractor
        if hasattr(fe, "embed_clips"):
            return fe.embed_clips(wave, clip_offsets)           # one fused pass (csrc/embed.hip)
        # protocol-only extractor: the reference's composition, stage by stage (pipeline.py:402-414)
        import torch
        host = wave.detach().cpu().numpy()
        pooled = []
        for b in range(len(clip_offsets) - 1):
            segs = self.audio_segmenter.segment_audio(host[clip_offsets[b]:clip_offsets[b + 1]])
            feats = [t.to(self.device) for t in fe.extract_features(segs)]
            pooled.append(torch.mean(self.tpp.pool_features_batch(feats), dim=0))
        return torch.stack(pooled)

    def process_audio_batch(self, audio_paths: List[str], audio_dataset) -> "torch.Tensor":
        """pipeline.py:392-414.  `audio_dataset` only needs `.load_audio(path) -> 1-D float array`."""
        import torch
        waves = []
        for path in audio_paths:
            wav = audio_dataset.load_audio(path)
            if wav is None:
                raise RuntimeError(f"Failed to load '{path}'")       # pipeline.py:398-399
            wav = np.asarray(wav, np.float32)
            if wav.ndim > 1:
                raise ValueError("Expected 1D audio array")          # segmenter.py:18-19
            waves.append(wav)
        offs = np.zeros(len(waves) + 1, np.int64)
        np.cumsum([len(w) for w in waves], out=offs[1:])
        flat = torch.from_numpy(np.concatenate(waves) if waves else np.zeros(0, np.float32)).to(self.device)
        return self.embed_waves(flat, offs)                          # [batch, tpp_output_dim]

    # ---- retrieve -----------------------------------------------------------------------------------------
    def retrieve_similar_vectors(self, query_vectors, query_paths: Optional[List[str]] = None, exclude_self: bool = True,
                                 return_info: bool = False, return_distances: bool = False):
        """pipeline.py:449-532 with the same four return arities.  The search stays on the device; only the
        [B, K+10] ids/distances come to the host for the basename exclusion (:491-509), and the kept neighbour
        rows are gathered on the device in one launch instead of one index.reconstruct call each (:503)."""
        import torch
        B = query_vectors.shape[0]
        K = int(self.config.top_k)
        D = self.tpp.get_output_dim()

        def pack(vec, lbl, paths, dist):
            if return_info and return_distances:
                return vec, lbl, paths, dist
            if return_info:
                return vec, lbl, paths
            if return_distances:
                return vec, lbl, dist
            return vec, lbl

        index = self.vector_db.index
        if index is None or getattr(index, "ntotal", 0) == 0:                   # pipeline.py:465-476
            return pack(torch.zeros(B, K, D, device=self.device), torch.zeros(B, K, device=self.device),
                        [[""] * K for _ in range(B)], torch.full((B, K), float("nan"), device=self.device))

        exclude_ids = set()
        if exclude_self and query_paths is not None:
            exclude_ids = {os.path.basename(p) for p in query_paths}            # pipeline.py:463
        k_search = K + (10 if exclude_self else 0)                              # pipeline.py:478
        q = query_vectors.detach().to(self.device, torch.float32)
        try:
            dists_t, idxs_t = self.vector_db.search_batch(q, k=k_search)
            dists, idxs = dists_t.cpu().numpy(), idxs_t.cpu().numpy()
        except Exception:                                                       # pipeline.py:481-483
            dists = np.zeros((B, 0), dtype=np.float32)
            idxs = np.zeros((B, 0), dtype=np.int64)

        paths_db, labels_db = self.vector_db.vector_paths, self.vector_db.vector_labels
        train_ids = getattr(self, "training_file_ids", set())
        chosen = np.full((B, K), -1, np.int64)
        lbl = np.zeros((B, K), np.float32)
        dist = np.full((B, K), np.nan, np.float32)
        all_paths = []
        for b in range(B):
            n, row_paths = 0, []
            for ii, dd in zip(idxs[b], dists[b]):
                ii = int(ii)
                if ii < 0:          # unfilled slot; the reference would wrap to vector_paths[-1] here (pipeline.py:495)
                    continue
                fname = os.path.basename(paths_db[ii])
                if exclude_self:
                    if query_paths is not None:
                        if fname in exclude_ids:
                            continue
                    elif fname in train_ids:
                        continue
                chosen[b, n], lbl[b, n], dist[b, n] = ii, labels_db[ii], float(dd)
                row_paths.append(paths_db[ii])
                n += 1
                if n == K:
                    break
            all_paths.append(row_paths + [""] * (K - n))                        # pipeline.py:511-515
        vec_tensor = index.reconstruct_batch(torch.from_numpy(chosen).to(self.device))   # [B,K,D], zeros where id == -1
        return pack(vec_tensor, torch.from_numpy(lbl).to(self.device), all_paths, torch.from_numpy(dist).to(self.device))
