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

    # ---- build the store ------------------------------------------------------------------------------------
    def build_vector_database(self, batches, audio_dataset, save: bool = True):
        """pipeline.py:416-447 without the host round trip: every batch is embedded on the GPU and appended to the
        HBM-resident store as a device tensor (the reference does .cpu().numpy() -> np.vstack -> index.add).
        `batches` yields dicts like the reference's DataLoader does: {'path': [...], 'label': [...], 'metadata': ...}."""
        self.training_file_ids.clear()
        n = 0
        for batch in batches:
            paths = list(batch["path"])
            vecs = self.process_audio_batch(paths, audio_dataset)            # [b, D] on the device
            labels = [int(l) if not hasattr(l, "item") else int(l.item()) for l in batch["label"]]
            speakers = self._extract_batch_speakers(batch.get("metadata"), len(paths))
            self.training_file_ids.update(os.path.basename(p) for p in paths)     # pipeline.py:437-438
            self.vector_db.add_vectors(vecs, paths, labels, {"speaker_id": speakers})
            n += len(paths)
        if save:
            self.vector_db.save()                                             # pipeline.py:446
        return n

    @staticmethod
    def _extract_batch_speakers(metas, batch_size):
        """the speaker_id column the reference keeps beside each vector (pipeline.py:433-443)"""
        if isinstance(metas, dict) and "speaker_id" in metas:
            sp = list(metas["speaker_id"])
            return (sp + ["unknown"] * batch_size)[:batch_size]
        if isinstance(metas, (list, tuple)) and len(metas) == batch_size:
            return [m.get("speaker_id", "unknown") if isinstance(m, dict) else "unknown" for m in metas]
        return ["unknown"] * batch_size

    # ---- retrieve -----------------------------------------------------------------------------------------
    def retrieve_similar_vectors(self, query_vectors, query_paths: Optional[List[str]] = None, exclude_self: bool = True,
                                 return_info: bool = False, return_distances: bool = False):
        """pipeline.py:449-532 with the same four return arities.  Search, basename exclusion (:491-509, through
        per-row tags), compaction to K, label and neighbour-row gathers (:503) all stay on the device; nothing is
        copied to the host unless `return_info` asks for the path strings.  Unfilled search slots (id -1) are skipped
        instead of wrapping to vector_paths[-1] as the reference does (:495)."""
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
        except Exception:                                                       # pipeline.py:481-483: swallow, return padding
            dists_t = idxs_t = None

        # exclusion + compaction + gathers on the device (csrc/knn.hip k_filter_topk, k_gather_rows): basenames are
        # compared through their 63-bit tags, so the only per-row host work left is building the optional path lists
        from .vector_database import path_tag
        excl = None
        if exclude_self:
            names = exclude_ids if query_paths is not None else getattr(self, "training_file_ids", set())
            if names:
                excl = torch.tensor(sorted({path_tag(n) for n in names}), dtype=torch.int64, device=self.device)
        if idxs_t is None:
            idxs_t = torch.zeros((B, 0), dtype=torch.int64, device=self.device)
            dists_t = torch.zeros((B, 0), dtype=torch.float32, device=self.device)
        dist_t, chosen_t = self.vector_db.filter_hits(dists_t, idxs_t, K, excl)
        valid = chosen_t >= 0
        rows_t = (chosen_t - index.id_base).clamp(min=0)
        lbl_t = torch.where(valid, self.vector_db.labels_device()[rows_t], torch.zeros((), device=self.device))   # 0.0 pad (:513)
        all_paths = None
        if return_info:
            paths_db = self.vector_db.vector_paths
            all_paths = [[paths_db[i - index.id_base] if i >= 0 else "" for i in row] for row in chosen_t.cpu().tolist()]
        vec_tensor = index.reconstruct_batch(chosen_t)                          # [B,K,D], zeros where id == -1 (pipeline.py:512)
        return pack(vec_tensor, lbl_t, all_paths, dist_t)

    # ---- the online path ------------------------------------------------------------------------------------
    def predict(self, audio_path: str, audio_dataset, radad_model, threshold: float = 0.5):
        """pipeline.py:1038-1103 for ONE clip: segment -> embed (:1047) -> retrieve with the clip's own basename excluded (:1049-1051)
        -> when that leaves nothing, retrieve again WITHOUT the exclusion (:1052-1055) -> RADADModel (:1076) -> sigmoid -> the
        reference's result dict (:1096-1103).  `audio_dataset` needs `.load_audio(path)` (the reference builds an AudioDataset
        around librosa, :1043); `radad_model` is this build's RADADModel (inference forward, csrc/proj.hip) or any module with the
        reference's forward(retrieved_vectors, tpp_vector).  Everything between the loaded samples and the logit stays on the GPU."""
        import logging
        import torch
        index = self.vector_db.index
        if index is None or getattr(index, "ntotal", 0) == 0:
            logging.warning("Vector DB is empty or not loaded. Retrieval will return zero neighbors.")      # :1039-1040
        radad_model.eval()                                                                                  # :1042
        with torch.no_grad():
            tpp_vec = self.process_audio_batch([audio_path], audio_dataset)                                 # [1, D]
            vecs, lbls, npaths = self.retrieve_similar_vectors(tpp_vec, query_paths=[audio_path], exclude_self=True, return_info=True)
            if torch.count_nonzero(vecs) == 0:                                                              # :1052
                vecs, lbls, npaths = self.retrieve_similar_vectors(tpp_vec, query_paths=[audio_path], exclude_self=False,
                                                                   return_info=True)
            if vecs.ndim == 2:                                                                              # :1057-1060
                vecs = vecs.unsqueeze(1)
            elif vecs.ndim == 1:
                vecs = vecs.unsqueeze(0).unsqueeze(1)
            logits = radad_model(vecs, tpp_vec)                                                             # :1076
            if logits.ndim == 1:
                logits = logits.unsqueeze(-1)
            flat = logits.detach().float().view(-1)
            prob = torch.sigmoid(flat).mean().item()                                                        # :1082
            pred = "spoof" if prob >= float(threshold) else "bona-fide"                                     # :1083
            logit = flat.mean().item()
            neigh_labels = [int(x) for x in lbls.squeeze(0).detach().cpu().tolist()] if isinstance(lbls, torch.Tensor) else \
                (list(lbls[0]) if isinstance(lbls, list) and len(lbls) else [])                             # :1087-1090
            neigh_paths = npaths[0] if isinstance(npaths, list) and len(npaths) else []
            retrieved = [{"file": os.path.basename(pth) if pth else "", "path": pth, "label": lab}
                         for lab, pth in zip(neigh_labels, neigh_paths)]
            return {"prediction": pred, "probability_spoof": float(prob), "logit": float(logit), "retrieved_labels": neigh_labels,
                    "retrieved_files": [r["file"] for r in retrieved], "retrieved": retrieved}
