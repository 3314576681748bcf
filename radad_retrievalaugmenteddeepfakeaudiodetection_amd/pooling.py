"""TemporalPyramidPooling -- same call surface as the reference's pooling.py:4-122, on the GPU.

pool_features([T,F]) -> [sum(levels)*F]: per level, adaptive max/avg over time into `level` bins
(bin i = rows [floor(i*T/l), ceil((i+1)*T/l)), the torch adaptive-pool rule used at pooling.py:76,79), flattened
bin-major / feature-minor (:84), levels concatenated in order (:103).  In the fused embedding path this stage
runs inside k_proj_pool; this class is the stand-alone operator (csrc/embed.hip k_tpp).
"""
from typing import List

import numpy as np

from . import _lib


class TemporalPyramidPooling:
    def __init__(self, config):
        import torch
        self.config = config
        self.levels = list(config.tpp_levels)
        self.pooling_type = config.tpp_pooling_type
        self.device = torch.device(config.device)
        self._levels_arr = np.ascontiguousarray(np.asarray(self.levels, np.int32))

    def _mode(self):
        if self.pooling_type == "max":
            return _lib.POOL_MAX
        if self.pooling_type == "avg":
            return _lib.POOL_AVG
        raise ValueError(f"Unsupported pooling type: {self.pooling_type}")      # pooling.py:81

    def _pool(self, feats, row_offsets):
        import torch
        lib = _lib.load()
        F = feats.shape[-1]
        n = len(row_offsets) - 1
        out = torch.empty((n, sum(self.levels) * F), device=feats.device, dtype=torch.float32)
        offs = np.ascontiguousarray(np.asarray(row_offsets, np.int64))
        with torch.cuda.device(feats.device):
            _lib.check(lib.radad_tpp_forward(feats.data_ptr(), offs.ctypes.data_as(_lib.c_i64p), n, F,
                                             self._levels_arr.ctypes.data_as(_lib.c_i32p), len(self.levels), self._mode(),
                                             out.data_ptr(), feats.device.index, _lib.stream_ptr(feats.device)),
                       "radad_tpp_forward")
        return out

    def pool_features(self, features):
        """pooling.py:88-103.  features: [T, F] tensor (moved to config.device like the reference does, :90-91)."""
        import torch
        if features.device != self.device:
            features = features.to(self.device)
        _lib.require_cuda(features, "features")
        if features.dim() != 2:
            raise ValueError("expected features of shape [sequence_length, feature_dim]")
        f = features.contiguous().float()
        return self._pool(f, [0, f.shape[0]])[0]

    def pool_features_batch(self, features_batch: List):
        """pooling.py:105-117: list of [T_i, F] -> [B, sum(levels)*F] (one launch for the whole list)."""
        import torch
        if not features_batch:
            return torch.empty(0, device=self.device)
        fs = [f.to(self.device).contiguous().float() for f in features_batch]
        offs = np.zeros(len(fs) + 1, np.int64)
        np.cumsum([f.shape[0] for f in fs], out=offs[1:])
        return self._pool(torch.cat(fs, dim=0), offs)

    def get_output_dim(self) -> int:
        """pooling.py:119-122."""
        return sum(self.levels) * self.config.feature_dim
