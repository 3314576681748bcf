"""MI355X-native implementation of RADAD's segment -> embed -> retrieve hot path.

Host-side mirror of the reference's call surface (AudioSegmenter, the extractor protocol,
TemporalPyramidPooling, VectorDatabase(.index), ProjectionLayer, and the two hot pipeline methods) over the
C ABI of libradad_hip.so (include/radad_hip.h, csrc/*.hip).  There is no CPU fallback: without the
shared library (or without a ROCm device) the operators raise.
"""
from .config import Config
from .segmenter import AudioSegmenter
from .pooling import TemporalPyramidPooling
from .vector_database import HipFlatIndex, HipIVFFlatIndex, VectorDatabase
from .feature_extractor import (MelProjectionFeatureExtractor, Wav2Vec2FeatureExtractor, WavLMFeatureExtractor,
                                WhisperFeatureExtractor, build_feature_extractor)
from .pipeline import HotPathPipeline
from .projection import ProjectionLayer
from .radad_model import DetectionModel, RADADModel
from .sharded import ReplicatedSearch, ShardedSearch, shard_bounds

__all__ = ["Config", "AudioSegmenter", "TemporalPyramidPooling", "HipFlatIndex", "HipIVFFlatIndex", "VectorDatabase",
           "MelProjectionFeatureExtractor", "Wav2Vec2FeatureExtractor", "WhisperFeatureExtractor", "WavLMFeatureExtractor",
           "build_feature_extractor", "HotPathPipeline", "ProjectionLayer", "DetectionModel", "RADADModel",
           "ShardedSearch", "ReplicatedSearch", "shard_bounds"]
