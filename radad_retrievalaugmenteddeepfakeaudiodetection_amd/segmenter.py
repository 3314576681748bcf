"""AudioSegmenter -- same call surface as the reference's segmenter.py:5-49.

Segmentation is index arithmetic, so on the GPU it is not a kernel: it is the (clip, segment) ->
sample-range plan the front-end kernel loads through (csrc/embed.hip build_plan).  This class keeps the
reference's host-side API for callers that still want the list of arrays, and exposes `plan()` -- the CSR
form of the same rule -- for the batched device path.
"""
from typing import List

import numpy as np


class AudioSegmenter:
    """Fixed-length overlapped windows (segmenter.py:8-13)."""

    def __init__(self, config):
        self.config = config
        self.sample_rate = config.sample_rate
        self.segment_length = int(config.segment_length * config.sample_rate)
        self.segment_overlap = config.segment_overlap
        self.hop_length = int(self.segment_length * (1 - self.segment_overlap))

    def num_segments(self, total_samples: int) -> int:
        """segmenter.py:25 -- max(1, (N-L)//hop + 1); samples past the last full window are dropped."""
        return max(1, (int(total_samples) - self.segment_length) // self.hop_length + 1)

    def segment_audio(self, audio: np.ndarray) -> List[np.ndarray]:
        """segmenter.py:15-39: views of `audio` when no padding is needed; a clip shorter than one segment
        is zero-padded (and, as in the reference, promoted to float64 by np.zeros' default dtype)."""
        if len(audio.shape) > 1:
            raise ValueError("Expected 1D audio array")
        total = len(audio)
        segments = []
        for i in range(self.num_segments(total)):
            start = i * self.hop_length
            seg = audio[start:min(start + self.segment_length, total)]
            if len(seg) < self.segment_length:
                seg = np.concatenate([seg, np.zeros(self.segment_length - len(seg))])
            segments.append(seg)
        return segments

    def plan(self, clip_lengths):
        """CSR plan for a batch of clips stored back to back.
        Returns (clip_offsets[B+1], seg_start[S] absolute, seg_valid[S], clip_seg_offsets[B+1])."""
        clip_lengths = np.asarray(clip_lengths, np.int64)
        clip_offsets = np.zeros(len(clip_lengths) + 1, np.int64)
        np.cumsum(clip_lengths, out=clip_offsets[1:])
        seg_start, seg_valid, clip_seg = [], [], [0]
        for b, n in enumerate(clip_lengths):
            ns = self.num_segments(int(n))
            for i in range(ns):
                st = i * self.hop_length
                seg_start.append(int(clip_offsets[b]) + st)
                seg_valid.append(max(0, min(self.segment_length, int(n) - st)))
            clip_seg.append(clip_seg[-1] + ns)
        return (clip_offsets, np.asarray(seg_start, np.int64), np.asarray(seg_valid, np.int32),
                np.asarray(clip_seg, np.int64))
