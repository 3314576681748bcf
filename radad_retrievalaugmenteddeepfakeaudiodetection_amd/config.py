"""Config -- the attribute bag the hot path reads (mirror of the reference's config.py:21-115).

Only the attributes the segment -> embed -> retrieve path touches are kept, with the reference's names
and defaults (config.py:37-39 audio, :48-49 pyramid, :52 index type, :56 top_k, :59-60 projection dims,
:74-76 vector-db knobs), plus the knobs this build adds, read the same `getattr(config, name, default)`
way the reference reads its optional ones.  No heavy imports at module top (the reference's config.py
imports torchaudio/faiss/librosa there, config.py:1-16).
"""
import os


class Config:
    def __init__(self):
        # data paths (config.py:23-27)
        self.data_root = os.environ.get("DATA_ROOT", "/tmp/radad_data")
        self.vector_db_path = os.path.join(self.data_root, "vector_db")
        # audio processing (config.py:37-39)
        self.sample_rate = 16000
        self.segment_length = 2.0
        self.segment_overlap = 0.5
        # temporal pyramid pooling (config.py:48-49)
        self.tpp_levels = [1, 2, 4]
        self.tpp_pooling_type = "max"
        # vector database / retrieval (config.py:52-56, 74-76)
        self.vector_db_index_type = "L2"
        self.vector_db_nprobe = 32
        self.top_k = 5
        self.use_float16 = False
        self.vector_add_batch_size = 10000
        # projection layer (config.py:59-60, 80)
        self.projection_hidden_dim = 256
        self.projection_output_dim = 128
        self.projection_dropout = 0.1
        self.use_mixed_precision = False
        self.use_gradient_checkpointing = False
        self.fuse_attention_ops = True
        # detection head (config.py:63, 82-86; the later assignment of detection_dropout wins there too)
        self.detection_hidden_dims = [64, 32]
        self.detection_dropout = 0.1
        self.use_batch_norm = True
        self.use_layer_norm = False
        # device (config.py:89)
        import torch
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        # extractor selection (config.py:92).  "melproj" is this build's native extractor.
        self.feature_extractor_type = "melproj"
        # the reference's three encoders (config.py:42-45).  The reference fetches these NAMES from the hub; here each must be a local
        # directory in save_pretrained layout (nothing is downloaded): feature_extractor.py's adapters run the front-end in HIP
        self.wav2vec2_model_name = "facebook/wav2vec2-base-960h"
        self.whisper_model_name = "openai/whisper-base"
        self.wavlm_model_name = "microsoft/wavlm-base"
        self.wav2vec2_layers_to_use = [-4, -3, -2, -1]
        # ---- knobs added by this build -------------------------------------------------------------
        self.feature_dim = 512              # F of the frame projection (injected by main.py:66 in the reference)
        self.melproj_normalize = True       # per-segment zero-mean/unit-variance before the spectrogram
        self.melproj_padded_samples = 0     # 0: spectrogram of the segment itself; 480000: HF's 30 s padding
        self.melproj_seed = 20251003        # seed of the synthetic projection weights when no file is given
        self.melproj_weights_path = None    # optional .npz with 'w' [80,F] and 'b' [F]
        # kernel choices (speed only, results unchanged; A/B measurements and parity tests): which log-mel kernel ...
        self.melproj_share_frames = True    # frames that overlapping segments share are transformed once per clip
        self.melproj_logmel_fft = True      # ... as a radix FFT on the vector ALU (False: DFT-as-GEMM on the f16 matrix pipe)
        self.melproj_logmel_f32 = False     # True: the folded DFT on the fp32 matrix pipe (per segment)
        # ... and which scan (None = the library's default): radad_knn_set_option
        self.knn_hi_plane = None            # False: every batch scans on the fp32 kernels (no f16 plane)
        self.knn_centre = None              # -1 / 0 / 1: decide per store / never / always centre the f16 plane
        self.knn_smallq_hi = None           # False: batches of <= 16 queries stream the fp32 rows
        self.knn_wide_min_q = None          # smallest batch on the 256-query tile scan (default 17)
        self.knn_dense = None               # 0: fp32 stores of a few thousand rows stay on the register-list kernels
        self.knn_live_floor = None          # 1: the tile scan runs ONE launch that raises its floors inside it (default: one launch per phase)

    def update(self, **kwargs):
        """config.py:109-115."""
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)
            else:
                raise ValueError(f"Invalid configuration parameter: {key}")
