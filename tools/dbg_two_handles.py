import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
B = 1024
cfg = R.Config(); cfg.update(device=dev, feature_dim=512, tpp_levels=[1], tpp_pooling_type="max")
fe, fe2 = R.MelProjectionFeatureExtractor(cfg), R.MelProjectionFeatureExtractor(cfg)
wave = torch.empty(B * 64000, device=dev)
_lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, 64000, 1234, 0, _lib.stream_ptr(dev)))
H = B // 2
w1, w2 = wave[:H * 64000], wave[H * 64000:]
o_h = np.arange(H + 1, dtype=np.int64) * 64000
def t(fn, n=40):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one handle, half batch          %.4f ms" % t(lambda: fe.embed_clips(w1, o_h)))
print("one handle, half batch x2       %.4f ms" % t(lambda: (fe.embed_clips(w1, o_h), fe.embed_clips(w1, o_h))))
print("one handle, two different halves %.4f ms" % t(lambda: (fe.embed_clips(w1, o_h), fe.embed_clips(w2, o_h))))
print("two handles alternating          %.4f ms" % t(lambda: (fe.embed_clips(w1, o_h), fe2.embed_clips(w2, o_h))))
print("one handle, whole batch          %.4f ms" % t(lambda: fe.embed_clips(wave, np.arange(B + 1, dtype=np.int64) * 64000)))
