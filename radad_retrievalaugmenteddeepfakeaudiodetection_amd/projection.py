"""ProjectionLayer -- inference forward of the reference's projection.py:8-160 on the GPU.

Parameter names and shapes equal the reference's fused layout (`fuse_attention_ops=True`, config.py:79), so a
state_dict trained with the reference loads unchanged (the unfused layout of projection.py:32-47 -- the same layers inside two
nn.Sequential -- is renamed on load, and written back under those names by a layer configured with fuse_attention_ops=False):
    attention_score [H,D]  attention_final [1,H]  cst_hidden [H,D]  cst_output [D,H]
    weight_sum [H,D]  normalization (LayerNorm H, eps 1e-6)  unified_embedding [O,H]
forward([B,K,D]) -> [B,O] runs csrc/proj.hip (eval semantics: dropout is the identity).  Training stays in
the reference; a forward that needs gradients raises instead of silently falling back.
"""
import ctypes as C

from . import _lib

import torch
import torch.nn as nn


class ProjectionLayer(nn.Module):
    def __init__(self, config, input_dim: int):
        super().__init__()
        self.config = config
        self.input_dim = int(input_dim)
        self.device = torch.device(getattr(config, "device", "cuda"))
        self.hidden_dim = int(config.projection_hidden_dim)
        self.output_dim = int(config.projection_output_dim)
        D, H, O = self.input_dim, self.hidden_dim, self.output_dim
        self.attention_score = nn.Linear(D, H)          # projection.py:29
        self.attention_final = nn.Linear(H, 1)          # :30
        self.cst_hidden = nn.Linear(D, H)               # :40
        self.cst_output = nn.Linear(H, D)               # :41
        self.weight_sum = nn.Linear(D, H)               # :50
        self.normalization = nn.LayerNorm(H, eps=1e-6)  # :51
        self.unified_embedding = nn.Linear(H, O)        # :52
        for m in self.modules():                        # :58-66
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        self.to(self.device)
        # The reference has a second parameter layout (`fuse_attention_ops=False`, projection.py:32-47): the same four Linear layers
        # inside two nn.Sequential -- attention_score.{0,2}, cst_attention.{0,2}.  The arithmetic is identical (tanh / relu between
        # them, projection.py:69-76 vs :78-90), so such a state_dict loads here by renaming, and a layer configured that way
        # writes those names back out (state_dict() round-trips with the reference module of the same configuration).
        self.fuse_operations = bool(getattr(config, "fuse_attention_ops", True))
        self._register_load_state_dict_pre_hook(self._rename_unfused_keys)
        if not self.fuse_operations:
            self._register_state_dict_hook(self._emit_unfused_keys)
        self._ws = None
        self._fold = None        # (key, w54t [H,H], b54 [H]): W5 (W4 h + b4) + b5 folded once per set of weights

    _UNFUSED = (("attention_score.0.", "attention_score."), ("attention_score.2.", "attention_final."),
                ("cst_attention.0.", "cst_hidden."), ("cst_attention.2.", "cst_output."))

    @classmethod
    def _rename_unfused_keys(cls, state_dict, prefix, *args):
        for old, new in cls._UNFUSED:
            for leaf in ("weight", "bias"):
                k = prefix + old + leaf
                if k in state_dict:
                    state_dict[prefix + new + leaf] = state_dict.pop(k)

    @classmethod
    def _emit_unfused_keys(cls, module, state_dict, prefix, local_metadata):
        for old, new in reversed(cls._UNFUSED):          # (attention_final -> attention_score.2 before attention_score -> .0)
            for leaf in ("weight", "bias"):
                k = prefix + new + leaf
                if k in state_dict:
                    state_dict[prefix + old + leaf] = state_dict.pop(k)
        return state_dict

    def forward(self, input_embeddings: torch.Tensor) -> torch.Tensor:
        """projection.py:108-117 (eval path).  input [B, top_k, D] -> [B, output_dim]."""
        if input_embeddings.device != self.device:
            input_embeddings = input_embeddings.to(self.device)
        if torch.is_grad_enabled() and (input_embeddings.requires_grad or any(p.requires_grad for p in self.parameters())) \
                and self.training:
            raise RuntimeError("ProjectionLayer here is inference-only (HIP forward); call .eval() / torch.no_grad(), "
                               "or train with the reference module and load its state_dict")
        _lib.require_cuda(input_embeddings, "input_embeddings")
        x = input_embeddings.detach().contiguous().float()
        if x.dim() != 3 or x.shape[2] != self.input_dim:
            raise ValueError(f"expected [B, K, {self.input_dim}], got {tuple(x.shape)}")
        B, K, D = x.shape
        lib = _lib.load()
        need = lib.radad_projection_workspace_bytes(B, K, D, self.hidden_dim, self.output_dim)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        w, keep = self._weights(x.device)
        out = torch.empty((B, self.output_dim), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.radad_projection_forward(C.byref(w), x.data_ptr(), B, K, D, self.hidden_dim, self.output_dim,
                                                    out.data_ptr(), self._ws.data_ptr(), int(self._ws.numel()),
                                                    x.device.index, _lib.stream_ptr(x.device)), "radad_projection_forward")
        return out

    def _weights(self, device):
        """C struct of device pointers (+ the tensors that must outlive the launch) with the cached W5*W4 fold."""
        lib = _lib.load()
        w = _lib.ProjWeights()
        keep = []   # contiguous fp32 views stay alive until the launch is enqueued

        def ptr(t):
            t = t.detach().contiguous().float()
            keep.append(t)
            return t.data_ptr()
        w.w1, w.b1 = ptr(self.attention_score.weight), ptr(self.attention_score.bias)
        w.w2, w.b2 = ptr(self.attention_final.weight), ptr(self.attention_final.bias)
        w.w3, w.b3 = ptr(self.cst_hidden.weight), ptr(self.cst_hidden.bias)
        w.w4, w.b4 = ptr(self.cst_output.weight), ptr(self.cst_output.bias)
        w.w5, w.b5 = ptr(self.weight_sum.weight), ptr(self.weight_sum.bias)
        w.ln_g, w.ln_b = ptr(self.normalization.weight), ptr(self.normalization.bias)
        w.w6, w.b6 = ptr(self.unified_embedding.weight), ptr(self.unified_embedding.bias)
        # in-place updates (load_state_dict, optimizer steps) bump _version; a new Parameter changes data_ptr
        src = (self.cst_output.weight, self.cst_output.bias, self.weight_sum.weight, self.weight_sum.bias)
        key = tuple((t.data_ptr(), t._version) for t in src) + (str(device),)
        if self._fold is None or self._fold[0] != key:
            H = self.hidden_dim
            w54t = torch.empty((H, H), device=device, dtype=torch.float32)
            b54 = torch.empty((H,), device=device, dtype=torch.float32)
            with torch.cuda.device(device):
                _lib.check(lib.radad_projection_fold(C.byref(w), self.input_dim, H, w54t.data_ptr(), b54.data_ptr(),
                                                     device.index, _lib.stream_ptr(device)), "radad_projection_fold")
            self._fold = (key, w54t, b54)
        w.w54t, w.b54 = self._fold[1].data_ptr(), self._fold[2].data_ptr()
        return w, keep

    def forward_batch(self, input_embeddings_list: list) -> torch.Tensor:
        """projection.py:119-122."""
        return self.forward(torch.stack(input_embeddings_list).to(self.device))

    @torch.no_grad()
    def get_attention_weights(self, input_embeddings: torch.Tensor) -> torch.Tensor:
        """projection.py:125-130: softmax over the K neighbours of  W2 tanh(W1 x + b1) + b2  -> [B, K, 1].
        The two Linear layers run on radad_linear_forward (split-K MFMA GEMM with the tanh fused); the softmax over the
        K (= 5) scores per row is a torch op on the [B, K] result."""
        x = _lib.require_cuda(input_embeddings.to(self.device), "input_embeddings").detach().contiguous().float()
        if x.dim() != 3 or x.shape[2] != self.input_dim:
            raise ValueError(f"expected [B, K, {self.input_dim}], got {tuple(x.shape)}")
        B, K, D = x.shape
        lib = _lib.load()
        H = self.hidden_dim

        def linear(inp, lin, act, n_out, n_in):
            rows = inp.shape[0]
            w, b = lin.weight.detach().contiguous().float(), lin.bias.detach().contiguous().float()
            out = torch.empty((rows, n_out), device=inp.device, dtype=torch.float32)
            need = lib.radad_linear_workspace_bytes(rows, n_out, n_in)
            ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=inp.device)
            with torch.cuda.device(inp.device):
                _lib.check(lib.radad_linear_forward(inp.data_ptr(), n_in, w.data_ptr(), n_in, b.data_ptr(), act, rows, n_out, n_in,
                                                    out.data_ptr(), n_out, ws.data_ptr(), int(ws.numel()), inp.device.index,
                                                    _lib.stream_ptr(inp.device)), "radad_linear_forward")
            return out
        hidden = linear(x.reshape(B * K, D), self.attention_score, 1, H, D)          # tanh fused
        scores = linear(hidden, self.attention_final, 0, 1, H).reshape(B, K, 1)
        return torch.softmax(scores, dim=1)

    def memory_efficient_forward(self, input_embeddings: torch.Tensor, chunk_size: int = 32) -> torch.Tensor:
        """projection.py:132-138: the forward in chunks of `chunk_size` rows (kept for callers; the HIP forward's workspace
        is a few MB at any batch size, so chunking buys nothing here)."""
        if input_embeddings.size(0) <= chunk_size:
            return self.forward(input_embeddings)
        return torch.cat([self.forward(input_embeddings[i:i + chunk_size]) for i in range(0, input_embeddings.size(0), chunk_size)], dim=0)

    def profile_performance(self, input_shape: tuple, num_iterations: int = 100):
        """projection.py:140-153: average wall time of a forward on random input (prints, and returns ms per iteration)."""
        import time
        dummy = torch.randn(input_shape, device=self.device)
        was_training = self.training
        self.eval()
        for _ in range(10):
            self.forward(dummy)
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        for _ in range(num_iterations):
            self.forward(dummy)
        torch.cuda.synchronize(self.device)
        ms = (time.time() - t0) / max(1, num_iterations) * 1000
        self.train(was_training)
        print(f"Avg forward: {ms:.2f} ms/iter")
        return ms

    def get_flops(self, input_shape: tuple) -> int:
        """projection.py:155-160."""
        B, K, D = input_shape
        flops = B * K * (D * self.hidden_dim + self.hidden_dim)
        flops += B * K * (D * self.hidden_dim + self.hidden_dim * D)
        flops += B * (D * self.hidden_dim + self.hidden_dim * self.output_dim)
        return flops
