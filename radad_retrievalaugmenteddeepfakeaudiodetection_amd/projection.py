"""ProjectionLayer -- inference forward of the reference's projection.py:8-160 on the GPU.

Parameter names and shapes equal the reference's fused layout (`fuse_attention_ops=True`, config.py:79), so a
state_dict trained with the reference loads unchanged:
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
        self._ws = None
        self._fold = None        # (key, w54t [H,H], b54 [H]): W5 (W4 h + b4) + b5 folded once per set of weights

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

    def get_flops(self, input_shape: tuple) -> int:
        """projection.py:155-160."""
        B, K, D = input_shape
        flops = B * K * (D * self.hidden_dim + self.hidden_dim)
        flops += B * K * (D * self.hidden_dim + self.hidden_dim * D)
        flops += B * (D * self.hidden_dim + self.hidden_dim * self.output_dim)
        return flops
